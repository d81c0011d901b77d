"""Hamiltonian Monte Carlo drivers for ``ExactGP.fit()`` (host side of gpax/gp.py:512-534).

The reference hands ``_gp_prior`` to NumPyro's ``NUTS`` inside ``MCMC(num_warmup, num_samples, ...)``;
NumPyro/JAX are not part of this stack, so the sampler lives here.  It only ever sees a callable
``value_and_grad(u) -> (log_density, gradient)`` in unconstrained space -- on the GPU that is ONE call of
``gpx_lml_grad`` (log-marginal likelihood and its closed-form gradient from one bordered factorisation).

* ``nuts``: the No-U-Turn sampler with multinomial trajectory sampling and biased progressive
  sub-tree selection (Hoffman & Gelman 2014 alg. 6; Betancourt 2017), dual-averaging step-size
  adaptation to ``target_accept_prob`` and Stan-style windowed adaptation of a diagonal mass matrix --
  the defaults of NumPyro's ``NUTS`` (``max_tree_depth=10``, ``target_accept_prob=0.8``,
  ``adapt_step_size``, ``adapt_mass_matrix``, ``dense_mass=False``).
* ``hmc``: fixed number of leapfrog steps, same adaptation.

Random streams differ from JAX's threefry: agreement with the reference is statistical.
"""

from __future__ import annotations

import math

import numpy as np


class _DualAveraging:
    """Nesterov dual averaging of log(step size) (Hoffman & Gelman 2014, section 3.2)."""

    def __init__(self, eps0, target, gamma=0.05, t0=10.0, kappa=0.75):
        self.target, self.gamma, self.t0, self.kappa = target, gamma, t0, kappa
        self.restart(eps0)

    def restart(self, eps0):
        self.mu = math.log(10.0 * eps0)
        self.hbar, self.log_eps_bar, self.m = 0.0, 0.0, 0

    def update(self, accept_stat):
        self.m += 1
        m = self.m
        self.hbar = (1 - 1 / (m + self.t0)) * self.hbar + (self.target - accept_stat) / (m + self.t0)
        log_eps = self.mu - math.sqrt(m) / self.gamma * self.hbar
        eta = m ** (-self.kappa)
        self.log_eps_bar = eta * log_eps + (1 - eta) * self.log_eps_bar
        return math.exp(log_eps)

    @property
    def final(self):
        return math.exp(self.log_eps_bar)


class _Welford:
    def __init__(self, dim):
        self.n, self.mean, self.m2 = 0, np.zeros(dim), np.zeros(dim)

    def add(self, x):
        self.n += 1
        d = x - self.mean
        self.mean += d / self.n
        self.m2 += d * (x - self.mean)

    def variance(self):
        """Regularised as in Stan / NumPyro: shrink towards 1e-3 * identity."""
        n = self.n
        var = self.m2 / max(n - 1, 1)
        return (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0))


def _windows(num_warmup):
    """Stan's adaptation schedule: [(start, end)] of the slow (mass-matrix) windows."""
    if num_warmup < 20:
        return []
    init_buf, term_buf, base = 75, 50, 25
    if init_buf + base + term_buf > num_warmup:
        init_buf, term_buf = int(0.15 * num_warmup), int(0.1 * num_warmup)
        base = num_warmup - init_buf - term_buf
    out, start, size = [], init_buf, base
    end_slow = num_warmup - term_buf
    while start < end_slow:
        end = start + size
        if end + 2 * size > end_slow:   # the last window absorbs the remainder
            end = end_slow
        out.append((start, end))
        start, size = end, 2 * size
    return out


def _leapfrog(vg, u, r, grad, eps, minv):
    r = r + 0.5 * eps * grad
    u = u + eps * minv * r
    logp, grad = vg(u)
    r = r + 0.5 * eps * grad
    return u, r, logp, grad


def _energy(logp, r, minv):
    with np.errstate(over="ignore", invalid="ignore"):   # a wild trial step of the step-size search: rejected as inf
        h = -logp + 0.5 * float(r @ (minv * r))
    return h if np.isfinite(h) else np.inf


def _reasonable_step(vg, u, logp, grad, minv, rng, eps=1.0):
    """Heuristic of Hoffman & Gelman alg. 4: double/halve until the one-step acceptance crosses 1/2."""
    r = rng.standard_normal(u.size) / np.sqrt(minv)
    h0 = _energy(logp, r, minv)

    def delta(e):
        _, r1, lp1, _ = _leapfrog(vg, u, r, grad, e, minv)
        h1 = _energy(lp1, r1, minv) if np.isfinite(lp1) else np.inf
        return h0 - h1

    d = delta(eps)
    d = d if np.isfinite(d) else -np.inf
    a = 1.0 if d > math.log(0.5) else -1.0
    for _ in range(60):
        if not (a * d > -a * math.log(2.0)):
            break
        eps *= 2.0 ** a
        d = delta(eps)
        d = d if np.isfinite(d) else -np.inf
    return float(min(max(eps, 1e-8), 1e3))


class _Tree:
    __slots__ = ("u_minus", "r_minus", "g_minus", "u_plus", "r_plus", "g_plus", "u_prop", "lp_prop", "g_prop",
                 "log_w", "turning", "diverging", "sum_accept", "n_steps")


def _uturn(u_minus, u_plus, r_minus, r_plus, minv):
    du = u_plus - u_minus
    return (du @ (minv * r_minus) < 0) or (du @ (minv * r_plus) < 0)


def _build_tree(vg, u, r, grad, direction, depth, eps, h0, minv, rng, max_delta=1000.0):
    if depth == 0:
        u1, r1, lp1, g1 = _leapfrog(vg, u, r, grad, direction * eps, minv)
        h1 = _energy(lp1, r1, minv) if np.isfinite(lp1) else np.inf
        dh = h1 - h0
        if not np.isfinite(dh):
            dh = np.inf
        t = _Tree()
        t.u_minus = t.u_plus = t.u_prop = u1
        t.r_minus = t.r_plus = r1
        t.g_minus = t.g_plus = t.g_prop = g1
        t.lp_prop = lp1
        t.log_w = -dh
        t.turning = False
        t.diverging = dh > max_delta
        t.sum_accept = min(1.0, math.exp(-dh)) if dh > 0 else 1.0
        t.n_steps = 1
        return t
    a = _build_tree(vg, u, r, grad, direction, depth - 1, eps, h0, minv, rng, max_delta)
    if a.turning or a.diverging:
        return a
    if direction > 0:
        b = _build_tree(vg, a.u_plus, a.r_plus, a.g_plus, direction, depth - 1, eps, h0, minv, rng, max_delta)
    else:
        b = _build_tree(vg, a.u_minus, a.r_minus, a.g_minus, direction, depth - 1, eps, h0, minv, rng, max_delta)
    t = _Tree()
    if direction > 0:
        t.u_minus, t.r_minus, t.g_minus = a.u_minus, a.r_minus, a.g_minus
        t.u_plus, t.r_plus, t.g_plus = b.u_plus, b.r_plus, b.g_plus
    else:
        t.u_minus, t.r_minus, t.g_minus = b.u_minus, b.r_minus, b.g_minus
        t.u_plus, t.r_plus, t.g_plus = a.u_plus, a.r_plus, a.g_plus
    t.log_w = np.logaddexp(a.log_w, b.log_w)
    # uniform progressive sampling inside the sub-tree
    if np.isfinite(b.log_w) and math.log(rng.uniform() + 1e-300) < b.log_w - t.log_w:
        t.u_prop, t.lp_prop, t.g_prop = b.u_prop, b.lp_prop, b.g_prop
    else:
        t.u_prop, t.lp_prop, t.g_prop = a.u_prop, a.lp_prop, a.g_prop
    t.diverging = b.diverging
    t.turning = b.turning or _uturn(t.u_minus, t.u_plus, t.r_minus, t.r_plus, minv)
    t.sum_accept = a.sum_accept + b.sum_accept
    t.n_steps = a.n_steps + b.n_steps
    return t


def _nuts_transition(vg, u, logp, grad, eps, minv, rng, max_tree_depth):
    r0 = rng.standard_normal(u.size) / np.sqrt(minv)
    h0 = _energy(logp, r0, minv)
    u_minus = u_plus = u
    r_minus = r_plus = r0
    g_minus = g_plus = grad
    prop = (u, logp, grad)
    log_w = 0.0   # the initial point has weight exp(-(h0 - h0)) = 1
    sum_accept, n_steps, diverged = 0.0, 0, False
    for depth in range(max_tree_depth):
        direction = 1.0 if rng.uniform() < 0.5 else -1.0
        if direction > 0:
            t = _build_tree(vg, u_plus, r_plus, g_plus, direction, depth, eps, h0, minv, rng)
            u_plus, r_plus, g_plus = t.u_plus, t.r_plus, t.g_plus
        else:
            t = _build_tree(vg, u_minus, r_minus, g_minus, direction, depth, eps, h0, minv, rng)
            u_minus, r_minus, g_minus = t.u_minus, t.r_minus, t.g_minus
        sum_accept += t.sum_accept
        n_steps += t.n_steps
        if t.diverging:
            diverged = True
            break
        if t.turning:
            break
        # biased progressive sampling between the old tree and the new sub-tree
        if np.isfinite(t.log_w) and math.log(rng.uniform() + 1e-300) < t.log_w - log_w:
            prop = (t.u_prop, t.lp_prop, t.g_prop)
        log_w = np.logaddexp(log_w, t.log_w)
        if _uturn(u_minus, u_plus, r_minus, r_plus, minv):
            break
    return prop, sum_accept / max(n_steps, 1), n_steps, diverged


def _hmc_transition(vg, u, logp, grad, eps, minv, rng, num_leapfrog):
    eps = eps * rng.uniform(0.8, 1.2)   # jitter: a fixed trajectory length resonates with the target's periods
    r0 = rng.standard_normal(u.size) / np.sqrt(minv)
    h0 = _energy(logp, r0, minv)
    un, rn, lpn, gn = u, r0, logp, grad
    for _ in range(num_leapfrog):
        un, rn, lpn, gn = _leapfrog(vg, un, rn, gn, eps, minv)
        if not np.isfinite(lpn):
            break
    h1 = _energy(lpn, rn, minv) if np.isfinite(lpn) else np.inf
    dh = h1 - h0
    acc = min(1.0, math.exp(-dh)) if (np.isfinite(dh) and dh > 0) else (1.0 if np.isfinite(dh) else 0.0)
    if rng.uniform() < acc:
        return (un, lpn, gn), acc, num_leapfrog, False
    return (u, logp, grad), acc, num_leapfrog, not np.isfinite(dh)


def sample(value_and_grad, u0, num_warmup, num_samples, rng, algorithm="nuts", target_accept_prob=0.8,
           max_tree_depth=10, num_leapfrog=8, adapt_step_size=True, adapt_mass_matrix=True, step_size=None):
    """Run one chain.  Returns ``(draws (num_samples, dim), info)`` with ``info`` holding the adapted
    step size, inverse mass diagonal, mean acceptance statistic, divergences and gradient evaluations."""
    if algorithm not in ("nuts", "hmc"):
        raise ValueError("algorithm must be 'nuts' or 'hmc'")
    n_grad = [0]

    def vg(u):
        n_grad[0] += 1
        lp, g = value_and_grad(u)
        if not np.isfinite(lp) or not np.all(np.isfinite(g)):
            return -np.inf, np.zeros_like(u)
        return float(lp), np.asarray(g, dtype=np.float64)

    u = np.asarray(u0, dtype=np.float64).copy()
    dim = u.size
    logp, grad = vg(u)
    if not np.isfinite(logp):
        raise ValueError("the log-density is not finite at the initial point")
    minv = np.ones(dim)
    eps = float(step_size) if step_size is not None else _reasonable_step(vg, u, logp, grad, minv, rng)
    da = _DualAveraging(eps, target_accept_prob)
    windows = _windows(num_warmup) if adapt_mass_matrix else []
    window_end = {e: s for s, e in windows}
    in_window = np.zeros(num_warmup + 1, dtype=bool)
    for s, e in windows:
        in_window[s:e] = True
    welford = _Welford(dim)
    draws = np.empty((num_samples, dim))
    accs, n_div, n_steps_total = [], 0, 0
    for it in range(num_warmup + num_samples):
        if algorithm == "nuts":
            (u, logp, grad), acc, n_steps, div = _nuts_transition(vg, u, logp, grad, eps, minv, rng, max_tree_depth)
        else:
            (u, logp, grad), acc, n_steps, div = _hmc_transition(vg, u, logp, grad, eps, minv, rng, num_leapfrog)
        n_steps_total += n_steps
        if it < num_warmup:
            if adapt_step_size:
                eps = da.update(acc)
            if in_window[it]:
                welford.add(u)
            if (it + 1) in window_end and welford.n >= 5:
                minv = welford.variance()
                welford = _Welford(dim)
                if adapt_step_size:
                    eps = _reasonable_step(vg, u, logp, grad, minv, rng, eps=da.final)
                    da.restart(eps)
            if it + 1 == num_warmup and adapt_step_size:
                eps = da.final
        else:
            draws[it - num_warmup] = u
            accs.append(acc)
            n_div += int(div)
    info = dict(step_size=eps, inverse_mass=minv, mean_accept=float(np.mean(accs)) if accs else float("nan"),
                divergences=n_div, num_grad_evals=n_grad[0], num_leapfrog_total=n_steps_total)
    return draws, info


class _LockStep:
    """Gathers one gradient request per running chain, evaluates them with ONE batched call and hands the results back:
    the chains (one thread each, plain ``sample`` inside) advance in lock-step, leapfrog step by leapfrog step."""

    def __init__(self, n_chains, batch_fn):
        import threading

        self.cv = threading.Condition()
        self.batch_fn, self.active = batch_fn, n_chains
        self.pending, self.results, self.error = {}, {}, None
        self.calls, self.points = 0, 0

    def _flush(self):
        ids = sorted(self.pending)
        try:
            outs = self.batch_fn([self.pending[i] for i in ids])
            self.calls += 1
            self.points += len(ids)
            self.results.update(zip(ids, outs))
        except BaseException as e:   # every waiting chain must wake up and fail
            self.error = e
        self.pending.clear()
        self.cv.notify_all()

    def call(self, cid, u):
        with self.cv:
            self.pending[cid] = u
            if len(self.pending) == self.active:
                self._flush()
            while cid not in self.results and self.error is None:
                self.cv.wait()
            if self.error is not None:
                raise RuntimeError("a batched gradient call failed") from self.error
            return self.results.pop(cid)

    def done(self):
        with self.cv:
            self.active -= 1
            if self.pending and len(self.pending) == self.active:
                self._flush()


def sample_lockstep(value_and_grad_batch, u0s, num_warmup, num_samples, rngs, **kw):
    """``sample`` for several chains whose gradient evaluations are batched: ``value_and_grad_batch(list of u)`` returns a
    list of ``(log_density, gradient)``.  Each chain runs the ordinary sampler on its own random stream, so its draws are
    those of a sequential run with the same stream.  Returns ``[(draws, info), ...]`` in chain order; every ``info``
    carries ``batched_calls`` / ``batched_points`` (GPU calls issued and points evaluated by them, all chains)."""
    import threading

    n = len(u0s)
    gate = _LockStep(n, value_and_grad_batch)
    out, errs = [None] * n, [None] * n

    def run(cid):
        try:
            out[cid] = sample(lambda u: gate.call(cid, u), u0s[cid], num_warmup, num_samples, rngs[cid], **kw)
        except BaseException as e:
            errs[cid] = e
        finally:
            gate.done()

    threads = [threading.Thread(target=run, args=(i,), name=f"gpx-chain-{i}") for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    for _, info in out:
        info["batched_calls"], info["batched_points"] = gate.calls, gate.points
    return out
