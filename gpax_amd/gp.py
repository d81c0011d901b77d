"""Gaussian-process model objects with gpax's Python surface (``gpax/gp.py``).

Everything numeric on the hot path -- gram matrices, the Cholesky factor, the log-marginal
likelihood, the posterior mean / covariance and the MVN draws -- runs in the HIP library through
``gpax_amd._lib.Core`` (C ABI of ``include/gpx.h``).  What stays in host Python is what is plain
Python in the reference too: transforms, mean functions, hyper-parameter bookkeeping and the
HMC driver that calls the log-marginal.

Deviations from the reference, on purpose:
  * ``use_cholesky`` is accepted but the HIP path ALWAYS uses the Cholesky branch
    (gpax/gp.py:295-301); the reference default is the explicit inverse (:303-310), which agrees
    with it to ~cond(K)*eps (tests/test_oracle.py::test_golden_inverse_branch_gap_recorded).
  * ``ExactGP.fit()`` runs the NUTS sampler of ``gpax_amd.mcmc`` (NumPyro/JAX are not part of this
    stack) on the log-marginal likelihood and its closed-form gradient from ``gpx_lml_grad``;
    ``VariationalInferenceGP.fit()`` maximises the single-sample reparameterised ELBO of a mean-field
    normal guide with Adam.  The random streams differ from JAX's, so agreement with the reference
    is statistical, not bitwise.
"""

from __future__ import annotations

import hashlib
import itertools
import math

import numpy as np

try:   # 13 us for the (4096, 8) training inputs
    from xxhash import xxh3_64_intdigest as _digest
except ImportError:   # pragma: no cover
    def _digest(a):
        """64-bit content hash without xxhash (a checksum such as adler32 is not one: an in-place edit that collides
        would silently reuse stale device data)."""
        return int.from_bytes(hashlib.blake2b(memoryview(np.ascontiguousarray(a)).cast("B"), digest_size=8).digest(), "little")

from gpax_amd import _lib, mcmc, state
from gpax_amd.kernels import Kernel
from gpax_amd.utils.prior_utils import Distribution as _Distribution
from gpax_amd.means import ConstantMean, Mean
from gpax_amd.transforms import IdentityTransform, NormalizeTransform, ScaleTransform, Transform

DATA_TYPES = (np.ndarray, type(None))
Y_STD_DATA_TYPES = DATA_TYPES + (float, int)


_PositiveBijection = _Distribution.from_unconstrained   # the default exp bijection (fast_terms assumes it)

_model_uid = itertools.count(1)   # never reused, unlike id(): CPython hands a freed model's id() to the next one


def _fingerprint(a):
    """Content key of an array (shape + a 64-bit digest of its bytes): what decides whether data that is already
    resident in HBM, or a cached derived array, is still the caller's data -- in-place edits included."""
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    return (a.shape, a.dtype.str, _digest(a))


class GPSample:
    """Container for GP results (gpax/gp.py:52-96)."""

    def __init__(self, hp=None, y=None, mu=None, sd=None):
        self._hp, self._y, self._mu, self._sd = hp, y, mu, sd

    @property
    def y(self):
        """Samples over the GP, shape ``(hp, gp, L)``."""
        return self._y

    @property
    def hp(self):
        return self._hp

    @property
    def mu(self):
        if self._y.shape[0] == 1:
            return self._mu.squeeze()
        return np.mean(self._y, axis=(0, 1)).squeeze()

    @property
    def sd(self):
        if self._y.shape[0] == 1:
            return self._sd.squeeze()
        return np.std(self._y, axis=(0, 1)).squeeze()

    @property
    def ci(self):
        return self.mu - 2.0 * self.sd, self.mu + 2.0 * self.sd


class GaussianProcess:
    """Core Gaussian process class (gpax/gp.py:99-501); same constructor keywords."""

    def __init__(
        self,
        kernel,
        x=None,
        y=None,
        y_std=None,
        observation_noise=False,
        hp_samples=100,
        gp_samples=10,
        input_transform="default",
        output_transform="default",
        metadata=None,
        use_cholesky=False,
        mean=None,
    ):
        if not isinstance(kernel, Kernel):
            raise TypeError("kernel must be a gpax_amd.kernels.Kernel")
        if not isinstance(x, DATA_TYPES) or not isinstance(y, DATA_TYPES):
            raise TypeError("x and y must be numpy arrays or None")
        if not isinstance(y_std, Y_STD_DATA_TYPES):
            raise TypeError("y_std must be an array, a float, an int or None")
        if not isinstance(observation_noise, bool) or not isinstance(use_cholesky, bool):
            raise TypeError("observation_noise and use_cholesky must be bool")
        for name, v in (("hp_samples", hp_samples), ("gp_samples", gp_samples)):
            if not isinstance(v, int) or v <= 0:
                raise ValueError(f"{name} must be a positive int")
        self.kernel = kernel
        self.x, self.y, self.y_std = x, y, y_std
        self.observation_noise = observation_noise
        self.hp_samples, self.gp_samples = hp_samples, gp_samples
        self.input_transform = ScaleTransform() if input_transform == "default" else input_transform
        self.output_transform = NormalizeTransform() if output_transform == "default" else output_transform
        for t in (self.input_transform, self.output_transform):
            if not isinstance(t, (Transform, type(None))):
                raise TypeError("transforms must be Transform instances or None")
        self.metadata = metadata if metadata is not None else {}
        self.use_cholesky = use_cholesky
        self.mean = mean if mean is not None else ConstantMean()
        if not isinstance(self.mean, Mean):
            raise TypeError("mean must be a gpax_amd.means.Mean")
        self._is_fit = False
        self._uid = next(_model_uid)
        self._post_init()

    # gpax/gp.py:171-193
    def _post_init(self):
        if self.x is None and self.y_std is not None:
            raise ValueError("y_std cannot be set if no x or y data is given")
        if self.y is not None and isinstance(self.y_std, (float, int)):
            self.y_std = np.ones(self.y.shape) * self.y_std
        if (self.y is None) ^ (self.x is None):
            raise ValueError("x and y must either both be None or not")
        if self.input_transform is None or self.x is None:
            self.input_transform = IdentityTransform()
        if self.output_transform is None or self.y is None:
            self.output_transform = IdentityTransform()
        self.input_transform.fit(self.x)
        self.output_transform.fit(self.y)
        self._class_specific_post_init()

    def _class_specific_post_init(self):
        pass

    def _memo(self, name, key, make):
        """Per-instance cache of a derived array, keyed by the CONTENT of the array it is derived from and the identity
        of the transform (a Transform is fitted once and immutable afterwards; the model keeps it alive).  Replacing
        or editing x / y in place invalidates the entry, as the reference's recompute-on-every-call would."""
        memo = self.__dict__.setdefault("_memo_store", {})
        hit = memo.get(name)
        if hit is None or hit[0] != key:
            hit = (key, make())
            memo[name] = hit
        return hit[1]

    @property
    def x_transformed(self):
        return self._memo("x", (_fingerprint(self.x), id(self.input_transform)),
                          lambda: self.input_transform.forward(self.x, transforms_as="mean"))

    @property
    def y_transformed(self):
        return self._memo("y", (_fingerprint(self.y), id(self.output_transform)),
                          lambda: np.atleast_1d(self.output_transform.forward(self.y, transforms_as="mean").squeeze()))

    @property
    def y_std_transformed(self):
        y_std = self.output_transform.forward(self.y_std, transforms_as="std")
        if y_std is None:
            return None
        return np.atleast_1d(y_std.squeeze())

    # ------------------------------------------------------------------ the HIP path
    def _core(self):
        return state.get_core()

    def _make_train_resident(self):
        """jax.device_put(x, state.device) of gpax/gp.py:527: once per model."""
        core = self._core()
        x = np.ascontiguousarray(self.x_transformed)
        # the model's never-reused token + the content of what it would upload (the memo key of x_transformed: raw x and
        # the transform): a recycled id(), a same-shape successor or an in-place edit can never alias stale device data
        key = (self._uid, self.__dict__["_memo_store"]["x"][0])
        if getattr(core, "_gp_owner", None) != key:
            core.set_train(x)
            core._gp_owner = key
            core._gp_test = None
            core._gp_factor = None
        return core, x

    def _kernel_struct(self, k_params, k_noise, k_jitter, x_rows):
        kp = dict(k_params)
        noise = np.asarray(k_noise, dtype=np.float64).reshape(-1)
        if noise.size not in (1, x_rows):
            raise ValueError("k_noise must be a scalar or one value per training point")
        return _lib.make_params(
            self.kernel.KIND, kp["k_scale"], kp["k_length"], kp.get("k_period"), noise, k_jitter, True
        )

    def _split(self, params):
        params = dict(params)   # the values are only read (a deepcopy here cost 13 us per NUTS gradient call)
        m_params = {k: v for k, v in params.items() if "m_" in k}  # substring filters, gpax/gp.py:230-231
        k_params = {k: v for k, v in params.items() if "k_" in k}
        k_noise = k_params.pop("k_noise")
        k_jitter = k_params.pop("k_jitter")
        return m_params, k_params, k_noise, k_jitter

    def log_marginal(self, params):
        """Log-density of the ``y_sampled`` site of ``_gp_prior`` (gpax/gp.py:145-151) at the data:
        MVN(loc=m, covariance_matrix=k).log_prob(y).  As in the reference model, ``y_std`` is not
        used here (only in prediction)."""
        m_params, k_params, k_noise, k_jitter = self._split(params)
        core, x = self._make_train_resident()
        if core.m != 0:
            core.set_test(np.zeros((0, x.shape[1])))
            core._gp_test = None
        core._gp_factor = None
        r = self.y_transformed - self.mean(x, **m_params)
        info, logp, _, _ = core.fit_predict(self._kernel_struct(k_params, k_noise, k_jitter, x.shape[0]), r, None, 0.0)
        return logp if info == 0 else float("nan")

    def log_marginal_and_grad(self, params):
        """``log_marginal`` together with its gradient with respect to every entry of ``params``
        that is a kernel hyper-parameter (``gpx_lml_grad``: closed form, one bordered factorisation)
        or a mean-function parameter (``alpha . dm/dtheta`` with ``alpha = K^-1 (y - m)`` from the same
        call and the O(N) mean function differentiated by central differences on the host).  This is
        what jax.grad of the NumPyro potential provides in the reference (gpax/gp.py:515-530)."""
        m_params, k_params, k_noise, k_jitter = self._split(params)
        core, x = self._make_train_resident()
        core._gp_factor = None
        y = self.y_transformed
        r = y - self.mean(x, **m_params)
        info, logp, grads, alpha = core.lml_grad(self._kernel_struct(k_params, k_noise, k_jitter, x.shape[0]), r)
        if info != 0:
            return float("nan"), {}
        out = {k: np.asarray(v, dtype=np.float64) for k, v in grads.items() if k in params}
        out.update(self._mean_param_grads(x, m_params, alpha))
        return logp, out

    def _mean_param_grads(self, x, m_params, alpha):
        """d logp / d theta_m = alpha . dm/dtheta_m.  A Mean that implements ``_mean_grad(x, **params)`` (name -> dm/dtheta,
        shape (N,) or (plate, N)) is differentiated exactly -- what jax.grad gives the reference under NUTS
        (gpax/gp.py:145-151; jnp.piecewise / jnp.where have zero derivative in the break point almost everywhere);
        without the hook: central differences on the O(N) host function."""
        out = {}
        exact = self.mean._mean_grad(x, **m_params) if m_params else None
        for name, val in m_params.items():
            val = np.atleast_1d(np.asarray(val, dtype=np.float64))
            if exact is not None and name in exact:
                dm = np.asarray(exact[name], dtype=np.float64).reshape(val.size, -1)
                out[name] = (np.broadcast_to(dm, (val.size, alpha.shape[0])) @ alpha).reshape(val.shape)
                continue
            g = np.zeros(val.size)
            for i in range(val.size):
                h = 1e-6 * max(1.0, abs(val.flat[i]))
                up, dn = val.copy(), val.copy()
                up.flat[i] += h
                dn.flat[i] -= h
                dm = (self.mean(x, **{**m_params, name: up}) - self.mean(x, **{**m_params, name: dn})) / (2 * h)
                g[i] = float(alpha @ np.broadcast_to(dm, alpha.shape))
            out[name] = g.reshape(val.shape)
        return out

    def log_marginal_and_grad_batch(self, params_list):
        """``log_marginal_and_grad`` for B parameter sets in ONE schedule on the GPU (``gpx_lml_grad_batch``): what the
        reference gets from running its NUTS chains vectorised (``chain_method``, gpax/gp.py:507-509, 517-525).  Returns a
        list of ``(logp, grads)``; a draw's numbers are those of the single call, bit for bit."""
        if len(params_list) == 1:
            return [self.log_marginal_and_grad(params_list[0])]
        core, x = self._make_train_resident()
        core._gp_factor = None
        y = self.y_transformed
        split = [self._split(p) for p in params_list]
        plist = [self._kernel_struct(k_params, k_noise, k_jitter, x.shape[0]) for _, k_params, k_noise, k_jitter in split]
        r = np.stack([y - self.mean(x, **m_params) for m_params, _, _, _ in split])
        info, logp, grads, alpha = core.lml_grad_batch(plist, r)
        res = []
        for b, (m_params, _, _, _) in enumerate(split):
            if info[b] != 0:
                res.append((float("nan"), {}))
                continue
            out = {k: np.asarray(v, dtype=np.float64) for k, v in grads[b].items() if k in params_list[b]}
            out.update(self._mean_param_grads(x, m_params, alpha[b]))
            res.append((float(logp[b]), out))
        return res

    def _gp_prior(self, x=None, y=None, params=None, rng=None):
        """The GP model of gpax/gp.py:133-151.  Without NumPyro there are no sample sites: with
        ``y`` this returns the joint log-density (priors + MVN log-prob) at ``params`` (drawn from the
        priors when omitted); without ``y`` it returns one draw of ``y_sampled`` from the prior."""
        rng = rng if rng is not None else state.get_rng_key()[1]
        if params is None:
            params = {**self.mean.sample_parameters(rng), **self.kernel.sample_parameters(rng)}
        if y is not None:
            lp = self.log_marginal(params)
            for name, p in {**self.mean._params, **self.kernel._params}.items():
                if not p.is_constant:
                    lp += float(np.sum(p.value.log_prob(params[name])))
            return lp
        x = self.x_transformed if x is None else np.asarray(x, dtype=np.float64)
        m_params, k_params, k_noise, k_jitter = self._split(params)
        mean = self.mean(x, **m_params)
        cov = self.kernel(x, x, k_noise=k_noise, k_jitter=k_jitter, **k_params)
        _, draw = self._core().mvn_sample(mean, cov, rng.standard_normal((1, x.shape[0])))
        return draw[0]

    def _get_mean_and_covariance_unconditioned(self, x_new, params):
        """gpax/gp.py:212-245."""
        m_params, k_params, k_noise, k_jitter = self._split(params)
        mean = self.mean(x_new, **m_params)
        noise = k_noise if self.observation_noise else 0.0
        # cov = kernel(x_new, x_new, k_noise=noise, k_jitter, **k_params) (gpax/gp.py:240-244), built on the device at
        # the resident x_new (gpx_prior_cov)
        core = self._core()
        x_new = np.ascontiguousarray(np.asarray(x_new, dtype=np.float64))
        if core.d != x_new.shape[1] or core.n == 0:   # no training inputs on this handle (prior-only model): d from x_new
            core.set_train(x_new[:1])
            core._gp_owner = core._gp_factor = None
        self._set_test(core, x_new)
        prm = _lib.make_params(self.kernel.KIND, k_params["k_scale"], k_params["k_length"], k_params.get("k_period"), None, k_jitter, True)
        cov = core.prior_cov(prm, float(np.asarray(noise, dtype=np.float64).reshape(-1)[0]))
        return mean, cov

    @staticmethod
    def _set_test(core, x_new):
        test_key = (x_new.shape, _digest(x_new))
        if core._gp_test != test_key:
            core.set_test(x_new)
            core._gp_test = test_key

    def _posterior(self, x_new, params, want_cov):
        m_params, k_params, k_noise, k_jitter = self._split(params)
        core, x = self._make_train_resident()
        x_new = np.ascontiguousarray(np.asarray(x_new, dtype=np.float64))
        self._set_test(core, x_new)
        y_std = self.y_std_transformed
        noise_train = y_std**2 if y_std is not None else k_noise  # gpax/gp.py:289-292
        new_noise = float(np.asarray(k_noise).reshape(-1)[0]) if self.observation_noise else 0.0  # :283-286
        prm = self._kernel_struct(k_params, noise_train, k_jitter, x.shape[0])
        r = self.y_transformed - self.mean(x, **m_params)
        m_new = np.broadcast_to(self.mean(x_new, **m_params), (x_new.shape[0],))
        # The reference re-builds and re-factorises k_train on every call (gpax/gp.py:293-304); here the
        # factor is cached on the device and reused while hyper-parameters and data are unchanged
        # (e.g. the 100 candidate batches of AcquisitionFunction._optimize_halton).
        fkey = hash((self.kernel.KIND, repr(sorted((k, np.asarray(v).tobytes()) for k, v in k_params.items())),
                     np.asarray(noise_train).tobytes(), np.asarray(k_jitter).tobytes(), r.tobytes()))
        if getattr(core, "_gp_factor", None) == (core._gp_owner, fkey):
            _, mean, var, cov = core.posterior(m_new, new_noise, want_cov=want_cov)
            return (mean, cov) if want_cov else (mean, var)
        if not want_cov:
            _, _, mean, var = core.fit_predict(prm, r, m_new, new_noise)
            core._gp_factor = (core._gp_owner, fkey)
            return mean, var
        core.fit_factor(prm)
        core.log_marginal(r)
        core._gp_factor = (core._gp_owner, fkey)
        _, mean, var, cov = core.posterior(m_new, new_noise, want_cov=True)
        return mean, cov

    def _get_mean_and_covariance(self, x_new, params):
        """gpax/gp.py:247-312: posterior mean (M,) and covariance (M,M) at already-transformed
        ``x_new`` for one hyper-parameter set."""
        return self._posterior(x_new, params, want_cov=True)

    # upstream (ziatdinovmax/gpax) name for the same thing
    get_mvn_posterior = _get_mean_and_covariance

    def _sample_gp_given_single_hp(self, rng, x_new, params, n, condition_on_data=True):
        """gpax/gp.py:314-338."""
        params = {**self._prior_constants(), **params}
        if condition_on_data:
            mean, cov = self._get_mean_and_covariance(x_new, params)
        else:
            mean, cov = self._get_mean_and_covariance_unconditioned(x_new, params)
        mean = np.broadcast_to(mean, (x_new.shape[0],))
        _, sampled = self._core().mvn_sample(mean, cov, rng.standard_normal((n, x_new.shape[0])))
        return mean, np.sqrt(np.diagonal(cov)), sampled

    def _prior_constants(self):
        """Constants of the priors (e.g. k_jitter) as arrays of shape (plate,)."""
        out = {}
        for name, p in {**self.mean._params, **self.kernel._params}.items():
            if p.is_constant:
                out[name] = np.full((p.plate,), float(p.value))
        return out

    def _get_hp_samples_from_prior(self, rng, fast):
        """gpax/gp.py:340-350."""
        hp = {**self.mean.sample_parameters(rng, self.hp_samples), **self.kernel.sample_parameters(rng, self.hp_samples)}
        n = self.hp_samples
        if fast:
            hp = {k: np.median(v, axis=0, keepdims=True) for k, v in hp.items()}
            n = 1
        return hp, n

    def _get_hp_samples_from_posterior(self, fast):
        raise NotImplementedError

    def _sample_over_hps(self, rng, x_new, hp, n_hp, condition_on_data):
        """The vmap over hyper-parameter draws of gpax/gp.py:352-391: a rank's draws run as ONE batched schedule on the
        GPU (``gpx_sample_batch``: per draw gram + factor + posterior mean / covariance + the M x M factor of the
        covariance + the MVN draws, every kernel launched once for all draws; covariances stay in HBM), in chunks
        sized to a memory budget.  With several ranks the draws are dealt round-robin over them."""
        from gpax_amd import parallel

        # one independent child stream per draw, so the result does not depend on how the draws
        # are dealt over the ranks or chunked
        seeds = rng.integers(0, 2**63 - 1, size=n_hp)
        x_new = np.ascontiguousarray(np.asarray(x_new, dtype=np.float64))
        m, gp_n = x_new.shape[0], self.gp_samples
        consts = self._prior_constants()

        def run(indices):
            core = self._core()
            if condition_on_data:
                core, x = self._make_train_resident()
                y_std = self.y_std_transformed
                n = x.shape[0]
            else:
                x, y_std, n = None, None, 1
                if core.d != x_new.shape[1] or core.n == 0:
                    core.set_train(x_new[:1])
                    core._gp_owner = None
            # gpx_sample_batch ends in batch_invalidate() in BOTH modes (the batch reuses the factor buffers): whatever
            # factor another model cached on this shared handle is gone
            core._gp_factor = None
            self._set_test(core, x_new)
            # chunk size from a 16 GiB budget: A and L of the bordered factor + the three M x M sampling matrices per draw
            npad, mpad = -(-n // 128) * 128, -(-m // 128) * 128
            per_draw = 8 * (2 * (npad + mpad + 64) * npad + 3 * mpad * mpad + 2 * (gp_n + 64) * mpad)
            chunk = int(max(1, min(64, (16 << 30) // max(per_draw, 1))))
            mus, sds, ys = [], [], []
            for c0 in range(0, len(indices), chunk):
                idx = indices[c0:c0 + chunk]
                plist, rs, mns, nns, eps = [], [], [], [], []
                for i in idx:
                    p = {**consts, **{k: v[i] for k, v in hp.items()}}
                    m_params, k_params, k_noise, k_jitter = self._split(p)
                    noise_train = y_std**2 if y_std is not None else k_noise           # gpax/gp.py:289-292
                    if not condition_on_data:
                        noise_train = None
                    plist.append(_lib.make_params(self.kernel.KIND, k_params["k_scale"], k_params["k_length"], k_params.get("k_period"),
                                                  noise_train, k_jitter, True))
                    nns.append(float(np.asarray(k_noise).reshape(-1)[0]) if self.observation_noise else 0.0)   # :283-286 / :236-239
                    if condition_on_data:
                        rs.append(self.y_transformed - self.mean(x, **m_params))
                    mns.append(np.broadcast_to(self.mean(x_new, **m_params), (m,)))
                    eps.append(np.random.default_rng(seeds[i]).standard_normal((gp_n, m)))
                _, mu, sd, y = core.sample_batch(plist, np.stack(rs) if rs else None, np.stack(mns), np.asarray(nns), np.stack(eps),
                                                 conditioned=condition_on_data)
                mus.append(mu); sds.append(sd); ys.append(y)
            return np.concatenate(mus), np.concatenate(sds), np.concatenate(ys)

        mu, sd, ys = parallel.map_draws_batched(run, n_hp)
        return hp, mu, sd, ys

    # the reference's private sampling entry points (gpax/gp.py:352-397), same names and return values
    def _sample_prior(self, rng, x_new, condition_on_data, fast):
        hp, n = self._get_hp_samples_from_prior(rng, fast)
        return self._sample_over_hps(rng, x_new, hp, n, condition_on_data)

    def _sample_unconditioned_prior(self, rng, x_new, fast):
        return self._sample_prior(rng, x_new, False, fast)

    def _sample_conditioned_prior(self, rng, x_new, fast):
        return self._sample_prior(rng, x_new, True, fast)

    def _sample_posterior(self, rng, x_new, fast=True):
        hp, n = self._get_hp_samples_from_posterior(fast)
        return self._sample_over_hps(rng, x_new, hp, n, True)

    def _pre_sample(self, x_new):
        """gpax/gp.py:393-397 (the device_put happens inside the library: gpx_set_test)."""
        _, rng = state.get_rng_key()
        return rng, np.asarray(x_new, dtype=np.float64)

    def _fit_transforms(self):
        """gpax/gp.py:168-193 under its reference name."""
        self._post_init()

    def sample(self, x_new, fast=False):
        """gpax/gp.py:399-462: returns a GPSample with ``y`` of shape (hp, gp, M)."""
        x_new = self.input_transform.forward(x_new, transforms_as="mean")
        _, rng = state.get_rng_key()
        if self.x is None and self.y is None:
            hp, n = self._get_hp_samples_from_prior(rng, fast)
            hp, mu, sd, y = self._sample_over_hps(rng, x_new, hp, n, False)
        elif not self._is_fit:
            hp, n = self._get_hp_samples_from_prior(rng, fast)
            hp, mu, sd, y = self._sample_over_hps(rng, x_new, hp, n, True)
        else:
            hp, n = self._get_hp_samples_from_posterior(fast)
            hp, mu, sd, y = self._sample_over_hps(rng, x_new, hp, n, True)
        hp = {k: np.atleast_1d(v) for k, v in hp.items()}
        # transform back (gpax/gp.py:458-460; the (1,1) statistics broadcast over (hp, gp, M))
        y = self.output_transform.reverse(y, transforms_as="mean")
        mu = self.output_transform.reverse(mu, transforms_as="mean")
        sd = self.output_transform.reverse(sd, transforms_as="std")
        return GPSample(hp=hp, y=y, mu=mu, sd=sd)

    def _fast_predict(self, rng, x_new):
        """gpax/gp.py:464-484: median hyper-parameters -> posterior mean and sqrt(diag cov)."""
        if self.x is None and self.y is None:
            hps, _ = self._get_hp_samples_from_prior(rng, True)
            condition = False
        elif not self._is_fit:
            hps, _ = self._get_hp_samples_from_prior(rng, True)
            condition = True
        else:
            hps, _ = self._get_hp_samples_from_posterior(True)
            condition = True
        hps = {**self._prior_constants(), **{k: v[0] for k, v in hps.items()}}
        if condition:
            mu, var = self._posterior(x_new, hps, want_cov=False)  # only diag(cov) is used (:484)
            return mu, np.sqrt(var)
        mu, cov = self._get_mean_and_covariance_unconditioned(x_new, hps)
        return np.broadcast_to(mu, (x_new.shape[0],)), np.sqrt(np.diagonal(cov))

    def _output_reverse_is_affine(self):
        """The device acquisition epilogue folds ``output_transform.reverse`` in as an affine map (true of the built-in
        Identity / Scale / Normalize transforms): a third probe point catches a user Transform that is not, and the
        acquisition function then evaluates its analytic form on the host from ``predict`` (still the HIP posterior)."""
        t = self.output_transform
        if isinstance(t, (IdentityTransform, ScaleTransform, NormalizeTransform)):
            return True
        a, b, c = (float(np.asarray(t.reverse(np.full(1, v), transforms_as="mean")).ravel()[0]) for v in (0.0, 1.0, -2.5))
        s1, s3 = (float(np.asarray(t.reverse(np.full(1, v), transforms_as="std")).ravel()[0]) for v in (1.0, 3.0))
        return abs(c - (a - 2.5 * (b - a))) <= 1e-9 * (abs(a) + abs(b) + 1.0) and abs(s3 - 3.0 * s1) <= 1e-9 * (abs(s1) + 1.0)

    def _acquisition_fast(self, x_new, kind, param, want_values=True):
        """``predict(x_new, fast=True)`` (gpax/gp.py:464-501) followed by the analytic acquisition function
        (gpax/acquisition.py:260-266, 291-298) WITHOUT the round trip: the posterior mean / variance of all candidates
        stay in HBM and the EI / UCB values and their arg-max are an epilogue over them (``gpx_acquisition``).  The output
        transform's reverse is affine and is applied inside that epilogue.  Conditioned models only.
        Returns ``(values or None, argmax, best)``."""
        x_t = self.input_transform.forward(x_new, transforms_as="mean")
        _, rng = state.get_rng_key()
        if not self._is_fit:
            hps, _ = self._get_hp_samples_from_prior(rng, True)
        else:
            hps, _ = self._get_hp_samples_from_posterior(True)
        hps = {**self._prior_constants(), **{k: v[0] for k, v in hps.items()}}
        self._posterior(x_t, hps, want_cov=False)        # leaves (mean, var) of every candidate on the device
        t = self.output_transform
        shift = float(np.asarray(t.reverse(np.zeros(1), transforms_as="mean")).ravel()[0])
        mu_scale = float(np.asarray(t.reverse(np.ones(1), transforms_as="mean")).ravel()[0]) - shift
        sd_scale = float(np.asarray(t.reverse(np.ones(1), transforms_as="std")).ravel()[0])
        return self._core().acquisition(kind, param, mu_scale, shift, sd_scale, want_values)

    def predict(self, x_new, fast=False):
        """gpax/gp.py:486-501."""
        if not fast:
            sampled = self.sample(x_new, False)
            return sampled.mu.squeeze(), sampled.sd.squeeze()
        x_new = self.input_transform.forward(x_new, transforms_as="mean")
        _, rng = state.get_rng_key()
        mu, sd = self._fast_predict(rng, x_new)
        mu = self.output_transform.reverse(mu, transforms_as="mean")
        sd = self.output_transform.reverse(sd, transforms_as="std")
        return mu.squeeze(), sd.squeeze()

    def fit(self):
        raise NotImplementedError


class _ParamSpace:
    """Bijection between the learnable hyper-parameters (non-constant Parameters, with plates)
    and one unconstrained vector -- what NumPyro's NUTS samples in."""

    def __init__(self, gp):
        self.items = []
        for name, p in {**gp.mean._params, **gp.kernel._params}.items():
            if not p.is_constant:
                self.items.append((name, p.value, p.plate))
        self.dim = sum(pl for _, _, pl in self.items)
        self.constants = gp._prior_constants()

    def unpack(self, u):
        params, log_jac, log_prior, o = dict(self.constants), 0.0, 0.0, 0
        for name, dist, plate in self.items:
            x, lj = dist.from_unconstrained(u[o : o + plate])
            params[name] = np.asarray(x, dtype=np.float64)
            log_jac += float(np.sum(lj))
            log_prior += float(np.sum(dist.log_prob(x)))
            o += plate
        return params, log_prior + log_jac

    def init_to_median(self):
        return np.concatenate([np.atleast_1d(d.to_unconstrained(np.full(pl, d.median))) for _, d, pl in self.items])

    def _prior_terms(self, u):
        """Host half of the potential at ``u``: the constrained parameters, d x / d u, the gradient of the log-prior +
        log|Jacobian| and its value (``None`` where the point is outside the support or overflows: a rejected -inf)."""
        params, g_prior, dx, o = dict(self.constants), np.zeros(self.dim), np.zeros(self.dim), 0
        log_prior = 0.0
        with np.errstate(all="ignore"):   # far-out trial points of the step-size search overflow to a rejected -inf
            for name, dist, plate in self.items:
                uu = u[o : o + plate]
                fast = getattr(dist, "fast_terms", None)
                if fast is not None and type(dist).from_unconstrained is _PositiveBijection:   # exp bijection, closed forms
                    x, dxdu, gterm, lp = fast(np.asarray(uu, dtype=np.float64))
                    params[name] = x
                    log_prior += lp
                else:
                    x, dxdu, gterm = dist.grad_terms(uu)
                    _, lj = dist.from_unconstrained(uu)
                    params[name] = np.asarray(x, dtype=np.float64)
                    log_prior += float(np.asarray(dist.log_prob(x)).sum() + np.asarray(lj).sum())
                dx[o : o + plate], g_prior[o : o + plate] = dxdu, gterm
                o += plate
        if not math.isfinite(log_prior):
            return None
        for v in params.values():   # (ndarray methods: the np.all / np.isfinite wrappers cost 20 us per call here)
            a = np.asarray(v, dtype=np.float64)
            if not (np.isfinite(a).all() and (np.abs(a) < 1e150).all()):
                return None
        return params, g_prior, dx, log_prior

    def _finish(self, terms, logp, grads):
        _, g_prior, dx, log_prior = terms
        val = logp + log_prior
        if not np.isfinite(val):
            return -np.inf, np.zeros(self.dim)
        g, o = g_prior.copy(), 0
        for name, _, plate in self.items:
            gv = grads[name].reshape(-1)
            g[o : o + plate] += dx[o : o + plate] * (gv if gv.size == plate else np.broadcast_to(gv, (plate,)))
            o += plate
        return val, g

    def value_and_grad(self, gp, u):
        """Log joint density in unconstrained space (priors + log|Jacobian| + log-marginal) and its
        gradient: the potential NumPyro differentiates for NUTS / SVI, one ``gpx_lml_grad`` call."""
        terms = self._prior_terms(u)
        if terms is None:
            return -np.inf, np.zeros(self.dim)
        logp, grads = gp.log_marginal_and_grad(terms[0])
        return self._finish(terms, logp, grads)

    def value_and_grad_batch(self, gp, us):
        """``value_and_grad`` at several points with ONE batched GPU call (``gpx_lml_grad_batch``) for those inside the
        support: the potentials of chains stepping in lock-step."""
        terms = [self._prior_terms(u) for u in us]
        live = [i for i, t in enumerate(terms) if t is not None]
        res = [(-np.inf, np.zeros(self.dim))] * len(us)
        if live:
            outs = gp.log_marginal_and_grad_batch([terms[i][0] for i in live])
            for i, (logp, grads) in zip(live, outs):
                res[i] = self._finish(terms[i], logp, grads)
        return res

    def constrain(self, u):
        out, o = {}, 0
        for name, dist, plate in self.items:
            out[name] = np.asarray(dist.from_unconstrained(u[o : o + plate])[0], dtype=np.float64)
            o += plate
        return out


class ExactGP(GaussianProcess):
    """gpax/gp.py:504-544: fully Bayesian GP; ``fit()`` samples the hyper-parameter posterior."""

    def __init__(self, *args, mcmc=None, num_warmup=2000, num_chains=1, chain_method="sequential", mcmc_run_kwargs=None, **kw):
        for name, v in (("num_warmup", num_warmup), ("num_chains", num_chains)):
            if not isinstance(v, int) or v <= 0:
                raise ValueError(f"{name} must be a positive int")
        self.mcmc = mcmc
        self.num_warmup, self.num_chains, self.chain_method = num_warmup, num_chains, chain_method
        self.mcmc_run_kwargs = mcmc_run_kwargs if mcmc_run_kwargs is not None else {}
        super().__init__(*args, **kw)

    def fit(self):
        """gpax/gp.py:512-534: NUTS over the hyper-parameters with ``init_to_median``, ``num_warmup``
        adaptation steps and ``hp_samples`` kept draws per chain (``gpax_amd.mcmc``).  Every leapfrog
        step is one ``gpx_lml_grad`` call (log-marginal + closed-form gradient).  ``mcmc_run_kwargs``
        may carry ``algorithm`` ("nuts" | "hmc"), ``target_accept_prob``, ``max_tree_depth``,
        ``num_leapfrog`` (hmc), ``adapt_mass_matrix``, ``step_size``."""
        key, rng = state.get_rng_key()
        space = _ParamSpace(self)
        kw = dict(self.mcmc_run_kwargs)
        draws, infos = {name: [] for name, _, _ in space.items}, []
        # one independent child stream per chain: a chain's draws do not depend on how the chains are scheduled
        rngs = [rng] if self.num_chains == 1 else [np.random.default_rng(sq) for sq in np.random.SeedSequence(int(rng.integers(0, 2**63 - 1))).spawn(self.num_chains)]
        if self.num_chains > 1 and self.chain_method != "sequential":
            # the reference's "vectorized" / "parallel" chain methods (gpax/gp.py:507-509): the chains step in LOCK-STEP,
            # every leapfrog step of all of them is one gpx_lml_grad_batch call
            results = mcmc.sample_lockstep(lambda us: space.value_and_grad_batch(self, us), [space.init_to_median() for _ in rngs],
                                           self.num_warmup, self.hp_samples, rngs, **kw)
        else:   # chain_method "sequential" (gpax/gp.py:509, the reference default)
            results = [mcmc.sample(lambda u: space.value_and_grad(self, u), space.init_to_median(), self.num_warmup, self.hp_samples, r, **kw)
                       for r in rngs]
        for d, info in results:
            for row in d:
                for name, v in space.constrain(row).items():
                    draws[name].append(v)
            infos.append(info)
        self.mcmc = {k: np.stack(v) for k, v in draws.items()}
        self.metadata["fit_key"] = np.array(key)
        self.metadata["mcmc_info"] = infos
        self._is_fit = True

    def _get_hp_samples_from_posterior(self, fast):
        """gpax/gp.py:536-544."""
        hp = dict(self.mcmc)
        if fast:
            hp = {k: np.median(v, axis=0, keepdims=True) for k, v in hp.items()}
        n = len(next(iter(hp.values())))
        return hp, n


class _Adam:
    """numpyro.optim.Adam(step_size, b1, b2, eps) on one flat vector (ascent on the ELBO)."""

    def __init__(self, dim, step_size=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.step_size, self.b1, self.b2, self.eps = step_size, b1, b2, eps
        self.m, self.v, self.t = np.zeros(dim), np.zeros(dim), 0

    def step(self, x, grad):
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * grad
        self.v = self.b2 * self.v + (1 - self.b2) * grad * grad
        mhat = self.m / (1 - self.b1**self.t)
        vhat = self.v / (1 - self.b2**self.t)
        return x + self.step_size * mhat / (np.sqrt(vhat) + self.eps)


def _softplus(x):
    return np.logaddexp(0.0, x)


class VariationalInferenceGP(GaussianProcess):
    """gpax/gp.py:547-605: stochastic variational inference over the hyper-parameters with a
    mean-field normal guide in unconstrained space (NumPyro's ``AutoNormal``: ``loc`` initialised at a
    uniform draw in (-2, 2), ``scale`` at 0.1 through a softplus), the single-particle
    reparameterised ``Trace_ELBO`` and Adam (``b1 = 0.5``, ``step_size = 1e-3`` unless given,
    gpax/gp.py:565-569).  Each optimiser step is one ``gpx_lml_grad`` call.  Predictions use the
    guide median only (``hp_samples`` is forced to 1, gpax/gp.py:570-576)."""

    def __init__(self, *args, num_steps=100, optimizer_kwargs=None, hp_samples=1, svi=None, kernel_params=None, **kw):
        if not isinstance(num_steps, int) or num_steps <= 0:
            raise ValueError("num_steps must be a positive int")
        self.num_steps = num_steps
        self.optimizer_kwargs = dict(optimizer_kwargs) if optimizer_kwargs is not None else {}
        self.svi, self.kernel_params = svi, kernel_params
        super().__init__(*args, hp_samples=hp_samples, **kw)

    def _class_specific_post_init(self):
        self.optimizer_kwargs.setdefault("b1", 0.5)
        self.optimizer_kwargs.setdefault("step_size", 1e-3)
        if self.hp_samples != 1:
            import warnings

            warnings.warn(
                f"hp_samples is set to {self.hp_samples} but will be overridden to 1. In VariationalInferenceGP, "
                "only the median value of the hyperparameters are used for predictions."
            )
            self.hp_samples = 1

    def _print_summary(self):
        """gpax/gp.py:558-563."""
        print("\nInferred GP parameters")
        for k, vals in self.kernel_params["params"].items():
            print(k, " " * (15 - len(k)), np.around(vals, 4))

    def fit(self):
        key, rng = state.get_rng_key()
        space = _ParamSpace(self)
        dim = space.dim
        loc = rng.uniform(-2.0, 2.0, dim)                 # init_to_uniform(radius=2), AutoNormal's default
        rho = np.full(dim, math.log(math.expm1(0.1)))      # softplus(rho) = 0.1 = AutoNormal's init_scale
        opt = _Adam(2 * dim, **self.optimizer_kwargs)
        losses = np.empty(self.num_steps)
        for it in range(self.num_steps):
            eps = rng.standard_normal(dim)
            scale = _softplus(rho)
            u = loc + scale * eps
            val, g = space.value_and_grad(self, u)
            entropy = float(np.sum(np.log(scale))) + 0.5 * dim * (1.0 + math.log(2 * math.pi))
            losses[it] = -(val + entropy)                  # Trace_ELBO loss = -ELBO, one particle
            if np.isfinite(val):
                dscale_drho = 1.0 / (1.0 + np.exp(-rho))
                grad = np.concatenate([g, (g * eps + 1.0 / scale) * dscale_drho])
                both = opt.step(np.concatenate([loc, rho]), grad)
                loc, rho = both[:dim], both[dim:]
        self.svi = {"loc": loc, "scale": _softplus(rho), "space": space}
        self.kernel_params = {"params": space.constrain(loc), "losses": losses}
        if state.verbose > 0:
            self._print_summary()
        self.metadata["fit_key"] = np.array(key)
        self._is_fit = True

    def _get_hp_samples_from_posterior(self, fast):
        """gpax/gp.py:593-605: the guide median (the unconstrained ``loc`` pushed through the bijection)."""
        if fast:
            import warnings

            warnings.warn(
                "Note setting fast = True for VIGP will do nothing. VIGP only uses the median value of the "
                "hyperparameters by default.",
                RuntimeWarning,
            )
        hp = {k: np.atleast_1d(v)[None] for k, v in self.svi["space"].constrain(self.svi["loc"]).items()}
        return hp, 1
