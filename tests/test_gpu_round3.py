"""Round-3 parity cases (``-m gpu``, through the C ABI): the sizes ``ExactGP.fit()`` / ``sample()`` / the C5 acquisition step are
benchmarked at, which the earlier suites reached only with finiteness checks.

  * ``gpx_lml_grad`` at N = 2048 / 4096 against the oracle's closed form, incl. a run whose K^-1 accumulation goes through
    GROUPED (K = 256 / 512) bulk phases and `grad_reduce_kernel` over > 100 tiles (gpax/gp.py:515-530 is what it replaces);
  * ``gpx_sample_batch`` with M = 1024 / 2500 test points: the M x M posterior covariance is factorised by the split-panel
    schedule (>= 8 block columns), as C5's ``sample()`` would (gpax/gp.py:314-338, 382-391);
  * ``gpx_acquisition`` at C5 scale (N = 8192, 10 000 candidates) against the host formulae on ``gpx_posterior`` outputs
    (gpax/acquisition.py:260-266, 291-298);
  * the factor-cache interleaving of ADVICE round 2 (conditioned predict / prior-only sample / the same predict again).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_GRAD_REL = 1e-7    # of the largest gradient component (a sum of N^2 terms G_ij dK_ij, |K^-1| ~ cond(K))
TOL_LOGP_REL = 1e-10
TOL_ALPHA_REL = 1e-7


@pytest.fixture(scope="module")
def core():
    import gpax_amd

    c = gpax_amd.Core(0)
    yield c
    c.close()


def _problem(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (n, d))
    y = np.sin(3.0 * x.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(n)
    return x, (y - y.mean()) / (y.std() + 1e-8) - 1.0, rng


@pytest.mark.parametrize("kind,n,d,ard,noise,opts", [
    (0, 2048, 4, True, 0.02, {}),                                  # 16 block columns, default grouping (ungrouped)
    (0, 2048, 4, True, 0.02, {"pair_min": 4, "quad_min": 8}),      # the same problem through K = 512 and K = 256 passes
    (1, 4096, 3, False, 0.05, {}),                                 # the size profiles/*_fit_bench.txt quotes (2.3 ms per call)
    (0, 4096, 2, False, 0.01, {"pair_min": 6, "quad_min": 12}),
])
def test_lml_grad_matches_oracle_at_fit_sizes(core, kind, n, d, ard, noise, opts):
    import gpax_amd
    from oracle import gp_oracle as orc

    x, r, rng = _problem(n, d, seed=7 * n + d)
    k_length = rng.uniform(0.7, 1.4, d) if ard else np.array([0.9])
    k_scale = 1.2
    saved = {k: core.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            core.set_option(k, v)
        core.set_train(x)
        p = gpax_amd.make_params(kind, k_scale, k_length, None, np.array([noise]), 1e-6, True)
        info, logp, grads, alpha = core.lml_grad(p, r)
        info2, logp2, grads2, alpha2 = core.lml_grad(p, r)     # repeatable bit for bit (fixed-order reduction)
    finally:
        for k, v in saved.items():
            core.set_option(k, v)
    assert info == 0 and info2 == 0
    assert logp == logp2 and np.array_equal(alpha, alpha2) and all(np.array_equal(grads[k], grads2[k]) for k in grads)
    logp_ref, g_ref, alpha_ref = orc.lml_grad(kind, x, r, k_scale, k_length, None, noise, 1e-6)
    assert abs(logp - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
    assert np.abs(alpha - alpha_ref).max() <= TOL_ALPHA_REL * np.abs(alpha_ref).max()
    scale = max(np.abs(v).max() for v in g_ref.values())
    assert set(grads) == set(g_ref)
    for name, ref in g_ref.items():
        assert grads[name].shape == ref.shape, name
        assert np.abs(grads[name] - ref).max() <= TOL_GRAD_REL * scale, (name, grads[name], ref)


def test_lml_grad_grouping_does_not_move_the_result_beyond_rounding(core):
    """Grouped and ungrouped K^-1 accumulation are different summation orders of the same sum: equal to ~1e-12 relative."""
    import gpax_amd

    x, r, rng = _problem(2048, 4, seed=3)
    core.set_train(x)
    p = gpax_amd.make_params(0, 1.0, np.array([1.1]), None, np.array([0.03]), 1e-6, True)
    _, logp_a, g_a, _ = core.lml_grad(p, r)
    saved = {k: core.get_option(k) for k in ("pair_min", "quad_min")}
    try:
        core.set_option("pair_min", 2); core.set_option("quad_min", 4)
        _, logp_b, g_b, _ = core.lml_grad(p, r)
    finally:
        for k, v in saved.items():
            core.set_option(k, v)
    scale = max(np.abs(v).max() for v in g_a.values())
    assert abs(logp_a - logp_b) <= 1e-11 * abs(logp_a)
    assert all(np.abs(g_a[k] - g_b[k]).max() <= 1e-9 * scale for k in g_a)


@pytest.mark.parametrize("conditioned", [True, False])
@pytest.mark.parametrize("n,m,d,kind,b", [(300, 1024, 2, 0, 2), (500, 2500, 2, 1, 2)])
def test_device_resident_sampling_with_a_split_schedule_covariance_factor(core, conditioned, n, m, d, kind, b):
    """``gpx_sample_batch`` where the M x M posterior (or prior) covariance needs the split-panel schedule (8 and 20 block
    columns): mean, sd and ``mean + eps chol(cov)^T`` against the oracle with the SAME eps."""
    import gpax_amd
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(n + m)
    x, y, xn, _ = orc.synthetic_problem(n, d, m, seed=n, kind=kind)
    draws = [dict(k_scale=np.array([float(np.exp(0.2 * rng.standard_normal()))]), k_length=np.exp(0.2 * rng.standard_normal(1)),
                  k_noise=np.array([0.01 + 0.02 * rng.uniform()]), k_jitter=np.array([1e-6])) for _ in range(b)]
    plist = [gpax_amd.make_params(kind, q["k_scale"], q["k_length"], None, q["k_noise"], q["k_jitter"]) for q in draws]
    nsamp = 3
    eps = rng.standard_normal((b, nsamp, m))
    r = np.stack([y - 1.0] * b)
    m_new = np.ones((b, m))
    new_noise = np.array([q["k_noise"][0] for q in draws])     # observation_noise=True: cov >= k_noise on the diagonal
    core.set_train(x)
    core.set_test(xn)
    info, mean, sd, smp = core.sample_batch(plist, r if conditioned else None, m_new, new_noise, eps, conditioned=conditioned)
    assert (info == 0).all() and smp.shape == (b, nsamp, m)
    for i, q in enumerate(draws):
        if conditioned:
            mean_ref, cov_ref = orc.mean_and_covariance(kind, x, y, xn, dict(q), observation_noise=True)
        else:
            mean_ref, cov_ref = orc.mean_and_covariance_unconditioned(kind, xn, dict(q), observation_noise=True)
            mean_ref = np.broadcast_to(mean_ref, (m,))
        assert np.abs(mean[i] - mean_ref).max() <= 1e-8
        assert np.abs(sd[i] - np.sqrt(np.diag(cov_ref))).max() <= 1e-8
        ref = mean_ref[None, :] + eps[i] @ orc.cholesky_lower(cov_ref).T
        # chol(cov) amplifies differences in cov by ~1/sqrt(lambda_min(cov)) <= 1/sqrt(k_noise) = 10
        assert np.abs(smp[i] - ref).max() <= 1e-6, np.abs(smp[i] - ref).max()


def test_acquisition_epilogue_at_c5_scale(core):
    """BASELINE.json configs[4] at full size: Matern-5/2, d = 2, N = 8192, structured (piecewise) prior mean, 10 000
    candidates in one posterior call; the device EI / UCB values and arg-max against the oracle's formulae
    (gpax/acquisition.py:260-266, 291-298) evaluated on the host from the SAME call's (mean, var)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    n, m = 8192, 10000
    rng = np.random.default_rng(7)
    x = rng.uniform(0.0, 3.0, (n, 2))
    xn = rng.uniform(0.0, 3.0, (m, 2))
    y = orc.piecewise_mean(x) + 0.05 * rng.standard_normal(n)
    core.set_train(x)
    core.set_test(xn)
    p = gpax_amd.make_params(gpax_amd.MATERN52, 1.0, 1.0, None, np.array([0.01]), 1e-6, True)
    info, logp, mean, var = core.fit_predict(p, y - orc.piecewise_mean(x), orc.piecewise_mean(xn), 0.0)
    assert info == 0 and np.isfinite(logp) and (var > 0).all()
    sd = np.sqrt(var)
    y_max = float(y.max())
    for kind, param, want in ((0, y_max, orc.expected_improvement(mean, sd, y_max)), (1, 4.0, orc.upper_confidence_bound(mean, sd, 4.0))):
        vals, idx, best = core.acquisition(kind, param)
        tol = 1e-11 * max(1.0, np.abs(want).max())
        assert np.abs(vals - want).max() <= tol
        assert idx == int(np.argmax(want)) and abs(best - want.max()) <= tol
    # an affine output transform folded in (Normalize-style): mu = 3 mean - 2, sd = 3 sd
    vals, idx, best = core.acquisition(0, y_max, 3.0, -2.0, 3.0)
    want = orc.expected_improvement(3.0 * mean - 2.0, 3.0 * sd, y_max)
    assert np.abs(vals - want).max() <= 1e-11 * max(1.0, np.abs(want).max()) and idx == int(np.argmax(want))
    # the resident (mean, var) is dropped by whatever rewrites the result block
    core.set_test(xn[:10])
    with pytest.raises(gpax_amd.GpxError):
        core.acquisition(0, y_max)


def test_prior_only_sample_between_two_cached_predicts():
    """a prior-only model's ``sample()`` on the shared handle between two identical conditioned
    ``predict(fast=True)`` calls of another model used to leave that model's factor-cache key in place while
    ``gpx_sample_batch`` had dropped the factor (GPX_ERR_STATE on the third call)."""
    import gpax_amd
    from gpax_amd import state

    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (200, 2))
    y = np.sin(2 * x[:, 0]) + 0.1 * rng.standard_normal(200)
    xs = rng.uniform(-1, 1, (40, 2))
    a = gpax_amd.ExactGP(kernel=gpax_amd.RBFKernel(), x=x, y=y)
    b = gpax_amd.ExactGP(kernel=gpax_amd.RBFKernel(), hp_samples=3, gp_samples=2)      # prior only, same d
    state.set_rng_key(5)
    mu1, sd1 = a.predict(xs, fast=True)
    state.set_rng_key(6)
    s = b.sample(xs)
    assert s.y.shape == (3, 2, 40)
    state.set_rng_key(5)
    mu2, sd2 = a.predict(xs, fast=True)
    assert np.allclose(mu1, mu2, rtol=0, atol=1e-12) and np.allclose(sd1, sd2, rtol=0, atol=1e-12)


def test_exact_mean_gradient_hook_against_central_differences():
    """``Mean._mean_grad`` (the derivatives jax.grad gives the reference for the structured mean of
    examples/04_Structured_GP.ipynb cell 10): the beta-derivatives agree with central differences of the mean function,
    and d/dt is exactly 0 where central differences of the step are noise."""
    import importlib.util
    import os

    import gpax_amd

    spec = importlib.util.spec_from_file_location("structured_gp", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "structured_gp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(1)
    x = rng.uniform(0.05, 3.0, (60, 1))
    mean = mod.PiecewiseMeanPrior()
    y = mean(x, m_t=1.7, m_beta1=4.5, m_beta2=2.5) + rng.normal(0.0, 0.1, 60)
    gp = gpax_amd.ExactGP(kernel=gpax_amd.MaternKernel(), mean=mean, input_transform=None, output_transform=None, x=x, y=y)
    params = {"k_scale": np.array([1.0]), "k_length": np.array([0.8]), "k_noise": np.array([0.02]), "k_jitter": np.array([1e-6]),
              "m_t": np.array([1.6]), "m_beta1": np.array([4.0]), "m_beta2": np.array([2.2])}
    logp, g = gp.log_marginal_and_grad(params)
    assert np.isfinite(logp) and g["m_t"].shape == (1,) and g["m_t"][0] == 0.0
    for name in ("m_beta1", "m_beta2", "k_length", "k_scale"):
        h = 1e-6
        up, dn = dict(params), dict(params)
        up[name] = params[name] + h
        dn[name] = params[name] - h
        fd = (gp.log_marginal(up) - gp.log_marginal(dn)) / (2 * h)
        assert abs(fd - g[name][0]) <= 1e-5 * max(1.0, abs(fd)), (name, fd, g[name])

    class NoHook(mod.PiecewiseMeanPrior):
        def _mean_grad(self, x, **params):
            return None

    gp2 = gpax_amd.ExactGP(kernel=gpax_amd.MaternKernel(), mean=NoHook(), input_transform=None, output_transform=None, x=x, y=y)
    _, g2 = gp2.log_marginal_and_grad(params)      # central differences of the host mean function
    assert abs(g2["m_beta1"][0] - g["m_beta1"][0]) <= 1e-5 * max(1.0, abs(g["m_beta1"][0]))
    assert abs(g2["m_beta2"][0] - g["m_beta2"][0]) <= 1e-5 * max(1.0, abs(g["m_beta2"][0]))


@pytest.mark.parametrize("kind,n,d,ard,b,opts", [
    (0, 60, 2, True, 3, {}),                      # one block column: the single-stream schedule
    (1, 300, 3, False, 4, {}),                    # three block columns: split schedule, in-kernel hand-overs (<= 4 draws)
    (2, 190, 3, True, 2, {}),                     # periodic, ARD length-scales
    (0, 1024, 8, False, 4, {}),                   # the size the bench's "grad_batched" entry runs
    (0, 1400, 4, True, 6, {"pair_min": 4, "quad_min": 8}),   # > 4 draws: kernel-boundary hand-overs; grouped K^-1 passes
])
def test_batched_gradient_calls_equal_single_calls_and_the_oracle(core, kind, n, d, ard, b, opts):
    """``gpx_lml_grad_batch``: B gradient evaluations in one schedule (the reference's vectorised NUTS chains,
    gpax/gp.py:507-509, 517-525).  Per draw: the bits of a single ``gpx_lml_grad`` call, and the oracle's closed form."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, r0, rng = _problem(n, d, seed=n + b)
    draws = []
    for _ in range(b):
        ell = rng.uniform(0.7, 1.5, d) if ard else np.array([float(rng.uniform(0.7, 1.5))])
        per = (np.array([float(rng.uniform(1.5, 2.5))]) if kind == 2 else None)
        draws.append((float(rng.uniform(0.8, 1.4)), ell, per, float(rng.uniform(0.01, 0.05))))
    plist = [gpax_amd.make_params(kind, s, ell, per, np.array([nz]), 1e-6, True) for s, ell, per, nz in draws]
    r = np.stack([r0 + 0.1 * i for i in range(b)])     # a different prior mean per chain
    saved = {k: core.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            core.set_option(k, v)
        core.set_train(x)
        info, logp, grads, alpha = core.lml_grad_batch(plist, r)
        info2, logp2, grads2, alpha2 = core.lml_grad_batch(plist, r)
        singles = [core.lml_grad(plist[i], r[i]) for i in range(b)]
    finally:
        for k, v in saved.items():
            core.set_option(k, v)
    assert (info == 0).all() and np.array_equal(logp, logp2) and np.array_equal(alpha, alpha2)
    for i, (s, ell, per, nz) in enumerate(draws):
        si, sl, sg, sa = singles[i]
        assert si == 0 and sl == logp[i] and np.array_equal(sa, alpha[i])
        assert all(np.array_equal(sg[k], grads[i][k]) and np.array_equal(grads[i][k], grads2[i][k]) for k in sg)
        if n <= 400 or i == 0:      # the oracle forms an (n, n, d) tensor
            logp_ref, g_ref, alpha_ref = orc.lml_grad(kind, x, r[i], s, ell, per, nz, 1e-6)
            assert abs(logp[i] - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
            assert np.abs(alpha[i] - alpha_ref).max() <= TOL_ALPHA_REL * np.abs(alpha_ref).max()
            scale = max(np.abs(v).max() for v in g_ref.values())
            for name, ref in g_ref.items():
                assert np.abs(grads[i][name] - ref).max() <= TOL_GRAD_REL * scale, (name, grads[i][name], ref)


def test_batched_gradient_reports_a_bad_draw_and_leaves_the_others_alone(core):
    import gpax_amd

    x, r0, _ = _problem(300, 2, seed=2)
    good = np.full(300, 0.01)
    bad = good.copy()
    bad[137] = -5.0
    plist = [gpax_amd.make_params(0, 1.0, 1.0, None, v, 1e-6) for v in (good, bad, good)]
    core.set_train(x)
    info, logp, grads, alpha = core.lml_grad_batch(plist, np.stack([r0] * 3))
    assert list(info) == [0, 138, 0]
    assert np.isfinite(logp[[0, 2]]).all() and all(np.isnan(v).all() for v in grads[1].values())
    i0, l0, g0, a0 = core.lml_grad(plist[0], r0)
    assert l0 == logp[0] == logp[2] and np.array_equal(a0, alpha[2])


def test_fit_with_lockstep_chains_equals_sequential_chains():
    """``ExactGP.fit(num_chains = 3, chain_method = "vectorized")``: the chains' leapfrog steps are batched into
    ``gpx_lml_grad_batch`` calls; every chain keeps the draws of the sequential run (same random streams, and a batched
    gradient carries the bits of the single call)."""
    import gpax_amd
    from gpax_amd import state

    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (160, 2))
    y = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + 0.1 * rng.standard_normal(160)
    fits = {}
    for method in ("sequential", "vectorized"):
        state.set_rng_key(17)
        gp = gpax_amd.ExactGP(kernel=gpax_amd.RBFKernel(), x=x, y=y, hp_samples=12, num_warmup=25, num_chains=3, chain_method=method)
        gp.fit()
        fits[method] = gp
    a, b = fits["sequential"], fits["vectorized"]
    assert set(a.mcmc) == set(b.mcmc)
    for k in a.mcmc:
        assert a.mcmc[k].shape[0] == 36 and np.array_equal(a.mcmc[k], b.mcmc[k]), k
    info = b.metadata["mcmc_info"]
    total = sum(i["num_grad_evals"] for i in info)
    assert info[0]["batched_calls"] < total and info[0]["batched_points"] == total    # fewer GPU calls than gradient evaluations


@pytest.mark.parametrize("n,m", [(1500, 90), (4096, 300)])
def test_first_column_publication_of_bulk_launches_does_not_move_a_bit(core, n, m):
    """Option ``bulk_first_col`` (default on): ungrouped bulk launches enumerate their first block column first and those
    tiles publish themselves, so the panel stream's next column update and the band kernel's phase 0 wait for that counter
    instead of the launch's completion event.  Only the ORDER in which workgroups take tiles changes: same bits."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(n, 6, m, seed=n)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_train(x)
    core.set_test(xn)
    try:
        core.set_option("bulk_first_col", 0)
        ref = core.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
        core.set_option("bulk_first_col", 1)
        for _ in range(3):
            got = core.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
            assert got[0] == ref[0] == 0 and got[1] == ref[1] and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    finally:
        core.set_option("bulk_first_col", 1)


def test_results_through_the_pinned_mirror_across_growing_and_shrinking_problems(core):
    """Single-draw steps hand their results to the host through a pinned mirror the last kernels write themselves, and up to
    N = 1024 their hyper-parameters travel in the first launch's kernel arguments (api.hip: ensure_mirror, upload_params_batch).
    Both must survive what re-allocates the pinned buffers around them -- a larger N (the upload ring grows), a larger M (the
    mirror grows), a smaller problem afterwards -- and the interleaving of the two entry points that share the result block;
    every step is checked against the oracle (gpax/gp.py:247-312 for the posterior, :145-151 for the log-marginal)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    for kind, n, m, d in ((1, 30, 200, 1), (0, 1024, 100, 1), (0, 64, 700, 2), (1, 1100, 50, 3), (0, 40, 10, 1)):
        x, r, rng = _problem(n, d, seed=n + m)
        xn = rng.uniform(-1, 1, (m, d))
        k_length = np.array([0.8])
        for noise in (0.03, 0.05):   # two draws per size: the second reuses every buffer of the first
            p = gpax_amd.make_params(kind, 1.1, k_length, None, np.array([noise]), 1e-6, True)
            core.set_train(x); core.set_test(xn); core.set_targets(r, np.ones(m))
            core.fit_predict_async(p, 0.0)
            info, logp, mean, var = core.fetch_results()
            info_g, logp_g, grads, alpha = core.lml_grad(p, r)
            assert info == 0 and info_g == 0
            prm = {"k_scale": 1.1, "k_length": k_length, "k_noise": noise, "k_jitter": 1e-6}
            logp_ref, mean_ref, var_ref = orc.fit_predict(kind, x, r + 1.0, xn, prm, mean_value=1.0)
            assert abs(logp - logp_ref) <= TOL_LOGP_REL * abs(logp_ref) and abs(logp_g - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
            assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8
            _, g_ref, alpha_ref = orc.lml_grad(kind, x, r, 1.1, k_length, None, noise, 1e-6)
            assert np.abs(alpha - alpha_ref).max() <= TOL_ALPHA_REL * np.abs(alpha_ref).max()
            scale = max(np.abs(v).max() for v in g_ref.values())
            for name, ref in g_ref.items():
                assert np.abs(grads[name] - ref).max() <= TOL_GRAD_REL * scale, (name, n)


def test_gram_on_the_matrix_cores_variants_agree_bitwise_and_match_the_oracle(core):
    """RBF / Matern grams run as an augmented matrix product on fp64 MFMA (gram.hip: the reference's own |a|^2 - 2ab + |b|^2
    form, gpax/kernels.py:39-51, clipped at zero) in two variants -- a self-staging tile kernel for launches of at most two
    rounds of workgroups and a prep launch + LDS-free tile kernel beyond -- and, up to 256 points, inside the one-launch
    build of the bordered matrix.  A given entry must not depend on which variant built it (a batched call and a single-draw
    call of the same problem pick different variants), and every variant holds the 1e-12 of the K parity tests."""
    import gpax_amd
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(5)
    for kind, d in ((gpax_amd.RBF, 8), (gpax_amd.MATERN52, 16), (gpax_amd.RBF, 3), (gpax_amd.MATERN52, 62)):
        n_big, n_small = 4200, 300    # 66 x 66 = 4356 tiles (> 16 per CU: the prep variant) against 5 x 5 (self-staging)
        x = rng.uniform(-1, 1, (n_big, d))
        ell = rng.uniform(0.7, 1.6, d)
        p = gpax_amd.make_params(kind, 1.3, ell, None, np.array([0.05]), 1e-6, True)
        k_big = core.gram(p, x, x)
        k_small = core.gram(p, x[:n_small], x[:n_small])
        blk = k_big[:n_small, :n_small].copy()
        blk[np.diag_indices(n_small)] = k_small[np.diag_indices(n_small)]   # (the diagonal rule is keyed on the shapes: both carry it)
        assert np.array_equal(np.diag(k_big)[:n_small], np.diag(k_small))
        assert np.array_equal(blk, k_small), (kind, d)
        rows = rng.choice(n_big, 200, replace=False)
        ref = orc.gram(kind, x[rows], x, 1.3, ell, None, np.array([0.05]), 1e-6, False)
        got = k_big[rows].copy()
        got[np.arange(200), rows] -= 0.05 + 1e-6
        assert np.abs(got - ref).max() <= 1e-12, (kind, d)
    # beyond 62 dimensions the direct-difference kernel takes over; ragged shapes through both tile kernels
    for n1, n2, d in ((130, 70, 63), (130, 70, 100), (700, 650, 5), (4500, 70, 4)):
        x1, x2 = rng.uniform(-1, 1, (n1, d)), rng.uniform(-1, 1, (n2, d))
        ell = rng.uniform(0.7, 1.6, d) * np.sqrt(d / 4.0)
        for kind in (gpax_amd.RBF, gpax_amd.MATERN52):
            p = gpax_amd.make_params(kind, 0.9, ell, None, np.array([0.0]), 0.0, False)
            assert np.abs(core.gram(p, x1, x2) - orc.gram(kind, x1, x2, 0.9, ell, None, np.array([0.0]), 0.0, False)).max() <= 1e-12, (n1, n2, d, kind)


def test_gram_of_coincident_and_far_points(core):
    """The expansion form can come out a few 1e-16 below zero for coincident points (the reference clips, gpax/kernels.py:50):
    duplicates must give exactly k_scale (RBF) or the Matern value at r = 1e-6, far points underflow to zero without a NaN."""
    import gpax_amd

    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, (150, 8))
    x[75:] = x[:75]                         # every point twice
    for kind in (gpax_amd.RBF, gpax_amd.MATERN52):
        p = gpax_amd.make_params(kind, 2.0, 0.3, None, np.array([0.0]), 0.0, False)
        k = core.gram(p, x, x)
        assert np.isfinite(k).all() and (k >= 0).all()
        dup = k[np.arange(75), np.arange(75) + 75]
        if kind == gpax_amd.RBF:
            assert np.abs(dup - 2.0).max() <= 1e-14
        else:
            assert np.abs(dup - 2.0 * (1 + np.sqrt(5) * 1e-6) * np.exp(-np.sqrt(5) * 1e-6)).max() <= 1e-10
    far = np.vstack([np.zeros((1, 8)), np.full((1, 8), 1000.0)])
    k = core.gram(gpax_amd.make_params(gpax_amd.RBF, 1.0, 0.5, None, np.array([0.0]), 0.0, False), far, far)
    assert np.array_equal(k, np.eye(2))
