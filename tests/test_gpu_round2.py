"""GPU parity tests added in round 2 (``-m gpu``), all through the C ABI:

  * ill-conditioned / jitter-only regimes the reference's own tests use (``y_std = 0.0``, gpax/_tests/test_gp.py:10)
    at N = 1024 and 4096, with a conditioning bound the test computes;
  * L against the oracle at the bench size, K blocks at the C3 / C5 sizes;
  * the unconditioned path (gpax/gp.py:212-245) numerically, through ``gpx_prior_cov``;
  * the HIP posterior against the reference's DEFAULT branch, the explicit inverse (gpax/gp.py:303-310);
  * the fall-back the in-kernel hand-overs rely on (forced time-out -> GPX_ERR_RETRY -> kernel boundaries -> same bits);
  * two handles stepping concurrently from two threads;
  * device residency keyed on content, not on ``id()``;
  * batched hyper-parameter draws (gpax's vmap, gpax/gp.py:354-364, 384-390): bit-identical per draw to the single-draw
    path, parity per draw against the oracle; batched device-resident sampling.
"""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def core():
    import gpax_amd

    c = gpax_amd.Core(0)
    yield c
    c.close()


def test_runtime_is_the_one_the_library_was_built_against():
    """The GPU tests must run on the HIP runtime the library was compiled for (``gpx_runtime_info``): a process that
    imports PyTorch first would bind the library to torch's bundled runtime of another ROCm release."""
    from gpax_amd import _lib

    ok, text = _lib.runtime_info()
    assert ok, text
    assert "/opt/rocm" in text, text


def _cond_upper(k, diag_shift, iters=30):
    """kappa_2(K) <= lambda_max / (k_noise + k_jitter): K = K0 + shift * I with K0 positive semi-definite, so
    lambda_min >= shift (attained numerically for these smooth kernels); lambda_max by power iteration."""
    v = np.ones(k.shape[0]) / np.sqrt(k.shape[0])
    lam = 1.0
    for _ in range(iters):
        w = k @ v
        lam = float(np.linalg.norm(w))
        v = w / lam
    return lam / diag_shift


@pytest.mark.parametrize("n", [1024, 4096])
@pytest.mark.parametrize("k_scale,k_length,k_noise", [(1.0, 1.0, 0.0), (1.0, 5.0, 0.0), (100.0, 1.0, 0.0), (100.0, 5.0, 1e-3)])
def test_ill_conditioned_parity(core, n, k_scale, k_length, k_noise):
    """Jitter-only diagonals (``y_std = 0.0`` in the reference's own test matrix -> diagonal rule adds 1e-6 only), long
    length-scales and large scales: kappa(K) up to ~1e12.  K must agree to 1e-12 * k_scale regardless.  L, alpha and
    the posterior are compared under a bound the test computes from kappa: two backward-stable factorisations of the
    same matrix differ by at most ~ kappa * eps * |L| (Higham, Accuracy and Stability, Thm 10.8); the explicit
    W_k = inv(L_kk) solves of the HIP path add kappa(L_kk) * eps per block, which the same bound covers."""
    import gpax_amd
    from oracle import gp_oracle as orc

    d, m = 8, 64
    x, y, xn, p = orc.synthetic_problem(n, d, m, seed=n)
    p = dict(p, k_scale=np.array([k_scale]), k_length=np.array([k_length]), k_noise=np.array([k_noise]))
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_option("nb", 0)
    core.set_train(x)
    core.set_test(xn)
    info, k, l = core.fit_factor(prm, want_k=True, want_l=True)
    assert info == 0
    k_ref = orc.rbf(x, x, p["k_scale"], p["k_length"], p["k_noise"], p["k_jitter"])
    assert np.abs(k - k_ref).max() <= 1e-12 * k_scale
    l_ref = orc.cholesky_lower(k_ref)
    kappa = _cond_upper(k_ref, k_noise + 1e-6)
    lmax = np.abs(l_ref).max()
    err_l = np.abs(l - l_ref).max()
    # measured on MI355X (round 2): |dL|max <= 2.4e-11 in every regime below, i.e. the north-star 1e-10 holds even at
    # kappa ~ 5e10; the kappa * eps * |L|max figure printed beside it is the worst case the theory allows
    bound = 1e-10
    print(f"\n[ill-cond] n={n} scale={k_scale} len={k_length} noise={k_noise}: kappa<={kappa:.2e} |dL|max={err_l:.2e} "
          f"(kappa*eps*|L|max={kappa * EPS * lmax:.2e}) |L L^T - K|max={np.abs(l @ l.T - k).max():.2e}")
    assert err_l <= bound
    # backward stability does not degrade with kappa: the factor reproduces K to a few ulps of its scale
    assert np.abs(l @ l.T - k).max() <= 64 * EPS * k_scale * np.sqrt(n)
    r = y - 1.0
    info, logp, alpha = core.log_marginal(r, want_alpha=True)
    logp_ref, _, alpha_ref = orc.mvn_log_prob(y, np.ones(n), k_ref)
    # the log-marginal is dominated by the quadratic form r^T K^-1 r here (~ |r|^2 / 1e-6): relative error ~ kappa * eps
    rel = abs(logp - logp_ref) / abs(logp_ref)
    err_a = np.abs(alpha - alpha_ref).max() / np.abs(alpha_ref).max()
    print(f"[ill-cond]   logp rel {rel:.2e}  alpha rel {err_a:.2e}")
    # measured: logp rel <= 5.1e-10 (kappa 9e8), alpha rel <= 1.8e-8 (kappa 3.7e9): 1e-10 / 1e-8 wherever kappa * eps allows
    # it, a small fraction of kappa * eps beyond
    assert info == 0 and rel <= max(1e-10, 0.02 * kappa * EPS)
    assert err_a <= max(1e-8, 0.1 * kappa * EPS)


def test_factor_matches_oracle_at_bench_size(core):
    """|L - L_oracle| <= 1e-10 and |K - K_oracle| <= 1e-12 at the bench workload itself (C2: N = 4096, d = 8, RBF)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(4096, 8, 1000, seed=0)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_option("nb", 0)
    core.set_train(x)
    info, k, l = core.fit_factor(prm, want_k=True, want_l=True)
    k_ref = orc.rbf(x, x, p["k_scale"], p["k_length"], p["k_noise"], p["k_jitter"])
    assert info == 0 and np.abs(k - k_ref).max() <= 1e-12
    assert np.abs(l - orc.cholesky_lower(k_ref)).max() <= 1e-10


@pytest.mark.parametrize("n,d,kind", [(16384, 16, 1), (8192, 2, 1)])
def test_gram_blocks_at_c3_c5_sizes(core, n, d, kind):
    """K of the C3 / C5 problems (Matern-5/2) on 512 random rows against the oracle, 1e-12; and at the C5 size the factor
    against the oracle's LAPACK factor, 1e-10."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, _, p = orc.synthetic_problem(n, d, 8, seed=3, kind=kind)
    prm = gpax_amd.make_params(kind, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_option("nb", 0)
    core.set_train(x)
    info, k, l = core.fit_factor(prm, want_k=True, want_l=(n <= 8192))
    assert info == 0
    rows = np.random.default_rng(7).choice(n, 512, replace=False)
    blk = orc.gram(kind, x[rows], x, p["k_scale"], p["k_length"], None, 0.0, 0.0, False)
    blk[np.arange(512), rows] += p["k_noise"][0] + p["k_jitter"][0]
    assert np.abs(k[rows] - blk).max() <= 1e-12
    if l is not None:
        k_ref = orc.gram(kind, x, x, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
        assert np.abs(l - orc.cholesky_lower(k_ref)).max() <= 1e-10


@pytest.mark.parametrize("observation_noise", [False, True])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_unconditioned_mean_and_covariance_match_oracle(observation_noise, kind):
    """``_get_mean_and_covariance_unconditioned`` (gpax/gp.py:212-245) against the oracle, ``observation_noise`` on and
    off, through ``gpx_prior_cov``; and the sampling path built on it keeps the reference's shapes."""
    import gpax_amd
    from gpax_amd.utils.prior_utils import LogNormal, Parameter
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(5)
    x_new = rng.uniform(-1, 1, (150, 3))
    kernel = [gpax_amd.RBFKernel(), gpax_amd.MaternKernel(), gpax_amd.PeriodicKernel(k_period=Parameter(LogNormal(0.0, 1.0), 1))][kind]
    gp = gpax_amd.ExactGP(kernel, observation_noise=observation_noise, hp_samples=3, gp_samples=4)
    params = {"k_scale": np.array([1.7]), "k_length": np.array([0.8]), "k_noise": np.array([0.03]), "k_jitter": np.array([1e-6])}
    if kind == 2:
        params["k_period"] = np.array([1.3])
    mean, cov = gp._get_mean_and_covariance_unconditioned(x_new, params)
    mean_ref, cov_ref = orc.mean_and_covariance_unconditioned(kind, x_new, params, observation_noise=observation_noise)
    assert np.abs(np.broadcast_to(mean, (150,)) - mean_ref).max() <= 1e-15
    assert cov.shape == (150, 150) and np.abs(cov - cov_ref).max() <= 1e-12
    expect = 1.7 + 1e-6 + (0.03 if observation_noise else 0.0)
    assert abs(cov[0, 0] - expect) <= 1e-12 + (1e-5 if kind == 1 else 0.0)
    s = gp.sample(x_new)
    assert s.y.shape == (3, 4, 150)


def test_posterior_against_the_reference_default_inverse_branch(core):
    """The reference's DEFAULT posterior forms K^-1 explicitly (``use_cholesky=False``, gpax/gp.py:303-310); the HIP path
    always takes the Cholesky branch (:295-301).  The two agree to ~kappa * eps: the gap between the HIP outputs and the
    oracle's inverse branch is measured and asserted here, beside the (much smaller) gap to the Cholesky branch."""
    import gpax_amd
    from oracle import gp_oracle as orc

    for n, noise in [(512, 0.006744897501960818), (1024, 0.006744897501960818), (512, 1e-4)]:
        x, y, xn, p = orc.synthetic_problem(n, 8, 200, seed=n)
        p = dict(p, k_noise=np.array([noise]))
        prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
        core.set_train(x)
        core.set_test(xn)
        core.fit_factor(prm)
        core.log_marginal(y - 1.0)
        _, mean, var, cov = core.posterior(np.ones(200), 0.0, want_cov=True)
        mean_c, cov_c = orc.mean_and_covariance(0, x, y, xn, p, use_cholesky=True)
        mean_i, cov_i = orc.mean_and_covariance(0, x, y, xn, p, use_cholesky=False)
        k_ref = orc.rbf(x, x, p["k_scale"], p["k_length"], p["k_noise"], p["k_jitter"])
        kappa = _cond_upper(k_ref, noise + 1e-6)
        gap_c = max(np.abs(mean - mean_c).max(), np.abs(cov - cov_c).max())
        gap_i = max(np.abs(mean - mean_i).max(), np.abs(cov - cov_i).max())
        print(f"\n[inv-branch] n={n} noise={noise:g}: kappa<={kappa:.2e}  HIP vs cholesky branch {gap_c:.2e}  HIP vs inverse branch {gap_i:.2e}"
              f"  (oracle cholesky vs inverse {max(np.abs(mean_c - mean_i).max(), np.abs(cov_c - cov_i).max()):.2e})")
        assert gap_c <= 1e-8
        assert gap_i <= max(1e-8, 64 * kappa * EPS)   # what separates the two reference branches themselves


def test_forced_handover_timeout_retries_with_kernel_boundaries():
    """``debug_force_timeout`` makes the first resident wait of the next factorisation give up at once: the kernel
    reports info = -7, the handle switches ``resident`` off, the synchronous entry point repeats the call by itself and
    returns the same bits as a handle that ran with kernel-boundary hand-overs from the start.  The asynchronous pair
    surfaces GPX_ERR_RETRY from ``gpx_fetch_results`` and ``Core.fetch_results`` re-issues the step."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(1500, 6, 90, seed=31)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    ref_core = gpax_amd.Core(0)
    ref_core.set_option("resident", 0)
    ref_core.set_train(x)
    ref_core.set_test(xn)
    ref = ref_core.fit_predict(prm, y - 1.0, np.ones(90), 0.0)
    ref_g = ref_core.lml_grad(prm, y - 1.0)
    ref_core.close()
    for mode in ("sync", "async", "grad"):
        c = gpax_amd.Core(0)
        assert c.get_option("resident") == 2
        c.set_train(x)
        c.set_test(xn)
        c.set_option("debug_force_timeout", 1)
        if mode == "sync":
            got = c.fit_predict(prm, y - 1.0, np.ones(90), 0.0)
        elif mode == "async":
            c.set_targets(y - 1.0, np.ones(90))
            c.fit_predict_async(prm, 0.0)
            got = c.fetch_results()
        else:
            g = c.lml_grad(prm, y - 1.0)
            assert g[0] == 0 and g[1] == ref_g[1] and all(np.array_equal(g[2][k], ref_g[2][k]) for k in g[2]) and np.array_equal(g[3], ref_g[3])
            got = c.fit_predict(prm, y - 1.0, np.ones(90), 0.0)
        assert c.get_option("resident") == 0 and c.get_option("debug_force_timeout") == 0
        assert got[0] == ref[0] == 0 and got[1] == ref[1] and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
        got2 = c.fit_predict(prm, y - 1.0, np.ones(90), 0.0)   # and it stays usable
        assert got2[1] == ref[1]
        c.close()
    logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
    assert abs(ref[1] - logp_ref) <= 1e-10 * abs(logp_ref) and np.abs(ref[2] - mean_ref).max() <= 1e-8


def test_two_handles_stepping_concurrently_from_two_threads():
    """Two ``Core``s (multi-chain NUTS, several models) share the GPU's hardware queues: stepped concurrently from two
    host threads, each must return exactly what it returns alone -- whether or not a resident wait times out under the
    contention (the retry is transparent)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    probs = [orc.synthetic_problem(1100, 4, 60, seed=1), orc.synthetic_problem(1500, 6, 90, seed=2)]
    cores, prms, refs = [], [], []
    for x, y, xn, p in probs:
        c = gpax_amd.Core(0)
        c.set_train(x)
        c.set_test(xn)
        prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
        cores.append(c); prms.append(prm)
        refs.append((c.fit_predict(prm, y - 1.0, np.ones(xn.shape[0]), 0.0), c.lml_grad(prm, y - 1.0)))
    errors = []

    def work(i):
        x, y, xn, p = probs[i]
        try:
            for it in range(25):
                got = cores[i].fit_predict(prms[i], y - 1.0, np.ones(xn.shape[0]), 0.0)
                assert got[0] == 0 and got[1] == refs[i][0][1], (i, it, got[1], refs[i][0][1])
                assert np.array_equal(got[2], refs[i][0][2]) and np.array_equal(got[3], refs[i][0][3]), (i, it)
                g = cores[i].lml_grad(prms[i], y - 1.0)
                assert g[1] == refs[i][1][1] and np.array_equal(g[3], refs[i][1][3]), (i, it)
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for c in cores:
        c.close()
    assert not errors, errors
    for (x, y, xn, p), (fp, _) in zip(probs, refs):
        logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
        assert abs(fp[1] - logp_ref) <= 1e-10 * abs(logp_ref) and np.abs(fp[2] - mean_ref).max() <= 1e-8


def test_models_of_one_shape_never_see_each_others_inputs():
    """Device residency is keyed on a never-reused model token and the CONTENT of x (not on ``id()``, which CPython
    recycles): models of one shape built, used and deleted in sequence -- and a model whose x is edited in place -- each
    get the oracle's posterior for their own data."""
    import gc

    import gpax_amd
    from gpax_amd import state
    from oracle import gp_oracle as orc

    state.set_rng_key(7)
    rng = np.random.default_rng(0)
    x_new = rng.uniform(0, 1, (40, 2))
    params = {"k_scale": np.array([1.0]), "k_length": np.array([0.7]), "k_noise": np.array([0.01]), "k_jitter": np.array([1e-6])}

    def check(gp, x, y):
        mean, cov = gp._get_mean_and_covariance(x_new, params)
        mean_ref, cov_ref = orc.mean_and_covariance(0, x, y, x_new, params)
        assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(cov - cov_ref).max() <= 1e-8

    for trial in range(6):
        x = rng.uniform(0, 1, (130, 2))
        y = np.sin(4 * x.sum(1)) + 0.05 * rng.standard_normal(130)
        gp = gpax_amd.ExactGP(gpax_amd.RBFKernel(), x, y, input_transform=None, output_transform=None)
        check(gp, x, y)
        if trial == 3:   # in-place edit of the inputs and of the targets
            x += 0.05
            check(gp, x, y)
            y *= 1.5
            check(gp, x, y)
        del gp
        gc.collect()


# ----------------------------------------------------------------------------- batched hyper-parameter draws

def _draws(rng, b, d, ard):
    out = []
    for _ in range(b):
        ell = np.exp(0.3 * rng.standard_normal(d if ard else 1))
        out.append(dict(k_scale=np.array([float(np.exp(0.3 * rng.standard_normal()))]), k_length=ell,
                        k_noise=np.array([0.002 + 0.02 * rng.uniform()]), k_jitter=np.array([1e-6])))
    return out


@pytest.mark.parametrize("n,d,m,kind,ard,b", [(25, 1, 30, 0, False, 5), (200, 2, 77, 1, False, 4), (256, 3, 64, 2, True, 3),
                                               (640, 4, 129, 0, True, 6), (1333, 5, 77, 0, False, 4), (2000, 8, 100, 1, False, 3),
                                               (4096, 8, 1000, 0, False, 8)])
def test_batched_draws_are_bit_identical_to_single_draws(core, n, d, m, kind, ard, b):
    """``gpx_fit_predict_batch``: B draws in ONE schedule.  Each draw's log-marginal, posterior mean and variance carry
    the SAME BITS as a ``gpx_fit_predict`` call with that draw's parameters (the schedule -- which kernel touches which
    tile at which step -- does not depend on the batch size), and match the oracle per draw."""
    import gpax_amd
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(n + b)
    x, y, xn, p0 = orc.synthetic_problem(n, d, m, seed=n, kind=kind)
    draws = _draws(rng, b, d, ard)
    per = [np.exp(0.2 * rng.standard_normal(1)) for _ in range(b)] if kind == 2 else [None] * b
    means = rng.uniform(0.5, 1.5, b)                     # a different prior mean per draw (ConstantMean with a sampled m_mean)
    plist = [gpax_amd.make_params(kind, q["k_scale"], q["k_length"], per[i], q["k_noise"], q["k_jitter"]) for i, q in enumerate(draws)]
    r = np.stack([y - means[i] for i in range(b)])
    m_new = np.stack([np.full(m, means[i]) for i in range(b)])
    new_noise = np.array([q["k_noise"][0] if i % 2 else 0.0 for i, q in enumerate(draws)])
    core.set_option("nb", 0)
    core.set_train(x)
    core.set_test(xn)
    # Two schedules: "batch_group" 0 = the single-draw grouping thresholds (the bits of DEFAULT single-draw calls); the
    # default throughput mode for >= 4 draws groups the bulk updates four panels at a time throughout, which is what a
    # single-draw call does with pair_min = 1, quad_min = 2.
    for batch_group, single_opts in ((0, {}), (4, {"pair_min": 1, "quad_min": 2} if b >= 4 else {})):
        core.set_option("batch_group", batch_group)
        info, logp, mean, var = core.fit_predict_batch(plist, r, m_new, new_noise)
        assert (info == 0).all()
        info2, logp2, mean2, var2 = core.fit_predict_batch(plist, r, m_new, new_noise)
        assert np.array_equal(logp, logp2) and np.array_equal(mean, mean2) and np.array_equal(var, var2)   # repeatable, bit for bit
        try:
            for name, v in single_opts.items():
                core.set_option(name, v)
            for i in range(b):
                i1, l1, m1, v1 = core.fit_predict(plist[i], r[i], m_new[i], float(new_noise[i]))
                assert i1 == 0 and l1 == logp[i], (batch_group, i, l1, logp[i])
                assert np.array_equal(m1, mean[i]) and np.array_equal(v1, var[i]), (batch_group, i)
                if n <= 2000 or i < 2:
                    q = dict(draws[i])
                    if kind == 2:
                        q["k_period"] = per[i]
                    logp_ref, mean_ref, var_ref = orc.fit_predict(kind, x, y, xn, q, mean_value=means[i], observation_noise=bool(i % 2))
                    assert abs(l1 - logp_ref) <= 1e-10 * abs(logp_ref)
                    assert np.abs(m1 - mean_ref).max() <= 1e-8 * max(1.0, np.abs(mean_ref).max()) and np.abs(v1 - var_ref).max() <= 1e-8
        finally:
            core.set_option("pair_min", 28)
            core.set_option("quad_min", 40)
            core.set_option("batch_group", 4)


def test_batched_draws_report_info_per_draw(core):
    """One draw of the batch is not positive definite (a negative per-point noise): its info is the LAPACK index and its
    outputs are NaN; the other draws are untouched."""
    import gpax_amd

    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (300, 2))
    xn = rng.uniform(-1, 1, (20, 2))
    good = np.full(300, 0.01)
    bad = good.copy()
    bad[137] = -5.0
    plist = [gpax_amd.make_params(0, 1.0, 1.0, None, v, 1e-6) for v in (good, bad, good)]
    r = rng.standard_normal((3, 300))
    core.set_train(x)
    core.set_test(xn)
    info, logp, mean, var = core.fit_predict_batch(plist, r, None, None)
    assert list(info) == [0, 138, 0]
    assert np.isfinite(logp[[0, 2]]).all() and np.isfinite(mean[[0, 2]]).all() and not np.isfinite(logp[1])
    i0, l0, m0, v0 = core.fit_predict(plist[0], r[0], None, 0.0)
    assert l0 == logp[0] and np.array_equal(m0, mean[0])


@pytest.mark.parametrize("conditioned", [True, False])
@pytest.mark.parametrize("n,m,d,kind,b", [(130, 50, 2, 0, 4), (700, 300, 3, 1, 3)])
def test_batched_device_resident_sampling_matches_oracle(core, conditioned, n, m, d, kind, b):
    """``gpx_sample_batch``: per draw the posterior (or prior) mean, sd = sqrt(diag cov) and mean + eps chol(cov)^T with
    the covariance and its factor kept in HBM (gpax/gp.py:314-338 under the vmap of :352-391), against the oracle's
    ``mean_and_covariance[_unconditioned]`` + ``eps @ chol(cov).T`` with the SAME eps."""
    import gpax_amd
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(n + m)
    x, y, xn, _ = orc.synthetic_problem(n, d, m, seed=n, kind=kind)
    draws = _draws(rng, b, d, False)
    plist = [gpax_amd.make_params(kind, q["k_scale"], q["k_length"], None, q["k_noise"], q["k_jitter"]) for q in draws]
    obs = True
    nsamp = 7
    eps = rng.standard_normal((b, nsamp, m))
    r = np.stack([y - 1.0] * b)
    m_new = np.ones((b, m))
    new_noise = np.array([q["k_noise"][0] for q in draws]) if obs else None
    core.set_train(x)
    core.set_test(xn)
    info, mean, sd, smp = core.sample_batch(plist, r if conditioned else None, m_new, new_noise, eps, conditioned=conditioned)
    assert (info == 0).all() and smp.shape == (b, nsamp, m)
    for i, q in enumerate(draws):
        prm = dict(q)
        if conditioned:
            mean_ref, cov_ref = orc.mean_and_covariance(kind, x, y, xn, prm, observation_noise=obs)
        else:
            mean_ref, cov_ref = orc.mean_and_covariance_unconditioned(kind, xn, prm, observation_noise=obs)
            mean_ref = np.broadcast_to(mean_ref, (m,))
        assert np.abs(mean[i] - mean_ref).max() <= 1e-8
        assert np.abs(sd[i] - np.sqrt(np.diag(cov_ref))).max() <= 1e-8
        ref = mean_ref[None, :] + eps[i] @ orc.cholesky_lower(cov_ref).T
        # chol(cov) amplifies differences in cov by ~1/sqrt(lambda_min(cov)) (>= 1/sqrt(noise + jitter))
        assert np.abs(smp[i] - ref).max() <= 1e-6, np.abs(smp[i] - ref).max()


def test_device_acquisition_epilogue_matches_oracle_and_host_path():
    """EI / UCB as a device epilogue over the resident (mu, sd) (``gpx_acquisition``) with the default Scale / Normalize
    transforms: the values against the oracle's formulae on ``predict(fast=True)`` outputs, the arg-max, and the Halton
    optimiser evaluating all its candidates in ONE posterior call against the same candidates pushed through in the
    reference's batches of 100 (gpax/acquisition.py:166-216)."""
    import gpax_amd
    from gpax_amd import state
    from gpax_amd.acquisition import ExpectedImprovement, UpperConfidenceBound
    from oracle import gp_oracle as orc
    from scipy.stats import qmc

    rng = np.random.default_rng(4)
    x = rng.uniform(0.0, 3.0, (300, 2))
    y = 5.0 + 2.0 * np.sin(2 * x[:, 0]) * np.cos(x[:, 1]) + 0.05 * rng.standard_normal(300)
    gp = gpax_amd.ExactGP(kernel=gpax_amd.MaternKernel(), x=x, y=y)          # default transforms
    bounds = np.array([[0.0, 0.0], [3.0, 3.0]])
    cand = rng.uniform(0.0, 3.0, (1234, 2))
    for acq, ref in ((ExpectedImprovement(bounds), lambda mu, sd: orc.expected_improvement(mu, sd, y.max())),
                     (UpperConfidenceBound(bounds, beta=4.0), lambda mu, sd: orc.upper_confidence_bound(mu, sd, 4.0)),
                     (UpperConfidenceBound(bounds, beta=0.0), lambda mu, sd: mu),
                     (UpperConfidenceBound(bounds, beta=float("inf")), lambda mu, sd: sd)):
        state.set_rng_key(11)
        vals = acq(gp, cand)
        state.set_rng_key(11)
        mu, sd = gp.predict(cand, fast=True)
        want = ref(mu, sd)
        assert vals.shape == (1234,) and np.abs(vals - want).max() <= 1e-11 * max(1.0, np.abs(want).max())
        state.set_rng_key(11)
        _, idx, best = gp._acquisition_fast(cand, acq.DEVICE_KIND, acq._device_param(gp), want_values=False)
        # (this second call meets the cached factor and takes the solve path of gpx_posterior: same numbers to rounding)
        assert idx == int(np.argmax(vals)) and abs(best - vals[idx]) <= 1e-11 * max(1.0, abs(vals[idx]))
    # the optimiser: one call for all candidates == the reference's batched evaluation of the same candidates
    ei = ExpectedImprovement(bounds)
    state.set_rng_key(21)
    x_star, val = ei.optimize(gp, 1500)
    state.set_rng_key(21)
    key, _ = state.get_rng_key()
    samples = qmc.scale(qmc.Halton(d=2, seed=key).random(n=1500), bounds[0], bounds[1])
    vals = ei(gp, samples)                       # same key sequence: one more get_rng_key inside
    assert np.array_equal(x_star, samples[int(np.argmax(vals))][None, :]) and abs(val - vals.max()) <= 1e-11 * max(1.0, abs(val))
    host = ExpectedImprovement(bounds, fast=True)
    host.DEVICE_KIND = None                      # the host formula on predict() outputs, in batches of 100
    state.set_rng_key(21)
    x_host, val_host = host.optimize(gp, 1500)   # (re-draws the prior medians per batch, like the reference: compare loosely)
    assert x_host.shape == x_star.shape and np.isfinite(val_host)


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [1, 16, 17, 100, 128, 129, 144, 255, 256, 300, 511, 640, 700])
def test_first_failing_pivot_is_reported_wherever_it_falls(core, bad):
    """LAPACK-style info = the order of the first leading minor that is not positive definite, whichever wave of the
    dataflow potrf_diag (one wave per 16 x 16 tile row, each running its own leaf) or whichever diagonal block meets it:
    first / last column of a tile, of a 128-block, of the matrix (the reference returns NaNs silently)."""
    import gpax_amd

    n = 700
    rng = np.random.default_rng(bad)
    x = rng.uniform(-1, 1, (n, 2))
    noise = np.full(n, 0.01)
    noise[bad - 1] = -5.0   # diagonal entry 1 - 5 + jitter < 0
    core.set_train(x)
    prm = gpax_amd.make_params(0, 1.0, 1.0, None, noise, 1e-6)
    info, _, l = core.fit_factor(prm, want_l=True)
    assert info == bad
    assert np.isfinite(l[: bad - 1, : bad - 1]).all()
