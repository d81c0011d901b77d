"""GPU parity tests proper (``-m gpu``): the HIP path, called through the C ABI
(``gpax_amd._lib.Core`` -> ``libgpx_hip.so``), against the committed golden fixtures
and the oracle on seeded inputs.

Tolerances (fp64; ``north_star``: 1e-10 on K, L and the log-marginal):
  K            max-abs <= 1e-12          (measured ~1e-15: device exp/sin vs libm)
  L            max-abs <= 1e-10
  log-marginal relative <= 1e-10         (its magnitude is O(N))
  alpha, posterior mean / cov            <= 1e-8 * scale  (cond(K)*eps amplification)
"""

import numpy as np
import pytest

from conftest import GOLDEN_NAMES, load_golden

pytestmark = pytest.mark.gpu

TOL_K = 1e-12
TOL_L = 1e-10
TOL_LOGP_REL = 1e-10
TOL_POST = 1e-8


@pytest.fixture(scope="module")
def core():
    import gpax_amd

    c = gpax_amd.Core(0)
    yield c
    c.close()


def _params(g, noise=True, apply_noise=True):
    import gpax_amd

    p = g["params"]
    k_noise = None
    if noise:
        k_noise = g["y_std"] ** 2 if g["y_std"] is not None else p["k_noise"]
    return gpax_amd.make_params(
        g["kind"], p["k_scale"], p["k_length"], p.get("k_period"), k_noise, p["k_jitter"], apply_noise
    )


def _new_noise(g):
    return float(g["params"]["k_noise"][0]) if g["observation_noise"] else 0.0


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_gram_matches_golden(core, name):
    g = load_golden(name)
    k = core.gram(_params(g), g["x"], g["x"])
    assert np.abs(k - g["K"]).max() <= TOL_K
    kc = core.gram(_params(g, noise=False, apply_noise=False), g["x_new"], g["x"])
    assert kc.shape == g["K_cross"].shape
    assert np.abs(kc - g["K_cross"]).max() <= TOL_K


@pytest.mark.parametrize("nb", [64, 128])
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_factor_lml_posterior_match_golden(core, name, nb):
    g = load_golden(name)
    core.set_option("nb", nb)
    core.set_train(g["x"])
    core.set_test(g["x_new"])
    info, k, l = core.fit_factor(_params(g), want_k=True, want_l=True)
    assert info == 0
    assert np.abs(k - g["K"]).max() <= TOL_K
    assert np.abs(l - g["L"]).max() <= TOL_L
    info, logp, alpha = core.log_marginal(g["y"] - g["m_train"], want_alpha=True)
    assert info == 0
    assert abs(logp - float(g["logp"])) <= TOL_LOGP_REL * abs(float(g["logp"]))
    assert np.abs(alpha - g["alpha"]).max() <= TOL_POST * max(1.0, np.abs(g["alpha"]).max())
    info, mean, var, cov = core.posterior(g["m_new"], _new_noise(g), want_cov=True)
    scale = max(1.0, np.abs(g["post_mean"]).max())
    assert np.abs(mean - g["post_mean"]).max() <= TOL_POST * scale
    assert np.abs(cov - g["post_cov"]).max() <= TOL_POST
    assert np.abs(var - np.diag(g["post_cov"])).max() <= TOL_POST


@pytest.mark.parametrize("lookahead", [0, 1])
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_fused_fit_predict_matches_golden(core, name, lookahead):
    """lookahead 0: everything on one stream; 1: the split-panel schedule (the default: the chain stream carries only
    the diagonal band)."""
    g = load_golden(name)
    core.set_option("nb", 0)   # automatic again
    core.set_option("lookahead", lookahead)
    core.set_train(g["x"])
    core.set_test(g["x_new"])
    info, logp, mean, var = core.fit_predict(_params(g), g["y"] - g["m_train"], g["m_new"], _new_noise(g))
    core.set_option("lookahead", 1)
    assert info == 0
    assert abs(logp - float(g["logp"])) <= TOL_LOGP_REL * abs(float(g["logp"]))
    scale = max(1.0, np.abs(g["post_mean"]).max())
    assert np.abs(mean - g["post_mean"]).max() <= TOL_POST * scale
    assert np.abs(var - np.diag(g["post_cov"])).max() <= TOL_POST
    # the fused call leaves a cached factor: the separate posterior must agree with it
    info, mean2, var2, _ = core.posterior(g["m_new"], _new_noise(g))
    assert np.abs(mean2 - mean).max() <= 1e-10 * scale and np.abs(var2 - var).max() <= 1e-10


def test_gram_edge_shapes(core):
    """Ragged / odd / rectangular shapes, d spanning the LDS chunk (16), the diagonal
    rule keyed on shapes (gpax/kernels.py:124), vector noise, empty inputs."""
    import gpax_amd
    from oracle import gp_oracle as orc

    rng = np.random.default_rng(0)
    for n1, n2, d in [(1, 1, 1), (7, 5, 3), (65, 63, 17), (130, 130, 33), (64, 129, 2), (200, 1, 16)]:
        x1, x2 = rng.uniform(-1, 1, (n1, d)), rng.uniform(-1, 1, (n2, d))
        ell = rng.uniform(0.5, 2.0, d)
        for kind, per in ((gpax_amd.RBF, None), (gpax_amd.MATERN52, None), (gpax_amd.PERIODIC, rng.uniform(0.5, 2, d))):
            p = gpax_amd.make_params(kind, 1.7, ell, per, np.array([0.3]), 1e-6, True)
            k = core.gram(p, x1, x2)
            ref = orc.gram(kind, x1, x2, 1.7, ell, per, np.array([0.3]), 1e-6, True)
            assert k.shape == (n1, n2)
            assert np.abs(k - ref).max() <= TOL_K, (n1, n2, d, kind)
    # same shape, different points: the rule still fires; vector noise lands on the diagonal
    x1, x2 = rng.uniform(-1, 1, (50, 4)), rng.uniform(-1, 1, (50, 4))
    v = rng.uniform(0.1, 0.3, 50)
    k = core.gram(gpax_amd.make_params(gpax_amd.RBF, 1.0, 1.0, None, v, 1e-3, True), x1, x2)
    assert np.abs(k - orc.rbf(x1, x2, 1.0, 1.0, v, 1e-3, True)).max() <= TOL_K
    k = core.gram(gpax_amd.make_params(gpax_amd.RBF, 1.0, 1.0, None, v, 1e-3, False), x1, x2)
    assert np.abs(k - orc.rbf(x1, x2, 1.0, 1.0, apply_noise=False)).max() <= TOL_K
    # empty
    assert core.gram(gpax_amd.make_params(gpax_amd.RBF, 1.0, 1.0), np.zeros((0, 3)), np.zeros((4, 3))).shape == (0, 4)


@pytest.mark.parametrize("n,d,kind", [(1, 1, 0), (2, 1, 1), (100, 2, 0), (1000, 8, 0), (1500, 16, 1), (777, 3, 2)])
def test_against_oracle_seeded(core, n, d, kind):
    """Sizes the oracle finishes in well under a second, incl. N not a multiple of the panel."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(n, d, 37, seed=n + d, kind=kind)
    core.set_option("nb", 0)   # automatic again
    core.set_train(x)
    core.set_test(xn)
    prm = gpax_amd.make_params(kind, p["k_scale"], p["k_length"], p.get("k_period"), p["k_noise"], p["k_jitter"])
    info, logp, mean, var = core.fit_predict(prm, y - 1.0, np.ones(37), 0.0)
    assert info == 0
    logp_ref, mean_ref, var_ref = orc.fit_predict(kind, x, y, xn, p)
    assert abs(logp - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= TOL_POST * max(1.0, np.abs(mean_ref).max())
    assert np.abs(var - var_ref).max() <= TOL_POST
    info, k, l = core.fit_factor(prm, want_k=True, want_l=True)
    kp = {a: b for a, b in p.items() if a not in ("k_noise", "k_jitter")}
    k_ref = orc.gram(kind, x, x, k_noise=p["k_noise"], k_jitter=p["k_jitter"], **kp)
    assert np.abs(k - k_ref).max() <= TOL_K
    assert np.abs(l - orc.cholesky_lower(k_ref)).max() <= TOL_L


def test_split_schedule_is_deterministic(core):
    """No tile is ever updated by two streams at once: repeated runs of the split-panel schedule at N=2000
    (16 panels, ragged) are bit-identical, and agree with the one-stream schedule and the oracle to rounding
    (the schedules differ in which GEMM variant updates a column block, so not bitwise)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(2000, 8, 100, seed=21)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_train(x)
    core.set_test(xn)
    core.set_option("lookahead", 0)
    _, _, l_ref = core.fit_factor(prm, want_l=True)
    core.set_option("lookahead", 1)
    _, _, l0 = core.fit_factor(prm, want_l=True)
    for _ in range(4):
        _, _, l = core.fit_factor(prm, want_l=True)
        assert np.array_equal(l, l0)
    assert np.abs(l0 - l_ref).max() <= 1e-12
    assert np.abs(l0 - orc.cholesky_lower(orc.rbf(x, x, 1.0, 1.0, p["k_noise"], 1e-6))).max() <= TOL_L


@pytest.mark.parametrize("resident", [3, 2, 1, 0])
@pytest.mark.parametrize("n,m,grouped", [(200, 250, False), (1100, 90, False), (1333, 77, True)])
def test_handover_modes_are_bit_identical(core, resident, n, m, grouped):
    """The in-kernel hand-overs of the split-panel schedule (resident 3: ONE potrf_diag workgroup walks the whole diagonal
    and the fused band kernel sleeps on device flags; 2, the default: one potrf_diag launch per block, started resident;
    1: potrf_diag only; 0: every hand-over is a kernel boundary, which is also the mode a handle falls back to after a
    timed-out wait) only move WHEN a kernel runs: all four return the very same bits -- and the oracle's numbers.
    (The reference bits are taken with resident = 0.)"""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(n, 4, m, seed=n)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    if grouped:
        core.set_option("nb", 64)
        core.set_option("pair_min", 6)
        core.set_option("quad_min", 12)
    core.set_option("split_min_blocks", 2)   # two block columns run on one stream by default: keep the split machinery covered there
    core.set_train(x)
    core.set_test(xn)
    try:
        core.set_option("resident", 0)
        ref = core.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
        core.set_option("resident", resident)
        for _ in range(3):
            got = core.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
            assert got[0] == ref[0] == 0 and got[1] == ref[1]
            assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    finally:
        core.set_option("split_min_blocks", 3)
        core.set_option("resident", 2)
        core.set_option("nb", 0)
        core.set_option("pair_min", 28)
        core.set_option("quad_min", 40)
    logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
    assert abs(ref[1] - logp_ref) <= 1e-10 * abs(logp_ref)
    assert np.abs(ref[2] - mean_ref).max() <= 1e-8 and np.abs(ref[3] - var_ref).max() <= 1e-8


@pytest.mark.parametrize("pair_min,quad_min", [(0, 0), (1, 0), (6, 0), (13, 0), (1, 1), (4, 10), (0, 7)])
def test_grouped_bulk_updates_match_the_oracle(core, pair_min, quad_min):
    """The split-panel schedule groups the bulk updates (the last 2 or 4 panels in one K = 2 nb / 4 nb pass)
    while more than pair_min / quad_min block columns remain: never grouped, always paired, always in fours and
    several switch points all give the same factor, posterior and log-marginal."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(1333, 5, 77, seed=13)      # 21 panels of 64, ragged
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_option("nb", 64)
    core.set_option("pair_min", pair_min)
    core.set_option("quad_min", quad_min)
    core.set_train(x)
    core.set_test(xn)
    _, _, l = core.fit_factor(prm, want_l=True)
    info, logp, mean, var = core.fit_predict(prm, y - 1.0, np.ones(77), 0.0)
    core.set_option("nb", 0)   # automatic again
    core.set_option("pair_min", 28)
    core.set_option("quad_min", 40)
    assert info == 0
    k_ref = orc.rbf(x, x, p["k_scale"], p["k_length"], p["k_noise"], p["k_jitter"])
    assert np.abs(l - orc.cholesky_lower(k_ref)).max() <= TOL_L
    logp_ref, mean_ref, var_ref = orc.fit_predict(orc.RBF, x, y, xn, p)
    assert abs(logp - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= TOL_POST and np.abs(var - var_ref).max() <= TOL_POST


@pytest.mark.parametrize("option,value", [("resident", 0), ("resident", 1), ("resident", 3)])
def test_optional_schedule_features_match_the_default(core, option, value):
    """The in-kernel hand-overs of the chain (default resident = 2: potrf_diag and the band kernel start resident and sleep on
    flags raised by their producers) against kernel-boundary hand-overs (0), the half-way mode (1) and the one-workgroup diagonal (3):
    same bits as the default schedule, for the factor as well as the fused step."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(1500, 6, 90, seed=31)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_train(x)
    core.set_test(xn)
    _, _, l0 = core.fit_factor(prm, want_l=True)
    info0, logp0, mean0, var0 = core.fit_predict(prm, y - 1.0, np.ones(90), 0.0)
    core.set_option(option, value)
    for _ in range(3):
        _, _, l1 = core.fit_factor(prm, want_l=True)
        info1, logp1, mean1, var1 = core.fit_predict(prm, y - 1.0, np.ones(90), 0.0)
        assert info0 == 0 and info1 == 0
        assert np.array_equal(l0, l1) and logp0 == logp1 and np.array_equal(mean0, mean1) and np.array_equal(var0, var1)
    core.set_option(option, 2)
    logp_ref, mean_ref, var_ref = orc.fit_predict(orc.RBF, x, y, xn, p)
    assert abs(logp1 - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)


def test_grouped_schedule_at_default_thresholds_matches_the_oracle(core):
    """N = 6300 -> 50 block columns of 128: with the default thresholds the bulk updates run four panels at a time
    (k < 4), then two (k < 18), then one -- the schedule large problems get -- against the oracle's LAPACK path."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(6300, 8, 257, seed=41)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    assert core.get_option("pair_min") == 28 and core.get_option("quad_min") == 40 and core.get_option("nb") == 128
    core.set_train(x)
    core.set_test(xn)
    info, logp, mean, var = core.fit_predict(prm, y - 1.0, np.ones(257), 0.0)
    assert info == 0
    logp_ref, mean_ref, var_ref = orc.fit_predict(orc.RBF, x, y, xn, p)
    assert abs(logp - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= TOL_POST and np.abs(var - var_ref).max() <= TOL_POST
    info, logp_g, grads, alpha = core.lml_grad(prm, y - 1.0)      # the gradient path groups down to 8 columns
    assert info == 0 and abs(logp_g - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
    assert all(np.isfinite(v).all() for v in grads.values())


def test_full_size_properties(core):
    """BASELINE config C2 (N=4096, d=8, RBF, M=1000) through size-independent properties:
    K symmetric, L L^T = K, logp and posterior against the oracle (LAPACK, ~1 s)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    n, d, m = 4096, 8, 1000
    x, y, xn, p = orc.synthetic_problem(n, d, m, seed=0)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_option("nb", 0)   # automatic again
    core.set_train(x)
    core.set_test(xn)
    info, k, l = core.fit_factor(prm, want_k=True, want_l=True)
    assert info == 0
    assert np.abs(k - k.T).max() <= 1e-15
    assert np.abs(np.diag(k) - (1.0 + p["k_noise"][0] + 1e-6)).max() <= 1e-15
    assert np.abs(l @ l.T - k).max() <= 1e-12
    assert np.array_equal(l, np.tril(l))
    info, logp, mean, var = core.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
    logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
    assert abs(logp - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= TOL_POST
    assert np.abs(var - var_ref).max() <= TOL_POST
    assert (var > 0).all()


def test_not_positive_definite_reports_info(core):
    """JAX's Cholesky returns NaNs silently on a non-PD matrix; the C ABI additionally
    returns LAPACK-style info > 0."""
    import gpax_amd

    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (200, 2))
    noise = np.full(200, 0.01)
    noise[137] = -5.0  # diagonal entry 1 - 5 + jitter < 0: the leading minor of order 138 is not PD
    core.set_train(x)
    prm = gpax_amd.make_params(0, 1.0, 1.0, None, noise, 1e-6)
    info, _, l = core.fit_factor(prm, want_l=True)
    assert info == 138
    assert np.isfinite(l[:137]).all() and not np.isfinite(l[137:, 137:]).all()


def test_call_order_errors(core):
    import gpax_amd

    c = gpax_amd.Core(0)
    with pytest.raises(gpax_amd.GpxError):
        c.fit_factor(gpax_amd.make_params(0, 1.0, 1.0))  # no training inputs
    c.set_train(np.zeros((4, 2)) + np.arange(4)[:, None])
    with pytest.raises(gpax_amd.GpxError):
        c.log_marginal(np.zeros(4))  # no factor yet
    with pytest.raises(gpax_amd.GpxError):
        c.fit_factor(gpax_amd.make_params(0, 1.0, np.ones(3)))  # k_length has neither 1 nor d entries
    with pytest.raises(gpax_amd.GpxError):
        c.fit_factor(gpax_amd.make_params(2, 1.0, 1.0))  # periodic without k_period (reference quirk F7)
    c.close()


def _solve_identity(core, kind, x, r, prm, noise_plus_jitter, n_probe, m_total, seed):
    """Size-independent check of factorisation + solves + cross-gram + posterior mean: predicting AT training
    points x_i gives mean_i - m_i = k(x_i, X) alpha = r_i - (k_noise + k_jitter) alpha_i, because K alpha = r and
    the cross-covariance carries no diagonal term (gpax/gp.py:282, apply_noise=False).  alpha comes from
    gpx_lml_grad (U z, a different code path from the border rows that carry the posterior)."""
    rng = np.random.default_rng(seed)
    n = x.shape[0]
    idx = rng.choice(n, n_probe, replace=False)
    x_new = np.concatenate([x[idx], rng.uniform(x.min(), x.max(), (m_total - n_probe, x.shape[1]))])
    core.set_train(x)
    core.set_test(x_new)
    info, logp, mean, var = core.fit_predict(prm, r, np.zeros(m_total), 0.0)
    assert info == 0 and np.isfinite(logp)
    info2, logp2, mean2, var2 = core.fit_predict(prm, r, np.zeros(m_total), 0.0)
    assert info2 == 0 and logp2 == logp and np.array_equal(mean, mean2) and np.array_equal(var, var2)   # idempotent, bit for bit
    info, logp_g, _, alpha = core.lml_grad(prm, r)
    assert info == 0 and abs(logp_g - logp) <= 1e-9 * abs(logp)
    resid = mean[:n_probe] + noise_plus_jitter * alpha[idx] - r[idx]
    scale = max(1.0, np.abs(r).max(), noise_plus_jitter * np.abs(alpha).max())
    assert np.abs(resid).max() <= 1e-7 * scale, np.abs(resid).max()
    assert (var > 0).all() and (var[:n_probe] <= noise_plus_jitter * (1 + 1e-6)).all()   # 0 < var_i <= noise at a training point
    return logp, alpha


def test_full_size_properties_c3(core):
    """BASELINE config C3 (Matern-5/2, d=16, N=16384) at full size, without the oracle (too slow there):
    L L^T v = K v on random probe vectors, sum log L_ii and r^T alpha against logp, and the solve identity."""
    import gpax_amd
    from oracle import gp_oracle as orc

    n, d, m = 16384, 16, 1000
    x, y, _, p = orc.synthetic_problem(n, d, m, seed=3, kind=1)
    prm = gpax_amd.make_params(1, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_option("nb", 0)
    r = y - 1.0
    nj = float(p["k_noise"][0] + p["k_jitter"][0])
    logp, alpha = _solve_identity(core, 1, x, r, prm, nj, 400, m, seed=5)
    info, k, l = core.fit_factor(prm, want_k=True, want_l=True)
    assert info == 0 and np.array_equal(l, np.tril(l))
    rng = np.random.default_rng(6)
    v = rng.standard_normal((n, 3))
    kv = k @ v
    assert np.abs(l @ (l.T @ v) - kv).max() <= 1e-10 * np.abs(kv).max()
    logp_host = -0.5 * float(r @ alpha) - float(np.log(np.diag(l)).sum()) - 0.5 * n * np.log(2 * np.pi)
    assert abs(logp - logp_host) <= 1e-9 * abs(logp_host)


def test_full_size_properties_c5(core):
    """BASELINE config C5 (structured mean, Matern-5/2, d=2, N=8192, 10 000 candidates) at full size: solve identity
    with the piecewise prior mean (examples/04_Structured_GP.ipynb cell 10) and analytic EI finite and >= 0."""
    import gpax_amd
    from oracle import gp_oracle as orc
    from scipy.special import ndtr

    n, d, m = 8192, 2, 10000
    rng = np.random.default_rng(11)
    x = rng.uniform(0.0, 3.0, (n, d))
    y = orc.piecewise_mean(x) + 0.05 * rng.standard_normal(n)
    prm = gpax_amd.make_params(1, 1.0, np.array([1.0]), None, np.array([0.01]), 1e-6)
    core.set_option("nb", 0)
    r = y - orc.piecewise_mean(x)
    _solve_identity(core, 1, x, r, prm, 0.01 + 1e-6, 500, m, seed=12)
    info, mean, var, _ = core.posterior(np.full(m, 0.25), 0.0)   # cached factor, another prior mean vector
    assert info == 0
    sd, y_max = np.sqrt(var), float(y.max())
    u = (mean - y_max) / sd
    ei = sd * np.exp(-0.5 * u * u) / np.sqrt(2 * np.pi) + (mean - y_max) * ndtr(u)   # gpax/acquisition.py:291-298
    assert np.isfinite(ei).all() and (ei >= -1e-12).all()


def test_full_size_properties_c4_one_gpu(core):
    """BASELINE config C4's matrix (RBF, d=8, N=32768) on ONE GPU: the solve identity only (K and L are 8.6 GB each)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    n, d, m = 32768, 8, 1000
    x, y, _, p = orc.synthetic_problem(n, d, m, seed=4)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    core.set_option("nb", 0)
    _solve_identity(core, 0, x, y - 1.0, prm, float(p["k_noise"][0] + p["k_jitter"][0]), 400, m, seed=8)


@pytest.mark.parametrize("n,m,d,kind", [(129, 1, 2, 0), (200, 3200, 3, 0), (257, 5, 1, 1), (385, 3300, 4, 0), (256, 128, 2, 2),
                                        (640, 129, 8, 1), (1025, 1, 3, 0), (130, 6500, 2, 0)])
def test_edge_shapes_of_the_split_schedule(core, n, m, d, kind):
    """Shapes at the edges of the schedule's special cases: two and three block columns, one test point, a border taller
    than the look-ahead tile switch (so that the last panel falls back to the stream-level wait), a ragged last block --
    three calls each (the second and third reuse every buffer, flag and event)."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(n, d, m, seed=n + m, kind=kind)
    prm = gpax_amd.make_params(kind, p["k_scale"], p["k_length"], p.get("k_period"), p["k_noise"], p["k_jitter"])
    core.set_train(x)
    core.set_test(xn)
    logp_ref, mean_ref, var_ref = orc.fit_predict(kind, x, y, xn, p)
    for _ in range(3):
        info, logp, mean, var = core.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
        assert info == 0 and abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
        assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8
