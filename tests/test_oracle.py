"""Pin the oracle (CPU, no GPU): closed forms, independent implementations
(scipy.stats, scikit-learn) and the committed golden fixtures.

The reference itself holds no numeric vectors for this path (its tests assert
shapes only, gpax/_tests/test_gp.py:43-47) and cannot be imported here, so these
checks are what anchors the oracle; see oracle/gp_oracle.py header ("parity unpinned").
"""

import math

import numpy as np
import pytest
from scipy.stats import multivariate_normal

from oracle import gp_oracle as orc

from conftest import load_golden

HP = dict(k_scale=np.array([1.3]), k_length=np.array([0.8]), k_noise=np.array([0.01]), k_jitter=np.array([1e-6]))


def _mean_fn_for(g):
    return orc.piecewise_mean if "piecewise" in g["name"] else orc.constant_mean


# ------------------------------------------------------------------- closed forms


def test_n1_closed_form():
    x = np.array([[0.3]])
    y = np.array([0.7])
    k = orc.rbf(x, x, 2.0, 1.0, k_noise=0.5, k_jitter=1e-6)
    assert k.shape == (1, 1) and k[0, 0] == pytest.approx(2.0 + 0.5 + 1e-6, abs=1e-15)
    logp, L, alpha = orc.mvn_log_prob(y, np.array([1.0]), k)
    v = 2.5 + 1e-6
    assert L[0, 0] == pytest.approx(math.sqrt(v), abs=1e-15)
    assert logp == pytest.approx(-0.5 * (0.3**2) / v - 0.5 * math.log(2 * math.pi * v), abs=1e-14)
    assert alpha[0] == pytest.approx(-0.3 / v, abs=1e-15)


def test_n2_closed_form():
    x = np.array([[0.0], [1.0]])
    for fn, c in ((orc.rbf, math.exp(-0.5)), (orc.matern52, None)):
        k = fn(x, x, 1.0, 1.0, k_noise=0.1, k_jitter=0.0)
        if c is None:
            r = math.sqrt(1 + 1e-12)
            c = (1 + math.sqrt(5) * r + 5 / 3) * math.exp(-math.sqrt(5) * r)
        assert k[0, 1] == pytest.approx(c, abs=1e-15) and k[1, 0] == k[0, 1]
        a = k[0, 0]
        L = orc.cholesky_lower(k)
        assert L[0, 0] == pytest.approx(math.sqrt(a), abs=1e-15)
        assert L[1, 0] == pytest.approx(c / math.sqrt(a), abs=1e-15)
        assert L[1, 1] == pytest.approx(math.sqrt(k[1, 1] - c * c / a), abs=1e-15)
        y = np.array([0.2, -0.4])
        logp, _, _ = orc.mvn_log_prob(y, np.zeros(2), k)
        det = a * k[1, 1] - c * c
        quad = (k[1, 1] * y[0] ** 2 - 2 * c * y[0] * y[1] + a * y[1] ** 2) / det
        assert logp == pytest.approx(-0.5 * quad - 0.5 * math.log(det) - math.log(2 * math.pi), abs=1e-13)


def test_periodic_closed_form():
    x = np.array([[0.0, 0.0], [0.25, 0.5]])
    k = orc.periodic(x, x, 1.5, np.array([1.0, 2.0]), np.array([1.0, 2.0]), apply_noise=False)
    s = (math.sin(math.pi * 0.25) / 1.0) ** 2 + (math.sin(math.pi * 0.5 / 2.0) / 2.0) ** 2
    assert k[0, 1] == pytest.approx(1.5 * math.exp(-2 * s), abs=1e-15)
    assert k[0, 0] == pytest.approx(1.5, abs=1e-15)


# ------------------------------------------------------------ diagonal rule & shapes


def test_diagonal_rule():
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (7, 3))
    z = rng.uniform(-1, 1, (7, 3))  # same SHAPE, different points: rule still fires (kernels.py:124)
    base = orc.rbf(x, z, 1.0, 1.0, apply_noise=False)
    with_noise = orc.rbf(x, z, 1.0, 1.0, k_noise=0.2, k_jitter=0.01)
    assert np.allclose(with_noise - base, 0.21 * np.eye(7), atol=1e-16)
    z5 = z[:5]
    assert np.array_equal(orc.rbf(x, z5, 1.0, 1.0, k_noise=0.2), orc.rbf(x, z5, 1.0, 1.0, apply_noise=False))
    v = rng.uniform(0.1, 0.3, 7)  # y_std**2 vector noise lands on the diagonal (gp.py:289-293)
    kv = orc.rbf(x, x, 1.0, 1.0, k_noise=v, k_jitter=1e-6)
    assert np.allclose(np.diag(kv), 1.0 + v + 1e-6, atol=1e-15)
    assert np.allclose(kv - np.diag(np.diag(kv)), base_offdiag(orc.rbf(x, x, 1.0, 1.0, apply_noise=False)))


def base_offdiag(k):
    return k - np.diag(np.diag(k))


def test_expansion_vs_direct_distance():
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (300, 8))
    r2e = orc.squared_distance(x, x, np.array([1.0]))
    r2d = orc.squared_distance_direct(x, x, np.array([1.0]))
    assert np.abs(r2e - r2d).max() < 2e-14
    assert (r2e >= 0).all()
    ke, kd = np.exp(-0.5 * r2e), np.exp(-0.5 * r2d)
    assert np.abs(ke - kd).max() < 5e-15


def test_ard_lengthscale_matches_prescaling():
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (20, 3))
    ell = np.array([0.5, 1.0, 2.0])
    assert np.allclose(orc.rbf(x, x, 1.0, ell, apply_noise=False), orc.rbf(x / ell, x / ell, 1.0, 1.0, apply_noise=False), atol=1e-15)


# -------------------------------------------------------- independent implementations


@pytest.mark.parametrize("n,d", [(30, 1), (120, 4)])
def test_logp_vs_scipy_stats(n, d):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, (n, d))
    y = rng.standard_normal(n)
    for fn in (orc.rbf, orc.matern52):
        k = fn(x, x, HP["k_scale"], HP["k_length"], HP["k_noise"], HP["k_jitter"])
        logp, L, alpha = orc.mvn_log_prob(y, np.ones(n), k)
        ref = multivariate_normal(mean=np.ones(n), cov=k, allow_singular=False).logpdf(y)
        assert logp == pytest.approx(ref, rel=1e-9)
        assert np.abs(L @ L.T - k).max() < 1e-13
        assert np.abs(k @ alpha - (y - 1.0)).max() < 1e-9


def test_kernels_vs_sklearn():
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, ExpSineSquared, Matern

    rng = np.random.default_rng(3)
    x, z = rng.uniform(-1, 1, (40, 5)), rng.uniform(-1, 1, (25, 5))
    k = orc.rbf(x, z, 1.3, 0.8, apply_noise=False)
    assert np.abs(k - (ConstantKernel(1.3) * RBF(0.8))(x, z)).max() < 1e-14
    k = orc.matern52(x, z, 1.3, 0.8, apply_noise=False)
    # sklearn has no 1e-12 under the sqrt: the two agree to ~sqrt5*1e-6 * derivative ~ 1e-11
    assert np.abs(k - (ConstantKernel(1.3) * Matern(0.8, nu=2.5))(x, z)).max() < 1e-10
    x1, z1 = x[:, :1], z[:, :1]
    k = orc.periodic(x1, z1, 1.3, 0.8, 0.6, apply_noise=False)
    # ExpSineSquared = exp(-2 sin^2(pi |d| / p) / l^2): the reference formula for d = 1
    assert np.abs(k - (ConstantKernel(1.3) * ExpSineSquared(0.8, 0.6))(x1, z1)).max() < 1e-14


def test_posterior_vs_sklearn_gpr():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel

    rng = np.random.default_rng(4)
    x, xn = rng.uniform(-1, 1, (60, 2)), rng.uniform(-1, 1, (15, 2))
    y = np.sin(3 * x.sum(1)) + 0.05 * rng.standard_normal(60)
    p = dict(HP)
    mean, cov = orc.mean_and_covariance(orc.RBF, x, y, xn, p, mean_fn=lambda a: np.zeros(a.shape[0]))
    gpr = GaussianProcessRegressor(ConstantKernel(1.3) * RBF(0.8), alpha=0.01 + 1e-6, optimizer=None).fit(x, y)
    mu_s, cov_s = gpr.predict(xn, return_cov=True)
    assert np.abs(mean - mu_s).max() < 1e-9
    # reference adds jitter to k_new's diagonal (gp.py:287 with noise=0): remove it to compare
    assert np.abs(cov - 1e-6 * np.eye(15) - cov_s).max() < 1e-9
    assert gpr.log_marginal_likelihood_value_ == pytest.approx(
        orc.mvn_log_prob(y, np.zeros(60), orc.rbf(x, x, **{k: v for k, v in p.items()}))[0], rel=1e-11
    )


def test_cholesky_vs_inverse_branch():
    rng = np.random.default_rng(5)
    x, xn = rng.uniform(-1, 1, (80, 3)), rng.uniform(-1, 1, (20, 3))
    y = rng.standard_normal(80)
    a = orc.mean_and_covariance(orc.RBF, x, y, xn, HP, use_cholesky=True)
    b = orc.mean_and_covariance(orc.RBF, x, y, xn, HP, use_cholesky=False)
    assert np.abs(a[0] - b[0]).max() < 1e-7 and np.abs(a[1] - b[1]).max() < 1e-7


def test_unconditioned():
    rng = np.random.default_rng(6)
    xn = rng.uniform(-1, 1, (9, 2))
    m, c = orc.mean_and_covariance_unconditioned(orc.RBF, xn, HP, observation_noise=False)
    assert np.array_equal(m, np.ones(9))
    assert np.allclose(np.diag(c), 1.3 + 1e-6)
    _, c2 = orc.mean_and_covariance_unconditioned(orc.RBF, xn, HP, observation_noise=True)
    assert np.allclose(np.diag(c2), 1.3 + 0.01 + 1e-6)


def test_fit_predict_matches_posterior():
    x, y, xn, p = orc.synthetic_problem(150, 8, 30, seed=9)
    for kind in (orc.RBF, orc.MATERN52):
        logp, mean, var = orc.fit_predict(kind, x, y, xn, p)
        m2, c2 = orc.mean_and_covariance(kind, x, y, xn, p, use_cholesky=True)
        k = orc.gram(kind, x, x, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
        assert logp == pytest.approx(orc.mvn_log_prob(y, np.ones(150), k)[0], rel=1e-13)
        assert np.abs(mean - m2).max() < 1e-10
        assert np.abs(var - np.diag(c2)).max() < 1e-10


# ----------------------------------------------------------------------- transforms


def _dummy_data():
    """gpax/_tests/conftest.py:6-12 restated (numpy seed 123, N=100, d=2)."""
    np.random.seed(123)
    x = (np.random.random(size=(100, 2)) * 10 / 2) ** 2 + 2
    return x, 10 * x**2


@pytest.mark.parametrize("cls", [orc.ScaleOracle, orc.NormalizeOracle])
@pytest.mark.parametrize("as_", ["mean", "std"])
def test_transform_round_trip(cls, as_):
    """Mirrors gpax/_tests/test_transforms.py:11-34."""
    x, _ = _dummy_data()
    t = cls().fit(x)
    x1 = t.forward(x, as_)
    assert np.allclose(x, t.reverse(x1, as_))
    if cls is orc.ScaleOracle and as_ == "mean":
        assert np.allclose(x1.min(0), -1.0) and np.allclose(x1.max(0), 1.0)


def test_acquisition_closed_forms():
    mu, sd = np.array([0.0, 1.0]), np.array([1.0, 2.0])
    ei = orc.expected_improvement(mu, sd, 0.0)
    assert ei[0] == pytest.approx(1 / math.sqrt(2 * math.pi), abs=1e-15)
    assert orc.upper_confidence_bound(mu, sd, 4.0)[1] == pytest.approx(1 + 2 * math.sqrt(4 + 1e-12))
    assert np.array_equal(orc.upper_confidence_bound(mu, sd, math.inf), sd)
    assert np.array_equal(orc.upper_confidence_bound(mu, sd, 0.0), mu)


# -------------------------------------------------------------------- golden fixtures


def test_oracle_reproduces_golden(golden):
    """The committed vectors are what the oracle computes today (regression pin) and
    are internally consistent (L L^T = K, K alpha = y - m, K symmetric)."""
    g = golden
    p = g["params"]
    kp = {k: v for k, v in p.items() if k not in ("k_noise", "k_jitter")}
    noise = g["y_std"] ** 2 if g["y_std"] is not None else p["k_noise"]
    k = orc.gram(g["kind"], g["x"], g["x"], k_noise=noise, k_jitter=p["k_jitter"], **kp)
    assert np.abs(k - g["K"]).max() < 1e-14
    assert np.array_equal(g["K"], g["K"].T)
    assert np.abs(g["L"] @ g["L"].T - g["K"]).max() < 1e-13
    logp, L, alpha = orc.mvn_log_prob(g["y"], g["m_train"], g["K"])
    assert np.abs(L - g["L"]).max() < 1e-12
    assert logp == pytest.approx(float(g["logp"]), rel=1e-12)
    assert np.abs(alpha - g["alpha"]).max() <= 1e-9 * max(1.0, np.abs(g["alpha"]).max())
    mean, cov = orc.mean_and_covariance(
        g["kind"], g["x"], g["y"], g["x_new"], p, y_std=g["y_std"],
        observation_noise=g["observation_noise"], mean_fn=_mean_fn_for(g),
    )
    scale = max(1.0, np.abs(g["post_mean"]).max())
    assert np.abs(mean - g["post_mean"]).max() < 1e-9 * scale
    assert np.abs(cov - g["post_cov"]).max() < 1e-9
    assert g["post_mean"].shape == (g["x_new"].shape[0],) and g["post_cov"].shape == (g["x_new"].shape[0],) * 2


def test_golden_inverse_branch_gap_recorded():
    """The reference DEFAULT is the explicit inverse (gp.py:129,303-310); the HIP path
    always uses Cholesky.  Record that the two branches agree to ~cond*eps on the
    reference's own fixture recipe."""
    g = load_golden("sin25_rbf")
    assert np.abs(g["post_mean"] - g["post_mean_inv"]).max() < 1e-6
    assert np.nanmax(np.abs(g["post_sd"] - g["post_sd_inv"])) < 1e-4


# ---------------------------------------------------------------- gradient of the log-marginal


@pytest.mark.parametrize("kind", [orc.RBF, orc.MATERN52, orc.PERIODIC])
@pytest.mark.parametrize("ard", [False, True])
def test_lml_grad_matches_central_differences(kind, ard):
    """The oracle's closed-form gradient (the checker of gpx_lml_grad) against central differences of the
    oracle's own log-marginal -- the only pin available: the reference obtains this by jax.grad and holds no
    numbers for it (parity unpinned)."""
    rng = np.random.default_rng(3 + kind)
    n, d = 40, 3
    x, r = rng.uniform(-1, 1, (n, d)), rng.standard_normal(n)
    kl = rng.uniform(0.5, 1.5, d) if ard else np.array([0.8])
    kp = (rng.uniform(1.5, 2.5, d) if ard else np.array([2.0])) if kind == orc.PERIODIC else None
    s, nz, h = 1.3, 0.05, 1e-6

    def f(s=s, kl=kl, kp=kp, nz=nz):
        return orc.mvn_log_prob(r, np.zeros(n), orc.gram(kind, x, x, s, kl, kp, nz, 1e-6, True))[0]

    logp, g, alpha = orc.lml_grad(kind, x, r, s, kl, kp, nz)
    assert abs(logp - f()) < 1e-10
    assert np.allclose(alpha, np.linalg.solve(orc.gram(kind, x, x, s, kl, kp, nz, 1e-6, True), r), rtol=1e-8)
    assert abs((f(s=s + h) - f(s=s - h)) / (2 * h) - g["k_scale"][0]) < 1e-5
    assert abs((f(nz=nz + h) - f(nz=nz - h)) / (2 * h) - g["k_noise"][0]) < 1e-4
    for i in range(kl.size):
        e = np.zeros_like(kl)
        e[i] = h
        assert abs((f(kl=kl + e) - f(kl=kl - e)) / (2 * h) - g["k_length"][i]) < 1e-5
    if kind == orc.PERIODIC:
        for i in range(kp.size):
            e = np.zeros_like(kp)
            e[i] = h
            assert abs((f(kp=kp + e) - f(kp=kp - e)) / (2 * h) - g["k_period"][i]) < 1e-5
    assert g["k_length"].shape == kl.shape
