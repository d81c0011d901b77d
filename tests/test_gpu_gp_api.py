"""GPU tests of the gpax-compatible Python surface (``-m gpu``): the reference's own shape tests
(gpax/_tests/test_gp.py:8-47) plus numeric checks of predict / sample / acquisition against the oracle."""

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _dummy_sinusoidal_data():
    """gpax/_tests/conftest.py:15-30 restated."""
    np.random.seed(0)
    x = np.random.uniform(-1.0, 1.0, 25)
    y = np.sin(10 * x) + np.random.normal(0.0, 0.1, 25)
    return x.reshape(-1, 1), y.squeeze(), np.linspace(-1.1, 1.1, 300).reshape(-1, 1)


@pytest.mark.parametrize("gp_factory", ["ExactGP", "VariationalInferenceGP"])
@pytest.mark.parametrize("y_std", [None, 0.0, 1.0])
@pytest.mark.parametrize("model_state", ["unconditioned_prior", "prior", "posterior"])
def test_ExactGP_sample_shapes(gp_factory, y_std, model_state):
    """gpax/_tests/test_gp.py:8-47: {ExactGP, VariationalInferenceGP} x RBFKernel."""
    from gpax_amd import gp as gp_mod
    from gpax_amd import state
    from gpax_amd.kernels import RBFKernel

    HP_SAMPLES, GP_SAMPLES = 17, 33
    if gp_factory == "ExactGP":
        kwargs = {"hp_samples": HP_SAMPLES, "gp_samples": GP_SAMPLES, "num_warmup": 100}
    else:   # gpax/_tests/test_gp.py:23-31: VI uses the guide median only
        HP_SAMPLES = 1
        kwargs = {"hp_samples": HP_SAMPLES, "gp_samples": GP_SAMPLES, "num_steps": 10}
    ExactGP = getattr(gp_mod, gp_factory)
    x, y, x_grid = _dummy_sinusoidal_data()
    x_grid = x_grid[::5]
    state.set_rng_key(0)
    kernel = RBFKernel()
    if model_state == "unconditioned_prior":
        if y_std is not None:
            with pytest.raises(ValueError):
                ExactGP(kernel=kernel, x=None, y=None, y_std=y_std, **kwargs)
            return
        gp = ExactGP(kernel=kernel, x=None, y=None, y_std=y_std, **kwargs)
    elif model_state == "prior":
        gp = ExactGP(kernel=kernel, x=x, y=y, y_std=y_std, **kwargs)
    else:
        gp = ExactGP(kernel=kernel, x=x, y=y, y_std=y_std, **kwargs)
        gp.fit()
    result = gp.sample(x_grid)
    yy = result.y
    assert yy.ndim == 3
    assert yy.shape[0] == HP_SAMPLES
    assert yy.shape[1] == GP_SAMPLES
    assert yy.shape[2] == x_grid.shape[0]
    assert result.mu.shape == (x_grid.shape[0],) and result.sd.shape == (x_grid.shape[0],)


def test_get_mvn_posterior_matches_golden():
    """_get_mean_and_covariance (gpax/gp.py:247-312) through the Python surface, on the reference's
    fixture recipe after the default transforms."""
    from gpax_amd.gp import ExactGP
    from gpax_amd.kernels import RBFKernel

    g = load_golden("sin25_rbf")
    x, y, x_grid = _dummy_sinusoidal_data()
    gp = ExactGP(kernel=RBFKernel(), x=x, y=y)
    assert np.abs(gp.x_transformed - g["x"]).max() < 1e-15 and np.abs(gp.y_transformed - g["y"]).max() < 1e-14
    params = {k: v for k, v in g["params"].items()}
    mean, cov = gp.get_mvn_posterior(g["x_new"], params)
    assert np.abs(mean - g["post_mean"]).max() < 1e-8 and np.abs(cov - g["post_cov"]).max() < 1e-8
    k = gp.kernel(g["x"], g["x"], **params)
    assert np.abs(k - g["K"]).max() < 1e-12
    assert abs(gp.log_marginal(params) - float(g["logp"])) < 1e-10 * abs(float(g["logp"]))


def test_fast_predict_matches_oracle_with_transforms():
    """predict(fast=True) before fit: prior-median hyper-parameters (gpax/gp.py:464-501), raw units."""
    from gpax_amd import state
    from gpax_amd.gp import ExactGP
    from gpax_amd.kernels import MaternKernel
    from oracle import gp_oracle as orc

    x, y, x_grid = _dummy_sinusoidal_data()
    state.set_rng_key(3)
    gp = ExactGP(kernel=MaternKernel(), x=x, y=y, hp_samples=2001)
    mu, sd = gp.predict(x_grid, fast=True)
    # the medians actually used
    state.set_rng_key(3)
    _, rng = state.get_rng_key()
    hp, _ = gp._get_hp_samples_from_prior(rng, True)
    p = {k: v[0] for k, v in hp.items()}
    sx, sy = orc.ScaleOracle().fit(x), orc.NormalizeOracle().fit(y)
    m_ref, s_ref = orc.fast_predict(orc.MATERN52, sx.forward(x), sy.forward(y).squeeze(), sx.forward(x_grid), p)
    mu_ref = sy.reverse(m_ref, "mean").squeeze()
    sd_ref = sy.reverse(s_ref, "std").squeeze()
    assert np.abs(mu - mu_ref).max() < 1e-8 and np.abs(sd - sd_ref).max() < 1e-8


def test_mvn_sample_statistics_and_exact_transform():
    """K12: out = mean + eps @ chol(cov)^T exactly for the given eps; the draws have the right moments."""
    import gpax_amd
    from oracle import gp_oracle as orc

    core = gpax_amd.state.get_core()
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (150, 2))
    cov = orc.rbf(x, x, 1.3, 0.7, 0.05, 1e-6)
    mean = rng.standard_normal(150)
    eps = rng.standard_normal((37, 150))
    info, out = core.mvn_sample(mean, cov, eps)
    assert info == 0
    ref = mean[None, :] + eps @ orc.cholesky_lower(cov).T
    assert np.abs(out - ref).max() < 1e-10
    info, big = core.mvn_sample(None, cov, rng.standard_normal((20000, 150)))
    emp = big.T @ big / 20000
    assert np.abs(emp - cov).max() < 0.08


def test_fit_recovers_noise_scale():
    """Statistical parity of fit(): on data from a known GP the posterior medians land near the truth."""
    from gpax_amd import state
    from gpax_amd.gp import ExactGP
    from gpax_amd.kernels import RBFKernel
    from gpax_amd.utils.prior_utils import HalfNormal, Parameter

    rng = np.random.default_rng(5)
    x = np.sort(rng.uniform(-1, 1, 60)).reshape(-1, 1)
    y = np.sin(4 * x[:, 0]) + 0.1 * rng.standard_normal(60)
    state.set_rng_key(1)
    gp = ExactGP(kernel=RBFKernel(k_noise=Parameter(HalfNormal(0.1), 1)), x=x, y=y, hp_samples=60, num_warmup=150)
    gp.fit()
    assert gp.metadata["mcmc_info"][0]["divergences"] <= 3
    hp, n = gp._get_hp_samples_from_posterior(False)
    assert n == 60 and set(hp) == {"k_noise", "k_scale", "k_length"}
    noise_med = float(np.median(hp["k_noise"]))
    # noise variance in normalised units: (0.1 / std(y))^2 ~ 0.02
    assert 0.005 < noise_med < 0.08
    mu, sd = gp.predict(x, fast=True)
    assert np.sqrt(np.mean((mu - np.sin(4 * x[:, 0])) ** 2)) < 0.08


def test_acquisition_ei_ucb_and_factor_reuse():
    """Config C5 in miniature: structured (piecewise) mean + EI over Halton candidates (gpax/acquisition.py)."""
    from gpax_amd import state
    from gpax_amd.acquisition import ExpectedImprovement, UpperConfidenceBound
    from gpax_amd.gp import ExactGP
    from gpax_amd.kernels import MaternKernel
    from gpax_amd.means import Mean
    from oracle import gp_oracle as orc

    class PiecewiseMean(Mean):  # examples/04_Structured_GP.ipynb cell 10 with fixed parameters
        def _mean_function(self, x):
            return orc.piecewise_mean(x)

    rng = np.random.default_rng(1)
    x = rng.uniform(0.0, 3.0, (200, 2))
    y = orc.piecewise_mean(x) + rng.normal(0.0, 0.1, 200)
    state.set_rng_key(2)
    gp = ExactGP(kernel=MaternKernel(), mean=PiecewiseMean(), input_transform=None, output_transform=None, x=x, y=y)
    bounds = np.array([[0.0, 0.0], [3.0, 3.0]])
    cand = rng.uniform(0.0, 3.0, (250, 2))
    mu, sd = gp.predict(cand, fast=True)
    state.set_rng_key(2)
    _, r = state.get_rng_key()
    hp, _ = gp._get_hp_samples_from_prior(r, True)
    p = {k: v[0] for k, v in hp.items()}
    m_ref, s_ref = orc.fast_predict(orc.MATERN52, x, y, cand, p, mean_fn=orc.piecewise_mean)
    assert np.abs(mu - m_ref).max() < 1e-7 and np.abs(sd - s_ref).max() < 1e-7
    # every predict() advances the rng key and re-draws the prior medians (as in the reference,
    # gpax/gp.py:470-478): pin the key to compare like with like
    state.set_rng_key(7)
    ei = ExpectedImprovement(bounds)(gp, cand)
    state.set_rng_key(7)
    mu7, sd7 = gp.predict(cand, fast=True)
    assert np.abs(ei - orc.expected_improvement(mu7, sd7, y.max())).max() < 1e-12
    state.set_rng_key(7)
    ucb = UpperConfidenceBound(bounds, beta=4.0)(gp, cand)
    assert np.abs(ucb - orc.upper_confidence_bound(mu7, sd7, 4.0)).max() < 1e-12
    x_star, val = ExpectedImprovement(bounds).optimize(gp, 1000)
    assert x_star.shape == (1, 2) and np.isfinite(val) and (x_star >= 0).all() and (x_star <= 3).all()


def test_reference_private_sampling_entry_points():
    """gpax/gp.py:352-397: _sample_*prior / _sample_posterior return (hp, mu, sd, sampled) with the (hp, gp, M)
    layout, and kernels._squared_distance reproduces the oracle's distances through the HIP gram kernel."""
    from gpax_amd import kernels, state
    from gpax_amd.gp import ExactGP
    from oracle import gp_oracle as orc

    x, y, x_grid = _dummy_sinusoidal_data()
    state.set_rng_key(3)
    gp = ExactGP(kernel=kernels.RBFKernel(), x=x, y=y, hp_samples=5, gp_samples=7)
    rng, xg = gp._pre_sample(gp.input_transform.forward(x_grid[::10], transforms_as="mean"))
    for fn, n_hp in ((gp._sample_unconditioned_prior, 5), (gp._sample_conditioned_prior, 5)):
        hp, mu, sd, ys = fn(rng, xg, False)
        assert ys.shape == (n_hp, 7, xg.shape[0]) and mu.shape == (n_hp, xg.shape[0]) and sd.shape == mu.shape
        assert set(hp) == {"k_noise", "k_jitter", "k_scale", "k_length"}
    hp, mu, sd, ys = gp._sample_conditioned_prior(rng, xg, True)
    assert ys.shape == (1, 7, xg.shape[0])
    r2 = kernels._squared_distance(x, x[:5], np.array([0.7]))
    assert np.abs(r2 - orc.squared_distance_direct(x, x[:5], np.array([0.7]))).max() < 1e-9
