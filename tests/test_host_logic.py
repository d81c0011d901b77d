"""CPU tests of the host-side mirror of gpax's interface (no GPU): transforms (the reference's own
round-trip tests, gpax/_tests/test_transforms.py:11-34), prior plumbing, constructor validation
(gpax/gp.py:171-184), rng key chain (gpax/state.py:87-126)."""

import math

import numpy as np
import pytest

from gpax_amd import state
from gpax_amd.transforms import IdentityTransform, NormalizeTransform, ScaleTransform
from gpax_amd.utils.prior_utils import HalfNormal, LogNormal, Normal, Parameter, Uniform

from oracle import gp_oracle as orc


def _dummy_data():
    """gpax/_tests/conftest.py:6-12 restated."""
    np.random.seed(123)
    x = (np.random.random(size=(100, 2)) * 10 / 2) ** 2 + 2
    return x, 10 * x**2


@pytest.mark.parametrize("transforms_as", ["mean", "std"])
@pytest.mark.parametrize("factory", [ScaleTransform, NormalizeTransform, IdentityTransform])
def test_other_transforms(factory, transforms_as):
    """gpax/_tests/test_transforms.py:21-34."""
    x, _ = _dummy_data()
    t = factory()
    t.fit(x)
    x1 = t.forward(x, transforms_as=transforms_as)
    x2 = t.reverse(x1, transforms_as=transforms_as)
    assert np.allclose(x, x2)
    if transforms_as == "mean" and factory is ScaleTransform:
        assert np.allclose(x1.min(axis=0), -1.0) and np.allclose(x1.max(axis=0), 1.0)


def test_transforms_match_oracle_and_rules():
    x, _ = _dummy_data()
    for cls, ocls in ((ScaleTransform, orc.ScaleOracle), (NormalizeTransform, orc.NormalizeOracle)):
        t, o = cls(), ocls().fit(x)
        t.fit(x)
        for as_ in ("mean", "std"):
            assert np.array_equal(t.forward(x, as_), o.forward(x, as_))
            assert np.array_equal(t.reverse(x, as_), o.reverse(x, as_))
    t = ScaleTransform()
    with pytest.raises(RuntimeError):
        t.forward(x)  # before fit (gpax/transforms.py:151-152)
    t.fit(x)
    with pytest.raises(RuntimeError):
        t.fit(x)  # fit twice (gpax/transforms.py:101-102)
    with pytest.raises(ValueError):
        t.forward(x, transforms_as="bogus")
    assert t.forward(None) is None
    ident = IdentityTransform()
    ident.fit(None)
    assert ident.forward(np.arange(3.0)).shape == (3, 1)  # _ensure_2d (gpax/transforms.py:61-64)


def test_distributions():
    rng = np.random.default_rng(0)
    hn = HalfNormal(0.01)
    assert hn.median == pytest.approx(orc.K_NOISE_MEDIAN, rel=1e-14)
    s = hn.sample(rng, (200000,))
    assert np.median(s) == pytest.approx(hn.median, rel=2e-2) and (s >= 0).all()
    ln = LogNormal(0.0, 1.0)
    assert ln.median == 1.0 and ln.mean == pytest.approx(math.exp(0.5))
    from scipy import stats

    xs = np.array([0.1, 0.5, 2.0])
    assert np.allclose(ln.log_prob(xs), stats.lognorm(s=1.0).logpdf(xs))
    assert np.allclose(hn.log_prob(xs), stats.halfnorm(scale=0.01).logpdf(xs))
    assert np.allclose(Uniform(0.5, 2.5).log_prob(np.array([1.0])), -math.log(2.0))
    assert np.allclose(Normal(1.0, 2.0).log_prob(xs), stats.norm(1.0, 2.0).logpdf(xs))
    for d in (hn, ln, Uniform(0.5, 2.5), Normal(0.0, 1.0)):
        u = np.array([-0.3, 0.2, 1.1])
        x, lj = d.from_unconstrained(u)
        assert np.allclose(d.to_unconstrained(x), u)
        h = 1e-6
        num = (d.from_unconstrained(u + h)[0] - d.from_unconstrained(u - h)[0]) / (2 * h)
        assert np.allclose(np.log(num), lj, atol=1e-6)


def test_parameter_and_prior_plumbing():
    from gpax_amd.kernels import RBFKernel

    with pytest.raises(ValueError):
        Parameter("nope")
    with pytest.raises(ValueError):
        Parameter(1.0, plate=0)
    k = RBFKernel()
    assert sorted(k.params) == ["k_jitter", "k_length", "k_noise", "k_scale"]  # prefix rule, prior_utils.py:145-155
    assert k.params["k_jitter"] == 1e-6 and isinstance(k.params["k_noise"], HalfNormal)
    k.k_length.plate = 3  # ARD as in examples/02: kernel.k_length.plate = 2
    draw = k.sample_parameters(np.random.default_rng(1))
    assert draw["k_length"].shape == (3,) and draw["k_scale"].shape == (1,) and draw["k_jitter"][0] == 1e-6
    many = k.sample_parameters(np.random.default_rng(1), 17)
    assert many["k_length"].shape == (17, 3) and many["k_noise"].shape == (17, 1)
    assert k.get_sample_parameter_means()["k_jitter"] == 1e-6


def test_constructor_validation_without_gpu():
    """gpax/gp.py:171-184 and gpax/_tests/test_gp.py:31-35: shape/None misuse raises at construction."""
    from gpax_amd.gp import ExactGP
    from gpax_amd.kernels import RBFKernel

    x, y = np.linspace(-1, 1, 5).reshape(-1, 1), np.arange(5.0)
    for y_std in (0.0, 1.0):
        with pytest.raises(ValueError):
            ExactGP(kernel=RBFKernel(), x=None, y=None, y_std=y_std)
    with pytest.raises(ValueError):
        ExactGP(kernel=RBFKernel(), x=x, y=None)
    with pytest.raises(TypeError):
        ExactGP(kernel="rbf", x=x, y=y)
    with pytest.raises(ValueError):
        ExactGP(kernel=RBFKernel(), x=x, y=y, hp_samples=0)
    gp = ExactGP(kernel=RBFKernel(), x=x, y=y, y_std=0.5)
    assert gp.y_std.shape == y.shape  # scalar y_std broadcast (gpax/gp.py:179-180)
    assert np.allclose(gp.x_transformed.min(), -1) and np.allclose(gp.y_transformed.mean(), 0, atol=1e-12)
    gp0 = ExactGP(kernel=RBFKernel(), x=None, y=None)
    assert isinstance(gp0.input_transform, IdentityTransform) and isinstance(gp0.output_transform, IdentityTransform)
    assert gp0.mean(np.zeros((4, 1))).tolist() == [1.0] * 4  # ConstantMean() defaults to 1.0 (gpax/means.py:75)


def test_rng_key_chain():
    """gpax/state.py:87-126: the key advances by seeding numpy with itself."""
    state.set_rng_key(0)
    k1, _ = state.get_rng_key()
    k2, _ = state.get_rng_key()
    np.random.seed(0)
    e1 = np.random.randint(low=0, high=2**32 - 1)
    np.random.seed(e1)
    e2 = np.random.randint(low=0, high=2**32 - 1)
    assert (k1, k2) == (e1, e2)
    state.set_rng_key(0)
    assert state.get_rng_key()[0] == k1


def test_param_space_bijection():
    from gpax_amd.gp import ExactGP, _ParamSpace
    from gpax_amd.kernels import RBFKernel

    x, y = np.linspace(-1, 1, 5).reshape(-1, 1), np.arange(5.0)
    k = RBFKernel()
    k.k_length.plate = 2
    gp = ExactGP(kernel=k, x=x, y=y)
    sp = _ParamSpace(gp)
    assert sp.dim == 4  # k_noise(1) + k_scale(1) + k_length(2); k_jitter is a constant
    u0 = sp.init_to_median()
    p, lp = sp.unpack(u0)
    assert p["k_scale"][0] == pytest.approx(1.0) and p["k_noise"][0] == pytest.approx(orc.K_NOISE_MEDIAN)
    assert p["k_jitter"][0] == 1e-6 and np.isfinite(lp)


# ------------------------------------------------------------------ samplers and prior gradients (host only)


def test_prior_grad_terms_match_central_differences():
    from gpax_amd.utils.prior_utils import HalfNormal, LogNormal, Normal, Uniform

    u, h = np.array([-0.7, 0.4, 1.3]), 1e-6
    for dist in (HalfNormal(0.5), LogNormal(0.3, 1.2), Uniform(1.0, 3.0), Normal(0.5, 2.0)):
        def f(v):
            x, lj = dist.from_unconstrained(v)
            return dist.log_prob(x) + lj

        x, dx, g = dist.grad_terms(u)
        assert np.allclose(x, dist.from_unconstrained(u)[0])
        assert np.allclose(g, (f(u + h) - f(u - h)) / (2 * h), atol=1e-6)
        assert np.allclose(dx, (dist.from_unconstrained(u + h)[0] - dist.from_unconstrained(u - h)[0]) / (2 * h), atol=1e-7)
        assert np.allclose(dist.to_unconstrained(x), u)


def test_warmup_windows_follow_the_stan_schedule():
    from gpax_amd import mcmc

    assert mcmc._windows(10) == []
    assert mcmc._windows(1000) == [(75, 100), (100, 150), (150, 250), (250, 450), (450, 950)]
    w = mcmc._windows(100)
    assert w[0][0] == 15 and w[-1][1] == 90 and all(a < b for a, b in w)


@pytest.mark.parametrize("algorithm", ["nuts", "hmc"])
def test_sampler_recovers_a_badly_scaled_gaussian(algorithm):
    from gpax_amd import mcmc

    sd, mu = np.array([0.1, 1.0, 10.0]), np.array([1.0, -2.0, 0.5])

    def vg(u):
        z = (u - mu) / sd
        return -0.5 * float(z @ z), -z / sd

    draws, info = mcmc.sample(vg, np.zeros(3), 500, 3000, np.random.default_rng(1), algorithm=algorithm, num_leapfrog=12)
    assert draws.shape == (3000, 3) and info["divergences"] == 0
    assert np.all(np.abs(draws.mean(0) - mu) < 0.25 * sd)
    assert np.all(np.abs(draws.std(0) / sd - 1.0) < 0.15)
    assert np.allclose(info["inverse_mass"], sd**2, rtol=0.5)       # the mass matrix adapted to the scales


def test_nuts_is_unbiased_on_a_skewed_target():
    """u = log Gamma(3): mean digamma(3), variance trigamma(3)."""
    from scipy.special import polygamma

    from gpax_amd import mcmc

    def vg(u):
        return float(np.sum(3 * u - np.exp(u))), 3 - np.exp(u)

    draws, info = mcmc.sample(vg, np.zeros(2), 500, 6000, np.random.default_rng(8))
    assert np.all(np.abs(draws.mean(0) - polygamma(0, 3)) < 0.05)
    assert np.all(np.abs(draws.var(0) - polygamma(1, 3)) < 0.05)
    assert 0.7 < info["mean_accept"] < 0.98


def test_sampler_rejects_bad_arguments():
    from gpax_amd import mcmc

    with pytest.raises(ValueError):
        mcmc.sample(lambda u: (0.0, np.zeros(1)), np.zeros(1), 5, 5, np.random.default_rng(0), algorithm="gibbs")
    with pytest.raises(ValueError):
        mcmc.sample(lambda u: (-np.inf, np.zeros(1)), np.zeros(1), 5, 5, np.random.default_rng(0))


def test_adam_matches_the_textbook_update():
    from gpax_amd.gp import _Adam

    opt = _Adam(2, step_size=0.1, b1=0.5, b2=0.9)
    x = opt.step(np.zeros(2), np.array([1.0, -2.0]))
    assert np.allclose(x, [0.1, -0.1], atol=1e-6)       # first step: m_hat / sqrt(v_hat) = sign(grad)


def test_reference_named_helpers_exist():
    """Names a gpax user may reach for (gpax/gp.py:352-397, gpax/state.py:30-58, gpax/acquisition.py:31-131,
    gpax/kernels.py:33-51) exist with the reference's signatures."""
    import inspect

    from gpax_amd import acquisition, gp, kernels, state

    for name in ("_sample_prior", "_sample_unconditioned_prior", "_sample_conditioned_prior", "_sample_posterior",
                 "_pre_sample", "_fit_transforms", "get_mvn_posterior", "_get_mean_and_covariance",
                 "_get_mean_and_covariance_unconditioned", "_sample_gp_given_single_hp", "_fast_predict"):
        assert callable(getattr(gp.GaussianProcess, name)), name
    assert list(inspect.signature(gp.GaussianProcess._sample_prior).parameters)[1:] == ["rng", "x_new", "condition_on_data", "fast"]
    assert callable(gp.VariationalInferenceGP._print_summary)
    for name in ("_monte_carlo_evaluation", "_analytic_evaluation", "_optimize_halton", "optimize"):
        assert callable(getattr(acquisition.AcquisitionFunction, name)), name
    assert "must have dimensions ('N', 'd')" in acquisition.x_shape_err_msg(np.zeros((2, 3, 4)), ("N", "d"))
    assert kernels._add_jitter(1.0, 0.5) == 1.5 and callable(kernels._squared_distance)
    v = state.verbose
    with state.silent_mode():
        assert state.verbose == 0
    with state.debug_mode():
        assert state.verbose == 2
    assert state.verbose == v
    acq = acquisition.UpperConfidenceBound(np.array([[0.0], [1.0]]), q=2)
    with pytest.raises(ValueError):
        acq._analytic_evaluation(None, np.zeros((3, 1)))
    with pytest.raises(ValueError):
        acq._monte_carlo_evaluation(None, np.zeros((3, 1)))


def test_lockstep_chains_equal_sequential_chains_on_a_toy_density():
    """``mcmc.sample_lockstep`` (the chains of ``ExactGP.fit(num_chains > 1, chain_method="vectorized")``: one thread per
    chain, every round of gradient requests served by ONE batched call) gives each chain the draws of a sequential run on
    the same random stream, also when chains finish at different times, and fails as a whole when the batched call fails."""
    import numpy as np
    import pytest

    from gpax_amd import mcmc

    def vg(u):
        return float(-0.5 * np.sum(u * u) - 0.1 * np.sum(u**4)), -u - 0.4 * u**3

    sizes = []

    def vgb(us):
        sizes.append(len(us))
        return [vg(u) for u in us]

    seq = [mcmc.sample(vg, np.ones(3), 40, 25, np.random.default_rng(s)) for s in (1, 2, 3)]
    lock = mcmc.sample_lockstep(vgb, [np.ones(3)] * 3, 40, 25, [np.random.default_rng(s) for s in (1, 2, 3)])
    for (d0, i0), (d1, i1) in zip(seq, lock):
        assert np.array_equal(d0, d1) and i0["num_grad_evals"] == i1["num_grad_evals"] and i0["step_size"] == i1["step_size"]
    assert max(sizes) == 3 and min(sizes) >= 1 and sum(sizes) == sum(i["num_grad_evals"] for _, i in lock)
    assert lock[0][1]["batched_calls"] == len(sizes) < sum(sizes)

    def broken(us):
        raise ValueError("boom")

    with pytest.raises((RuntimeError, ValueError)):
        mcmc.sample_lockstep(broken, [np.ones(2)] * 2, 5, 5, [np.random.default_rng(s) for s in (1, 2)])
