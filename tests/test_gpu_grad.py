"""GPU parity of ``gpx_lml_grad`` (log-marginal likelihood + its gradient with respect to the kernel
hyper-parameters, the quantity jax.grad supplies to NUTS / SVI in the reference, gpax/gp.py:515-530,
583-587) against the oracle's closed form (``oracle.gp_oracle.lml_grad``, itself checked against central
differences of the oracle log-marginal in tests/test_oracle.py).

Tolerance: the gradient is a sum of N^2 terms G_ij dK_ij with G = alpha alpha^T - K^-1 and
|K^-1| ~ cond(K): relative 1e-7 of the largest gradient component (measured ~1e-10), written below.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_GRAD_REL = 1e-7
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


CASES = [
    # kind, n, d, ard_length, ard_period, noise
    (0, 25, 1, False, False, 0.006744897501960818),
    (0, 60, 2, True, False, 0.02),        # one block column at either panel width
    (1, 128, 3, True, False, 0.03),
    (0, 300, 3, True, False, 0.05),
    (0, 517, 8, False, False, 0.01),
    (0, 640, 20, True, False, 0.02),      # two dimension chunks
    (1, 257, 2, False, False, 0.01),
    (1, 384, 16, True, False, 0.05),
    (2, 130, 2, True, True, 0.05),
    (2, 200, 1, False, False, 0.01),
    (2, 190, 3, True, False, 0.03),       # ARD length-scales, one shared period
    (2, 190, 3, False, True, 0.03),
]


@pytest.mark.parametrize("kind,n,d,ard_l,ard_p,noise", CASES)
@pytest.mark.parametrize("nb", [0, 64, 128])
def test_lml_grad_matches_oracle(core, kind, n, d, ard_l, ard_p, noise, nb):
    import gpax_amd
    from oracle import gp_oracle as orc

    x, r, rng = _problem(n, d, seed=100 * kind + n)
    k_length = rng.uniform(0.6, 1.6, d) if ard_l else np.array([0.9])
    k_period = (rng.uniform(1.5, 2.5, d) if ard_p else np.array([2.0])) if kind == 2 else None
    k_scale = 1.3
    core.set_option("nb", nb)
    core.set_train(x)
    p = gpax_amd.make_params(kind, k_scale, k_length, k_period, np.array([noise]), 1e-6, True)
    info, logp, grads, alpha = core.lml_grad(p, r)
    core.set_option("nb", 0)   # automatic again
    assert info == 0
    logp_ref, g_ref, alpha_ref = orc.lml_grad(kind, x, r, k_scale, k_length, k_period, noise, 1e-6)
    assert abs(logp - logp_ref) <= TOL_LOGP_REL * abs(logp_ref)
    assert np.abs(alpha - alpha_ref).max() <= TOL_ALPHA_REL * np.abs(alpha_ref).max()
    scale = max(np.abs(v).max() for v in g_ref.values())
    assert set(grads) == set(g_ref)
    for name, ref in g_ref.items():
        assert grads[name].shape == ref.shape, name
        assert np.abs(grads[name] - ref).max() <= TOL_GRAD_REL * scale, (name, grads[name], ref)


def test_lml_grad_agrees_with_central_differences_of_the_gpu_logp(core):
    """Self-consistency on the device: d logp / d theta from gpx_lml_grad against central differences of
    gpx_fit_predict's logp (what ExactGP.fit() used before the closed form existed)."""
    import gpax_amd

    x, r, rng = _problem(700, 4, seed=9)
    k_length = rng.uniform(0.7, 1.4, 4)
    core.set_train(x)
    core.set_test(np.zeros((0, 4)))

    def logp_at(s, ell, nz):
        p = gpax_amd.make_params(0, s, ell, None, np.array([nz]), 1e-6, True)
        info, lp, _, _ = core.fit_predict(p, r, None, 0.0)
        assert info == 0
        return lp

    s, nz, h = 1.1, 0.04, 1e-5
    p = gpax_amd.make_params(0, s, k_length, None, np.array([nz]), 1e-6, True)
    info, logp, grads, _ = core.lml_grad(p, r)
    assert info == 0 and abs(logp - logp_at(s, k_length, nz)) <= 1e-9 * abs(logp)
    fd_s = (logp_at(s + h, k_length, nz) - logp_at(s - h, k_length, nz)) / (2 * h)
    fd_n = (logp_at(s, k_length, nz + h) - logp_at(s, k_length, nz - h)) / (2 * h)
    assert abs(fd_s - grads["k_scale"][0]) <= 1e-5 * max(1.0, abs(fd_s))
    assert abs(fd_n - grads["k_noise"][0]) <= 1e-5 * max(1.0, abs(fd_n))
    for i in range(4):
        e = np.zeros(4)
        e[i] = h
        fd = (logp_at(s, k_length + e, nz) - logp_at(s, k_length - e, nz)) / (2 * h)
        assert abs(fd - grads["k_length"][i]) <= 1e-5 * max(1.0, abs(fd))


def test_lml_grad_leaves_a_usable_factor_cache(core):
    """After the gradient call the cached factor / z serve gpx_posterior like after gpx_fit_predict."""
    import gpax_amd
    from oracle import gp_oracle as orc

    x, r, rng = _problem(333, 2, seed=4)
    x_new = rng.uniform(-1, 1, (50, 2))
    core.set_train(x)
    core.set_test(x_new)
    p = gpax_amd.make_params(0, 1.0, np.array([1.0]), None, np.array([0.01]), 1e-6, True)
    info, logp, _, _ = core.lml_grad(p, r)
    assert info == 0
    info, mean, var, _ = core.posterior(np.ones(50), 0.0)
    assert info == 0
    params = {"k_scale": np.array([1.0]), "k_length": np.array([1.0]), "k_noise": np.array([0.01]), "k_jitter": np.array([1e-6])}
    _, mean_ref, var_ref = orc.fit_predict(0, x, r + 1.0, x_new, params)
    assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8


def test_lml_grad_vector_noise_and_not_pd(core):
    import gpax_amd

    x, r, _ = _problem(200, 2, seed=5)
    core.set_train(x)
    noise = np.full(200, 0.01)
    p = gpax_amd.make_params(0, 1.0, np.array([1.0]), None, noise, 1e-6, True)
    info, logp, grads, _ = core.lml_grad(p, r)
    assert info == 0 and np.isfinite(logp) and grads["k_noise"][0] == 0.0   # a per-point noise vector is data
    noise[137] = -5.0
    p = gpax_amd.make_params(0, 1.0, np.array([1.0]), None, noise, 1e-6, True)
    info, logp, grads, _ = core.lml_grad(p, r)
    assert info == 138 and all(np.isnan(v).all() for v in grads.values())
