"""Generate the committed golden fixtures under ``tests/golden/``.

Run from the repo root:  ``python tests/golden/make_golden.py``

The reference cannot be imported here (``jax``/``numpyro`` absent, see
``oracle/gp_oracle.py`` header), so these vectors come from the *oracle* -- the
NumPy/SciPy restatement of the reference formulae -- on inputs built from the
reference's own fixture RECIPES (re-stated below, not imported):

  * ``dummy_sinusoidal_data``   gpax/_tests/conftest.py:15-30  (numpy seed 0, N=25, 300-pt grid)
  * ``SimpleSinusoidal1d.default_dataset(256)``  gpax/utils/experiments.py:356-365
    with the seed ``state.get_rng_key()`` hands out from the initial key 0
    (gpax/state.py:87-126: ``np.random.seed(0); randint(0, 2**32-1)``)
  * structured-GP data of examples/04_Structured_GP.ipynb cell 4 (numpy seed 1),
    widened to d=2 / N=256 as config C5's family
  * uniform d=8 / d=16 inputs of the benchmark recipe

Each ``.npz`` holds inputs AND expected outputs (K, L, logp, alpha, posterior mean /
cov / sd), so tests on the GPU box need neither the reference nor this script.
"""

from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import gp_oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

HP = dict(
    k_scale=np.array([1.0]),
    k_length=np.array([1.0]),
    k_noise=np.array([orc.K_NOISE_MEDIAN]),
    k_jitter=np.array([orc.K_JITTER]),
)


def _case(name, kind, x, y, x_new, params, mean_fn=orc.constant_mean, y_std=None, observation_noise=False):
    k_params = {k: v for k, v in params.items() if k not in ("k_noise", "k_jitter")}
    noise = (np.asarray(y_std) ** 2) if y_std is not None else params["k_noise"]
    k_train = orc.gram(kind, x, x, k_noise=noise, k_jitter=params["k_jitter"], **k_params)
    m_train = mean_fn(x)
    logp, L, alpha = orc.mvn_log_prob(y, m_train, k_train)
    k_cross = orc.gram(kind, x_new, x, apply_noise=False, **k_params)
    mean, cov = orc.mean_and_covariance(
        kind, x, y, x_new, params, y_std=y_std, observation_noise=observation_noise, use_cholesky=True, mean_fn=mean_fn
    )
    mean_inv, cov_inv = orc.mean_and_covariance(
        kind, x, y, x_new, params, y_std=y_std, observation_noise=observation_noise, use_cholesky=False, mean_fn=mean_fn
    )
    out = dict(
        kind=np.int64(kind),
        x=x,
        y=y,
        x_new=x_new,
        m_train=m_train,
        m_new=mean_fn(x_new),
        observation_noise=np.int64(observation_noise),
        K_tril=k_train[np.tril_indices(k_train.shape[0])],  # K is symmetric: packed lower triangle
        L_tril=L[np.tril_indices(L.shape[0])],
        logp=np.float64(logp),
        alpha=alpha,
        K_cross=k_cross,
        post_mean=mean,
        post_cov=cov,
        post_sd=np.sqrt(np.diag(cov)),
        post_mean_inv=mean_inv,
        post_sd_inv=np.sqrt(np.diag(cov_inv)),
    )
    for k, v in params.items():
        out["hp_" + k] = np.asarray(v, dtype=np.float64)
    if y_std is not None:
        out["y_std"] = np.asarray(y_std, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: N={x.shape[0]} d={x.shape[1]} M={x_new.shape[0]} logp={logp:.12g}")


def dummy_sinusoidal_data():
    """gpax/_tests/conftest.py:15-30 restated."""
    np.random.seed(0)
    x = np.random.uniform(-1.0, 1.0, 25)
    y = np.sin(10 * x) + np.random.normal(0.0, 0.1, 25)
    return x.reshape(-1, 1), y.squeeze(), np.linspace(-1.1, 1.1, 300).reshape(-1, 1)


def simple_sinusoidal_1d(n):
    """gpax/utils/experiments.py:356-365 with gpax/state.py:87-126's first key."""
    np.random.seed(0)
    seed = np.random.randint(low=0, high=2**32 - 1)
    np.random.seed(seed)
    x = np.random.uniform(-1.0, 1.0, n).reshape(-1, 1)
    y = np.atleast_1d(np.sin(10.0 * x).squeeze()) + np.random.normal(0.0, 0.1, n)
    return x, y.squeeze()


def main():
    # 1. reference test fixture after ScaleTransform / NormalizeTransform (gpax/gp.py:122-128 defaults)
    x, y, grid = dummy_sinusoidal_data()
    sx = orc.ScaleOracle().fit(x)
    sy = orc.NormalizeOracle().fit(y)
    xt, yt, gt = sx.forward(x), sy.forward(y).squeeze(), sx.forward(grid)
    _case("sin25_rbf", orc.RBF, xt, yt, gt, HP)
    g3 = gt[::3]  # variants use every third grid point to keep the fixtures small
    _case("sin25_matern", orc.MATERN52, xt, yt, g3, HP)
    _case("sin25_rbf_ystd", orc.RBF, xt, yt, g3, HP, y_std=sy.forward(np.full(25, 0.1), "std").squeeze())
    _case("sin25_rbf_obsnoise", orc.RBF, xt, yt, g3, HP, observation_noise=True)
    per = dict(HP, k_period=np.array([0.7]))
    _case("sin25_periodic", orc.PERIODIC, xt, yt, g3, per)

    # 2. config C1: SimpleSinusoidal1d.default_dataset(256), 100 test points
    x, y = simple_sinusoidal_1d(256)
    sx = orc.ScaleOracle().fit(x)
    sy = orc.NormalizeOracle().fit(y)
    grid = np.linspace(-1.1, 1.1, 100).reshape(-1, 1)
    _case("c1_sin256_rbf", orc.RBF, sx.forward(x), sy.forward(y).squeeze(), sx.forward(grid), HP)

    # 3. C2 family (d=8 RBF), C3 family (d=16 Matern), ARD lengthscale, C5 family (d=2 structured mean)
    for n in (64, 512):
        x, y, xn, p = orc.synthetic_problem(n, 8, 40, seed=n)
        _case(f"c2_d8_n{n}_rbf", orc.RBF, x, y, xn, p)
    x, y, xn, p = orc.synthetic_problem(256, 16, 40, seed=3, kind=orc.MATERN52)
    _case("c3_d16_n256_matern", orc.MATERN52, x, y, xn, p)
    x, y, xn, p = orc.synthetic_problem(200, 3, 33, seed=5)
    p = dict(p, k_length=np.array([0.7, 1.3, 2.1]), k_scale=np.array([1.7]))
    _case("ard_d3_n200_rbf", orc.RBF, x, y, xn, p)
    x, y, xn, p = orc.synthetic_problem(130, 2, 17, seed=6, kind=orc.PERIODIC)
    p = dict(p, k_length=np.array([0.9, 1.4]), k_period=np.array([1.3, 0.8]))
    _case("ard_d2_n130_periodic", orc.PERIODIC, x, y, xn, p)

    rng = np.random.default_rng(1)
    x = rng.uniform(0.0, 3.0, (256, 2))
    y = orc.piecewise_mean(x) + rng.normal(0.0, 0.1, 256)
    xn = rng.uniform(0.0, 3.0, (64, 2))
    _case("c5_d2_n256_matern_piecewise", orc.MATERN52, x, y, xn, HP, mean_fn=orc.piecewise_mean)


if __name__ == "__main__":
    main()
