"""The reference's structured-GP example (examples/04_Structured_GP.ipynb, BASELINE.json configs[4] in miniature) on the
MI355X-native path: a piecewise power-law prior mean with learnable ``m_t``, ``m_beta1``, ``m_beta2`` under NUTS.

The reference differentiates the mean through ``jax.grad`` (``jnp.piecewise`` has zero derivative with respect to the break
point almost everywhere); here the mean ships the same derivatives in closed form through ``Mean._mean_grad``, which
``ExactGP.fit()`` contracts with ``alpha = K^-1 (y - m)`` from the gradient call of the HIP library.

    python examples/structured_gp.py          # needs an MI355X and the built library
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gpax_amd.means import Mean  # noqa: E402
from gpax_amd.utils.prior_utils import LogNormal, Parameter, Uniform  # noqa: E402


class PiecewiseMeanPrior(Mean):
    """``where(x < t, x**beta1, x**beta2)`` on the first input column (notebook cell 10), priors as there:
    t ~ Uniform(0.5, 2.5), beta1, beta2 ~ LogNormal(0, 1)."""

    def __init__(self):
        self.m_t = Parameter(Uniform(0.5, 2.5))
        self.m_beta1 = Parameter(LogNormal(0.0, 1.0))
        self.m_beta2 = Parameter(LogNormal(0.0, 1.0))

    @staticmethod
    def _x0(x):
        x = np.asarray(x, dtype=np.float64)
        return x[:, 0] if x.ndim == 2 else x

    def _mean_function(self, x, m_t, m_beta1, m_beta2):
        x0 = self._x0(x)
        t, b1, b2 = (float(np.asarray(v).reshape(-1)[0]) for v in (m_t, m_beta1, m_beta2))
        return np.where(x0 < t, x0**b1, x0**b2)

    def _mean_grad(self, x, m_t, m_beta1, m_beta2):
        """d/dbeta x**beta = x**beta log x on the branch that is active; d/dt = 0 almost everywhere."""
        x0 = self._x0(x)
        t, b1, b2 = (float(np.asarray(v).reshape(-1)[0]) for v in (m_t, m_beta1, m_beta2))
        left = x0 < t
        with np.errstate(divide="ignore", invalid="ignore"):
            lx = np.where(x0 > 0, np.log(np.where(x0 > 0, x0, 1.0)), 0.0)
        return {"m_t": np.zeros_like(x0), "m_beta1": np.where(left, x0**b1 * lx, 0.0), "m_beta2": np.where(left, 0.0, x0**b2 * lx)}


def main():
    import gpax_amd
    from gpax_amd.gp import ExactGP
    from gpax_amd.kernels import MaternKernel

    gpax_amd.state.set_rng_key(1)
    rng = np.random.default_rng(1)
    true = {"m_t": 1.7, "m_beta1": 4.5, "m_beta2": 2.5}
    x = rng.uniform(0.0, 3.0, (15, 1))
    mean = PiecewiseMeanPrior()
    y = mean(x, **true) + rng.normal(0.0, 0.1, 15)
    gp = ExactGP(kernel=MaternKernel(), mean=mean, input_transform=None, output_transform=None, x=x, y=y, y_std=None,
                 hp_samples=200, num_warmup=300, observation_noise=False)
    gp.fit()
    for k in ("m_t", "m_beta1", "m_beta2"):
        v = gp.mcmc[k].ravel()
        print(f"{k}: posterior median {np.median(v):.3f} (true {true[k]})")
    x_grid = np.linspace(0.0, 3.0, 200)[:, None]
    mu, sd = gp.predict(x_grid, fast=True)
    print("max |mu - truth| on the grid:", float(np.abs(mu - mean(x_grid, **true)).max()))


if __name__ == "__main__":
    main()
