"""The reference's first example (examples/00: ExactGP on a noisy 1-d sinusoid, BASELINE.json configs[0]) on the
MI355X-native path: prior samples, NUTS fit, posterior prediction, one Bayesian-optimisation step.

    python examples/quickstart.py          # needs an MI355X and the built library (python -c "import __graft_entry__ as g; g.build()")
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gpax_amd import ExactGP, RBFKernel, state  # noqa: E402
from gpax_amd.acquisition import ExpectedImprovement, UpperConfidenceBound  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    x = rng.uniform(-1.0, 1.0, 256).reshape(-1, 1)                 # SimpleSinusoidal1d.default_dataset(256)-like
    y = np.sin(10.0 * x[:, 0]) + rng.normal(0.0, 0.1, 256)
    x_grid = np.linspace(-1.1, 1.1, 100).reshape(-1, 1)

    state.set_rng_key(0)
    gp = ExactGP(kernel=RBFKernel(), x=x, y=y, hp_samples=100, gp_samples=10, num_warmup=300)
    prior = gp.sample(x_grid)                                      # hyper-parameters from the priors, conditioned on the data
    print("prior-hp samples      :", prior.y.shape, "(hp, gp, M)")

    gp.fit()                                                       # NUTS; every leapfrog step is one GPU call
    info = gp.metadata["mcmc_info"][0]
    hp, _ = gp._get_hp_samples_from_posterior(False)
    print(f"NUTS                  : {info['num_grad_evals']} gradient calls, accept {info['mean_accept']:.2f}, "
          f"divergences {info['divergences']}")
    print("posterior medians     :", {k: float(np.median(v)) for k, v in hp.items()})

    mu, sd = gp.predict(x_grid, fast=True)                         # median hyper-parameters
    rmse = float(np.sqrt(np.mean((mu - np.sin(10.0 * x_grid[:, 0])) ** 2)))
    print(f"fast predict          : rmse vs truth {rmse:.3f}, mean sd {sd.mean():.3f}")
    full = gp.sample(x_grid)                                       # all 100 hyper-parameter draws x 10 GP draws
    print("posterior samples     :", full.y.shape)

    bounds = np.array([[-1.1], [1.1]])
    for acq in (UpperConfidenceBound(bounds, beta=4.0), ExpectedImprovement(bounds)):
        x_star, value = acq.optimize(gp, n=512, method="Halton")
        print(f"{acq.name:22s}: next point {np.ravel(x_star)[0]:+.3f} (value {value:.3f})")


if __name__ == "__main__":
    main()
