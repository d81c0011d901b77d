"""End-to-end ExactGP.fit() (NUTS on the GPU log-marginal + closed-form gradient): wall time, gradient calls, ms per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpax_amd import state
from gpax_amd.gp import ExactGP, VariationalInferenceGP
from gpax_amd.kernels import RBFKernel
from gpax_amd.utils.prior_utils import LogNormal, Parameter
for n, d in ((256, 1), (1024, 4), (4096, 8)):
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (n, d)); y = np.sin(3 * x.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(n)
    state.set_rng_key(1)
    gp = ExactGP(kernel=RBFKernel(k_length=Parameter(LogNormal(0.0, 1.0), d)), x=x, y=y, hp_samples=50, num_warmup=100)
    t0 = time.perf_counter(); gp.fit(); dt = time.perf_counter() - t0
    info = gp.metadata["mcmc_info"][0]
    print(f"N={n} d={d}: NUTS 100 warm-up + 50 draws: {dt:.2f} s, {info['num_grad_evals']} gradient calls, {dt / info['num_grad_evals'] * 1e3:.2f} ms/call, "
          f"accept {info['mean_accept']:.2f}, divergences {info['divergences']}")
    with state.silent_mode():
        vi = VariationalInferenceGP(kernel=RBFKernel(k_length=Parameter(LogNormal(0.0, 1.0), d)), x=x, y=y, num_steps=200, optimizer_kwargs={"step_size": 0.02})
        t0 = time.perf_counter(); vi.fit(); dt = time.perf_counter() - t0
    print(f"        SVI 200 steps: {dt:.2f} s ({dt / 200 * 1e3:.2f} ms/step)")
