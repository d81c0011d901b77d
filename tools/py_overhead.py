import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from gpax_amd import state
from gpax_amd.gp import ExactGP, _ParamSpace
from gpax_amd.kernels import RBFKernel
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (25, 1)); y = np.sin(10 * x[:, 0]) + 0.1 * rng.standard_normal(25)
gp = ExactGP(kernel=RBFKernel(), x=x, y=y, hp_samples=100, num_warmup=200)
space = _ParamSpace(gp)
u = space.init_to_median()
for _ in range(50): space.value_and_grad(gp, u)
t0 = time.perf_counter()
for _ in range(1000): space.value_and_grad(gp, u)
print("value_and_grad per call: %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
core = state.get_core()
p = gp._kernel_struct({"k_scale": np.array([1.0]), "k_length": np.array([1.0])}, np.array([0.01]), np.array([1e-6]), 25)
r = gp.y_transformed - 1.0
t0 = time.perf_counter()
for _ in range(1000): core.lml_grad(p, r)
print("core.lml_grad per call: %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): space.value_and_grad(gp, u)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
state.set_rng_key(1)
t0 = time.perf_counter(); gp.fit(); dt = time.perf_counter() - t0
print("fit: %.2f s for 300 iterations -> %.0f it/s, %d grad calls" % (dt, 300 / dt, gp.metadata["mcmc_info"][0]["num_grad_evals"]))
