"""gpx_posterior against a cached factor (predict on new points after a fit): python tools/posterior_bench.py [N M d kind]
GPX_SOLVE_GROUP=1 selects round 2's column-by-column solve."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 2
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(0)
x = rng.uniform(0, 3, (n, d)); xn = rng.uniform(0, 3, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn)
p = gpax_amd.make_params(kind, 1.0, 1.0, None, np.array([0.01]), 1e-6)
c.fit_factor(p)
c.log_marginal(y)
reps = 5 if n * m > 2e7 else 20
for _ in range(2): r = c.posterior(np.zeros(m), 0.0)
t0 = time.perf_counter()
for _ in range(reps): r = c.posterior(np.zeros(m), 0.0)
ms = (time.perf_counter() - t0) / reps * 1e3
fl = float(n) * n * m + 2.0 * n * m
print(f"N={n} M={m} d={d} kind={kind} group={os.environ.get('GPX_SOLVE_GROUP', 'default')}: posterior {ms:.3f} ms  {fl / ms * 1e-9:.1f} TFLOP/s  mean[0]={r[1][0]:.12g} var[0]={r[2][0]:.6e}")
