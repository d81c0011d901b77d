"""Cached-factor posterior (gpx_posterior after gpx_fit_predict): the acquisition loop's unit of work (C5)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 2
rng = np.random.default_rng(0)
x = rng.uniform(0, 3, (n, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x)
p = gpax_amd.make_params(1, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
c.set_test(np.zeros((0, d)))
t0 = time.perf_counter(); c.fit_predict(p, y, None, 0.0); t1 = time.perf_counter()
print(f"N={n}: fit (factor + logp) {1e3 * (t1 - t0):.2f} ms (first call)")
for m in (100, 1000, 10000):
    xn = rng.uniform(0, 3, (m, d))
    c.set_test(xn)
    c.posterior(np.zeros(m), 0.0)
    t0 = time.perf_counter()
    for _ in range(3): info, mean, var, _ = c.posterior(np.zeros(m), 0.0)
    dt = (time.perf_counter() - t0) / 3
    print(f"  posterior mean+var for M={m}: {dt * 1e3:.2f} ms  ({n * n * m / dt * 1e-12:.1f} TFLOP/s on N^2 M)")
