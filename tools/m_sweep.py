"""Step time against the number of test points M at fixed N (what the border rows cost the factorisation): python tools/m_sweep.py N M1 M2 ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n = int(sys.argv[1]); ms = [int(a) for a in sys.argv[2:]]
d = 8
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x)
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
K = 100 if n <= 4096 else 10
for rep in range(2):
    for m in ms:
        xn = rng.uniform(-1, 1, (m, d))
        c.set_test(xn); c.set_targets(y, np.ones(m))
        for _ in range(max(K // 5, 2)): c.fit_predict_async(p, 0.0); c.fetch_results()
        t0 = time.perf_counter()
        for _ in range(K): c.fit_predict_async(p, 0.0); r = c.fetch_results()
        dt = (time.perf_counter() - t0) / K
        fl = n ** 3 / 3 + n * n * m
        print(f"N={n} M={m}: {dt * 1e3:.3f} ms/step  {fl / dt * 1e-12:.1f} TFLOP/s  logp={r[1]:.6f}", flush=True)
