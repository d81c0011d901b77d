"""A/B of one tuning knob of the library: python tools/opt_ab.py NAME V1,V2,... N [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
name, vals, n = sys.argv[1], [int(v) for v in sys.argv[2].split(",")], int(sys.argv[3])
d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
K = int(sys.argv[4]) if len(sys.argv) > 4 else (100 if n <= 4096 else 5)
for rep in range(2):
    for v in vals:
        c.set_option(name, v)
        for _ in range(max(K // 5, 2)): c.fit_predict_async(p, 0.0); c.fetch_results()
        t0 = time.perf_counter()
        for _ in range(K): c.fit_predict_async(p, 0.0); r = c.fetch_results()
        print(f"{name}={v:6d} N={n}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step  logp={r[1]:.6f}", flush=True)
