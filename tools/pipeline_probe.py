"""What the per-step host synchronisation costs: K steps with a fetch after each one against K steps enqueued back to back
(one fetch at the end; the steps are independent, the streams order them).  python tools/pipeline_probe.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
K = 100 if n <= 4096 else 10
for _ in range(10): c.fit_predict_async(p, 0.0); c.fetch_results()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K): c.fit_predict_async(p, 0.0); r = c.fetch_results()
    a = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K): c.fit_predict_async(p, 0.0)
    r2 = c.fetch_results()
    b = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K): c.fit_predict_async(p, 0.0)
    e = (time.perf_counter() - t0) / K
    c.fetch_results()
    print(f"N={n}: fetch per step {a * 1e3:.3f} ms/step; back to back {b * 1e3:.3f} ms/step (host enqueue alone {e * 1e3:.3f}); logp {r[1]:.6f} {r2[1]:.6f}", flush=True)
