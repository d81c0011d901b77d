import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import gpax_amd
n=4096; d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
K=300
for rep in range(2):
    for pm, qm in ((28, 40), (28, 28), (24, 28), (28, 30), (20, 28), (24, 24)):
        c.set_option("pair_min", pm); c.set_option("quad_min", qm)
        for _ in range(40): c.fit_predict_async(p, 0.0); c.fetch_results()
        t0 = time.perf_counter()
        for _ in range(K): c.fit_predict_async(p, 0.0); r = c.fetch_results()
        print(f"pair_min={pm} quad_min={qm}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step", flush=True)
