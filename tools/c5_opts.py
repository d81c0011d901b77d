"""Config C5's shape (N = 8192, M = 10 000 test points, d = 2, Matern): fit+predict step against the grouping thresholds.
python tools/c5_opts.py "pair_min=28,quad_min=40" "pair_min=1,quad_min=2" ...   [N=8192 M=10000 via env]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n = int(os.environ.get("N", 8192)); m = int(os.environ.get("M", 10000)); d = 2
rng = np.random.default_rng(0)
x = rng.uniform(0, 3, (n, d)); xn = rng.uniform(0, 3, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(1, 1.0, 1.0, None, np.array([0.01]), 1e-6)
fl = n ** 3 / 3 + float(n) * n * (m + 1)
for rep in range(2):
    for cfg in sys.argv[1:]:
        for kv in cfg.split(","):
            c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        for _ in range(2): c.fit_predict_async(p, 0.0); c.fetch_results()
        t0 = time.perf_counter()
        for _ in range(5): c.fit_predict_async(p, 0.0); r = c.fetch_results()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{cfg:28s} N={n} M={m}: {ms:.3f} ms/step  {fl / ms * 1e-9:.1f} TFLOP/s  logp={r[1]:.6f}", flush=True)
