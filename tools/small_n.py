"""Latency of one fit+predict step and one gradient call at small N (the sizes of the reference's examples)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpax_amd import _lib
if os.environ.get("GPX_AB_LIB"):   # time an older build of the library (tools/ab/...)
    os.environ["GPX_LIB_COMPAT"] = "1"
    _lib.LIB_PATH = os.path.abspath(os.environ["GPX_AB_LIB"])
import gpax_amd
c = gpax_amd.Core(0)
rng = np.random.default_rng(0)
for n, m in ((25, 300), (64, 100), (128, 100), (256, 100), (512, 100), (1024, 100)):
    d = 1
    x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
    c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
    p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.05]), 1e-6)
    for _ in range(20): c.fit_predict_async(p, 0.0); c.fetch_results()
    t0 = time.perf_counter()
    for _ in range(200): c.fit_predict_async(p, 0.0); c.fetch_results()
    tf = (time.perf_counter() - t0) / 200
    for _ in range(10): c.lml_grad(p, y)
    t0 = time.perf_counter()
    for _ in range(200): c.lml_grad(p, y)
    tg = (time.perf_counter() - t0) / 200
    print(f"N={n:5d} M={m}: fit+predict {tf*1e6:7.1f} us   logp+grad {tg*1e6:7.1f} us")
