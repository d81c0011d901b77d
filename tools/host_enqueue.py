"""How long the host needs to ENQUEUE one fit+predict step (gpx_fit_predict_async returns when everything is queued)
against the step itself: python tools/host_enqueue.py [N ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
c = gpax_amd.Core(0)
for n in [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192]:
    d, m = 8, 1000
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
    c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
    p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
    for _ in range(10): c.fit_predict_async(p, 0.0); c.fetch_results()
    K = 50
    te = tf = 0.0
    for _ in range(K):
        t0 = time.perf_counter(); c.fit_predict_async(p, 0.0); t1 = time.perf_counter(); c.fetch_results(); t2 = time.perf_counter()
        te += t1 - t0; tf += t2 - t1
    print(f"N={n}: enqueue {te / K * 1e3:.3f} ms, then fetch (wait + copy) {tf / K * 1e3:.3f} ms; step {(te + tf) / K * 1e3:.3f} ms")
