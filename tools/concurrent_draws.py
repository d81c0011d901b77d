"""Aggregate throughput of K independent fit+predict pipelines on ONE GPU (K handles, K host threads):
python tools/concurrent_draws.py N [K1,K2,...] [steps]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
for K in ks:
    cores = []
    for i in range(K):
        c = gpax_amd.Core(0)
        c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
        cores.append(c)
    prm = [gpax_amd.make_params(0, 1.0 + 0.01 * i, 1.0, None, np.array([0.0067]), 1e-6) for i in range(K)]
    out = [None] * K
    def work(i, reps):
        for _ in range(reps):
            cores[i].fit_predict_async(prm[i], 0.0)
            out[i] = cores[i].fetch_results()
    def run(reps):
        th = [threading.Thread(target=work, args=(i, reps)) for i in range(K)]
        for t in th: t.start()
        for t in th: t.join()
    run(5)
    t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
    print(f"N={n} K={K}: {dt / steps * 1e3:.3f} ms per round of {K} steps -> {n * K * steps / dt:.3e} points/s  ({dt / steps / K * 1e3:.3f} ms per step)  logp0={out[0][1]:.4f}", flush=True)
    for c in cores: c.close()
