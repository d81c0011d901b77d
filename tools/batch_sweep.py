"""Grouping thresholds x batch size at one N: ms per batch, points/s, aggregate TFLOP/s.
   python tools/batch_sweep.py [N] [B,B,...] [pair:quad,pair:quad,...] [la_rows]"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import gpax_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Bs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8]
grid = [tuple(int(u) for u in v.split(":")) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(28, 40), (16, 40), (8, 40), (4, 40), (8, 16), (4, 8), (2, 4)]
la = int(sys.argv[4]) if len(sys.argv) > 4 else None
d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn)
if la is not None: c.set_option("la_rows", la)
flops = n ** 3 / 3 + n * n * (m + 1)
for B in Bs:
    plist = [gpax_amd.make_params(0, 1.0 + 0.05 * i, 1.0, None, np.array([0.0067]), 1e-6) for i in range(B)]
    r = np.stack([y] * B); mn = np.ones((B, m))
    for pm, qm in grid:
        c.set_option("pair_min", pm); c.set_option("quad_min", qm)
        reps = max(3, int(40 / B)) if n <= 4096 else 3
        for _ in range(2): res = c.fit_predict_batch(plist, r, mn, None)
        t0 = time.perf_counter()
        for _ in range(reps): res = c.fit_predict_batch(plist, r, mn, None)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"N={n} B={B:3d} pair_min={pm:3d} quad_min={qm:3d} la_rows={c.get_option('la_rows')}: {ms:8.3f} ms/batch {ms / B:7.3f} ms/draw "
              f"{n * B / ms * 1e-3:6.2f} M points/s {B * flops / ms * 1e-9:5.1f} TFLOP/s  logp0={res[1][0]:.6f}", flush=True)
