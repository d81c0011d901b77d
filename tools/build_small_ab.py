import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import gpax_amd
from oracle import gp_oracle as orc
c = gpax_amd.Core(0)
for n, m, d, kind in ((25, 300, 1, 0), (64, 100, 2, 1), (100, 50, 3, 2), (128, 100, 8, 0), (37, 1, 2, 0)):
    x, y, xn, p = orc.synthetic_problem(n, d, m, seed=n, kind=kind)
    prm = gpax_amd.make_params(kind, p["k_scale"], p["k_length"], p.get("k_period"), p["k_noise"], p["k_jitter"])
    c.set_train(x); c.set_test(xn)
    ref = orc.fit_predict(kind, x, y, xn, p)
    res = {}
    for bs in (0, 1):
        c.set_option("build_small", bs)
        for _ in range(20): r = c.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
        t0 = time.perf_counter()
        for _ in range(300): r = c.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
        res[bs] = (r, (time.perf_counter() - t0) / 300 * 1e6)
    same = res[0][0][1] == res[1][0][1] and np.array_equal(res[0][0][2], res[1][0][2]) and np.array_equal(res[0][0][3], res[1][0][3])
    ok = abs(res[1][0][1] - ref[0]) <= 1e-10 * abs(ref[0]) and np.abs(res[1][0][2] - ref[1]).max() <= 1e-8
    print(f"n={n} m={m} kind={kind}: separate {res[0][1]:.1f} us, one launch {res[1][1]:.1f} us, bit-identical={same}, oracle ok={ok}")
