"""Soak test of the default schedule (in-kernel hand-overs): many steps at several sizes, every result compared
bitwise with the first one of its size (the schedule is deterministic) and info checked."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpax_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "default":   # python tools/soak.py [lib.so]: a diagnostic build of the library
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import gpax_amd
rng = np.random.default_rng(0)
c = gpax_amd.Core(0)
t0 = time.time()
total = 0
# (200, 250) and (256, 1000): two block columns with a tall border -- the shape at which the last border panel once raced
# with the panel stream's last column update
for n, m, reps in ((200, 250, 3000), (256, 1000, 2000), (300, 17, 400), (700, 50, 400), (1500, 100, 300), (2500, 333, 200), (4096, 1000, 300), (6000, 10, 50)):
    d = 5
    c.set_option("split_min_blocks", 2 if n <= 256 else 3)   # two block columns run on one stream by default: soak the split path there too
    x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
    c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
    p = gpax_amd.make_params(1, 1.3, rng.uniform(0.5, 1.5, d), None, np.array([0.01]), 1e-6)
    ref = None
    for i in range(reps):
        c.fit_predict_async(p, 0.0)
        info, logp, mean, var = c.fetch_results()
        assert info == 0, (n, i, info)
        if ref is None: ref = (logp, mean.copy(), var.copy())
        else: assert logp == ref[0] and np.array_equal(mean, ref[1]) and np.array_equal(var, ref[2]), (n, i)
        if i % 7 == 0:
            info, lp, g, a = c.lml_grad(p, y)
            assert info == 0 and abs(lp - logp) <= 1e-9 * abs(logp), (n, i, lp, logp)
        total += 1
    print(f"n={n}: {reps} steps ok, logp={ref[0]:.6f}", flush=True)
print(f"soak ok: {total} steps in {time.time()-t0:.1f} s")
