"""In-situ duration of every bulk-update launch of one N=4096 step (the kernels' own dispatch timestamps, profile mode 3)."""
import os, sys
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
for _ in range(10): c.fit_predict_async(p, 0.0); c.fetch_results()
c.profile_reset(); c.profile_mode(3)
c.fit_predict_async(p, 0.0); c.fetch_results(); c.synchronize()
path = "/tmp/t_dump.csv"
c._check(c._lib.gpx_profile_dump(c._h, path.encode()), "dump")
rows = [l.strip().split(",") for l in open(path)][1:]
t0 = None
for r in rows:
    st, a, b = int(r[0]) if r[0].isdigit() else r[0], float(r[1]), float(r[2])
    if t0 is None: t0 = a
    print(f"stage {st}: start {a - t0:8.1f} us  dur {b - a:6.1f} us")
