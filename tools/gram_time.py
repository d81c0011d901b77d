"""Gram-build kernel time inside the fit+predict step (the kernel's own dispatch timestamps, profile mode 3) and its HBM
rate against the algorithmic bytes (8 B per lower-tile entry written + the inputs): python tools/gram_time.py [N d kind]
GPX_GRAM_SLAB_MIN=1000000000 selects the 64 x 64-tile kernel (round 2), the default the slab-shaped one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpax_amd import _lib
if os.environ.get("GPX_AB_LIB"):   # time an older build of the library (tools/ab/...)
    os.environ["GPX_LIB_COMPAT"] = "1"
    _lib.LIB_PATH = os.path.abspath(os.environ["GPX_AB_LIB"])
import gpax_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
m = 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(kind, 1.0, 1.0, np.array([2.0]) if kind == 2 else None, np.array([0.0067]), 1e-6)
K = 20 if n <= 4096 else 4
for _ in range(max(K // 4, 2)): c.fit_predict_async(p, 0.0); c.fetch_results()
c.profile_reset(); c.profile_mode(3)
for _ in range(K): c.fit_predict_async(p, 0.0); r = c.fetch_results()
g = c.profile_get(0)
us = g["ms"] / g["launches"] * 1e3
gbs = g["bytes"] / g["launches"] / (us * 1e-6) * 1e-9
print(f"N={n} d={d} kind={kind} slab_min={os.environ.get('GPX_GRAM_SLAB_MIN', 'default')}: gram {us:.1f} us/launch, {gbs:.0f} GB/s = {gbs / 8000:.3f} of 8 TB/s ({gbs / 5700:.2f} of the 5.7 TB/s write ceiling); logp {r[1]:.9g}")
