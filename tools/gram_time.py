import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from gpax_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "default": _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import gpax_amd
n=4096; d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
for _ in range(20): c.fit_predict_async(p, 0.0); c.fetch_results()
c.profile_reset(); c.profile_mode(3)
for _ in range(50): c.fit_predict_async(p, 0.0); r = c.fetch_results()
g = c.profile_get(0)
print(sys.argv[1] if len(sys.argv) > 1 else "default", "gram us/launch", g["ms"]/g["launches"]*1e3, "logp", r[1])
