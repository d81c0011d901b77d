"""A few fit+predict steps at small N for a rocprofv3 --kernel-trace --memory-copy-trace run (what sits between two steps)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m, d = 100, 1
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.05]), 1e-6)
for _ in range(30): c.fit_predict_async(p, 0.0); c.fetch_results()
