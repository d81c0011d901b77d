import sys, time
import numpy as np
sys.path.insert(0, '.')
import gpax_amd
from oracle import gp_oracle as orc
c = gpax_amd.Core(0)
for n, m in ((60, 60), (60, 10), (25, 300), (64, 64), (65, 65), (100, 100)):
    x, y, xn, p = orc.synthetic_problem(n, 1, m, seed=3)
    c.set_train(x); c.set_test(xn)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    t0 = time.time()
    info, logp, mean, var = c.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
    lr, mr, vr = orc.fit_predict(0, x, y, xn, p)
    info2, mean2, var2, _ = c.posterior(np.ones(m), 0.0)
    print(n, m, "nb", c.get_option("nb"), "info", info, "logp rel", abs(logp - lr) / abs(lr), "mean", np.abs(mean - mr).max(), "var", np.abs(var - vr).max(),
          "| cached posterior: mean", np.abs(mean2 - mr).max(), "var", np.abs(var2 - vr).max(), f"{time.time()-t0:.2f}s")
