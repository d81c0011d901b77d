"""Edge shapes of the fit+predict path against the oracle (run after schedule changes): python tools/edge_cases.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
from oracle import gp_oracle as orc
c = gpax_amd.Core(0)
bad = 0
for n, m, d, kind in ((129, 1, 2, 0), (200, 3200, 3, 0), (257, 5, 1, 1), (385, 3300, 4, 0), (256, 128, 2, 2), (640, 129, 8, 1), (1025, 1, 3, 0),
                      (130, 6500, 2, 0), (384, 0, 2, 0)):
    x, y, xn, p = orc.synthetic_problem(n, d, max(m, 1), seed=n + m, kind=kind)
    per = p.get("k_period")
    prm = gpax_amd.make_params(kind, p["k_scale"], p["k_length"], per, p["k_noise"], p["k_jitter"])
    c.set_train(x)
    if m > 0:
        c.set_test(xn)
        ref = orc.fit_predict(kind, x, y, xn, p)
        for rep in range(3):
            info, logp, mean, var = c.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
            ok = info == 0 and abs(logp - ref[0]) <= 1e-10 * abs(ref[0]) and np.abs(mean - ref[1]).max() <= 1e-8 and np.abs(var - ref[2]).max() <= 1e-8
            if not ok: bad += 1
        print(f"n={n} m={m} d={d} kind={kind}: {'ok' if ok else 'MISMATCH'}  logp={logp:.9g} (oracle {ref[0]:.9g}) max|dmean|={np.abs(mean - ref[1]).max():.1e}", flush=True)
    else:
        k = [orc.rbf, orc.matern52][kind](x, x, p["k_scale"], p["k_length"], p["k_noise"], p["k_jitter"])
        logp_ref = orc.mvn_log_prob(y, np.ones(n), k)[0]
        c.fit_factor(prm)
        lp = c.log_marginal(y - 1.0)
        lp = lp[1] if isinstance(lp, tuple) and len(lp) > 1 else (lp[0] if isinstance(lp, tuple) else lp)
        ok = abs(lp - logp_ref) <= 1e-10 * abs(logp_ref)
        if not ok: bad += 1
        print(f"n={n} m=0: {'ok' if ok else 'MISMATCH'} logp={lp:.9g} (oracle {logp_ref:.9g})", flush=True)
print("edge cases:", "all ok" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
