import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import gpax_amd
rng = np.random.default_rng(1)
c = gpax_amd.Core(0)
t0 = time.time()
for n, m, reps, kind in ((8192, 500, 40, 0), (12000, 100, 12, 1), (16384, 1000, 6, 0), (1000, 3000, 200, 2)):
    d = 3
    x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
    c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
    per = rng.uniform(1.5, 2.5, d) if kind == 2 else None
    p = gpax_amd.make_params(kind, 1.1, rng.uniform(0.5, 1.5, d), per, np.array([0.02]), 1e-6)
    ref = None
    for i in range(reps):
        c.fit_predict_async(p, 0.0)
        info, logp, mean, var = c.fetch_results()
        assert info == 0, (n, i, info)
        if ref is None: ref = (logp, mean.copy(), var.copy())
        else: assert logp == ref[0] and np.array_equal(mean, ref[1]) and np.array_equal(var, ref[2]), (n, i)
        if i % 5 == 0:
            info, lp, g, a = c.lml_grad(p, y)
            assert info == 0 and abs(lp - logp) <= 1e-9 * abs(logp), (n, i, lp, logp)
            if i == 0: gref = {k: v.copy() for k, v in g.items()}
            else: assert all(np.array_equal(g[k], gref[k]) for k in g), (n, i)
    print(f"n={n} m={m} kind={kind}: {reps} steps ok, logp={ref[0]:.6f}", flush=True)
print(f"soak2 ok in {time.time()-t0:.1f} s")
