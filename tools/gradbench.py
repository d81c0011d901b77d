import sys, time, numpy as np
sys.path.insert(0, '.')
import gpax_amd
from oracle import gp_oracle as orc
core = gpax_amd.Core(0)
rng = np.random.default_rng(0)
for n, d, ard in ((1024, 8, True), (4096, 8, False), (4096, 8, True), (8192, 8, True)):
    x = rng.uniform(-1, 1, (n, d)); y = np.sin(3*x.sum(1)/np.sqrt(d)) + 0.1*rng.standard_normal(n)
    r = (y-y.mean())/(y.std()+1e-8) - 1.0
    kl = rng.uniform(0.8, 1.2, d) if ard else np.array([1.0])
    core.set_train(x); core.set_test(np.zeros((0, d)))
    p = gpax_amd.make_params(0, 1.0, kl, None, np.array([0.0067]), 1e-6, True)
    core.lml_grad(p, r)
    t0 = time.perf_counter()
    for _ in range(5): info, logp, g, a = core.lml_grad(p, r)
    tg = (time.perf_counter()-t0)/5
    core.fit_predict(p, r, None, 0.0)
    t0 = time.perf_counter()
    for _ in range(5): core.fit_predict(p, r, None, 0.0)
    tf = (time.perf_counter()-t0)/5
    msg = f"n={n} ard={ard} grad {tg*1e3:.2f} ms  logp-only {tf*1e3:.2f} ms"
    if n <= 4096:
        lr, gr, ar = orc.lml_grad(0, x, r, 1.0, kl, None, 0.0067, 1e-6)
        sc = max(np.abs(v).max() for v in gr.values())
        msg += f"  relerr grad {max(np.abs(g[k]-gr[k]).max() for k in gr)/sc:.2e} alpha {np.abs(a-ar).max()/np.abs(ar).max():.2e} logp {abs(logp-lr)/abs(lr):.2e}"
    print(msg, flush=True)
