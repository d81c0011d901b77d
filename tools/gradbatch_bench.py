"""Batched gradient calls (gpx_lml_grad_batch) against the same number of single gpx_lml_grad calls, and the lock-step NUTS
fit against sequential chains: python tools/gradbatch_bench.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
from gpax_amd import state
c = gpax_amd.Core(0)
d = 8
for n, B in ((256, 4), (1024, 4), (1024, 8), (2048, 4), (4096, 4)):
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (n, d)); y = rng.standard_normal(n)
    c.set_train(x)
    plist = [gpax_amd.make_params(0, float(np.exp(0.3 * rng.standard_normal())), np.exp(0.3 * rng.standard_normal(d)), None, np.array([0.0067]), 1e-6, True) for _ in range(B)]
    r = np.stack([y] * B)
    for _ in range(3): c.lml_grad_batch(plist, r); c.lml_grad(plist[0], r[0])
    K = 20 if n <= 1024 else 5
    t0 = time.perf_counter()
    for _ in range(K): c.lml_grad_batch(plist, r)
    tb = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        for i in range(B): c.lml_grad(plist[i], r[i])
    ts = (time.perf_counter() - t0) / K
    print(f"N={n} B={B}: batched {tb * 1e3:.3f} ms per batch ({B / tb:.0f} gradient calls/s), {B} single calls {ts * 1e3:.3f} ms ({B / ts:.0f}/s): {ts / tb:.2f}x", flush=True)
c.close()
rng = np.random.default_rng(3)
for n in (256, 1024):
    x = rng.uniform(-1, 1, (n, 2)); y = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + 0.1 * rng.standard_normal(n)
    for method in ("sequential", "vectorized"):
        state.set_rng_key(17)
        gp = gpax_amd.ExactGP(kernel=gpax_amd.RBFKernel(), x=x, y=y, hp_samples=20, num_warmup=40, num_chains=4, chain_method=method)
        t0 = time.perf_counter(); gp.fit(); t = time.perf_counter() - t0
        ng = sum(i["num_grad_evals"] for i in gp.metadata["mcmc_info"])
        print(f"fit N={n} 4 chains {method}: {t:.2f} s, {ng} gradient evaluations, {ng / t:.0f} /s", flush=True)
