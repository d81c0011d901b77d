"""A/B of option SETS with the bulk kernel's in-situ rate: python tools/opt_ab3.py N "a=1,b=2" "a=0,b=2" ...
Per configuration: ms per step (free-running), then a profile-mode-3 pass: bulk launches per step, their summed time per step,
their average rate (TFLOP/s, algorithmic flops / the kernels' own dispatch timestamps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n = int(sys.argv[1]); cfgs = sys.argv[2:]
d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
K = 100 if n <= 4096 else (20 if n <= 8192 else 5)
REPS = int(os.environ.get("REPS", "3"))
for rep in range(REPS):
    for cfg in cfgs:
        for kv in cfg.split(","):
            c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        for _ in range(max(K // 5, 2)): c.fit_predict_async(p, 0.0); c.fetch_results()
        t0 = time.perf_counter()
        for _ in range(K): c.fit_predict_async(p, 0.0); r = c.fetch_results()
        ms = (time.perf_counter() - t0) / K * 1e3
        c.profile_reset(); c.profile_mode(3)
        for _ in range(K): c.fit_predict_async(p, 0.0); c.fetch_results()
        tr = c.profile_get(gpax_amd._lib.STAGE_NAMES.index("trailing_update"))
        c.profile_mode(0)
        print(f"{cfg:36s} N={n}: {ms:.3f} ms/step  bulk: {tr['launches'] / K:.0f} launches, {tr['ms'] / K:.3f} ms/step, "
              f"{tr['flops'] / (tr['ms'] * 1e-3) * 1e-12:.1f} TFLOP/s  logp={r[1]:.6f}", flush=True)
