"""The RCCL driver on ONE GPU through a one-rank communicator, torch imported first (the order bench.py uses):
python tools/dist_single.py N [steps]"""
import os, sys, time
import torch  # noqa: F401  (first, as in bench.py: its bundled ROCm libraries are the ones that get loaded)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
from gpax_amd.dist import DistributedStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
for force, early in ((False, "0"), (True, "0"), (True, "1")):
    os.environ["GPX_DIST_EARLY_DIAG"] = early
    st = DistributedStep(x, xn, y, device=0, force_comm=force)
    st.fit_predict(p, np.ones(m), 0.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        info, logp, mean, var = st.fit_predict(p, np.ones(m), 0.0)
    dt = (time.perf_counter() - t0) / steps
    print(f"N={n} one rank, communicator={force}, early diagonal={early}: {dt * 1e3:.2f} ms/step  info={info} logp={logp:.6f}", flush=True)
    st.close()
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
c.fit_predict_async(p, 0.0); c.fetch_results()
t0 = time.perf_counter(); c.fit_predict_async(p, 0.0); r = c.fetch_results()
print(f"single-GPU schedule: {(time.perf_counter() - t0) * 1e3:.2f} ms/step logp={r[1]:.6f}")
