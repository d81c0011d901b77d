"""Where a fit+predict step's wall time goes, with three event pairs per step (profile mode 2): the whole
enqueued step on the chain stream (gram start .. epilogue end), the gram launches and the epilogue."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gpax_amd
n, d, m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
for mode in (0, 2):
    c.profile_reset(); c.profile_mode(mode)
    for _ in range(10): c.fit_predict_async(p, 0.0); c.fetch_results()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K): c.fit_predict_async(p, 0.0); c.fetch_results()
    wall = (time.perf_counter() - t0) / K * 1e3
    msg = f"mode {mode}: wall {wall:.3f} ms/step"
    if mode:
        st = {gpax_amd._lib.STAGE_NAMES[i]: c.profile_get(i) for i in range(8)}
        tot = st["factor_total"]["ms"] / st["factor_total"]["launches"]
        gr = st["gram"]["ms"] / (st["factor_total"]["launches"])
        ep = st["epilogue"]["ms"] / st["factor_total"]["launches"]
        msg += f" | on-GPU step {tot:.3f} ms (gram launches {gr:.3f}, epilogue {ep:.3f}) -> host gap between steps {wall - tot:.3f} ms"
    print(msg)
c.profile_mode(0)
