import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpax_amd
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(np.zeros((0, d)))
p = gpax_amd.make_params(0, 1.0, rng.uniform(0.8, 1.2, d), None, np.array([0.0067]), 1e-6)
for _ in range(3): c.lml_grad(p, y)
t0 = time.perf_counter()
for _ in range(10): c.lml_grad(p, y)
print(f"grad {(time.perf_counter()-t0)/10*1e3:.3f} ms")
c.profile_reset(); c.profile_mode(1)
for _ in range(5): c.lml_grad(p, y)
st = {gpax_amd._lib.STAGE_NAMES[i]: c.profile_get(i) for i in range(8)}
c.profile_mode(0)
for k, v in st.items():
    if v["launches"]: print(f"  {k:18s} {v['ms']/5:8.3f} ms/call  {v['launches']//5:4d} launches  {v['flops']/5/(v['ms']/5*1e-3)*1e-12 if v['ms'] else 0:6.1f} TF")
