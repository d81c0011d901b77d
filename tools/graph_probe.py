import subprocess, sys
code = r'''
import sys, numpy as np
sys.path.insert(0, ".")
n, m, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if depth >= 10:
    import torch
    torch.cuda.set_device(0)
    if depth >= 20: torch.cuda.synchronize()
    depth = 3
import gpax_amd
core = gpax_amd.Core(0)
core.set_option("depth", depth)
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, 4)); xn = rng.uniform(-1, 1, (m, 4)); y = rng.standard_normal(n)
p = gpax_amd.make_params(0, 1.0, np.array([1.0]), None, np.array([0.01]), 1e-6, True)
core.set_train(x); core.set_test(xn); core.set_targets(y, np.ones(m))
core.fit_predict_async(p, 0.0); core.fetch_results()
core.set_option("graph", 1)
for _ in range(3):
    core.fit_predict_async(p, 0.0); r = core.fetch_results()
print("ok", n, m, depth, r[1])
'''
for n, m, depth in [(512, 1000, 3), (512, 1000, 10), (512, 1000, 20)]:
    r = subprocess.run([sys.executable, "-c", code, str(n), str(m), str(depth)], capture_output=True, text=True)
    print(n, m, depth, "rc", r.returncode, r.stdout.strip()[-60:])
