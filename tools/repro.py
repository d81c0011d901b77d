import sys
import numpy as np
sys.path.insert(0, '.')
import gpax_amd
c = gpax_amd.Core(0)
rng = np.random.default_rng(0)
def grad(n, d=3):
    x = rng.uniform(-1, 1, (n, d)); y = rng.standard_normal(n)
    c.set_train(x)
    p = gpax_amd.make_params(0, 1.3, rng.uniform(0.6, 1.6, d), None, np.array([0.05]), 1e-6)
    try:
        info, logp, g, a = c.lml_grad(p, y)
        print("grad n", n, "nb", c.get_option("nb"), "info", info, "logp", logp, flush=True)
    except Exception as e:
        print("grad n", n, "FAILED", str(e)[:80], flush=True)
for seq in ([300], [25, 300], [128, 300], [300, 300]):
    c.set_option("resident", 2)
    print("sequence", seq)
    for n in seq: grad(n)
