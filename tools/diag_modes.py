"""Which outputs differ between hand-over modes / batch at one size (diagnostic)."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import gpax_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d = 8
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
c = gpax_amd.Core(0)
c.set_train(x); c.set_test(xn)
p = gpax_amd.make_params(0, 1.3, 1.1, None, np.array([0.0067]), 1e-6)
out = {}
for res in (2, 1, 0):
    c.set_option("resident", res)
    _, _, L = c.fit_factor(p, want_l=True)
    fp = c.fit_predict(p, y, np.ones(m), 0.0)
    out[res] = (L, fp)
c.set_option("resident", 2)
c.set_option("batch_group", 0)
b = c.fit_predict_batch([p, p, p, p], np.stack([y] * 4), np.ones((4, m)), None)
ref = out[2]
for res in (1, 0):
    L, fp = out[res]
    dl = np.abs(L - ref[0]); 
    print(f"resident {res} vs 2: L differs at {np.count_nonzero(dl)} entries; first rows {np.unique(np.nonzero(dl)[0])[:5]} cols {np.unique(np.nonzero(dl)[1])[:5]}; logp equal {fp[1] == ref[1][1]}; "
          f"mean diff entries {np.count_nonzero(fp[2] != ref[1][2])}; var diff entries {np.count_nonzero(fp[3] != ref[1][3])}")
for i in range(4):
    print(f"batch draw {i} vs single resident2: logp equal {b[1][i] == ref[1][1]}, mean diff {np.count_nonzero(b[2][i] != ref[1][2])} (idx {np.nonzero(b[2][i] != ref[1][2])[0][:8]}), var diff {np.count_nonzero(b[3][i] != ref[1][3])}"
          f" | vs resident0: mean diff {np.count_nonzero(b[2][i] != out[0][1][2])}")
