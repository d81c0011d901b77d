import sys, time, numpy as np
sys.path.insert(0, '.')
import gpax_amd
core = gpax_amd.Core(0)
n, d, m = int(sys.argv[1]), 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d))
y = np.sin(3*x.sum(1)/np.sqrt(d)) + 0.1*rng.standard_normal(n); y = (y-y.mean())/y.std()
p = gpax_amd.make_params(0, 1.0, np.array([1.0]), None, np.array([0.0067]), 1e-6, True)
core.set_train(x); core.set_test(xn); core.set_targets(y-1.0, np.ones(m))
for _ in range(2):
    core.fit_predict_async(p, 0.0); core.fetch_results()
t0 = time.perf_counter()
for _ in range(4):
    core.fit_predict_async(p, 0.0); r = core.fetch_results()
dt = (time.perf_counter()-t0)/4
print(f"n={n} {dt*1e3:.2f} ms  {(n**3/3+n*n*(m+1))/dt*1e-12:.1f} TF  logp={r[1]:.6f}")
