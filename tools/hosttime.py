import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, gpax_amd
from bench import synthetic_problem
n,d,m=4096,8,1000
x,y,xn=synthetic_problem(n,d,m)
for graph in (0,1):
  for la in (1,0):
    c=gpax_amd.Core(0); c.set_option("graph",graph); c.set_option("lookahead",la)
    c.set_train(x); c.set_test(xn); c.set_targets(y-1.0,np.ones(m))
    p=gpax_amd.make_params(0,1.0,1.0,None,np.array([0.0067]),1e-6)
    for _ in range(3): c.fit_predict_async(p,0.0); c.fetch_results()
    te=[];tt=[]
    for _ in range(10):
        t0=time.perf_counter(); c.fit_predict_async(p,0.0); t1=time.perf_counter(); c.fetch_results(); t2=time.perf_counter()
        te.append(t1-t0); tt.append(t2-t0)
    print(f"graph={graph} lookahead={la}: host enqueue {np.median(te)*1e3:.3f} ms, total {np.median(tt)*1e3:.3f} ms")
    c.close()
