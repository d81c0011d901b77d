import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, gpax_amd
from bench import synthetic_problem
n,d,m=4096,8,1000
x,y,xn=synthetic_problem(n,d,m)
c=gpax_amd.Core(0)
c.set_train(x); c.set_test(xn); c.set_targets(y-1.0,np.ones(m))
p=gpax_amd.make_params(0,1.0,1.0,None,np.array([0.0067]),1e-6)
for _ in range(3): c.fit_predict_async(p,0.0); c.fetch_results()
c.profile_reset(); c.profile_mode(1)
c.fit_predict_async(p,0.0); c.fetch_results()
c.profile_dump("gpurun_out/timeline.csv")
rows=[l.strip().split(",") for l in open("gpurun_out/timeline.csv")][1:]
names={0:"gram",1:"gramX",2:"POTRF",3:"trsm",4:"BULK",5:"epi",6:"TOTAL",7:"la"}
ev=[(int(r[0]),float(r[1]),float(r[2])) for r in rows]
for s,a,b in ev[:52]: print(f"{names[s]:6s} {a:8.1f} -> {b:8.1f}  ({b-a:6.1f})")
tot=[e for e in ev if e[0]==6][0]; print("total", tot[2]-tot[1])
# critical stream idle estimate: sum of gaps between consecutive potrf/trsm and small-upd events
