import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import codes_of_ipd_ssn_amg_method_amd as ipd
N = int(sys.argv[1]); out = sys.argv[2]
rs = np.random.RandomState(1)
c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
l = l * r.sum() / l.sum()
one = np.ones(N)
ws = ipd.APDWorkspace(1, c, r, l, one, one, gama=np.inf)
ws.warmup(0.0, 100)
amg = dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5, cycle="w", isnsp=1, inter=1)
res = ws.run(amg, ipd.MatlabRand(5489), iters=int(sys.argv[3]) if len(sys.argv) > 3 else 100)
recs = ws.records()
with open(out, "w") as f:
    for r_ in recs:
        f.write(json.dumps({k: (float(v) if isinstance(v, (float, np.floating)) else int(v) if isinstance(v, (int, np.integer)) else str(v)) for k, v in r_.items()}) + "\n")
print(res["k"], res["converged"], len(recs))
