import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import codes_of_ipd_ssn_amg_method_amd as ipd
N = int(sys.argv[1]); iters = int(sys.argv[2])
rs = np.random.RandomState(1)
c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
one = np.ones(N)
ws = ipd.APDWorkspace(2, c, r, l, one, one, mu=0.65 * min(r.sum(), l.sum()), phi=np.ones(N * N))
amg = dict(retol=1e-11, bigph=1, maxit=40, theta=1 / 4, smoth=10, cycle="w", isnsp=1, inter=1)
ws.warmup(0.0, 100)
for k in range(1, iters + 1):
    t0 = time.perf_counter()
    out = ws.run(amg, ipd.MatlabRand(5489) if k == 1 else None, iters=k) if False else None
    break
t0 = time.perf_counter()
out = ws.run(amg, ipd.MatlabRand(5489), iters=iters)
print("iters", iters, "time %.3f" % (time.perf_counter() - t0), out["k"], out["nrec"], flush=True)
recs = ws.records()
print([ (r_["k"], r_["itamg"], r_["info0"]) for r_ in recs][-12:])
