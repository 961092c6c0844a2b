import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np
import codes_of_ipd_ssn_amg_method_amd as ipd
from bench_driver import problem
G = "tests/golden"
def approx(recs):
    return sum(1 for a, b in zip(recs, recs[1:]) if a["k"] == b["k"] and a["E"] == b["E"] and a["info0"] == b["info0"])
d = np.load(f"{G}/data1_500.npz"); o = np.ones(500)
out = ipd.APD_SsN_Class1(d["c"], d["r"], d["l"], o, o, np.inf, rng=ipd.MatlabRand(5489))
print("bundled cls1", out["reuse_stats"], approx(out["records"]), flush=True)
d = np.load(f"{G}/data4_500.npz")
out = ipd.APD_SsN_Class2(d["c"], d["r"], d["l"], o, o, float(d["mu"]), np.ones(250000), rng=ipd.MatlabRand(5489))
print("bundled cls2", out["reuse_stats"], approx(out["records"]), flush=True)
for N in (256, 1024):
    for cls in (1, 2):
        pr = problem(cls, N); one = np.ones(N)
        if cls == 1:
            out = ipd.APD_SsN_Class1(pr["c"], pr["r"], pr["l"], one, one, np.inf, rng=ipd.MatlabRand(5489))
        else:
            out = ipd.APD_SsN_Class2(pr["c"], pr["r"], pr["l"], one, one, pr["mu"], np.ones(N * N), rng=ipd.MatlabRand(5489))
        print("synthetic", N, "cls", cls, out["reuse_stats"], approx(out["records"]), "k", out["k"], flush=True)
