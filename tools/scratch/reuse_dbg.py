import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import codes_of_ipd_ssn_amg_method_amd as ipd
G = "tests/golden"; o = np.ones(500)
d = np.load(f"{G}/data1_500.npz")
out = ipd.APD_SsN_Class1(d["c"], d["r"], d["l"], o, o, np.inf, rng=ipd.MatlabRand(5489))
print(out["reuse_stats"])
