#!/usr/bin/env python3
"""One ASAt + Hybrid_AMG call (after a warm-up call) for rocprofv3 kernel traces:
  tests/one_newton.py [tree|hub|golden40]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import codes_of_ipd_ssn_amg_method_amd as ipd  # noqa: E402
from tests import problems as PR  # noqa: E402
from tests.test_golden_oracle import load, problem_from  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "tree"
if kind == "golden40":
    pd = problem_from(load("class1_500_k40.npz"))
else:
    s = PR.mask_hub(1024, 1024, seed=2) if kind == "hub" else PR.mask_tree(1024, 1024, seed=2)
    pd = PR.make_prob(1024, 1024, s)
opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="w", isnsp=1, inter=1,
            guess=None, fnode=None)
for rep in range(2):
    t0 = time.perf_counter()
    pd["H0"] = ipd.ASAt(pd["s"], pd["p"], pd["q"])
    t1 = time.perf_counter()
    z, it, res, info = ipd.Hybrid_AMG(pd, opts, ipd.MatlabRand())
    t2 = time.perf_counter()
    print("rep %d: ASAt %.3f ms, Hybrid_AMG %.3f ms, its %d" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), it))
