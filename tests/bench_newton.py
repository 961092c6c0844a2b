#!/usr/bin/env python3
"""End-to-end time of one semismooth-Newton linear solve through the MATLAB-facing
(host-pointer) entry points: ASAt + Hybrid_AMG on the realistic golden fixtures
(m=n=500, captured from the restated Class 1 driver) and on synthetic m=n=1024 masks.
Prints milliseconds per call; with --oracle also the CPU oracle's time."""
import argparse
import glob
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import codes_of_ipd_ssn_amg_method_amd as ipd
    from tests import problems as PR
    from tests.test_golden_oracle import GOLD, load, problem_from
    cases = []
    for f in sorted(glob.glob(os.path.join(GOLD, "class1_500_k*.npz"))):
        g = load(os.path.basename(f))
        cases.append(("golden500 k=%d E=%d" % (int(g["k"]), int(g["E"])), problem_from(g)))
    for name, s in [("tree1024", PR.mask_tree(1024, 1024, seed=2)), ("hub1024", PR.mask_hub(1024, 1024, seed=2))]:
        cases.append((name + " E=%d" % int(s.sum()), PR.make_prob(1024, 1024, s)))
    opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="w", isnsp=1, inter=1,
                guess=None, fnode=None)
    for name, pd in cases:
        ipd.ASAt(pd["s"], pd["p"], pd["q"])  # warm
        t0 = time.perf_counter()
        for _ in range(args.reps):
            H0 = ipd.ASAt(pd["s"], pd["p"], pd["q"])
        t_asat = (time.perf_counter() - t0) / args.reps
        pd["H0"] = H0
        ipd.Hybrid_AMG(pd, opts, ipd.MatlabRand())
        t0 = time.perf_counter()
        for _ in range(args.reps):
            z, it, res, info = ipd.Hybrid_AMG(pd, opts, ipd.MatlabRand())
        t_h = (time.perf_counter() - t0) / args.reps
        line = "%-28s ASAt %7.3f ms   Hybrid_AMG %8.3f ms  (its %d, comps %d, res %.1e)" % (
            name, 1e3 * t_asat, 1e3 * t_h, it, info[0], res)
        if args.oracle:
            from oracle import ipd_oracle as O
            t0 = time.perf_counter()
            Ho = O.ASAt(pd["s"], pd["p"], pd["q"])
            to_a = time.perf_counter() - t0
            pd2 = dict(pd)
            pd2["H0"] = Ho
            t0 = time.perf_counter()
            O.Hybrid_AMG(pd2, opts, O.matlab_rng())
            to_h = time.perf_counter() - t0
            line += "   | oracle: ASAt %7.1f ms  Hybrid %8.1f ms" % (1e3 * to_a, 1e3 * to_h)
        print(line)


if __name__ == "__main__":
    main()
