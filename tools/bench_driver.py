#!/usr/bin/env python3
"""Whole-solve timing of the device drivers (rows f1/f2) and the bandwidth of the fused
evaluation pass.  (The CPU restatement is timed by tests/time_oracle_drivers.py: only tests/ may
import oracle/.)

  python tools/bench_driver.py [--sizes 500,1024] [--classes 1,2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import codes_of_ipd_ssn_amg_method_amd as ipd   # noqa: E402


def problem(cls, N, seed=1):
    rs = np.random.RandomState(seed)
    c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
    if cls == 1:
        return dict(c=c, r=r, l=l * r.sum() / l.sum())
    return dict(c=c, r=r, l=l, mu=0.65 * min(r.sum(), l.sum()))


def run(cls, N, pr, cycle="w"):
    one = np.ones(N)
    t0 = time.perf_counter()
    if cls == 1:
        ws = ipd.APDWorkspace(1, pr["c"], pr["r"], pr["l"], one, one, gama=np.inf)
        amg = dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5, cycle=cycle, isnsp=1, inter=1)
    else:
        ws = ipd.APDWorkspace(2, pr["c"], pr["r"], pr["l"], one, one, mu=pr["mu"], phi=np.ones(N * N))
        amg = dict(retol=1e-11, bigph=1, maxit=40, theta=1 / 4, smoth=10, cycle=cycle, isnsp=1, inter=1)
    ipd.get_ctx().sync()
    t1 = time.perf_counter()
    ws.warmup(0.0, 100)
    t2 = time.perf_counter()
    out = ws.run(amg, ipd.MatlabRand(5489))
    t3 = time.perf_counter()
    lls = np.array([r["ll"] for r in ws.records()] or [0])
    ll_stats = dict(mean=float(lls.mean()), p50=float(np.median(lls)), max=int(lls.max()),
                    zero_frac=float((lls == 0).mean()))
    prof = None
    if os.environ.get("IPD_PROFILE"):
        import ctypes
        sec = (ctypes.c_double * 16)()
        cnt = (ctypes.c_int64 * 16)()
        ns = ipd._lib.lib.ipd_prof_read(sec, cnt, 1)
        names = ["asat", "build_Ae", "components", "amg_setup", "amg_solve", "small_blocks",
                 "eval", "begin_end"]
        prof = {names[i]: [round(sec[i], 4), int(cnt[i])] for i in range(ns)}
    ws.begin(8)
    ms, by = ws.bench_eval(200)
    rec = dict(cls=cls, N=N, upload_s=t1 - t0, warmup_s=t2 - t1, apd_s=t3 - t2, k=out["k"],
               converged=out["converged"], fval=out["fval"], newton_steps=out["nrec"],
               SumAMG=out["SumAMG"], eval_us=1e3 * ms / 200, eval_GBps=by * 200 / (ms * 1e-3) / 1e9,
               eval_bytes=by, ll=ll_stats)
    if prof:
        rec["profile_s_calls"] = prof
    ws.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="500,1024")
    ap.add_argument("--classes", default="1,2")
    ap.add_argument("--cycle", default="w", choices=["v", "w"],
                    help="AMG cycle of the inner solver (the reference scripts use w)")
    a = ap.parse_args()
    for N in [int(x) for x in a.sizes.split(",")]:
        for cls in [int(x) for x in a.classes.split(",")]:
            pr = problem(cls, N)
            rec = run(cls, N, pr, a.cycle)
            rec["cycle"] = a.cycle
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
