#!/usr/bin/env python3
"""Row f3: what sharing the hierarchy between AMG4POT's two right-hand sides changes
(Class2/AMG4POT.m:46-47).  Three modes on the reference's bundled Class 2 problem (data4-500) and
the synthetic m=n=1024 one:
  default            the second setup shares the rand-independent levels 1-2 of the first (donor):
                     bit-identical to two full setups
  IPD_NO_DONOR=1     two full setups (the reference's way)
  IPD_REUSE_HIERARCHY=1   ONE hierarchy for both right-hand sides: the second solve uses levels >= 3
                     built with other random numbers than the reference's second setup would draw
Reports iterations, Newton steps, total AMG cycles, objective, and the largest deviation of the
final iterate from the default run.   python tools/bench_reuse.py"""
import os
import subprocess
import sys
import json
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def one(which):
    import codes_of_ipd_ssn_amg_method_amd as ipd
    if which == "bundled":
        d = np.load(os.path.join(ROOT, "tests", "golden", "data4_500.npz"))
        N = int(d["r"].size)
        c, r, l, mu = d["c"].ravel(), d["r"].ravel(), d["l"].ravel(), float(d["mu"])
    else:
        N = 1024
        rs = np.random.RandomState(1)
        c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
        mu = 0.65 * min(r.sum(), l.sum())
    one_ = np.ones(N)
    ws = ipd.APDWorkspace(2, c, r, l, one_, one_, mu=mu, phi=np.ones(N * N))
    amg = dict(retol=1e-11, bigph=1, maxit=40, theta=1 / 4, smoth=10, cycle="w", isnsp=1, inter=1)
    ws.warmup(0.0, 100)
    ipd.get_ctx().sync()
    t0 = time.perf_counter()
    out = ws.run(amg, ipd.MatlabRand(5489))
    t = time.perf_counter() - t0
    u, v, lam, bk = ws.state()
    np.save("/tmp/reuse_%s_%s.npy" % (which, os.environ.get("MODE", "default")), u)
    print(json.dumps(dict(problem=which, mode=os.environ.get("MODE", "default"), k=out["k"], converged=bool(out["converged"]),
                          newton_steps=out["nrec"], SumAMG=out["SumAMG"], fval=out["fval"], seconds=round(t, 4))))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
        sys.exit(0)
    for which in ("bundled", "synthetic1024"):
        for mode, env in (("default", {}), ("two_full_setups", {"IPD_NO_DONOR": "1"}),
                          ("one_hierarchy", {"IPD_REUSE_HIERARCHY": "1"})):
            e = dict(os.environ, MODE=mode, **env)
            subprocess.run([sys.executable, os.path.abspath(__file__), which], env=e, check=True)
        a = np.load("/tmp/reuse_%s_default.npy" % which)
        for mode in ("two_full_setups", "one_hierarchy"):
            b = np.load("/tmp/reuse_%s_%s.npy" % (which, mode))
            print("   %s: max |x - x_default| = %.3e (|x|_max %.3e)" % (mode, np.max(np.abs(a - b)), np.max(np.abs(a))))
