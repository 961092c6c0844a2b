#!/usr/bin/env python3
"""max |rel_resk(device) - rel_resk(oracle)| on the three captured M = 2048 Newton systems (scratch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import codes_of_ipd_ssn_amg_method_amd as ipd
from oracle import ipd_oracle as O
from tests.test_gpu_bench_workload import options
import tests.test_gpu_resident_remote as T

class Req: pass
for kcap in (9, 20, 30):
    rq = Req(); rq.param = kcap
    Ae, f, n, want = T.newton_system.__wrapped__(ipd, rq) if hasattr(T.newton_system, "__wrapped__") else T.newton_system.__pytest_wrapped__.obj(ipd, rq)
    x0 = np.zeros(Ae.shape[0])
    for cycle in ("v", "w"):
        opts = options(cycle, n)
        o = dict(opts); o.update(guess=x0)
        xo, ito, rro, relko, rhoko = O.Class_AMG(Ae, f, o, O.matlab_rng(5489))
        sgn = np.where(np.random.RandomState(11).random_sample(f.size) < 0.5, -1.0, 1.0)
        _, itp, _, relkp, _ = O.Class_AMG(Ae, f * (1.0 + 2.2e-16 * sgn), o, O.matlab_rng(5489))
        kk = min(ito, itp) + 1
        dd = np.abs(np.asarray(relko[:kk]) - np.asarray(relkp[:kk]))
        print("k%d %s ORACLE vs ORACLE(f perturbed by 1 ulp): it %d/%d max|d| %.3e at %d" % (kcap + 1, cycle, ito, itp, dd.max(), int(dd.argmax())))
        for tag, kv in (("default", {}), ("NO_POLY", {"IPD_NO_POLY": "1"}), ("NO_RESIDENT", {"IPD_NO_RESIDENT": "1"}),
                        ("NO_RESIDENT+NO_POLY", {"IPD_NO_RESIDENT": "1", "IPD_NO_POLY": "1"})):
            os.environ.update(kv)
            h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489))
            x, it, rr, relk, rhok = h.solve(f, x0)
            for k_ in kv: os.environ.pop(k_)
            k = min(it, ito) + 1
            d = np.abs(np.asarray(relk[:k]) - np.asarray(relko[:k]))
            print("k%d %s %-20s it %d/%d max|d| %.3e at %d  (rel there %.3e)" % (kcap + 1, cycle, tag, it, ito, d.max(), int(d.argmax()), relko[int(d.argmax())]))
            h.close()
