#!/usr/bin/env python3
"""Times the CPU restatement of the drivers (oracle/drivers.py, inner_solver = 4) on the synthetic
problems tools/bench_driver.py uses -- the "CPU oracle" column of DESIGN.md section 6.  Lives under
tests/ because only tests/ may import oracle/.

  python tests/time_oracle_drivers.py [--sizes 500] [--classes 1,2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import drivers as D   # noqa: E402


def problem(cls, N, seed=1):
    rs = np.random.RandomState(seed)
    c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
    if cls == 1:
        return dict(c=c, r=r, l=l * r.sum() / l.sum())
    return dict(c=c, r=r, l=l, mu=0.65 * min(r.sum(), l.sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="500")
    ap.add_argument("--classes", default="1,2")
    a = ap.parse_args()
    for N in [int(x) for x in a.sizes.split(",")]:
        for cls in [int(x) for x in a.classes.split(",")]:
            pr = problem(cls, N)
            one = np.ones(N)
            t0 = time.perf_counter()
            if cls == 1:
                st = D.warmup_class1(pr["c"], pr["r"], pr["l"], one, one, np.inf, 100)
                t1 = time.perf_counter()
                ref = D.apd_ssn_class1(pr["c"], pr["r"], pr["l"], one, one, np.inf, inner="amg", start=st)
            else:
                st = D.warmup_class2(pr["c"], pr["r"], pr["l"], one, one, pr["mu"], np.ones(N * N), 100)
                t1 = time.perf_counter()
                ref = D.apd_ssn_class2(pr["c"], pr["r"], pr["l"], one, one, pr["mu"], np.ones(N * N),
                                       inner="amg", start=st)
            t2 = time.perf_counter()
            print(json.dumps(dict(cls=cls, N=N, oracle_warmup_s=t1 - t0, oracle_apd_s=t2 - t1,
                                  oracle_k=ref["k"], oracle_fval=ref["fval"])), flush=True)


if __name__ == "__main__":
    main()
