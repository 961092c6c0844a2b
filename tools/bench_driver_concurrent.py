#!/usr/bin/env python3
"""Throughput of T independent Class 1 solves running concurrently on ONE GPU: each host thread
owns an ipd context (its own HIP stream and arenas) and one problem.  A single solve is a chain of
latency-bound launches that leaves the device mostly idle, so independent problems overlap.

  python tools/bench_driver_concurrent.py [--n 1024] [--threads 1,2,4,8]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import codes_of_ipd_ssn_amg_method_amd as ipd           # noqa: E402
from codes_of_ipd_ssn_amg_method_amd import _lib as L   # noqa: E402

AMG = dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5, cycle="w", isnsp=1, inter=1)


def problem(N, seed):
    rs = np.random.RandomState(seed)
    c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
    return c, r, l * r.sum() / l.sum()


def worker(N, seed, out, idx, barrier):
    ctx = L.Context(int(os.environ.get("IPD_DEVICE", "0")))
    c, r, l = problem(N, seed)
    one = np.ones(N)
    ws = ipd.APDWorkspace(1, c, r, l, one, one, gama=np.inf, ctx=ctx)
    barrier.wait()
    t0 = time.perf_counter()
    ws.warmup(0.0, 100)
    res = ws.run(AMG, ipd.MatlabRand(5489))
    out[idx] = (time.perf_counter() - t0, res["k"], res["converged"], res["fval"])
    ws.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--threads", default="1,2,4,8")
    a = ap.parse_args()
    for T in [int(x) for x in a.threads.split(",")]:
        out = [None] * T
        barrier = threading.Barrier(T + 1)
        th = [threading.Thread(target=worker, args=(a.n, 1 + i, out, i, barrier)) for i in range(T)]
        for t in th:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        print(json.dumps(dict(N=a.n, threads=T, wall_s=wall, solves_per_s=T / wall,
                              per_solve_s=[round(o[0], 3) for o in out],
                              all_converged=all(o[2] for o in out), k=[o[1] for o in out])), flush=True)


if __name__ == "__main__":
    main()
