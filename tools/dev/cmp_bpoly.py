import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import codes_of_ipd_ssn_amg_method_amd as ipd
import bench
from tests.test_gpu_bench_workload import env, options, solve_mode
os.environ["IPD_NO_RESIDENT_DEEP"] = "1"
for k in [int(a) for a in sys.argv[1:]] or [8, 12, 16, 20, 24, 28, 32, 36, 40, 44]:
    Ae, f, guess, nf, s, bk1, tk = bench.capture_newton_system(ipd, 2048, k)
    res = {}
    for tag, kv in (("bpoly", {}), ("nobpoly", {"IPD_NO_BPOLY": "1"})):
        with env(**kv):
            h = ipd.AMGHierarchy(Ae, options("w", nf), ipd.MatlabRand(5489))
        x, it, rr, relk, rhok = h.solve(f, guess)
        res[tag] = (it, rr, np.asarray(relk)[:8], h.level_forms(), h.level_sizes())
        h.close()
    a, b = res["bpoly"], res["nobpoly"]
    print(k, a[4], "forms", a[3], "its", a[0], b[0], "rel", "%.2e %.2e" % (a[1], b[1]))
    if a[0] != b[0] or abs(a[1] - b[1]) > 1e-9:
        print("   DIFF bpoly  ", a[2]); print("        nobpoly", b[2])
