import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
import codes_of_ipd_ssn_amg_method_amd as ipd
from codes_of_ipd_ssn_amg_method_amd import _lib
from ctypes import byref, c_double, c_int, c_int32
m = n = 1024
for kind in ("tree", "hub"):
    s = bench.build_mask(m, n, kind, 1.0)
    Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
    for cyc in ("v", "w"):
        opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle=cyc, isnsp=1, inter=1, fnode=n)
        for nomid in ("0", "1"):
            os.environ["IPD_NO_MID"] = nomid
            os.environ["IPD_DEBUG_LEVELS"] = "1"
            h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
            mode, grid, tmo = c_int32(), c_int32(), c_int32()
            _lib.check(_lib.lib.ipd_amg_solve_mode(h.handle, byref(mode), byref(grid), byref(tmo)))
            db = _lib.DeviceBuffer.from_array(f); dx = _lib.DeviceBuffer.from_array(guess)
            ms, bpc = c_double(), c_double()
            for rep in range(2):
                _lib.check(_lib.lib.ipd_amg_bench_cycles(h.handle, db.ptr, dx.ptr, c_int(50), byref(ms), byref(bpc)))
            t0 = time.perf_counter()
            x, it, rel, relk, rhok = h.solve(f, guess)
            t1 = time.perf_counter() - t0
            print(kind, cyc, "nomid", nomid, "mode", mode.value, "levels", h.level_sizes(), "ms/cycle %.4f" % (ms.value / 50), "solve: it", it, "rel %.2e" % rel, "wall %.2f ms" % (1e3 * t1))
