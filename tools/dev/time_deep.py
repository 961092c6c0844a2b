import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from ctypes import byref, c_double, c_int
import codes_of_ipd_ssn_amg_method_amd as ipd
from codes_of_ipd_ssn_amg_method_amd import _lib
import bench
from tests.test_gpu_bench_workload import env, options, solve_mode, resident_kernel_name
N = 2048
ks = [int(a) for a in sys.argv[1].split(",")]
variants = [("default", {}), ("G 230", {"IPD_RESIDENT_G": "230"}), ("G 255", {"IPD_RESIDENT_G": "255"}), ("no poly4", {"IPD_NO_RES_POLY4": "1"}), ("v-cycle", {"CYCLE": "v"}), ("skip1 (poly streams)", {"IPD_DEBUG_SKIP": "1"}), ("skip2 (pcg)", {"IPD_DEBUG_SKIP": "2"}),
            ("skip15 (all tail work)", {"IPD_DEBUG_SKIP": "15"}), ("launches", {"IPD_NO_RESIDENT_DEEP": "1"})]
for k in ks:
    Ae, f, guess, nf, s, bk1, tk = bench.capture_newton_system(ipd, N, k)
    whole = Ae.shape[0] == 2 * N
    for name, kv in variants:
        kv = dict(kv)
        cyc = kv.pop("CYCLE", "w")
        with env(**kv):
            h = ipd.AMGHierarchy(Ae, options(cyc, nf), ipd.MatlabRand(5489))
            if whole:
                h.attach_mask_operator(np.ones(N), np.ones(N), tk)
            db = _lib.DeviceBuffer.from_array(f); dx = _lib.DeviceBuffer.from_array(guess)
            ms, bpc = c_double(), c_double()
            _lib.check(_lib.lib.ipd_amg_bench_cycles(h.handle, db.ptr, dx.ptr, c_int(5), byref(ms), byref(bpc)))
            dx = _lib.DeviceBuffer.from_array(guess)
            _lib.check(_lib.lib.ipd_amg_bench_cycles(h.handle, db.ptr, dx.ptr, c_int(20), byref(ms), byref(bpc)))
        print(k, h.level_sizes(), name, resident_kernel_name(h), "grid", solve_mode(h)[1], "%.4f ms/cycle" % (ms.value / 20))
        h.close()
