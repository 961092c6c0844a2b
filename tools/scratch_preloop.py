import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ctypes import byref, c_double, c_int, c_int64
import codes_of_ipd_ssn_amg_method_amd as ipd
from codes_of_ipd_ssn_amg_method_amd import _lib
m = n = 1024
s = bench.build_mask(m, n, "bernoulli", 1.0)
Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="v", isnsp=1, inter=1, fnode=n)
h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
db = _lib.DeviceBuffer.from_array(f); dx = _lib.DeviceBuffer.from_array(guess)
st = (c_int64 * 10)(); ms = c_double()
for c in (1, 1, 5, 5, 20):
    _lib.check(_lib.lib.ipd_amg_bench_resident(h.handle, db.ptr, dx.ptr, c_int(c), byref(ms), st))
    ticks = int(st[3])
    print("cycles %d: kernel (events) %.1f us, loop (in-kernel, workgroup 0) %.1f us -> outside the loop %.1f us" % (c, 1e3 * ms.value, ticks / 100.0, 1e3 * ms.value - ticks / 100.0))
