import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import codes_of_ipd_ssn_amg_method_amd as ipd
from codes_of_ipd_ssn_amg_method_amd import _lib
from ctypes import byref, c_double, c_int
m = n = 1024
s = bench.build_mask(m, n, "bernoulli", 1.0)
Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="v", isnsp=1, inter=1, fnode=n)
h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
if "--xmask" in sys.argv:
    print("xmask", h.attach_mask_transfers(np.ones(m), np.ones(n), bench.TK))
db = _lib.DeviceBuffer.from_array(f); dx = _lib.DeviceBuffer.from_array(guess)
ctx = _lib.get_ctx()
def run(c):
    ms, bpc = c_double(), c_double()
    _lib.check(_lib.lib.ipd_amg_bench_cycles(h.handle, db.ptr, dx.ptr, c_int(c), byref(ms), byref(bpc)))
    return ms.value
run(5)
for c in (1, 2, 5, 10, 20, 50, 200):
    ev = []; wall = []
    for rep in range(5):
        ctx.sync(); t0 = time.perf_counter(); e = run(c); ctx.sync(); wall.append(time.perf_counter() - t0); ev.append(e)
    print("cycles %4d  events %8.1f us (%.2f us/cycle)   wall %8.1f us (%.2f us/cycle)" % (c, 1e3 * min(ev), 1e3 * min(ev) / c, 1e6 * min(wall), 1e6 * min(wall) / c))
