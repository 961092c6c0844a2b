#!/usr/bin/env python3
"""Level-resident kernel with a remote tail (csrc/ipd_resident.h, ResDesc::remote) against the
multi-launch path on Newton systems dumped from the m=n=1024 Class 1 driver run: ms per W cycle of
the fixed hierarchy and wall time of a whole solve.
  python tools/bench_remote_tail.py [first-last]"""
import glob
import os
import subprocess
import sys
import time
from ctypes import byref, c_double, c_int, c_int32, c_int64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np            # noqa: E402
import scipy.sparse as sp     # noqa: E402

rng_calls = sys.argv[1] if len(sys.argv) > 1 else "60-200"
os.makedirs("/tmp/dump", exist_ok=True)
if not glob.glob("/tmp/dump/s*.bin"):
    env = dict(os.environ, IPD_DUMP_SYSTEM="/tmp/dump/s", IPD_DUMP_CALLS=rng_calls)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_driver.py"), "--sizes", "1024", "--classes", "1"],
                   env=env, capture_output=True, text=True)
from tests.read_system_dump import read           # noqa: E402
import codes_of_ipd_ssn_amg_method_amd as ipd     # noqa: E402
from codes_of_ipd_ssn_amg_method_amd import _lib  # noqa: E402

files = sorted(glob.glob("/tmp/dump/s*.bin"), key=lambda p: int(os.path.basename(p)[1:-4]))
done = 0
for path in files[::int(os.environ.get("STRIDE", "7"))]:
    Ae, f, nf = read(path)
    ncomp, lab = sp.csgraph.connected_components(Ae)
    if ncomp != 1:
        continue
    opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="w", isnsp=1, inter=1, fnode=nf)
    res = {}
    for off in ("0", "1"):
        os.environ["IPD_NO_RESIDENT_REMOTE"] = off
        h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
        mode, grid = c_int32(), c_int32()
        _lib.check(_lib.lib.ipd_amg_solve_mode(h.handle, byref(mode), byref(grid), None))
        g = np.zeros(Ae.shape[0])
        h.solve(f, g)
        t0 = time.perf_counter()
        for _ in range(5):
            x, it, rel, relk, rhok = h.solve(f, g)
        t = (time.perf_counter() - t0) / 5
        db = _lib.DeviceBuffer.from_array(f)
        dx = _lib.DeviceBuffer.from_array(g)
        ms, bpc = c_double(), c_double()
        _lib.check(_lib.lib.ipd_amg_bench_cycles(h.handle, db.ptr, dx.ptr, c_int(20), byref(ms), byref(bpc)))
        extra = ""
        if mode.value == 2:
            st = (c_int64 * 10)()
            m2 = c_double()
            _lib.check(_lib.lib.ipd_amg_bench_resident(h.handle, db.ptr, dx.ptr, c_int(20), byref(m2), st))
            wait, tot, nh, ticks, bar1, store, bar2, xfer, tail = [int(v) for v in st][:9]
            busy = int(st[9])
            mhz = tot / (ticks / 100.0)
            extra = " | per cycle: %.0f hand-offs, wait %.1f us, tail (incl. waiting for it) %.1f us of which the tail workgroup is busy %.1f us, total %.1f us" % (
                nh / 20, wait / mhz / 20, tail / mhz / 20, busy / mhz / 20, tot / mhz / 20)
        res[off] = (mode.value, grid.value, it, rel, 1e3 * t, ms.value / 20, extra)
        lv = [(h.level_dims(k)) for k in range(1, h.J + 1)]
        h.close()
    print(os.path.basename(path), lv)
    for k, v in res.items():
        print("   NO_RESIDENT_REMOTE=%s mode %d grid %d its %d rel %.1e solve wall %.3f ms; %.4f ms/cycle%s" % ((k,) + v))
    done += 1
    if done >= int(os.environ.get("COUNT", "8")):
        break
