import os, sys, time, glob, subprocess
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp
os.makedirs("/tmp/dump", exist_ok=True)
if not glob.glob("/tmp/dump/s*.bin"):
    env = dict(os.environ, IPD_DUMP_SYSTEM="/tmp/dump/s", IPD_DUMP_CALLS="60-200")
    subprocess.run([sys.executable, "tools/bench_driver.py", "--sizes", "1024", "--classes", "1"], env=env, capture_output=True, text=True)
from tests.read_system_dump import read
import codes_of_ipd_ssn_amg_method_amd as ipd
from codes_of_ipd_ssn_amg_method_amd import _lib
from ctypes import byref, c_double, c_int, c_int32
files = sorted(glob.glob("/tmp/dump/s*.bin"), key=lambda p: int(os.path.basename(p)[1:-4]))
done = 0
for path in files[::10]:
    Ae, f, nf = read(path)
    ncomp, lab = sp.csgraph.connected_components(Ae)
    if ncomp != 1:
        continue
    opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="w", isnsp=1, inter=1, fnode=nf)
    res = {}
    for nomid in ("0", "1"):
        os.environ["IPD_NO_MID"] = nomid
        h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
        mode = c_int32(); _lib.check(_lib.lib.ipd_amg_solve_mode(h.handle, byref(mode), None, None))
        g = np.zeros(Ae.shape[0])
        h.solve(f, g)
        t0 = time.perf_counter()
        for _ in range(5):
            x, it, rel, relk, rhok = h.solve(f, g)
        t = (time.perf_counter() - t0) / 5
        db = _lib.DeviceBuffer.from_array(f); dx = _lib.DeviceBuffer.from_array(g)
        ms, bpc = c_double(), c_double()
        _lib.check(_lib.lib.ipd_amg_bench_cycles(h.handle, db.ptr, dx.ptr, c_int(20), byref(ms), byref(bpc)))
        res[nomid] = (mode.value, it, rel, 1e3 * t, ms.value / 20)
        lv = [(h.level_dims(k)) for k in range(1, h.J + 1)]
    print(os.path.basename(path), lv)
    for k, v in res.items():
        print("   NO_MID=%s mode %d its %d rel %.1e solve wall %.3f ms; bench %.4f ms/cycle" % ((k,) + v))
    done += 1
    if done >= 6:
        break
