#!/usr/bin/env python3
"""Times the single-workgroup sub-cycle kernel on a realistic hierarchy (tree-like mask)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import codes_of_ipd_ssn_amg_method_amd as ipd          # noqa: E402
from codes_of_ipd_ssn_amg_method_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kcap = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# a realistic Newton system: run the device driver kcap iterations and take the next active set
rs = np.random.RandomState(1)
c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
l = l * r.sum() / l.sum()
one = np.ones(N)
ws = ipd.APDWorkspace(1, c, r, l, one, one, gama=np.inf)
ws.warmup(0.0, 100)
amg = dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5, cycle="w", isnsp=1, inter=1)
ws.run(amg, ipd.MatlabRand(5489), iters=kcap)
lam = ws.state()[2]
sc = ws.begin(kcap + 1)
ev = ws.eval(lam)
print("captured k=%d E=%d bk1=%.3e tk=%.3e" % (kcap + 1, ev["E"], sc["bk1"], sc["tk"]))
import scipy.sparse as sp
H0 = ipd.ASAt(ev["s"], one, one)                       # Hybrid_AMG.m:17-24 with p = q = 1, T = 0
Q0 = sp.diags(np.concatenate([one, -one]))
Ae = sp.csr_matrix(sc["bk1"] * (Q0 @ Q0) + (1.0 / sc["tk"]) * ((Q0 @ H0) @ Q0))
ws.close()
opts = dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5, cycle="w", isnsp=1, inter=1,
            fnode=N)
h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489))
print("levels", [h.level_dims(k) for k in range(1, h.J + 1)])
ms = ctypes.c_double()
ks = ctypes.c_int32()
st = (ctypes.c_int64 * 8)()
L.check(L.lib.ipd_amg_bench_subcycle(h.handle, 200, ctypes.byref(ms), ctypes.byref(ks), st))
print("k_sub", ks.value, "us per launch", 1e3 * ms.value / 200,
      "shader MHz", st[0], "stages us: rootcopy %.2f cycle %.2f" % (
                                                         (st[2] - st[1]) / 100.0,
                                                         (st[3] - st[2]) / 100.0),
      "| tiny %.2f blk-sweeps %.2f resid+restrict %.2f prolong %.2f" % tuple(st[i] / 100.0 for i in (4, 5, 6, 7)))
