"""ctypes loader of oracle/cpu_cycle.c (TEST INFRASTRUCTURE / CPU BASELINE, see that file):
feeds it the hierarchy the Python oracle built and runs the Class_AMG loop body on host cores."""
import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libipd_cpu_cycle.so")


def build():
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)


def _load():
    # idle OpenMP threads sleep instead of spinning (the box's CPU share is smaller than the machine)
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "cpu_cycle.c")):
        build()
    lib = ctypes.CDLL(SO)
    lib.ipdo_cycles.restype = ctypes.c_double
    return lib


class CpuCycle:
    """h: oracle.ipd_oracle.Hierarchy (1-based lists Ack, Prok, Rk)."""

    def __init__(self, h, isnsp):
        self.lib = _load()
        self.keep = []
        self.N = h.Ack[1].shape[0]
        self.lib.ipdo_reset(ctypes.c_int(h.J), ctypes.c_int(int(h.smoth_it)), ctypes.c_int(int(isnsp)))
        for k in range(1, h.J + 1):
            self._set(k, 0, h.Ack[k])
            if k < h.J:
                self._set(k, 1, h.Rk[k])
                self._set(k, 2, sp.csr_matrix(sp.csr_matrix(h.Rk[k]).T))
            if k >= 2:
                self._set(k, 3, h.Prok[k])
                self._set(k, 4, sp.csr_matrix(sp.csr_matrix(h.Prok[k]).T))

    def _set(self, k, which, M):
        M = sp.csr_matrix(M)
        M.sort_indices()
        rp = np.ascontiguousarray(M.indptr, np.int32)
        ci = np.ascontiguousarray(M.indices, np.int32)
        va = np.ascontiguousarray(M.data, np.float64)
        self.keep += [rp, ci, va]
        self.lib.ipdo_set(ctypes.c_int(k), ctypes.c_int(which), ctypes.c_int(M.shape[0]), ctypes.c_int(M.shape[1]),
                          rp.ctypes.data_as(ctypes.c_void_p), ci.ctypes.data_as(ctypes.c_void_p),
                          va.ctypes.data_as(ctypes.c_void_p))

    def max_threads(self):
        return int(self.lib.ipdo_max_threads())

    def run(self, b, x0, cycles, wcycle=False, threads=1):
        """-> x, seconds, residual norms (cycles + 1)."""
        b = np.ascontiguousarray(b, np.float64)
        x = np.array(x0, np.float64, copy=True)
        res = np.zeros(cycles + 1)
        sec = self.lib.ipdo_cycles(b.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_int(int(cycles)), ctypes.c_int(1 if wcycle else 0),
                                   ctypes.c_int(int(threads)), res.ctypes.data_as(ctypes.c_void_p))
        return x, float(sec), res
