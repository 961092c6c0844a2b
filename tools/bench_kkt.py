#!/usr/bin/env python3
"""ASAt / Ax / Aty on device-resident inputs: time per call and fraction of the 8 TB/s HBM peak
against the algorithmic bytes of DESIGN.md section 3 (ASAt: mn + 8M + 16(2E+M); Ax, Aty: 8mn + 16M).
Wall clock around back-to-back calls of the C ABI's *_dev entry points (ASAt includes its one host
read-back of the entry count); run under `rocprofv3 --kernel-trace --stats` for per-kernel rows.
  python tools/bench_kkt.py [--n1 1024] [--reps 100]"""
import argparse
import os
import sys
import time
from ctypes import byref, c_int64, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=100)
    a = ap.parse_args()
    from codes_of_ipd_ssn_amg_method_amd import _lib
    from tests import problems as PR
    lib, ctx = _lib.lib, _lib.get_ctx()
    m = n = a.n1
    M, mn = m + n, m * n
    p = _lib.DeviceBuffer.from_array(np.ones(m))
    q = _lib.DeviceBuffer.from_array(np.ones(n))
    rs = np.random.RandomState(0)
    x = _lib.DeviceBuffer.from_array(rs.randn(mn))
    y = _lib.DeviceBuffer.from_array(rs.randn(M))
    z = _lib.DeviceBuffer(8 * mn)
    yo = _lib.DeviceBuffer(8 * M)
    peak = 8000.0

    def timeit(fn, reps):
        for _ in range(3):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps

    t = timeit(lambda: _lib.check(lib.ipd_ax_dev(ctx.handle, x.ptr, p.ptr, q.ptr, c_int64(m), c_int64(n), yo.ptr)), a.reps)
    b = 8.0 * mn + 16.0 * M
    print("Ax    m=n=%d: %7.2f us  %6.0f GB/s  frac %.3f" % (m, 1e6 * t, b / t / 1e9, b / t / 1e9 / peak))
    t = timeit(lambda: _lib.check(lib.ipd_aty_dev(ctx.handle, y.ptr, p.ptr, q.ptr, c_int64(m), c_int64(n), z.ptr)), a.reps)
    print("Aty   m=n=%d: %7.2f us  %6.0f GB/s  frac %.3f" % (m, 1e6 * t, b / t / 1e9, b / t / 1e9 / peak))
    masks = [("rho=1", PR.mask_bernoulli(m, n, 1.0)), ("rho=1/8", PR.mask_bernoulli(m, n, 0.125)),
             ("rho=1/64", PR.mask_bernoulli(m, n, 1.0 / 64)), ("tree", PR.mask_tree(m, n, seed=2))]
    for name, s in masks:
        ds = _lib.DeviceBuffer.from_array(s)
        E = int(s.sum())

        def one():
            H = c_void_p()
            _lib.check(lib.ipd_asat_dev(ctx.handle, ds.ptr, p.ptr, q.ptr, c_int64(m), c_int64(n), byref(H)))
            lib.ipd_dmat_destroy(H)

        t = timeit(one, a.reps)
        b = mn + 8.0 * M + 16.0 * (2 * E + M)
        print("ASAt  m=n=%d %-8s E=%8d: %7.2f us  %6.0f GB/s  frac %.4f" % (m, name, E, 1e6 * t, b / t / 1e9, b / t / 1e9 / peak))


if __name__ == "__main__":
    main()
