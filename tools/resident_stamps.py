#!/usr/bin/env python3
"""In-kernel stamps of the level-resident solve kernel on the metric workload (bench.py's
system): where a cycle's time goes inside workgroup 0 -- waiting in hand-off sweeps against
everything else (row dot products, reductions, barriers, transfers, tail solve).
  python tools/resident_stamps.py [--n1 1024] [--cycle v] [--cycles 200]"""
import argparse
import os
import sys
from ctypes import byref, c_double, c_int, c_int64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=1024)
    ap.add_argument("--rho", type=float, default=1.0)
    ap.add_argument("--cycle", default="v")
    ap.add_argument("--cycles", type=int, default=200)
    ap.add_argument("--no-poly2", action="store_true", help="level 2 as sweeps (33 hand-offs per V cycle)")
    a = ap.parse_args()
    import codes_of_ipd_ssn_amg_method_amd as ipd
    from codes_of_ipd_ssn_amg_method_amd import _lib
    m = n = a.n1
    s = bench.build_mask(m, n, "bernoulli", a.rho)
    Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
    opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle=a.cycle, isnsp=1, inter=1, fnode=n)
    h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
    print("level 1 <-> 2 transfers from the bit mask:", h.attach_mask_transfers(np.ones(m), np.ones(n), bench.TK))
    if a.cycle == "v" and not a.no_poly2:
        print("level 2 composed over a visit (ipd_amg_attach_level2_poly):", h.attach_level2_poly())
    db = _lib.DeviceBuffer.from_array(f)
    dx = _lib.DeviceBuffer.from_array(guess)
    st = (c_int64 * 10)()
    ms = c_double()
    for rep in range(3):
        _lib.check(_lib.lib.ipd_amg_bench_resident(h.handle, db.ptr, dx.ptr, c_int(a.cycles), byref(ms), st))
        wait, tot, nh, ticks, bar1, store, bar2, xfer, tail = [int(v) for v in st][:9]
        clk_mhz = tot / (ticks / 100.0)
        print("cycles=%d  %.3f ms  -> %.2f us/cycle, %d hand-offs (%.1f per cycle, %.2f us each); "
              "workgroup 0: waiting in sweeps %.1f %% (%.2f us per hand-off), shader clock %.0f MHz"
              % (a.cycles, ms.value, 1e3 * ms.value / a.cycles, nh, nh / a.cycles,
                 1e3 * ms.value / nh, 100.0 * wait / tot, wait / clk_mhz / nh, clk_mhz))
        rest = tot - wait - bar1 - store - bar2 - xfer - tail
        print("   per cycle (us): transfers P'rr, P e_2 %.2f | tail level %.2f" % (
            xfer / clk_mhz / a.cycles, tail / clk_mhz / a.cycles))
        print("   per hand-off (us): row work %.2f | barrier before publish %.2f | sweep wait %.2f | "
              "store+sums %.2f | closing barrier %.2f" % tuple(v / clk_mhz / nh for v in (rest, bar1, wait, store, bar2)))


if __name__ == "__main__":
    main()
