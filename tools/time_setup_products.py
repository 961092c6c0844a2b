"""Setup time with the Galerkin products through the row kernel vs the dense-tile kernel
(csrc/ipd_sparse.hip), on bench.py's Newton system with a dense mask (BASELINE regime D).
  python tools/time_setup_products.py [N ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

import codes_of_ipd_ssn_amg_method_amd as ipd

sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048]
for N in sizes:
    s = bench.build_mask(N, N, "bernoulli", 1.0)
    Ae = bench.build_newton_system(ipd, N, N, s)[0]
    o = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="v", isnsp=1, fnode=N)
    for gmin in ("rows", None):
        if gmin is None:
            os.environ.pop("IPD_PRODUCT", None)
        else:
            os.environ["IPD_PRODUCT"] = gmin
        ipd.AMGHierarchy(Ae, o, ipd.MatlabRand()).close()
        t0 = time.perf_counter()
        for _ in range(3):
            ipd.AMGHierarchy(Ae, o, ipd.MatlabRand()).close()
        print(f"m=n={N} products={'rows' if gmin else 'auto'} setup "
              f"{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
