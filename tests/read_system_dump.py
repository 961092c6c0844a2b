"""Reader for the IPD_DUMP_SYSTEM files (csrc/ipd_hybrid.hip dump_system): the rescaled Newton
system Ae*u = f of one Hybrid_AMG call of a driver run, so that it can be put through the oracle.

  python tests/read_system_dump.py FILE [v|w]     # oracle Class_AMG on the largest component
"""
import sys

import numpy as np
import scipy.sparse as sp


def read(path):
    with open(path, "rb") as fh:
        M, nf, nnz = np.fromfile(fh, np.int64, 3)
        rp = np.fromfile(fh, np.int32, M + 1)
        ci = np.fromfile(fh, np.int32, nnz)
        va = np.fromfile(fh, np.float64, nnz)
        f = np.fromfile(fh, np.float64, M)
    return sp.csr_matrix((va, ci, rp), shape=(M, M)), f, int(nf)


if __name__ == "__main__":
    import os
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ipd_oracle as O
    Ae, f, nf = read(sys.argv[1])
    cyc = sys.argv[2] if len(sys.argv) > 2 else "w"
    ncomp, lab = sp.csgraph.connected_components(Ae)
    pk = np.flatnonzero(lab == np.argmax(np.bincount(lab)))
    Ak = sp.csr_matrix(Ae[pk, :][:, pk])
    print("M", Ae.shape[0], "nnz", Ae.nnz, "components", ncomp, "largest", len(pk), "nnz", Ak.nnz)
    o = O.amg_options_class1(cyc)
    o.update(fnode=int((pk < nf).sum()), isnsp=1, retol=1e-11, maxit=30, smoth=5, bigph=1)
    t0 = time.time()
    h = O.amg_setup(Ak, o, O.matlab_rng())
    print("oracle setup", time.time() - t0, "s levels", h.level_sizes(), flush=True)
    fk = f[pk]
    x = np.zeros(len(pk))
    r0 = np.linalg.norm(fk)
    for it in range(4):
        r = fk - Ak @ x
        x = x + (O.MG_Wcycle(h, r, 1) if cyc == "w" else O.MG_Vcycle(h, r, 1))
        print("cycle", it + 1, "rel res", np.linalg.norm(fk - Ak @ x) / r0, flush=True)
