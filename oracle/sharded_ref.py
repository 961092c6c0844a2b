"""CPU model of the DEVICE formulation of the cycle -- TEST INFRASTRUCTURE ONLY.

The HIP kernels do not apply the reference's explicit smoother matrices; they use
 (1) the fused Gauss-Seidel half sweeps on the bipartite level
     (Rk{1}*(r - A e) == forward F-then-C half sweeps, Rk{1}' == backward C-then-F),
 (2) xig = 1'(r - A e) = 1'r - (A1)'e  for the kernel-space correction,
 (3) row-block ownership with an all-gather after every launch when sharded.
This module restates exactly that formulation in NumPy so that CPU tests can check
it against the literal oracle (oracle/ipd_oracle.py, MG_Vcycle.m:14-41) and exercise
the world_size > 1 decomposition over torch.distributed/gloo.
`gather(vec, lo, hi)` must make rows [lo,hi) of `vec` consistent across ranks.
"""
import numpy as np

from . import ipd_oracle as O


def _slices(lo, hi, rank, G, min_rows=1):
    rows = hi - lo
    if G <= 1 or rows % G or rows < min_rows:
        return lo, hi, False
    cnt = rows // G
    return lo + rank * cnt, lo + (rank + 1) * cnt, True


class ShardedCycle:
    def __init__(self, h: O.Hierarchy, nf1: int, rank=0, G=1, gather=None, min_rows=1):
        self.h, self.nf1, self.rank, self.G = h, nf1, rank, G
        self.gather = gather or (lambda v, lo, hi: None)
        self.min_rows = min_rows
        self.dinv, self.Axi, self.xx = {}, {}, {}
        for k in range(1, h.J):
            A = h.Ack[k]
            d = A.diagonal()
            self.dinv[k] = 1.0 / d if (k == 1 and nf1 > 0) else 0.5 * (1.0 / d)
            self.Axi[k] = A @ np.ones(A.shape[0])
            self.xx[k] = float(self.Axi[k].sum())

    def _rows(self, lo, hi, fn, outs):
        r0, r1, sharded = _slices(lo, hi, self.rank, self.G, self.min_rows)
        fn(r0, r1)
        if sharded:
            for v in outs:
                self.gather(v, lo, hi)

    def _half(self, k, r, eold, w, enew, lo, hi, u0, u1, isnsp, ezero):
        A = self.h.Ack[k]
        c = 0.0
        if isnsp:
            c = (r.sum() - (0.0 if ezero else self.Axi[k] @ eold)) / self.xx[k]
        y = np.zeros_like(eold) if ezero else eold.copy()
        y[u0:u1] = w[u0:u1]

        def fn(r0, r1):
            s = A[r0:r1, :] @ y
            eo = 0.0 if ezero else eold[r0:r1]
            g = r[r0:r1] - s - (self.Axi[k][r0:r1] * c if isnsp else 0.0)
            wv = eo + self.dinv[k][r0:r1] * g
            w[r0:r1] = wv
            enew[r0:r1] = wv + c
        self._rows(lo, hi, fn, [enew, w])

    def sweep(self, k, r, e, isnsp, post, ezero):
        N = self.h.Ack[k].shape[0]
        enew, w = np.zeros(N), np.zeros(N)
        nf = self.nf1 if k == 1 else 0
        if nf == 0:
            self._half(k, r, e, w, enew, 0, N, 0, 0, isnsp, ezero)
        else:
            f0, f1 = (nf, N) if post else (0, nf)
            s0, s1 = (0, nf) if post else (nf, N)
            self._half(k, r, e, w, enew, f0, f1, 0, 0, isnsp, ezero)
            self._half(k, r, e, w, enew, s0, s1, f0, f1, isnsp, ezero)
        return enew

    def cycle(self, r, isnsp, wcycle=False, k=1, e=None):
        h = self.h
        if k == h.J:
            return O.PCG(h.Ack[k], r)[0]
        A = h.Ack[k]
        N = A.shape[0]
        ezero = e is None
        e = np.zeros(N) if e is None else e
        for _ in range(h.smoth_it):
            e = self.sweep(k, r, e, isnsp, False, ezero)
            ezero = False
        rr = np.zeros(N)

        def resid(r0, r1):
            rr[r0:r1] = r[r0:r1] - A[r0:r1, :] @ e
        self._rows(0, N, resid, [rr])
        P = h.Prok[k + 1]
        Pt = O._csr(P.T)
        rc = np.zeros(P.shape[1])

        def restrict(r0, r1):
            rc[r0:r1] = Pt[r0:r1, :] @ rr
        self._rows(0, P.shape[1], restrict, [rc])
        ec = self.cycle(rc, isnsp, wcycle, k + 1)
        if wcycle and k + 1 < h.J:
            ec = self.cycle(rc, isnsp, wcycle, k + 1, ec)

        def prolong(r0, r1):
            e[r0:r1] = e[r0:r1] + P[r0:r1, :] @ ec
        self._rows(0, N, prolong, [e])
        for _ in range(h.smoth_it):
            e = self.sweep(k, r, e, isnsp, True, False)
        return e
