"""CPU restatement of the reference's Class 1 driver -- TEST INFRASTRUCTURE ONLY.

`Class1/warmup_class1.m` (A-ADMM warm start) and the APD / semismooth-Newton loops of
`Class1/APD_SsN_Class1.m:101-275`, used to produce REALISTIC (s, bk1, tk, z) tuples
for fixtures (SURVEY.md section 8c/8d "Regime R").  The drivers themselves are out of
scope for the product; nothing here is imported by the package.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import ipd_oracle as O


def warmup_class1(c, r, l, p, q, gama, maxit=100):
    """`warmup_class1.m:22-96` with res = 0, maxit finite (the driver's call, `:53-59`)."""
    m, n = len(l), len(r)
    M = m + n
    prox = lambda x: np.minimum(np.maximum(0.0, x), gama)
    b = np.concatenate([r, l])
    Atb = O.Aty(b, p, q)
    z0 = np.zeros(M)
    muf, gk, bk = 0.0, 1.0, 1.0
    xk = np.zeros(m * n)
    vk, wk, pik = xk.copy(), xk.copy(), xk.copy()
    lk = np.concatenate([z0, xk])
    for _ in range(maxit):
        ak = bk
        bk1 = bk / (1 + ak)
        gk1 = (gk + muf * ak) / (1 + ak)
        etafk = (1 + ak) * gk + muf * ak
        sgk = 1 / bk1
        etagk = (1 + ak) * bk
        wwk = (ak * pik + wk) / (1 + ak)
        wxk = (ak * gk * vk + (gk + muf * ak) * xk) / etafk
        hlk = lk - 1 / bk * np.concatenate([O.Ax(xk, p, q) - b, xk - wk]) \
            + ak / bk * np.concatenate([z0, -(pik - wk)])
        cAw = -Atb - wk
        cAlk = O.Aty(hlk[:M], p, q) + hlk[M:]
        dd = etafk * wxk - ak ** 2 * (c + cAlk + sgk * cAw)
        tt = sgk * ak ** 2
        sg = 1 + etafk / tt
        xk1 = (dd - O.Aty(O.invAAt(O.Ax(dd, p, q), p, q, sg), p, q)) / (etafk + tt)
        vk1 = xk1 + (xk1 - xk) / ak
        blk = lk + ak / bk * np.concatenate([O.Ax(vk1, p, q) - b, vk1 - pik])
        wk1 = prox(wwk - ak ** 2 / etagk * (-blk[M:]))
        pik1 = wk1 + (wk1 - wk) / ak
        lk1 = lk + ak / bk * np.concatenate([O.Ax(vk1, p, q) - b, vk1 - pik1])
        gk, bk, xk, vk, wk, pik, lk = gk1, bk1, xk1, vk1, wk1, pik1, lk1
    return xk, lk[:M]


def apd_ssn_class1(c, r, l, p, q, gama, capture=(), inner="direct", maxit=100, rng=None,
                   amg_cycle="w", verbose=False):
    """`APD_SsN_Class1.m:30-275`.  `capture` = iterable of (k, ssn_it) pairs (1-based) whose
    Newton systems are recorded as dicts(s, bk1, tk, z, k, ssn).  inner = "direct"
    (inner_solver 1, `:146-148`) or "amg" (inner_solver 4, `:160-161`, oracle Hybrid_AMG)."""
    m, n = len(l), len(r)
    M = m + n
    prox = lambda x: np.minimum(np.maximum(0.0, x), gama)
    b = np.concatenate([r, l])
    KKT_Tol, bk = 1e-6, 1.0
    SsN_IT, SsN_Tol1, nu, delta, ll_max = 50, 1e-11, 0.2, 0.9, 500
    xk, lk = warmup_class1(c, r, l, p, q, gama, 100)
    vk = xk.copy()
    kkt_l0 = np.linalg.norm(O.Ax(xk, p, q) - b)
    kkt_x0 = np.linalg.norm(xk - prox(xk - c - O.Aty(lk, p, q)))
    kkt_x, kkt_l = kkt_x0, kkt_l0
    rng = rng or O.matlab_rng()
    want = set(capture)
    captured, log = [], []
    opts = O.amg_options_class1(amg_cycle)
    Tz = sp.csr_matrix((M, M))
    for k in range(1, maxit + 1):
        resk = max(kkt_x, kkt_l)
        ak = np.sqrt(k ** 2 * bk)
        bk1 = bk / (1 + ak)
        tk = bk * (1 + ak) / ak ** 2
        SsN_Tol = max(bk1 / k ** 2, SsN_Tol1)
        wk = -c + bk * (xk + ak * vk) / ak ** 2
        wlk = bk1 * (lk - 1 / bk * (O.Ax(xk, p, q) - b)) - b
        ssn_it = 0
        lk_new = lk
        zk = 1 / tk * (wk - O.Aty(lk_new, p, q))
        Fk_new = bk1 * lk_new - O.Ax(prox(zk), p, q) - wlk
        while np.linalg.norm(Fk_new) > SsN_Tol:
            ssn_it += 1
            lk_old = lk_new
            zk = 1 / tk * (wk - O.Aty(lk_old, p, q))
            s = (zk >= 0) & (zk <= gama)
            H0 = O.ASAt(s, p, q)
            Fk_old = bk1 * lk_old - O.Ax(prox(zk), p, q) - wlk
            if (k, ssn_it) in want:
                captured.append(dict(k=k, ssn=ssn_it, s=np.packbits(s), mn=m * n, bk1=bk1, tk=tk,
                                     z=-Fk_old, E=int(s.sum())))
            if inner == "direct":
                Jk = bk1 * sp.identity(M) + (Tz + H0) / tk
                zeta = spla.spsolve(sp.csc_matrix(Jk), -Fk_old)
                it_in = 1
            else:
                pd = dict(bk1=bk1, tk=tk, q=q, p=p, T=Tz, H0=H0, z=-Fk_old)
                zeta, it_in, _, _ = O.Hybrid_AMG(pd, opts, rng)
            f0 = bk1 / 2 * np.linalg.norm(lk_old) ** 2 - wlk @ lk_old
            cF_old = f0 + 0.5 * tk * np.linalg.norm(prox(zk)) ** 2
            ll = 0
            ress = abs(Fk_old @ zeta)
            while True:
                lk_new = lk_old + delta ** ll * zeta
                f0 = bk1 / 2 * np.linalg.norm(lk_new) ** 2 - wlk @ lk_new
                zk = 1 / tk * (wk - O.Aty(lk_new, p, q))
                cF_new = f0 + 0.5 * tk * np.linalg.norm(prox(zk)) ** 2
                if not (cF_new > cF_old - nu * delta ** ll * ress) or ll == ll_max:
                    break
                ll += 1
            Fk_new = bk1 * lk_new - O.Ax(prox(zk), p, q) - wlk
            log.append(dict(k=k, ssn=ssn_it, E=int(s.sum()), it=it_in, Fk=np.linalg.norm(Fk_new)))
            if np.linalg.norm(Fk_new) <= SsN_Tol:
                break
            if abs(np.linalg.norm(Fk_old) - np.linalg.norm(Fk_new)) < SsN_Tol / 100:
                break
            if ssn_it == SsN_IT:
                break
        lk1 = lk_new
        xk1 = prox(zk)
        vk1 = xk1 + (xk1 - xk) / ak
        kl = np.linalg.norm(O.Ax(xk1, p, q) - b)
        kx = np.linalg.norm(xk1 - prox(xk1 - c - O.Aty(lk1, p, q)))
        rr = max(kx / (1 + kkt_x0), kl / (1 + kkt_l0))
        if bk1 < 1e-8 and rr > resk:
            xk1, lk1, vk1, bk1 = xk, lk, xk, rng.random_sample()   # restart (:245-249)
        bk, xk, lk, vk = bk1, xk1, lk1, vk1
        kkt_l = np.linalg.norm(O.Ax(xk, p, q) - b)
        kkt_x = np.linalg.norm(xk - prox(xk - c - O.Aty(lk, p, q)))
        rr = max(kkt_x / (1 + kkt_x0), kkt_l / (1 + kkt_l0))
        if verbose:
            print("APD it=%3d KKT=%.2e f=%.6f ssn=%d" % (k, rr, c @ xk, ssn_it))
        if rr <= KKT_Tol:
            return dict(converged=True, k=k, fval=float(c @ xk), captured=captured, log=log)
    return dict(converged=False, k=maxit, fval=float(c @ xk), captured=captured, log=log)
