"""CPU restatement of the reference's drivers -- TEST INFRASTRUCTURE ONLY.

`Class1/warmup_class1.m`, `Class2/warmup_class2.m` (A-ADMM warm starts) and the APD /
semismooth-Newton loops of `Class1/APD_SsN_Class1.m:101-275` and
`Class2/APD_SsN_Class2.m:95-285`.  They produce REALISTIC (s, bk1, tk, z) tuples for
fixtures (SURVEY.md section 8c/8d "Regime R") and are the checker of the device drivers
(rows f1/f2, `csrc/ipd_driver.hip`).  Nothing here is imported by the package.
Parity unpinned: MATLAB is not available, see ipd_oracle.py's header.
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
                   amg_cycle="w", verbose=False, start=None, prob=2, bk0=1.0, vk0=None):
    """`APD_SsN_Class1.m:30-275`.  `capture` = iterable of (k, ssn_it) pairs (1-based) whose
    Newton systems are recorded as dicts(s, bk1, tk, z, k, ssn).  inner = "direct"
    (inner_solver 1, `:146-148`) or "amg" (inner_solver 4, `:160-161`, oracle Hybrid_AMG)."""
    m, n = len(l), len(r)
    M = m + n
    prox = lambda x: np.minimum(np.maximum(0.0, x), gama)
    b = np.concatenate([r, l])
    KKT_Tol, bk = 1e-6, 1.0
    SsN_IT, SsN_Tol1, nu, delta, ll_max = 50, 1e-11, 0.2, 0.9, 500
    xk, lk = start if start is not None else warmup_class1(c, r, l, p, q, gama, 100)
    vk = xk.copy() if vk0 is None else np.asarray(vk0, float).copy()
    bk = bk0
    hist = dict(fxk=[float(c @ xk)], KKT_xk=[], KKT_lk=[], SsN_itnum=[])
    kkt_l0 = np.linalg.norm(O.Ax(xk, p, q) - b)
    kkt_x0 = np.linalg.norm(xk - prox(xk - c - O.Aty(lk, p, q)))
    kkt_x, kkt_l = kkt_x0, kkt_l0
    hist["KKT_xk"].append(kkt_x0)
    hist["KKT_lk"].append(kkt_l0)
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
            def merit(f0, zk):                                   # :183-187 / :192-196
                if prob < 3:
                    return f0 + 0.5 * tk * np.linalg.norm(prox(zk)) ** 2
                return f0 + 0.5 * tk * (np.linalg.norm(zk) ** 2 - np.linalg.norm(zk - prox(zk)) ** 2)

            f0 = bk1 / 2 * np.linalg.norm(lk_old) ** 2 - wlk @ lk_old
            cF_old = merit(f0, zk)
            ll = 0
            ress = abs(Fk_old @ zeta)
            while True:
                lk_new = lk_old + delta ** ll * zeta
                f0 = bk1 / 2 * np.linalg.norm(lk_new) ** 2 - wlk @ lk_new
                zk = 1 / tk * (wk - O.Aty(lk_new, p, q))
                cF_new = merit(f0, zk)
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
        hist["fxk"].append(float(c @ xk))
        hist["KKT_xk"].append(kkt_x)
        hist["KKT_lk"].append(kkt_l)
        hist["SsN_itnum"].append(ssn_it)
        if verbose:
            print("APD it=%3d KKT=%.2e f=%.6f ssn=%d" % (k, rr, c @ xk, ssn_it))
        if rr <= KKT_Tol:
            return dict(converged=True, k=k, fval=float(c @ xk), captured=captured, log=log,
                        xk=xk, lk=lk, bk=bk, **hist)
    return dict(converged=False, k=maxit, fval=float(c @ xk), captured=captured, log=log,
                xk=xk, lk=lk, bk=bk, vk=vk, **hist)


# ---------------------------------------------------------------------------
# Class 2 (partial optimal transport)
# ---------------------------------------------------------------------------
def _H(u, p, q, phi, m, n):
    """`[Ax(xk)+[yk;zk]; phi'*xk]` (`APD_SsN_Class2.m:45`)."""
    mn = m * n
    return np.concatenate([O.Ax(u[:mn], p, q) + u[mn:], [phi @ u[:mn]]])


def _Ht(lam, p, q, phi, m, n):
    """`[Aty(lk(1:m+n))+lk(m+n+1)*phi; lk(1:m+n)]` (`APD_SsN_Class2.m:122`)."""
    M = m + n
    return np.concatenate([O.Aty(lam[:M], p, q) + lam[M] * phi, lam[:M]])


def warmup_class2(c, r, l, p, q, mu, phi, maxit=100):
    """`warmup_class2.m:19-100` with res = 0 and a finite maxit."""
    m, n = len(l), len(r)
    M, mn = m + n, m * n
    U = mn + M
    b = np.concatenate([r, l, [mu]])
    Htb = np.concatenate([O.Aty(b[:M], p, q) + b[-1] * phi, b[:M]])
    wc = np.concatenate([c, np.zeros(M)])
    z0 = np.zeros(M + 1)
    muf, gk, bk = 0.0, 1.0, 1.0
    uk = np.zeros(U)
    vk, wk, pik = uk.copy(), uk.copy(), uk.copy()
    lk = np.concatenate([z0, uk])
    for _ in range(maxit):
        ak = bk
        bk1 = bk / (1 + ak)
        gk1 = (gk + muf * ak) / (1 + ak)
        etafk = (1 + ak) * gk + muf * ak
        sgk = 1 / bk1
        etagk = (1 + ak) * bk
        wwk = (ak * pik + wk) / (1 + ak)
        wuk = (ak * gk * vk + (gk + muf * ak) * uk) / etafk
        hlk = lk - 1 / bk * np.concatenate([_H(uk, p, q, phi, m, n) - b, uk - wk]) \
            + ak / bk * np.concatenate([z0, -(pik - wk)])
        cAw = -Htb - wk
        cAlk = hlk[M + 1:] + _Ht(hlk[:M + 1], p, q, phi, m, n)
        dd = etafk * wuk - ak ** 2 * (wc + cAlk + sgk * cAw)
        tt = sgk * ak ** 2
        sg = 1 + etafk / tt
        Hdd = np.concatenate([O.Ax(dd[:mn], p, q) + dd[mn:], [phi @ dd[:mn]]])
        ff = O.invHHt(Hdd, p, q, sg, phi)
        uk1 = (dd - np.concatenate([O.Aty(ff[:M], p, q) + ff[-1] * phi, ff[:M]])) / (etafk + tt)
        vk1 = uk1 + (uk1 - uk) / ak
        b0 = _H(vk1, p, q, phi, m, n) - b
        blk = lk + ak / bk * np.concatenate([b0, vk1 - pik])
        wk1 = np.maximum(0.0, wwk - ak ** 2 / etagk * (-blk[M + 1:]))
        pik1 = wk1 + (wk1 - wk) / ak
        lk1 = lk + ak / bk * np.concatenate([b0, vk1 - pik1])
        gk, bk, uk, vk, wk, pik, lk = gk1, bk1, uk1, vk1, wk1, pik1, lk1
    return uk, lk[:M + 1]


def apd_ssn_class2(c, r, l, p, q, mu, phi, inner="direct", maxit=100, rng=None, start=None,
                   verbose=False):
    """`APD_SsN_Class2.m:29-285`; inner = "direct" (inner_solver 1, `:151-156`) or "amg"
    (inner_solver 4, oracle AMG4POT)."""
    m, n = len(l), len(r)
    M, mn = m + n, m * n
    prox = lambda x: np.maximum(0.0, x)
    b = np.concatenate([r, l, [mu]])
    wc = np.concatenate([c, np.zeros(M)])
    KKT_Tol, bk = 1e-6, 1.0
    SsN_IT, SsN_Tol1, nu, delta, ll_max = 50, 1e-10, 0.2, 0.9, 500
    uk, lk = start if start is not None else warmup_class2(c, r, l, p, q, mu, phi, 100)
    vk = uk.copy()
    rng = rng or O.matlab_rng()
    opts = dict(retol=1e-11, bigph=1, maxit=40, theta=1 / 4, smoth=10, cycle="w", isnsp=1,
                inter=1, guess=None)

    def kkts(u, lam):
        x, y, z = u[:mn], u[mn:mn + n], u[mn + n:]
        kl = np.linalg.norm(_H(u, p, q, phi, m, n) - b)
        kz = np.linalg.norm(z - np.maximum(z - lam[n:M], 0))
        ky = np.linalg.norm(y - np.maximum(y - lam[:n], 0))
        kx = np.linalg.norm(x - np.maximum(x - c - (O.Aty(lam[:M], p, q) + lam[M] * phi), 0))
        return np.array([kx, ky, kz, kl])

    k0 = kkts(uk, lk)
    kk = k0.copy()
    hist = dict(fxk=[float(c @ uk[:mn])], KKT_xk=[k0[0]], KKT_yk=[k0[1]], KKT_zk=[k0[2]],
                KKT_lk=[k0[3]], SsN_itnum=[])
    log = []
    for k in range(1, maxit + 1):
        resk = kk.max()
        ak = np.sqrt(k ** 2 * bk)
        bk1 = bk / (1 + ak)
        tk = bk * (1 + ak) / ak ** 2
        SsN_Tol = max(bk1 / k ** 2, SsN_Tol1)
        wk = -wc + bk * (uk + ak * vk) / ak ** 2
        wlk = bk1 * (lk - 1 / bk * (_H(uk, p, q, phi, m, n) - b)) - b
        ssn_it = 0
        lk_new = lk

        def F_of(lam):
            zk = 1 / tk * (wk - _Ht(lam, p, q, phi, m, n))
            return zk, bk1 * lam - _H(prox(zk), p, q, phi, m, n) - wlk

        zk, Fk_new = F_of(lk_new)
        while np.linalg.norm(Fk_new) > SsN_Tol:
            ssn_it += 1
            lk_old = lk_new
            zk, Fk_old = F_of(lk_old)
            s = zk[:mn] >= 0
            t = zk[mn:] >= 0
            H0 = O.ASAt(s, p, q)
            T = sp.diags(t.astype(float))
            if inner == "direct":
                ss = O.Ax(s * phi, p, q)
                cT = sp.bmat([[T, None], [None, sp.csr_matrix((1, 1))]])
                cH0 = sp.bmat([[H0, ss[:, None]], [ss[None, :], np.array([[phi @ (s * phi)]])]])
                Jk = bk1 * sp.identity(M + 1) + (cT + cH0) / tk
                zeta = spla.spsolve(sp.csc_matrix(Jk), -Fk_old)
                it_in = 1
            else:
                pd = dict(bk1=bk1, tk=tk, q=q, p=p, s=s, T=T, H0=H0, z=-Fk_old, phi=phi)
                zeta, it_in, _, _ = O.AMG4POT(pd, opts, rng)
            f0 = bk1 / 2 * np.linalg.norm(lk_old) ** 2 - wlk @ lk_old
            cF_old = f0 + 0.5 * tk * np.linalg.norm(prox(zk)) ** 2
            ress = abs(Fk_old @ zeta)
            ll = 0
            while True:
                lk_new = lk_old + delta ** ll * zeta
                f0 = bk1 / 2 * np.linalg.norm(lk_new) ** 2 - wlk @ lk_new
                zk, Fk_new = F_of(lk_new)
                cF_new = f0 + 0.5 * tk * np.linalg.norm(prox(zk)) ** 2
                if not (cF_new > cF_old - nu * delta ** ll * ress) or ll == ll_max:
                    break
                ll += 1
            log.append(dict(k=k, ssn=ssn_it, E=int(s.sum()), it=it_in, ll=ll,
                            Fk=np.linalg.norm(Fk_new)))
            if np.linalg.norm(Fk_new) <= SsN_Tol:
                break
            if abs(np.linalg.norm(Fk_old) - np.linalg.norm(Fk_new)) < SsN_Tol:
                break
            if ssn_it == SsN_IT:
                break
        lk1 = lk_new
        uk1 = prox(zk)
        vk1 = uk1 + (uk1 - uk) / ak
        kk1 = kkts(uk1, lk1)
        if bk1 < 1e-8 and (kk1 / (1 + k0)).max() > resk:
            uk1, lk1, vk1, bk1 = uk, lk, uk, 10 * bk1                       # restart (:230-234)
        bk, uk, lk, vk = bk1, uk1, lk1, vk1
        kk = kkts(uk, lk)
        rr = (kk / (1 + k0)).max()
        hist["fxk"].append(float(c @ uk[:mn]))
        for name, val in zip(("KKT_xk", "KKT_yk", "KKT_zk", "KKT_lk"), kk):
            hist[name].append(val)
        hist["SsN_itnum"].append(ssn_it)
        if verbose:
            print("APD it=%3d KKT=%.2e f=%.6f ssn=%d" % (k, rr, c @ uk[:mn], ssn_it))
        if rr <= KKT_Tol:
            return dict(converged=True, k=k, fval=float(c @ uk[:mn]), log=log, uk=uk, lk=lk,
                        bk=bk, **hist)
    return dict(converged=False, k=maxit, fval=float(c @ uk[:mn]), log=log, uk=uk, lk=lk, bk=bk,
                **hist)
