"""Host-side mirror of the reference's MATLAB function signatures.

Same names, argument meaning and error behaviour as the reference's ``.m``
files; every call goes through the C ABI of libipdamg.so
(hand-written HIP, gfx950).  MATLAB itself is not available in this pipeline,
so this Python layer plays the role of the MEX gateway for tests and benches;
the gateway a MATLAB maintainer would compile is ``mex/ipd_mex.cpp``.

Differences forced by the host language, all documented in INTEGRATION.md:
 * indices are 0-based;
 * functions that consume MATLAB's global ``rand`` stream take a ``rng``
   argument (``MatlabRand``; seed 5489 reproduces MATLAB's default stream);
 * the reference's ``global Ack Prok J smoth_it Rk`` is the ``AMGHierarchy``
   handle; ``MG_Vcycle``/``MG_Wcycle`` take it as first argument.
"""
from __future__ import annotations

from ctypes import byref, c_double, c_int, c_int32, c_int64, c_void_p, POINTER

import numpy as np
import scipy.sparse as sp

from . import _lib as L
from ._lib import CscIn, IpdError, MatlabRand, bptr, check, csc_out_to_scipy, dptr, f64, get_ctx, \
    iptr, lib, u8

__all__ = [
    "Ax", "Aty", "ASAt", "invAAt", "invHHt", "strength", "cf_split", "mis_set", "transfer",
    "Class_AMG", "AMGHierarchy", "MG_Vcycle", "MG_Wcycle", "PCG", "components", "Hybrid_AMG",
    "AMG4POT", "MatlabRand", "IpdError", "amg_options", "APDWorkspace", "warmup_class1",
    "warmup_class2", "APD_SsN_Class1", "APD_SsN_Class2", "twogrid_bigph", "twogrid", "Hybrid_twogrid",
    "aug_PCG", "PCG4POT", "load_input", "sparse_multiply", "spd_solve",
]


def _h():
    return get_ctx().handle


# ---------------------------------------------------------------------------
# options structs
# ---------------------------------------------------------------------------
def amg_options(retol=None, bigph=None, maxit=None, theta=None, smoth=None, cycle=None,
                isnsp=None, inter=None, guess=None, fnode=None) -> dict:
    """``struct('retol',..,'bigph',..,...)`` with ``[]`` spelled ``None``."""
    return dict(retol=retol, bigph=bigph, maxit=maxit, theta=theta, smoth=smoth, cycle=cycle,
                isnsp=isnsp, inter=inter, guess=guess, fnode=fnode)


def _opts_struct(o: dict | None) -> L.ipd_amg_opts:
    s = L.ipd_amg_opts()
    lib.ipd_amg_opts_init(byref(s))
    if o is None:
        # Class_AMG.m:22-23 (nargin == 2 defaults; note cycle = 1 selects no cycle)
        o = dict(retol=1e-12, bigph=0, maxit=20, theta=1 / 4, smoth=10, cycle=1, isnsp=1, inter=1)
    for key in ("retol", "theta"):
        if o.get(key) is not None:
            setattr(s, key, float(o[key]))
    for key in ("bigph", "maxit", "smoth", "isnsp", "inter"):
        if o.get(key) is not None:
            setattr(s, key, int(o[key]))
    if o.get("cycle") is not None:
        cyc = o["cycle"]
        s.cycle = ord(cyc) if isinstance(cyc, str) and len(cyc) == 1 else int(cyc)
    if o.get("fnode") is not None:
        s.fnode = int(o["fnode"])
    return s


# ---------------------------------------------------------------------------
# L0 / L1
# ---------------------------------------------------------------------------
def Ax(x, p, q) -> np.ndarray:
    """``y = Ax(x,p,q)`` (``Ax.m:2``)."""
    p, q, x = f64(p), f64(q), f64(x)
    m, n = p.size, q.size
    if x.size != m * n:
        raise ValueError("Ax: length(x) must be length(p)*length(q)")
    y = np.empty(m + n)
    check(lib.ipd_ax(_h(), dptr(x), dptr(p), dptr(q), c_int64(m), c_int64(n), dptr(y)))
    return y


def Aty(y, p, q) -> np.ndarray:
    """``z = Aty(y,p,q)`` (``Aty.m:2``)."""
    p, q, y = f64(p), f64(q), f64(y)
    m, n = p.size, q.size
    if y.size < m + n:
        raise ValueError("Aty: y needs n+m entries")
    z = np.empty(m * n)
    check(lib.ipd_aty(_h(), dptr(y), dptr(p), dptr(q), c_int64(m), c_int64(n), dptr(z)))
    return z


def ASAt(s, p, q) -> sp.csc_matrix:
    """``H = ASAt(s,p,q)`` (``ASAt.m:2``); ``s`` logical of length m*n."""
    p, q, s = f64(p), f64(q), u8(s)
    m, n = p.size, q.size
    if s.size != m * n:
        raise ValueError("ASAt: length(s) must be length(p)*length(q)")
    out = L.ipd_csc_out()
    check(lib.ipd_asat(_h(), bptr(s), dptr(p), dptr(q), c_int64(m), c_int64(n), byref(out)))
    return csc_out_to_scipy(out)


def invAAt(x, p, q, sg1=None, sg2=None) -> np.ndarray:
    """``y = invAAt(x,p,q,sg1,sg2)`` (``invAAt.m:1``; nargin rules ``:7-12``)."""
    if sg1 is None:
        sg1, sg2 = 1.0, 1.0
    elif sg2 is None:
        sg2 = sg1
    p, q, x = f64(p), f64(q), f64(x)
    m, n = p.size, q.size
    y = np.empty(m + n)
    check(lib.ipd_inv_aat(_h(), dptr(x), dptr(p), dptr(q), c_int64(m), c_int64(n), c_double(sg1),
                          c_double(sg2), dptr(y)))
    return y


def invHHt(v, p, q, sg, phi) -> np.ndarray:
    """``y = invHHt(v,p,q,sg,phi)`` (``Class2/invHHt.m:1``)."""
    p, q, v, phi = f64(p), f64(q), f64(v), f64(phi)
    m, n = p.size, q.size
    y = np.empty(m + n + 1)
    check(lib.ipd_inv_hht(_h(), dptr(v), dptr(p), dptr(q), c_int64(m), c_int64(n), c_double(sg),
                          dptr(phi), dptr(y)))
    return y


# ---------------------------------------------------------------------------
# L2: setup pieces
# ---------------------------------------------------------------------------
def strength(A, which: int = 2) -> sp.csc_matrix:
    """``S = strength(A,which)`` (``AMG/strength.m:1``)."""
    a = CscIn(A)
    out = L.ipd_csc_out()
    check(lib.ipd_strength(_h(), a.ref(), c_int(which), byref(out)))
    return csc_out_to_scipy(out)


def cf_split(S):
    """``[indC,indF] = cf_split(S)`` (``AMG/cf_split.m:1``)."""
    a = CscIn(S)
    n = a.struct.nrows
    indC = np.zeros(n, np.uint8)
    indF = np.zeros(n, np.uint8)
    check(lib.ipd_cf_split(_h(), a.ref(), bptr(indC), bptr(indF)))
    return indC.astype(bool), indF.astype(bool)


def mis_set(A, theta: float = 0.025, rng: MatlabRand | None = None):
    """``[isC,isF,As] = mis_set(A,theta)`` (``AMG/mis_set.m:1``)."""
    rng = rng or MatlabRand()
    a = CscIn(A)
    n = a.struct.nrows
    isC = np.zeros(n, np.uint8)
    isF = np.zeros(n, np.uint8)
    out = L.ipd_csc_out()
    check(lib.ipd_mis_set(_h(), a.ref(), c_double(theta), rng.handle, bptr(isC), bptr(isF),
                          byref(out)))
    return isC.astype(bool), isF.astype(bool), csc_out_to_scipy(out)


def sparse_multiply(A, B) -> sp.csc_matrix:
    """``A*B`` for sparse operands with the summation order of MATLAB's sparse ``mtimes`` (every
    entry accumulated in ascending inner index; ``transfer.m:66`` forms ``Pro'*A*Pro`` with it).
    The device kernels are the ones the setup uses (row kernel or register tiles)."""
    a, b = CscIn(A), CscIn(B)
    if a.struct.ncols != b.struct.nrows:
        raise ValueError("sparse_multiply: inner dimensions differ")
    da, db, dc = c_void_p(), c_void_p(), c_void_p()
    out = L.ipd_csc_out()
    try:
        check(lib.ipd_dmat_upload(_h(), a.ref(), c_int32(0), byref(da)))
        check(lib.ipd_dmat_upload(_h(), b.ref(), c_int32(0), byref(db)))
        check(lib.ipd_dmat_multiply(_h(), da, db, byref(dc)))
        check(lib.ipd_dmat_download(_h(), dc, byref(out)))
    finally:
        for d in (da, db, dc):
            if d:
                lib.ipd_dmat_destroy(d)
    return csc_out_to_scipy(out)


def spd_solve(A, B) -> np.ndarray:
    """``X = A \\ B`` for sparse symmetric positive definite ``A`` and dense ``B`` -- the
    reference's direct solves (``APD_SsN_Class1.m:148``, ``APD_SsN_Class2.m:155``,
    ``AMG/transfer.m:58``), a blocked dense Cholesky on the device."""
    a = CscIn(A)
    B = np.asarray(B, dtype=np.float64)
    one = B.ndim == 1
    B2 = B.reshape(-1, 1) if one else B
    n, nrhs = B2.shape
    if n != a.struct.nrows:
        raise ValueError("spd_solve: B must have as many rows as A")
    Bf = np.asfortranarray(B2)
    X = np.empty_like(Bf, order="F")
    check(lib.ipd_spd_solve(_h(), a.ref(), Bf.ctypes.data_as(POINTER(c_double)), c_int64(nrhs),
                            X.ctypes.data_as(POINTER(c_double))))
    return X[:, 0].copy() if one else np.ascontiguousarray(X)


def transfer(A, amg_options: dict, level: int = 2, rng: MatlabRand | None = None):
    """``[Ac,Pro,~,indC] = transfer(A,amg_options)`` (``AMG/transfer.m:1``);
    ``level`` is the reference's ``global J``."""
    rng = rng or MatlabRand()
    a = CscIn(A)
    o = _opts_struct(amg_options)
    Ac, Pro = L.ipd_csc_out(), L.ipd_csc_out()
    indC = np.zeros(a.struct.nrows, np.uint8)
    check(lib.ipd_transfer(_h(), a.ref(), byref(o), c_int(level), rng.handle, byref(Ac),
                           byref(Pro), bptr(indC)))
    return csc_out_to_scipy(Ac), csc_out_to_scipy(Pro), indC.astype(bool)


# ---------------------------------------------------------------------------
# L3: hierarchy, cycles, PCG
# ---------------------------------------------------------------------------
class AMGHierarchy:
    """Device-resident ``Ack/Prok/Rk/J`` (``AMG/Class_AMG.m:42-85``)."""

    def __init__(self, A, amg_options: dict, rng: MatlabRand | None = None, ctx=None):
        self.ctx = ctx or get_ctx()      # an own L.Context = an own HIP stream (concurrent use)
        self.rng = rng or MatlabRand()
        self._a = CscIn(A)
        self.opts = dict(amg_options) if amg_options is not None else None
        o = _opts_struct(amg_options)
        self.handle = c_void_p()
        check(lib.ipd_amg_setup(self.ctx.handle, self._a.ref(), byref(o), self.rng.handle,
                                byref(self.handle)))
        self.maxit = int(o.maxit) if o.maxit >= 0 else 50
        self.N = int(self._a.struct.nrows)

    @property
    def J(self) -> int:
        return int(lib.ipd_amg_num_levels(self.handle))

    def attach_mask_operator(self, p, q, tk) -> bool:
        """Matrix-free level 1 (``ipd_amg_attach_mask_operator``): True when level 1 is exactly
        Hybrid_AMG's rescaled operator for these ``p, q, tk`` and the 1-bit-per-entry sweeps are
        now in use; False leaves the CSR kernels in place."""
        p_, q_ = f64(p), f64(q)
        dp = L.DeviceBuffer.from_array(p_, self.ctx)
        dq = L.DeviceBuffer.from_array(q_, self.ctx)
        got = c_int32(0)
        check(lib.ipd_amg_attach_mask_operator(self.handle, dp.ptr, dq.ptr, c_int64(p_.size),
                                               c_int64(q_.size), c_double(float(tk)), byref(got)))
        self.ctx.sync()
        return bool(got.value)

    def attach_mask_transfers(self, p, q, tk) -> bool:
        """``ipd_amg_attach_mask_transfers``: the level-resident kernel's level 1 <-> 2 transfers from the
        bit mask (True when in use)."""
        p_, q_ = f64(p), f64(q)
        dp = L.DeviceBuffer.from_array(p_, self.ctx)
        dq = L.DeviceBuffer.from_array(q_, self.ctx)
        got = c_int32(0)
        check(lib.ipd_amg_attach_mask_transfers(self.handle, dp.ptr, dq.ptr, c_int64(p_.size),
                                                c_int64(q_.size), c_double(float(tk)), byref(got)))
        self.ctx.sync()
        return bool(got.value)

    def attach_level2_poly(self) -> bool:
        """``ipd_amg_attach_level2_poly``: level 2 of the level-resident kernel composed over a whole visit
        (three levels, one-row tail, V cycle: ONE hand-off per visit instead of ten).  True when in use."""
        got = c_int32(0)
        check(lib.ipd_amg_attach_level2_poly(self.handle, byref(got)))
        self.ctx.sync()
        return bool(got.value)

    def level_dims(self, k: int):
        rows, nnz = c_int64(), c_int64()
        check(lib.ipd_amg_level_dims(self.handle, c_int(k), byref(rows), byref(nnz)))
        return int(rows.value), int(nnz.value)

    def level_sizes(self):
        return [self.level_dims(k)[0] for k in range(1, self.J + 1)]

    def level_forms(self):
        """forms[k - 1] for level k = 1..J: bit mask of how the level runs inside the single-workgroup
        LDS images (include/ipd_amg.h, ipd_amg_level_forms); 0 = in no image."""
        buf = (c_int32 * (self.J + 1))()
        check(lib.ipd_amg_level_forms(self.handle, buf, c_int32(self.J + 1)))
        return [int(buf[k]) for k in range(1, self.J + 1)]

    def A(self, k: int) -> sp.csc_matrix:
        out = L.ipd_csc_out()
        check(lib.ipd_amg_get_A(self.handle, c_int(k), byref(out)))
        return csc_out_to_scipy(out)

    def P(self, k: int) -> sp.csc_matrix:
        out = L.ipd_csc_out()
        check(lib.ipd_amg_get_P(self.handle, c_int(k), byref(out)))
        return csc_out_to_scipy(out)

    def cmask(self, k: int) -> np.ndarray:
        rows = self.level_dims(k - 1)[0]
        m = np.zeros(rows, np.uint8)
        check(lib.ipd_amg_get_cmask(self.handle, c_int(k), bptr(m)))
        return m.astype(bool)

    def solve(self, b, guess=None):
        """Solve phase of Class_AMG: ``x, it, rel_res, rel_resk, rhok``."""
        b = f64(b)
        x = np.empty(self.N)
        it = c_int32()
        rel = c_double()
        rel_resk = np.full(self.maxit + 2, np.nan)
        rhok = np.full(self.maxit + 2, np.nan)
        g = f64(guess) if guess is not None else None
        check(lib.ipd_amg_solve(self.handle, dptr(b), dptr(g) if g is not None else None, dptr(x),
                                byref(it), byref(rel), dptr(rel_resk), dptr(rhok)))
        n = it.value + 1
        return x, int(it.value), float(rel.value), rel_resk[:n].copy(), rhok[:n].copy()

    def cycle_bytes(self) -> float:
        v = c_double()
        check(lib.ipd_amg_cycle_bytes(self.handle, byref(v)))
        return float(v.value)

    def close(self):
        if self.handle:
            lib.ipd_amg_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def MG_Vcycle(h: AMGHierarchy, r, isnsp=0, k: int = 1) -> np.ndarray:
    """``e = MG_Vcycle(r,isnsp,k)`` (``AMG/MG_Vcycle.m:2``); ``k`` is 1-based."""
    r = f64(r)
    e = np.empty(h.level_dims(k)[0])
    check(lib.ipd_amg_vcycle(h.handle, dptr(r), c_int(int(isnsp)), c_int(k), dptr(e)))
    return e


def MG_Wcycle(h: AMGHierarchy, r, isnsp=0, k: int = 1, e=None) -> np.ndarray:
    """``e = MG_Wcycle(r,isnsp,k,e)`` (``AMG/MG_Wcycle.m:2``)."""
    r = f64(r)
    out = np.empty(h.level_dims(k)[0])
    ein = f64(e) if e is not None else None
    check(lib.ipd_amg_wcycle(h.handle, dptr(r), c_int(int(isnsp)), c_int(k),
                             dptr(ein) if ein is not None else None, dptr(out)))
    return out


def Class_AMG(A, b, amg_options: dict | None = None, rng: MatlabRand | None = None):
    """``[x,it,rel_res,rel_resk,rhok] = Class_AMG(A,b,amg_options)``
    (``AMG/Class_AMG.m:1``)."""
    h = AMGHierarchy(A, amg_options, rng)
    try:
        guess = None if amg_options is None else amg_options.get("guess")
        return h.solve(b, guess)
    finally:
        h.close()


def PCG(H, e, pcg_options: dict | None = None):
    """``[d,it,res,resk] = PCG(H,e,pcg_options)`` (``PCG.m:1``)."""
    e = f64(e)
    a = CscIn(H)
    o = L.ipd_pcg_opts()
    lib.ipd_pcg_opts_init(byref(o))
    guess = None
    if pcg_options is not None:
        if pcg_options.get("retol") is not None:
            o.retol = float(pcg_options["retol"])
        if pcg_options.get("maxit") is not None:
            o.maxit = int(pcg_options["maxit"])
        if pcg_options.get("precd") is not None:
            o.precd = int(pcg_options["precd"])
        if pcg_options.get("nf") is not None:
            o.nf = int(pcg_options["nf"])
        if pcg_options.get("guess") is not None:
            guess = f64(pcg_options["guess"])
        if o.precd == 5 and "nf" not in pcg_options:
            raise ValueError("SSOR for bigraph requires pcg_options.nf!!!")   # PCG.m:64
    maxit = int(o.maxit) if o.maxit >= 0 else 10000
    d = np.empty(e.size)
    it = c_int64()
    res = c_double()
    resk = np.zeros(maxit)
    check(lib.ipd_pcg(_h(), a.ref(), dptr(e), dptr(guess) if guess is not None else None, byref(o),
                      dptr(d), byref(it), byref(res), dptr(resk)))
    return d, int(it.value), float(res.value), resk


# ---------------------------------------------------------------------------
# L4
# ---------------------------------------------------------------------------
def components(A):
    """``[blocks,sizes,p,r] = components(A)`` (``components.m:1``), 0-based."""
    a = CscIn(A)
    n = int(a.struct.nrows)
    blocks = np.zeros(n, np.int64)
    sizes = np.zeros(n, np.int64)
    p = np.zeros(n, np.int64)
    r = np.zeros(n + 1, np.int64)
    nc = c_int64()
    check(lib.ipd_components(_h(), a.ref(), iptr(blocks), iptr(sizes), iptr(p), iptr(r), byref(nc)))
    k = int(nc.value)
    return blocks, sizes[:k].copy(), p, r[:k + 1].copy()


def _prob_struct(pd: dict, need_pot: bool):
    keep = []
    p, q = f64(pd["p"]), f64(pd["q"])
    m, n = p.size, q.size
    T = pd.get("T")
    t = None
    if T is not None:
        t = f64(sp.csr_matrix(T).diagonal() if sp.issparse(T) else np.asarray(T))
        if not np.any(t):
            t = None
    H0 = CscIn(pd["H0"])
    z = f64(pd["z"])
    s = L.ipd_prob()
    s.m, s.n, s.bk1, s.tk = m, n, float(pd["bk1"]), float(pd["tk"])
    s.p, s.q = dptr(p), dptr(q)
    s.t = dptr(t) if t is not None else None
    s.H0 = POINTER(L.ipd_csc)(H0.struct)
    s.z = dptr(z)
    keep += [p, q, t, H0, z]
    if need_pot:
        sm = u8(pd["s"])
        phi = f64(pd["phi"])
        s.s, s.phi = bptr(sm), dptr(phi)
        keep += [sm, phi]
    return s, keep, m, n


def Hybrid_AMG(prob_data: dict, amg_options: dict, rng: MatlabRand | None = None):
    """``[zeta,itamg,resamg,info] = Hybrid_AMG(prob_data,amg_options)``
    (``Hybrid_AMG.m:1``)."""
    rng = rng or MatlabRand()
    s, keep, m, n = _prob_struct(prob_data, False)
    o = _opts_struct(amg_options)
    zeta = np.empty(m + n)
    it = c_int32()
    res = c_double()
    info = np.zeros(2, np.int64)
    check(lib.ipd_hybrid_amg(_h(), byref(s), byref(o), rng.handle, dptr(zeta), byref(it),
                             byref(res), iptr(info)))
    return zeta, int(it.value), float(res.value), info


def Hybrid_twogrid(prob_data: dict, amg_options: dict, rng: MatlabRand | None = None):
    """``[zeta,itamg,resamg,info] = Hybrid_twogrid(prob_data,amg_options)``
    (``Hybrid_twogrid.m:1``)."""
    rng = rng or MatlabRand()
    s, keep, m, n = _prob_struct(prob_data, False)
    o = _opts_struct(amg_options)
    zeta = np.empty(m + n)
    it = c_int32()
    res = c_double()
    info = np.zeros(2, np.int64)
    check(lib.ipd_hybrid_twogrid(_h(), byref(s), byref(o), rng.handle, dptr(zeta), byref(it),
                                 byref(res), iptr(info)))
    return zeta, int(it.value), float(res.value), info


def twogrid_bigph(A, b, amg_options: dict | None = None):
    """``[x,it,rel_res,rel_resk,rhok] = twogrid_bigph(A,b,amg_options)``
    (``AMG/twogrid_bigph.m:1``); ``amg_options.fnode`` is required."""
    b = f64(b)
    if amg_options is None:                       # :6-9 (fnode = 0 cannot work; kept as an error)
        amg_options = dict(retol=1e-12, maxit=20, fnode=0, smoth=10, isnsp=1, guess=None)
    a = CscIn(A)
    o = _opts_struct(dict(amg_options))
    maxit = int(o.maxit) if o.maxit >= 0 else 50
    x = np.empty(b.size)
    it = c_int32()
    rel = c_double()
    rel_resk = np.full(maxit + 2, np.nan)
    rhok = np.full(maxit + 2, np.nan)
    g = amg_options.get("guess")
    g = f64(g) if g is not None else None
    check(lib.ipd_twogrid_bigph(_h(), a.ref(), dptr(b), dptr(g) if g is not None else None, byref(o),
                                dptr(x), byref(it), byref(rel), dptr(rel_resk), dptr(rhok)))
    k = it.value + 1
    return x, int(it.value), float(rel.value), rel_resk[:k].copy(), rhok[:k].copy()


def twogrid(A, b, amg_options: dict | None = None, rng: MatlabRand | None = None):
    """``[x,it,rel_res,rel_resk,rhok] = twogrid(A,b,amg_options)`` (``AMG/twogrid.m:1``)."""
    b = f64(b)
    if amg_options is None:                                               # :6-9
        amg_options = dict(retol=1e-12, bigph=0, maxit=20, smoth=10, isnsp=1, guess=None)
    if amg_options.get("bigph") and not (amg_options.get("fnode") or 0) > 0:
        raise ValueError("bigph = 1 requires fnode > 0")                  # :24-26
    rng = rng or MatlabRand()
    a = CscIn(A)
    o = _opts_struct(dict(amg_options))
    maxit = int(o.maxit) if o.maxit >= 0 else 50
    x = np.empty(b.size)
    it = c_int32()
    rel = c_double()
    rel_resk = np.full(maxit + 2, np.nan)
    rhok = np.full(maxit + 2, np.nan)
    g = amg_options.get("guess")
    g = f64(g) if g is not None else None
    check(lib.ipd_twogrid(_h(), a.ref(), dptr(b), dptr(g) if g is not None else None, byref(o),
                          rng.handle, dptr(x), byref(it), byref(rel), dptr(rel_resk), dptr(rhok)))
    k = it.value + 1
    return x, int(it.value), float(rel.value), rel_resk[:k].copy(), rhok[:k].copy()


def _pcg_opts_struct(o: dict | None) -> L.ipd_pcg_opts:
    s = L.ipd_pcg_opts()
    lib.ipd_pcg_opts_init(byref(s))
    if o:
        if o.get("retol") is not None:
            s.retol = float(o["retol"])
        if o.get("maxit") is not None:
            s.maxit = int(o["maxit"])
        if o.get("precd") is not None:
            s.precd = int(o["precd"])
    return s


def aug_PCG(prob_data: dict, pcg_options: dict | None = None):
    """``[zeta,itpcg,respcg,info] = aug_PCG(prob_data,pcg_options)`` (``aug_PCG.m:1``)."""
    s, keep, m, n = _prob_struct(prob_data, False)
    o = _pcg_opts_struct(pcg_options)
    zeta = np.empty(m + n)
    it = c_int64()
    res = c_double()
    info = np.zeros(2, np.int64)
    check(lib.ipd_aug_pcg(_h(), byref(s), byref(o), dptr(zeta), byref(it), byref(res), iptr(info)))
    return zeta, int(it.value), float(res.value), info


def PCG4POT(prob_data: dict, pcg_options: dict | None = None):
    """``[zeta,itpcg,respcg,info] = PCG4POT(prob_data,pcg_options)``
    (``Class2/PCG4POT.m:1``)."""
    s, keep, m, n = _prob_struct(prob_data, True)
    o = _pcg_opts_struct(pcg_options)
    zeta = np.empty(m + n + 1)
    it = c_int64()
    res = c_double()
    info = np.zeros(2, np.int64)
    check(lib.ipd_pcg4pot(_h(), byref(s), byref(o), dptr(zeta), byref(it), byref(res), iptr(info)))
    return zeta, int(it.value), float(res.value), info


def AMG4POT(prob_data: dict, amg_options: dict, str_: str = "amg", rng: MatlabRand | None = None):
    """``[zeta,itamg,resamg,info] = AMG4POT(prob_data,amg_options,str)``
    (``Class2/AMG4POT.m:1``); ``str`` = ``'amg'`` or ``'twogrid'`` (``:44-51``)."""
    if str_ not in ("amg", "twogrid"):
        raise ValueError("AMG4POT: str must be 'amg' or 'twogrid'")
    rng = rng or MatlabRand()
    s, keep, m, n = _prob_struct(prob_data, True)
    o = _opts_struct(amg_options)
    zeta = np.empty(m + n + 1)
    it = c_int32()
    res = c_double()
    info = np.zeros(2, np.int64)
    fn = lib.ipd_amg4pot if str_ == "amg" else lib.ipd_amg4pot_twogrid
    check(fn(_h(), byref(s), byref(o), rng.handle, dptr(zeta), byref(it), byref(res), iptr(info)))
    return zeta, int(it.value), float(res.value), info


# ---------------------------------------------------------------------------
# L5: drivers and warm starts
# ---------------------------------------------------------------------------
class APDWorkspace:
    """The workspace of ``Class1/APD_SsN_Class1.m`` / ``Class2/APD_SsN_Class2.m`` in HBM.

    ``cls=1``: ``c, r, l, p, q, gama`` as loaded from ``InputData/data1-*.mat`` (``gama`` scalar,
    ``inf`` allowed, or an mn-vector); ``cls=2``: ``c (= C(:)), r, l, p, q, mu, phi`` of
    ``data4-*.mat``.  The scripts' variables are reachable as attributes/methods: ``state()``
    -> ``(uk, vk, lk, bk)``, ``history()`` -> ``fxk, KKT_xk, KKT_lk[, KKT_yk, KKT_zk],
    SsN_itnum``, ``records()`` -> what the scripts print per Newton step.
    """

    def __init__(self, cls, c, r, l, p, q, gama=np.inf, mu=0.0, phi=None, ctx=None):
        # ctx: an own L.Context (own HIP stream and arenas) lets several workspaces run
        # concurrently from several host threads; default = the process-wide context
        self._ctx = ctx
        self.cls = int(cls)
        self._keep = [f64(c), f64(r), f64(l), f64(p), f64(q)]
        c_, r_, l_, p_, q_ = self._keep
        m, n = p_.size, q_.size
        if l_.size != m or r_.size != n or c_.size != m * n:
            raise ValueError("APDWorkspace: need length(l)=length(p)=m, length(r)=length(q)=n, "
                             "length(c)=m*n")
        d = L.ipd_apd_data()
        d.cls, d.m, d.n = self.cls, m, n
        d.c, d.r, d.l, d.p, d.q = dptr(c_), dptr(r_), dptr(l_), dptr(p_), dptr(q_)
        d.mu = float(mu)
        d.gama_scalar = np.inf
        if self.cls == 1:
            if np.ndim(gama) == 0 or np.size(gama) == 1:
                d.gama_scalar = float(np.asarray(gama).reshape(-1)[0])
            else:
                g = f64(gama)
                if g.size != m * n:
                    raise ValueError("gama must be a scalar or an m*n vector")
                self._keep.append(g)
                d.gama = dptr(g)
        else:
            if phi is None:
                raise ValueError("class 2 needs phi")
            ph = f64(phi)
            if ph.size != m * n:
                raise ValueError("phi must have m*n entries")
            self._keep.append(ph)
            d.phi = dptr(ph)
        self.m, self.n = m, n
        self.M = m + n
        self.L = self.M + (1 if self.cls == 2 else 0)
        self.U = m * n + (self.M if self.cls == 2 else 0)
        self.handle = c_void_p()
        check(lib.ipd_apd_create((ctx or get_ctx()).handle, byref(d), byref(self.handle)))

    def close(self):
        if getattr(self, "handle", None):
            lib.ipd_apd_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- warm start -----------------------------------------------------------
    def warmup(self, res=None, maxit=None):
        """``warmup_class1.m:3-20`` nargin rules: no res -> (1e-1, inf); no maxit -> inf."""
        if res is None:
            res, maxit = 1e-1, np.inf
        elif maxit is None:
            maxit = np.inf
        mi = -1 if np.isinf(maxit) else int(maxit)
        check(lib.ipd_apd_warmup(self.handle, c_double(float(res)), c_int64(mi)))
        u, _, lam, _ = self.state()
        return u, lam

    # -- workspace ------------------------------------------------------------
    def set_state(self, u=None, v=None, lam=None, bk=1.0):
        u_ = f64(u) if u is not None else None
        v_ = f64(v) if v is not None else None
        l_ = f64(lam) if lam is not None else None
        for a, size, name in ((u_, self.U, "u"), (v_, self.U, "v"), (l_, self.L, "lam")):
            if a is not None and a.size != size:
                raise ValueError(f"{name} must have {size} entries")
        check(lib.ipd_apd_set_state(self.handle, dptr(u_) if u_ is not None else None,
                                    dptr(v_) if v_ is not None else None,
                                    dptr(l_) if l_ is not None else None, c_double(float(bk))))

    def state(self):
        u, v, lam = np.empty(self.U), np.empty(self.U), np.empty(self.L)
        bk = c_double()
        check(lib.ipd_apd_get_state(self.handle, dptr(u), dptr(v), dptr(lam), byref(bk)))
        return u, v, lam, bk.value

    # -- the loop -------------------------------------------------------------
    def options(self, **kw) -> L.ipd_apd_opts:
        o = L.ipd_apd_opts()
        lib.ipd_apd_opts_init(c_int32(self.cls), byref(o))
        for key, val in kw.items():
            setattr(o, key, val)
        return o

    def run(self, amg_options: dict, rng: MatlabRand | None = None, iters: int | None = None,
            **opts) -> dict:
        o = self.options(**opts)
        ao = _opts_struct(amg_options)
        rng = rng or MatlabRand()
        res = L.ipd_apd_result()
        check(lib.ipd_apd_run(self.handle, byref(o), byref(ao), rng.handle,
                              c_int32(o.maxit if iters is None else int(iters)), byref(res)))
        return dict(converged=bool(res.converged), k=res.k, fval=res.fval, kkt=list(res.kkt),
                    rr=res.rr, SumAMG=res.sum_amg, TotalAMG=res.total_amg, FailAMG=res.fail_amg,
                    MaxAMG=res.max_amg, restarts=res.restarts, nrec=res.nrec)

    def history(self) -> dict:
        names = ["fxk", "KKT_xk", "KKT_lk", "KKT_yk", "KKT_zk", "SsN_itnum"]
        out = {}
        for which, name in enumerate(names):
            if self.cls == 1 and name in ("KKT_yk", "KKT_zk"):
                continue
            cnt = c_int64()
            check(lib.ipd_apd_history(self.handle, c_int32(which), None, c_int64(0), byref(cnt)))
            buf = np.empty(max(cnt.value, 1))
            check(lib.ipd_apd_history(self.handle, c_int32(which), dptr(buf), c_int64(cnt.value),
                                      byref(cnt)))
            out[name] = buf[:cnt.value].copy()
        return out

    def records(self) -> list:
        cnt = c_int64()
        check(lib.ipd_apd_records(self.handle, None, c_int64(0), byref(cnt)))
        arr = (L.ipd_ssn_rec * max(cnt.value, 1))()
        check(lib.ipd_apd_records(self.handle, arr, c_int64(cnt.value), byref(cnt)))
        keys = [f[0] for f in L.ipd_ssn_rec._fields_]
        return [{k: getattr(arr[i], k) for k in keys} for i in range(cnt.value)]

    def reuse_stats(self):
        """Row f3: (Newton steps solved with AMG, steps whose system repeated the previous one,
        setups that shared the previous step's levels 1-2)."""
        a, b, c = c_int64(0), c_int64(0), c_int64(0)
        check(lib.ipd_apd_reuse_stats(self.handle, byref(a), byref(b), byref(c)))
        return a.value, b.value, c.value

    # -- building blocks ------------------------------------------------------
    def begin(self, k: int):
        vals = (c_double * 3)()
        check(lib.ipd_apd_begin(self.handle, c_int32(int(k)), vals))
        return dict(bk1=vals[0], tk=vals[1], ak=vals[2])

    def eval(self, lam):
        lam = f64(lam)
        if lam.size != self.L:
            raise ValueError(f"lam must have {self.L} entries")
        s = np.empty(self.m * self.n, np.uint8)
        t = np.empty(self.M)
        F = np.empty(self.L)
        vals = (c_double * 6)()
        check(lib.ipd_apd_eval(self.handle, dptr(lam), bptr(s), dptr(t) if self.cls == 2 else None,
                               dptr(F), vals))
        return dict(s=s.astype(bool), t=(t != 0) if self.cls == 2 else None, Fk=F, bk1=vals[0],
                    tk=vals[1], ak=vals[2], Fk_norm=vals[3], cFk=vals[4], E=int(vals[5]))

    def bench_eval(self, reps: int = 100):
        ms, by = c_double(), c_double()
        check(lib.ipd_apd_bench_eval(self.handle, c_int32(int(reps)), byref(ms), byref(by)))
        return ms.value, by.value


def warmup_class1(c, r, l, p, q, gama, res=None, maxit=None):
    """``[xk,lk] = warmup_class1(c,r,l,p,q,gama,res,maxit)`` (``Class1/warmup_class1.m:2``)."""
    if res is not None and maxit is not None and res == 0 and np.isinf(maxit):
        raise ValueError("res = 0 and maxit = inf")
    ws = APDWorkspace(1, c, r, l, p, q, gama=gama)
    try:
        return ws.warmup(res, maxit)
    finally:
        ws.close()


def warmup_class2(c, r, l, p, q, mu, phi, res=None, maxit=None):
    """``[uk,lk] = warmup_class2(c,r,l,p,q,mu,phi,res,maxit)`` (``Class2/warmup_class2.m:1``)."""
    if res is not None and maxit is not None and res == 0 and np.isinf(maxit):
        raise ValueError("res = 0 and maxit = inf")
    ws = APDWorkspace(2, c, r, l, p, q, mu=mu, phi=phi)
    try:
        return ws.warmup(res, maxit)
    finally:
        ws.close()


def _run_script(ws: APDWorkspace, amg_opts: dict, rng, warm, opts) -> dict:
    try:
        if warm is not None:
            ws.warmup(*warm)
        out = ws.run(amg_opts, rng, **opts)
        u, v, lam, bk = ws.state()
        out.update(ws.history())
        out.update(uk=u, vk=v, lk=lam, bk=bk, records=ws.records(), reuse_stats=ws.reuse_stats())
        out["xk"] = u[:ws.m * ws.n]
        return out
    finally:
        ws.close()


def APD_SsN_Class1(c, r, l, p, q, gama=np.inf, prob=2, rng: MatlabRand | None = None,
                   amg_opts: dict | None = None, **opts) -> dict:
    """The script ``Class1/APD_SsN_Class1.m`` with ``inner_solver = 4`` on the workspace it
    loads (``:27``); returns the variables it leaves behind.  Warm start as ``:53-59``."""
    amg_opts = amg_opts or dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5, cycle="w",
                                isnsp=1, inter=1, guess=None)                          # :87-88
    warm = (0.0, 100) if prob > 0 else (5e-2, np.inf)                                  # :53-58
    ws = APDWorkspace(1, c, r, l, p, q, gama=gama)
    return _run_script(ws, amg_opts, rng, warm, dict(prob=int(prob), **opts))


def APD_SsN_Class2(c, r, l, p, q, mu, phi, rng: MatlabRand | None = None,
                   amg_opts: dict | None = None, **opts) -> dict:
    """The script ``Class2/APD_SsN_Class2.m`` with ``inner_solver = 4`` (AMG4POT 'amg')."""
    amg_opts = amg_opts or dict(retol=1e-11, bigph=1, maxit=40, theta=1 / 4, smoth=10, cycle="w",
                                isnsp=1, inter=1, guess=None)
    ws = APDWorkspace(2, c, r, l, p, q, mu=mu, phi=phi)
    return _run_script(ws, amg_opts, rng, (0.0, 100), opts)


def load_input(path: str) -> dict:
    """``load('./InputData/data1-*.mat')`` / ``data4-*.mat`` (``APD_SsN_Class1.m:27``,
    ``APD_SsN_Class2.m``): the reference's MAT-v5 workspaces as float64 arrays.

    The bundled files store ``p``, ``q`` (and ``phi``) as ``uint8`` and ``m``, ``n`` as ``uint16``;
    MATLAB promotes them silently in mixed arithmetic, here they are cast once (SURVEY A-12).
    Returns ``dict(cls, m, n, c, r, l, p, q, gama)`` for Class 1 and
    ``dict(cls, m, n, c, r, l, p, q, mu, phi)`` for Class 2 (``c = C(:)``, column-major)."""
    import scipy.io
    d = scipy.io.loadmat(path)
    vec = lambda k: np.asarray(d[k], dtype=np.float64).reshape(-1, order="F")
    out = dict(r=vec("r"), l=vec("l"), p=vec("p"), q=vec("q"))
    out["m"], out["n"] = out["l"].size, out["r"].size
    if "mu" in d:                                           # Class 2
        out["cls"] = 2
        out["c"] = vec("c") if "c" in d else vec("C")
        out["mu"] = float(np.asarray(d["mu"]).reshape(-1)[0])
        out["phi"] = vec("phi")
    else:
        out["cls"] = 1
        out["c"] = vec("c")
        g = vec("gama")
        out["gama"] = float(g[0]) if g.size == 1 or np.all(g == g[0]) else g
    if out["c"].size != out["m"] * out["n"]:
        raise ValueError("load_input: length(c) != m*n")
    return out
