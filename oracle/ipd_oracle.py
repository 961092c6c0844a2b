"""CPU oracle for the IPD-SsN-AMG hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy/SciPy restatement of the reference's MATLAB
functions on the hot path.  It is the *checker* for the
HIP library: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped product
(``codes_of_ipd_ssn_amg_method_amd``) never imports, links or calls it.

PARITY UNPINNED: the reference is pure MATLAB and neither MATLAB nor Octave
exists in the build container or on the GPU box; the reference ships no tests,
golden vectors or expected outputs.  What pins this
restatement instead is (1) the algebraic identities in
``tests/test_oracle_identities.py`` (ASAt == A*diag(s)*A' with the explicit A of
``Class1/APD_SsN_Class1.m:47``, Rk{1} == inv(tril(Ae)), Galerkin symmetry ...),
(2) the MT19937(5489) == MATLAB ``rand`` equivalence, (3) convergence of the
restated drivers on the reference's bundled ``InputData`` inputs.

Every function cites the reference ``file:line`` it follows (paths relative to
the reference root).  MATLAB semantics that are reproduced on purpose are
listed in SURVEY.md appendix A ("quirks").

Conventions
-----------
* All matrices are ``scipy.sparse.csr_matrix`` float64 with sorted indices and
  no explicit zeros (MATLAB never stores explicit zeros).  Symmetric operators
  are identical in CSR and MATLAB's CSC.
* Indices are 0-based here; MATLAB's 1-based values are shifted by one.
* Unknown ordering of the KKT system: ``[0..n)`` = column constraints (r),
  ``[n..n+m)`` = row constraints (l)  (``Class1/APD_SsN_Class1.m:33``).
* Summation order: every sparse product accumulates each output entry in
  ascending inner index with a separate multiply and add (no FMA) -- this is
  MATLAB's column-Gustavson order restated row-wise and is what
  SciPy's ``csr_matmat`` / ``csr_matvec`` do on sorted inputs
  (``tests/test_oracle_identities.py::test_spgemm_order`` checks it bit for bit
  against a pure-Python loop).
* ``rng`` arguments are ``numpy.random.RandomState`` objects: MATLAB's default
  ``rand`` stream (mt19937ar, seed 5489, 53-bit doubles) equals
  ``RandomState(5489).random_sample``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

__all__ = [
    "Ax", "Aty", "ASAt", "build_A", "invAAt", "invHHt",
    "strength", "strength_mask", "cf_split", "mis_set", "transfer",
    "Hierarchy", "amg_setup", "MG_Vcycle", "MG_Wcycle", "PCG", "Class_AMG",
    "components", "build_Ae", "Hybrid_AMG", "AMG4POT", "matlab_rng",
    "amg_options_class1", "amg_options_class2",
]


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def matlab_rng(seed: int = 5489) -> np.random.RandomState:
    """MATLAB's default global stream (mt19937ar).  ``rng('default')`` is seed
    5489 (== MATLAB ``rng(0)``); ``rand(n,1)`` == ``random_sample(n)``."""
    return np.random.RandomState(seed)


def _csr(M) -> sp.csr_matrix:
    """Canonical form: CSR float64, duplicates summed, explicit zeros dropped,
    indices sorted (MATLAB sparse invariants)."""
    M = sp.csr_matrix(M, dtype=np.float64)
    M.sum_duplicates()
    M.eliminate_zeros()
    M.sort_indices()
    return M


def _row_of(M: sp.csr_matrix) -> np.ndarray:
    """Row index of every stored entry of a CSR matrix."""
    return np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))


def _spgemm(X: sp.csr_matrix, Y: sp.csr_matrix) -> sp.csr_matrix:
    """Ordered sparse product (ascending inner index per output entry, exact
    zeros dropped afterwards, as MATLAB's sparse mtimes does)."""
    return _csr(_csr(X) @ _csr(Y))


def amg_options_class1(cycle: str = "w") -> dict:
    """``Class1/APD_SsN_Class1.m:87-88``."""
    return dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5,
                cycle=cycle, isnsp=1, inter=1, guess=None, fnode=None)


def amg_options_class2(cycle: str = "w") -> dict:
    """``Class2/APD_SsN_Class2.m:80-81``."""
    return dict(retol=1e-11, bigph=1, maxit=40, theta=1 / 4, smoth=10,
                cycle=cycle, isnsp=1, inter=1, guess=None, fnode=None)


# ----------------------------------------------------------------------------
# L0: matrix-free A operators            (Ax.m, Aty.m, invAAt.m, Class2/invHHt.m)
# ----------------------------------------------------------------------------
def build_A(p: np.ndarray, q: np.ndarray) -> sp.csr_matrix:
    """Explicit ``A = [kron(speye(n),p'); kron(q',speye(m))]``
    (``Class1/APD_SsN_Class1.m:47``); only used by identity tests."""
    m, n = len(p), len(q)
    top = sp.kron(sp.identity(n), np.asarray(p, float).reshape(1, m))
    bot = sp.kron(np.asarray(q, float).reshape(1, n), sp.identity(m))
    return _csr(sp.vstack([top, bot]))


def Ax(x: np.ndarray, p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """``Ax.m:10-13``: ``X = reshape(x,m,n); y = [X'*p; X*q]`` (column-major)."""
    m, n = len(p), len(q)
    X = np.asarray(x, float).reshape((m, n), order="F")
    return np.concatenate([X.T @ p, X @ q])


def Aty(y: np.ndarray, p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """``Aty.m:10-13``: ``z = vec(p*y1' + y2*q')``."""
    m, n = len(p), len(q)
    y1, y2 = y[:n], y[n:n + m]
    z1 = np.outer(p, y1)
    z2 = np.outer(y2, q)
    return (z1 + z2).reshape(-1, order="F")


def invAAt(x, p, q, sg1=None, sg2=None) -> np.ndarray:
    """``invAAt.m:7-20``: ``(diag(sg1,sg2) + A*A') \\ x`` in closed form."""
    if sg1 is None:
        sg1, sg2 = 1.0, 1.0
    elif sg2 is None:
        sg2 = sg1
    m, n = len(p), len(q)
    np_ = np.linalg.norm(p) ** 2
    nq = np.linalg.norm(q) ** 2
    vn, vm = x[:n], x[n:n + m]
    den = sg1 * sg2 + sg1 * nq + sg2 * np_
    yn = vn / (sg1 + np_) + (np_ / (sg1 + np_) * (q @ vn) - p @ vm) * q / den
    ym = vm / (sg2 + nq) + (nq / (sg2 + nq) * (p @ vm) - q @ vn) * p / den
    return np.concatenate([yn, ym])


def invHHt(v, p, q, sg, phi) -> np.ndarray:
    """``Class2/invHHt.m:7-17``: ``(sg*I + H*H') \\ v`` for ``H=(G,IY,IZ)``."""
    m, n = len(p), len(q)
    t = sg + np.linalg.norm(phi) ** 2
    l = Ax(phi, p, q)
    Vl = invAAt(l, p, q, sg + 1)
    s = t - l @ Vl
    v1, v2 = v[:n + m], v[-1]
    Vv1 = invAAt(v1, p, q, sg + 1)
    y1 = s * Vv1 + (l @ Vv1) * Vl - v2 * Vl
    y2 = v2 - l @ Vv1
    return np.concatenate([y1, [y2]]) / s


# ----------------------------------------------------------------------------
# L1: KKT assembly                                                    (ASAt.m)
# ----------------------------------------------------------------------------
def ASAt(s: np.ndarray, p: np.ndarray, q: np.ndarray) -> sp.csr_matrix:
    """``ASAt.m:14-19``.  ``Y = sparse(reshape(s,m,n))``, ``U = P*Y``,
    ``Q = Y*R``, ``H = [diag(U'*p)  R*U' ; P*Q  diag(Q*q)]``.

    Entry-wise: ``H(j,j) = sum_i (p_i*Y_ij)*p_i`` (ascending i),
    ``H(n+i,n+i) = sum_j (Y_ij*q_j)*q_j`` (ascending j),
    ``H(n+i,j) = p_i*(Y_ij*q_j)``, ``H(j,n+i) = q_j*(p_i*Y_ij)``.
    """
    p = np.asarray(p, float)
    q = np.asarray(q, float)
    m, n = len(p), len(q)
    Y = _csr(np.asarray(s, float).reshape((m, n), order="F") != 0)
    P = sp.diags(p, format="csr")
    R = sp.diags(q, format="csr")
    U = _csr(P @ Y)            # U_ij = p_i * Y_ij
    Q = _csr(Y @ R)            # Q_ij = Y_ij * q_j
    d1 = _csr(U.T) @ p         # (U'*p)_j : sequential over ascending i
    d2 = Q @ q                 # (Q*q)_i  : sequential over ascending j
    H = sp.bmat([[sp.diags(d1), _csr(R @ _csr(U.T))],
                 [_csr(P @ Q), sp.diags(d2)]], format="csr")
    return _csr(H)


# ----------------------------------------------------------------------------
# L2: AMG setup                    (AMG/strength.m, cf_split.m, mis_set.m, transfer.m)
# ----------------------------------------------------------------------------
def strength(A: sp.csr_matrix, which: int = 2) -> sp.csr_matrix:
    """``AMG/strength.m:6-18``: strength *values* ``S(i,j) = -a_ij / min(max_row(i),
    max_row(j))`` with ``max_row = max(D-A,[],2)`` (implicit zeros count, so it is
    never negative; ``<=0 -> Inf``)."""
    A = _csr(A)
    N = A.shape[0]
    ia = _row_of(A)
    ja = A.indices
    off = ia != ja
    ia, ja, s0 = ia[off], ja[off], -A.data[off]      # find(D - A): diagonal cancels
    max_row = np.zeros(N)                            # the diagonal of D-A is an implicit 0
    np.maximum.at(max_row, ia, s0)
    max_row[max_row <= 0] = np.inf
    if which == 1:
        sa = s0 / max_row[ia]
    else:
        sa = s0 / np.minimum(max_row[ia], max_row[ja])
    S = sp.csr_matrix((sa, (ia, ja)), shape=(N, N))
    S.sort_indices()        # keeps explicit zeros out: s0 != 0 and max_row > 0
    S.eliminate_zeros()     # s0/Inf == 0 entries vanish exactly as in MATLAB's sparse()
    return S


def strength_mask(A: sp.csr_matrix, theta: float) -> sp.csr_matrix:
    """``strength(A) >= theta`` (``AMG/mis_set.m:25``, ``transfer.m:28,33``): sparse
    logical, no diagonal, negative strengths dropped."""
    S = strength(A)
    keep = S.data >= theta
    ia = _row_of(S)[keep]
    As = sp.csr_matrix((np.ones(keep.sum()), (ia, S.indices[keep])), shape=S.shape)
    As.sort_indices()
    return As


def cf_split(S: sp.csr_matrix):
    """``AMG/cf_split.m:6-16`` (dead in the reference, named by north_star):
    sequential greedy first-visit C/F split on ``graph(S)``.  ``graph(S)`` of a
    symmetric logical matrix: neighbours of k = stored off-diagonal columns of
    row k (self loops are ignored for the split: k is marked C first and
    ``indF(kk)`` would re-mark it, so we exclude them explicitly -- SURVEY a10)."""
    S = _csr(S)
    N = S.shape[0]
    indF = np.zeros(N, bool)
    indC = np.zeros(N, bool)
    indU = np.ones(N, bool)
    for k in range(N):
        if indU[k]:
            kk = S.indices[S.indptr[k]:S.indptr[k + 1]]
            kk = kk[kk != k]
            indC[k] = True
            indU[k] = False
            indF[kk] = True
            indU[kk] = False
    return indC, indF


def mis_set(A: sp.csr_matrix, theta: float, rng: np.random.RandomState):
    """``AMG/mis_set.m:9-67`` -- the LIVE C/F split (randomised MIS, iFEM).

    Random consumption (SURVEY A-6/A-13): exactly ``sum(deg>0)`` doubles in the
    normal branch (``:35``), ``N0`` doubles in the degenerate branch (``:31``).
    Returns ``isC, isF, As`` plus a dict with the consumed random vector (so the
    HIP path can be fed the identical numbers).
    """
    A = _csr(A)
    N = A.shape[0]
    isF = np.zeros(N, bool)
    isC = np.zeros(N, bool)
    N0 = min(int(math.floor(math.sqrt(N))) + 1, 25)                     # :12
    As = strength_mask(A, theta)                                          # :25
    AsT = _csr(As.T)
    deg = np.diff(AsT.indptr).astype(np.float64)                          # :28-29 column counts
    info = dict(branch="mis", rand=None, rounds=0)
    if (deg > 0).sum() < 0.25 * math.sqrt(N):                             # :30
        rv = rng.random_sample(N0)
        picks = np.ceil(rv * N).astype(np.int64) - 1                      # :31 (1-based ceil)
        isC[picks] = True
        isF = ~isC                                                        # :32
        info.update(branch="degenerate", rand=rv)
        return isC, isF, As, info
    idx = deg > 0
    rv = rng.random_sample(int(idx.sum()))
    deg[idx] = deg[idx] + 0.1 * rv                                        # :35
    info["rand"] = rv
    isF[deg == 0] = True                                                  # :40
    isU = np.ones(N, bool)                                                # :41
    ri = _row_of(As)
    cj = As.indices
    upper = ri < cj                                                       # triu(.,1)
    eu_i, eu_j = ri[upper], cj[upper]
    n_isU = N
    while isC.sum() < N / 2 and n_isU > N0:                               # :42
        info["rounds"] += 1
        isS = deg > 0                                                     # :44-45
        both = isS[eu_i] & isS[eu_j]                                      # edges of As(S,S)
        i, j = eu_i[both], eu_j[both]
        ge = deg[i] >= deg[j]                                             # :50
        isS[j[ge]] = False                                                # :51
        isS[i[~ge]] = False                                               # :52
        isC[isS] = True                                                   # :53
        touched = ri[isC[cj]]                                             # :56 rows of As(:,isC)
        isF[touched] = True                                               # :57
        isU = ~(isF | isC)                                                # :58
        deg[~isU] = 0                                                     # :59
        n_isU = int(isU.sum())
        if n_isU <= N0:                                                   # :61-64
            isC[isU] = True
            n_isU = 0
    iso = np.diff(As.indptr) == 0                                         # :67 sum(As,2)==0
    isC[iso] = True
    isF[iso] = False
    return isC, isF, As, info


def _row_scale_div(M: sp.csr_matrix, d: np.ndarray) -> sp.csr_matrix:
    """``diag(d) \\ M``: MATLAB solves a diagonal system by division."""
    M = _csr(M).copy()
    M.data = M.data / np.repeat(d, np.diff(M.indptr))
    return M


def transfer(A: sp.csr_matrix, opts: dict, J: int, rng: np.random.RandomState):
    """``AMG/transfer.m:17-66``.  ``J`` is the reference's ``global J`` (current
    number of levels, 1-based).  Returns ``Ac, Pro, info``."""
    A = _csr(A)
    N = A.shape[0]
    theta = 1 / 4 if opts.get("theta") is None else opts["theta"]
    isnsp = 0 if opts.get("isnsp") is None else opts["isnsp"]
    inter = 1 if opts.get("inter") is None else opts["inter"]
    info = dict()
    if J == 1 and opts.get("bigph"):                                      # :19
        Nf = int(opts["fnode"])
        Nc = N - Nf
        Aff = A[:Nf, :Nf]
        Afc = _csr(A[:Nf, Nf:])
        dff = Aff.diagonal()
        if Aff.nnz != np.count_nonzero(dff):
            raise ValueError("bigph level-1 transfer needs a diagonal Aff (transfer.m:20-21)")
        W = _row_scale_div(Afc, -dff)                                     # :21  W = -Aff\Afc
        if isnsp == 1:                                                    # :22-24
            rs = W @ np.ones(Nc)
            W = _row_scale_div(W, rs)
        Pro = _csr(sp.vstack([W, sp.identity(Nc)]))                       # :25
        info.update(isC=np.arange(N) >= Nf, isF=np.arange(N) < Nf, mis=None)
    else:
        isC, isF, As, minfo = mis_set(A, theta, rng)                      # :41
        if np.any(isC & isF) or not np.all(isC | isF):
            # SURVEY A-6: undecided nodes would index out of range in transfer.m:46-47
            raise RuntimeError("mis_set left nodes in neither/both sets")
        C_node = np.flatnonzero(isC)
        F_node = np.flatnonzero(isF)
        Nc, Nf = len(C_node), len(F_node)
        pp = np.concatenate([F_node, C_node])                             # :46
        AA = _csr(A[pp, :][:, pp])
        Aff = _csr(AA[:Nf, :Nf])
        Afc = _csr(AA[:Nf, Nf:])
        if inter < 2:                                                     # :48
            dff = Aff.diagonal()
            W1 = _row_scale_div(Afc, -dff)                                # :49
            as_ = _csr(sp.identity(Nf) + As[F_node, :][:, F_node])        # :50
            Affs = _csr(Aff.multiply(as_))                                # :51
            X = _row_scale_div(Affs, -dff)                                # ((-Dff)\Affs)
            W2 = _spgemm(X, W1)                                           #   ... * W1
            # :52-56 quirk A-3: the isempty() test is always true for Nf>0,
            # so W = W1 + 0.5*W2 regardless of `inter`.
            W = _csr(W1 + 0.5 * W2) if Nf > 0 else _csr(W1 + inter * W2)
        else:                                                             # :57-58 ideal interp.
            W = spla.spsolve(sp.csc_matrix(-Aff), sp.csc_matrix(Afc))
            if not sp.issparse(W):                 # one coarse node: SciPy returns a 1-D array
                W = np.asarray(W).reshape(Nf, Nc)
            W = _csr(W)
            W.eliminate_zeros()                    # MATLAB never stores explicit zeros
        if isnsp == 1:                                                    # :60-62
            rs = W @ np.ones(Nc)
            W = _row_scale_div(W, rs)
        P = _csr(sp.vstack([W, sp.identity(Nc)]))
        inv = np.empty(N, np.int64)
        inv[pp] = np.arange(N)
        Pro = _csr(P[inv, :])                                             # :63 Pro(p,:) = P
        info.update(isC=isC, isF=isF, mis=minfo, As=As)
    T1 = _spgemm(_csr(Pro.T), A)                                          # :66 left to right
    Ac = _spgemm(T1, Pro)
    return Ac, Pro, info


# ----------------------------------------------------------------------------
# L3: hierarchy + cycles        (AMG/Class_AMG.m, MG_Vcycle.m, MG_Wcycle.m, PCG.m)
# ----------------------------------------------------------------------------
@dataclass
class Hierarchy:
    """The reference's ``global Ack Prok J smoth_it Rk`` (``Class_AMG.m:42-47``).
    Lists are 1-based like MATLAB cells: index 0 is unused."""
    Ack: List[Optional[sp.csr_matrix]] = field(default_factory=lambda: [None])
    Prok: List[Optional[sp.csr_matrix]] = field(default_factory=lambda: [None])
    Rk: List[Optional[sp.csr_matrix]] = field(default_factory=lambda: [None])
    J: int = 0
    smoth_it: int = 0
    info: List[Optional[dict]] = field(default_factory=lambda: [None])

    def level_sizes(self):
        return [self.Ack[k].shape[0] for k in range(1, self.J + 1)]

    def level_nnz(self):
        return [self.Ack[k].nnz for k in range(1, self.J + 1)]


def _fill_amg_defaults(o: Optional[dict], b) -> dict:
    """``Class_AMG.m:20-34`` (the empty-field defaults; the nargin==2 set is
    quirk A-8 and selects no cycle at all)."""
    if o is None:
        return dict(retol=1e-12, bigph=0, maxit=20, theta=1 / 4, smoth=10, cycle=1,
                    isnsp=1, inter=1, guess=np.zeros_like(b), fnode=None)
    o = dict(o)
    dflt = dict(retol=1e-12, bigph=0, maxit=50, theta=1 / 4, smoth=3, cycle="v",
                isnsp=0, inter=1)
    for key, val in dflt.items():
        if o.get(key) is None:
            o[key] = val
    if o.get("guess") is None:
        o["guess"] = np.zeros_like(b)
    return o


def coarsest_threshold(N: int) -> int:
    """``1 + fix(size(A,1)^(1/3))`` evaluated in floating point (quirk A-2)."""
    return 1 + int(math.floor(float(N) ** (1.0 / 3.0)))


def amg_setup(A: sp.csr_matrix, opts: dict, rng: np.random.RandomState,
              max_levels: int = 40) -> Hierarchy:
    """Setup phase of ``AMG/Class_AMG.m:41-85``."""
    A = _csr(A)
    if opts.get("bigph"):
        if opts.get("fnode") is None or opts["fnode"] <= 0:               # :36-40
            raise ValueError("amg_options.bigph = 1 requires Nf > 0")
    h = Hierarchy()
    h.smoth_it = int(opts["smoth"])
    h.J = 1
    h.Ack.append(A)
    h.Prok.append(None)
    h.info.append(None)
    dofk = A.shape[0]
    if opts.get("bigph"):                                                 # :48-59
        Nf = int(opts["fnode"])
        Nc = dofk - Nf
        dA = A.diagonal()
        U = _csr(A[:Nf, Nf:])
        invV = sp.diags(1.0 / dA[:Nf])
        invT = sp.diags(1.0 / dA[Nf:])
        low = _csr(_csr(_csr(-invT) @ _csr(U.T)) @ _csr(invV))           # -invT*U'*invV
        R1 = sp.bmat([[invV, None], [low, invT]], format="csr")
        h.Rk.append(_csr(R1))
    else:                                                                 # :72
        h.Rk.append(_csr(sp.diags(0.5 * (1.0 / A.diagonal()))))
    thr = coarsest_threshold(A.shape[0])
    Ak = A
    while Ak.shape[0] > thr:                                              # :76
        if h.J >= max_levels:
            raise RuntimeError("coarsening stalled")
        Ak, Pro, tinfo = transfer(h.Ack[h.J], opts, h.J, rng)             # :78
        h.J += 1
        h.Ack.append(Ak)
        h.Prok.append(Pro)
        h.info.append(tinfo)
        h.Rk.append(_csr(sp.diags(0.5 * (1.0 / Ak.diagonal()))))          # :84
    return h


def ichol0(H):
    """``ichol(H)`` with MATLAB's default options (``PCG.m:46``): zero-fill incomplete Cholesky,
    ``type = 'nofill'``, no drop tolerance, ``michol = 'off'``, ``diagcomp = 0``, lower factor.
    ``L`` has the pattern of ``tril(H)`` and
    ``L(i,k) = (H(i,k) - sum_{j<k} L(i,j) L(k,j)) / L(k,k)``,
    ``L(i,i) = sqrt(H(i,i) - sum_{j<i} L(i,j)^2)`` (sums over the common pattern, ascending j).
    The algorithm is the published IC(0); MATLAB's implementation is closed source, so rounding
    order is this restatement's.  Raises like MATLAB on a nonpositive pivot."""
    Hl = sp.tril(_csr(H), 0, format="csr")
    Hl.sort_indices()
    N = Hl.shape[0]
    rp, ci, va = Hl.indptr, Hl.indices, Hl.data.astype(float).copy()
    ldg = np.zeros(N)
    wrow = np.zeros(N)
    for i in range(N):
        b, e = rp[i], rp[i + 1]
        if e == b or ci[e - 1] != i:
            raise ValueError("Encountered nonpositive pivot.")
        for t in range(b, e - 1):
            k = ci[t]
            kb, ke = rp[k], rp[k + 1] - 1            # row k without its diagonal
            sdot = 0.0
            for u in range(kb, ke):
                sdot += va[u] * wrow[ci[u]]
            va[t] = (va[t] - sdot) / ldg[k]
            wrow[k] = va[t]
        sq = 0.0
        for t in range(b, e - 1):
            sq += va[t] * va[t]
        d = va[e - 1] - sq
        if not d > 0.0:
            raise ValueError("Encountered nonpositive pivot.")
        ldg[i] = math.sqrt(d)
        va[e - 1] = ldg[i]
        wrow[ci[b:e - 1]] = 0.0
    return sp.csr_matrix((va, ci.copy(), rp.copy()), shape=Hl.shape)


def PCG(H, e, pcg_options: Optional[dict] = None):
    """``PCG.m:18-88`` (Shewchuk B3).  Preconditioners 1 (none), 2 (Jacobi),
    3 (SSOR w=1.5) and 5 (bigraph SSOR) are restated; 4 is ``ichol(H)`` with MATLAB's
    defaults, i.e. IC(0) (``ichol0`` below; MATLAB's kernel is closed source: parity unpinned).
    Returns ``d, it, res, resk``."""
    e = np.asarray(e, float)
    if pcg_options is None:
        pcg_options = dict(guess=np.zeros_like(e), retol=1e-11, maxit=1e4, precd=2)
    o = dict(pcg_options)
    if o.get("guess") is None:
        o["guess"] = np.zeros_like(e)
    if o.get("retol") is None:
        o["retol"] = 1e-11
    if o.get("maxit") is None:
        o["maxit"] = 1e4
    if o.get("precd") is None:
        o["precd"] = 2
    ii = int(o["precd"])
    d0 = np.asarray(o["guess"], float)
    tol = float(o["retol"])
    maxit = int(o["maxit"])
    H = _csr(H)
    if ii == 1:
        prec = lambda r: r
    elif ii == 2:
        Pd = H.diagonal()
        prec = lambda r: r / Pd
    elif ii == 3:
        w = 1.5
        Lo = sp.tril(H, -1, format="csr")
        Up = sp.triu(H, 1, format="csr")
        Dg = sp.diags(H.diagonal(), format="csr")
        lo_m = sp.csr_matrix(Dg + w * Lo)
        up_m = sp.csr_matrix(Dg + w * Up)

        def prec(r):                                                      # PCG.m:96-99
            p1 = spla.spsolve_triangular(lo_m, r, lower=True)
            p2 = Dg @ p1
            # `w*(2-w) * (D+wU) \ p2` parses as ((w*(2-w))*(D+wU)) \ p2
            return spla.spsolve_triangular(sp.csr_matrix(w * (2 - w) * up_m), p2, lower=False)
    elif ii == 5:
        if "nf" not in o:
            raise ValueError("SSOR for bigraph requires pcg_options.nf!!!")
        w = 1.5
        Nf = int(o["nf"])
        dH = H.diagonal()
        U = _csr(H[:Nf, Nf:])
        invV = sp.diags(1.0 / dH[:Nf])
        invT = sp.diags(1.0 / dH[Nf:])
        Pm = w * (2 - w) * sp.bmat(
            [[invV + w ** 2 * (invV @ U @ invT @ U.T @ invV), -w * (invV @ U @ invT)],
             [-w * (invT @ U.T @ invV), invT]], format="csr")
        prec = lambda r: Pm @ r
    elif ii == 4:                                                         # PCG.m:44-50
        if not sp.issparse(H):
            raise ValueError("iC requires H is sparse!")
        Lc = ichol0(H)
        Lt = _csr(Lc.T)

        def prec(r):                                                      # :100-101  P\r ; P'\p
            return spla.spsolve_triangular(Lt, spla.spsolve_triangular(Lc, r, lower=True), lower=False)
    else:
        raise ValueError("precd must be 1..5")

    it = 0
    r = e - H @ d0                                                        # :68
    p = prec(r)
    delta_new = float(r @ p)
    d = d0.copy()
    delta_0 = delta_new
    resk = []
    while it < maxit and delta_new > tol ** 2 * delta_0:                  # :76
        delta_old = delta_new
        qv = H @ p
        alpha = delta_old / float(qv @ p)
        d = d + alpha * p
        r = r - alpha * qv
        w_ = prec(r)
        delta_new = float(r @ w_)
        beta = delta_new / delta_old
        p = w_ + beta * p
        it += 1
        with np.errstate(invalid="ignore", divide="ignore"):
            resk.append(math.sqrt(abs(delta_new / delta_0)) if delta_0 != 0 else float("nan"))
    with np.errstate(invalid="ignore", divide="ignore"):
        res = math.sqrt(abs(delta_new / delta_0)) if delta_0 != 0 else float("nan")   # r=0 -> NaN
    return d, it, res, np.array(resk)


def _smooth(A, R, r, e, isnsp, nu):
    """The smoothing loops of ``MG_Vcycle.m:14-25`` / ``:33-41``."""
    N = A.shape[0]
    if isnsp:
        xi = np.ones(N)
        xx = float((A.T @ xi) @ xi)          # xi'*A*xi, evaluated left to right
        Axi = A @ xi
        for _ in range(nu):
            g = r - A @ e
            xig = float(xi @ g)
            g = xi * (xig / xx) + R @ (g - Axi * (xig / xx))
            e = e + g
    else:
        for _ in range(nu):
            e = e + R @ (r - A @ e)
    return e


def MG_Vcycle(h: Hierarchy, r: np.ndarray, isnsp=0, k: int = 1) -> np.ndarray:
    """``AMG/MG_Vcycle.m:9-45`` (recursive, zero initial guess)."""
    R = h.Rk[k]
    A = h.Ack[k]
    if k < h.J:
        Rt = _csr(R.T)
        e = _smooth(A, R, r, np.zeros_like(r), isnsp, h.smoth_it)        # :14-25
        rr = r - A @ e                                                   # :27
        rrc = _csr(h.Prok[k + 1].T) @ rr
        eec = MG_Vcycle(h, rrc, isnsp, k + 1)                            # :29
        e = e + h.Prok[k + 1] @ eec                                      # :31
        e = _smooth(A, Rt, r, e, isnsp, h.smoth_it)                      # :33-41
        return e
    d, _, _, _ = PCG(A, r)                                               # :43
    return d


def MG_Wcycle(h: Hierarchy, r: np.ndarray, isnsp=0, k: int = 1, e=None) -> np.ndarray:
    """``AMG/MG_Wcycle.m:10-46``: two recursive corrections, the second one
    starting from the first one's result (``:28-30``)."""
    if e is None:
        e = np.zeros_like(r)
    R = h.Rk[k]
    A = h.Ack[k]
    if k < h.J:
        Rt = _csr(R.T)
        e = _smooth(A, R, r, e, isnsp, h.smoth_it)                       # :15-24
        rr = r - A @ e                                                   # :26
        rrc = _csr(h.Prok[k + 1].T) @ rr
        eec = MG_Wcycle(h, rrc, isnsp, k + 1)                            # :28
        eec = MG_Wcycle(h, rrc, isnsp, k + 1, eec)                       # :30
        e = e + h.Prok[k + 1] @ eec                                      # :32
        e = _smooth(A, Rt, r, e, isnsp, h.smoth_it)                      # :34-42
        return e
    d, _, _, _ = PCG(A, r)                                               # :44 (guess ignored)
    return d


def amg_solve(h: Hierarchy, b: np.ndarray, opts: dict):
    """Solve phase of ``AMG/Class_AMG.m:86-109``."""
    A = h.Ack[1]
    maxit = int(opts["maxit"])
    it = 0
    rhok = np.full(maxit + 1, np.nan)
    rel_resk = np.ones(maxit + 1)
    x = np.array(opts["guess"], float)
    res0 = np.linalg.norm(A @ x - b)                                     # :89
    if res0 == 0:                                                        # :91-92
        return x, 0, 0.0, np.array([0.0]), np.array([np.inf])
    it = 1                                                               # :94 (1-based)
    rel_res = 1.0
    while rel_resk[it - 1] > opts["retol"] and it <= maxit:              # :95
        r = b - A @ x
        if opts["cycle"] == "v":
            x = x + MG_Vcycle(h, r, opts["isnsp"])
        if opts["cycle"] == "w":
            x = x + MG_Wcycle(h, r, opts["isnsp"])
        res = np.linalg.norm(A @ x - b)                                  # :103
        rel_res = res / res0
        rel_resk[it] = rel_res
        rhok[it] = res / np.linalg.norm(r)
        it += 1
        if rhok[it - 1] > 1:                                             # :106
            break
    rel_resk = rel_resk[:it]
    rhok = rhok[:it]
    it -= 1                                                              # :108
    return x, it, rel_res, rel_resk, rhok


def Class_AMG(A, b, amg_options: Optional[dict], rng: np.random.RandomState,
              return_hierarchy: bool = False):
    """``AMG/Class_AMG.m:1-111``: ``[x,it,rel_res,rel_resk,rhok]``."""
    b = np.asarray(b, float)
    o = _fill_amg_defaults(amg_options, b)
    h = amg_setup(A, o, rng)
    out = amg_solve(h, b, o)
    return out + (h,) if return_hierarchy else out


# ----------------------------------------------------------------------------
# L4: problem-level solvers        (components.m, Hybrid_AMG.m, Class2/AMG4POT.m)
# ----------------------------------------------------------------------------
def components(A: sp.csr_matrix):
    """``components.m:32-55``: connected components of a symmetric pattern.

    MATLAB obtains them from ``dmperm`` whose block ORDER is undocumented and
    cannot be reproduced here (parity unpinned; it only influences the order in
    which components are visited and hence the ``rand`` stream and
    ``info(2)``).  This restatement numbers components by their smallest member
    and lists the members of each component in ascending order.
    Returns ``blocks`` (0-based labels), ``sizes``, ``p`` (permutation), ``r``
    (block boundaries, length k+1)."""
    A = _csr(A)
    n, m = A.shape
    if n != m:
        raise ValueError("Adjacency matrix must be square")
    pattern = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape)
    ncomp, lab = sp.csgraph.connected_components(pattern, directed=False)
    # scipy labels in order of first occurrence == smallest member
    sizes = np.bincount(lab, minlength=ncomp)
    p = np.argsort(lab, kind="stable")
    r = np.concatenate([[0], np.cumsum(sizes)])
    return lab.astype(np.int64), sizes.astype(np.int64), p.astype(np.int64), r.astype(np.int64)


def build_Ae(H0, T, p, q, bk1, tk):
    """``Hybrid_AMG.m:17-24``: ``Q0 = diag([q;-p])``, ``A0 = Q0*H0*Q0``,
    ``Q = Q0*Q0``, ``K = Q0*T*Q0``, ``Ae = bk1*Q + 1/tk*(K+A0)``."""
    qp = np.concatenate([np.asarray(q, float), -np.asarray(p, float)])
    if np.any(qp == 0):
        raise ValueError("p or q contains 0 !!!!!")
    M = len(qp)
    Q0 = sp.diags(qp, format="csr")
    A0 = _csr(_csr(Q0 @ _csr(H0)) @ Q0)
    Q = _csr(Q0 @ Q0)
    K = _csr(_csr(Q0 @ _csr(T)) @ Q0)
    Ae = _csr(bk1 * Q + 1 / tk * _csr(K + A0))
    return Ae, A0, Q, K, Q0, qp


def Hybrid_AMG(prob_data: dict, amg_options: dict, rng: np.random.RandomState,
               N0: int = 100, trace: Optional[list] = None, solver=None):
    """``Hybrid_AMG.m:12-113``: ``[zeta,itamg,resamg,info]``.

    ``trace`` (optional list) receives one dict per Class_AMG call with the
    hierarchy and residual history, for parity tests."""
    bk1, tk = prob_data["bk1"], prob_data["tk"]
    q, p = np.asarray(prob_data["q"], float), np.asarray(prob_data["p"], float)
    H0, z, T = prob_data["H0"], np.asarray(prob_data["z"], float), prob_data["T"]
    Ae, A0, Q, K, Q0, qp = build_Ae(H0, T, p, q, bk1, tk)
    f = qp * z                                                           # :24
    M = len(qp)
    blocks, sizes, ps, rs = components(A0)                               # :27
    num_comp = len(sizes)
    dK = K.diagonal()
    n = len(q)
    o = dict(amg_options)
    u = np.zeros(M)
    if num_comp == 1:                                                    # :30-48
        o["isnsp"] = 0 if dK.sum() else 1                                # :32-38
        o["fnode"] = n
        o["guess"] = bk1 * tk * rng.random_sample(M)                     # :40
        if solver is not None:
            x, itamg, resamg, rel_resk, rhok = solver(Ae, f, o)
            h = None
        else:
            x, itamg, resamg, rel_resk, rhok, h = Class_AMG(Ae, f, o, rng, True)
        if trace is not None:
            trace.append(dict(pk=np.arange(M), h=h, rel_resk=rel_resk, rhok=rhok, it=itamg,
                              isnsp=o["isnsp"], fnode=n, guess=o["guess"], x=x))
        u = x
        it_num = 1
    else:                                                                # :50-107
        itamg, resamg, it_num = 0, 0.0, 0
        large = np.flatnonzero(sizes > N0)                               # :53
        for k in large:                                                  # :55
            pk = np.sort(ps[rs[k]:rs[k + 1]])     # SURVEY A-9: F side first, ascending
            Aek = _csr(Ae[pk, :][:, pk])
            fk = f[pk]
            o["isnsp"] = 0 if dK[pk].sum() else 1                        # :60-66
            o["fnode"] = int((pk < n).sum())                             # :68 (pk<=n, 1-based)
            o["guess"] = bk1 * tk * rng.random_sample(len(pk))           # :69
            if solver is not None:
                dk, itk, resk_, rel_resk, rhok = solver(Aek, fk, o)
                h = None
            else:
                dk, itk, resk_, rel_resk, rhok, h = Class_AMG(Aek, fk, o, rng, True)
            if trace is not None:
                trace.append(dict(pk=pk, h=h, rel_resk=rel_resk, rhok=rhok, it=itk,
                                  isnsp=o["isnsp"], fnode=o["fnode"], guess=o["guess"], x=dk))
            u[pk] = dk
            itamg = max(itamg, itk)
            resamg = max(resamg, resk_)
            it_num = int(k) + 1                                          # :80 (1-based k)
        small = np.flatnonzero(sizes[blocks] <= N0)                      # :85
        if small.size:
            order = np.argsort(blocks[small], kind="stable")             # :86
            pk = small[order]
            Aes = _csr(bk1 * Q[pk, :][:, pk] + 1 / tk * _csr(K[pk, :][:, pk] + A0[pk, :][:, pk]))
            u[pk] = spla.spsolve(sp.csc_matrix(Aes), f[pk])              # :91 direct
    zeta = qp * u                                                        # :113
    return zeta, itamg, resamg, np.array([num_comp, it_num])


def twogrid_bigph(A, b, amg_options: Optional[dict] = None):
    """``AMG/twogrid_bigph.m:1-53`` with ``twogrid_it`` (``:55-84``): two-level method on the
    bigraph blocks; coarse solve ``PCG(Ac,rrc,struct('retol',[],'maxit',1e2,'precd',2))``.
    Returns ``x, it, rel_res, rel_resk, rhok``."""
    b = np.asarray(b, float)
    if amg_options is None:
        amg_options = dict(retol=1e-12, maxit=20, fnode=0, smoth=10, isnsp=1, guess=np.zeros_like(b))
    o = dict(amg_options)
    retol = 0 if o.get("retol") is None else o["retol"]                  # :11-15
    maxit = 50 if o.get("maxit") is None else int(o["maxit"])
    smoth = 3 if o.get("smoth") is None else int(o["smoth"])
    isnsp = 0 if o.get("isnsp") is None else int(o["isnsp"])
    guess = np.zeros_like(b) if o.get("guess") is None else np.asarray(o["guess"], float)
    A = _csr(A)
    N = A.shape[0]
    Nf = int(o["fnode"])
    Nc = N - Nf
    Aff, Afc, Acc = A[:Nf, :Nf], _csr(A[:Nf, Nf:]), A[Nf:, Nf:]           # :22-23
    invV = sp.diags(1.0 / Aff.diagonal(), format="csr")
    invT = sp.diags(1.0 / Acc.diagonal(), format="csr")
    R = _csr(sp.bmat([[invV, None], [-_spgemm(_spgemm(invT, _csr(Afc.T)), invV), invT]]))  # :26
    W = _csr(-(sp.diags(1.0 / Aff.diagonal()) @ Afc))                     # :28 Aff\Afc, Aff diagonal
    if isnsp == 1:
        W = _csr(sp.diags(1.0 / np.asarray(W @ np.ones(Nc)).ravel()) @ W)  # :29-31
    Pro = _csr(sp.vstack([W, sp.identity(Nc, format="csr")]))
    Ac = _spgemm(_spgemm(_csr(Pro.T), A), Pro)                            # :33 left to right
    return _twogrid_loop(A, b, R, Pro, Ac, retol, maxit, smoth, isnsp, guess)


def _twogrid_loop(A, b, R, Pro, Ac, retol, maxit, smoth, isnsp, guess):
    """The iteration shared by ``twogrid_bigph.m:35-52`` and ``twogrid.m:67-84`` with their
    (identical) ``twogrid_it`` / ``twogrid_iteration``."""
    N = A.shape[0]
    Rt = _csr(R.T)
    xi = np.ones(N)
    Axi = A @ xi
    xx = xi @ Axi

    def it_(r):                                                           # twogrid_it :55-84
        e = np.zeros_like(r)
        for Rm in (R, None, Rt):
            if Rm is None:
                rr = r - A @ e
                eec = PCG(Ac, Pro.T @ rr, dict(retol=None, maxit=100, precd=2, guess=None))[0]
                e = e + Pro @ eec
                continue
            for _ in range(smoth):
                g = r - A @ e
                if isnsp:
                    xig = xi @ g
                    g = xi * (xig / xx) + Rm @ (g - Axi * (xig / xx))
                else:
                    g = Rm @ g
                e = e + g
        return e

    it = 0
    rhok = np.full(maxit + 2, np.nan)
    rel_resk = np.ones(maxit + 2)
    x = guess.copy()
    res0 = np.linalg.norm(A @ x - b)
    if res0 == 0:
        return x, 0, 0.0, np.array([0.0]), np.array([np.inf])
    it = 1
    rel_res = 1.0
    while rel_resk[it - 1] > retol and it <= maxit:                       # :42 (1-based it)
        r = b - A @ x
        x = x + it_(r)
        res = np.linalg.norm(A @ x - b)
        rel_res = res / res0
        rel_resk[it] = rel_res
        rhok[it] = res / np.linalg.norm(r)
        it += 1
        if rhok[it - 1] > 1:
            break
    return x, it - 1, rel_res, rel_resk[:it].copy(), rhok[:it].copy()



def twogrid(A, b, amg_options: Optional[dict], rng: np.random.RandomState):
    """``AMG/twogrid.m:1-85``: the two-level method for a general matrix (``bigph = 0``: Jacobi
    smoother ``.5*D^{-1}``, C/F split by ``mis_set(A,1/4)``, interpolation ``W1 + 0.5*W2`` -- the
    ``~isempty`` test of ``:58`` is always true) or a bigraph (``bigph = 1``: as twogrid_bigph)."""
    b = np.asarray(b, float)
    if amg_options is None:
        amg_options = dict(retol=1e-12, bigph=0, maxit=20, smoth=10, isnsp=1, guess=np.zeros_like(b))
    o = dict(amg_options)
    retol = 0 if o.get("retol") is None else o["retol"]                  # :11-16
    bigph = 0 if o.get("bigph") is None else int(o["bigph"])
    maxit = 50 if o.get("maxit") is None else int(o["maxit"])
    smoth = 3 if o.get("smoth") is None else int(o["smoth"])
    isnsp = 0 if o.get("isnsp") is None else int(o["isnsp"])
    guess = np.zeros_like(b) if o.get("guess") is None else np.asarray(o["guess"], float)
    fnode = 0 if o.get("fnode") is None else int(o["fnode"])
    if bigph and fnode <= 0:
        raise ValueError("bigph = 1 requires fnode > 0")                  # :24-26
    if bigph:
        return twogrid_bigph(A, b, dict(retol=retol, maxit=maxit, smoth=smoth, isnsp=isnsp,
                                        guess=guess, fnode=fnode))
    A = _csr(A)
    N = A.shape[0]
    R = _csr(0.5 * sp.diags(1.0 / A.diagonal()))                          # :40
    Ac, Pro, _ = transfer(A, dict(bigph=0, theta=1 / 4, isnsp=isnsp, inter=1), 2, rng)  # :50-66
    return _twogrid_loop(A, b, R, Pro, Ac, retol, maxit, smoth, isnsp, guess)


def Hybrid_twogrid(prob_data: dict, amg_options: dict, rng: np.random.RandomState,
                   trace: Optional[list] = None):
    """``Hybrid_twogrid.m``: `Hybrid_AMG` with `twogrid_bigph` on the connected system / the large
    components (``:39,63``); everything else (rescaling, isnsp rule, guesses, small blocks) is the
    same text."""
    return Hybrid_AMG(prob_data, amg_options, rng, trace=trace,
                      solver=lambda A, f, o: twogrid_bigph(A, f, o))


def aug_PCG(prob_data: dict, pcg_options: dict):
    """``aug_PCG.m:11-36``: PCG on the system augmented with the kernel vectors of A0 (one
    indicator vector per connected component)."""
    bk1, tk = prob_data["bk1"], prob_data["tk"]
    q, p = np.asarray(prob_data["q"], float), np.asarray(prob_data["p"], float)
    H0, z, T = prob_data["H0"], np.asarray(prob_data["z"], float), prob_data["T"]
    Ae, A0, Q, K, Q0, qp = build_Ae(H0, T, p, q, bk1, tk)
    f = qp * z
    M = len(qp)
    blocks, sizes, _, _ = components(A0)                                  # :24
    nc = len(sizes)
    Y = sp.csr_matrix((np.ones(M), (np.arange(M), blocks)), shape=(M, nc))  # :25
    QK = _csr(bk1 * Q + 1 / tk * K)                                       # :27
    augAe = sp.bmat([[Y.T @ QK @ Y, Y.T @ QK], [QK @ Y, Ae]], format="csr")  # :28
    augf = np.concatenate([Y.T @ f, f])
    o = dict(pcg_options)
    o["guess"] = np.zeros(nc + M)                                         # :29
    o["precd"] = 2                                                        # :32
    U, itpcg, respcg, _ = PCG(augAe, augf, o)
    u = Y @ U[:nc] + U[nc:]                                               # :35
    return qp * u, itpcg, respcg, np.array([nc, 1])


def PCG4POT(prob_data: dict, pcg_options: dict):
    """``Class2/PCG4POT.m:27-39``: the Sherman-Morrison reduction of `AMG4POT` with `aug_PCG`."""
    p, q = prob_data["p"], prob_data["q"]
    bk1, tk = prob_data["bk1"], prob_data["tk"]
    phi = np.asarray(prob_data["phi"], float)
    z = np.asarray(prob_data["z"], float)
    s = np.asarray(prob_data["s"], float)
    z1, z2 = z[:-1], z[-1]
    epss, sg = bk1, 1 / tk
    phi_e = epss + sg * (phi @ (s * phi))
    v = Ax(s * phi, p, q)
    w = z1 - sg / phi_e * z2 * v
    pd = dict(prob_data)
    pd["z"] = v
    vv, it1, res1, info1 = aug_PCG(pd, pcg_options)
    pd["z"] = w
    ww, it2, res2, info2 = aug_PCG(pd, pcg_options)
    tt = sg ** 2 / (phi_e - sg ** 2 * (v @ vv))
    zeta1 = ww + tt * vv * (v @ ww)
    zeta2 = (z2 - sg * (v @ zeta1)) / phi_e
    return (np.concatenate([zeta1, [zeta2]]), max(it1, it2), max(res1, res2),
            np.maximum(info1, info2))


def AMG4POT(prob_data: dict, amg_options: dict, rng: np.random.RandomState,
            str_: str = "amg", trace: Optional[list] = None):
    """``Class2/AMG4POT.m:27-55``: bordered partial-OT system through
    Sherman-Morrison and two ``Hybrid_AMG`` solves on the same ``Ae``."""
    if str_ not in ("amg", "twogrid"):
        raise ValueError("AMG4POT: str must be 'amg' or 'twogrid'")
    hybrid = Hybrid_AMG if str_ == "amg" else Hybrid_twogrid               # :44-51
    p, q = prob_data["p"], prob_data["q"]
    bk1, tk = prob_data["bk1"], prob_data["tk"]
    phi = np.asarray(prob_data["phi"], float)
    z = np.asarray(prob_data["z"], float)
    s = np.asarray(prob_data["s"], float)
    z1, z2 = z[:-1], z[-1]
    epss, sg = bk1, 1 / tk
    phi_e = epss + sg * (phi @ (s * phi))                                # :33
    v = Ax(s * phi, p, q)                                                # :34
    w = z1 - sg / phi_e * z2 * v
    pd = dict(prob_data)
    pd["z"] = v
    vv, it1, res1, info1 = hybrid(pd, amg_options, rng, trace=trace)      # :46
    pd["z"] = w
    ww, it2, res2, info2 = hybrid(pd, amg_options, rng, trace=trace)      # :47
    tt = sg ** 2 / (phi_e - sg ** 2 * (v @ vv))                          # :53
    # :54 reads `ww + tt*vv*v'*ww`, which MATLAB evaluates left to right: the M x M outer product
    # (tt*vv)*v' times ww through a dense BLAS gemv whose summation order is the library's
    # (unpinnable).  The associative form below differs from it by rounding only (~1e-16
    # relative, far inside the 1e-10 bar); it is the one deliberate departure from the oracle's
    # "reproduce MATLAB's order" rule.
    zeta1 = ww + tt * vv * (v @ ww)
    zeta2 = (z2 - sg * (v @ zeta1)) / phi_e
    zeta = np.concatenate([zeta1, [zeta2]])
    return zeta, max(it1, it2), max(res1, res2), np.maximum(info1, info2)
