"""Pins the CPU oracle (oracle/ipd_oracle.py) through algebraic identities and
known answers -- the reference ships no golden vectors, so the
oracle is "parity unpinned" w.r.t. real MATLAB output; these are what pin it."""
import math

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import ipd_oracle as O
from tests import problems as PR


def test_matlab_rand_stream():
    # MATLAB: rng('default'); rand(5,1) -> 0.8147 0.9058 0.1270 0.9134 0.6324
    v = O.matlab_rng().random_sample(5)
    assert np.allclose(v, [0.8147, 0.9058, 0.1270, 0.9134, 0.6324], atol=5e-5)


@pytest.mark.parametrize("m,n", [(7, 5), (16, 16), (33, 20)])
def test_ax_aty_against_explicit_A(m, n):
    rs = np.random.RandomState(0)
    p, q = 0.5 + rs.random_sample(m), 0.5 + rs.random_sample(n)
    A = O.build_A(p, q)
    x = rs.randn(m * n)
    y = rs.randn(m + n)
    assert np.allclose(O.Ax(x, p, q), A @ x, rtol=1e-13, atol=1e-13)
    assert np.allclose(O.Aty(y, p, q), A.T @ y, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("m,n,rho", [(6, 9, 0.3), (16, 16, 0.05), (20, 13, 1.0), (8, 8, 0.0)])
def test_asat_is_A_diag_s_At(m, n, rho):
    rs = np.random.RandomState(1)
    p, q = 0.5 + rs.random_sample(m), 0.5 + rs.random_sample(n)
    s = PR.mask_bernoulli(m, n, rho, seed=5)
    H = O.ASAt(s, p, q)
    A = O.build_A(p, q)
    Href = A @ sp.diags(s.astype(float)) @ A.T
    assert abs(H - Href).max() <= 1e-13 * max(1.0, abs(Href).max())
    assert (H != H.T).nnz == 0
    assert H.nnz == (H != 0).sum()          # no explicit zeros
    # graph-Laplacian structure after the Q0 scaling of Hybrid_AMG.m:23
    qp = np.concatenate([q, -p])
    A0 = sp.diags(qp) @ H @ sp.diags(qp)
    assert np.allclose(A0 @ np.ones(m + n), 0, atol=1e-12)
    off = A0 - sp.diags(A0.diagonal())
    assert off.nnz == 0 or off.max() <= 0


def test_inv_aat_hht_residuals():
    rs = np.random.RandomState(2)
    m, n = 9, 7
    p, q = 0.5 + rs.random_sample(m), 0.5 + rs.random_sample(n)
    A = O.build_A(p, q).toarray()
    x = rs.randn(m + n)
    sg1, sg2 = 0.7, 1.9
    y = O.invAAt(x, p, q, sg1, sg2)
    Mx = np.diag(np.r_[sg1 * np.ones(n), sg2 * np.ones(m)]) + A @ A.T
    assert np.allclose(Mx @ y, x, atol=1e-12)
    phi = rs.random_sample(m * n)
    sg = 0.3
    # H = (G, IY, IZ) with G = [A; phi'] and IY, IZ the slack identities of the n column and m
    # row constraints: sg*I + H*H' = blkdiag((sg+1) I_{n+m}, sg) + G*G'   (Class2/invHHt.m:8-9)
    G = np.vstack([A, phi[None, :]])
    Mh = np.diag(np.r_[(sg + 1) * np.ones(m + n), sg]) + G @ G.T
    v = rs.randn(m + n + 1)
    yv = O.invHHt(v, p, q, sg, phi)
    assert np.allclose(Mh @ yv, v, atol=1e-10)


def _py_spgemm(X, Y):
    """Reference Gustavson product: ascending inner index, separate mul/add."""
    X, Y = sp.csr_matrix(X), sp.csr_matrix(Y)
    X.sort_indices(); Y.sort_indices()
    out = {}
    for i in range(X.shape[0]):
        for e in range(X.indptr[i], X.indptr[i + 1]):
            k, a = X.indices[e], X.data[e]
            for t in range(Y.indptr[k], Y.indptr[k + 1]):
                prod = a * Y.data[t]
                key = (i, Y.indices[t])
                out[key] = out.get(key, 0.0) + prod
    return out


def test_spgemm_order():
    """The oracle relies on SciPy's csr_matmat accumulating in ascending inner
    index without FMA; check bit-for-bit against a pure-Python loop."""
    rs = np.random.RandomState(3)
    X = sp.random(40, 30, 0.3, random_state=rs, format="csr") * 1.37
    Y = sp.random(30, 25, 0.3, random_state=rs, format="csr") * 0.91
    Z = O._spgemm(X, Y).tocoo()
    ref = _py_spgemm(X, Y)
    got = {(int(i), int(j)): v for i, j, v in zip(Z.row, Z.col, Z.data)}
    ref = {k: v for k, v in ref.items() if v != 0.0}
    assert got.keys() == ref.keys()
    assert all(got[k] == ref[k] for k in ref)          # bitwise


def test_csr_matvec_is_sequential():
    rs = np.random.RandomState(4)
    M = sp.random(20, 50, 0.6, random_state=rs, format="csr")
    x = rs.randn(50)
    y = M @ x
    for i in range(20):
        acc = 0.0
        for e in range(M.indptr[i], M.indptr[i + 1]):
            acc = acc + M.data[e] * x[M.indices[e]]
        assert acc == y[i]


def test_strength_definition():
    A = PR.random_sym_graph_laplacian(30, seed=5)
    S = O.strength(A).toarray()
    Ad = A.toarray()
    off = -(Ad - np.diag(np.diag(Ad)))
    mr = np.maximum(off.max(axis=1), 0)
    mr[mr <= 0] = np.inf
    ref = off / np.minimum.outer(mr, mr)
    np.fill_diagonal(ref, 0)
    assert np.array_equal(S, ref)


def test_cf_split_is_lexicographic_mis():
    A = PR.random_sym_graph_laplacian(60, deg=2, seed=6)
    As = O.strength_mask(A, 0.25)
    As = ((As + As.T) > 0).astype(float)          # graph(S) needs symmetry
    indC, indF = O.cf_split(sp.csr_matrix(As))
    assert not np.any(indC & indF) and np.all(indC | indF)
    Ad = As.toarray() > 0
    for k in range(60):
        lower_c = np.any(indC[:k] & Ad[k, :k])
        assert indC[k] == (not lower_c)


def test_level1_smoother_is_gauss_seidel_inverse():
    """Rk{1} == inv(tril(Ae)) and Rk{1}' == inv(triu(Ae)) in [F|C] order."""
    m = n = 12
    s = PR.mask_tree(m, n, seed=7)
    pd = PR.make_prob(m, n, s)
    H0 = O.ASAt(s, pd["p"], pd["q"])
    Ae = O.build_Ae(H0, pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    o = O.amg_options_class1("v"); o["fnode"] = n; o["isnsp"] = 1
    h = O.amg_setup(Ae, o, O.matlab_rng())
    R = h.Rk[1].toarray()
    assert np.allclose(R @ np.tril(Ae.toarray()), np.eye(m + n), atol=1e-10)
    assert np.allclose(R.T @ np.triu(Ae.toarray()), np.eye(m + n), atol=1e-10)


@pytest.mark.parametrize("isnsp", [0, 1])
def test_transfer_properties(isnsp):
    m = n = 24
    s = PR.mask_tree(m, n, seed=8)
    pd = PR.make_prob(m, n, s)
    H0 = O.ASAt(s, pd["p"], pd["q"])
    Ae = O.build_Ae(H0, pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    o = O.amg_options_class1("v"); o["fnode"] = n; o["isnsp"] = isnsp
    h = O.amg_setup(Ae, o, O.matlab_rng())
    assert h.J >= 3
    assert h.level_sizes()[-1] <= O.coarsest_threshold(m + n)
    for k in range(2, h.J + 1):
        P, Af, Ac = h.Prok[k], h.Ack[k - 1], h.Ack[k]
        if isnsp:
            assert np.allclose(P @ np.ones(P.shape[1]), 1.0, atol=1e-12)   # row-normalised
        assert abs(Ac - Ac.T).max() <= 1e-9 * abs(Ac).max()
        if isnsp:   # constants are preserved => 1'Ac1 == 1'A1
            assert math.isclose(Ac.sum(), Af.sum(), rel_tol=1e-9)
        isC = h.info[k]["isC"]
        assert P.shape == (Af.shape[0], int(isC.sum()))
        assert np.array_equal(P[np.flatnonzero(isC), :].toarray(), np.eye(int(isC.sum())))


def test_coarsest_threshold_quirk():
    # SURVEY A-2: evaluated in floating point (512^(1/3) = 7.99..)
    assert [O.coarsest_threshold(N) for N in (512, 1000, 1024, 2048, 4096)] == [8, 10, 11, 13, 16]


def test_components_vs_scipy():
    rs = np.random.RandomState(9)
    G = sp.random(80, 80, 0.015, random_state=rs, format="csr")
    G = G + G.T + sp.identity(80)
    blocks, sizes, p, r = O.components(G)
    nc, lab = sp.csgraph.connected_components(G, directed=False)
    assert len(sizes) == nc
    assert np.array_equal(blocks, lab)
    for k in range(nc):
        members = p[r[k]:r[k + 1]]
        assert np.array_equal(np.sort(members), np.flatnonzero(lab == k))
        assert np.all(np.diff(members) > 0)


def test_pcg_solves_spd():
    A = PR.random_sym_graph_laplacian(40, seed=10, eps=0.5)
    b = np.random.RandomState(11).randn(40)
    for precd in (1, 2, 3):
        d, it, res, resk = O.PCG(A, b, dict(guess=None, retol=1e-11, maxit=1000, precd=precd))
        assert np.allclose(A @ d, b, atol=1e-8)
        assert it == len(resk) and res <= 1e-11
    d, it, res, _ = O.PCG(A, np.zeros(40))
    assert it == 0 and math.isnan(res) and not d.any()   # r=0 -> res=NaN


@pytest.mark.parametrize("cycle", ["v", "w"])
@pytest.mark.parametrize("kind", ["tree", "bern"])
def test_hybrid_amg_matches_direct(cycle, kind):
    m = n = 48
    s = PR.mask_tree(m, n, seed=12, connect=False) if kind == "tree" else PR.mask_bernoulli(m, n, 0.08)
    pd = PR.make_prob(m, n, s)
    pd["H0"] = O.ASAt(s, pd["p"], pd["q"])
    tr = []
    zeta, it, res, info = O.Hybrid_AMG(pd, O.amg_options_class1(cycle), O.matlab_rng(), trace=tr)
    Jk = pd["bk1"] * sp.identity(m + n) + (pd["T"] + pd["H0"]) / pd["tk"]
    zd = spla.spsolve(sp.csc_matrix(Jk), pd["z"])
    assert np.linalg.norm(Jk @ zeta - pd["z"]) <= 1e-8 * np.linalg.norm(pd["z"])
    assert np.linalg.norm(zeta - zd) <= 1e-6 * np.linalg.norm(zd)
    for t in tr:
        assert t["it"] < 30 and np.all(t["rhok"][1:] < 1.0)
        assert np.all(np.diff(t["rel_resk"]) < 0)           # monotone residual history


def test_amg4pot_matches_bordered_direct():
    m = n = 32
    rs = np.random.RandomState(13)
    s = PR.mask_bernoulli(m, n, 0.1, seed=14)
    t = (rs.random_sample(m + n) < 0.7).astype(float)
    pd = PR.make_prob(m, n, s, t=t)
    pd["z"] = rs.randn(m + n + 1)
    pd["phi"] = np.ones(m * n)
    pd["H0"] = O.ASAt(s, pd["p"], pd["q"])
    zeta, it, res, info = O.AMG4POT(pd, O.amg_options_class2("w"), O.matlab_rng())
    A = O.build_A(pd["p"], pd["q"])
    S = sp.diags(s.astype(float))
    sg, eps = 1 / pd["tk"], pd["bk1"]
    v = A @ (S @ pd["phi"])
    Hfull = sp.bmat([[pd["T"] + A @ S @ A.T, sp.csr_matrix(v[:, None])],
                     [sp.csr_matrix(v[None, :]), sp.csr_matrix([[pd["phi"] @ (S @ pd["phi"])]])]])
    He = eps * sp.identity(m + n + 1) + sg * Hfull
    ref = spla.spsolve(sp.csc_matrix(He), pd["z"])
    assert np.linalg.norm(zeta - ref) <= 1e-6 * np.linalg.norm(ref)


def test_ichol0_is_the_cholesky_factor_when_nothing_is_dropped():
    """IC(0) (oracle `ichol0`, the restatement of MATLAB's default `ichol`, PCG.m:46) equals the
    complete Cholesky factor when tril(H) already holds all of its fill: a dense SPD matrix and an
    arrowhead matrix; on a general sparse SPD matrix L*L' matches H on H's pattern."""
    import scipy.sparse as sp
    rs = np.random.RandomState(3)
    G = rs.standard_normal((12, 12))
    Hd = G @ G.T + 12 * np.eye(12)
    L = O.ichol0(sp.csr_matrix(Hd)).toarray()
    assert np.allclose(L, np.linalg.cholesky(Hd), rtol=1e-12, atol=1e-12)
    n = 9
    Ha = np.diag(2.0 + rs.random_sample(n))
    Ha[-1, :-1] = Ha[:-1, -1] = 0.3 * rs.random_sample(n - 1)
    Ha[-1, -1] = 4.0
    assert np.allclose(O.ichol0(sp.csr_matrix(Ha)).toarray(), np.linalg.cholesky(Ha), atol=1e-13)
    S = sp.random(40, 40, density=0.08, random_state=rs, format="csr")
    Hs = (S + S.T + sp.identity(40) * 6.0).tocsr()
    Ls = O.ichol0(Hs)
    R = (Ls @ Ls.T - Hs).toarray()
    mask = Hs.toarray() != 0
    assert np.abs(R[mask]).max() <= 1e-13 * np.abs(Hs).max()
    with pytest.raises(ValueError):
        O.ichol0(sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 1.0]])))


def test_ideal_interpolation_identity_and_single_coarse_node():
    """`inter = 2`, W = -Aff \\ Afc (AMG/transfer.m:57-58): A_ff*W + A_fc = 0 on the oracle's result, and a
    hierarchy whose last transfer has ONE coarse node (SciPy's spsolve returns a 1-D array there)
    still builds and converges."""
    import scipy.sparse as sp
    A = PR.random_sym_graph_laplacian(200, deg=3, seed=7) + sp.identity(200) * 0.05
    o = O.amg_options_class1("v")
    o.update(bigph=0, isnsp=0, inter=2)
    Ac, Pro, info = O.transfer(A, o, 2, O.matlab_rng())
    isC = info["isC"]
    F, C = np.flatnonzero(~isC), np.flatnonzero(isC)
    Ad = A.toarray()
    W = Pro.toarray()[F, :]
    assert np.abs(Ad[np.ix_(F, F)] @ W + Ad[np.ix_(F, C)]).max() <= 1e-12 * np.abs(Ad).max() * max(1.0, np.abs(W).max())
    assert np.array_equal(Pro.toarray()[C, :], np.eye(len(C)))
    A2 = PR.random_sym_graph_laplacian(400, deg=4, seed=3) + sp.identity(400) * 0.02
    b = np.random.RandomState(1).standard_normal(400)
    opts = dict(retol=1e-10, bigph=0, maxit=40, theta=0.25, smoth=2, cycle="v", isnsp=0, inter=2, guess=None)
    x, it, rr, relk, rhok, h = O.Class_AMG(A2, b, opts, O.matlab_rng(), return_hierarchy=True)
    assert h.level_sizes()[-1] == 1 and rr <= 1e-10 and np.linalg.norm(A2 @ x - b) <= 1e-9 * np.linalg.norm(b)
