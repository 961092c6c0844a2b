"""GPU parity of the AMG setup: strength, cf_split, mis_set, transfer, hierarchy.
Bar: BIT-EXACT against the oracle (C/F masks, Pro, Ac on every level)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ipd_oracle as O
from tests import problems as PR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def csc_equal(A, B):
    A = sp.csc_matrix(A); B = sp.csc_matrix(B)
    A.sort_indices(); B.sort_indices()
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr)
            and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data))


def newton_matrix(m, n, s, t=None):
    pd = PR.make_prob(m, n, s, t=t)
    H0 = O.ASAt(s, pd["p"], pd["q"])
    Ae = O.build_Ae(H0, pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    return Ae, pd


@pytest.mark.parametrize("N,deg,seed", [(1, 1, 0), (17, 2, 1), (200, 3, 2), (777, 5, 3)])
@pytest.mark.parametrize("which", [1, 2])
def test_strength(ipd, N, deg, seed, which):
    A = PR.random_sym_graph_laplacian(N, deg=deg, seed=seed)
    assert csc_equal(ipd.strength(A, which), O.strength(A, which))


@pytest.mark.parametrize("N,deg,seed", [(1, 1, 0), (40, 2, 1), (500, 3, 2), (2048, 4, 3)])
def test_cf_split(ipd, N, deg, seed):
    A = PR.random_sym_graph_laplacian(N, deg=deg, seed=seed)
    As = O.strength_mask(A, 0.25)
    S = sp.csr_matrix(((As + As.T) > 0).astype(float))
    refC, refF = O.cf_split(S)
    gotC, gotF = ipd.cf_split(S)
    assert np.array_equal(gotC, refC) and np.array_equal(gotF, refF)


def test_cf_split_path_graph_worst_case(ipd):
    N = 300     # a path needs ~N/2 dependent rounds
    S = sp.diags([np.ones(N - 1), np.ones(N - 1)], [-1, 1], format="csr")
    gotC, gotF = ipd.cf_split(S)
    assert np.array_equal(gotC, np.arange(N) % 2 == 0) and np.array_equal(gotF, ~gotC)


@pytest.mark.parametrize("N,deg,seed", [(30, 2, 1), (400, 3, 2), (1500, 6, 3)])
def test_mis_set(ipd, N, deg, seed):
    A = PR.random_sym_graph_laplacian(N, deg=deg, seed=seed)
    refC, refF, refAs, info = O.mis_set(A, 0.25, O.matlab_rng())
    rng = ipd.MatlabRand()
    gotC, gotF, gotAs = ipd.mis_set(A, 0.25, rng)
    assert np.array_equal(gotC, refC) and np.array_equal(gotF, refF)
    assert csc_equal(gotAs, refAs)
    assert rng.consumed == len(info["rand"])


def test_mis_set_degenerate_branch(ipd):
    N = 100   # diagonal matrix: no strong connections -> random N0 picks (mis_set.m:30-34)
    A = sp.diags(np.arange(1.0, N + 1), format="csr")
    refC, refF, _, info = O.mis_set(A, 0.25, O.matlab_rng())
    assert info["branch"] == "degenerate"
    rng = ipd.MatlabRand()
    gotC, gotF, _ = ipd.mis_set(A, 0.25, rng)
    assert np.array_equal(gotC, refC) and np.array_equal(gotF, refF)
    assert rng.consumed == 11


CASES = [
    ("tree64", 64, 64, lambda: PR.mask_tree(64, 64, seed=1), None),
    ("tree_rect", 150, 90, lambda: PR.mask_tree(150, 90, seed=2), None),
    ("bern256", 256, 256, lambda: PR.mask_bernoulli(256, 256, 0.02, seed=3), None),
    ("dense96", 96, 96, lambda: PR.mask_bernoulli(96, 96, 1.0), None),
    ("half128", 128, 128, lambda: PR.mask_bernoulli(128, 128, 0.5), None),
]


@pytest.mark.parametrize("name,m,n,mk,t", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("isnsp", [0, 1])
def test_hierarchy_bit_exact(ipd, name, m, n, mk, t, isnsp):
    s = mk()
    Ae, pd = newton_matrix(m, n, s)
    ncomp = sp.csgraph.connected_components(Ae)[0]
    if ncomp > 1:   # Class_AMG is only called on connected components
        lab = sp.csgraph.connected_components(Ae)[1]
        big = np.argmax(np.bincount(lab))
        pk = np.flatnonzero(lab == big)
        Ae = sp.csr_matrix(Ae[pk, :][:, pk])
        fnode = int((pk < n).sum())
    else:
        fnode = n
    o = O.amg_options_class1("v"); o.update(fnode=fnode, isnsp=isnsp)
    ho = O.amg_setup(Ae, o, O.matlab_rng())
    rng = ipd.MatlabRand()
    h = ipd.AMGHierarchy(Ae, o, rng)
    assert h.J == ho.J
    assert h.level_sizes() == ho.level_sizes()
    for k in range(1, ho.J + 1):
        assert csc_equal(h.A(k), ho.Ack[k]), f"Ack{{{k}}} differs"
    for k in range(2, ho.J + 1):
        assert csc_equal(h.P(k), ho.Prok[k]), f"Prok{{{k}}} differs"
        assert np.array_equal(h.cmask(k), ho.info[k]["isC"])
    assert rng.consumed == sum(len(i["mis"]["rand"]) for i in ho.info[2:] if i and i.get("mis"))
    h.close()


def test_transfer_standalone(ipd):
    A = PR.random_sym_graph_laplacian(300, deg=3, seed=5)
    o = O.amg_options_class1("v"); o.update(bigph=0, isnsp=0)
    Ac, Pro, info = O.transfer(A, o, 2, O.matlab_rng())
    gAc, gPro, gC = ipd.transfer(A, o, 2, ipd.MatlabRand())
    assert csc_equal(gAc, Ac) and csc_equal(gPro, Pro) and np.array_equal(gC, info["isC"])


@pytest.mark.parametrize("N,deg,seed", [(33, 2, 1), (300, 3, 5), (777, 5, 3), (1024, 4, 8)])
@pytest.mark.parametrize("isnsp", [0, 1])
def test_transfer_small_level_paths(ipd, N, deg, seed, isnsp):
    """`transfer` on levels of at most 1024 rows goes through the one-launch mis_set (k_mis_small): the
    split, Pro, Ac and the NUMBER of random numbers consumed against the oracle, with the launch-per-step
    form (IPD_NO_MIS_SMALL=1) beside it."""
    import os
    A = PR.random_sym_graph_laplacian(N, deg=deg, seed=seed)
    o = O.amg_options_class1("v"); o.update(bigph=0, isnsp=isnsp)
    rr = O.matlab_rng()
    Ac, Pro, info = O.transfer(A, o, 2, rr)
    used = len(info["mis"]["rand"])
    for kv in ({}, {"IPD_NO_MIS_SMALL": "1"}):
        os.environ.update(kv)
        try:
            rng = ipd.MatlabRand()
            gAc, gPro, gC = ipd.transfer(A, o, 2, rng)
        finally:
            for k in kv:
                os.environ.pop(k)
        assert csc_equal(gAc, Ac) and csc_equal(gPro, Pro) and np.array_equal(gC, info["isC"]), kv
        assert rng.consumed == used, (kv, rng.consumed, used)


def test_transfer_small_level_degenerate_and_replay(ipd):
    """(a) hardly any strong connection (mis_set.m:30-34): the one-launch form reports it and the host
    takes the branch; (b) a replay stream that holds exactly the numbers the level consumes -- fewer than
    the N the one-launch form peeks at."""
    N = 100
    A = sp.diags(np.arange(1.0, N + 1), format="csr") + 1e-9 * PR.random_sym_graph_laplacian(N, deg=2, seed=1)
    A = sp.csr_matrix(A)
    o = O.amg_options_class1("v"); o.update(bigph=0, isnsp=0)
    refC, refF, _, info = O.mis_set(A, 0.25, O.matlab_rng())
    rng = ipd.MatlabRand()
    if info["branch"] == "degenerate":
        try:
            gAc, gPro, gC = ipd.transfer(A, o, 2, rng)
            assert np.array_equal(gC, refC) and rng.consumed == len(info["rand"])
        except RuntimeError as exc:      # the reference itself may index out of range here (SURVEY A-6)
            assert "coarse" in str(exc) or "mis_set" in str(exc)
    B = PR.random_sym_graph_laplacian(400, deg=3, seed=2)
    rr = O.matlab_rng()
    Ac, Pro, info = O.transfer(B, o, 2, rr)
    vals = np.asarray(info["mis"]["rand"], float)
    got = ipd.transfer(B, o, 2, ipd.MatlabRand(replay=vals))
    assert csc_equal(got[0], Ac) and csc_equal(got[1], Pro)


@pytest.mark.parametrize("isnsp", [0, 1])
def test_transfer_ideal_interpolation(ipd, isnsp):
    """`inter = 2`: W = -Aff \\ Afc (AMG/transfer.m:57-58).  MATLAB and the oracle (SuperLU) solve
    with sparse factorisations, the device with a dense Cholesky of Aff: the C/F split is the same
    bits (it does not depend on `inter`), Pro and Ac agree to rounding, A_ff*W + A_fc = 0."""
    A = PR.random_sym_graph_laplacian(260, deg=3, seed=7) + sp.identity(260) * 0.05
    o = O.amg_options_class1("v"); o.update(bigph=0, isnsp=isnsp, inter=2)
    Ac, Pro, info = O.transfer(A, o, 2, O.matlab_rng())
    gAc, gPro, gC = ipd.transfer(A, o, 2, ipd.MatlabRand())
    assert np.array_equal(gC, info["isC"])
    assert gPro.shape == Pro.shape and gAc.shape == Ac.shape
    assert abs(gPro - Pro).max() <= 1e-11 * abs(Pro).max()
    assert abs(gAc - Ac).max() <= 1e-10 * abs(Ac).max()
    if isnsp == 0:      # the defining equation of the ideal interpolation, on the device result
        isC = info["isC"]
        F, C = np.flatnonzero(~isC), np.flatnonzero(isC)
        Ad = A.toarray()
        W = gPro.toarray()[F, :]
        assert np.abs(Ad[np.ix_(F, F)] @ W + Ad[np.ix_(F, C)]).max() <= 1e-12 * np.abs(Ad).max() * max(1.0, np.abs(W).max())
    else:
        assert np.allclose(gPro.toarray().sum(axis=1), 1.0, atol=1e-12)


def test_class_amg_with_ideal_interpolation(ipd):
    """Class_AMG with inter = 2 on the non-bigraph path: same level sizes as the oracle, same
    iteration count (+-1), residual history to 1e-8."""
    A = PR.random_sym_graph_laplacian(400, deg=4, seed=3) + sp.identity(400) * 0.02
    b = np.random.RandomState(1).standard_normal(400)
    o = dict(retol=1e-10, bigph=0, maxit=40, theta=0.25, smoth=2, cycle="v", isnsp=0, inter=2)
    xr, itr, rrr, relk, rhok = O.Class_AMG(A, b, dict(o, guess=None), O.matlab_rng())
    x, it, rr, relkd, _ = ipd.Class_AMG(A, b, dict(o, guess=None), ipd.MatlabRand())
    assert abs(it - itr) <= 1
    k = min(len(relk), len(relkd))
    assert np.all(np.abs(np.asarray(relkd[:k]) - np.asarray(relk[:k])) <= 1e-8 + 1e-5 * np.asarray(relk[:k]))
    assert np.linalg.norm(A @ x - b) <= 1e-9 * np.linalg.norm(b)


def test_setup_errors(ipd):
    A = PR.random_sym_graph_laplacian(50, seed=1)
    with pytest.raises(ipd.IpdError) as ei:
        ipd.AMGHierarchy(A, dict(bigph=1, fnode=None, smoth=1))
    assert "requires Nf > 0" in str(ei.value)       # Class_AMG.m:36-40


@pytest.mark.parametrize("name,m,n,mk,t", [c for c in CASES if c[0] in ("dense96", "half128")],
                         ids=["dense96", "half128"])
def test_hierarchy_bit_exact_tile_product(ipd, monkeypatch, name, m, n, mk, t):
    """Galerkin products through the dense-tile kernel (csrc/ipd_sparse.hip k_gemm_ordered),
    forced here at sizes the oracle can follow: same bits as the oracle's ordered product."""
    monkeypatch.setenv("IPD_PRODUCT", "tiles")
    monkeypatch.setenv("IPD_INTERP", "split")   # interpolation rows through the product too
    Ae, pd = newton_matrix(m, n, mk())
    o = O.amg_options_class1("v"); o.update(fnode=n, isnsp=1)
    ho = O.amg_setup(Ae, o, O.matlab_rng())
    h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    assert h.level_sizes() == ho.level_sizes()
    for k in range(1, ho.J + 1):
        assert csc_equal(h.A(k), ho.Ack[k]), f"Ack{{{k}}} differs"
    for k in range(2, ho.J + 1):
        assert csc_equal(h.P(k), ho.Prok[k]), f"Prok{{{k}}} differs"
    h.close()


@pytest.mark.parametrize("name,m,n,mk,t", CASES, ids=[c[0] for c in CASES])
def test_hierarchy_bit_exact_split_interpolation(ipd, monkeypatch, name, m, n, mk, t):
    """The interpolation build in product form (csrc/ipd_setup.hip k_w_split_*; default only for
    long rows) on every setup case, sparse ones included, with the row product kernel."""
    monkeypatch.setenv("IPD_INTERP", "split")
    monkeypatch.setenv("IPD_PRODUCT", "rows")
    s = mk()
    Ae, pd = newton_matrix(m, n, s)
    lab = sp.csgraph.connected_components(Ae)[1]
    pk = np.flatnonzero(lab == np.argmax(np.bincount(lab)))
    Ae = sp.csr_matrix(Ae[pk, :][:, pk])
    o = O.amg_options_class1("v"); o.update(fnode=int((pk < n).sum()), isnsp=1)
    ho = O.amg_setup(Ae, o, O.matlab_rng())
    h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    assert h.level_sizes() == ho.level_sizes()
    for k in range(2, ho.J + 1):
        assert csc_equal(h.A(k), ho.Ack[k]), f"Ack{{{k}}} differs"
        assert csc_equal(h.P(k), ho.Prok[k]), f"Prok{{{k}}} differs"
    h.close()


@pytest.mark.parametrize("name,m,n,mk,t", CASES, ids=[c[0] for c in CASES])
def test_hierarchy_bit_exact_block_interpolation(ipd, monkeypatch, name, m, n, mk, t):
    """The interpolation build by k_build_W (a workgroup per row, a barrier per strong F neighbour: the form
    rows of 96 entries and more still take) forced on every setup case; the default for shorter rows is the
    pipelined one-wave kernel k_build_W_w, which the unforced tests above compare with the oracle."""
    monkeypatch.setenv("IPD_INTERP", "block")
    s = mk()
    Ae, pd = newton_matrix(m, n, s)
    lab = sp.csgraph.connected_components(Ae)[1]
    pk = np.flatnonzero(lab == np.argmax(np.bincount(lab)))
    Ae = sp.csr_matrix(Ae[pk, :][:, pk])
    o = O.amg_options_class1("v"); o.update(fnode=int((pk < n).sum()), isnsp=1)
    ho = O.amg_setup(Ae, o, O.matlab_rng())
    h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    assert h.level_sizes() == ho.level_sizes()
    for k in range(2, ho.J + 1):
        assert csc_equal(h.A(k), ho.Ack[k]), f"Ack{{{k}}} differs"
        assert csc_equal(h.P(k), ho.Prok[k]), f"Prok{{{k}}} differs"
    h.close()


def test_first_and_later_hierarchies_of_a_context(ipd):
    """The first hierarchy a context builds fetches every entry count as it goes (the counting launches' tails post
    them); from the second on the counts of a level stay on the device until its last compaction posts them, the
    arrays are sized by dense bounds and the consumers scan the row counts themselves (csrc/ipd_setup.hip "lazy
    counts", ipd_internal.h scan_head / ScanTail).  Same bits either way, on a fresh context so that the first
    build really is one."""
    from codes_of_ipd_ssn_amg_method_amd import _lib
    m, n = 300, 260
    Ae, pd = newton_matrix(m, n, PR.mask_bernoulli(m, n, 0.03, seed=9))
    lab = sp.csgraph.connected_components(Ae)[1]
    pk = np.flatnonzero(lab == np.argmax(np.bincount(lab)))
    Ae = sp.csr_matrix(Ae[pk, :][:, pk])
    o = O.amg_options_class1("w"); o.update(fnode=int((pk < n).sum()), isnsp=1)
    ho = O.amg_setup(Ae, o, O.matlab_rng())
    ctx = _lib.Context(0)
    for build in range(3):
        h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand(), ctx=ctx)
        assert h.level_sizes() == ho.level_sizes(), build
        for k in range(2, ho.J + 1):
            assert csc_equal(h.A(k), ho.Ack[k]), (build, k)
            assert csc_equal(h.P(k), ho.Prok[k]), (build, k)
        h.close()


@pytest.mark.parametrize("rho", [1.0, 0.4])
def test_tile_product_matches_row_product(ipd, monkeypatch, rho):
    """At a size where the tile kernel is the default choice, both product kernels give the same
    hierarchy bit for bit (ragged sizes: the zero padding of the dense operands is exercised)."""
    m, n = 530, 470
    Ae, pd = newton_matrix(m, n, PR.mask_bernoulli(m, n, rho, seed=4))
    o = O.amg_options_class1("v"); o.update(fnode=n, isnsp=1)
    hs = []
    for kind in ("rows", "tiles"):
        monkeypatch.setenv("IPD_PRODUCT", kind)
        monkeypatch.setenv("IPD_INTERP", "single" if kind == "rows" else "split")
        hs.append(ipd.AMGHierarchy(Ae, o, ipd.MatlabRand()))
    a, b = hs
    assert a.level_sizes() == b.level_sizes()
    for k in range(2, a.J + 1):
        assert csc_equal(a.A(k), b.A(k)), f"level {k}"
        assert csc_equal(a.P(k), b.P(k)), f"level {k}"
    a.close(); b.close()
