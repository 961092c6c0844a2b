"""GPU parity of the solve phase: MG_Vcycle / MG_Wcycle / PCG / Class_AMG.
Bar (north_star): residual-per-cycle within 1e-10 of the oracle; iteration
counts identical.  Solution vectors are compared through residuals (the systems
are nearly singular, SURVEY section 7 hard part iii)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ipd_oracle as O
from tests import problems as PR
from tests.test_gpu_setup import newton_matrix

pytestmark = pytest.mark.gpu

RES_TOL = 1e-10


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def test_pcg(ipd):
    A = PR.random_sym_graph_laplacian(200, seed=10, eps=0.5)
    b = np.random.RandomState(11).randn(200)
    for precd in (1, 2):
        o = dict(guess=None, retol=1e-11, maxit=1000, precd=precd)
        d, it, res, resk = ipd.PCG(A, b, o)
        do, ito, reso, resko = O.PCG(A, b, o)
        assert abs(it - ito) <= 1
        assert np.linalg.norm(A @ d - b) <= 1e-9 * np.linalg.norm(b)
        assert np.allclose(d, do, rtol=1e-8, atol=1e-10)
        k = min(it, ito) - 2
        assert np.allclose(resk[:k], resko[:k], rtol=1e-6)
    d, it, res, _ = ipd.PCG(A, np.zeros(200))
    assert it == 0 and np.isnan(res) and not d.any()
    with pytest.raises(ipd.IpdError):
        ipd.PCG(A, b, dict(precd=6))


def test_pcg_incomplete_cholesky_preconditioner(ipd):
    """precd 4: P = ichol(H) with MATLAB's defaults = IC(0) (PCG.m:44-50,100-101), built and applied
    on the device, against the oracle's restatement (`ichol0`): iteration counts, solution, residual
    history; on a dense matrix IC(0) is the exact factor and PCG stops after one step; an
    indefinite matrix raises like MATLAB's "nonpositive pivot"."""
    import scipy.sparse as sp
    A = PR.random_sym_graph_laplacian(300, seed=10, eps=0.5)
    b = np.random.RandomState(11).randn(300)
    o = dict(guess=0.1 * np.random.RandomState(3).randn(300), retol=1e-11, maxit=1000, precd=4)
    d, it, res, resk = ipd.PCG(A, b, o)
    do, ito, reso, resko = O.PCG(A, b, o)
    assert abs(it - ito) <= 1 and it < O.PCG(A, b, dict(o, precd=2))[1]
    assert np.linalg.norm(A @ d - b) <= 1e-9 * np.linalg.norm(b)
    assert np.allclose(d, do, rtol=1e-8, atol=1e-10)
    k = min(it, ito) - 2
    assert np.allclose(resk[:k], resko[:k], rtol=1e-6)
    m, n = 40, 30                                   # a Newton operator of the drivers (Jk-like)
    s = PR.mask_bernoulli(m, n, 0.3, seed=3)
    pd = PR.make_prob(m, n, s, pq_random=True)
    Ae = O.build_Ae(O.ASAt(s, pd["p"], pd["q"]), pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    f = np.random.RandomState(5).randn(m + n)
    o4 = dict(guess=None, retol=1e-11, maxit=5000, precd=4)
    d, it, res, _ = ipd.PCG(Ae, f, o4)
    do, ito, reso, _ = O.PCG(Ae, f, o4)
    assert abs(it - ito) <= 2
    assert np.linalg.norm(d - do) <= 1e-7 * np.linalg.norm(do)
    G = np.random.RandomState(2).randn(50, 50)
    Hd = sp.csc_matrix(G @ G.T + 50 * np.eye(50))
    d, it, res, _ = ipd.PCG(Hd, np.ones(50), dict(guess=None, retol=1e-11, maxit=100, precd=4))
    assert it <= 2 and np.linalg.norm(Hd @ d - 1.0) <= 1e-10 * np.sqrt(50)
    with pytest.raises(ipd.IpdError, match="nonpositive pivot"):
        ipd.PCG(sp.csc_matrix(np.array([[1.0, 2.0], [2.0, 1.0]])), np.ones(2), dict(precd=4))


def test_pcg_ssor_preconditioners(ipd):
    """precd 3 (SSOR, w = 1.5, PCG.m:40-44,96-99) on a general SPD matrix and precd 5 (SSOR on the
    bigraph blocks, :52-62, needs pcg_options.nf) on a rescaled Newton operator."""
    A = PR.random_sym_graph_laplacian(300, seed=10, eps=0.5)
    b = np.random.RandomState(11).randn(300)
    o = dict(guess=0.1 * np.random.RandomState(3).randn(300), retol=1e-11, maxit=1000, precd=3)
    d, it, res, resk = ipd.PCG(A, b, o)
    do, ito, reso, resko = O.PCG(A, b, o)
    assert abs(it - ito) <= 1
    assert np.linalg.norm(A @ d - b) <= 1e-9 * np.linalg.norm(b)
    assert np.allclose(d, do, rtol=1e-8, atol=1e-10)
    k = min(it, ito) - 2
    assert np.allclose(resk[:k], resko[:k], rtol=1e-6)
    m, n = 60, 50
    s = PR.mask_bernoulli(m, n, 0.2, seed=3)
    pd = PR.make_prob(m, n, s, pq_random=True)
    Ae = O.build_Ae(O.ASAt(s, pd["p"], pd["q"]), pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    f = np.random.RandomState(5).randn(m + n)
    o5 = dict(guess=None, retol=1e-11, maxit=5000, precd=5, nf=n)
    d, it, res, resk = ipd.PCG(Ae, f, o5)
    do, ito, reso, resko = O.PCG(Ae, f, o5)
    assert abs(it - ito) <= 2
    assert np.linalg.norm(Ae @ d - f) <= 1e-8 * np.linalg.norm(f)
    assert np.linalg.norm(d - do) <= 1e-7 * np.linalg.norm(do)
    with pytest.raises(ValueError, match="requires pcg_options.nf"):
        ipd.PCG(Ae, f, dict(precd=5))


CASES = [
    ("tree64", 64, 64, lambda: PR.mask_tree(64, 64, seed=1)),
    ("tree_rect", 150, 90, lambda: PR.mask_tree(150, 90, seed=2)),
    ("tree256", 256, 256, lambda: PR.mask_tree(256, 256, seed=3)),
    ("dense96", 96, 96, lambda: PR.mask_bernoulli(96, 96, 1.0)),
    ("bern128", 128, 128, lambda: PR.mask_bernoulli(128, 128, 0.2)),
]


def _connected(Ae):
    return sp.csgraph.connected_components(Ae)[0] == 1


@pytest.mark.parametrize("name,m,n,mk", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("isnsp", [0, 1])
@pytest.mark.parametrize("cycle", ["v", "w"])
def test_single_cycle_matches_oracle(ipd, name, m, n, mk, isnsp, cycle):
    s = mk()
    Ae, pd = newton_matrix(m, n, s)
    assert _connected(Ae)
    o = O.amg_options_class1(cycle); o.update(fnode=n, isnsp=isnsp)
    ho = O.amg_setup(Ae, o, O.matlab_rng())
    h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    r = np.random.RandomState(5).randn(m + n)
    for k in range(1, ho.J + 1):
        rk = r[:ho.Ack[k].shape[0]]
        if cycle == "v":
            eo = O.MG_Vcycle(ho, rk, isnsp, k)
            e = ipd.MG_Vcycle(h, rk, isnsp, k)
        else:
            eo = O.MG_Wcycle(ho, rk, isnsp, k)
            e = ipd.MG_Wcycle(h, rk, isnsp, k)
        A = ho.Ack[k]
        # compare through the residual (near-null-space component is ill-determined)
        scale = np.linalg.norm(rk)
        assert np.linalg.norm(A @ (e - eo)) <= 1e-9 * scale, (k, np.linalg.norm(A @ (e - eo)) / scale)
    if cycle == "w" and ho.J >= 3:
        e0 = np.random.RandomState(6).randn(ho.Ack[2].shape[0]) * 1e-3
        rk = r[:ho.Ack[2].shape[0]]
        eo = O.MG_Wcycle(ho, rk, isnsp, 2, e0)
        e = ipd.MG_Wcycle(h, rk, isnsp, 2, e0)
        assert np.linalg.norm(ho.Ack[2] @ (e - eo)) <= 1e-9 * np.linalg.norm(rk)
    h.close()


@pytest.mark.parametrize("name,m,n,mk", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("isnsp", [0, 1])
@pytest.mark.parametrize("cycle", ["v", "w"])
def test_class_amg_residual_history(ipd, name, m, n, mk, isnsp, cycle):
    s = mk()
    Ae, pd = newton_matrix(m, n, s)
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    guess = pd["bk1"] * pd["tk"] * np.random.RandomState(4).random_sample(m + n)
    o = O.amg_options_class1(cycle); o.update(fnode=n, isnsp=isnsp, guess=guess)
    xo, ito, relo, rko, rhoo = O.Class_AMG(Ae, f, o, O.matlab_rng())
    x, it, rel, rk, rho = ipd.Class_AMG(Ae, f, o, ipd.MatlabRand())
    # Same residual history up to RES_TOL.  The cycle count may only differ when the
    # deciding residual sits within RES_TOL of retol (1e-11 is below the comparison bar).
    k = min(len(rk), len(rko))
    assert np.max(np.abs(rk[:k] - rko[:k])) <= RES_TOL, np.max(np.abs(rk[:k] - rko[:k]))
    if it != ito:
        assert abs(it - ito) == 1 and abs(rko[k - 1] - o["retol"]) <= RES_TOL, (it, ito, rk, rko)
    big = rko[:k] > 1e-8
    assert np.allclose(rk[:k][big], rko[:k][big], rtol=1e-6)
    assert np.linalg.norm(Ae @ x - f) <= max(10 * relo, 1e-10) * np.linalg.norm(Ae @ guess - f)


def test_class_amg_zero_rhs_and_no_cycle_quirk(ipd):
    m = n = 32
    Ae, pd = newton_matrix(m, n, PR.mask_tree(m, n, seed=9))
    o = O.amg_options_class1("v"); o.update(fnode=n, isnsp=1, guess=np.zeros(m + n))
    x, it, rel, rk, rho = ipd.Class_AMG(Ae, np.zeros(m + n), o, ipd.MatlabRand())
    assert it == 0 and rel == 0 and not x.any()          # Class_AMG.m:91-92
    # quirk A-8: a cycle value that is neither 'v' nor 'w' applies no correction
    o2 = dict(o); o2["cycle"] = 1; o2["maxit"] = 3
    f = np.ones(m + n)
    x, it, rel, rk, rho = ipd.Class_AMG(Ae, f, o2, ipd.MatlabRand())
    assert it == 3 and np.allclose(rk, 1.0) and not x.any()


def test_vcycle_linearity_full_size(ipd):
    """Size-independent property at BASELINE size (m=n=1024, M=2048): one V-cycle
    is a linear map, e(a r1 + b r2) == a e(r1) + b e(r2)."""
    m = n = 1024
    s = PR.mask_tree(m, n, seed=21)
    Ae, pd = newton_matrix(m, n, s)
    o = O.amg_options_class1("v"); o.update(fnode=n, isnsp=1)
    h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    assert h.level_sizes()[0] == 2048 and h.J >= 4
    rs = np.random.RandomState(8)
    r1, r2 = rs.randn(m + n), rs.randn(m + n)
    e1, e2 = ipd.MG_Vcycle(h, r1, 1), ipd.MG_Vcycle(h, r2, 1)
    e12 = ipd.MG_Vcycle(h, 0.3 * r1 - 1.7 * r2, 1)
    lin = 0.3 * e1 - 1.7 * e2
    A = sp.csr_matrix(Ae)
    assert np.linalg.norm(A @ (e12 - lin)) <= 1e-9 * np.linalg.norm(A @ lin)
    # and the cycle contracts the residual
    assert np.linalg.norm(r1 - A @ e1) < 0.5 * np.linalg.norm(r1)
    h.close()


@pytest.mark.parametrize("cycle", ["v", "w"])
@pytest.mark.parametrize("isnsp", [0, 1])
def test_single_workgroup_solver_matches_multi_launch_path(ipd, cycle, isnsp):
    """Small hierarchies are solved by ONE single-workgroup launch (k_solve_small); it must
    reproduce the multi-launch path (and hence the oracle) to the same tolerance."""
    import os
    m = n = 200
    s = PR.mask_tree(m, n, seed=11)
    Ae, pd = newton_matrix(m, n, s)
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    guess = pd["bk1"] * pd["tk"] * np.random.RandomState(4).random_sample(m + n)
    o = O.amg_options_class1(cycle)
    o.update(fnode=n, isnsp=isnsp, guess=guess)
    xs, its, rels, rks, rhos = ipd.Class_AMG(Ae, f, o, ipd.MatlabRand())
    os.environ["IPD_NO_SMALL"] = "1"
    try:
        xm, itm, relm, rkm, rhom = ipd.Class_AMG(Ae, f, o, ipd.MatlabRand())
    finally:
        os.environ.pop("IPD_NO_SMALL")
    xo, ito, relo, rko, rhoo = O.Class_AMG(Ae, f, o, O.matlab_rng())
    k = min(len(rks), len(rkm), len(rko))
    assert np.max(np.abs(rks[:k] - rkm[:k])) <= RES_TOL and np.max(np.abs(rks[:k] - rko[:k])) <= RES_TOL
    assert abs(its - itm) <= 1 and abs(its - ito) <= 1
    assert abs(np.linalg.norm(Ae @ xs - f) - np.linalg.norm(Ae @ xm - f)) <= 1e-9 * np.linalg.norm(f)
