"""Row f4 of SURVEY.md section 8: the alternate inner solvers of the drivers -- inner_solver 5
(`Hybrid_twogrid`, `AMG/twogrid_bigph.m`, `AMG4POT(...,'twogrid')`), 3 (`aug_PCG`, `PCG4POT`) and
2 (plain Jacobi-PCG on Jk) -- against their restatements in oracle/ipd_oracle.py."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from oracle import drivers as D             # noqa: E402
from oracle import ipd_oracle as O          # noqa: E402
from tests import problems as PR            # noqa: E402

OPTS = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="w", isnsp=1, inter=1,
            guess=None)


def ipd():
    import codes_of_ipd_ssn_amg_method_amd as pkg
    return pkg


def prob(m, n, rho, seed, pot=False, t=None):
    s = PR.mask_bernoulli(m, n, rho, seed=seed)
    pd = PR.make_prob(m, n, s, t=t, pq_random=True)
    pd["H0"] = O.ASAt(s, pd["p"], pd["q"])
    if pot:
        pd["z"] = np.random.RandomState(5).randn(m + n + 1)
        pd["phi"] = 0.5 + np.random.RandomState(6).random_sample(m * n)
    return pd


def direct(pd):
    M = pd["m"] + pd["n"]
    J = pd["bk1"] * sp.identity(M) + (pd["T"] + pd["H0"]) / pd["tk"]
    return sp.linalg.spsolve(sp.csc_matrix(J), pd["z"][:M])


@pytest.mark.parametrize("m,n,rho,isnsp", [(60, 50, 0.2, 1), (90, 80, 0.06, 1), (70, 64, 0.3, 0)])
def test_twogrid_bigph_matches_oracle(m, n, rho, isnsp):
    pd = prob(m, n, rho, 3)
    Ae = O.build_Ae(pd["H0"], pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    comp = O.components(sp.csr_matrix(Ae))[1]
    if len(comp) != 1:
        pytest.skip("mask is not connected")
    b = np.random.RandomState(2).standard_normal(m + n)
    if isnsp:
        b -= b.mean()
    o = dict(retol=1e-11, maxit=30, smoth=5, isnsp=isnsp, fnode=n,
             guess=1e-4 * np.random.RandomState(4).random_sample(m + n))
    x, it, rr, hist, rho_k = ipd().twogrid_bigph(sp.csc_matrix(Ae), b, o)
    xr, itr, rrr, histr, _ = O.twogrid_bigph(Ae, b, o)
    assert it == itr or abs(rr - o["retol"]) < 1e-10
    k = min(len(hist), len(histr))
    assert np.allclose(hist[:k], histr[:k], rtol=1e-6, atol=1e-10)
    A = sp.csr_matrix(Ae)
    assert np.linalg.norm(A @ x - b) <= 1e-9 * np.linalg.norm(b)


@pytest.mark.parametrize("N,isnsp", [(120, 1), (300, 1), (90, 0)])
def test_twogrid_general_matrix_matches_oracle(N, isnsp):
    """`AMG/twogrid.m` with bigph = 0: Jacobi smoother, mis_set coarsening (consumes rand)."""
    A = PR.random_sym_graph_laplacian(N, deg=3, seed=1)
    b = np.random.RandomState(2).standard_normal(N)
    o = dict(retol=1e-10, bigph=0, maxit=25, smoth=3, isnsp=isnsp, guess=None)
    rng = ipd().MatlabRand(5489)
    x, it, rr, hist, _ = ipd().twogrid(sp.csc_matrix(A), b, o, rng)
    orng = O.matlab_rng()
    xr, itr, rrr, histr, _ = O.twogrid(A, b, o, orng)
    assert it == itr
    assert np.allclose(hist, histr, rtol=1e-6, atol=1e-12)
    assert np.linalg.norm(A @ (x - xr)) <= 1e-9 * np.linalg.norm(b)
    with pytest.raises(ValueError, match="bigph = 1 requires fnode > 0"):
        ipd().twogrid(sp.csc_matrix(A), b, dict(o, bigph=1))


@pytest.mark.parametrize("m,n,rho", [(90, 80, 0.06), (40, 36, 0.12), (120, 110, 0.02)])
def test_hybrid_twogrid_matches_oracle(m, n, rho):
    pd = prob(m, n, rho, 3)
    z, it, res, info = ipd().Hybrid_twogrid(pd, OPTS, ipd().MatlabRand(5489))
    zr, itr, resr, infor = O.Hybrid_twogrid(pd, OPTS, O.matlab_rng())
    assert list(info) == list(infor) and it == itr
    ref = direct(pd)
    assert np.linalg.norm(z - ref) <= 1e-8 * np.linalg.norm(ref)
    assert np.linalg.norm(z - zr) <= 1e-8 * np.linalg.norm(zr)


def test_amg4pot_twogrid_matches_oracle():
    m, n = 70, 64
    t = (np.random.RandomState(9).random_sample(m + n) < 0.7).astype(float)
    pd = prob(m, n, 0.1, 4, pot=True, t=t)
    pd["s"] = PR.mask_bernoulli(m, n, 0.1, seed=4)
    o = dict(OPTS, smoth=10, maxit=40)
    z, it, res, info = ipd().AMG4POT(pd, o, "twogrid", ipd().MatlabRand(5489))
    zr, itr, resr, infor = O.AMG4POT(pd, o, O.matlab_rng(), "twogrid")
    assert it == itr and list(info) == list(infor)
    assert np.linalg.norm(z - zr) <= 1e-8 * np.linalg.norm(zr)


@pytest.mark.parametrize("m,n,rho", [(40, 36, 0.12), (90, 80, 0.06), (64, 64, 0.01)])
def test_aug_pcg_matches_oracle(m, n, rho):
    pd = prob(m, n, rho, 3)
    po = dict(retol=1e-11, maxit=10000, precd=2, guess=None)
    z, it, res, info = ipd().aug_PCG(pd, po)
    zr, itr, resr, infor = O.aug_PCG(pd, po)
    assert list(info) == list(infor)
    assert abs(it - itr) <= 2                      # the stopping test sits at the rounding floor
    ref = direct(pd)
    assert np.linalg.norm(z - ref) <= 1e-7 * np.linalg.norm(ref)
    assert np.linalg.norm(z - zr) <= 1e-7 * np.linalg.norm(zr)


def test_pcg4pot_matches_oracle():
    m, n = 50, 44
    t = (np.random.RandomState(9).random_sample(m + n) < 0.7).astype(float)
    pd = prob(m, n, 0.1, 4, pot=True, t=t)
    pd["s"] = PR.mask_bernoulli(m, n, 0.1, seed=4)
    po = dict(retol=1e-11, maxit=10000, precd=2, guess=None)
    z, it, res, info = ipd().PCG4POT(pd, po)
    zr, itr, resr, infor = O.PCG4POT(pd, po)
    assert list(info) == list(infor) and abs(it - itr) <= 2
    assert np.linalg.norm(z - zr) <= 1e-7 * np.linalg.norm(zr)


@pytest.mark.parametrize("solver", [1, 2, 3, 5])
def test_class1_driver_with_alternate_inner_solvers(solver):
    """All inner solvers solve the same Newton systems to 1e-11, so the APD histories agree."""
    from tests.test_gpu_driver import problem, ws_of
    pr = problem(1, 30, 26, seed=1)
    start = D.warmup_class1(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], np.inf, 100)
    ref = D.apd_ssn_class1(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], np.inf, inner="direct",
                           start=start)
    ws = ws_of(1, pr)
    ws.set_state(start[0], start[0], start[1], 1.0)
    out = ws.run(OPTS, ipd().MatlabRand(5489), inner_solver=solver)
    assert out["converged"] and out["k"] == ref["k"]
    assert abs(out["fval"] - ref["fval"]) <= 1e-7
    hist = ws.history()
    assert np.all(np.abs(hist["KKT_xk"] - np.asarray(ref["KKT_xk"])) <= 1e-6 * (1 + np.asarray(ref["KKT_xk"])))
    ws.close()


@pytest.mark.parametrize("solver", [1, 2, 3, 5])
def test_class2_driver_with_alternate_inner_solvers(solver):
    from tests.test_gpu_driver import problem, ws_of
    pr = problem(2, 26, 30, seed=1)
    start = D.warmup_class2(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], pr["mu"], pr["phi"], 100)
    ref = D.apd_ssn_class2(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], pr["mu"], pr["phi"],
                           inner="direct", start=start)
    ws = ws_of(2, pr)
    ws.set_state(start[0], start[0], start[1], 1.0)
    out = ws.run(dict(OPTS, smoth=10, maxit=40), ipd().MatlabRand(5489), inner_solver=solver)
    assert out["converged"] and out["k"] == ref["k"]
    assert abs(out["fval"] - ref["fval"]) <= 1e-7
    ws.close()


def test_out_of_range_inner_solver_fails_loudly():
    from tests.test_gpu_driver import problem, ws_of
    pr = problem(2, 8, 8, seed=1)
    ws = ws_of(2, pr)
    for bad in (0, 6):
        with pytest.raises(ipd().IpdError):
            ws.run(OPTS, ipd().MatlabRand(5489), inner_solver=bad)
    ws.close()


@pytest.mark.parametrize("cls", [1, 2])
def test_direct_inner_solver_matches_the_oracle_step_by_step(cls):
    """inner_solver = 1 (`zeta = Jk \\ (-Fk_old)`, APD_SsN_Class1.m:146-148, Class2 :152-156) against
    the restated driver with SciPy's sparse direct solve: same iteration count, Newton-step counts
    and histories (1e-8: two exact solves of the same systems differ by rounding only)."""
    from tests.test_gpu_driver import problem, ws_of
    pr = problem(cls, 40, 33, seed=2)
    if cls == 1:
        start = D.warmup_class1(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], np.inf, 100)
        ref = D.apd_ssn_class1(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], np.inf, inner="direct",
                               start=start)
    else:
        start = D.warmup_class2(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], pr["mu"], pr["phi"], 100)
        ref = D.apd_ssn_class2(pr["c"], pr["r"], pr["l"], pr["p"], pr["q"], pr["mu"], pr["phi"],
                               inner="direct", start=start)
    ws = ws_of(cls, pr)
    ws.set_state(start[0], start[0], start[1], 1.0)
    out = ws.run(OPTS, ipd().MatlabRand(5489), inner_solver=1)
    hist = ws.history()
    recs = ws.records()
    ws.close()
    assert out["converged"] and out["k"] == ref["k"]
    assert abs(out["fval"] - ref["fval"]) <= 1e-9 * (1 + abs(ref["fval"]))
    assert all(r["itamg"] == 1 and r["resamg"] == 0.0 and r["info0"] == 0 for r in recs)   # :148
    for key in ("KKT_xk", "KKT_lk"):
        a, b = hist[key], np.asarray(ref[key])
        assert a.shape == b.shape and np.all(np.abs(a - b) <= 1e-8 * (1 + np.abs(b))), key
    ssn, ssn_ref = hist["SsN_itnum"].astype(int), np.asarray(ref["SsN_itnum"]).astype(int)
    half = len(ssn_ref) // 2
    assert np.array_equal(ssn[:half], ssn_ref[:half]) and np.abs(ssn - ssn_ref).max() <= 1


@pytest.mark.parametrize("n,nrhs", [(1, 1), (37, 1), (64, 3), (65, 70), (200, 1), (333, 129)])
def test_spd_mldivide_matches_numpy(n, nrhs):
    """`A \\ B` for sparse symmetric positive definite A (ipd_spd_solve: blocked dense Cholesky on
    the device) against numpy; ragged sizes around the 64-row block."""
    import scipy.sparse as sp
    rs = np.random.RandomState(n + nrhs)
    G = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rs, format="csr")
    A = (G @ G.T + sp.identity(n) * (0.5 + rs.random_sample())).tocsc()
    B = rs.standard_normal((n, nrhs))
    X = ipd().spd_solve(A, B)
    ref = np.linalg.solve(A.toarray(), B)
    assert X.shape == ref.shape
    assert np.linalg.norm(X - ref) <= 1e-11 * np.linalg.cond(A.toarray()) * np.linalg.norm(ref)
    assert np.linalg.norm(A @ X - B) <= 1e-12 * (np.linalg.norm(A.toarray()) * np.linalg.norm(X) + np.linalg.norm(B))


def test_spd_mldivide_rejects_an_indefinite_matrix():
    import scipy.sparse as sp
    A = sp.csc_matrix(np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(ipd().IpdError):
        ipd().spd_solve(A, np.ones((2, 1)))
