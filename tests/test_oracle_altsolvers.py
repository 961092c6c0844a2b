"""CPU checks of the restated alternate inner solvers (row f4): every one of them solves the
same Newton system He*zeta = z as the direct solve, and the two-grid code is the two-level
special case of the multilevel restatement."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import ipd_oracle as O
from tests import problems as PR

OPTS = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="w", isnsp=1, inter=1,
            guess=None)


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
    return spla.spsolve(sp.csc_matrix(J), pd["z"][:M])


def direct_pot(pd):
    """The bordered system of APD_SsN_Class2.m:151-155 solved directly."""
    m, n = pd["m"], pd["n"]
    M = m + n
    s = np.asarray(pd["s"], float)
    ss = O.Ax(s * pd["phi"], pd["p"], pd["q"])
    cT = sp.bmat([[pd["T"], None], [None, sp.csr_matrix((1, 1))]])
    cH0 = sp.bmat([[pd["H0"], ss[:, None]], [ss[None, :], np.array([[pd["phi"] @ (s * pd["phi"])]])]])
    J = pd["bk1"] * sp.identity(M + 1) + (cT + cH0) / pd["tk"]
    return spla.spsolve(sp.csc_matrix(J), pd["z"])


@pytest.mark.parametrize("m,n,rho", [(40, 36, 0.12), (90, 80, 0.06)])
def test_inner_solvers_agree_with_the_direct_solve(m, n, rho):
    pd = prob(m, n, rho, 3)
    ref = direct(pd)
    po = dict(retol=1e-11, maxit=10000, precd=2, guess=None)
    for z in (O.Hybrid_AMG(pd, OPTS, O.matlab_rng())[0], O.Hybrid_twogrid(pd, OPTS, O.matlab_rng())[0],
              O.aug_PCG(pd, po)[0]):
        assert np.linalg.norm(z - ref) <= 1e-9 * np.linalg.norm(ref)


def test_pot_solvers_agree_with_the_direct_bordered_solve():
    m, n = 50, 44
    t = (np.random.RandomState(9).random_sample(m + n) < 0.7).astype(float)
    pd = prob(m, n, 0.1, 4, pot=True, t=t)
    pd["s"] = PR.mask_bernoulli(m, n, 0.1, seed=4)
    ref = direct_pot(pd)
    o = dict(OPTS, smoth=10, maxit=40)
    po = dict(retol=1e-11, maxit=10000, precd=2, guess=None)
    for z in (O.AMG4POT(pd, o, O.matlab_rng(), "amg")[0], O.AMG4POT(pd, o, O.matlab_rng(), "twogrid")[0],
              O.PCG4POT(pd, po)[0]):
        assert np.linalg.norm(z - ref) <= 1e-8 * np.linalg.norm(ref)


def test_twogrid_bigph_is_the_two_level_hierarchy():
    """Same operators as level 1 -> 2 of Class_AMG (transfer.m:19-25, Class_AMG.m:48-59)."""
    m, n = 60, 50
    pd = prob(m, n, 0.2, 3)
    Ae = O.build_Ae(pd["H0"], pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    if len(O.components(sp.csr_matrix(Ae))[1]) != 1:
        pytest.skip("mask is not connected")
    b = np.random.RandomState(2).standard_normal(m + n)
    b -= b.mean()
    o = dict(retol=1e-11, maxit=30, smoth=5, isnsp=1, fnode=n, guess=None)
    x, it, rr, hist, rhok = O.twogrid_bigph(Ae, b, o)
    assert rr <= 1e-11 and it <= 30
    assert np.linalg.norm(Ae @ x - b) <= 1e-10 * np.linalg.norm(b)
    Ac, Pro, _ = O.transfer(Ae, dict(bigph=1, fnode=n, isnsp=1, theta=0.25, inter=1), 1, O.matlab_rng())
    assert Ac.shape == (m, m) and Pro.shape == (m + n, m)
    assert np.allclose(np.asarray(Pro.sum(axis=1)).ravel(), 1.0)      # isnsp: rows sum to one


def test_twogrid_general_and_defaults():
    A = PR.random_sym_graph_laplacian(100, deg=3, seed=1)
    b = np.random.RandomState(2).standard_normal(100)
    x, it, rr, hist, _ = O.twogrid(A, b, dict(retol=1e-10, bigph=0, maxit=40, smoth=3, isnsp=1, guess=None),
                                   O.matlab_rng())
    assert rr <= 1e-10 and np.all(np.diff(hist) < 0)
    with pytest.raises(ValueError):
        O.twogrid(A, b, dict(bigph=1), O.matlab_rng())
