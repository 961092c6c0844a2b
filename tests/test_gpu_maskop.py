"""Matrix-free level-1 operator: the Gauss-Seidel sweeps of the bipartite level read
the active-set mask (1 bit per entry) instead of the CSR arrays.  It must (a) be accepted only
for matrices that have exactly Hybrid_AMG's rescaled form and (b) reproduce the CSR sweeps to
rounding (the row sums are added in a different order)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from oracle import ipd_oracle as O          # noqa: E402
from tests import problems as PR            # noqa: E402


def system(m, n, rho, seed, pq_random):
    s = PR.mask_bernoulli(m, n, rho, seed=seed)
    pd = PR.make_prob(m, n, s, pq_random=pq_random)
    H0 = O.ASAt(s, pd["p"], pd["q"])
    Ae = O.build_Ae(H0, pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    return pd, sp.csc_matrix(Ae)


@pytest.mark.parametrize("m,n,rho,pq", [(64, 64, 1.0, False), (100, 70, 0.5, True),
                                        (130, 257, 0.15, True), (33, 40, 0.9, True)])
@pytest.mark.parametrize("cycle", ["v", "w"])
def test_mask_operator_matches_csr_sweeps(m, n, rho, pq, cycle):
    import codes_of_ipd_ssn_amg_method_amd as ipd
    pd, Ae = system(m, n, rho, 5, pq)
    opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle=cycle, isnsp=1, inter=1,
                fnode=n)
    rs = np.random.RandomState(1)
    r = rs.standard_normal(m + n)
    r -= r.mean()
    import os
    os.environ["IPD_NO_SMALL"] = "1"       # the multi-launch path is the one that uses the operator
    try:
        h0 = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489))
        h1 = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489))
    finally:
        del os.environ["IPD_NO_SMALL"]
    assert h1.attach_mask_operator(pd["p"], pd["q"], pd["tk"])
    cyc = ipd.MG_Vcycle if cycle == "v" else ipd.MG_Wcycle
    e0 = cyc(h0, r, 1, 1)
    e1 = cyc(h1, r, 1, 1)
    A = sp.csr_matrix(Ae)
    assert np.linalg.norm(A @ (e1 - e0)) <= 1e-11 * np.linalg.norm(r)
    x0, it0, rr0, hist0, _ = h0.solve(r, np.zeros(m + n))
    x1, it1, rr1, hist1, _ = h1.solve(r, np.zeros(m + n))
    assert it0 == it1
    assert np.allclose(hist0[:it0 + 1], hist1[:it1 + 1], rtol=1e-6, atol=1e-12)


def test_mask_operator_rejects_other_matrices():
    import codes_of_ipd_ssn_amg_method_amd as ipd
    m, n = 48, 40
    pd, Ae = system(m, n, 0.6, 7, True)
    opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle="v", isnsp=1, inter=1,
                fnode=n)
    # the operator itself is fine for this matrix ...
    assert ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489)).attach_mask_operator(pd["p"], pd["q"], pd["tk"])
    # ... (a) but not with wrong scale vectors
    h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489))
    assert not h.attach_mask_operator(pd["p"] * 1.001, pd["q"], pd["tk"])
    assert not h.attach_mask_operator(pd["p"], pd["q"], pd["tk"] * 1.5)
    # (b) one off-diagonal value perturbed (still symmetric, still an M-matrix pattern)
    B = sp.lil_matrix(Ae)
    rows, cols = sp.triu(Ae, 1).nonzero()
    i, j = rows[3], cols[3]
    B[i, j] *= 1.0 + 1e-6
    B[j, i] = B[i, j]
    hb = ipd.AMGHierarchy(sp.csc_matrix(B), opts, ipd.MatlabRand(5489))
    assert not hb.attach_mask_operator(pd["p"], pd["q"], pd["tk"])
    # (a non-bipartite level 1 cannot get this far: bigph setup refuses it, transfer.m:20-21)
    # (c) sparse masks (fewer than 16 entries per row) keep the padded CSR sweeps: they are faster
    pds, Aes = system(m, n, 0.05, 7, True)
    assert not ipd.AMGHierarchy(Aes, opts, ipd.MatlabRand(5489)).attach_mask_operator(pds["p"], pds["q"], pds["tk"])
    # the rejected hierarchies still solve with the CSR kernels
    r = np.random.RandomState(2).standard_normal(m + n)
    x, it, rr, _, _ = hb.solve(r - r.mean(), np.zeros(m + n))
    assert rr <= 1e-10


def test_hybrid_amg_with_mask_operator(monkeypatch):
    """IPD_MASKOP=1 lets Hybrid_AMG attach the operator itself (dense mask, multi-launch path):
    same routing and cycle count as the CSR sweeps, solutions equal to solver tolerance."""
    import codes_of_ipd_ssn_amg_method_amd as ipd
    m = n = 96
    s = PR.mask_bernoulli(m, n, 0.8, seed=3)
    pd = PR.make_prob(m, n, s, pq_random=True)
    pd["H0"] = O.ASAt(s, pd["p"], pd["q"])
    opts = O.amg_options_class1("w")
    monkeypatch.setenv("IPD_NO_SMALL", "1")
    z0, it0, res0, info0 = ipd.Hybrid_AMG(pd, opts, ipd.MatlabRand())
    monkeypatch.setenv("IPD_MASKOP", "1")
    z1, it1, res1, info1 = ipd.Hybrid_AMG(pd, opts, ipd.MatlabRand())
    assert np.array_equal(info0, info1) and abs(it0 - it1) <= 1
    assert res1 <= 1e-10
    assert np.linalg.norm(z1 - z0) <= 1e-8 * max(1.0, np.linalg.norm(z0))
