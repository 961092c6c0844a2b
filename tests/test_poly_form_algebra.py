"""The polynomial form of a level visit (csrc/ipd_cycle.hip: k_pack_poly, k_bpoly_*)
restated in numpy and checked against the oracle's own smoothing loops (oracle/ipd_oracle.py:_smooth,
i.e. AMG/MG_Vcycle.m:14-41): the nu sweeps, the residual and the transfers of a visit are two dense
maps.  This pins the ALGEBRA the kernels implement; their arithmetic is checked on the GPU against the
oracle's solves (tests/test_gpu_resident_remote.py, tests/test_gpu_cycle.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ipd_oracle as O


def level(N, Nc, seed):
    rs = np.random.RandomState(seed)
    B = sp.random(N, N, density=0.3, random_state=rs, format="csr")
    A = (B + B.T).toarray()
    A = A + np.diag(np.abs(A).sum(axis=1) + 0.05)          # symmetric, diagonally dominant
    P = sp.random(N, Nc, density=0.25, random_state=rs, format="csr").toarray()
    P[:Nc, :] += np.eye(Nc)
    dinv = 0.5 / np.diag(A)                                 # Jacobi level: Class_AMG.m:60-62
    return A, P, dinv, rs


def stacked_operators(A, P, dinv, isnsp, nu):
    """What pack_bpoly / k_pack_poly produce: M2a, M1, w and the rows stacked below them."""
    N = A.shape[0]
    one = np.ones(N)
    Axi = A @ one
    xx = float(one @ Axi)
    u = (1.0 - dinv * Axi) / xx if isnsp else np.zeros(N)
    cs = one @ A                                            # column sums 1'A
    S = np.eye(N) - (dinv[:, None] * A + np.outer(u, cs))   # I - Rg A
    powers = [np.eye(N)]
    for _ in range(nu):
        powers.append(S @ powers[-1])
    M1 = powers[nu]
    ssum = sum(powers[:nu])                                 # I + S + ... + S^(nu-1)
    M2a = ssum * dinv[None, :]
    w = ssum @ u
    T1 = P.T @ A
    return dict(M1=M1, M2a=M2a, w=w, Mr_low=P.T - T1 @ M2a, Me_low=-T1 @ M1, W_low=-T1 @ w, Mc=M1 @ P)


@pytest.mark.parametrize("isnsp", [0, 1])
@pytest.mark.parametrize("N,Nc", [(14, 3), (40, 11), (66, 11), (100, 34)])
def test_a_visit_is_two_dense_maps(N, Nc, isnsp):
    nu = 5
    A, P, dinv, rs = level(N, Nc, 7 * N + isnsp)
    R = sp.diags(dinv).tocsr()
    As = sp.csr_matrix(A)
    r = rs.standard_normal(N)
    e0 = rs.standard_normal(N)
    ec = rs.standard_normal(Nc)
    M = stacked_operators(A, P, dinv, isnsp, nu)
    sumr = float(np.ones(N) @ r)
    for e in (np.zeros(N), e0):                             # first visit (zero start) and MG_Wcycle.m:30's
        pre = O._smooth(As, R, r, e.copy(), isnsp, nu)      # :14-25
        rc = P.T @ (r - A @ pre)                            # :27
        post = O._smooth(As, R, r, pre + P @ ec, isnsp, nu)  # :31-41
        pre_poly = M["M2a"] @ r + M["M1"] @ e + M["w"] * sumr
        rc_poly = M["Mr_low"] @ r + M["Me_low"] @ e + M["W_low"] * sumr
        post_poly = M["M2a"] @ r + M["M1"] @ pre + M["Mc"] @ ec + M["w"] * sumr
        scale = 1.0 + np.abs(pre).max() + np.abs(post).max()
        assert np.abs(pre - pre_poly).max() <= 1e-11 * scale
        assert np.abs(rc - rc_poly).max() <= 1e-11 * (scale + np.abs(rc).max())
        assert np.abs(post - post_poly).max() <= 1e-11 * scale
