"""GPU parity: Ax / Aty / ASAt / invAAt / invHHt through the C ABI vs the oracle.
Bar: ASAt and Aty bit-exact (integer pattern work + 2 mul + 1 add per entry);
Ax, invAAt, invHHt to 1e-13 relative (parallel summation order)."""
import numpy as np
import pytest

from oracle import ipd_oracle as O
from tests import problems as PR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


SHAPES = [(1, 1), (3, 5), (64, 64), (65, 63), (130, 257), (256, 256), (500, 500)]


@pytest.mark.parametrize("m,n", SHAPES)
def test_ax_aty(ipd, m, n):
    rs = np.random.RandomState(m * 1000 + n)
    p, q = 0.5 + rs.random_sample(m), 0.5 + rs.random_sample(n)
    x = rs.randn(m * n)
    y = rs.randn(m + n)
    got = ipd.Ax(x, p, q)
    ref = O.Ax(x, p, q)
    assert np.max(np.abs(got - ref)) <= 1e-13 * max(1.0, np.max(np.abs(ref))) * max(m, n)
    z = ipd.Aty(y, p, q)
    assert np.array_equal(z, O.Aty(y, p, q))          # bit-exact


def _csc_equal(A, B):
    A = A.tocsc(); B = B.tocsc()
    A.sort_indices(); B.sort_indices()
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr)
            and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data))


@pytest.mark.parametrize("m,n", SHAPES)
@pytest.mark.parametrize("rho", [0.0, 0.02, 0.5, 1.0])
def test_asat_bit_exact(ipd, m, n, rho):
    rs = np.random.RandomState(7)
    p, q = 0.5 + rs.random_sample(m), 0.5 + rs.random_sample(n)
    s = PR.mask_bernoulli(m, n, rho, seed=m + n)
    H = ipd.ASAt(s, p, q)
    assert _csc_equal(H, O.ASAt(s, p, q))


def test_asat_small_path_sequences(ipd):
    """The one-launch assembly of small active sets (k_asat_small) is chosen from the PREVIOUS call's
    entry count: sequences of calls on one context -- small after small (the path itself), a mask that
    outgrows the guess (detected, redone by the general path), squares that are exactly zero (their
    explicit zeros dropped as MATLAB's sparse() drops them), sizes that are no multiple of 64, E = 0."""
    rs = np.random.RandomState(3)
    for m, n in [(1024, 1024), (1000, 1024), (130, 257), (65, 63)]:
        p, q = 0.5 + rs.random_sample(m), 0.5 + rs.random_sample(n)
        masks = [PR.mask_tree(m, n, seed=4), PR.mask_tree(m, n, seed=5), PR.mask_bernoulli(m, n, 1.0 / 64, seed=6),
                 np.zeros(m * n, np.uint8), PR.mask_tree(m, n, seed=7), PR.mask_bernoulli(m, n, 0.6, seed=8),
                 PR.mask_tree(m, n, seed=9)]
        for s in masks:
            assert _csc_equal(ipd.ASAt(s, p, q), O.ASAt(s, p, q)), (m, n, int(s.sum()))
        pz, qz = p.copy(), q.copy()
        pz[::7] = 0.0
        qz[3::5] = 0.0
        for s in (PR.mask_tree(m, n, seed=10), PR.mask_tree(m, n, seed=11)):
            assert _csc_equal(ipd.ASAt(s, pz, qz), O.ASAt(s, pz, qz)), (m, n, "zero squares")


def test_asat_tree_mask_and_unit_pq(ipd):
    m, n = 300, 200
    s = PR.mask_tree(m, n, seed=4)
    H = ipd.ASAt(s, np.ones(m), np.ones(n))
    assert _csc_equal(H, O.ASAt(s, np.ones(m), np.ones(n)))
    assert (H != H.T).nnz == 0


def test_asat_linearity_full_size(ipd):
    """Size-independent property at the BASELINE size (m=n=1024): for disjoint masks
    ASAt(s1)+ASAt(s2) == ASAt(s1|s2), and 1'H1 == sum_ij s_ij (p_i+q_j)^2."""
    m = n = 1024
    rs = np.random.RandomState(5)
    p, q = np.ones(m), np.ones(n)
    s = PR.mask_bernoulli(m, n, 0.03, seed=11)
    half = rs.random_sample(m * n) < 0.5
    s1, s2 = s * half, s * (~half)
    H, H1, H2 = ipd.ASAt(s, p, q), ipd.ASAt(s1, p, q), ipd.ASAt(s2, p, q)
    assert abs(H - (H1 + H2)).max() == 0
    assert H.sum() == 4.0 * s.sum()


def test_asat_errors(ipd):
    with pytest.raises(ValueError):
        ipd.ASAt(np.zeros(5), np.ones(2), np.ones(2))


def test_inv_aat_hht(ipd):
    rs = np.random.RandomState(3)
    m, n = 70, 45
    p, q = 0.5 + rs.random_sample(m), 0.5 + rs.random_sample(n)
    x = rs.randn(m + n)
    for args in [(), (0.7,), (0.7, 1.9)]:
        ref = O.invAAt(x, p, q, *args)
        got = ipd.invAAt(x, p, q, *args)
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-13)
    phi = rs.random_sample(m * n)
    v = rs.randn(m + n + 1)
    ref = O.invHHt(v, p, q, 0.3, phi)
    got = ipd.invHHt(v, p, q, 0.3, phi)
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-13)
