"""GPU parity of the ordered sparse product (a7 of SURVEY §8: the Galerkin product's primitive),
called directly through `ipd_dmat_multiply`: both device kernels -- the row kernel and the
register-tile kernel of csrc/ipd_sparse.hip -- against the oracle's ordered product, BIT-EXACT,
on ragged shapes (single rows, sizes off the 64x64 tile grid, empty rows and columns)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ipd_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def rand_sparse(nr, nc, dens, seed, empty_rows=0):
    rs = np.random.RandomState(seed)
    M = sp.random(nr, nc, density=dens, random_state=rs, format="lil", data_rvs=rs.standard_normal)
    for r in rs.choice(nr, size=min(empty_rows, nr), replace=False):
        M[r, :] = 0
    M = sp.csr_matrix(M)
    M.eliminate_zeros()
    return M


def same_bits(A, B):
    A = sp.csc_matrix(A); B = sp.csc_matrix(B)
    A.sort_indices(); B.sort_indices()
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr)
            and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data))


SHAPES = [
    (1, 300, 200, 0.9, 0.9),      # a single output row (the 1-row coarsest level of a dense mask)
    (64, 64, 64, 1.0, 1.0),       # exactly one tile
    (65, 17, 129, 0.7, 0.6),      # every dimension off the tile grid
    (200, 333, 77, 0.05, 0.5),    # sparse left, denser right
    (150, 150, 150, 0.02, 0.02),  # sparse both (the usual coarse levels)
    (300, 40, 500, 1.0, 0.3),
    (37, 512, 3, 0.5, 1.0),       # three output columns
    # row counts beyond one pass of the scans that ride on the count launches (2048 entries per pass of a
    # 256-thread producer's tail, ipd_internal.h ScanTail) and beyond what a consumer scans for itself (4096)
    (2500, 60, 300, 0.3, 0.5),    # long rows of Y: the 256-thread row kernel or the tiles, either with a tail
    (5000, 40, 64, 0.2, 0.9),     # short rows of Y: the one-wave row kernel and a scan launch
]


@pytest.mark.parametrize("kind", ["rows", "tiles"])
@pytest.mark.parametrize("nr,nk,nc,dx,dy", SHAPES)
def test_product_bit_exact(ipd, monkeypatch, kind, nr, nk, nc, dx, dy):
    monkeypatch.setenv("IPD_PRODUCT", kind)
    X = rand_sparse(nr, nk, dx, seed=nr + nk, empty_rows=nr // 10)
    Y = rand_sparse(nk, nc, dy, seed=nk + nc + 1, empty_rows=nk // 10)
    got = ipd.sparse_multiply(X, Y)
    assert same_bits(got, O._spgemm(X, Y))


def test_product_exact_cancellation_is_dropped(ipd, monkeypatch):
    """An entry whose terms cancel exactly is not stored (MATLAB's sparse mtimes drops it)."""
    X = sp.csr_matrix(np.array([[1.0, 1.0], [2.0, 0.0]]))
    Y = sp.csr_matrix(np.array([[3.0, 5.0], [-3.0, 1.0]]))
    for kind in ("rows", "tiles"):
        monkeypatch.setenv("IPD_PRODUCT", kind)
        got = ipd.sparse_multiply(X, Y)
        assert got.nnz == 3 and got[0, 0] == 0.0
        assert same_bits(got, O._spgemm(X, Y))


def test_product_default_choice_matches(ipd, monkeypatch):
    """Whatever the time model picks, the bits are the row kernel's (a size where it picks tiles)."""
    X = rand_sparse(700, 900, 0.8, seed=5)
    Y = rand_sparse(900, 650, 0.8, seed=6)
    monkeypatch.delenv("IPD_PRODUCT", raising=False)
    auto = ipd.sparse_multiply(X, Y)
    monkeypatch.setenv("IPD_PRODUCT", "rows")
    assert same_bits(auto, ipd.sparse_multiply(X, Y))


def test_product_errors(ipd):
    with pytest.raises(ValueError):
        ipd.sparse_multiply(sp.identity(3, format="csr"), sp.identity(4, format="csr"))
