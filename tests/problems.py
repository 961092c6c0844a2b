"""Seeded synthetic inputs shared by the CPU and GPU tests.

Regime D ("dense stress"): s ~ Bernoulli(rho).  Regime R surrogate: a tree-like
active set with about m+n entries (every row picks a column, every column picks a
row, plus a few extras), which reproduces the realistic structure nnz(s) ~ m+n,
several connected components, 4-6 AMG levels.
"""
import numpy as np
import scipy.sparse as sp

BK1, TK = 0.0038, 0.0255  # the measured k=8 values quoted in SURVEY.md 8d


def mask_bernoulli(m, n, rho, seed=2):
    return (np.random.RandomState(seed).random_sample(m * n) < rho).astype(np.uint8)


def mask_tree(m, n, extra=0.05, seed=2, connect=True):
    rs = np.random.RandomState(seed)
    Y = np.zeros((m, n), np.uint8)
    Y[np.arange(m), rs.randint(0, n, m)] = 1
    Y[rs.randint(0, m, n), np.arange(n)] = 1
    k = int(extra * (m + n))
    Y[rs.randint(0, m, k), rs.randint(0, n, k)] = 1
    if connect:  # a path through the rows makes the graph connected
        for i in range(m - 1):
            j = int(np.flatnonzero(Y[i])[0])
            Y[i + 1, j] = 1
    return Y.reshape(-1, order="F").copy()


def make_prob(m, n, s, seed=3, t=None, pq_random=False):
    rs = np.random.RandomState(seed)
    z = rs.randn(m + n)
    if pq_random:
        p = 0.5 + rs.random_sample(m)
        q = 0.5 + rs.random_sample(n)
    else:
        p, q = np.ones(m), np.ones(n)
    T = sp.diags(t if t is not None else np.zeros(m + n), format="csr")
    return dict(m=m, n=n, p=p, q=q, s=s, bk1=BK1, tk=TK, z=z, T=T)


def random_sym_graph_laplacian(N, deg=4, seed=0, eps=1e-3):
    """eps*I + weighted graph Laplacian of a random connected graph."""
    rs = np.random.RandomState(seed)
    rows = np.concatenate([np.arange(N - 1), rs.randint(0, N, deg * N)])
    cols = np.concatenate([np.arange(1, N), rs.randint(0, N, deg * N)])
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    w = 0.5 + rs.random_sample(rows.size)
    W = sp.csr_matrix((w, (rows, cols)), shape=(N, N))
    W = W + W.T
    Lp = sp.diags(np.asarray(W.sum(axis=1)).ravel()) - W
    return sp.csr_matrix(Lp + eps * sp.identity(N))
