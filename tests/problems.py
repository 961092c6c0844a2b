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
    """Tree-like active set, nnz ~ 2(m+n): every row and every column picks a random
    partner, a few extras; `connect` threads a path row i - column pi(i) - row i+1
    through a random permutation so that the graph is connected WITHOUT hub nodes
    (bounded degrees, as in the measured active sets of SURVEY appendix B)."""
    rs = np.random.RandomState(seed)
    Y = np.zeros((m, n), np.uint8)
    Y[np.arange(m), rs.randint(0, n, m)] = 1
    Y[rs.randint(0, m, n), np.arange(n)] = 1
    k = int(extra * (m + n))
    Y[rs.randint(0, m, k), rs.randint(0, n, k)] = 1
    if connect:
        pi = rs.permutation(max(m, n)) % n
        for i in range(m):
            Y[i, pi[i]] = 1
            if i + 1 < m:
                Y[i + 1, pi[i]] = 1
        for j in range(n):  # columns no row reached hang off a random row
            if not Y[:, j].any():
                Y[rs.randint(0, m), j] = 1
    return Y.reshape(-1, order="F").copy()


def mask_hub(m, n, seed=2):
    """Active set with a few high-degree columns: level 2 fills in (up to ~75 % dense),
    the peak-E behaviour SURVEY appendix B reports for the level-2 operator."""
    rs = np.random.RandomState(seed)
    Y = np.zeros((m, n), np.uint8)
    Y[np.arange(m), rs.randint(0, n, m)] = 1
    Y[rs.randint(0, m, n), np.arange(n)] = 1
    hubs = rs.choice(n, size=max(1, n // 64), replace=False)
    for j in hubs:
        Y[rs.random_sample(m) < 0.3, j] = 1
    Y[:, hubs[0]] = 1            # connects everything
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
