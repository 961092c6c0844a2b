"""The opt-in single-workgroup whole-solve kernel for realistic hierarchies (csrc/ipd_mid.h,
IPD_MID=1): same cycle counts and residual histories as the multi-launch path, on a hierarchy
whose level 3 fits the LDS image and on one whose level 3 is walked in chunks."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from tests import problems as PR
from tests.test_gpu_bench_workload import env, options, same_history, solve_mode
from tests.test_gpu_setup import newton_matrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def mask_sparse_tree(m, n, extra, seed):
    """A path row i - column pi(i) - row i+1 through random permutations plus `extra` random
    entries: 2-3 entries per row and column, as in the active sets of late Newton steps."""
    rs = np.random.RandomState(seed)
    Y = np.zeros((m, n), np.uint8)
    pr, pc = rs.permutation(m), rs.permutation(n)
    for k in range(max(m, n)):
        Y[pr[k % m], pc[k % n]] = 1
        Y[pr[(k + 1) % m], pc[k % n]] = 1
    Y[rs.randint(0, m, extra), rs.randint(0, n, extra)] = 1
    return Y.reshape(-1, order="F").copy()


@pytest.mark.parametrize("m,n,extra,cycle", [(700, 640, 0, "w"), (1024, 1024, 100, "v"), (1024, 1024, 600, "w")])
def test_mid_kernel_matches_multilaunch(ipd, m, n, extra, cycle):
    s = mask_sparse_tree(m, n, extra, seed=11)
    Ae, pd = newton_matrix(m, n, s)
    if sp.csgraph.connected_components(Ae)[0] != 1:
        pytest.skip("mask not connected")
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    opts = options(cycle, n)
    # (the level-resident kernel with a remote tail would take the first hierarchy ahead of both)
    with env(IPD_MID=1, IPD_NO_RESIDENT_REMOTE=1):
        h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
    with env(IPD_NO_RESIDENT_REMOTE=1):
        hc = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
    if solve_mode(h)[0] != 3:
        pytest.skip("hierarchy %s (nnz %s) is not taken by the single-workgroup kernel" % (
            h.level_sizes(), [h.level_dims(k)[1] for k in range(1, h.J + 1)]))
    assert solve_mode(hc)[0] == 0
    x, it, rel, relk, rhok = h.solve(f, None)
    xc, itc, relc, relkc, rhokc = hc.solve(f, None)
    same_history(it, relk, itc, relkc)
    assert np.linalg.norm(Ae @ (x - xc)) <= 1e-9 * np.linalg.norm(f)
