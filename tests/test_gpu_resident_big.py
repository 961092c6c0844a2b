"""Mask-form level-resident kernel (csrc/ipd_resident_big.h): level 1 of up to 4096 rows as the bit
mask, level 2 in registers, LDS holding only gather targets -- the kernel behind `bench.py --n1 2048`
(BASELINE config 4's size on one GPU).  Reference behaviour: AMG/Class_AMG.m:86-109,
AMG/MG_Vcycle.m:12-45, AMG/MG_Wcycle.m:13-46, PCG.m:68-87.

Forced (IPD_RESIDENT_BIG=1) on sizes the oracle solves in seconds: residual histories against the
ORACLE to 1e-10 and A(x - x_oracle) <= 1e-9 |f|, and against k_resident / the multi-launch path; then
m = n = 2048 itself against the multi-launch path (mask sweeps), K loop bodies and a whole solve."""
import numpy as np
import pytest
import scipy.sparse as sp

import bench
from oracle import ipd_oracle as O
from tests import problems as PR
from tests.test_gpu_bench_workload import (bench_cycles, env, options, resident_kernel_name, same_history,
                                            solve_mode)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def _system(m, n, rho, pq_random, seed=5):
    s = PR.mask_bernoulli(m, n, rho, seed=seed)
    pd = PR.make_prob(m, n, s, pq_random=pq_random)
    H0 = O.ASAt(s, pd["p"], pd["q"])
    Ae = sp.csr_matrix(O.build_Ae(H0, pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0])
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    return pd, Ae, f


@pytest.mark.parametrize("m,n,rho,pq,isnsp,cycle", [
    (512, 512, 1.0, False, 1, "v"),
    (1024, 1024, 1.0, False, 1, "w"),
    (700, 900, 1.0, True, 1, "v"),
    (1000, 1000, 0.9, True, 1, "w"),
    (1024, 1024, 0.5, True, 0, "v"),
])
def test_forced_big_kernel_against_oracle_and_resident(ipd, m, n, rho, pq, isnsp, cycle):
    pd, Ae, f = _system(m, n, rho, pq)
    if sp.csgraph.connected_components(Ae)[0] != 1:
        pytest.skip("mask not connected")
    guess = pd["bk1"] * pd["tk"] * np.random.RandomState(4).random_sample(m + n)
    opts = options(cycle, n, isnsp=isnsp)
    hc = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
    if hc.J != 3 or hc.level_sizes()[2] != 1:
        pytest.skip("hierarchy %s: the big kernel takes three levels with a one-row tail" % (hc.level_sizes(),))
    with env(IPD_RESIDENT_BIG=1):
        h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
        assert h.attach_mask_operator(pd["p"], pd["q"], pd["tk"])
    mode, grid, _ = solve_mode(h)
    assert mode == 2 and grid == -(-max(m, n) // 8), (mode, grid)
    x, it, rel, relk, rhok = h.solve(f, guess)
    assert solve_mode(h)[2] == 0, "no hand-off timed out"
    xc, itc, relc, relkc, rhokc = hc.solve(f, guess)
    same_history(it, relk, itc, relkc)
    A = sp.csr_matrix(Ae)
    assert np.linalg.norm(A @ (x - xc)) <= 1e-9 * np.linalg.norm(f)
    o = dict(opts)
    o.update(guess=guess)
    xo, ito, relo, relko, _ = O.Class_AMG(Ae, f, o, O.matlab_rng())
    same_history(it, relk, ito, relko)
    assert np.linalg.norm(A @ (x - xo)) <= 1e-9 * np.linalg.norm(f)
    # K loop bodies (what bench.py times) against the same hook of the other hierarchy
    a = bench_cycles(h, f, guess, 3)[0]
    b = bench_cycles(hc, f, guess, 3)[0]
    assert np.linalg.norm(A @ (a - b)) <= 5e-9 * np.linalg.norm(f)
    assert np.array_equal(a, bench_cycles(h, f, guess, 3)[0])      # run-to-run deterministic
    # zero right-hand side (Class_AMG.m:91-92)
    xz, itz, relz, relkz, rhokz = h.solve(np.zeros(m + n), None)
    assert itz == 0 and relkz[0] == 0.0 and np.isinf(rhokz[0]) and not xz.any()
    h.close()
    hc.close()


@pytest.fixture(scope="module")
def n2048(ipd):
    """m = n = 2048, regime D (BASELINE config 4's size) and the ORACLE's hierarchy of it: one
    O.amg_setup (about 35 s: the dense Galerkin products), shared by the V and the W case."""
    m = n = 2048
    s = bench.build_mask(m, n, "bernoulli", 1.0)
    Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
    o = dict(options("v", n))
    o.update(guess=guess)
    ho = O.amg_setup(Ae, o, O.matlab_rng())
    return m, n, Ae, f, guess, ho


@pytest.mark.parametrize("cycle", ["v", "w"])
def test_n2048_runs_in_the_big_kernel(ipd, n2048, cycle):
    """m = n = 2048, regime D (M = 4096: levels 4096 / 2048 / 1): the mask-form kernel on 256 workgroups
    -- k_resident_big<32>, the instantiation with 32 mask bits and 32 level-2 entries per lane, which only
    this size selects -- against the ORACLE directly (hierarchy sizes, K loop bodies through
    A(x - x_oracle), the residual history of a whole solve; AMG/Class_AMG.m:86-109, AMG/MG_Vcycle.m:12-45)
    and against the multi-launch path with the mask sweeps."""
    m, n, Ae, f, guess, ho = n2048
    opts = options(cycle, n)
    h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
    assert h.level_sizes() == [4096, 2048, 1] and solve_mode(h)[0] == 0
    assert h.level_sizes() == ho.level_sizes()
    assert [h.level_dims(k)[1] for k in range(1, h.J + 1)] == ho.level_nnz()
    assert h.attach_mask_operator(np.ones(m), np.ones(n), bench.TK)
    mode, grid, _ = solve_mode(h)
    assert (mode, grid) == (2, 256), (mode, grid)
    assert resident_kernel_name(h) == "k_resident_big<32,1,false>"
    with env(IPD_NO_RESIDENT_BIG=1):
        hc = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
        assert hc.attach_mask_operator(np.ones(m), np.ones(n), bench.TK)
    assert solve_mode(hc)[0] == 0
    K = 3
    a = bench_cycles(h, f, guess, K)[0]
    b = bench_cycles(hc, f, guess, K)[0]
    A = sp.csr_matrix(Ae)
    nf_ = np.linalg.norm(f)
    assert np.linalg.norm(A @ (a - b)) <= 5e-9 * nf_
    r0 = np.linalg.norm(A @ guess - f)
    assert np.linalg.norm(A @ a - f) < 1e-6 * r0
    # the oracle's K loop bodies on ITS hierarchy (Class_AMG.m:96-102 without the exit tests)
    xo = guess.copy()
    for _ in range(K):
        r = f - A @ xo
        xo = xo + (O.MG_Wcycle(ho, r, 1) if cycle == "w" else O.MG_Vcycle(ho, r, 1))
    assert np.linalg.norm(A @ (a - xo)) <= 1e-9 * nf_
    x, it, rel, relk, rhok = h.solve(f, guess)
    xc, itc, relc, relkc, rhokc = hc.solve(f, guess)
    assert solve_mode(h)[2] == 0
    same_history(it, relk, itc, relkc)
    assert np.linalg.norm(A @ (x - xc)) <= 1e-9 * nf_
    # a whole solve against O.Class_AMG's solve phase on the oracle's hierarchy
    o = dict(opts)
    o.update(guess=guess)
    xs, ito, relo, relko, rhoko = O.amg_solve(ho, f, o)
    same_history(it, relk, ito, relko)
    assert np.linalg.norm(A @ (x - xs)) <= 1e-9 * nf_
    h.close()
    hc.close()


@pytest.mark.parametrize("m,n,cycle", [(1024, 1024, "v"), (700, 900, "w")])
def test_rank_groups_are_bit_identical(ipd, m, n, cycle):
    """the mask-form kernel with its workgroups split into R rank groups, each polling a granule
    buffer of its OWN into which every publish is replicated (ResBigDesc::ranks, IPD_RESIDENT_RANKS=R) -- the
    data path of a row-block sharded run over peer-mapped xGMI buffers (one write-through store per peer, local
    polls), emulated on one GPU.  Nothing else is shared between workgroups: the iterates, cycle counts and
    residual histories equal the ungrouped run's bit for bit."""
    pd, Ae, f = _system(m, n, 1.0, m != n)
    guess = pd["bk1"] * pd["tk"] * np.random.RandomState(4).random_sample(m + n)
    opts = options(cycle, n)
    ref = None
    for R in (1, 2, 4, 8):
        with env(IPD_RESIDENT_BIG=1, IPD_RESIDENT_RANKS=R):
            h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
            assert h.attach_mask_operator(pd["p"], pd["q"], pd["tk"])
        assert resident_kernel_name(h).startswith("k_resident_big<") and resident_kernel_name(h).endswith(",1,false>")
        x, it, rel, relk, rhok = h.solve(f, guess)
        assert solve_mode(h)[2] == 0
        a = bench_cycles(h, f, guess, 3)[0]
        if ref is None:
            ref = (x, it, np.asarray(relk), a)
        else:
            assert it == ref[1] and np.array_equal(x, ref[0]) and np.array_equal(np.asarray(relk), ref[2]), R
            assert np.array_equal(a, ref[3]), R
        h.close()


def test_rank_groups_at_config4_size(ipd, n2048):
    """The same at m = n = 2048 (BASELINE config 4's size, 256 workgroups): eight rank groups of 32 workgroups --
    the partition north_star names -- against the ungrouped launch, bit for bit."""
    m, n, Ae, f, guess, ho = n2048
    opts = options("v", n)
    out = []
    for R in (1, 8):
        with env(IPD_RESIDENT_RANKS=R):
            h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
            assert h.attach_mask_operator(np.ones(m), np.ones(n), bench.TK)
        assert solve_mode(h)[:2] == (2, 256)
        out.append(bench_cycles(h, f, guess, 3)[0])
        h.close()
    assert np.array_equal(out[0], out[1])
