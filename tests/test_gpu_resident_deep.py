"""Mask-form level-resident kernel, DEEP mode (csrc/ipd_resident_big.h, round 4): REALISTIC hierarchies whose
level 1 exceeds k_resident's 2048 rows -- the Newton systems of the m = n = 2048 Class 1 driver run (levels
about 4096 / 2048 / 640 / 190 / 55 / 15).  Level 1 and the level 1 <-> 2 transfers from the active-set bit
mask, level 2 as register slices, level 3 in polynomial form, the remote tail workgroup rooted at level 4.

Reference behaviour: AMG/Class_AMG.m:86-109, AMG/MG_Wcycle.m:13-46, AMG/MG_Vcycle.m:12-45, Hybrid_AMG.m:40-41.
Checked against the ORACLE directly (hierarchy sizes, cycle counts, residual histories to max(1e-10, 2 x the
oracle's own one-ulp sensitivity) capped at 1e-9, A(x - x_oracle) <= 1e-9 |f|) and against the multi-launch
path (IPD_NO_RESIDENT_DEEP=1)."""
from ctypes import byref, c_int32

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ipd_oracle as O
from tests.test_gpu_bench_workload import (bench_cycles, env, options, resident_kernel_name, same_history,
                                            solve_mode)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def capture(ipd, N, kcap):
    """Ae, f, tk of the first Newton system of APD iteration kcap + 1 of the Class 1 device driver on the
    synthetic m = n = N problem of SURVEY 8d (Hybrid_AMG.m:17-24 with p = q = 1, T = 0)."""
    rs = np.random.RandomState(1)
    c, r, l = rs.random_sample(N * N), rs.random_sample(N), rs.random_sample(N)
    l = l * r.sum() / l.sum()
    one = np.ones(N)
    ws = ipd.APDWorkspace(1, c, r, l, one, one, gama=np.inf)
    ws.warmup(0.0, 100)
    amg = dict(retol=1e-11, bigph=1, maxit=30, theta=1 / 4, smoth=5, cycle="w", isnsp=1, inter=1)
    ws.run(amg, ipd.MatlabRand(5489), iters=kcap)
    lam = ws.state()[2]
    sc = ws.begin(kcap + 1)
    ev = ws.eval(lam)
    ws.close()
    H0 = ipd.ASAt(ev["s"], one, one)
    Q0 = sp.diags(np.concatenate([one, -one]))
    Ae = sp.csr_matrix(sc["bk1"] * (Q0 @ Q0) + (1.0 / sc["tk"]) * ((Q0 @ H0) @ Q0))
    f = Q0 @ np.random.RandomState(3).standard_normal(2 * N)
    return Ae, f, sc["tk"]


@pytest.fixture(scope="module", params=[12, 24], ids=["k13", "k25"])
def newton2048(ipd, request):
    N = 2048
    Ae, f, tk = capture(ipd, N, request.param)
    if sp.csgraph.connected_components(Ae)[0] != 1:
        pytest.skip("the captured system has several components (the mask form needs the whole Ae)")
    return N, Ae, f, tk


def _deep_hierarchy(ipd, Ae, N, tk, cycle):
    h = ipd.AMGHierarchy(Ae, options(cycle, N), ipd.MatlabRand(5489))
    one = np.ones(N)
    h.attach_mask_operator(one, one, tk)
    return h


@pytest.mark.parametrize("cycle", ["w", "v"])
def test_deep_mode_against_the_oracle_and_the_launches(ipd, newton2048, cycle):
    N, Ae, f, tk = newton2048
    # Hybrid_AMG.m:40 starts from bk1 * tk * rand(M, 1): a non-zero guess for the W case, zeros for the V case
    x0 = 1e-4 * np.random.RandomState(4).random_sample(2 * N) if cycle == "w" else np.zeros(2 * N)
    h = _deep_hierarchy(ipd, Ae, N, tk, cycle)
    mode, grid, _ = solve_mode(h)
    if mode != 2:
        pytest.skip("hierarchy %s / %s not taken by the mask-form kernel's deep mode" % (
            h.level_sizes(), [h.level_dims(k)[1] for k in range(1, h.J + 1)]))
    assert resident_kernel_name(h) in ("k_resident_big<4,2,true>", "k_resident_big<8,2,true>")
    assert h.J >= 5 and 129 <= grid <= 256
    from codes_of_ipd_ssn_amg_method_amd import _lib
    lev, root = c_int32(), c_int32()
    _lib.check(_lib.lib.ipd_amg_resident_levels(h.handle, byref(lev), byref(root)))
    assert (lev.value, root.value) in ((3, 4), (4, 5))      # (4, 5): level 4 resident as well (six levels and more)
    with env(IPD_NO_RESIDENT_DEEP=1):
        hc = _deep_hierarchy(ipd, Ae, N, tk, cycle)
    assert solve_mode(hc)[0] == 0 and hc.level_sizes() == h.level_sizes()
    x, it, rr, relk, rhok = h.solve(f, x0)
    assert solve_mode(h)[2] == 0, "no hand-off timed out"
    xc, itc, rrc, relkc, rhokc = hc.solve(f, x0)
    nf_ = np.linalg.norm(f)
    same_history(it, np.asarray(relk), itc, np.asarray(relkc), tol=2e-10)
    assert np.linalg.norm(Ae @ (x - xc)) <= 1e-9 * nf_
    # the kernel's own norm (|x| ~ 50, |Ae| ~ 1e3: A*x carries ~1e-11 |f| of rounding at this size)
    assert np.linalg.norm(Ae @ x - f) <= (1.05 * rr + 1e-11) * nf_
    # ... against the oracle directly
    o = dict(options(cycle, N))
    o.update(guess=x0)
    xo, ito, rro, relko, rhoko, ho = O.Class_AMG(Ae, f, o, O.matlab_rng(5489), return_hierarchy=True)
    assert h.level_sizes() == ho.level_sizes()
    assert [h.level_dims(k)[1] for k in range(1, h.J + 1)] == ho.level_nnz()
    sgn = np.where(np.random.RandomState(11).random_sample(f.size) < 0.5, -1.0, 1.0)
    _, itp, _, relkp, _ = O.Class_AMG(Ae, f * (1.0 + 2.2e-16 * sgn), o, O.matlab_rng(5489))
    kk = min(ito, itp) + 1
    sens = float(np.max(np.abs(np.asarray(relko[:kk]) - np.asarray(relkp[:kk]))))
    same_history(it, np.asarray(relk), ito, np.asarray(relko), tol=min(max(1e-10, 2.0 * sens), 1e-9))
    assert np.linalg.norm(Ae @ (x - xo)) <= 1e-9 * nf_
    assert it >= 3, relk
    # K loop bodies (what bench.py times): against the launches, and run-to-run identical bits
    a = bench_cycles(h, f, x0, 3)[0]
    b = bench_cycles(hc, f, x0, 3)[0]
    assert np.linalg.norm(Ae @ (a - b)) <= 5e-9 * nf_
    assert np.array_equal(a, bench_cycles(h, f, x0, 3)[0])
    # zero right-hand side (Class_AMG.m:91-92)
    xz, itz, relz, relkz, rhokz = h.solve(np.zeros(2 * N), None)
    assert itz == 0 and relkz[0] == 0.0 and not xz.any()
    h.close()
    hc.close()
