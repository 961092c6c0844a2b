"""Level-resident solve kernel with a REMOTE TAIL (csrc/ipd_resident.h, ResDesc::remote): realistic
hierarchies of the drivers' Newton systems have 4-6 levels; there the workgroups that keep levels
1-2 in registers hand r_3 = P3' rr_2 to ONE extra workgroup, which runs the single-workgroup
sub-cycle rooted at level 3 out of its LDS image (k_subcycle's code) and returns P3 e_3.

The system is a Newton system of the m=n=1024 Class 1 driver run (SURVEY 8d synthetic problem,
captured at APD iteration 31: levels about 2048 / 1024 / 320 / 100 / 30 / 10), i.e. BASELINE config
2's regime.  Reference behaviour: AMG/Class_AMG.m:86-109, AMG/MG_Vcycle.m:12-45,
AMG/MG_Wcycle.m:13-46.  Checked against the multi-launch path (IPD_NO_RESIDENT_REMOTE=1), which
tests/test_gpu_cycle.py ties to the oracle: same cycle counts, residual histories to 1e-10, timed
loop bodies to the rounding floor of A*x (5e-9 |f|), run-to-run identical bits."""
import os
from ctypes import byref, c_int32

import numpy as np
import pytest
import scipy.sparse as sp

from tests.test_gpu_bench_workload import bench_cycles, env, options, same_history, solve_mode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


@pytest.fixture(scope="module", params=[9, 20, 30], ids=["k10-long-level3", "k21-semi-root", "k31"])
def newton_system(ipd, request):
    """Ae, f of the Newton step the device driver reaches after 9 / 20 / 30 APD iterations: at k = 10
    level 3 (about 260 rows of 60 entries) is far too big for the tail's LDS, at k = 21 (about 310
    rows, 2 k entries) it would only be its semi-cached root -- both keep level 3 in registers with the
    tail rooted at level 4 -- and at k = 31 everything from level 3 down fits the tail's LDS image."""
    N, kcap = 1024, request.param
    want = {9: 3, 20: 3, 30: 2}[kcap]      # levels expected in the resident workgroups' registers
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
    H0 = ipd.ASAt(ev["s"], one, one)                         # Hybrid_AMG.m:17-24, p = q = 1, T = 0
    Q0 = sp.diags(np.concatenate([one, -one]))
    Ae = sp.csr_matrix(sc["bk1"] * (Q0 @ Q0) + (1.0 / sc["tk"]) * ((Q0 @ H0) @ Q0))
    f = Q0 @ np.random.RandomState(3).standard_normal(2 * N)
    ncomp, lab = sp.csgraph.connected_components(Ae)
    if ncomp > 1:      # Hybrid_AMG.m:55-70: the large component, F side (indices < n) first
        pk = np.flatnonzero(lab == np.argmax(np.bincount(lab)))
        assert len(pk) > 1500
        return sp.csr_matrix(Ae[pk, :][:, pk]), f[pk], int((pk < N).sum()), want
    return Ae, f, N, want


@pytest.mark.parametrize("cycle", ["v", "w"])
def test_remote_tail_matches_the_multi_launch_path(ipd, newton_system, cycle):
    Ae, f, n, want_levels = newton_system
    x0 = np.zeros(Ae.shape[0])
    h = ipd.AMGHierarchy(Ae, options(cycle, n), ipd.MatlabRand(5489))
    mode, grid, _ = solve_mode(h)
    assert h.J >= 4, h.level_sizes()
    if mode != 2:
        pytest.skip("hierarchy %s not taken by the resident kernel (level 3 does not fit the "
                    "sub-cycle's LDS image, or rows are not padded)" % h.level_sizes())
    assert grid == -(-max(n, Ae.shape[0] - n) // 8) + 1       # 8 rows of each block per workgroup + the tail
    from codes_of_ipd_ssn_amg_method_amd import _lib
    lev, root = c_int32(), c_int32()
    _lib.check(_lib.lib.ipd_amg_resident_levels(h.handle, byref(lev), byref(root)))
    assert (lev.value, root.value) in ((2, 3), (3, 4))
    # k = 10: level 3 is far too big for any LDS image, so it must be resident; for the other two the
    # planner's choice depends on a few hundred bytes of LDS budget -- what was measured when the test
    # was written is recorded in `want_levels`, either mode is valid and is checked the same way
    assert lev.value == want_levels or want_levels != 3 or h.level_dims(3)[1] <= 12 * h.level_dims(3)[0], (
        lev.value, want_levels, h.level_sizes())
    assert root.value == lev.value + 1
    with env(IPD_NO_RESIDENT_REMOTE=1):
        hc = ipd.AMGHierarchy(Ae, options(cycle, n), ipd.MatlabRand(5489))
    assert solve_mode(hc)[0] == 0 and hc.level_sizes() == h.level_sizes()
    x, it, rr, relk, rhok = h.solve(f, x0)
    xc, itc, rrc, relkc, rhokc = hc.solve(f, x0)
    assert solve_mode(h)[2] == 0                              # no launch gave up
    same_history(it, np.asarray(relk), itc, np.asarray(relkc))
    assert abs(rr - rrc) <= 1e-10 and rr <= 1e-8
    assert np.linalg.norm(Ae @ x - f) <= (1.01 * rr + 1e-12) * np.linalg.norm(f)   # the kernel's own norm
    assert np.linalg.norm(Ae @ (x - xc)) <= 1e-9 * np.linalg.norm(f)
    a = bench_cycles(h, f, x0, 3)[0]                          # what bench.py times
    b = bench_cycles(hc, f, x0, 3)[0]
    # |x| ~ 2.5e3 along the near-kernel vector and |Ae| ~ 1/tk ~ 1e3: A*x carries ~1e-8 of rounding
    assert np.linalg.norm(Ae @ (a - b)) <= 5e-9 * np.linalg.norm(f)
    again = bench_cycles(h, f, x0, 3)[0]                      # run-to-run deterministic
    assert np.array_equal(a, again)
    h.close()
    hc.close()
