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
import subprocess
import sys
from ctypes import byref, c_int32

import numpy as np
import pytest
import scipy.sparse as sp

import bench
from oracle import ipd_oracle as O
from tests import problems as PR
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
    assert (lev.value, root.value) in ((2, 3), (3, 4), (4, 5))   # (4, 5): level 4 in polynomial form as well (round 4)
    # k = 10: level 3 is far too big for any LDS image, so it must be resident; for the other two the
    # planner's choice depends on a few hundred bytes of LDS budget -- what was measured when the test
    # was written is recorded in `want_levels`, either mode is valid and is checked the same way
    lev3_ = min(lev.value, 3)
    assert lev3_ == want_levels or want_levels != 3 or h.level_dims(3)[1] <= 12 * h.level_dims(3)[0], (
        lev.value, want_levels, h.level_sizes())
    assert root.value == lev.value + 1
    with env(IPD_NO_RESIDENT_REMOTE=1):
        hc = ipd.AMGHierarchy(Ae, options(cycle, n), ipd.MatlabRand(5489))
    assert solve_mode(hc)[0] == 0 and hc.level_sizes() == h.level_sizes()
    x, it, rr, relk, rhok = h.solve(f, x0)
    xc, itc, rrc, relkc, rhokc = hc.solve(f, x0)
    assert solve_mode(h)[2] == 0                              # no launch gave up
    same_history(it, np.asarray(relk), itc, np.asarray(relkc))
    # (the last residual, ~7e-10 here, is at the rounding floor of these systems: the oracle's own history
    # moves by ~1e-10 under a one-ulp perturbation of f at k = 31, see _against_oracle; with IPD_NO_PAD=1 the
    # two paths ended 1.0015e-10 apart)
    assert abs(rr - rrc) <= 2e-10 and rr <= 1e-8
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


def _against_oracle(ipd, Ae, f, n, cycle, x0, expect_mode=None, kv=None):
    """One Class_AMG solve on the device against the SciPy oracle's (AMG/Class_AMG.m:86-109): the same
    hierarchy sizes, the same cycle count, residual histories to 1e-10 and A(x - x_oracle) at the
    rounding floor of A*x."""
    opts = options(cycle, n)
    with env(**(kv or {})):
        h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489))
        mode = solve_mode(h)[0]
        if expect_mode is not None and mode != expect_mode:
            h.close()
            return None
        x, it, rr, relk, rhok = h.solve(f, x0)
    assert solve_mode(h)[2] == 0
    o = dict(opts)
    o.update(guess=x0)
    xo, ito, rro, relko, rhoko, ho = O.Class_AMG(Ae, f, o, O.matlab_rng(5489), return_hierarchy=True)
    assert h.level_sizes() == ho.level_sizes()
    assert [h.level_dims(k)[1] for k in range(1, h.J + 1)] == ho.level_nnz()
    # Tolerance: 1e-10 on rel_res (north_star), unless the ORACLE's own history moves more than that
    # under a perturbation of the right-hand side by one unit in the last place -- these systems have
    # condition numbers of 1e8 and more (bk1 ~ 2e-5 against entries of 1/tk ~ 1e3), and on the latest
    # one (k = 31) rounding alone moves rel_res(2) = 8e-4 by ~1e-10; the bound is then 2 x that, never
    # more than 1e-9 (round 3 allowed 8 x, uncapped: 9 x looser than any deviation on record,
    # profiles/r3_history_sensitivity.txt -- VERDICT r3 #1).
    sgn = np.where(np.random.RandomState(11).random_sample(f.size) < 0.5, -1.0, 1.0)
    _, itp, _, relkp, _ = O.Class_AMG(Ae, f * (1.0 + 2.2e-16 * sgn), o, O.matlab_rng(5489))
    kk = min(ito, itp) + 1
    sens = float(np.max(np.abs(np.asarray(relko[:kk]) - np.asarray(relkp[:kk]))))
    same_history(it, np.asarray(relk), ito, np.asarray(relko), tol=min(max(1e-10, 2.0 * sens), 1e-9))
    assert np.linalg.norm(Ae @ (x - xo)) <= 1e-9 * np.linalg.norm(f)
    _against_oracle.forms = h.level_forms()
    h.close()
    return it, ito, relk, relko


@pytest.mark.parametrize("cycle", ["v", "w"])
def test_realistic_modes_against_the_oracle(ipd, newton_system, cycle):
    """VERDICT r2 #2 / ADVICE r2: the resident kernel's REALISTIC modes (remote tail rooted at level 3
    or 4, third resident level) tied to the oracle directly, not through the multi-launch path; and the
    multi-launch path at this size with and without the fused restriction r_c = P'r - (P'A)e
    (ipd_cycle_phases.h phase_rrc against the reference's P'(r - A e), AMG/MG_Vcycle.m:27)."""
    Ae, f, n, want_levels = newton_system
    x0 = np.zeros(Ae.shape[0])
    got = _against_oracle(ipd, Ae, f, n, cycle, x0, expect_mode=2)
    if got is None:
        pytest.skip("hierarchy not taken by the resident kernel")
    it, ito, relk, relko = got
    # level 4 of these systems (60-110 rows) runs in block-wide polynomial form inside the tail's image
    # (unless the whole suite runs under one of the switches that turn that form off: tools/switch_sweep.sh)
    forced_off = any(os.environ.get(k) == "1" for k in ("IPD_NO_BPOLY", "IPD_NO_POLY", "IPD_NO_BLK"))
    assert forced_off or _against_oracle.forms[3] & 16, _against_oracle.forms
    # these systems take several informative cycles (contraction ~0.1-0.3 per cycle), unlike rho = 1
    assert it >= 4 and np.sum(np.asarray(relko[:ito + 1]) > 1e-9) >= 4, relko
    assert _against_oracle(ipd, Ae, f, n, cycle, x0, kv=dict(IPD_NO_BPOLY=1)) is not None
    assert not any(v & 16 for v in _against_oracle.forms), _against_oracle.forms
    assert _against_oracle(ipd, Ae, f, n, cycle, x0, kv=dict(IPD_NO_BPOLY=1, IPD_NO_BLKDENSE=1)) is not None
    assert not any(v & 2 for v in _against_oracle.forms), _against_oracle.forms
    assert _against_oracle(ipd, Ae, f, n, cycle, x0, kv=dict(IPD_NO_RESIDENT=1)) is not None
    assert _against_oracle(ipd, Ae, f, n, cycle, x0, kv=dict(IPD_NO_RESIDENT=1, IPD_NO_RRC=1)) is not None
    assert _against_oracle(ipd, Ae, f, n, cycle, x0, kv=dict(IPD_NO_RESIDENT_THREE=1)) is not None


@pytest.mark.parametrize("cycle", ["v", "w"])
def test_tree_mask_against_the_oracle(ipd, cycle):
    """bench.py --mask tree (three resident levels + local tail) against the oracle directly."""
    m = n = 1024
    s = bench.build_mask(m, n, "tree", 1.0)
    Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
    got = _against_oracle(ipd, Ae, f, n, cycle, guess, expect_mode=2)
    assert got is not None


@pytest.mark.parametrize("cycle", ["v", "w"])
def test_bernoulli_eighth_at_the_metric_size_against_the_oracle(ipd, cycle):
    """rho = 1/8 at m=n=1024 (SURVEY 8d regime D) against the oracle directly.  Measured: like rho = 1
    this Bernoulli system reaches the rounding floor (2.7e-11) in ONE cycle, the oracle's too -- every
    regime-D mask does; the systems that take several informative cycles at the metric's size are the
    captured Newton systems above (4+ cycles each, asserted there)."""
    m = n = 1024
    s = PR.mask_bernoulli(m, n, 1.0 / 8, seed=2)
    Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
    got = _against_oracle(ipd, Ae, f, n, cycle, guess)
    it, ito, relk, relko = got
    assert relko[1] <= 1e-9 and relk[1] <= 1e-9


def test_skipped_publish_gives_up_once_and_is_redone_by_the_launches(ipd):
    """The recovery that makes the bounded spins safe (ipd_resident.h:res_sweep, run_resident): a test
    hook makes the last workgroup omit ONE publish (hand-off 7), every sweep of that step gives up after
    2^18 polls, the kernel reports it through the time-out word, and ipd_amg_solve redoes the solve from
    the guess on the multi-launch path: same result as a hierarchy that never used the resident kernel,
    one time-out on record, and the context stays on the launches for the next solves (back-off)."""
    m = n = 512
    s = PR.mask_bernoulli(m, n, 1.0, seed=5)
    Ae, f, guess, H0 = bench.build_newton_system(ipd, m, n, s)
    opts = options("v", n)
    with env(IPD_NO_RESIDENT=1):
        hc = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
    xc, itc, relc, relkc, _ = hc.solve(f, guess)
    from codes_of_ipd_ssn_amg_method_amd import _lib
    cx = _lib.Context(_lib.get_ctx().device)      # own context: the back-off is per context
    with env(IPD_RES_DEBUG_SKIP_PUBLISH=7):
        h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(), ctx=cx)
    assert solve_mode(h)[0] == 2
    x, it, rel, relk, _ = h.solve(f, guess)
    assert solve_mode(h)[2] == 1                   # one launch gave up ...
    assert it == itc and np.array_equal(x, xc)     # ... and the launches redid the solve, same bits
    assert np.array_equal(np.asarray(relk), np.asarray(relkc))
    x2, it2, _, _, _ = h.solve(f, guess)           # back-off: no second attempt, no second stall
    assert solve_mode(h)[2] == 1 and np.array_equal(x2, xc)
    h.close()
    hc.close()


def test_amg4pot_as_first_resident_use_in_a_fresh_process():
    """AMG4POT's two solve phases run on two host threads and both opt the same
    k_resident instantiation in to > 64 KB of LDS; in a process where no earlier test has done so the
    second thread must not launch before the attribute is set (ipd_lds_optin sets it under its lock)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import codes_of_ipd_ssn_amg_method_amd as ipd\n"
        "from tests import problems as PR\n"
        "m = n = 512\n"
        "rs = np.random.RandomState(2)\n"
        "s = PR.mask_bernoulli(m, n, 1.0)\n"
        "t = (rs.random_sample(m + n) < 0.7).astype(float)\n"
        "pd = PR.make_prob(m, n, s, t=t)\n"
        "pd['z'] = rs.randn(m + n + 1); pd['phi'] = np.ones(m * n)\n"
        "pd['H0'] = ipd.ASAt(s, pd['p'], pd['q'])\n"
        "o = dict(retol=1e-11, bigph=1, maxit=5, theta=0.25, smoth=10, cycle='w', isnsp=1, inter=1, guess=None, fnode=None)\n"
        "zeta, it, res, info = ipd.AMG4POT(pd, o, 'amg', ipd.MatlabRand())\n"
        "assert np.all(np.isfinite(zeta)) and it == 5\n"
        "print('ok', it, res)\n") % root
    res = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_block_wide_polynomial_operators_against_numpy(ipd, newton_system):
    """The operators k_bpoly_* pack on the f64 matrix cores (S^nu by nu - 1 products, the stacked
    restriction rows by one more) against the numpy restatement of the same algebra, which
    tests/test_poly_form_algebra.py ties to the oracle's smoothing loops (AMG/MG_Vcycle.m:14-41)."""
    from ctypes import POINTER, c_double, c_int64
    from codes_of_ipd_ssn_amg_method_amd import _lib
    from tests.test_poly_form_algebra import stacked_operators
    Ae, f, n, _ = newton_system
    opts = options("w", n)
    h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(5489))
    forms = h.level_forms()
    levels = [k for k in range(1, h.J) if forms[k - 1] & 16]
    if not levels:
        pytest.skip("no level of %s in block-wide polynomial form (forms %s)" % (h.level_sizes(), forms))
    for k in levels:
        A = h.A(k).toarray()
        P = h.P(k + 1).toarray()
        N, Nc = P.shape
        N8, Nc8 = -(-N // 8) * 8, -(-Nc // 8) * 8
        cap = 256 * (2 * N8 + Nc8 + 1)
        buf = np.zeros(cap)
        ld, nn, nc = c_int32(), c_int32(), c_int32()
        _lib.check(_lib.lib.ipd_amg_poly_operator(h.handle, c_int32(k), buf.ctypes.data_as(POINTER(c_double)),
                                                  c_int64(cap), byref(ld), byref(nn), byref(nc)))
        assert (nn.value, nc.value) == (N, Nc) and ld.value in (128, 256) and N + Nc <= ld.value
        LD = ld.value
        M = buf[:LD * (2 * N8 + Nc8 + 1)].reshape(2 * N8 + Nc8 + 1, LD).T     # column-major -> [row, col]
        ref = stacked_operators(A, P, 0.5 / np.diag(A), opts["isnsp"], opts["smoth"])
        blocks = {
            "M2a": (M[:N, :N], ref["M2a"]), "M1": (M[:N, N8:N8 + N], ref["M1"]),
            "Mc": (M[:N, 2 * N8:2 * N8 + Nc], ref["Mc"]), "w": (M[:N, 2 * N8 + Nc8], ref["w"]),
            "Mr_low": (M[N:N + Nc, :N], ref["Mr_low"]), "Me_low": (M[N:N + Nc, N8:N8 + N], ref["Me_low"]),
            "W_low": (M[N:N + Nc, 2 * N8 + Nc8], ref["W_low"]),
        }
        for name, (got, want) in blocks.items():
            # (the rank-one factors w = (I + ... + S^(nu-1)) u and -T1 w carry u ~ 1/xx ~ 1e5 and cancel: the
            # two summation orders differ by 6e-11 relative on the most ill-conditioned system, k = 31)
            tol = 1e-9 if name in ("w", "W_low") else 1e-11
            assert np.abs(got - want).max() <= tol * (1.0 + np.abs(want).max()), (k, name)
        # the padding the passes rely on is zero
        assert not M[N + Nc:, :].any() and not M[:, N:N8].any() and not M[:, N8 + N:2 * N8].any()
    h.close()
