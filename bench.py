#!/usr/bin/env python3
"""bench.py -- V-cycle throughput (DoF*cycles/s) + achieved GB/s on the m=n=1024 OT grid.

A "step" is one iteration of the Class_AMG loop body (AMG/Class_AMG.m:96-105:
residual, one MG_Vcycle, norm) on a fixed hierarchy, with everything resident in
HBM before the timed region.  Workload = BASELINE.json's metric configuration
("m=n=1024 OT grid"), regime D of SURVEY.md 8d (the roofline point):
s ~ Bernoulli(rho), seed 2; bk1=0.0038, tk=0.0255; z ~ N(0,1) seed 3; guess =
bk1*tk*U(0,1) seed 4; options as the Class 1 driver (smoth 5, theta 1/4, bigph 1,
isnsp 1), cycle 'v'.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rho R] [--cycle v|w]
                  [--mask bernoulli|tree] [--n1 1024] [--no-cpu-baseline]

N>1 is launched by `python -m torch.distributed.run --nproc-per-node N ...`
(one rank per GPU): `--mode sharded` (default) runs ONE system row-block sharded
over the N GPUs with RCCL all-gathers of the iterate (strong scaling, the curve
north_star asks for); `--mode replicas` runs N independent systems (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
HANDOFF_FLOOR_US = 1.0  # MI355X_MICROARCH.md, table row `handoff-1to1`: an idle tagged-granule hand-off, 0.8-1.0 us
BK1, TK = 0.0038, 0.0255


def build_mask(m, n, kind, rho, seed=2):
    if kind == "bernoulli":
        return (np.random.RandomState(seed).random_sample(m * n) < rho).astype(np.uint8)
    from tests.problems import mask_hub, mask_tree
    return mask_hub(m, n, seed=seed) if kind == "hub" else mask_tree(m, n, seed=seed)


def capture_newton_system(ipd, N, kcap):
    """A REALISTIC Newton system (regime R of SURVEY 8d): the Class 1 device driver on the synthetic
    m=n=N problem (seed 1, draw order c, r, l; Sum l = Sum r) is run for `kcap` APD iterations, and
    the first semismooth-Newton system of iteration kcap + 1 is assembled exactly as Hybrid_AMG.m:17-24
    does (p = q = 1, T = 0).  Returns Ae, f, guess, nf, s (its largest connected component)."""
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
    guess = np.zeros(2 * N)
    ncomp, lab = sp.csgraph.connected_components(Ae)
    nf = N
    if ncomp > 1:      # Hybrid_AMG.m:55-70: the large component, F side (indices < n) first
        pk = np.flatnonzero(lab == np.argmax(np.bincount(lab)))
        Ae, f, guess, nf = sp.csr_matrix(Ae[pk, :][:, pk]), f[pk], guess[pk], int((pk < N).sum())
    return Ae, f, guess, nf, ev["s"], sc["bk1"], sc["tk"]


def build_newton_system(ipd, m, n, s):
    """One semismooth-Newton system of the Class 1 driver in the rescaled form of
    Hybrid_AMG.m:17-24: Ae u = f.  H0 comes from the GPU ASAt."""
    p, q = np.ones(m), np.ones(n)
    H0 = ipd.ASAt(s, p, q)
    qp = np.concatenate([q, -p])
    Q0 = sp.diags(qp)
    A0 = (Q0 @ H0) @ Q0
    Ae = sp.csr_matrix(BK1 * (Q0 @ Q0) + (1.0 / TK) * A0)
    z = np.random.RandomState(3).randn(m + n)
    f = qp * z
    guess = BK1 * TK * np.random.RandomState(4).random_sample(m + n)
    return Ae, f, guess, H0


def cpu_baseline(Ae, f, guess, opts, n, budget_s=8.0):
    """CPU baselines on the host's cores, same hierarchy options and the same Class_AMG loop body
    (AMG/Class_AMG.m:96-103), bounded samples:
      * `port`: oracle/cpu_cycle.c, the C/OpenMP restatement (explicit smoother matrices, CSR
        SpMVs, as the reference does it), at 1 thread and at more threads up to all cores;
      * `scipy_oracle`: the SciPy oracle the parity tests use, 1 thread.
    `value` is the best C figure (a reported baseline, not the target; MATLAB exists on no box)."""
    from oracle import ipd_oracle as O
    from oracle.cpu_cycle import CpuCycle
    o = dict(opts)
    o.update(fnode=n, guess=guess)
    t0 = time.perf_counter()
    h = O.amg_setup(Ae, o, O.matlab_rng())
    t_setup = time.perf_counter() - t0
    A = h.Ack[1]
    M = A.shape[0]
    wc = opts["cycle"] == "w"
    cc = CpuCycle(h, opts["isnsp"])
    # cores this process may run on (the box's cgroup share, not the machine's core count:
    # OpenMP threads beyond it only spin against each other)
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    runs = []
    for thr in sorted({1, min(8, ncores), min(32, ncores), ncores}):
        x, sec, res = cc.run(f, guess, 1, wc, thr)            # warm-up (thread pool, caches)
        if sec > 2.0:                                          # oversubscribed: not a baseline
            continue
        x, sec, res = cc.run(f, guess, 3, wc, thr)            # cycle-time estimate
        cyc = int(max(10, min(2000, 2.5 / max(sec / 3, 1e-7))))  # ~2.5 s per thread count
        x, sec, res = cc.run(f, guess, cyc, wc, thr)
        runs.append({"threads": thr, "cycles": cyc, "seconds": sec, "ms_per_cycle": 1e3 * sec / cyc,
                     "value": M * cyc / sec, "rel_res_after": float(res[-1] / res[0])})
    best = max(runs, key=lambda r: r["value"])
    # the SciPy oracle (what the parity tests compare with), 1 thread
    x = guess.copy()
    cycles = 0
    t0 = time.perf_counter()
    while True:
        r = f - A @ x
        x = x + (O.MG_Wcycle(h, r, opts["isnsp"]) if wc else O.MG_Vcycle(h, r, opts["isnsp"]))
        np.linalg.norm(A @ x - f)
        cycles += 1
        el = time.perf_counter() - t0
        if el >= budget_s or cycles >= 2000:
            break
    return dict(value=best["value"], unit="DoF*cycles/s", cores=best["threads"], kind="port",
                sample="oracle/cpu_cycle.c (C/OpenMP, explicit smoother matrices as the reference): "
                       "%d %s-cycles of the same hierarchy in %.2f s at %d thread(s), the best of the "
                       "thread counts tried (%d cores usable by this process; oracle setup %.1f s excluded)" % (
                           best["cycles"], opts["cycle"].upper(), best["seconds"], best["threads"],
                           ncores, t_setup),
                ms_per_cycle=best["ms_per_cycle"], by_threads=runs,
                scipy_oracle={"value": M * cycles / el, "cores": 1, "ms_per_cycle": 1e3 * el / cycles,
                              "sample": "%d cycles in %.1f s (oracle/ipd_oracle.py, SciPy float64)" % (cycles, el)})


def pmc_summary_for(args):
    """The newest committed rocprofv3 PMC summary (tools/summarize_pmc.py) taken on the workload of
    this run, or None.  Summaries record their bench arguments under "_workload"; the ones written
    before that key existed were all taken on the default command."""
    import glob
    want = {"n1": args.n1, "mask": args.mask, "rho": float(args.rho), "cycle": args.cycle}
    if args.mask != "bernoulli":
        want.pop("rho")
    if args.mask == "newton":
        want["newton_k"] = int(args.newton_k)
    for pf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary*.json")), reverse=True):
        try:
            pm = json.load(open(pf))
        except Exception:
            continue
        wl = pm.get("_workload", {"n1": 1024, "mask": "bernoulli", "rho": 1.0, "cycle": "v"})
        if all(wl.get(k) == v for k, v in want.items()):
            return pm
    return None


def rank_command(n, argv, port):
    """The command line `bench.py --gpus N` runs when it is started without a launcher: one rank
    per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n, argv, run=None):
    """Starts the N ranks as a child process (never exec: this process may be profiled), forwards
    rank 0's JSON line to stdout, everything else to stderr, and returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = rank_command(n, argv, port)
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    limit = float(os.environ.get("IPD_BENCH_RANKS_TIMEOUT", "1500"))
    if run is not None:
        res = run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, timeout=limit)
    else:
        # own session: on a time-out the WHOLE group goes (the launcher and its rank processes -- ranks left
        # behind would keep holding the GPUs), and the caller gets an exit code, not a traceback
        import signal
        import types
        child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = child.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            child.communicate()
            sys.stderr.write("bench.py: the ranks did not finish within %.0f s; process group killed\n" % limit)
            return 5
        res = types.SimpleNamespace(returncode=child.returncode, stdout=out)
    line = None
    for ln in (res.stdout or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line)
        sys.stdout.flush()
    if res.returncode == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited 0 without a result line\n")
        return 4
    return res.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n1", type=int, default=1024)
    ap.add_argument("--rho", type=float, default=1.0)
    ap.add_argument("--mask", default="bernoulli", choices=["bernoulli", "tree", "hub", "newton"],
                    help="newton: a realistic Newton system captured from the Class 1 device driver "
                         "at APD iteration --newton-k + 1 (regime R of SURVEY 8d)")
    ap.add_argument("--newton-k", type=int, default=30)
    ap.add_argument("--cycle", default="v", choices=["v", "w"])
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replicas"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=1,
                    help="also time B independent systems replayed concurrently on B HIP streams "
                         "(reported beside `value`, which stays the single-system number)")
    ap.add_argument("--maskop", action="store_true",
                    help="level 1 through the 1-bit-per-entry mask operator whatever its size "
                         "(default: from 4 M level-1 entries on, as the solvers do: below that the "
                         "padded CSR sweep is the faster launch, 5.2 us against 5.7 us)")
    ap.add_argument("--no-maskop", action="store_true", help="CSR sweeps on level 1 always")
    ap.add_argument("--no-poly2", action="store_true",
                    help="level 2 of the resident kernel as sweeps (33 hand-offs per V cycle) instead of the "
                         "composed polynomial form (24)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # invoked plainly (`python bench.py --gpus N`): start the ranks ourselves, as a CHILD
        # process, before this process has loaded the HIP library or touched the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a job of %d rank(s)" % (args.gpus, world))

    # The HIP library is loaded BEFORE torch so that the system ROCm runtime is the
    # one in the process (torch bundles its own libamdhip64 with the same SONAME).
    import codes_of_ipd_ssn_amg_method_amd as ipd
    from codes_of_ipd_ssn_amg_method_amd import _lib
    from ctypes import byref, c_double, c_int, c_int64
    from ctypes import c_int32 as _ci32
    if "IPD_DEVICE" not in os.environ:
        from ctypes import c_int32 as _ci32
        ndev = _ci32(0)
        _lib.check(_lib.lib.ipd_device_count(byref(ndev)))
        # fewer GPUs than ranks: the ranks double up on the devices there are, and RCCL then
        # refuses the communicator (duplicate device) -- the loud failure below, not a hang
        os.environ["IPD_DEVICE"] = str(local_rank % ndev.value if ndev.value > 0 else local_rank)

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("gloo")  # control plane only; the data path is RCCL in-library

    m = n = args.n1
    M = m + n
    nf = n
    if args.mask == "newton":
        Ae, f, guess, nf, s, bk1_, tk_ = capture_newton_system(ipd, args.n1, args.newton_k)
        M = Ae.shape[0]
    else:
        s = build_mask(m, n, args.mask, args.rho)
        Ae, f, guess, H0 = build_newton_system(ipd, m, n, s)
    opts = dict(retol=1e-11, bigph=1, maxit=30, theta=0.25, smoth=5, cycle=args.cycle, isnsp=1,
                inter=1, fnode=nf)
    t0 = time.perf_counter()
    h = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand())
    setup_s = time.perf_counter() - t0
    ctx = _lib.get_ctx()
    # matrix-free level 1: same policy as the solvers (csrc/ipd_cycle_host.h amg_attach_maskop)
    maskop = False
    if args.mask != "newton" and not args.no_maskop and (args.maskop or Ae.nnz >= 4.0e6):
        from ctypes import c_int32 as _ci32
        dp = _lib.DeviceBuffer.from_array(np.ones(m))
        dq = _lib.DeviceBuffer.from_array(np.ones(n))
        got = _ci32(0)
        _lib.check(_lib.lib.ipd_amg_attach_mask_operator(h.handle, dp.ptr, dq.ptr, c_int64(m),
                                                         c_int64(n), c_double(TK), byref(got)))
        maskop = bool(got.value)
    if args.mask == "newton" and M == 2 * args.n1 and M > 2048 and not args.no_maskop:
        # a realistic system beyond k_resident's 2048 rows: the mask-form kernel's deep mode needs the bit mask
        # (what Hybrid_AMG does itself for a whole, connected Ae: csrc/ipd_cycle_host.h amg_attach_maskop)
        maskop_deep = h.attach_mask_operator(np.ones(m), np.ones(n), tk_)
    xmask = False
    if args.mask != "newton" and not args.no_maskop:
        # level-resident kernel: level 1 <-> 2 transfers from the bit mask (what Hybrid_AMG does itself)
        xmask = h.attach_mask_transfers(np.ones(m), np.ones(n), TK)
    # level 2 composed over a visit (ipd_amg_attach_level2_poly): the fixed-hierarchy throughput the metric is --
    # many cycles on ONE hierarchy -- is what its 0.5 ms pack is for; it applies to three-level hierarchies with a
    # one-row tail and V cycles (the metric's workload), and is reported in config.level2_form
    poly2 = False
    if args.mask != "newton" and args.cycle == "v" and not args.no_poly2 and world == 1:
        poly2 = h.attach_level2_poly()
    db = _lib.DeviceBuffer.from_array(f)
    dx = _lib.DeviceBuffer.from_array(guess)

    sharded = world > 1 and args.mode == "sharded"
    rccl = None
    if sharded:
        # RCCL communicator for the in-library all-gathers; the unique id travels over gloo.
        # A rank that cannot join makes EVERY rank exit non-zero: a replicas number printed
        # where the sharded one is expected would be read as the scaling curve.
        import torch
        ok, why = 1, ""
        try:
            idbuf = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
            if rank == 0:
                _lib.check(_lib.lib.ipd_comm_get_unique_id(_lib.bptr(idbuf)))
            t = torch.from_numpy(idbuf)
            dist.broadcast(t, 0)
            _lib.check(_lib.lib.ipd_comm_init(ctx.handle, _lib.bptr(idbuf), c_int(rank),
                                              c_int(world)))
        except Exception as exc:
            ok, why = 0, str(exc)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            sys.stderr.write("bench.py: rank %d of %d: RCCL communicator could not be set up%s; "
                             "no sharded number can be measured -- exiting (use --mode replicas for "
                             "independent systems)\n" % (rank, world, (": " + why) if why else
                                                         " on another rank"))
            sys.stderr.flush()
            dist.destroy_process_group()
            sys.exit(2)
        from ctypes import c_int32 as _ci32
        rr_, nn_ = _ci32(), _ci32()
        _lib.check(_lib.lib.ipd_comm_stats(ctx.handle, byref(rr_), byref(nn_), None, None, _ci32(1)))
        rccl = {"nranks_seen_by_rccl": int(nn_.value), "rank_seen_by_rccl": int(rr_.value)}
        if rccl["nranks_seen_by_rccl"] != world:
            sys.stderr.write("bench.py: RCCL reports %d ranks, launched %d\n" % (nn_.value, world))
            sys.exit(2)

    def run_with(fn, cycles):
        ms, bpc = c_double(), c_double()
        _lib.check(fn(h.handle, db.ptr, dx.ptr, c_int(cycles), byref(ms), byref(bpc)))
        return ms.value, bpc.value

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        ctx.sync()

    def timed(fn):
        """W warm-up steps, then exactly K steps between barriers; MAX wall over ranks."""
        if args.warmup > 0:
            run_with(fn, args.warmup)
        barrier()
        t0 = time.perf_counter()
        ev_ms, bpc = run_with(fn, args.steps)
        barrier()
        wall_ = time.perf_counter() - t0
        if dist is not None:
            import torch
            tt = torch.tensor([wall_], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall_ = float(tt.item())
        return wall_, ev_ms, bpc

    replicas_result = None
    if sharded:
        wall, ev_ms, bytes_per_cycle = timed(_lib.lib.ipd_amg_bench_cycles_sharded)
        ag, agv = c_int64(), c_int64()
        from ctypes import c_int32 as _ci32
        _lib.check(_lib.lib.ipd_comm_stats(ctx.handle, None, None, byref(ag), byref(agv), _ci32(0)))
        ncyc = args.steps + max(args.warmup, 0)
        rccl["allgather_launches_per_cycle"] = ag.value / ncyc
        rccl["allgather_vectors_per_cycle"] = agv.value / ncyc
        # the throughput-oriented alternative of SURVEY 8e, reported beside the sharded curve
        rwall, _, _ = timed(_lib.lib.ipd_amg_bench_cycles)
        replicas_result = {"value": args.steps * M * world / rwall, "unit": "DoF*cycles/s",
                           "scaling": "weak", "ms_per_step": 1e3 * rwall / args.steps,
                           "parallelism": "replicas x%d (independent systems, no collective)" % world}
    else:
        wall, ev_ms, bytes_per_cycle = timed(_lib.lib.ipd_amg_bench_cycles)

    batched = None
    if args.batch > 1 and world == 1:
        # B independent copies of the system, one context (HIP stream) and one host thread each:
        # the latency-bound launches of different systems overlap on the device
        import threading
        hs, bufs = [], []
        for _ in range(args.batch):
            cx = _lib.Context(ctx.device)
            hb = ipd.AMGHierarchy(Ae, opts, ipd.MatlabRand(), ctx=cx)
            if maskop:
                hb.attach_mask_operator(np.ones(m), np.ones(n), TK)
            hs.append((cx, hb))
            bufs.append((_lib.DeviceBuffer.from_array(f, cx), _lib.DeviceBuffer.from_array(guess, cx)))

        def one(i, cycles):
            ms_, bp_ = c_double(), c_double()
            _lib.check(_lib.lib.ipd_amg_bench_cycles(hs[i][1].handle, bufs[i][0].ptr, bufs[i][1].ptr,
                                                     c_int(cycles), byref(ms_), byref(bp_)))

        def all_(cycles):
            th = [threading.Thread(target=one, args=(i, cycles)) for i in range(args.batch)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()

        all_(max(args.warmup, 1))
        t0 = time.perf_counter()
        all_(args.steps)
        bw = time.perf_counter() - t0
        batched = {"systems": args.batch, "value": args.steps * M * args.batch / bw,
                   "unit": "DoF*cycles/s", "ms_per_step": 1e3 * bw / args.steps,
                   "note": "independent systems on separate HIP streams of one GPU"}

    units = args.steps * M * (world if (world > 1 and not sharded) else 1)
    value = units / wall

    # did the timed steps do the work?  relative residual of the iterate they left behind
    # (warm-up + timed loop bodies, Class_AMG.m:103-104); a run that does not contract is an error
    from ctypes import c_int32 as _ci32
    x_end = dx.to_array(np.float64, M)
    r0 = float(np.linalg.norm(Ae @ guess - f))
    rel_res_after = float(np.linalg.norm(Ae @ x_end - f)) / r0
    mode_, grid_, tmo_ = _ci32(), _ci32(), _ci32()
    _lib.check(_lib.lib.ipd_amg_solve_mode(h.handle, byref(mode_), byref(grid_), byref(tmo_)))
    resident = mode_.value == 2 and not sharded
    if not np.isfinite(rel_res_after) or rel_res_after > 1e-6:
        sys.stderr.write("bench.py: the timed cycles did not contract the residual "
                         "(rel_res_after_steps = %g)\n" % rel_res_after)
        sys.exit(3)
    if tmo_.value:
        sys.stderr.write("bench.py: %d launch(es) of the resident kernel gave up (workgroups not "
                         "co-resident) and were redone by the multi-launch path\n" % tmo_.value)
    nu = opts["smoth"]
    visits2 = 2 if args.cycle == "w" else 1
    # phases of the multi-launch path: sweeps (2 launches on the Gauss-Seidel level), residual,
    # restriction, coarse PCG, prolongation, top, norm; where P'A is small (csrc: T1.nnz <= 2^18)
    # residual + restriction are ONE launch, r_c = P'r - (P'A)e
    fused_rrc = [h.level_dims(k)[1] * 2 <= (1 << 18) for k in (1, 2)] if h.J == 3 else None
    launches_classic = 4 * nu + 2 + visits2 * (2 * nu + 4) + 1 + 2
    # which resident kernel ran and how many chip-wide hand-offs its timed launch made: reported by the
    # library (ipd_amg_resident_kernel), not re-derived from sizes here.  A launch of K loop
    # bodies makes K x (hand-offs per cycle) + 1: the residual norm after the last cycle.
    kname, handoffs, big, xmask_lib = "", None, False, False
    if resident:
        from ctypes import create_string_buffer
        nb_, ho_, cy_, xm_ = create_string_buffer(64), c_int64(), _ci32(), _ci32()
        _lib.check(_lib.lib.ipd_amg_resident_kernel(h.handle, nb_, _ci32(64), byref(ho_), byref(cy_), byref(xm_)))
        kname = nb_.value.decode()
        big = kname.startswith("k_resident_big")
        xmask_lib = bool(xm_.value)
        if cy_.value == args.steps and ho_.value > 0:
            handoffs = (ho_.value - 1) / float(args.steps)
        else:
            sys.stderr.write("bench.py: the library reports %d hand-offs over %d cycles for its last resident "
                             "launch (timed: %d cycles)\n" % (ho_.value, cy_.value, args.steps))
    result = {
        "metric": "V-cycle throughput (DoF*cycles/sec), m=n=%d OT grid" % m,
        "value": value, "unit": "DoF*cycles/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
        "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": ("class1-OT m=n=%d, regime-D mask %s rho=%g, %s-cycle AMG "
                                "(smoth 5, theta 1/4, bigph, isnsp), fixed hierarchy" % (
                                    m, args.mask, args.rho, args.cycle.upper()))
                   if args.mask != "newton" else
                   ("class1-OT m=n=%d, regime R: first Newton system of APD iteration %d of the device "
                    "driver run (bk1 %.3e, tk %.3e), %s-cycle AMG (smoth 5, theta 1/4, bigph, isnsp), "
                    "fixed hierarchy" % (m, args.newton_k + 1, bk1_, tk_, args.cycle.upper())),
                   "M": M, "E": int(s.sum()), "levels": h.level_sizes(),
                   "level_nnz": [h.level_dims(k)[1] for k in range(1, h.J + 1)],
                   "level1_operator": "bit mask + scale vectors" if maskop else "CSR",
                   "level2_form": ("polynomial, composed over a visit (ipd_amg_attach_level2_poly)" if poly2
                                   else "Jacobi sweeps"),
                   "resident_transfers": (("bit mask + scale vectors" if xmask_lib else "CSR rows of P', P")
                                          if resident else None),
                   "parallelism": ("row-block sharded x%d, RCCL all-gather" % world) if sharded
                   else ("replicas x%d" % world if world > 1 else "single GPU")},
        "cycle_bytes_algorithmic": bytes_per_cycle,
        "cycle_GBps_algorithmic": bytes_per_cycle * args.steps / wall / 1e9,
        "device_ms_per_step_events": ev_ms / args.steps,
        "rel_res_after_steps": rel_res_after,
        "execution": ({"mode": ("level-resident kernel, mask form (csrc/ipd_resident_big.h)" if big else
                                "level-resident kernel (csrc/ipd_resident.h)") + ": the timed steps are ONE launch",
                       "kernel": kname, "workgroups": grid_.value, "launches_per_cycle": 1.0 / args.steps,
                       "handoffs_per_cycle": handoffs, "resident_kernel_timeouts": tmo_.value}
                      if resident else
                      {"mode": "one launch per phase" + (", graph replay" if not sharded else ", eager + RCCL"),
                       "launches_per_cycle": launches_classic if h.J == 3 else None,
                       "residual_restriction_fused_estimate": fused_rrc}),
        "replicas": replicas_result,
        "batched": batched,
        "rccl": rccl,
        "setup_seconds_host_api": setup_s,
    }

    if rank == 0:
        # roofline of the dominant kernel (k_smooth): per-launch algorithmic bytes over the
        # per-launch duration measured with HIP events on the library's stream
        tot_bytes = tot_ms = 0.0
        launches = 0
        per_level = []
        from ctypes import c_int as _ci
        for k in range(1, h.J):
            ms, lps, bps = c_double(), _ci(), c_double()
            reps = 200
            _lib.check(_lib.lib.ipd_amg_bench_sweeps(h.handle, _ci(k), _ci(reps), byref(ms),
                                                     byref(lps), byref(bps)))
            per_level.append({"level": k,
                              "kernel": "k_smooth_mask" if (k == 1 and maskop) else "k_smooth",
                              "us_per_launch": 1e3 * ms.value / (reps * lps.value),
                              "bytes_per_launch": bps.value / lps.value,
                              "GBps": bps.value * reps / ms.value / 1e6})
            # weight = launches per cycle of this level (2 nu sweeps)
            w = 2 * opts["smoth"] * (2 ** (k - 1) if args.cycle == "w" and k + 1 < h.J else 1)
            if per_level[-1]["kernel"] != "k_smooth":
                # level 1 through the bit-mask operator: it does not stream the matrix, so it
                # is reported beside the roofline of k_smooth, not inside it
                nwords = n * ((m + 63) // 64) + m * ((n + 63) // 64)
                result["level1_mask_operator"] = {
                    "kernel": "k_smooth_mask", "us_per_launch": per_level[-1]["us_per_launch"],
                    "csr_equivalent_bytes_per_launch": per_level[-1]["bytes_per_launch"],
                    "actual_bytes_per_launch": (8 * nwords) / 2 + 8 * (4 * M + M // 2),
                    "launches_per_cycle": w * lps.value}
                continue
            tot_bytes += w * bps.value
            tot_ms += w * ms.value / reps
            launches += w * lps.value
        achieved = tot_bytes / tot_ms / 1e6 if tot_ms > 0 else 0.0  # GB/s
        # HBM traffic per launch of the same kernel from the committed rocprofv3 PMC passes
        # (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command, gfx950 x2 read
        # correction applied by tools/summarize_pmc.py); null when no profile is present
        # ... of THIS workload only: a summary taken on another mask / size / cycle says nothing
        # about this line, and `traffic` is then null
        pm = pmc_summary_for(args)
        traffic = None
        for pk_, d in (pm or {}).items():
            if not isinstance(d, dict) or "hbm_traffic_bytes_per_launch" not in d:
                continue
            if "k_smooth_mask" in pk_:
                if "level1_mask_operator" in result:
                    result["level1_mask_operator"]["traffic"] = d["hbm_traffic_bytes_per_launch"]
            elif "k_smooth" in pk_:
                traffic = d["hbm_traffic_bytes_per_launch"]
        ksm = {"bound": "hbm", "kernel": "k_smooth", "achieved": achieved,
               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "traffic": traffic,
               "avg_us_per_launch": 1e3 * tot_ms / launches if launches else None,
               "avg_bytes_per_launch": tot_bytes / launches if launches else None,
               "per_level": per_level}
        if resident:
            # The dominant kernel IS the whole timed region: one launch of the resident kernel runs the
            # `steps` loop bodies, the matrices of levels 1-2 are read ONCE into registers, and what
            # bounds a cycle is the latency of its chip-wide hand-offs, not bytes.
            #   bound "latency": floor = hand-offs per cycle x 1.0 us -- MI355X_MICROARCH.md's
            #     `handoff-1to1` row prices an idle tagged-granule hand-off at 0.8-1.0 us;
            #   achieved / frac: COUNTER-based -- HBM-side bytes of the launch (rocprofv3 FETCH_SIZE /
            #     WRITE_SIZE passes of this workload, gfx950 corrections of tools/summarize_pmc.py) over
            #     its duration, against 8 TB/s; null without a committed PMC summary of this workload;
            #   effective_GBps: the ALGORITHMIC bytes of SURVEY 8d (every sweep re-reads its matrix) over
            #     the time -- what a matrix-streaming cycle would have to sustain; never a roofline fraction
            #     (it exceeds the HBM peak at m=n=2048).
            us_cycle = 1e3 * ev_ms / args.steps
            tr = None
            d = (pm or {}).get("k_resident")
            if d:   # traffic = fixed part (matrix load) + per-cycle part, from two dispatches
                tr = d["hbm_traffic_bytes_fixed"] + d["hbm_traffic_bytes_per_cycle"] * args.steps
            ach = (tr / (ev_ms * 1e-3) / 1e9) if tr else None
            floor = handoffs * HANDOFF_FLOOR_US if handoffs else None
            result["roofline"] = {
                "bound": "latency", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach is not None else None, "traffic": tr,
                "achieved_kind": "measured HBM-side bytes (PMC) / launch duration",
                "handoffs_per_cycle": handoffs, "handoff_floor_us": HANDOFF_FLOOR_US,
                "latency_floor_us": floor, "us_per_cycle": us_cycle,
                "frac_of_latency_floor": (floor / us_cycle) if floor else None,
                "us_per_handoff": (us_cycle / handoffs) if handoffs else None,
                "cycles_per_launch": args.steps, "us_per_launch": 1e3 * ev_ms,
                "effective_GBps": bytes_per_cycle * args.steps / ev_ms / 1e6,
                "algorithmic_bytes_per_launch": bytes_per_cycle * args.steps,
                "limiter": "hand-off latency",
                "note": "latency-bound: one launch keeps the matrices in registers and exchanges only the "
                        "iterate (tagged 16-byte granules, sc1); frac is the counter-based HBM fraction, "
                        "frac_of_latency_floor = (hand-offs x 1.0 us) / measured cycle time; "
                        "effective_GBps is NOT traffic (see profiles/r4_resident_stamps.txt)",
                "multi_launch_k_smooth": ksm}
        else:
            result["roofline"] = ksm
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(Ae, f, guess, opts, nf)
            result["cpu_baseline"]["host_cores_available"] = os.cpu_count()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
