"""CPU tests of the device formulation and of the world_size > 1 decomposition.

1. The fused formulation used by the HIP kernels (half-sweep Gauss-Seidel, c from
   1'r - (A1)'e) equals the reference's literal smoother (oracle MG_Vcycle/MG_Wcycle).
2. Two gloo ranks that each own half of every row range and all-gather after every
   launch reproduce the single-rank result exactly (the N > 1 path of bench.py)."""
import os
import sys

import numpy as np
import pytest

from oracle import ipd_oracle as O
from oracle.sharded_ref import ShardedCycle
from tests import problems as PR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(m, n, mask, isnsp):
    s = PR.mask_tree(m, n, seed=3) if mask == "tree" else PR.mask_bernoulli(m, n, 0.6)
    pd = PR.make_prob(m, n, s)
    H0 = O.ASAt(s, pd["p"], pd["q"])
    Ae = O.build_Ae(H0, pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    o = O.amg_options_class1("v")
    o.update(fnode=n, isnsp=isnsp)
    return O.amg_setup(Ae, o, O.matlab_rng()), Ae


@pytest.mark.parametrize("mask", ["tree", "dense"])
@pytest.mark.parametrize("isnsp", [0, 1])
@pytest.mark.parametrize("wc", [False, True])
def test_fused_formulation_equals_literal_oracle(mask, isnsp, wc):
    m = n = 40
    h, Ae = _setup(m, n, mask, isnsp)
    r = np.random.RandomState(1).randn(m + n)
    lit = O.MG_Wcycle(h, r, isnsp) if wc else O.MG_Vcycle(h, r, isnsp)
    fused = ShardedCycle(h, nf1=n).cycle(r, isnsp, wc)
    # rounding only: the two forms associate the same sums differently (|e| ~ 1e2 |r| here)
    assert np.linalg.norm(Ae @ (fused - lit)) <= 1e-9 * np.linalg.norm(r)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = n = 32
    h, Ae = _setup(m, n, "tree", 1)

    def gather(v, lo, hi):
        cnt = (hi - lo) // world
        mine = torch.from_numpy(v[lo + rank * cnt: lo + (rank + 1) * cnt].copy())
        out = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(out, mine)
        v[lo:hi] = torch.cat(out).numpy()

    r = np.random.RandomState(2).randn(m + n)
    e = ShardedCycle(h, nf1=n, rank=rank, G=world, gather=gather).cycle(r, 1, True)
    # throughput aggregation as bench.py does it: MAX over ranks of the wall time
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, e, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_row_block_cycle():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(rk, world, port, q)) for rk in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = n = 32
    h, Ae = _setup(m, n, "tree", 1)
    r = np.random.RandomState(2).randn(m + n)
    single = ShardedCycle(h, nf1=n).cycle(r, 1, True)
    for rank, e, tmax in res:
        assert np.array_equal(e, single)          # sharding does not change a single bit
        assert tmax == pytest.approx(0.2)
