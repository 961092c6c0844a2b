"""Rank program of tests/test_gpu_sharded.py::test_rccl_ranks_match_emulation: one process per GPU
(torch.distributed.run), the RCCL communicator of csrc/ipd_dist.cpp set up over a gloo control
plane exactly as bench.py does, `cycles` loop bodies of the row-block sharded cycle;
rank 0 writes the iterate and what RCCL reports to argv[1] (.npz)."""
import os
import sys
from ctypes import byref, c_double, c_int, c_int32, c_int64

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, N1, mask, cycle, cycles = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["IPD_DEVICE"] = os.environ.get("LOCAL_RANK", "0")
    os.environ["IPD_NO_SMALL"] = "1"
    os.environ["IPD_NO_RESIDENT"] = "1"
    import codes_of_ipd_ssn_amg_method_amd as ipd            # before torch (same ROCm runtime)
    from codes_of_ipd_ssn_amg_method_amd import _lib
    import torch
    import torch.distributed as dist
    from oracle import ipd_oracle as O
    from tests import problems as PR
    from tests.test_gpu_setup import newton_matrix
    dist.init_process_group("gloo")
    m = n = N1
    s = PR.mask_bernoulli(m, n, 1.0) if mask == "dense" else PR.mask_tree(m, n, seed=3)
    Ae, pd = newton_matrix(m, n, s)
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    x0 = np.random.RandomState(4).random_sample(m + n) * 1e-4
    o = O.amg_options_class1(cycle)
    o.update(fnode=n, isnsp=1)
    h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    ctx = _lib.get_ctx()
    ident = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
    if rank == 0:
        _lib.check(_lib.lib.ipd_comm_get_unique_id(_lib.bptr(ident)))
    dist.broadcast(torch.from_numpy(ident), 0)
    _lib.check(_lib.lib.ipd_comm_init(ctx.handle, _lib.bptr(ident), c_int(rank), c_int(world)))
    db = _lib.DeviceBuffer.from_array(f)
    dx = _lib.DeviceBuffer.from_array(x0)
    ms, bpc = c_double(), c_double()
    _lib.check(_lib.lib.ipd_amg_bench_cycles_sharded(h.handle, db.ptr, dx.ptr, c_int(cycles), byref(ms),
                                                     byref(bpc)))
    x = dx.to_array(np.float64, m + n)
    rr, nn, ag, agv = c_int32(), c_int32(), c_int64(), c_int64()
    _lib.check(_lib.lib.ipd_comm_stats(ctx.handle, byref(rr), byref(nn), byref(ag), byref(agv), c_int32(0)))
    # every rank must hold the same iterate: compare through the control plane
    t = torch.from_numpy(x.copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(torch.equal(lo, hi))
    _lib.check(_lib.lib.ipd_comm_finalize(ctx.handle))
    if rank == 0:
        np.savez(out_path, x=x, nranks=nn.value, rank=rr.value, allgathers=ag.value, vectors=agv.value,
                 ranks_agree=same, ms=ms.value)
    dist.barrier()
    dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
