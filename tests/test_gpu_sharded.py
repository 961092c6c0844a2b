"""Row-block sharding of the cycle on ONE GPU:
 * emulate mode plays G owners back to back on the shared vectors, so the slicing
   logic (ranges, grids, the union of the slices) is exercised for G = 2, 4, 8 and
   must reproduce the unsharded iterate BIT FOR BIT;
 * a real RCCL communicator of size 1 exercises the communicator plumbing;
 * with >= 2 GPUs in the box, one process per GPU over RCCL (tests/sharded_ranks.py) must
   reproduce the emulation bit for bit (skipped on the pool's one-GPU boxes)."""
import os
from ctypes import byref, c_double, c_int

import numpy as np
import pytest

from oracle import ipd_oracle as O
from tests import problems as PR
from tests.test_gpu_setup import newton_matrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def _run(lib, fn, h, f, x0, cycles):
    from codes_of_ipd_ssn_amg_method_amd import _lib
    db = _lib.DeviceBuffer.from_array(f)
    dx = _lib.DeviceBuffer.from_array(x0)
    ms, bpc = c_double(), c_double()
    _lib.check(fn(h.handle, db.ptr, dx.ptr, c_int(cycles), byref(ms), byref(bpc)))
    x = dx.to_array(np.float64, len(x0))
    db.free()
    dx.free()
    return x


@pytest.mark.parametrize("mask,cycle,N1", [("dense", "v", 512), ("tree", "v", 512), ("tree", "w", 512),
                                           ("dense", "v", 2048)])
def test_sharded_emulation_is_bit_exact(ipd, mask, cycle, N1):
    """N1 = 2048 is BASELINE config 4's size (m=n=2048, M = 4096, 8 owners)."""
    from codes_of_ipd_ssn_amg_method_amd import _lib
    m = n = N1
    s = PR.mask_bernoulli(m, n, 1.0) if mask == "dense" else PR.mask_tree(m, n, seed=3)
    Ae, pd = newton_matrix(m, n, s)
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    x0 = np.random.RandomState(4).random_sample(m + n) * 1e-4
    o = O.amg_options_class1(cycle)
    o.update(fnode=n, isnsp=1)
    os.environ["IPD_NO_SMALL"] = "1"   # compare like with like: the multi-launch path
    os.environ["IPD_NO_RESIDENT"] = "1"
    try:
        h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    finally:
        os.environ.pop("IPD_NO_SMALL")
        os.environ.pop("IPD_NO_RESIDENT")
    ref = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles, h, f, x0, 3)
    os.environ["IPD_NO_GRAPH"] = "1"
    eager = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles, h, f, x0, 3)
    os.environ.pop("IPD_NO_GRAPH")
    assert np.array_equal(ref, eager)           # graph replay == eager launches
    for G in ((8,) if N1 > 512 else (2, 4, 8)):
        os.environ["IPD_SHARD_EMULATE"] = str(G)
        try:
            got = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles_sharded, h, f, x0, 3)
        finally:
            os.environ.pop("IPD_SHARD_EMULATE")
        assert np.array_equal(got, ref), G
    A = Ae
    assert np.linalg.norm(A @ ref - f) < 1e-3 * np.linalg.norm(A @ x0 - f)
    h.close()


@pytest.mark.parametrize("N1,G", [(512, 4), (2048, 8)])
def test_sharded_mask_operator_is_bit_exact(ipd, N1, G):
    """Sharded runs keep the matrix-free level 1 (the 1-bit-per-entry Gauss-Seidel half sweeps): G
    emulated owners, each sweeping its block of a half's rows, reproduce the unsharded mask-operator
    run BIT FOR BIT; N1 = 2048 with 8 owners is BASELINE config 4 as `bench.py --gpus 8 --n1 2048`
    runs it.  The operator is in use (attach returns True) and agrees with the CSR sweeps to rounding."""
    from codes_of_ipd_ssn_amg_method_amd import _lib
    m = n = N1
    Ae, pd = newton_matrix(m, n, PR.mask_bernoulli(m, n, 1.0))
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    x0 = np.random.RandomState(4).random_sample(m + n) * 1e-4
    o = O.amg_options_class1("v")
    o.update(fnode=n, isnsp=1)
    os.environ["IPD_NO_SMALL"] = "1"
    os.environ["IPD_NO_RESIDENT"] = "1"
    try:
        h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
        hc = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    finally:
        os.environ.pop("IPD_NO_SMALL")
        os.environ.pop("IPD_NO_RESIDENT")
    assert h.attach_mask_operator(pd["p"], pd["q"], pd["tk"])
    ref = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles, h, f, x0, 3)
    os.environ["IPD_SHARD_EMULATE"] = str(G)
    try:
        got = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles_sharded, h, f, x0, 3)
    finally:
        os.environ.pop("IPD_SHARD_EMULATE")
    assert np.array_equal(got, ref)
    csr = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles, hc, f, x0, 3)
    assert not np.array_equal(csr, ref)          # a different summation order: the operator DID run
    assert np.linalg.norm(Ae @ (csr - ref)) <= 1e-10 * np.linalg.norm(f)
    h.close()
    hc.close()


def test_rccl_communicator_of_one(ipd):
    from codes_of_ipd_ssn_amg_method_amd import _lib
    m = n = 256
    Ae, pd = newton_matrix(m, n, PR.mask_tree(m, n, seed=5))
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    x0 = np.zeros(m + n)
    o = O.amg_options_class1("v")
    o.update(fnode=n, isnsp=1)
    os.environ["IPD_NO_SMALL"] = "1"
    os.environ["IPD_NO_RESIDENT"] = "1"
    try:
        h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    finally:
        os.environ.pop("IPD_NO_SMALL")
        os.environ.pop("IPD_NO_RESIDENT")
    ref = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles, h, f, x0, 2)
    ident = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
    _lib.check(_lib.lib.ipd_comm_get_unique_id(_lib.bptr(ident)))
    ctx = _lib.get_ctx()
    _lib.check(_lib.lib.ipd_comm_init(ctx.handle, _lib.bptr(ident), c_int(0), c_int(1)))
    try:
        got = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles_sharded, h, f, x0, 2)
    finally:
        _lib.check(_lib.lib.ipd_comm_finalize(ctx.handle))
    assert np.array_equal(got, ref)
    h.close()


def _gpu_count():
    import torch
    return torch.cuda.device_count()       # does not initialise the GPU on this image


@pytest.mark.parametrize("mask,cycle,N1", [("dense", "v", 512), ("tree", "w", 512)])
def test_rccl_ranks_match_emulation(ipd, mask, cycle, N1, tmp_path):
    """The RCCL path itself (ncclAllGather after every sharded launch, csrc/ipd_dist.cpp): G real
    ranks, one per GPU, against G emulated owners on one GPU -- same bits, every rank holding the
    same iterate.  Needs G >= 2 GPUs: RCCL refuses two ranks on one device."""
    import socket
    import subprocess
    import sys
    from codes_of_ipd_ssn_amg_method_amd import _lib
    ngpu = _gpu_count()
    # one GPU (the pool's boxes): the rank program still runs with ONE rank, which checks the
    # harness (torchrun, gloo control plane, communicator) but not an all-gather
    G = 1 if ngpu < 2 else (2 if ngpu < 4 else 4)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "ranks.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "sharded_ranks.py"), out, str(N1), mask, cycle, "3"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    got = np.load(out)
    assert int(got["nranks"]) == G and bool(got["ranks_agree"])
    assert (int(got["allgathers"]) > 0) == (G > 1)
    m = n = N1
    s = PR.mask_bernoulli(m, n, 1.0) if mask == "dense" else PR.mask_tree(m, n, seed=3)
    Ae, pd = newton_matrix(m, n, s)
    f = np.concatenate([pd["q"], -pd["p"]]) * pd["z"]
    x0 = np.random.RandomState(4).random_sample(m + n) * 1e-4
    o = O.amg_options_class1(cycle)
    o.update(fnode=n, isnsp=1)
    os.environ["IPD_NO_SMALL"] = "1"
    os.environ["IPD_NO_RESIDENT"] = "1"
    try:
        h = ipd.AMGHierarchy(Ae, o, ipd.MatlabRand())
    finally:
        os.environ.pop("IPD_NO_SMALL")
        os.environ.pop("IPD_NO_RESIDENT")
    if G > 1:
        os.environ["IPD_SHARD_EMULATE"] = str(G)
        try:
            emu = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles_sharded, h, f, x0, 3)
        finally:
            os.environ.pop("IPD_SHARD_EMULATE")
    else:
        emu = _run(_lib.lib, _lib.lib.ipd_amg_bench_cycles, h, f, x0, 3)
    h.close()
    assert np.array_equal(got["x"], emu)
    if G == 1:
        pytest.skip("one GPU in this box: rank program checked with 1 rank only; the RCCL "
                    "all-gather needs >= 2 GPUs (RCCL refuses duplicate devices)")
