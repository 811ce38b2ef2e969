"""The multi-GPU path's sharding and reduce on CPU: world_size 2, gloo backend.
(On the MI355X node the same code runs with backend "nccl" = RCCL; the per-GPU arithmetic --
local sum, scaling, requantisation -- is libpb_hip.so and is covered by the gpu tests.)"""
import importlib
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth

NANT, NFLT = 5, 4096 * 6


def _planes(a):
    return torch.from_numpy(synth.gauss(100 + a, NFLT).astype(np.float32))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coadd = importlib.import_module("vlite-fast_amd.coadd")
    mine = coadd.antennas_of_rank(NANT, rank, world)
    local = torch.zeros(NFLT, dtype=torch.float32)
    for a in mine:                      # stands in for pb_coadd_local on this rank's GPU
        local += _planes(a)
    coadd.reduce_to_root(local, root=0)
    q.put((rank, mine, local.numpy().copy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_antenna_sharding_and_reduce_world2():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "tests", "golden"),
                                                os.environ.get("PYTHONPATH", "")])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][1] == [0, 2, 4] and got[1][1] == [1, 3]            # antenna a -> rank a mod world
    part0 = (_planes(0) + _planes(2)) + _planes(4)
    part1 = _planes(1) + _planes(3)
    assert np.array_equal(got[0][2], (part0 + part1).numpy())       # fp32 sum of the two partial sums


def test_single_process_reduce_is_identity():
    coadd = importlib.import_module("vlite-fast_amd.coadd")
    t = torch.arange(8, dtype=torch.float32)
    assert torch.equal(coadd.reduce_to_root(t.clone()), t)
    assert coadd.antennas_of_rank(16, 3, 8) == [3, 11]
