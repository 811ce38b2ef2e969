"""The RCCL leg of the incoherent sum on the one GPU there is: a world_size-1 "nccl" (= RCCL) process group,
three bench.py-style pipelined steps with the coadd on a stream of its own (pb_set_coadd_stream), the
collective issued on that stream, the root's asynchronous requantisation (pb_coadd_finish, non-blocking)
and the one-batch-late collection (pb_coadd_fetch_ptr age 1).  With one antenna the 1/sqrt(N) scale is 1,
so the coadded bytes must be byte for byte the excised-stream codes of the same second.
(The N > 1 arithmetic is the gloo test's; a real multi-GPU run needs a node this pool does not hand out.)"""
import os

import numpy as np
import pytest

from helpers import libpb, make_input

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("target", [False, True])
def test_rccl_world1_pipelined_coadd_equals_single_antenna_codes(target):
    """target: detect writes the plane to be reduced straight into the caller's buffer (pb_set_coadd_target, one
    buffer per set), the local sum launches nothing, pb_coadd_release follows the collective."""
    import torch
    import torch.distributed as dist
    lp = libpb()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29650 + os.getpid() % 300)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        R, S, NSETS, NSTEP = 16, 4, 2, 3
        data = [make_input(50 + k, R, S) for k in range(NSTEP)]
        h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=R, max_seg=S, keep_ave=True, nsets=NSETS)
        d_sums = [torch.zeros(S * h.ave_per_seg, dtype=torch.float32, device=dev) for _ in range(NSETS if target else 1)]
        if target:
            for st in range(NSETS):
                h.select_set(st)
                h.set_coadd_target(d_sums[st].data_ptr())
        ts = torch.cuda.Stream(device=dev)
        h.sync()
        h.set_coadd_stream(ts.cuda_stream)
        single, coadded = [], []
        for k in range(NSTEP + 1):
            if k < NSTEP:
                h.select_set(k % NSETS)
                for s in range(S):
                    h.submit_planar(0, s, data[k][s, 0], data[k][s, 1])
                h.process(S)
                d_sum = d_sums[k % NSETS] if target else d_sums[0]
                with torch.cuda.stream(ts):
                    h.coadd_local(S, d_sum.data_ptr())
                    dist.reduce(d_sum, dst=0, op=dist.ReduceOp.SUM)          # RCCL, on the coadd stream
                    h.coadd_finish(S, d_sum.data_ptr(), 1, blocking=False)
                    if target:
                        h.coadd_release()
            if k >= 1:
                h.select_set((k - 1) % NSETS)
                single.append(h.fetch(0, 0, S)["kur"].copy())
                # the coadded bytes of batch k-1: the previous finish (age 1) while batch k is in flight,
                # the latest one (age 0) after the last batch
                coadded.append(np.array(h.coadd_view(S, age=1 if k < NSTEP else 0), copy=True))
        h.sync()
        torch.cuda.synchronize()
        for k in range(NSTEP):
            assert np.array_equal(coadded[k], single[k]), "batch %d" % k
        assert len(set(c.tobytes() for c in coadded)) == NSTEP        # three different seconds went through
        h.close()
    finally:
        dist.destroy_process_group()


def _sel_and_dig_8b(v):
    """sel_and_dig_8b (src/pb_kernels.cu:711-735) on a compact plane: float(double(v) / 0.02957 + 127.5), clamped,
    truncated"""
    tmp = (v.astype(np.float64) / 0.02957 + 127.5).astype(np.float32)
    return np.where(tmp <= 0, 0, np.where(tmp >= 255, 255, np.minimum(tmp, 255).astype(np.uint8))).astype(np.uint8)


def test_rccl_world1_two_antennas_per_gpu_local_sum_path():
    """BASELINE configs[3]'s per-GPU shape: TWO antennas batched on the GPU, so the plane that is reduced is the
    local-sum kernel's output (pb_coadd_local really launches, no coadd-target shortcut), then the RCCL reduce on
    the coadd stream, the root's non-blocking requantisation with 1 / sqrt(2) and the one-batch-late collection.
    Expected bytes: sel_and_dig of (plane_0 + plane_1) / sqrt(2) in fp32, from the antennas' own fp32 planes."""
    import torch
    import torch.distributed as dist
    lp = libpb()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29350 + os.getpid() % 300)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        R, S, NSETS, NSTEP, A = 16, 4, 2, 3, 2
        data = [[make_input(70 + 10 * k + a, R, S) for a in range(A)] for k in range(NSTEP)]
        h = lp.PbHandle(device=0, nant=A, nbit=8, npol=1, rfi_mode=2, rows_per_seg=R, max_seg=S, keep_ave=True, nsets=NSETS)
        d_sum = torch.zeros(S * h.ave_per_seg, dtype=torch.float32, device=dev)
        ts = torch.cuda.Stream(device=dev)
        h.sync()
        h.set_coadd_stream(ts.cuda_stream)
        expect, coadded = [], []
        for k in range(NSTEP + 1):
            if k < NSTEP:
                h.select_set(k % NSETS)
                for a in range(A):
                    for s in range(S):
                        h.submit_planar(a, s, data[k][a][s, 0], data[k][a][s, 1])
                h.process(S)
            if k >= 1:
                # batch k-1 is complete once its bytes are here; its leg is queued one step late (bench.py's order)
                h.select_set((k - 1) % NSETS)
                planes = [h.fetch(a, 0, S, raw=False, kur=False, ave=True)["ave_kur"] for a in range(A)]
                with torch.cuda.stream(ts):
                    h.coadd_local(S, d_sum.data_ptr())
                    dist.reduce(d_sum, dst=0, op=dist.ReduceOp.SUM)          # RCCL, on the coadd stream
                    h.coadd_finish(S, d_sum.data_ptr(), A, blocking=False)
                ssum = (np.float32(0) + planes[0]) + planes[1]
                expect.append(_sel_and_dig_8b(ssum * np.float32(1.0 / np.sqrt(2.0))))
                if k >= 2:
                    coadded.append(np.array(h.coadd_view(S, age=1), copy=True))
        coadded.append(np.array(h.coadd_view(S, age=0), copy=True))
        h.sync()
        torch.cuda.synchronize()
        for k in range(NSTEP):
            assert np.array_equal(coadded[k], expect[k]), "batch %d" % k
        assert len(set(c.tobytes() for c in coadded)) == NSTEP
        h.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("A,target", [(1, True), (1, False), (2, False)])
def test_rccl_world1_sliced_layout_runs_as_a_unit(A, target):
    """The sliced layout's RCCL branch, executed inside coadd.IncoherentCoadd exactly as a multi-GPU rank runs it --
    RCCL all_to_all_single of the plane slices -> pb_coadd_tree over the slices -> pb_coadd_digitise -> RCCL gather of
    the code bytes -> pb_coadd_publish, all ordered on the leg's own stream, one batch behind the pipeline, with and
    without the coadd-target shortcut -- in the only world an RCCL group can have on a one-GPU box: one rank
    (`slice_world_of_one`, a test hook: the product only slices with more than one rank).  The published bytes must equal
    what the single-rank form (`layout="root"`: pb_coadd_local_tree + pb_coadd_finish) gives for the same batches, and
    the sel_and_dig of the antennas' own fp32 planes summed in the defined order.  What this cannot show is the
    exchange between DIFFERENT ranks: that is the gloo / threaded tests' part (tests/test_gpu_coadd_host.py)."""
    import importlib
    import torch
    import torch.distributed as dist
    from helpers import parity_sum
    lp = libpb()
    cmod = importlib.import_module("vlite-fast_amd.coadd")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29050 + os.getpid() % 300)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        R, S, NSETS, NSTEP = 16, 4, 2, 3
        data = [[make_input(110 + 10 * k + a, R, S) for a in range(A)] for k in range(NSTEP)]
        got = {}
        for layout in ("sliced", "root"):
            h = lp.PbHandle(device=0, nant=A, nbit=8, npol=1, rfi_mode=2, rows_per_seg=R, max_seg=S, keep_ave=True, nsets=NSETS)
            leg = cmod.IncoherentCoadd(h, A, dev, root=0, backend="nccl", order="tree", layout=layout, use_target=target,
                                       slice_world_of_one=True)
            assert leg.layout == layout and leg.world == 1 and leg.use_target == target
            out, planes = [], []
            for k in range(NSTEP + 1):
                if k < NSTEP:
                    h.select_set(k % NSETS)
                    for a in range(A):
                        for s in range(S):
                            h.submit_planar(a, s, data[k][a][s, 0], data[k][a][s, 1])
                    h.process(S)
                if k >= 1:
                    h.select_set((k - 1) % NSETS)
                    h.fetch(0, 0, S)                                  # batch k - 1 is complete: its leg may be queued
                    leg.queue((k - 1) % NSETS, S)
                    if k >= 2:
                        out.append(np.array(leg.coadded(S, age=1), copy=True))
            out.append(np.array(leg.coadded(S, age=0), copy=True))
            torch.cuda.synchronize()
            if layout == "root":
                for k in range(NSTEP):
                    h.select_set(k % NSETS)
                    if k >= NSTEP - NSETS:      # (the sets still hold the last NSETS batches' planes)
                        planes.append((k, [h.fetch(a, 0, S, raw=False, kur=False, ave=True)["ave_kur"] for a in range(A)]))
            got[layout] = (out, planes)
            leg.close()
            h.close()
        for k in range(NSTEP):
            assert np.array_equal(got["sliced"][0][k], got["root"][0][k]), "batch %d" % k
        assert len(set(c.tobytes() for c in got["sliced"][0])) == NSTEP
        for k, pl in got["root"][1]:
            want = _sel_and_dig_8b(parity_sum(pl) * np.float32(1.0 / np.sqrt(float(A))))
            assert np.array_equal(got["sliced"][0][k], want), "batch %d vs the planes" % k
    finally:
        dist.destroy_process_group()
