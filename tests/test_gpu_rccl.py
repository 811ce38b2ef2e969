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
