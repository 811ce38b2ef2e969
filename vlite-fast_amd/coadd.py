"""Incoherent antenna sum across GPUs: RCCL reduce over xGMI in place of the reference's
external MPI coadder (`agdadacoadd`, /root/reference/scripts/start_coadd:16,56-58, which is not
in the reference repository -- its arithmetic is unpinned).

Partitioning: antennas are independent until the sum, so antenna a lives on rank
a mod world; each rank pre-sums the fp32 pre-quantisation planes of its own antennas on its GPU
(pb_coadd_local), ONE reduce(SUM, fp32) per batch brings the partial sums to the root, and the
root scales by 1/sqrt(N_ant) (the reference's variance-preserving convention,
src/pb_kernels.cu:522,568,623) and requantises (pb_coadd_finish -> sel_and_dig).
Only the root needs the result, so a reduce (not an all-reduce) is the right collective: on
xGMI's point-to-point links a direct-to-root reduce of 7 x 20 MiB per second of data is far
below one link's bandwidth.
"""
import torch
import torch.distributed as dist


def antennas_of_rank(nant, rank, world):
    """Antenna indices owned by `rank`: a mod world == rank."""
    return [a for a in range(nant) if a % world == rank]


def reduce_to_root(t, root=0, group=None):
    """Sum `t` over ranks into the root's tensor (in place).  No-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM, group=group)
    return t


class IncoherentCoadd(object):
    """Per-rank driver: handle = PbHandle(keep_ave=True) holding this rank's antennas."""

    def __init__(self, handle, nant_total, device, root=0):
        self.h = handle
        self.nant_total = nant_total
        self.root = root
        self.sum = torch.zeros(handle.max_seg * handle.ave_per_seg, dtype=torch.float32, device=device)

    def step(self, nseg):
        """After handle.process(nseg): returns the coadded filterbank bytes on the root, None elsewhere."""
        self.h.coadd_local(nseg, self.sum.data_ptr())
        self.h.sync()                       # library streams -> visible to the collective's stream
        reduce_to_root(self.sum, self.root)
        rank = dist.get_rank() if dist.is_initialized() else 0
        if rank != self.root:
            return None
        torch.cuda.synchronize()
        return self.h.coadd_finish(nseg, self.sum.data_ptr(), self.nant_total)
