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

`IncoherentCoadd` is the per-rank driver of that leg; bench.py (N > 1), the coadder host
(coadd_host.py, BASELINE configs[3]) and the tests all go through it.
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
    """The incoherent-sum leg of one rank, pipelined behind the batches of a PbHandle.

    handle: PbHandle(keep_ave=True, nsets=n) holding this rank's antennas.  The leg has a stream of its own
    (pb_set_coadd_stream): local sum -> reduce -> the root's requantisation of batch k are ordered on it by the
    device and run beside the kernels of the batches after it; nothing here synchronises the host except the
    rehearsal back end ("gloo": the partial sums cross through host memory).

    With one local antenna and the in-library FFT the plane to be reduced IS the antenna's plane: detect writes it
    straight into this object's buffer of the batch's buffer set (pb_set_coadd_target) and the local sum launches
    nothing.  Otherwise pb_coadd_local sums the rank's antennas into one buffer.

    source="codes": the local sum is taken from the antennas' QUANTISED filterbank bytes (each code stands for the
    centre of its quantiser cell; pb_coadd_local_codes) -- what a coadder fed from the co rings has to work with.
    The reduce and the root's requantisation are the same.  For like-for-like comparisons; the
    default sums the fp32 planes.

    queue(set_index, nseg): call once the batch in that buffer set is known to be complete on the device (its
    filterbank bytes have been fetched), i.e. one step behind the batch itself -- no device-side wait for detect is
    then queued (a pending cross-stream wait on detect's event was measured to cost the pipeline 0.12 ms per step).
    coadded(nseg, age): the root's view of the coadded filterbank bytes of the latest (age 0) / previous (age 1)
    queue() call, in pinned host memory (waits for that batch's requantisation only).
    """

    def __init__(self, handle, nant_total, device, root=0, backend="nccl", group=None, use_target=None, parts=7,
                 source="planes"):
        if source not in ("planes", "codes"):
            raise ValueError("source must be 'planes' or 'codes'")
        self.source = source               # "codes": sum the antennas' quantised bytes (pb_coadd_local_codes)
        if source == "codes":
            use_target = False
        self.h = handle
        self.nant_total = int(nant_total)
        self.root = root
        self.backend = backend
        self.group = group
        self.parts = parts                 # timing experiments: 1 local sum, 2 reduce, 4 requantisation
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = handle.max_seg * handle.ave_per_seg
        if use_target is None:
            use_target = handle.nant == 1 and handle.cfg.fft_backend == 0
        self.sums = [torch.zeros(n, dtype=torch.float32, device=device)
                     for _ in range(handle.nsets if use_target else 1)]
        self.use_target = bool(use_target)
        if self.use_target:
            for st in range(handle.nsets):
                handle.select_set(st)
                handle.set_coadd_target(self.sums[st].data_ptr())
            handle.select_set(0)
        # (a CPU `device` exists for the host-logic tests: a stand-in handle, no stream, gloo)
        self.stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        handle.sync()
        handle.set_coadd_stream(self.stream.cuda_stream if self.stream is not None else 0)
        self.queued = 0
        self.timing = False                # bench.py: device time of the collective (event pairs on the leg's stream)
        self._pairs = []

    def reduce_ms(self):
        """(total device ms, calls) of the collectives queued since the last call, timing switched on; waits for them"""
        tot, n = 0.0, 0
        for a, b in self._pairs:
            b.synchronize()
            tot += a.elapsed_time(b)
            n += 1
        self._pairs = []
        return tot, n

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def buffer(self, set_index):
        return self.sums[set_index] if self.use_target else self.sums[0]

    def queue(self, set_index, nseg):
        h, ds = self.h, self.buffer(set_index)
        h.select_set(set_index)
        with self._on_stream():
            if self.parts & 1:
                if self.source == "codes":
                    h.coadd_local_codes(nseg, ds.data_ptr())
                else:
                    h.coadd_local(nseg, ds.data_ptr())
            if self.parts & 2 and self.world > 1:
                if self.backend == "nccl":
                    if self.timing:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record(self.stream)
                    dist.reduce(ds, dst=self.root, op=dist.ReduceOp.SUM, group=self.group)      # RCCL over xGMI
                    if self.timing:
                        ev[1].record(self.stream)
                        self._pairs.append(ev)
                else:                              # rehearsal (gloo): through host memory
                    if self.stream is not None:
                        self.stream.synchronize()
                    t = ds.cpu()
                    dist.reduce(t, dst=self.root, op=dist.ReduceOp.SUM, group=self.group)
                    ds.copy_(t)
            if self.rank == self.root and self.parts & 4:
                h.coadd_finish(nseg, ds.data_ptr(), self.nant_total, blocking=False)
            if self.use_target:
                h.coadd_release()
        self.queued += 1

    def coadded(self, nseg, age=0):
        if self.rank != self.root:
            return None
        return self.h.coadd_view(nseg, age=age)

    def step(self, nseg):
        """Unpipelined form: after handle.process(nseg) on the selected set, -> the coadded bytes on the root
        (a copy), None elsewhere."""
        self.h.sync()
        self.queue(self.h_cur_set(), nseg)
        if self.stream is not None:
            self.stream.synchronize()
        v = self.coadded(nseg, 0)
        return None if v is None else v.copy()

    def h_cur_set(self):
        return getattr(self.h, "cur_set", 0)

    def close(self):
        if self.use_target:
            self.h.sync()
            for st in range(self.h.nsets):
                self.h.select_set(st)
                self.h.set_coadd_target(0)
