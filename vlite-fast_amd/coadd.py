"""Incoherent antenna sum across GPUs, in place of the reference's external MPI coadder (`agdadacoadd`,
/root/reference/scripts/start_coadd:16,20-58, one rank per antenna ring under mpirun; it is not in the reference
repository -- its arithmetic is unpinned).

Partitioning: antennas are independent until the sum, so antenna a lives on rank a mod world.
What is summed are the fp32 pre-quantisation planes; the root scales by 1/sqrt(N_ant) (the reference's
variance-preserving convention, src/pb_kernels.cu:522,568,623) and requantises (pb_coadd_finish -> sel_and_dig).

The ORDER of the fp32 additions is defined, so that the coadded bytes are the same on 1, 2, 4, 8 (or any number of)
GPUs and can be checked against the oracle byte for byte (order="tree", the default):

    S(o, s) = plane of antenna o                    when o + s >= N_ant
            = S(o, 2 s) + S(o + s, 2 s)             otherwise (even members + odd members of {o, o+s, o+2s, ...})
    coadded = S(0, 1)

With world a power of two, rank r's antennas {r, r+W, ...} ARE the node S(r, W): the rank evaluates it on its own GPU
(pb_coadd_local_tree), ONE plane per rank goes to the root (a gather: point-to-point sends, one xGMI hop each, 7 x
20 MiB per second of data at 8 GPUs -- far below one link's bandwidth), and the root evaluates the top log2 W levels
over the partial sums in bit-reversed rank order (pb_coadd_tree).  Any other world size ships the antennas' planes
themselves and the root evaluates the whole tree: same bytes, more traffic.
Layout of the tree order across a power-of-two world (layout="auto" picks "sliced" with one output polarisation):
  "root"    every rank's plane is gathered to rank 0, which evaluates the top levels and requantises: 7 x 20 MiB per
            second of data into ONE GPU, and rank 0 does ~10 % more device work than the others (profiles/r05_notes.md);
  "sliced"  every rank evaluates the top levels for 1/W of the plane: ONE all-to-all of plane slices (every xGMI link of
            the full mesh carries 1/W of a plane, all links at once), pb_coadd_tree over the W received slices -- the same
            tree per element, hence the same bytes --, the slice requantised where it was summed (pb_coadd_digitise),
            and only code bytes gathered to rank 0 (7 x 0.65 MiB at 8 bits), which hands them out (pb_coadd_publish).
            The ranks' loads are equal.
order="fast" is one RCCL reduce(SUM, fp32) of the locally pre-summed planes to the root: the association is then the
collective's (ring / tree by topology and message size) and the bytes are not reproducible across world sizes; kept
for comparison (`--coadd-order fast`).

`IncoherentCoadd` is the per-rank driver of that leg; bench.py (N > 1), the coadder host (coadd_host.py, BASELINE
configs[3]) and the tests all go through it.
"""
import torch
import torch.distributed as dist


def antennas_of_rank(nant, rank, world):
    """Antenna indices owned by `rank`: a mod world == rank."""
    return [a for a in range(nant) if a % world == rank]


def tree_order(items):
    """The leaves of S over `items` (a list in index order) from left to right: even positions first, recursively.
    tree_order(range(8)) = [0, 4, 2, 6, 1, 5, 3, 7] (bit reversal); tree_order(range(5)) = [0, 4, 2, 1, 3].  The
    device evaluates T_n over leaves listed this way (include/pb_hip.h: pb_coadd_tree)."""
    items = list(items)
    if len(items) <= 1:
        return items
    return tree_order(items[0::2]) + tree_order(items[1::2])


def is_pow2(n):
    return n >= 1 and (n & (n - 1)) == 0


def reduce_to_root(t, root=0, group=None):
    """Sum `t` over ranks into the root's tensor (in place).  No-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM, group=group)
    return t


class IncoherentCoadd(object):
    """The incoherent-sum leg of one rank, pipelined behind the batches of a PbHandle.

    handle: PbHandle(keep_ave=True, nsets=n) holding this rank's antennas.  The leg has a stream of its own
    (pb_set_coadd_stream): local sum -> gather / reduce -> the root's sum and requantisation of batch k are ordered
    on it by the device and run beside the kernels of the batches after it; nothing here synchronises the host except
    the rehearsal back ends ("gloo", and "threads" = W ranks as threads of one process, threaded_ranks.py: the partial
    sums cross through host memory).

    order="tree" (default): the defined order of the module docstring -- pb_coadd_local_tree on every rank, a gather
    of the ranks' planes to the root, pb_coadd_tree there.  order="fast": pb_coadd_local (left to right over the
    rank's antennas) and one dist.reduce; the bytes then depend on the world size and the collective.

    With one local antenna and the in-library FFT the plane to be shipped IS the antenna's plane: detect writes it
    straight into this object's buffer of the batch's buffer set (pb_set_coadd_target) and the local step launches
    nothing.

    source="codes": the local sum is taken from the antennas' QUANTISED filterbank bytes (each code stands for the
    centre of its quantiser cell; pb_coadd_local_codes) -- what a coadder fed from the co rings has to work with.
    For like-for-like comparisons only: it always takes the "fast" route.

    queue(set_index, nseg): call once the batch in that buffer set is known to be complete on the device (its
    filterbank bytes have been fetched), i.e. one step behind the batch itself -- no device-side wait for detect is
    then queued (a pending cross-stream wait on detect's event was measured to cost the pipeline 0.12 ms per step).
    coadded(nseg, age): the root's view of the coadded filterbank bytes of the latest (age 0) / previous (age 1)
    queue() call, in pinned host memory (waits for that batch's requantisation only).
    """

    def __init__(self, handle, nant_total, device, root=0, backend="nccl", group=None, use_target=None, parts=7,
                 source="planes", order="tree", layout="auto", slice_world_of_one=False):
        if source not in ("planes", "codes"):
            raise ValueError("source must be 'planes' or 'codes'")
        if order not in ("tree", "fast"):
            raise ValueError("order must be 'tree' or 'fast'")
        if layout not in ("auto", "root", "sliced"):
            raise ValueError("layout must be 'auto', 'root' or 'sliced'")
        self.source = source               # "codes": sum the antennas' quantised bytes (pb_coadd_local_codes)
        if source == "codes":
            use_target = False
            order = "fast"
        self.order = order
        self.h = handle
        self.nant_total = int(nant_total)
        self.root = root
        self.backend = backend
        self.group = group
        self.parts = parts                 # timing experiments: 1 local sum, 2 collective, 4 root sum + requantisation
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if root != 0 and order == "tree":
            raise ValueError("the tree order is rooted at rank 0")
        if backend == "threads" and order == "fast" and self.world > 1:
            raise ValueError("the threaded rehearsal group has no dist.reduce: order 'fast' needs nccl or gloo")
        A, W, N = handle.nant, self.world, self.nant_total
        n = self.n = handle.max_seg * handle.ave_per_seg
        # what a rank ships: its node of the tree (one plane) when the world is a power of two, otherwise every
        # antenna's plane (padded to the largest rank's count: a gather moves equal pieces)
        self.node_per_rank = order == "fast" or is_pow2(W)
        self.ship = 1 if self.node_per_rank else -(-N // W)
        if order == "tree":
            # whatever the world size: a handle with another antenna count than the sharding gives this rank would
            # mis-address the root's leaves (a % W * ship + a // W) or, past `ship` planes, write beyond `sums`
            if len(antennas_of_rank(N, self.rank, W)) != A:
                raise ValueError("rank %d of %d holds %d antennas, the sharding a mod world gives it %d of %d"
                                 % (self.rank, W, A, len(antennas_of_rank(N, self.rank, W)), N))
            if max(A if self.node_per_rank else N, W) > 32:
                raise ValueError("the tree sum takes at most 32 leaves (PB_COADD_MAX_LEAVES)")
            self.local_order = tree_order(range(A))          # local antenna j is antenna rank + j * world
        if use_target is None:
            use_target = A == 1 and handle.cfg.fft_backend == 0
        self.sums = [torch.zeros(self.ship * n, dtype=torch.float32, device=device)
                     for _ in range(handle.nsets if use_target else 1)]
        self.use_target = bool(use_target)
        if self.use_target:
            for st in range(handle.nsets):
                handle.select_set(st)
                handle.set_coadd_target(self.sums[st].data_ptr())
            handle.select_set(0)
        # "sliced": a power-of-two world of more than one rank, the tree order, one output polarisation (a flat range
        # of the plane is then a range of the code stream).  slice_world_of_one (tests): also a world of ONE rank, so
        # that the whole sliced leg -- all-to-all, tree over the slices, pb_coadd_digitise, gather of the code bytes,
        # pb_coadd_publish, in stream order -- can run over RCCL on the one GPU a test box has.
        can_slice = (order == "tree" and (W > 1 or slice_world_of_one) and is_pow2(W)
                     and int(getattr(handle.cfg, "npol", 1)) == 1)
        if layout == "sliced" and not can_slice:
            raise ValueError("the sliced layout needs the tree order, a power-of-two world > 1 and npol = 1")
        if layout == "sliced" and parts != 7:
            raise ValueError("parts (timing experiments) isolates steps of the 'root' layout only: use layout='root'")
        if parts != 7 and layout == "auto":
            layout = "root"
        self.layout = "sliced" if (can_slice and layout != "root") else "root"
        self.gath = self.total = self.leaves = None
        if self.layout == "sliced":
            nbit = int(handle.cfg.nbit)
            self.recv = torch.zeros(n, dtype=torch.float32, device=device)            # W slices of n / W floats
            self.slice_sum = torch.zeros(n // W, dtype=torch.float32, device=device)
            self.slice_codes = torch.zeros(n // W * nbit // 8, dtype=torch.uint8, device=device)
            self.all_codes = (torch.zeros(n * nbit // 8, dtype=torch.uint8, device=device) if self.rank == root else None)
            self.nbit = nbit
        elif order == "tree" and W > 1 and self.rank == root:
            self.gath = torch.zeros(W * self.ship * n, dtype=torch.float32, device=device)
            self.total = torch.zeros(n, dtype=torch.float32, device=device)
            g0 = self.gath.data_ptr()
            if self.node_per_rank:
                self.leaves = [g0 + 4 * n * r for r in tree_order(range(W))]
            else:
                self.leaves = [g0 + 4 * n * ((a % W) * self.ship + a // W) for a in tree_order(range(N))]
        # (a CPU `device` exists for the host-logic tests: a stand-in handle, no stream, gloo)
        self.stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        handle.sync()
        handle.set_coadd_stream(self.stream.cuda_stream if self.stream is not None else 0)
        # bench.py --coadd-selftest --emulate-world W (one GPU): the ROOT's device work of a W-rank world -- the tree over
        # W gathered planes (stand-ins: copies of nothing in particular; results invalid) -- without the collective
        self.emulate = None
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

    def _collective(self, ds):
        """the ranks' planes to the root: dist.gather into self.gath (tree) / dist.reduce in place (fast)"""
        if self.backend == "nccl":
            if self.timing:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(self.stream)
            if self.order == "fast":
                dist.reduce(ds, dst=self.root, op=dist.ReduceOp.SUM, group=self.group)          # RCCL over xGMI
            else:
                into = list(self.gath.split(ds.numel())) if self.rank == self.root else None
                dist.gather(ds, into, dst=self.root, group=self.group)                          # RCCL send / recv
            if self.timing:
                ev[1].record(self.stream)
                self._pairs.append(ev)
            return
        # rehearsal (gloo): through host memory
        if self.stream is not None:
            self.stream.synchronize()
        t = ds.cpu()
        if self.order == "fast":
            dist.reduce(t, dst=self.root, op=dist.ReduceOp.SUM, group=self.group)
            ds.copy_(t)
        else:
            into = [torch.empty_like(t) for _ in range(self.world)] if self.rank == self.root else None
            dist.gather(t, into, dst=self.root, group=self.group)
            if into is not None:
                for r, piece in enumerate(into):
                    self.gath[r * t.numel():(r + 1) * t.numel()].copy_(piece)

    def _sliced(self, ds, nseg, nfl):
        """the ranks share the root's work: all-to-all of plane slices, the tree over the W slices received, the slice
        requantised here, code bytes gathered to the root (module docstring).  Runs on the leg's stream."""
        h, W = self.h, self.world
        sl = nfl // W                                   # floats per slice (nfl = nseg * ave_per_seg: a multiple of 4096)
        nb = sl * self.nbit // 8
        send, recv = ds[:nfl], self.recv[:nfl]
        timing = self.timing and self.backend == "nccl"
        if timing:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.stream)
        if self.backend == "nccl":
            dist.all_to_all_single(recv, send, group=self.group)                 # RCCL: every link of the mesh at once
        else:                                                                    # rehearsal (gloo): through host memory
            if self.stream is not None:
                self.stream.synchronize()
            t = send.cpu()
            r = torch.empty_like(t)
            dist.all_to_all_single(r, t, group=self.group)
            recv.copy_(r)
        g0 = recv.data_ptr()
        h.coadd_tree([g0 + 4 * sl * r for r in tree_order(range(W))], self.slice_sum.data_ptr(), sl)
        h.coadd_digitise(self.slice_sum.data_ptr(), sl, self.nant_total, self.slice_codes.data_ptr())
        mine = self.slice_codes[:nb]
        if self.backend == "nccl":
            into = list(self.all_codes[:W * nb].split(nb)) if self.rank == self.root else None
            dist.gather(mine, into, dst=self.root, group=self.group)
        else:
            if self.stream is not None:
                self.stream.synchronize()
            t = mine.cpu()
            into = [torch.empty_like(t) for _ in range(W)] if self.rank == self.root else None
            dist.gather(t, into, dst=self.root, group=self.group)
            if into is not None:
                for r, piece in enumerate(into):
                    self.all_codes[r * nb:(r + 1) * nb].copy_(piece)
        if timing:
            ev[1].record(self.stream)
            self._pairs.append(ev)
        if self.rank == self.root:
            h.coadd_publish(self.all_codes.data_ptr(), W * nb)

    def queue(self, set_index, nseg):
        h, ds = self.h, self.buffer(set_index)
        h.select_set(set_index)
        nfl = nseg * h.ave_per_seg
        final = ds
        with self._on_stream():
            if self.parts & 1:
                if self.source == "codes":
                    h.coadd_local_codes(nseg, ds.data_ptr())
                elif self.order == "fast":
                    h.coadd_local(nseg, ds.data_ptr())
                elif self.node_per_rank:
                    h.coadd_local_tree(nseg, self.local_order, ds.data_ptr())
                else:
                    for j in range(h.nant):
                        h.coadd_local_tree(nseg, [j], ds.data_ptr() + 4 * self.n * j)
            if self.layout == "sliced":
                self._sliced(ds, nseg, nfl)
                if self.use_target:
                    h.coadd_release()
                self.queued += 1
                return
            if self.parts & 2 and self.world > 1:
                self._collective(ds)
            if self.emulate is not None and self.emulate[0] == "sliced":
                _, leaves, tot, codes, W = self.emulate
                h.coadd_tree(leaves, tot.data_ptr(), nfl // W)
                h.coadd_digitise(tot.data_ptr(), nfl // W, self.nant_total, codes.data_ptr())
            elif self.emulate is not None:
                h.coadd_tree(self.emulate[1], self.emulate[2].data_ptr(), nfl)
                final = self.emulate[2]
            if self.rank == self.root and self.parts & 4:
                if self.order == "tree" and self.world > 1:
                    h.coadd_tree(self.leaves, self.total.data_ptr(), nfl)
                    final = self.total
                h.coadd_finish(nseg, final.data_ptr(), self.nant_total, blocking=False)
            if self.use_target:
                h.coadd_release()
        self.queued += 1

    def emulate_root_of(self, W, device, layout="root"):
        """timing only: queue() also runs the device work a W-rank world adds -- "root": rank 0's tree over W gathered
        planes; "sliced": EVERY rank's tree over W slices of 1/W of the plane and the slice's requantisation (then the
        coadded bytes are this handle's own, requantised by pb_coadd_finish as in a world of one)"""
        if layout == "sliced":
            g = torch.zeros(self.n, dtype=torch.float32, device=device)
            sl = self.n // W
            tot = torch.zeros(sl, dtype=torch.float32, device=device)
            codes = torch.zeros(sl, dtype=torch.uint8, device=device)
            self.emulate = ("sliced", [g.data_ptr() + 4 * sl * r for r in tree_order(range(W))], tot, codes, W)
            return
        g = torch.zeros(W * self.n, dtype=torch.float32, device=device)
        tot = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.emulate = (g, [g.data_ptr() + 4 * self.n * r for r in tree_order(range(W))], tot)

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
        """Detach from the handle: nothing of this object (its stream, its buffers) is referenced by the library
        afterwards, whatever is done with the handle later."""
        self.h.sync()
        if self.stream is not None:
            self.stream.synchronize()
        self.h.set_coadd_stream(0)
        if self.use_target:
            for st in range(self.h.nsets):
                self.h.select_set(st)
                self.h.set_coadd_target(0)
            self.h.select_set(0)
            self.use_target = False
