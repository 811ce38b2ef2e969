"""The fixed-order incoherent sum on the device (csrc/coadd_tree.hip through the C ABI): pb_coadd_tree over 1..32
caller-owned planes and pb_coadd_local_tree over a handle's own antennas, bit for bit against the order's DEFINITION
(helpers.parity_sum: antennas split by index parity, recursively -- DESIGN.md section 6), with planes whose fp32 sums
depend on the association (mixed magnitudes), and the emulation of an N-GPU run on one GPU: the partial sums an
8-, 4- or 2-rank world would ship, summed by the root's call, equal the single-GPU sum of all antennas.
Reference side this replaces: the external MPI coadder of scripts/start_coadd:16,20-58 (arithmetic unpinned)."""
import importlib

import numpy as np
import pytest

from helpers import count_tree, libpb, make_input, parity_sum  # noqa: F401

pytestmark = pytest.mark.gpu

coadd = importlib.import_module("vlite-fast_amd.coadd")


def _planes(n, nfloat, seed):
    rng = np.random.default_rng(seed)
    out = []
    for a in range(n):
        v = rng.standard_normal(nfloat).astype(np.float32) * np.float32(10.0 ** rng.integers(-2, 3))
        v[::97] = np.float32(-0.0)
        out.append(v)
    return out


def test_coadd_tree_every_leaf_count_is_the_defined_order():
    import torch
    lp = libpb()
    dev = torch.device("cuda", 0)
    nfloat = 4096 * 3 + 4
    with lp.PbHandle(device=0, nant=1, rows_per_seg=8, max_seg=1, keep_ave=True) as h:
        for N in range(1, 33):
            pl = _planes(N, nfloat, 100 + N)
            d = [torch.from_numpy(p).to(dev) for p in pl]
            dst = torch.full((nfloat,), float("nan"), dtype=torch.float32, device=dev)
            order = coadd.tree_order(range(N))
            h.coadd_tree([d[a].data_ptr() for a in order], dst.data_ptr(), nfloat)
            h.sync()
            torch.cuda.synchronize()
            got = dst.cpu().numpy()
            want = parity_sum(pl)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "N = %d" % N
            if N >= 3:          # and the association matters for these planes
                lr = pl[0]
                for p in pl[1:]:
                    lr = lr + p
                assert not np.array_equal(lr, want)
        # the plan of a W-rank world (W a power of two): every rank's node, then the root over the ranks' planes
        N = 16
        pl = _planes(N, nfloat, 7)
        d = [torch.from_numpy(p).to(dev) for p in pl]
        want = parity_sum(pl)
        for W in (1, 2, 4, 8, 16):
            parts = []
            for r in range(W):
                mine = coadd.antennas_of_rank(N, r, W)
                t = torch.empty(nfloat, dtype=torch.float32, device=dev)
                h.coadd_tree([d[mine[j]].data_ptr() for j in coadd.tree_order(range(len(mine)))], t.data_ptr(), nfloat)
                parts.append(t)
            tot = torch.empty(nfloat, dtype=torch.float32, device=dev)
            h.coadd_tree([parts[r].data_ptr() for r in coadd.tree_order(range(W))], tot.data_ptr(), nfloat)
            h.sync()
            torch.cuda.synchronize()
            assert np.array_equal(tot.cpu().numpy().view(np.uint32), want.view(np.uint32)), "W = %d" % W
        # refusals: too many leaves, a misaligned plane, the destination among the leaves
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_tree([d[0].data_ptr()] * 33, dst.data_ptr(), nfloat)
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_tree([d[0].data_ptr() + 4, d[1].data_ptr()], dst.data_ptr(), nfloat - 4)
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_tree([d[0].data_ptr(), dst.data_ptr()], dst.data_ptr(), nfloat)
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_tree([d[0].data_ptr(), d[1].data_ptr()], dst.data_ptr(), nfloat - 1)


@pytest.mark.parametrize("A", [1, 2, 3, 5])
def test_coadd_local_tree_over_a_handles_antennas(A):
    """a rank's node of the tree from the batch's own fp32 planes (what detect left in HBM), then the root's
    requantisation: sel_and_dig of the defined sum"""
    import torch
    lp = libpb()
    dev = torch.device("cuda", 0)
    R, S = 16, 2
    data = [make_input(300 + a, R, S, rfi=a % 2 == 0) for a in range(A)]
    with lp.PbHandle(device=0, nant=A, nbit=8, npol=1, rfi_mode=2, rows_per_seg=R, max_seg=S, keep_ave=True) as h:
        for a in range(A):
            for s in range(S):
                h.submit_planar(a, s, data[a][s, 0], data[a][s, 1])
        h.process(S)
        planes = [h.fetch(a, 0, S, raw=False, kur=False, ave=True)["ave_kur"] for a in range(A)]
        n = S * h.ave_per_seg
        dst = torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
        h.coadd_local_tree(S, coadd.tree_order(range(A)), dst.data_ptr())
        h.sync()
        torch.cuda.synchronize()
        want = parity_sum(planes)
        assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.view(np.uint32))
        codes = h.coadd_finish(S, dst.data_ptr(), A)
        v = want * np.float32(1.0 / np.sqrt(float(A)))
        tmp = (v.astype(np.float64) / 0.02957 + 127.5).astype(np.float32)
        ref = np.where(tmp <= 0, 0, np.where(tmp >= 255, 255, np.minimum(np.maximum(tmp, 0), 255).astype(np.uint8))).astype(np.uint8)
        assert np.array_equal(codes, ref)
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_local_tree(S, [A], dst.data_ptr())
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_local_tree(S + 1, [0], dst.data_ptr())


@pytest.mark.parametrize("nbit", [8, 4, 2])
def test_coadd_digitise_flat_range_and_publish(oracle, nbit):
    """pb_coadd_digitise on a flat slice of an npol = 1 plane = the oracle's sel_and_dig of the same samples (the slice a
    rank requantises in the sliced layout), at 8 / 4 / 2 bits, for slices that start and end inside segments; and
    pb_coadd_publish hands the assembled bytes out through pb_coadd_fetch_ptr.  npol = 2: refused."""
    import torch
    lp = libpb()
    dev = torch.device("cuda", 0)
    R, S = 16, 3
    rng = np.random.default_rng(9)
    with lp.PbHandle(device=0, nant=1, nbit=nbit, npol=1, rows_per_seg=R, max_seg=S, keep_ave=True) as h:
        n = S * h.ave_per_seg
        plane = (rng.standard_normal(n) * 3).astype(np.float32)
        plane[::501] = np.float32(100.0)
        plane[7::733] = np.float32(-100.0)
        nant = 5
        scale = np.float32(1.0 / np.sqrt(float(nant)))
        # the oracle quantises a full [ntime][6251] plane: put every segment's 4096 channels where it expects them
        want = b""
        for s_ in range(S):
            full = np.zeros((R // 8, oracle.NCHAN), np.float32)
            full[:, oracle.CHANMIN:oracle.CHANMIN + 4096] = (plane[s_ * h.ave_per_seg:(s_ + 1) * h.ave_per_seg] * scale).reshape(R // 8, 4096)
            want += oracle.sel_and_dig(full, R, npol=1, nbit=nbit).tobytes()
        want = np.frombuffer(want, np.uint8)
        d = torch.from_numpy(plane).to(dev)
        codes = torch.zeros(n * nbit // 8, dtype=torch.uint8, device=dev)
        W = 4
        sl = n // W
        for r in range(W):                      # four slices, as four ranks would requantise them
            h.coadd_digitise(d.data_ptr() + 4 * sl * r, sl, nant, codes.data_ptr() + sl * nbit // 8 * r)
        h.sync()
        torch.cuda.synchronize()
        assert np.array_equal(codes.cpu().numpy(), want)
        h.coadd_publish(codes.data_ptr(), codes.numel())
        assert np.array_equal(np.array(h.coadd_view(S, age=0), copy=True), want)
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_digitise(d.data_ptr(), 12, nant, codes.data_ptr())
        with pytest.raises((ValueError, lp.PbError)):
            h.coadd_publish(codes.data_ptr(), codes.numel() + 1)
    with lp.PbHandle(device=0, nant=1, nbit=8, npol=2, rows_per_seg=R, max_seg=1, keep_ave=True) as h2:
        with pytest.raises((ValueError, lp.PbError)):
            h2.coadd_digitise(d.data_ptr(), 4096, 1, codes.data_ptr())


def test_incoherent_coadd_leg_world1_rccl_gather_path():
    """coadd.IncoherentCoadd (the class bench.py and the coadder host run) in a world of ONE RCCL rank, two antennas:
    local tree -> (no collective) -> requantisation, pipelined one step late on the leg's stream; and with the
    collective forced (world 1 gather into the root's buffer) the same bytes."""
    import os
    import torch
    import torch.distributed as dist
    lp = libpb()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29250 + os.getpid() % 300)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        R, S, NSETS, NSTEP, A = 16, 4, 2, 3, 2
        data = [[make_input(170 + 10 * k + a, R, S) for a in range(A)] for k in range(NSTEP)]
        h = lp.PbHandle(device=0, nant=A, nbit=8, npol=1, rfi_mode=2, rows_per_seg=R, max_seg=S, keep_ave=True, nsets=NSETS)
        leg = coadd.IncoherentCoadd(h, A, dev, backend="nccl")
        assert leg.order == "tree" and not leg.use_target
        expect, coadded = [], []
        for k in range(NSTEP + 1):
            if k < NSTEP:
                h.select_set(k % NSETS)
                for a in range(A):
                    for s in range(S):
                        h.submit_planar(a, s, data[k][a][s, 0], data[k][a][s, 1])
                h.process(S)
            if k >= 1:
                h.select_set((k - 1) % NSETS)
                planes = [h.fetch(a, 0, S, raw=False, kur=False, ave=True)["ave_kur"] for a in range(A)]
                leg.queue((k - 1) % NSETS, S)
                v = parity_sum(planes) * np.float32(1.0 / np.sqrt(2.0))
                tmp = (v.astype(np.float64) / 0.02957 + 127.5).astype(np.float32)
                expect.append(np.where(tmp <= 0, 0, np.where(tmp >= 255, 255, np.clip(tmp, 0, 255).astype(np.uint8))).astype(np.uint8))
                if k >= 2:
                    coadded.append(np.array(leg.coadded(S, age=1), copy=True))
        coadded.append(np.array(leg.coadded(S, age=0), copy=True))
        for k in range(NSTEP):
            assert np.array_equal(coadded[k], expect[k]), "batch %d" % k
        # the RCCL gather itself (one rank: into the root's own buffer), then the root's tree over that one plane
        g = torch.zeros(S * h.ave_per_seg, dtype=torch.float32, device=dev)
        src = torch.arange(S * h.ave_per_seg, dtype=torch.float32, device=dev)
        dist.gather(src, [g], dst=0)
        torch.cuda.synchronize()
        assert torch.equal(g, src)
        # ... and the sliced layout's collectives as RCCL calls (one rank: identity): all_to_all_single on fp32, gather on uint8
        g2 = torch.zeros_like(src)
        dist.all_to_all_single(g2, src)
        cb = (torch.arange(4096, device=dev) % 251).to(torch.uint8)
        cg = torch.zeros_like(cb)
        dist.gather(cb, [cg], dst=0)
        torch.cuda.synchronize()
        assert torch.equal(g2, src) and torch.equal(cg, cb)
        leg.close()
        h.coadd_local_tree(S, [0, 1], g.data_ptr())          # the handle is usable after the leg has gone
        h.sync()
        h.close()
    finally:
        dist.destroy_process_group()
