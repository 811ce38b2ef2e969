"""The coadder host (vlite-fast_amd/coadd_host.py, BASELINE configs[3]) without a GPU: 2 / 3 / 5 / 8 ranks under gloo
-- 8 ranks x 16 antennas is configs[3]'s real shape -- the device replaced by a stand-in whose outputs are pure
functions of the samples it was given, so that every byte of every file can be predicted.  Under test: antenna a ->
rank a mod world (uneven shards included), streams aligned on their VDIF seconds (one antenna starts a second early,
one ends a second early), every antenna's own .fil / _kur.fil, the per-second incoherent-sum leg through
coadd.IncoherentCoadd (local tree -> dist.gather -> the root's tree and requantisation, collected one second late),
the single station-99 file with its SIGPROC header and the coadded ring.  The coadded bytes must equal the DEFINED
order of the fp32 additions (antennas split by index parity, recursively: DESIGN.md section 6) -- written down here
from that definition, not from the plan coadd.py derives from it -- and so be the SAME bytes for every world size.
Reference: scripts/start_coadd:16,20-58 (one coadder rank per antenna ring), src/process_baseband.cu:272-285
(station-99 name), :1416-1422 (coadd ring feed)."""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import synth
from helpers import count_tree, parity_sum

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, SEG = 8, 10
NSEC_SAMP = R * SEG * 12500
FPS = NSEC_SAMP // 5000
TRIM = 2 * R * 4096 // 16              # 8-bit, npol 1: bytes per segment == fp32 plane floats per segment
STATIONS = [7 + 3 * a for a in range(16)]


def _plane(codes):
    return (codes.astype(np.float32) - np.float32(128)) * np.float32(1.1)


def _mods():
    return [importlib.import_module("vlite-fast_amd." + m) for m in ("vdif", "sigproc", "dada", "coadd", "coadd_host")]


class FakeCoaddHandle(object):
    """Duck-typed PbHandle(keep_ave=True) with `nant` local antennas: host deframe; codes = the staged samples
    (pol 0 -> raw stream, pol 1 -> excised stream); fp32 plane = _plane(excised codes), values with full mantissas so
    that fp32 sums of them depend on the association; the "requantisation" keeps the LOW mantissa byte of the scaled
    sum, so that one ulp anywhere changes the file; pb_coadd_local / pb_coadd_tree / pb_coadd_finish act on the
    caller's buffers through their addresses, like the library."""

    def __init__(self, nant, nsets=2):
        self.nant, self.nsets, self.trim, self.max_seg, self.ave_per_seg = nant, nsets, TRIM, SEG, TRIM
        self.cfg = argparse.Namespace(fft_backend=1, npol=1, nbit=8)
        self.cur_set = 0
        self.staged = [[None] * nant for _ in range(nsets)]
        self.done = [[None] * nant for _ in range(nsets)]
        self.co = [None, None]
        self.co_last = 0
        self.calls = []
        self.vdif = importlib.import_module("vlite-fast_amd.vdif")

    def select_set(self, i):
        self.cur_set = i

    def reset_history(self, ant):
        pass

    def sync(self):
        pass

    def set_coadd_stream(self, s):
        pass

    def submit_vdif(self, ant, seg0, block, second=None, frame0=0):
        assert self.staged[self.cur_set][ant] is None, "buffer set refilled before it was processed"
        self.staged[self.cur_set][ant] = self.vdif.deframe_block(np.array(block, copy=True), second, frame0)
        self.calls.append(("submit", self.cur_set, ant, second))

    def process(self, nseg, inject_now=0):
        assert all(s is not None for s in self.staged[self.cur_set])
        self.done[self.cur_set], self.staged[self.cur_set] = self.staged[self.cur_set], [None] * self.nant
        self.calls.append(("process", self.cur_set))

    def fetch(self, ant, seg0, nseg, raw=True, kur=True, **kw):
        d = self.done[self.cur_set][ant]
        return dict(raw=d[0, :nseg * TRIM].copy() if raw else None, kur=d[1, :nseg * TRIM].copy() if kur else None)

    def _mem(self, ptr, n):
        return np.ctypeslib.as_array((C.c_float * n).from_address(ptr))

    def coadd_local(self, nseg, ptr, accumulate=False):
        m = self._mem(ptr, nseg * TRIM)
        s = np.zeros(nseg * TRIM, np.float32)
        for a in range(self.nant):
            s = s + _plane(self.done[self.cur_set][a][1, :nseg * TRIM])
        m[:] = s
        self.calls.append(("coadd_local", self.cur_set))

    def coadd_local_tree(self, nseg, order, ptr):
        m = self._mem(ptr, nseg * TRIM)
        m[:] = count_tree([_plane(self.done[self.cur_set][a][1, :nseg * TRIM]) for a in order])
        self.calls.append(("coadd_local_tree", self.cur_set, tuple(order)))

    def coadd_tree(self, leaf_ptrs, dst, nfloat):
        v = count_tree([self._mem(p, nfloat).copy() for p in leaf_ptrs])
        self._mem(dst, nfloat)[:] = v
        self.calls.append(("coadd_tree", len(leaf_ptrs)))

    def coadd_finish(self, nseg, ptr, nant_total, blocking=True):
        m = self._mem(ptr, nseg * TRIM)
        v = m * np.float32(1.0 / np.sqrt(float(nant_total)))
        slot = self.co_last ^ 1
        self.co[slot] = (v.view(np.uint32) & np.uint32(0xFF)).astype(np.uint8)
        self.co_last = slot

    def coadd_digitise(self, ptr, nfloat, nant_total, codes_ptr):
        v = self._mem(ptr, nfloat) * np.float32(1.0 / np.sqrt(float(nant_total)))
        out = np.ctypeslib.as_array((C.c_uint8 * nfloat).from_address(codes_ptr))
        out[:] = (v.view(np.uint32) & np.uint32(0xFF)).astype(np.uint8)
        self.calls.append(("coadd_digitise", nfloat))

    def coadd_publish(self, codes_ptr, nbytes):
        slot = self.co_last ^ 1
        self.co[slot] = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(codes_ptr)).copy()
        self.co_last = slot
        self.calls.append(("coadd_publish", nbytes))

    def coadd_view(self, nseg, age=0):
        return self.co[self.co_last if age == 0 else self.co_last ^ 1]

    def close(self):
        pass


def _second(ant, sec):
    p0 = synth.baseband_u8(9000 + 100 * ant + 2 * (sec - 3590), NSEC_SAMP)
    p1 = synth.baseband_u8(9001 + 100 * ant + 2 * (sec - 3590), NSEC_SAMP)
    p0[p0 == 0] = 1
    p1[p1 == 0] = 1
    return p0, p1


def _seconds_of(ant):
    """antenna 1 starts one second early, antenna 2 ends one second early"""
    first = 3599 if ant == 1 else 3600
    last = 3602 if ant == 2 else 3603
    return list(range(first, last + 1))


def _stream(ant):
    vdif = importlib.import_module("vlite-fast_amd.vdif")
    secs = _seconds_of(ant)
    hdr = vdif.ascii_header_format(vdif.writer_header(STATIONS[ant], 0.8718, -0.72452, "B0833-45", 58000.0, "19A-331", 33, secs[0]))
    body = b"".join(vdif.frame_block(*_second(ant, s), s, 33, STATIONS[ant]).tobytes() for s in secs)
    return hdr, body


def _worker(rank, world, port, tmp, NANT, order, layout="auto", transport="gloo", dist=None):
    import torch
    if transport == "gloo":
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    vdif, sigproc, dada, coadd, host = _mods()
    args = host.build_parser().parse_args(["--replay"] + ["unused"] * NANT + ["-b", "8", "-r", "2", "-w", "2", "--datadir", tmp,
                                           "--logdir", os.path.join(tmp, "logs"), "--rows-per-seg", str(R),
                                           "--dist-backend", transport, "--coadd-order", order, "--coadd-layout", layout,
                                           "--out-sink", os.path.join(tmp, "co_ring.bin")])
    mine = coadd.antennas_of_rank(NANT, rank, world)
    rings = {}
    for a in mine:
        hdr, body = _stream(a)
        r = dada.MemoryRing()
        r.write_header(hdr)
        r.write(np.frombuffer(body, np.uint8))
        r.end_of_data()
        rings[a] = r
    h = FakeCoaddHandle(len(mine), nsets=2)
    co = coadd.IncoherentCoadd(h, NANT, torch.device("cpu"), root=0, backend=transport, order=order, layout=layout)
    rc = host.run(args, rank=rank, world=world, local=0, rings=rings, handle=h, dist=dist, device=torch.device("cpu"), coadd=co)
    with open(os.path.join(tmp, "rc%d" % rank), "w") as f:
        f.write("%d %s" % (rc, [c for c in h.calls if c[0] in ("submit", "coadd_tree", "coadd_digitise", "coadd_publish")] + [("layout", co.layout)]))
    dist.barrier()
    if transport == "gloo":
        dist.destroy_process_group()


def _run_world(tmp_path, world, NANT, order, layout="auto", transport="gloo"):
    os.environ["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"),
                                                os.environ.get("PYTHONPATH", "")])
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    tmp = str(tmp_path)
    if transport == "threads":
        # the ranks as threads of THIS process (vlite-fast_amd/threaded_ranks.py): the rehearsal transport for boxes
        # that allow fewer processes on a card than configs[3] has ranks
        tr = importlib.import_module("vlite-fast_amd.threaded_ranks")
        tr.run_as_threads(world, lambda rank, w, dist: _worker(rank, w, 0, tmp, NANT, order, layout, "threads", dist), timeout=300)
        for r in range(world):
            assert (tmp_path / ("rc%d" % r)).read_text().startswith("0 ")
        return
    ctx = mp.get_context("spawn")
    port = 29900 + (os.getpid() * 7 + world * 13 + NANT) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, tmp, NANT, order, layout)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for r in range(world):
        assert (tmp_path / ("rc%d" % r)).read_text().startswith("0 ")


def _expected(NANT, secs):
    """per-antenna file bodies and fp32 planes over the coadded seconds"""
    raws, kurs, planes = [], [], []
    for a in range(NANT):
        raws.append(b"".join(_second(a, s)[0][:SEG * TRIM].tobytes() for s in secs))
        kurs.append(b"".join(_second(a, s)[1][:SEG * TRIM].tobytes() for s in secs))
        planes.append(_plane(np.concatenate([_second(a, s)[1][:SEG * TRIM] for s in secs])))
    return raws, kurs, planes


def _quantise(tot, NANT):
    v = tot * np.float32(1.0 / np.sqrt(float(NANT)))
    return (v.view(np.uint32) & np.uint32(0xFF)).astype(np.uint8).tobytes()


@pytest.mark.parametrize("world,NANT,layout,transport", [(8, 16, "auto", "gloo"), (8, 16, "root", "gloo"), (3, 7, "auto", "gloo"),
                                                         (5, 11, "auto", "gloo"), (4, 6, "sliced", "gloo"), (2, 4, "root", "gloo"),
                                                         (8, 16, "auto", "threads"), (8, 16, "root", "threads"),
                                                         (3, 7, "auto", "threads")])
def test_coadd_host_tree_order_files_and_sum(tmp_path, world, NANT, layout, transport):
    """(8, 16): two antennas per rank, configs[3] -- with the root's work spread over the ranks ("auto" -> "sliced":
    all-to-all of plane slices, every rank sums and requantises an eighth, code bytes gathered) and with every plane
    gathered to rank 0 ("root"): the SAME bytes; (3, 7) and (5, 11): worlds that are not powers of two ship every
    antenna's plane, shards 3/2/2 and 3/2/2/2/2; (4, 6): a power of two with uneven shards 2/2/1/1.
    transport "threads": the same ranks as threads of one process over torch's in-process group -- what the GPU test of
    the 8 x 16 shape uses, where eight processes on one card are not allowed (tests/test_gpu_coadd_host.py)."""
    _run_world(tmp_path, world, NANT, "tree", layout, transport)
    vdif, sigproc, dada, coadd, host = _mods()
    # the sum covers seconds 3600 and 3601: antenna 1's early second is skipped, antenna 2's stream ends with 3602,
    # which -- being its last -- is dropped
    secs = [3600, 3601]
    dmjd = 57570 + 3600 / 86400.
    raws, kurs, planes = _expected(NANT, secs)
    for a in range(NANT):
        hdr = sigproc.sigproc_header(STATIONS[a], 0.8718, -0.72452, "B0833-45", dmjd, 1, 8)
        assert (tmp_path / ("20160701_010000_muos_ea%02d.fil" % STATIONS[a])).read_bytes() == hdr + raws[a], a
        assert (tmp_path / ("20160701_010000_muos_ea%02d_kur.fil" % STATIONS[a])).read_bytes() == hdr + kurs[a], a
    want = _quantise(parity_sum(planes), NANT)
    co_hdr = sigproc.sigproc_header(99, 0.8718, -0.72452, "B0833-45", dmjd, 1, 8)
    assert (tmp_path / "20160701_010000_muos_ea99_kur.fil").read_bytes() == co_hdr + want
    ring = (tmp_path / "co_ring.bin").read_bytes()
    rh = vdif.ascii_header_parse(ring[:4096])
    assert rh["STATIONID"] == "99" and rh["SIGPROC_FILE"].endswith("_muos_ea99_kur.fil") and rh["NBIT"] == "8"
    assert ring[4096:] == want
    # every rank submitted its own antennas only, second by second in lockstep; the root summed one plane per rank
    # (power-of-two world) or one per antenna
    calls0 = eval((tmp_path / "rc0").read_text().split(" ", 1)[1])
    n0 = len(coadd.antennas_of_rank(NANT, 0, world))
    assert [(c[2], c[3]) for c in calls0 if c[0] == "submit"] == [(i, s) for s in secs for i in range(n0)]
    assert [c[1] for c in calls0 if c[0] == "coadd_tree"] == [world if coadd.is_pow2(world) else NANT] * len(secs)
    sliced = coadd.is_pow2(world) and layout != "root"
    assert ("layout", "sliced" if sliced else "root") in calls0
    nfl = SEG * TRIM
    assert [c[1] for c in calls0 if c[0] == "coadd_digitise"] == ([nfl // world] * len(secs) if sliced else [])
    assert [c[1] for c in calls0 if c[0] == "coadd_publish"] == ([nfl] * len(secs) if sliced else [])
    if sliced:          # every rank did its share: the tree over `world` slices and its slice's requantisation
        calls1 = eval((tmp_path / ("rc%d" % (world - 1))).read_text().split(" ", 1)[1])
        assert [c[1] for c in calls1 if c[0] == "coadd_tree"] == [world] * len(secs)
        assert [c[1] for c in calls1 if c[0] == "coadd_digitise"] == [nfl // world] * len(secs)
        assert not [c for c in calls1 if c[0] == "coadd_publish"]
    # the order really matters for these data (else the test could not tell a wrong plan from a right one)
    naive = np.float32(0)
    for pl in planes:
        naive = naive + pl
    if NANT > 2:
        assert _quantise(naive, NANT) != want


def test_tree_plan_matches_definition():
    """coadd.tree_order + T_n (the device's shape) == S(0, 1) for every antenna count, and the two-level plan of a
    power-of-two world (local trees, then the ranks' partial sums in tree order) is the same association"""
    coadd = _mods()[3]
    rng = np.random.default_rng(11)
    for N in range(1, 33):
        planes = [(rng.standard_normal(257) * 10.0 ** rng.integers(-3, 4)).astype(np.float32) for _ in range(N)]
        want = parity_sum(planes)
        assert np.array_equal(count_tree([planes[a] for a in coadd.tree_order(range(N))]), want), N
        for W in (1, 2, 4, 8, 16, 32):
            if W > N:
                continue
            parts = []
            for r in range(W):
                mine = coadd.antennas_of_rank(N, r, W)
                parts.append(count_tree([planes[mine[j]] for j in coadd.tree_order(range(len(mine)))]))
            assert np.array_equal(count_tree([parts[r] for r in coadd.tree_order(range(W))]), want), (N, W)
    assert coadd.tree_order(range(8)) == [0, 4, 2, 6, 1, 5, 3, 7] and coadd.tree_order(range(5)) == [0, 4, 2, 1, 3]


def test_coadd_host_fast_order_world2(tmp_path):
    """`--coadd-order fast`: left-to-right local sums and ONE dist.reduce; with two ranks the association is known"""
    NANT = 4
    _run_world(tmp_path, 2, NANT, "fast")
    sigproc = _mods()[1]
    raws, kurs, planes = _expected(NANT, [3600, 3601])
    # rank 0 holds antennas 0 and 2, rank 1 antennas 1 and 3; fp32 sums in that order, 1 / sqrt(4)
    tot = ((np.float32(0) + planes[0]) + planes[2]) + ((np.float32(0) + planes[1]) + planes[3])
    co_hdr = sigproc.sigproc_header(99, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, 8)
    assert (tmp_path / "20160701_010000_muos_ea99_kur.fil").read_bytes() == co_hdr + _quantise(tot, NANT)


def test_coadd_host_refuses_colliding_station_ids(tmp_path, monkeypatch):
    """two streams with one STATIONID would open the same .fil twice: refused (exit 1, nothing coadded) unless -w 0"""
    vdif, sigproc, dada, coadd, host = _mods()
    import torch
    monkeypatch.setattr(sys.modules[__name__], "STATIONS", [7, 7])
    for wflag, want_rc in (("2", 1), ("0", 0)):
        d = tmp_path / wflag
        d.mkdir()
        args = host.build_parser().parse_args(["--replay", "a", "b", "-b", "8", "-r", "2", "-w", wflag, "--datadir", str(d),
                                               "--logdir", str(d / "logs"), "--rows-per-seg", str(R), "--dist-backend", "gloo"])
        rings = {}
        for a in range(2):
            hdr, body = _stream(a)
            r = dada.MemoryRing()
            r.write_header(hdr)
            r.write(np.frombuffer(body, np.uint8))
            r.end_of_data()
            rings[a] = r
        h = FakeCoaddHandle(2, nsets=2)
        co = coadd.IncoherentCoadd(h, 2, torch.device("cpu"), backend="gloo")
        rc = host.run(args, rank=0, world=1, rings=rings, handle=h, device=torch.device("cpu"), coadd=co)
        assert rc == want_rc
        assert (d / "20160701_010000_muos_ea99_kur.fil").exists() == (want_rc == 0)
        assert not (d / "20160701_010000_muos_ea07.fil").exists()


def test_coadd_host_honours_the_source_list_with_w1(tmp_path, monkeypatch):
    """-w 1 (src/process_baseband.cu:880-923, per antenna here): an antenna whose source is on the site's list gets
    its .fil / _kur.fil, one whose source is not writes to /dev/null; the coadded file is written either way."""
    vdif, sigproc, dada, coadd, host = _mods()
    import torch
    lst = tmp_path / "allow.txt"
    lst.write_text("name B0833\n")
    monkeypatch.setenv("PB_WRITE_ALLOW", str(lst))
    args = host.build_parser().parse_args(["--replay", "a", "b", "-b", "8", "-r", "2", "-w", "1", "--datadir", str(tmp_path),
                                           "--logdir", str(tmp_path / "logs"), "--rows-per-seg", str(R), "--dist-backend", "gloo"])
    rings = {}
    for a in range(2):
        hdr, body = _stream(a)
        if a == 1:
            hdr = hdr.replace(b"B0833-45", b"J0000+00")
            assert b"J0000+00" in hdr and len(hdr) == 4096
        r = dada.MemoryRing()
        r.write_header(hdr)
        r.write(np.frombuffer(body, np.uint8))
        r.end_of_data()
        rings[a] = r
    h = FakeCoaddHandle(2, nsets=2)
    co = coadd.IncoherentCoadd(h, 2, torch.device("cpu"), backend="gloo")
    assert host.run(args, rank=0, world=1, rings=rings, handle=h, device=torch.device("cpu"), coadd=co) == 0
    assert (tmp_path / ("20160701_010000_muos_ea%02d_kur.fil" % STATIONS[0])).stat().st_size > 1000
    assert not (tmp_path / ("20160701_010000_muos_ea%02d_kur.fil" % STATIONS[1])).exists()
    assert not (tmp_path / ("20160701_010000_muos_ea%02d.fil" % STATIONS[1])).exists()
    assert (tmp_path / "20160701_010000_muos_ea99_kur.fil").stat().st_size > 1000
    log = "".join(p.read_text() for p in (tmp_path / "logs").iterdir())
    assert "matches target list" in log and "not on target list" in log


def test_incoherent_coadd_source_switch():
    """IncoherentCoadd(source=...): "planes" queues pb_coadd_local, "codes" pb_coadd_local_codes (and never the
    coadd-target shortcut, which hands detect's fp32 plane to the reduce); anything else is refused.  One process,
    no process group: the local sum and the root's requantisation only."""
    coadd = importlib.import_module("vlite-fast_amd.coadd")

    class H(FakeCoaddHandle):
        def coadd_local_codes(self, nseg, ptr, accumulate=False):
            # each code as the centre of its cell, (q - 127) * 0.02957 (include/pb_hip.h)
            m = self._mem(ptr, nseg * TRIM)
            s = np.zeros(nseg * TRIM, np.float32)
            for a in range(self.nant):
                q = self.done[self.cur_set][a][1, :nseg * TRIM].astype(np.float64)
                s = s + ((q - 127.0) * 0.02957).astype(np.float32)
            m[:] = s
            self.calls.append(("coadd_local_codes", self.cur_set))

        def set_coadd_target(self, ptr):
            self.calls.append(("target", ptr))

    rng = np.random.default_rng(5)
    for source in ("planes", "codes"):
        h = H(2)
        h.cfg.fft_backend = 0
        h.done[0] = [rng.integers(0, 256, (2, SEG * TRIM), dtype=np.uint8) for _ in range(2)]
        c = coadd.IncoherentCoadd(h, 2, "cpu", backend="gloo", source=source)
        assert not c.use_target
        out = c.step(SEG)
        assert out is not None and out.size == SEG * TRIM
        names = [x[0] for x in h.calls]
        assert ("coadd_local_codes" in names) == (source == "codes") and ("coadd_local_tree" in names) == (source == "planes")
        assert "coadd_local" not in names and c.order == ("tree" if source == "planes" else "fast")
        c.close()
    h1 = H(1)
    h1.cfg.fft_backend = 0
    assert coadd.IncoherentCoadd(h1, 1, "cpu", backend="gloo").use_target            # one antenna, planes: the shortcut
    h1 = H(1)
    h1.cfg.fft_backend = 0
    assert not coadd.IncoherentCoadd(h1, 1, "cpu", backend="gloo", source="codes").use_target
    with pytest.raises(ValueError):                          # one rank cannot hold 1 of 4 antennas under a mod world
        coadd.IncoherentCoadd(H(1), 4, "cpu", backend="gloo")
    with pytest.raises(ValueError):
        coadd.IncoherentCoadd(H(1), 1, "cpu", source="sum")
    with pytest.raises(ValueError):
        coadd.IncoherentCoadd(H(1), 1, "cpu", order="ring")
    with pytest.raises(ValueError):
        coadd.IncoherentCoadd(H(1), 1, "cpu", layout="sliced")          # a world of one has nothing to slice
    assert coadd.IncoherentCoadd(H(1), 1, "cpu", backend="gloo").layout == "root"
    hf = H(2)
    hf.done[0] = [rng.integers(0, 256, (2, SEG * TRIM), dtype=np.uint8) for _ in range(2)]
    cf = coadd.IncoherentCoadd(hf, 2, "cpu", backend="gloo", order="fast")
    cf.step(SEG)
    assert "coadd_local" in [x[0] for x in hf.calls]


def test_threaded_ranks_transport_and_the_leg_checks():
    """The rehearsal transport itself (vlite-fast_amd/threaded_ranks.py): ranks see their own rank, collectives move what
    torch.distributed says they move, a failing rank's exception reaches the caller (and does not hang the others) --
    and IncoherentCoadd's checks inside such a world: a handle with another antenna count than a mod world gives the
    rank is refused for EVERY world size (it would mis-address the root's leaves, or write past the shipped planes, in a
    world that is not a power of two), `parts` (timing experiments) never runs the sliced layout, the threaded group
    offers no dist.reduce so order="fast" is refused there."""
    import torch
    tr = importlib.import_module("vlite-fast_amd.threaded_ranks")
    coadd = importlib.import_module("vlite-fast_amd.coadd")

    def body(rank, world, dist):
        assert dist.get_rank() == rank and dist.get_world_size() == world and dist.get_backend() == "threaded"
        t = torch.arange(world * 2, dtype=torch.float32) + 100 * rank
        o = torch.empty_like(t)
        dist.all_to_all_single(o, t)
        assert o.tolist() == [100 * s + 2 * rank + i for s in range(world) for i in range(2)]
        mine = torch.full((3,), rank, dtype=torch.uint8)
        into = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, into, dst=0)
        out = {"gathered": [int(x[0]) for x in into] if rank == 0 else None}
        # the leg's checks (world 3: not a power of two -> every antenna's plane is shipped)
        n_mine = len(coadd.antennas_of_rank(7, rank, world))
        ok = coadd.IncoherentCoadd(FakeCoaddHandle(n_mine), 7, "cpu", backend="threads")
        out["layout"], out["ship"] = ok.layout, ok.ship
        for bad_n in (n_mine + 1, n_mine - 1):
            if bad_n >= 1:
                with pytest.raises(ValueError, match="sharding"):
                    coadd.IncoherentCoadd(FakeCoaddHandle(bad_n), 7, "cpu", backend="threads")
        with pytest.raises(ValueError, match="dist.reduce"):
            coadd.IncoherentCoadd(FakeCoaddHandle(n_mine), 7, "cpu", backend="threads", order="fast")
        dist.barrier()
        return out

    res = tr.run_as_threads(3, body, timeout=120)
    assert res[0]["gathered"] == [0, 1, 2] and all(r["layout"] == "root" and r["ship"] == 3 for r in res)

    def body4(rank, world, dist):
        h = FakeCoaddHandle(1)
        with pytest.raises(ValueError, match="parts"):
            coadd.IncoherentCoadd(h, 4, "cpu", backend="threads", layout="sliced", parts=1)
        a = coadd.IncoherentCoadd(FakeCoaddHandle(1), 4, "cpu", backend="threads", parts=1)       # auto + parts -> root
        b = coadd.IncoherentCoadd(FakeCoaddHandle(1), 4, "cpu", backend="threads")
        return a.layout, b.layout

    assert tr.run_as_threads(4, body4, timeout=120) == [("root", "sliced")] * 4

    def failing(rank, world, dist):
        if rank == 1:
            raise KeyError("rank one gives up")
        dist.barrier()                       # the others wait here; the group's termination event releases them
        return rank

    with pytest.raises(RuntimeError, match="threaded rank 1 failed"):
        tr.run_as_threads(3, failing, timeout=120)
    # and the process-wide torch.distributed state is as it was: no group left installed, and a run after a failed one
    # works (the group's termination event does not outlive the run that set it)
    import torch.distributed as dist
    assert not dist.is_initialized()
    assert tr.run_as_threads(3, lambda rank, world, d: (d.barrier(), rank)[1], timeout=120) == [0, 1, 2]
