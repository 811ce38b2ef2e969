"""End to end on the GPU: a genbase-style dump replayed through the process_baseband host
(vlite-fast_amd/process_baseband.py -> libpb_hip.so) must give .fil / _kur.fil files that are
byte-identical to the oracle's chain, including the reference's habit of dropping the last
second; and the VDIF gather on the GPU must equal the host demux."""
import importlib
import os

import numpy as np
import pytest

from helpers import libpb, make_input, oracle_run

pytestmark = pytest.mark.gpu

vdif = importlib.import_module("vlite-fast_amd.vdif")
sigproc = importlib.import_module("vlite-fast_amd.sigproc")
dada = importlib.import_module("vlite-fast_amd.dada")
pbmod = importlib.import_module("vlite-fast_amd.process_baseband")

R = 8
SEG = 10
FPS = R * SEG * 12500 // 5000       # frames per thread per (shortened) second = 200
T_UNIX = 1467334800                 # 2016-07-01 01:00:00 UTC -> epoch 33, second 3600


def _dump(path, data, station=7):
    """data: u8 [nsec*SEG][2][R*12500] -> dump file with one observation."""
    nsec = data.shape[0] // SEG
    hdr = vdif.writer_header(station, 0.8718, -0.72452, "B0833-45", 58000.0, "19A-331", 33, 3600)
    with open(path, "wb") as f:
        f.write(vdif.ascii_header_format(hdr))
        for s in range(nsec):
            p0 = np.concatenate([data[s * SEG + i, 0] for i in range(SEG)])
            p1 = np.concatenate([data[s * SEG + i, 1] for i in range(SEG)])
            f.write(vdif.frame_block(p0, p1, 3600 + s, 33, station).tobytes())


@pytest.mark.parametrize("nbit", [8, 2])
def test_replay_to_fil_is_byte_exact(tmp_path, oracle, nbit):
    nsec = 4
    data = make_input(11, R, nsec * SEG)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data)
    argv = ["-k", "40", "-K", "0", "-w", "2", "-b", str(nbit), "-P", "1", "-r", "2", "-g", "0", "-p", "0",
            "--replay", dump, "--datadir", str(tmp_path), "--logdir", str(tmp_path / "logs"), "--no-control",
            "--rows-per-seg", str(R), "--co-sink", str(tmp_path / "co.bin"), "--out-sink", str(tmp_path / "out.bin")]
    args = pbmod.build_parser().parse_args(argv)
    assert pbmod.run(args) == 0
    fb = tmp_path / "20160701_010000_muos_ea07.fil"
    fbk = tmp_path / "20160701_010000_muos_ea07_kur.fil"
    assert fb.exists() and fbk.exists()
    # oracle: the first nsec-1 seconds only (the reference drops the last one)
    nseg = (nsec - 1) * SEG
    res, _, _ = oracle_run(oracle, data[:nseg], R, rfi_mode=2, npol=1, nbit=nbit)
    hdr = sigproc.sigproc_header(7, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, nbit)
    assert fb.read_bytes() == hdr + b"".join(r.codes_raw.tobytes() for r in res)
    assert fbk.read_bytes() == hdr + b"".join(r.codes_kur.tobytes() for r in res)
    # coadd ring stand-in: header block + one write per segment of the excised stream
    co = (tmp_path / "co.bin").read_bytes()
    ch = vdif.ascii_header_parse(co[:4096])
    assert ch["NCHAN"] == "4096" and ch["NBIT"] == str(nbit) and ch["SIGPROC_FILE"].endswith("_muos_ea99_kur.fil")
    assert co[4096:] == b"".join(r.codes_kur.tobytes() for r in res)
    # heimdall ring stand-in: nothing before 10 s have been integrated (:1482-1494)
    assert len((tmp_path / "out.bin").read_bytes()) == 4096
    log = next((tmp_path / "logs").glob("*_process_*.log")).read_text()
    assert "Wrote" in log and "Proc Time" in log


def test_replay_through_a_fifo_equals_file_replay(tmp_path):
    """The live-ring stand-in: the same observation delivered through a FIFO in odd-sized pieces gives
    the same .fil bytes as the file replay."""
    import threading
    nsec = 3
    data = make_input(13, R, nsec * SEG)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data)
    outs = []
    for mode in ("file", "fifo"):
        d = tmp_path / mode
        d.mkdir()
        src = dump
        th = None
        if mode == "fifo":
            src = str(tmp_path / "ring.fifo")
            os.mkfifo(src)
            blob = open(dump, "rb").read()

            def feed():
                with open(src, "wb", buffering=0) as f:
                    for i in range(0, len(blob), 70001):
                        f.write(blob[i:i + 70001])

            th = threading.Thread(target=feed, daemon=True)
            th.start()
        argv = ["-b", "8", "-w", "2", "-r", "2", "--replay", src, "--datadir", str(d), "--logdir", str(d / "logs"),
                "--no-control", "--rows-per-seg", str(R)]
        assert pbmod.run(pbmod.build_parser().parse_args(argv)) == 0
        if th:
            th.join(timeout=20)
            assert not th.is_alive()
        outs.append([(d / n).read_bytes() for n in ("20160701_010000_muos_ea07.fil", "20160701_010000_muos_ea07_kur.fil")])
    assert outs[0] == outs[1] and len(outs[0][0]) > 1000


def test_out_ring_cadence_10s_then_1s(tmp_path, oracle):
    nsec = 13
    data = make_input(12, R, nsec * SEG, rfi=False, dropped=False)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data)
    sink = dada.FileSink(str(tmp_path / "out.bin"))
    args = pbmod.build_parser().parse_args(["-b", "8", "-w", "0", "--replay", dump, "--datadir", str(tmp_path),
                                            "--logdir", str(tmp_path), "--no-control", "--rows-per-seg", str(R)])
    assert pbmod.run(args, out_ring=sink) == 0
    trim = 2 * R * 4096 // 16
    assert sink.nwrites == [10 * SEG * trim, SEG * trim, SEG * trim]       # 12 s processed of 13


def test_gpu_vdif_gather_equals_host_demux():
    lp = libpb()
    data = make_input(13, R, SEG)
    p0 = np.concatenate([data[i, 0] for i in range(SEG)])
    p1 = np.concatenate([data[i, 1] for i in range(SEG)])
    blk = vdif.frame_block(p0, p1, 3600, 33, 7).reshape(-1, 5032).copy()
    rng = np.random.default_rng(0)
    perm = rng.permutation(blk.shape[0])
    perm = np.concatenate([[0], perm[perm != 0]])                  # the first frame anchors the block
    shuffled = blk[perm].copy()
    victim = 17
    shuffled[victim, 3] |= 0x80                                    # invalid-data bit -> zero fill
    ref = vdif.deframe_block(shuffled.ravel())
    with lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=SEG) as h1, lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=SEG) as h2:
        h1.submit_vdif(0, 0, shuffled.ravel())
        h1.process(SEG)
        a = h1.fetch(0, 0, SEG)
        n = R * 12500
        for s in range(SEG):
            h2.submit_planar(0, s, ref[0, s * n:(s + 1) * n], ref[1, s * n:(s + 1) * n])
        h2.process(SEG)
        b = h2.fetch(0, 0, SEG)
    assert np.array_equal(a["raw"], b["raw"]) and np.array_equal(a["kur"], b["kur"])
    assert (ref == 0).sum() >= 5000


def test_replay_with_dropped_frames_realigns(tmp_path, oracle):
    """A frame lost in the middle of second 1 and the first frame of second 2 lost: the .fil files must equal
    the oracle's over the same stream with exactly those frames zero-filled (the reference places every
    frame by its own header and loses nothing else, src/process_baseband.cu:1017-1034)."""
    nsec = 5
    data = make_input(17, R, nsec * SEG, rfi=False, dropped=False)
    hdr = vdif.writer_header(7, 0.8718, -0.72452, "B0833-45", 58000.0, "19A-331", 33, 3600)
    dump = str(tmp_path / "obs.uw")
    holes = {(1, 1, 57), (2, 0, 0)}                      # (second, thread, frame)
    zeroed = data.copy().reshape(nsec, SEG, 2, R * 12500)
    with open(dump, "wb") as f:
        f.write(vdif.ascii_header_format(hdr))
        for s in range(nsec):
            p0 = np.concatenate([data[s * SEG + i, 0] for i in range(SEG)])
            p1 = np.concatenate([data[s * SEG + i, 1] for i in range(SEG)])
            blk = vdif.frame_block(p0, p1, 3600 + s, 33, 7).reshape(FPS, 2, 5032)
            keep = np.ones((FPS, 2), bool)
            for (hs, ht, hf) in holes:
                if hs == s:
                    keep[hf, ht] = False
                    flat = zeroed[s, :, ht].reshape(-1)              # [SEG * R * 12500] of that pol (a copy)
                    flat[hf * 5000:(hf + 1) * 5000] = 0
                    zeroed[s, :, ht] = flat.reshape(SEG, R * 12500)
            f.write(blk[keep].tobytes())
    argv = ["-b", "8", "-w", "2", "-r", "2", "--replay", dump, "--datadir", str(tmp_path), "--logdir",
            str(tmp_path / "logs"), "--no-control", "--rows-per-seg", str(R)]
    assert pbmod.run(pbmod.build_parser().parse_args(argv)) == 0
    nseg = (nsec - 1) * SEG
    res, _, _ = oracle_run(oracle, zeroed.reshape(nsec * SEG, 2, R * 12500)[:nseg], R, rfi_mode=2, npol=1, nbit=8)
    sp = sigproc.sigproc_header(7, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, 8)
    assert (tmp_path / "20160701_010000_muos_ea07.fil").read_bytes() == sp + b"".join(r.codes_raw.tobytes() for r in res)
    assert (tmp_path / "20160701_010000_muos_ea07_kur.fil").read_bytes() == sp + b"".join(r.codes_kur.tobytes() for r in res)


def test_profile_pass_logs_the_stage_lines(tmp_path):
    """-t: the per-stage PROFILE block of the reference (src/process_baseband.cu:1538-1556), with the fused
    kernels under the labels of the stages they replace."""
    nsec = 3
    data = make_input(19, R, nsec * SEG, rfi=False, dropped=False)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data)
    argv = ["-b", "8", "-w", "0", "-t", "--replay", dump, "--datadir", str(tmp_path), "--logdir", str(tmp_path / "logs"),
            "--no-control", "--rows-per-seg", str(R)]
    assert pbmod.run(pbmod.build_parser().parse_args(argv)) == 0
    log = next((tmp_path / "logs").glob("*_process_*.log")).read_text()
    for label in ("Proc Time...", "Read Time...", "Copy To Dev.", "Kurtosis....", "FFT.........", "Normalize...", "Write......."):
        assert label in log
    import re
    assert re.search(r"FFT\.+([0-9]+\.[0-9]{3})\n", log)          # seconds, %.3f like the reference (tiny here)


@pytest.mark.parametrize("seed", range(8))
def test_gpu_vdif_gather_random_loss_patterns(seed):
    """Seeded random damage to one second of frames -- frames dropped, frames repeated, the invalid-data bit set, the
    order shuffled (the first frame stays: it anchors the block) -- through pb_submit_vdif (host frame index + the
    gather kernel, in place of the demux loop of src/process_baseband.cu:1015-1067) against the host deframer on the same
    bytes fed through pb_submit_planar: identical filterbank bytes, both streams."""
    lp = libpb()
    rng = np.random.default_rng(900 + seed)
    data = make_input(40 + seed, R, SEG, rfi=bool(rng.integers(0, 2)), dropped=False)
    p0 = np.concatenate([data[i, 0] for i in range(SEG)])
    p1 = np.concatenate([data[i, 1] for i in range(SEG)])
    blk = vdif.frame_block(p0, p1, 3600, 33, 7).reshape(-1, 5032).copy()
    nfr = blk.shape[0]
    # a block always holds a second's worth of slots (the host seals it that way): a LOST frame is a slot that holds
    # something else -- a repeat of another frame of the second (lands where that one lands: harmless) or a frame of
    # another second (outside the block's span: ignored); its own place stays zero
    damaged = blk.copy()
    lost = 1 + rng.choice(nfr - 1, size=int(rng.integers(0, nfr // 4)), replace=False)
    for i in lost:
        if rng.integers(0, 2):
            damaged[i] = blk[int(rng.integers(0, nfr))]
        else:
            w = damaged[i, :4].view("<u4").copy()
            w[0] = (w[0] & ~np.uint32(0x3FFFFFFF)) | np.uint32(3600 + int(rng.integers(1, 5)))
            damaged[i, :4] = w.view(np.uint8)
    keep = np.ones(nfr, bool)
    keep[lost] = False
    dup = lost
    rest = np.arange(1, nfr)
    rng.shuffle(rest)
    damaged = damaged[np.concatenate([[0], rest])].copy()
    for v in rng.choice(np.arange(1, damaged.shape[0]), size=int(rng.integers(0, 5)), replace=False):
        damaged[v, 3] |= 0x80                                          # invalid-data bit -> zero fill
    ref = vdif.deframe_block(damaged.ravel())
    with lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=SEG) as h1, lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=SEG) as h2:
        h1.submit_vdif(0, 0, damaged.ravel())
        h1.process(SEG)
        a = h1.fetch(0, 0, SEG)
        n = R * 12500
        for s in range(SEG):
            h2.submit_planar(0, s, ref[0, s * n:(s + 1) * n], ref[1, s * n:(s + 1) * n])
        h2.process(SEG)
        b = h2.fetch(0, 0, SEG)
    assert np.array_equal(a["raw"], b["raw"]) and np.array_equal(a["kur"], b["kur"]), (seed, int((~keep).sum()), dup.size)
