"""The host loop of process_baseband (vlite-fast_amd/process_baseband.py) without a GPU: ring in, frames
placed by their headers, seconds pipelined over the buffer sets, files and rings out, the control
socket.  The device is replaced by a stand-in that deframes on the host and returns bytes that are
a pure function of the samples it was given, so every byte of the outputs can be predicted.

Reference behaviour under test (src/process_baseband.cu): frames placed by (thread, frame number)
:1017-1034; a second closed by the first frame of another second :1019,1058 (hence the dropped last
second); a dropped frame costs that frame only; > 1 s skip -> EXIT_FAILURE :1069-1077; CMD_QUIT checked
once per second :1081 (src/utils.c:174-186: any 'Q' among up to 32 bytes of one non-blocking read);
out ring 10 s then 1 s :1482-1494; coadd ring one write per segment :1416-1422."""
import importlib
import os
import socket

import numpy as np
import pytest

import synth

vdif = importlib.import_module("vlite-fast_amd.vdif")
sigproc = importlib.import_module("vlite-fast_amd.sigproc")
dada = importlib.import_module("vlite-fast_amd.dada")
pbmod = importlib.import_module("vlite-fast_amd.process_baseband")

R, SEG = 8, 10
NSEC_SAMP = R * SEG * 12500            # samples per pol per (shortened) second
FPS = NSEC_SAMP // 5000                # frames per thread per second = 200
TRIM = 2 * R * 4096 // 16              # 8-bit, npol 1


class FakeHandle(object):
    """Duck-typed PbHandle: host deframe, output = bytes of the staged samples (pol 0 -> raw stream,
    pol 1 -> excised stream)."""

    def __init__(self, nsets=2):
        self.trim, self.nsets, self.cur = TRIM, nsets, 0
        self.staged = [None] * nsets
        self.done = [None] * nsets
        self.calls = []

    def select_set(self, i):
        assert 0 <= i < self.nsets
        self.cur = i

    def reset_history(self, ant):
        pass

    def submit_vdif(self, ant, seg0, block, second=None, frame0=0):
        assert self.staged[self.cur] is None, "buffer set refilled before its output was fetched"
        self.staged[self.cur] = vdif.deframe_block(np.array(block, copy=True), second, frame0)
        self.calls.append(("submit", self.cur, second))

    def process(self, nseg, inject_now=0):
        self.done[self.cur], self.staged[self.cur] = self.staged[self.cur], None
        self.calls.append(("process", self.cur, inject_now))

    def fetch(self, ant, seg0, nseg, raw=True, kur=True, **kw):
        d = self.done[self.cur]
        assert d is not None
        self.calls.append(("fetch", self.cur))
        return dict(raw=d[0, :nseg * TRIM].copy() if raw else None, kur=d[1, :nseg * TRIM].copy() if kur else None)

    def timers(self, reset=False):
        return {}

    def profile(self, on):
        pass

    def close(self):
        pass


def _frames(nsec, seed=5):
    """[nsec][FPS][2][5032] frames of an observation starting at VDIF second 3600, epoch 33"""
    out = []
    for s in range(nsec):
        p0 = synth.baseband_u8(seed * 100 + 2 * s, NSEC_SAMP)
        p1 = synth.baseband_u8(seed * 100 + 2 * s + 1, NSEC_SAMP)
        p0[p0 == 0] = 1
        p1[p1 == 0] = 1                      # zeros are reserved for "frame missing" in these tests
        out.append(vdif.frame_block(p0, p1, 3600 + s, 33, 7).reshape(FPS, 2, 5032).copy())
    return out


def _header():
    return vdif.ascii_header_format(vdif.writer_header(7, 0.8718, -0.72452, "B0833-45", 58000.0, "19A-331", 33, 3600))


def _args(tmp_path, extra=()):
    return pbmod.build_parser().parse_args(
        ["-b", "8", "-w", "2", "-r", "2", "--replay", "unused", "--datadir", str(tmp_path), "--logdir",
         str(tmp_path / "logs"), "--no-control", "--rows-per-seg", str(R)] + list(extra))


def _ring(stream_bytes):
    r = dada.MemoryRing()
    r.write_header(_header())
    r.write(np.frombuffer(stream_bytes, np.uint8))
    r.end_of_data()
    return r


def _expected(frames_by_sec, nsec_out):
    """what FakeHandle turns the first nsec_out seconds into: (.fil payload, _kur.fil payload)"""
    raw, kur = [], []
    for s in range(nsec_out):
        d = vdif.deframe_block(frames_by_sec[s].reshape(-1), 3600 + s, 0) if frames_by_sec[s].size else np.zeros((2, NSEC_SAMP), np.uint8)
        full = np.zeros((2, NSEC_SAMP), np.uint8)
        full[:, :d.shape[1]] = d[:, :NSEC_SAMP]
        raw.append(full[0, :SEG * TRIM])
        kur.append(full[1, :SEG * TRIM])
    return np.concatenate(raw).tobytes(), np.concatenate(kur).tobytes()


def _log(tmp_path):
    return next((tmp_path / "logs").glob("*_process_*.log")).read_text()


def test_clean_stream_pipelined_over_two_sets(tmp_path):
    fr = _frames(4)
    h = FakeHandle(nsets=2)
    co = dada.FileSink(str(tmp_path / "co.bin"))
    assert pbmod.run(_args(tmp_path), in_ring=_ring(b"".join(f.tobytes() for f in fr)), co_ring=co, handle=h) == 0
    raw, kur = _expected(fr, 3)                                # the last second is dropped (:1058-1064)
    hdr = sigproc.sigproc_header(7, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, 8)
    assert (tmp_path / "20160701_010000_muos_ea07.fil").read_bytes() == hdr + raw
    assert (tmp_path / "20160701_010000_muos_ea07_kur.fil").read_bytes() == hdr + kur
    assert co.nwrites == [TRIM] * 30                           # one write per segment
    # second k+1 is queued before the output of second k is collected
    order = [c[0] for c in h.calls]
    assert order[:5] == ["submit", "process", "submit", "process", "fetch"]
    assert [c[2] for c in h.calls if c[0] == "submit"] == [3600, 3601, 3602]


def test_dropped_frames_cost_only_themselves(tmp_path):
    """A frame missing early in second 1 (thread 1) and the very first frame of second 2 missing:
    every other frame of every second must still land in its place (the block that was one frame short
    swallowed the head of the next second, which has to be carried over, and the origin of a second is
    its number, not its first frame seen)."""
    fr = _frames(5)
    pieces = []
    for s, f in enumerate(fr):
        keep = np.ones((FPS, 2), bool)
        if s == 1:
            keep[3, 1] = False
        if s == 2:
            keep[0, 0] = False
        pieces.append(f[keep])                                  # [nkept][5032] in arrival order
        fr[s] = f[keep]
    h = FakeHandle(nsets=2)
    assert pbmod.run(_args(tmp_path), in_ring=_ring(b"".join(p.tobytes() for p in pieces)), handle=h) == 0
    raw, kur = _expected(fr, 4)
    hdr = sigproc.sigproc_header(7, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, 8)
    got_raw = (tmp_path / "20160701_010000_muos_ea07.fil").read_bytes()[len(hdr):]
    got_kur = (tmp_path / "20160701_010000_muos_ea07_kur.fil").read_bytes()[len(hdr):]
    assert len(got_raw) == len(raw) == 4 * SEG * TRIM
    a, b = np.frombuffer(got_kur, np.uint8), np.frombuffer(kur, np.uint8)
    assert np.array_equal(a, b)
    assert got_raw == raw
    # exactly the two missing frames are zeros (only the part of them inside the bytes FakeHandle returns)
    zeros_kur = (np.frombuffer(got_kur, np.uint8) == 0).sum()
    zeros_raw = (np.frombuffer(got_raw, np.uint8) == 0).sum()
    assert zeros_raw == 5000 and zeros_kur == 5000             # frame 0 of pol 0, frame 3 of pol 1


def test_quit_command_ends_the_run_with_status_0(tmp_path):
    fr = _frames(6)
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.bind(("127.0.0.1", 0))
    rx.setblocking(False)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)

    class QuitAfter(dada.MemoryRing):
        """sends the reader 'Q' once 2.5 seconds of frames have been handed out"""
        sent = False
        nread = 0

        def read(self, nbytes):
            out = dada.MemoryRing.read(self, nbytes)
            self.nread += len(out)
            if not self.sent and self.nread > 2.5 * 2 * FPS * 5032:
                tx.sendto(b"xQ", rx.getsockname())            # any byte of the datagram may be the command
                self.sent = True
            return out

    ring = QuitAfter()
    ring.write_header(_header())
    ring.write(np.frombuffer(b"".join(f.tobytes() for f in fr), np.uint8))
    ring.end_of_data()
    h = FakeHandle(nsets=2)
    assert pbmod.run(_args(tmp_path), in_ring=ring, handle=h, control_sock=rx) == 0
    assert "Received CMD_QUIT, indicating data taking is ceasing.  Exiting." in _log(tmp_path)
    nsub = len([c for c in h.calls if c[0] == "submit"])
    assert 1 <= nsub <= 3                                       # stopped early ...
    got = (tmp_path / "20160701_010000_muos_ea07.fil").read_bytes()
    hdr = sigproc.sigproc_header(7, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, 8)
    assert got == hdr + _expected(fr, nsub)[0]                  # ... and everything queued was written out
    rx.close()
    tx.close()


def test_major_data_skip_is_exit_failure(tmp_path):
    fr = _frames(3)
    late = vdif.frame_block(np.ones(NSEC_SAMP, np.uint8), np.ones(NSEC_SAMP, np.uint8), 3600 + 5, 33, 7)
    stream = b"".join(f.tobytes() for f in fr) + late.tobytes()
    h = FakeHandle(nsets=2)
    assert pbmod.run(_args(tmp_path), in_ring=_ring(stream), handle=h) == 1
    assert "Major data skip!  (3605 vs. 3602; thread = 0) Aborting this observation." in _log(tmp_path)


def test_single_buffer_set_handle_still_works(tmp_path):
    fr = _frames(3)
    h = FakeHandle(nsets=1)
    assert pbmod.run(_args(tmp_path), in_ring=_ring(b"".join(f.tobytes() for f in fr)), handle=h) == 0
    assert [c[0] for c in h.calls] == ["submit", "process", "fetch"] * 2
