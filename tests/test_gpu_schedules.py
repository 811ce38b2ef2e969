"""The stream graph of pb_process (csrc/pb_api.hip: five streams, eight events, branches by mode) under every
scheduling mode the library offers, at production size: batches of five full-size segments (R = 1024, half a second
of one antenna each) through
    buffer sets 1 / 2 / 3  x  detect beside the next channeliser or behind its own (PB_OVERLAP_DETECT)
    x  the channeliser that flags its own rows or kurtosis kernel + channeliser (PB_FUSE_KURTOSIS)
    x  the kurtosis pass beside or behind the previous channeliser (PB_KUR_EARLY; three sets, two kernels)
    x  input resident in the sets' buffers (pb_input_dev / nothing staged between calls, as bench.py runs) or
       re-staged before every call (device-to-device, or from host memory)
    +  detect's ring two / three chunks deep (PB_DETECT_DEPTH), taps = 4
must give IDENTICAL bytes for every batch, both streams, and the same final bandpass state: none of these switches may
change a result, and an ordering hole in the graph -- a wait dropped by the next overlap trick --
shows up here as a difference.  The first mode's first two segments are also compared with the oracle, so "identical"
means "identical to the right answer".  (The switches are read per handle at pb_create, which is what lets one
process hold all modes.)  Replaces tools/soak_equivalence.sh as the guard of src/process_baseband.cu:1108-1376's
order."""
import itertools

import numpy as np
import pytest

from helpers import NCHAN, libpb, oracle_run

pytestmark = pytest.mark.gpu
R, S, NB = 1024, 5, 5
N = R * 12500


def _input():
    """S segments x 2 pols of genbase-like noise on the GPU, with impulsive bursts in ~1 % of the 500-sample blocks and
    one dropped frame (code 0): flags, excised transforms, a NaN-kurtosis block"""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(20251)
    x = torch.randn(S * 2 * N, device="cuda", generator=g) * 16.9 + 128.5
    nblk = S * 2 * N // 500
    bad = torch.rand(nblk, device="cuda", generator=g) < 0.01
    burst = (torch.rand(S * 2 * N, device="cuda", generator=g) - 0.5) * 180.0
    x = (x + burst * bad.repeat_interleave(500)).clamp_(1, 255).to(torch.uint8)
    x[3 * 2 * N + 5000:3 * 2 * N + 10000] = 0
    torch.cuda.synchronize()
    return x


def _modes():
    out = []
    for nsets, fuse, feed in itertools.product((1, 2, 3), (1, 0), ("resident", "staged")):
        overlaps = (1, 0) if nsets >= 2 else (1,)
        earlies = (1, 0) if (nsets >= 3 and not fuse) else (1,)
        for ov, ke in itertools.product(overlaps, earlies):
            out.append(dict(nsets=nsets, fuse=fuse, feed=feed, overlap=ov, kur_early=ke))
    out.append(dict(nsets=3, fuse=1, feed="resident", overlap=1, kur_early=1, depth=2))
    out.append(dict(nsets=3, fuse=1, feed="resident", overlap=1, kur_early=1, depth=3))
    out.append(dict(nsets=2, fuse=0, feed="staged", overlap=1, kur_early=1, depth=3))
    out.append(dict(nsets=3, fuse=1, feed="host", overlap=1, kur_early=1))
    out.append(dict(nsets=2, fuse=0, feed="host", overlap=0, kur_early=1))
    return out


def _run(lp, monkeypatch, x, mode, taps=1, host=None):
    """NB batches (the same S segments every time: the bandpass state is what makes the batches differ), collected
    nsets - 1 calls late like bench.py -> ([raw bytes per batch], [excised bytes per batch], bandpass)"""
    monkeypatch.setenv("PB_OVERLAP_DETECT", str(mode["overlap"]))
    monkeypatch.setenv("PB_KUR_EARLY", str(mode["kur_early"]))
    monkeypatch.setenv("PB_FUSE_KURTOSIS", str(mode["fuse"]))
    monkeypatch.setenv("PB_DETECT_DEPTH", str(mode.get("depth", 0)))
    nsets = mode["nsets"]
    A = mode.get("nant", 1)
    raw, kur = [], []

    def stage(h):
        for a in range(A):                       # (antenna a gets the segments rotated by a: different data)
            for s in range(S):
                src = (s + a) % S
                if mode["feed"] == "host":
                    h.submit_planar(a, s, host[src, 0], host[src, 1])
                else:
                    base = x.data_ptr() + src * 2 * N
                    h.submit_planar_dev(a, s, base, base + N, N)

    def collect(h, b):
        h.select_set(b % nsets)
        raw.append(np.concatenate([h.fetch_view(a, 0, S).copy() for a in range(A)]))
        kur.append(np.concatenate([h.fetch_view(a, 1, S).copy() for a in range(A)]))

    with lp.PbHandle(nant=A, nbit=8, rfi_mode=2, rows_per_seg=R, max_seg=S, nsets=nsets, taps=taps) as h:
        if mode["feed"] == "resident":
            for st in range(nsets):
                h.select_set(st)
                stage(h)
            h.sync()
        for b in range(NB):
            h.select_set(b % nsets)
            if mode["feed"] != "resident":
                stage(h)
            h.process(S)
            if b >= nsets - 1:
                collect(h, b - (nsets - 1))
        for b in range(max(0, NB - (nsets - 1)), NB):
            collect(h, b)
        bps = [h.get_bandpass(a) for a in range(A)]
        bp = (np.concatenate([b[0] for b in bps]), np.concatenate([b[1] for b in bps]))
    return raw, kur, bp


def _name(m):
    return " ".join("%s=%s" % kv for kv in sorted(m.items()))


def test_every_scheduling_mode_gives_the_same_bytes(oracle, monkeypatch):
    lp = libpb()
    x = _input()
    host = x.cpu().numpy().reshape(S, 2, N)
    modes = _modes()
    assert len(modes) >= 29
    base = None
    for m in modes:
        got = _run(lp, monkeypatch, x, m, host=host)
        if base is None:
            base = got
            # identical to the RIGHT answer: the first batch's first two segments against the oracle's serial run
            res, _, _ = oracle_run(oracle, host[:2], R, rfi_mode=2, npol=1, nbit=8)
            trim = got[0][0].size // S
            assert np.array_equal(got[0][0][:2 * trim], np.concatenate([r.codes_raw for r in res])), "raw codes vs oracle"
            assert np.array_equal(got[1][0][:2 * trim], np.concatenate([r.codes_kur for r in res])), "excised codes vs oracle"
            assert (got[1][0] != got[0][0]).any()                       # something was excised
            assert len(set(b.tobytes() for b in got[0])) == NB          # the batches differ (bandpass state)
            continue
        for b in range(NB):
            assert np.array_equal(got[0][b], base[0][b]), "raw stream, batch %d: %s" % (b, _name(m))
            assert np.array_equal(got[1][b], base[1][b]), "excised stream, batch %d: %s" % (b, _name(m))
        for i in range(2):
            assert np.array_equal(got[2][i].view(np.uint32), base[2][i].view(np.uint32)), "bandpass %d: %s" % (i, _name(m))


def test_every_scheduling_mode_gives_the_same_bytes_taps4(monkeypatch):
    """the same for the 4-tap window (kurtosis pass + weights + history kernels on their own stream: the graph with
    the most cross-stream edges); its single-set result is what tests/test_gpu_pfb.py pins to the oracle"""
    lp = libpb()
    x = _input()
    host = x.cpu().numpy().reshape(S, 2, N)
    modes = [dict(nsets=1, fuse=1, feed="staged", overlap=1, kur_early=1)]
    for nsets, feed, ov, ke in itertools.product((2, 3), ("resident", "staged", "host"), (1, 0), (1, 0)):
        if ke == 0 and nsets < 3:
            continue
        modes.append(dict(nsets=nsets, fuse=1, feed=feed, overlap=ov, kur_early=ke))
    base = None
    for m in modes:
        got = _run(lp, monkeypatch, x, m, taps=4, host=host)
        if base is None:
            base = got
            assert (got[1][0] != got[0][0]).any() and len(set(b.tobytes() for b in got[0])) == NB
            continue
        for b in range(NB):
            assert np.array_equal(got[0][b], base[0][b]), "raw stream, batch %d: %s" % (b, _name(m))
            assert np.array_equal(got[1][b], base[1][b]), "excised stream, batch %d: %s" % (b, _name(m))
        for i in range(2):
            assert np.array_equal(got[2][i].view(np.uint32), base[2][i].view(np.uint32)), "bandpass %d: %s" % (i, _name(m))


def test_two_antennas_per_handle_scheduling_modes_agree(monkeypatch):
    """BASELINE configs[3]'s per-GPU shape (two antennas in one handle, detect's grid z = 2, ring two chunks deep):
    1 / 2 / 3 buffer sets, detect beside or behind, resident or re-staged input."""
    lp = libpb()
    x = _input()
    host = x.cpu().numpy().reshape(S, 2, N)
    modes = [dict(nsets=1, fuse=1, feed="staged", overlap=1, kur_early=1, nant=2)]
    for nsets, ov, feed in itertools.product((2, 3), (1, 0), ("resident", "staged")):
        modes.append(dict(nsets=nsets, fuse=1, feed=feed, overlap=ov, kur_early=1, nant=2))
    base = None
    for m in modes:
        got = _run(lp, monkeypatch, x, m, host=host)
        if base is None:
            base = got
            half = got[0][0].size // 2
            assert (got[0][0][:half] != got[0][0][half:]).any()          # the two antennas carry different data
            continue
        for b in range(NB):
            assert np.array_equal(got[0][b], base[0][b]), "raw stream, batch %d: %s" % (b, _name(m))
            assert np.array_equal(got[1][b], base[1][b]), "excised stream, batch %d: %s" % (b, _name(m))
        for i in range(2):
            assert np.array_equal(got[2][i].view(np.uint32), base[2][i].view(np.uint32)), "bandpass %d: %s" % (i, _name(m))
