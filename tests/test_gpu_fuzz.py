"""Seeded random configurations of the whole chain against the oracle, bit for bit.  The parametrised parity tests fix one
shape per mode; here rows per segment (8 ... 72, i.e. detect's 8-row and 32-row chunk forms, odd multiples of 8),
segments per call, batches, buffer sets, the lag with which batches are collected, RFI mode, output pols and bits, one or
two antennas per handle, which kernel computes the flags (PB_FUSE_KURTOSIS through debug_keep), and the input's content
(RFI bursts, dropped frames, a dead pol, saturated stretches, rows that are flagged whole) are all drawn per case from
a seed, so that combinations nobody wrote down are exercised too -- every 8/4/2-bit code of both streams and the final
bandpass state must equal the oracle's serial run over the same bytes (src/process_baseband.cu:1108-1376's order of
operations, src/pb_kernels.cu:23-735)."""
import numpy as np
import pytest

from helpers import NCHAN, libpb, make_input, oracle_run

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    R = int(rng.choice([8, 16, 24, 32, 40, 64, 72]))
    nseg = int(rng.integers(1, 4))
    nb = int(rng.integers(1, 4))
    nsets = int(rng.integers(1, 4))
    lag = int(rng.integers(0, nsets))                      # batches in flight before the host collects (< nsets)
    rfi_mode = int(rng.choice([0, 1, 2, 2, 2]))
    npol = int(rng.choice([1, 1, 2]))
    nbit = int(rng.choice([8, 8, 4, 2]))
    nant = int(rng.choice([1, 1, 2]))
    keep = bool(rng.integers(0, 2))
    return dict(R=R, nseg=nseg, nb=nb, nsets=nsets, lag=lag, rfi_mode=rfi_mode, npol=npol, nbit=nbit, nant=nant, keep=keep,
                rng=rng)


def _mutate(data, rng):
    """content the fixed fixtures do not have in these combinations"""
    nsegs, _, n = data.shape
    kind = int(rng.integers(0, 6))
    if kind == 0:                                          # one pol dead (code 128 = 0.0) for a whole segment
        data[int(rng.integers(0, nsegs)), int(rng.integers(0, 2))] = 128
    elif kind == 1:                                        # a saturated stretch crossing block boundaries
        s, p = int(rng.integers(0, nsegs)), int(rng.integers(0, 2))
        a = int(rng.integers(0, n - 3000))
        data[s, p, a:a + 2777] = 255
    elif kind == 2:                                        # dropped frames (code 0) scattered over both pols
        for _ in range(int(rng.integers(1, 5))):
            s, p = int(rng.integers(0, nsegs)), int(rng.integers(0, 2))
            a = 5000 * int(rng.integers(0, n // 5000))
            data[s, p, a:a + 5000] = 0
    elif kind == 3:                                        # a whole FFT row of square wave in both pols: weight 0
        s, row = int(rng.integers(0, nsegs)), int(rng.integers(0, n // 12500))
        for p in range(2):
            seg = data[s, p, row * 12500:(row + 1) * 12500]
            seg[:] = np.where((np.arange(seg.size) // 5) % 2 == 0, 220, 36)
    elif kind == 4:                                        # the first segment entirely dropped (bandpass initialised later)
        data[0] = 0
    return kind


@pytest.mark.parametrize("seed", range(64))
def test_random_configuration_is_bit_exact(oracle, seed):
    c = _case(seed)
    lp = libpb()
    R, nseg, nb, nsets, lag = c["R"], c["nseg"], c["nb"], c["nsets"], c["lag"]
    rfi_mode, npol, nbit, nant = c["rfi_mode"], c["npol"], c["nbit"], c["nant"]
    datas = []
    for a in range(nant):
        d = make_input(500 + 10 * seed + a, R, nseg * nb, rfi=bool(c["rng"].integers(0, 2)), dropped=bool(c["rng"].integers(0, 2)))
        _mutate(d, c["rng"])
        datas.append(d)
    names = {0: ("raw",), 1: ("kur",), 2: ("raw", "kur")}[rfi_mode]
    got = {(a, nm): [] for a in range(nant) for nm in names}

    def collect(h, b):
        h.select_set(b % nsets)
        for a in range(nant):
            out = h.fetch(a, 0, nseg, raw=rfi_mode != 1, kur=rfi_mode != 0)
            for nm in names:
                got[(a, nm)].append(np.array(out[nm], copy=True))

    with lp.PbHandle(nant=nant, nbit=nbit, npol=npol, rfi_mode=rfi_mode, rows_per_seg=R, max_seg=nseg, nsets=nsets,
                     debug_keep=c["keep"]) as h:
        for b in range(nb):
            h.select_set(b % nsets)
            for a in range(nant):
                for s in range(nseg):
                    h.submit_planar(a, s, datas[a][b * nseg + s, 0], datas[a][b * nseg + s, 1])
            h.process(nseg)
            if b >= lag:
                collect(h, b - lag)
        for b in range(max(0, nb - lag), nb):
            collect(h, b)
        bps = [h.get_bandpass(a) for a in range(nant)]
    for a in range(nant):
        res, bp_raw, bp_kur = oracle_run(oracle, datas[a], R, rfi_mode=rfi_mode, npol=npol, nbit=nbit)
        for nm in names:
            ref = np.concatenate([getattr(r, "codes_" + nm) for r in res])
            g = np.concatenate(got[(a, nm)])
            assert g.size == ref.size and np.array_equal(g, ref), \
                "case %s antenna %d %s: %d of %d bytes differ" % ({k: v for k, v in c.items() if k != "rng"}, a, nm,
                                                                 int((g != ref).sum()), ref.size)
        gr, gk = bps[a]
        same = lambda x, y: np.array_equal(np.ascontiguousarray(x, np.float32).view(np.uint32),
                                           np.ascontiguousarray(y, np.float32).view(np.uint32))
        if rfi_mode != 1:
            assert same(gr, bp_raw.reshape(2, NCHAN)[:, 2155:]), "raw bandpass state, case %d" % seed
        if rfi_mode == 2:
            assert same(gk, bp_kur.reshape(2, NCHAN)[:, 2155:]), "excised bandpass state, case %d" % seed
        if rfi_mode == 1:
            assert same(gk, bp_raw.reshape(2, NCHAN)[:, 2155:]), "bandpass state (mode 1), case %d" % seed
