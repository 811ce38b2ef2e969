"""taps = 4 (4-tap Hamming WOLA window of analysis/baseband.py:polyphase_filterbank) in the
streaming 8-bit path: the whole chain is bit-exact against the oracle's kernels composed around
the same fp32 FIR, rows carry over between pb_process calls, and the window-energy weights behave."""
import ctypes as C

import numpy as np
import pytest

from helpers import NCHAN, libpb, make_input

pytestmark = pytest.mark.gpu
R = 16
NFFT = 12500


def _taps32(oracle):
    """float32 FIR taps exactly as libpb_hip builds them (sequential double sums for the norms)."""
    ns, nw = NFFT, 4
    win = oracle.hamming_sym(nw * ns)
    norms = [1. / np.cumsum(win[j * ns:(j + 1) * ns] ** 2)[-1] for j in range(nw)]
    return np.stack([(win[j * ns:(j + 1) * ns] * norms[0] * (norms[j] if j else 1.0)).astype(np.float32)
                     for j in range(nw)])


def _oracle_pfb_chain(oracle, data, nbit=8):
    """RFI mode 0, npol 1: convertarray -> fp32 FIR (products, then left-to-right adds) -> oracle FFT
    -> detect_and_normalize2 -> pscrunch -> tscrunch -> sel_and_dig_8b, segment by segment."""
    L = oracle.lib()
    fp = C.POINTER(C.c_float)
    nseg = data.shape[0]
    t = _taps32(oracle)
    volts = [oracle.convertarray(np.concatenate([data[s, p] for s in range(nseg)])).reshape(-1, NFFT)
             for p in range(2)]
    rows = nseg * R
    spec = []
    for p in range(2):
        pad = np.concatenate([np.zeros((3, NFFT), np.float32), volts[p]])
        acc = t[0] * pad[0:rows]
        for j in (1, 2, 3):
            acc = (acc + t[j] * pad[j:j + rows]).astype(np.float32)
        spec.append(oracle.rfft(acc.ravel()))
    bp = np.zeros(2 * NCHAN, np.float32)
    scale = np.float32((12500.0 / 128000000 * 8) / 1.0)
    codes = []
    for s in range(nseg):
        fo = np.ascontiguousarray(np.stack([spec[0][s * R:(s + 1) * R], spec[1][s * R:(s + 1) * R]]))  # [pol][R][6251] c64
        fof = fo.view(np.float32)
        L.orc_detect_and_normalize2(fof.ctypes.data_as(fp), bp.ctypes.data_as(fp), C.c_float(scale), R)
        L.orc_pscrunch(fof.ctypes.data_as(fp), R * NCHAN)
        ave = np.zeros(R // 8 * NCHAN, np.float32)
        L.orc_tscrunch(fof.ctypes.data_as(fp), ave.ctypes.data_as(fp), ave.size)
        out = np.zeros(R // 8 * 4096, np.uint8)
        L.orc_sel_and_dig_8b(ave.ctypes.data_as(fp), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size, 1, R // 8)
        codes.append(out)
    return np.concatenate(codes)


def test_pfb_streaming_chain_bit_exact_across_calls(oracle):
    lp = libpb()
    nseg = 4
    data = make_input(51, R, nseg, rfi=False, dropped=False)
    ref = _oracle_pfb_chain(oracle, data)
    got = []
    with lp.PbHandle(nbit=8, rfi_mode=0, taps=4, rows_per_seg=R, max_seg=2) as h:
        for call in range(2):                                   # two calls of two segments: history carries over
            for s in range(2):
                h.submit_planar(0, s, data[call * 2 + s, 0], data[call * 2 + s, 1])
            h.process(2)
            got.append(h.fetch(0, 0, 2, kur=False)["raw"])
        # a new observation forgets the carried rows
        h.reset_history(0)
        h.reset_bandpass(0)
        for s in range(2):
            h.submit_planar(0, s, data[s, 0], data[s, 1])
        h.process(2)
        again = h.fetch(0, 0, 2, kur=False)["raw"]
    got = np.concatenate(got)
    assert np.array_equal(got, ref)
    assert np.array_equal(again, ref[:again.size])


def test_pfb_weights_and_excision(oracle):
    lp = libpb()
    nseg = 2
    data = make_input(52, R, nseg, rfi=False, dropped=True)   # chance flags + one zero-filled frame
    with lp.PbHandle(nbit=8, rfi_mode=2, taps=4, rows_per_seg=R, max_seg=nseg, debug_keep=True) as h:
        for s in range(nseg):
            h.submit_planar(0, s, data[s, 0], data[s, 1])
        h.process(nseg)
        o2 = h.fetch(0, 0, nseg)
        w = np.concatenate([h.debug_fetch(lp.DBG_ROWWEIGHT, 0, s) for s in range(nseg)])
        flags = np.concatenate([h.debug_fetch(lp.DBG_FLAGS, 0, s) for s in range(nseg)]).reshape(-1, 25)
    with lp.PbHandle(nbit=8, rfi_mode=0, taps=4, rows_per_seg=R, max_seg=nseg) as h:
        for s in range(nseg):
            h.submit_planar(0, s, data[s, 0], data[s, 1])
        h.process(nseg)
        o0 = h.fetch(0, 0, nseg, kur=False)
    assert np.array_equal(o2["raw"], o0["raw"])                 # the raw stream ignores flags
    # window-energy weights: ramp up over the first three rows, 1 where four clean rows contribute
    t = _taps32(oracle).astype(np.float64)
    E = (t.reshape(4, 25, 500) ** 2).sum(axis=2)
    expect = np.zeros(nseg * R)
    for g in range(nseg * R):
        s = 0.0
        for j in range(4):
            rr = g - 3 + j
            if rr >= 0:
                s += E[j][flags[rr] == 0].sum()
        expect[g] = s / E.sum()
    np.testing.assert_allclose(w, expect, rtol=2e-6, atol=1e-7)
    clean = np.array([g >= 3 and not flags[g - 3:g + 1].any() for g in range(nseg * R)])
    assert clean.any() and np.all(w[clean] == 1.0) and w[0] < w[1] < w[2] < 1.0
    assert (o2["kur"] != o2["raw"]).any()
