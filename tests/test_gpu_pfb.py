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


# ---------------------------------------------------------------------------------------------------------
# The taps = 4 configuration AS BENCHMARKED: two buffer sets (batches pipelined over the library's streams, the
# the history kernel of batch k recorded in ev_hist, which the next batch's weights wait for), RFI mode 2 (both streams; excision zeroes the flagged blocks of every
# contributing row and the row weight is the unflagged fraction of the window's energy), R = 1024 (the XCD-aware
# row mapping of k_channelize_pfb).  The oracle is the reference's kernels composed around the same fp32 FIR.

def _tap_energy_sequential(t):
    """E[j][b] exactly as pb_create evaluates it: a sequential double sum of t*t over the block, rounded to float"""
    E = np.empty((4, 25), np.float32)
    for j in range(4):
        for b in range(25):
            e = 0.0
            for v in t[j, b * 500:(b + 1) * 500].astype(np.float64):
                e += v * v
            E[j, b] = np.float32(e)
    return E


def _oracle_pfb_chain_mode2(oracle, data, Rr):
    """data: u8 [nseg_total][2][Rr*12500], the whole stream (all batches).  Returns (raw codes, excised codes,
    weights, flags) of RFI mode 2, npol 1, 8 bits: convertarray -> kurtosis -> compute_dagostino -> flags;
    raw: FIR -> FFT -> detect_and_normalize2 -> pscrunch -> tscrunch -> sel_and_dig_8b;
    excised: apply_kurtosis zeroing of every row -> FIR -> FFT -> detect_and_normalize3 (window-energy weights)
    -> pscrunch_weights -> tscrunch_weights -> sel_and_dig_8b."""
    L = oracle.lib()
    fp = C.POINTER(C.c_float)
    nseg = data.shape[0]
    rows = nseg * Rr
    t = _taps32(oracle)
    volts = [oracle.convertarray(np.concatenate([data[s, p] for s in range(nseg)])) for p in range(2)]
    kur = np.concatenate([oracle.kurtosis(v)[1] for v in volts])
    dag = oracle.compute_dagostino(kur)
    nb = rows * 25
    flags = (dag[:nb] > 3.0).reshape(rows, 25)
    # window-energy weights with the library's own E (sequential double sums)
    E = _tap_energy_sequential(t)
    tot = np.float32(0)
    for v in E.ravel():
        tot = np.float32(tot + v)
    w = np.empty(rows, np.float32)
    for g in range(rows):
        rr = [flags[g - 3 + j] if g - 3 + j >= 0 else np.ones(25, bool) for j in range(4)]
        if not np.any(rr):
            w[g] = np.float32(1.0)
            continue
        s = np.float32(0)
        for j in range(4):
            for b in range(25):
                if not rr[j][b]:
                    s = np.float32(s + E[j, b])
        w[g] = np.float32(s / tot)

    def spectra(v):
        v = v.reshape(-1, NFFT)
        pad = np.concatenate([np.zeros((3, NFFT), np.float32), v])
        acc = t[0] * pad[0:rows]
        for j in (1, 2, 3):
            acc = (acc + t[j] * pad[j:j + rows]).astype(np.float32)
        return oracle.rfft(acc.ravel())

    spec_raw = [spectra(volts[p]) for p in range(2)]
    spec_kur = []
    for p in range(2):
        z = volts[p].reshape(rows, 25, 500).copy()
        z[flags] = 0.0
        spec_kur.append(spectra(z.ravel()))
    scale = np.float32((12500.0 / 128000000 * 8) / 1.0)
    bp_raw = np.zeros(2 * NCHAN, np.float32)
    bp_kur = np.zeros(2 * NCHAN, np.float32)
    codes_raw, codes_kur = [], []
    u8p = C.POINTER(C.c_uint8)
    for s in range(nseg):
        sl = slice(s * Rr, (s + 1) * Rr)
        fo = np.ascontiguousarray(np.stack([spec_raw[0][sl], spec_raw[1][sl]])).view(np.float32)
        L.orc_detect_and_normalize2(fo.ctypes.data_as(fp), bp_raw.ctypes.data_as(fp), C.c_float(scale), Rr)
        L.orc_pscrunch(fo.ctypes.data_as(fp), Rr * NCHAN)
        ave = np.zeros(Rr // 8 * NCHAN, np.float32)
        L.orc_tscrunch(fo.ctypes.data_as(fp), ave.ctypes.data_as(fp), ave.size)
        out = np.zeros(Rr // 8 * 4096, np.uint8)
        L.orc_sel_and_dig_8b(ave.ctypes.data_as(fp), out.ctypes.data_as(u8p), out.size, 1, Rr // 8)
        codes_raw.append(out)
        fk = np.ascontiguousarray(np.stack([spec_kur[0][sl], spec_kur[1][sl]])).view(np.float32)
        kw = np.ascontiguousarray(np.concatenate([w[sl], w[sl]]))
        L.orc_detect_and_normalize3(fk.ctypes.data_as(fp), kw.ctypes.data_as(fp), bp_kur.ctypes.data_as(fp),
                                    C.c_float(scale), Rr)
        L.orc_pscrunch_weights(fk.ctypes.data_as(fp), kw.ctypes.data_as(fp), Rr * NCHAN)
        avek = np.zeros(Rr // 8 * NCHAN, np.float32)
        L.orc_tscrunch_weights(fk.ctypes.data_as(fp), avek.ctypes.data_as(fp), kw.ctypes.data_as(fp), avek.size)
        outk = np.zeros(Rr // 8 * 4096, np.uint8)
        L.orc_sel_and_dig_8b(avek.ctypes.data_as(fp), outk.ctypes.data_as(u8p), outk.size, 1, Rr // 8)
        codes_kur.append(outk)
    return np.concatenate(codes_raw), np.concatenate(codes_kur), w, flags


def _run_pipelined(lp, batches, Rr, rfi_mode, nsets, resident=False):
    """bench.py's order of calls: batch k goes to buffer set k mod nsets (re-staged every time), its bytes are
    collected nsets - 1 batches later.  resident (needs len(batches) <= nsets): every batch is staged into its set
    and the device is idle before the first pb_process, then the batches are processed back to back with no staging
    between them -- bench.py's HBM-resident input: nothing but the library's own events orders the kernels."""
    S = batches[0].shape[0]
    raw, kur, wts = [], [], []

    def collect(h, j):
        h.select_set(j % nsets)
        o = h.fetch(0, 0, S, raw=rfi_mode != 1, kur=rfi_mode != 0, weights=rfi_mode != 0)
        raw.append(o["raw"])
        kur.append(o["kur"])
        wts.append(o["weights"])

    with lp.PbHandle(nbit=8, rfi_mode=rfi_mode, taps=4, rows_per_seg=Rr, max_seg=S, nsets=nsets) as h:
        if resident:
            assert len(batches) <= nsets
            for k, d in enumerate(batches):
                h.select_set(k)
                for s in range(S):
                    h.submit_planar(0, s, d[s, 0], d[s, 1])
            h.sync()
        for k, d in enumerate(batches):
            h.select_set(k % nsets)
            for s in range(S):
                if not resident:
                    h.submit_planar(0, s, d[s, 0], d[s, 1])
            h.process(S)
            if k >= nsets - 1:
                collect(h, k - (nsets - 1))
        for j in range(max(0, len(batches) - (nsets - 1)), len(batches)):
            collect(h, j)
    return raw, kur, wts


def test_pfb_pipelined_two_sets_mode0_bit_exact(oracle):
    """(a) nsets = 2, four batches re-staged into the reused sets: the carried rows of batch k (its history kernel,
    recorded in ev_hist) reach batch k + 1 although its input was staged while batch k was still running."""
    lp = libpb()
    S, NB = 2, 4
    data = make_input(61, R, S * NB, rfi=False, dropped=False)
    ref = _oracle_pfb_chain(oracle, data)
    batches = [data[k * S:(k + 1) * S] for k in range(NB)]
    raw, _, _ = _run_pipelined(lp, batches, R, 0, 2)
    assert np.array_equal(np.concatenate(raw), ref)


@pytest.mark.parametrize("nsets", [1, 2, 3])
def test_pfb_rfi_mode2_both_streams_bit_exact(oracle, nsets):
    """(b) RFI mode 2 as benchmarked: RFI bursts, a row with every block flagged (weight 0 for four output rows'
    window share), a strongly flagged stretch, a dropped frame, three batches (flags of the carried rows are used
    by the next batch).  Raw AND excised codes bit-exact, weights bit-exact.  nsets = 3: the kurtosis pass of batch k + 1
    runs beside the channeliser of batch k (it no longer waits for it), the PFB weights behind the history kernel."""
    lp = libpb()
    S, NB = 2, 3
    data = make_input(62, R, S * NB)
    # four consecutive rows with every block flagged: one output row whose whole window is excised (weight 0, the
    # +inf convention of the power plane) and neighbours below MIN_WEIGHT
    for p in range(2):
        seg = data[3, p, 4 * NFFT:8 * NFFT]
        seg[:] = np.where((np.arange(seg.size) // 5) % 2 == 0, 220, 36)
    ref_raw, ref_kur, w, flags = _oracle_pfb_chain_mode2(oracle, data, R)
    assert flags.sum() > 20 and flags.all(axis=1).sum() >= 5
    batches = [data[k * S:(k + 1) * S] for k in range(NB)]
    raw, kur, wts = _run_pipelined(lp, batches, R, 2, nsets)
    gw = np.concatenate(wts)
    # pb_fetch reports what tscrunch_weights sees (rows below MIN_WEIGHT as 0)
    expect_w = np.where(w >= np.float32(0.2), w, np.float32(0))
    assert np.array_equal(gw.view(np.uint32), expect_w.view(np.uint32)), "window-energy weights differ"
    assert (w == 0).any() and ((w > 0) & (w < np.float32(0.2))).any() and ((w >= np.float32(0.2)) & (w < 1)).any()
    assert np.array_equal(np.concatenate(raw), ref_raw), "raw-stream codes differ"
    assert np.array_equal(np.concatenate(kur), ref_kur), "excised-stream codes differ"
    assert (ref_kur != ref_raw).any()


def test_pfb_three_sets_resident_input_history_ordering(oracle):
    """Three buffer sets with the input already on the device (no staging between the pb_process calls, as in
    bench.py): batch 0's history kernel runs on the main stream, batch 1's kurtosis pass and PFB weights on the
    kurtosis stream without waiting for batch 0's channeliser -- only ev_hist orders the weights of batch 1's first
    three rows behind the history kernel that keeps batch 0's last flags (with staging in between, the copies hid
    the missing order).  Flags in batch 0's last rows make those weights differ from "nothing flagged" and from the
    memset state "everything flagged"."""
    lp = libpb()
    S, NB = 2, 3
    data = make_input(64, R, S * NB)
    for b in (0, 1):                      # RFI in the last three rows of batches 0 and 1, one block each
        seg = b * S + S - 1
        # (the reference's window carries norms[0] * norms[j] on tap j >= 1, so the window's energy -- and with it the
        # weight of output row g -- is almost all in tap 0, i.e. in row g - 3: the first three output rows of a batch
        # are weighted by the flags of the previous batch's last three rows)
        for row, blk in ((R - 1, 7), (R - 2, 19), (R - 3, 3)):
            x = data[seg, 0, row * NFFT + blk * 500:row * NFFT + (blk + 1) * 500]
            x[:] = np.where((np.arange(500) // 5) % 2 == 0, 230, 26)
    ref_raw, ref_kur, w, flags = _oracle_pfb_chain_mode2(oracle, data, R)
    assert flags[S * R - 1].any() and flags[S * R - 2].any() and flags[S * R - 3].any()
    first3 = w[S * R:S * R + 3]
    assert ((first3 > 0) & (first3 < np.float32(0.999))).all(), first3
    batches = [data[k * S:(k + 1) * S] for k in range(NB)]
    raw, kur, wts = _run_pipelined(lp, batches, R, 2, 3, resident=True)
    gw = np.concatenate(wts)
    expect_w = np.where(w >= np.float32(0.2), w, np.float32(0))
    assert np.array_equal(gw.view(np.uint32), expect_w.view(np.uint32)), "window-energy weights differ"
    assert np.array_equal(np.concatenate(raw), ref_raw), "raw-stream codes differ"
    assert np.array_equal(np.concatenate(kur), ref_kur), "excised-stream codes differ"


def test_pfb_fullsize_two_segments_mode2_bit_exact(oracle):
    """(c) R = 1024 (the production segment: XCD-aware row mapping, 32-row detect chunks), two segments in one
    call, RFI mode 2, both streams bit-exact against the composed oracle."""
    lp = libpb()
    Rf = 1024
    data = make_input(63, Rf, 2)
    ref_raw, ref_kur, w, flags = _oracle_pfb_chain_mode2(oracle, data, Rf)
    raw, kur, wts = _run_pipelined(lp, [data], Rf, 2, 2)
    assert np.array_equal(raw[0], ref_raw), "raw-stream codes differ"
    assert np.array_equal(kur[0], ref_kur), "excised-stream codes differ"


def _run_many(lp, data, Rr, S, rfi_mode, nant, nsets=2):
    """batches of S segments through one handle of `nant` antennas (antenna a gets the data rolled by a segments):
    -> (raw, kur, weights) per antenna, concatenated over the batches"""
    nb = data.shape[0] // S
    out = [([], [], []) for _ in range(nant)]

    def collect(h, j):
        h.select_set(j % nsets)
        for a in range(nant):
            o = h.fetch(a, 0, S, raw=rfi_mode != 1, kur=True, weights=True)
            if rfi_mode != 1:
                out[a][0].append(o["raw"])
            out[a][1].append(o["kur"])
            out[a][2].append(o["weights"])

    with lp.PbHandle(nant=nant, nbit=8, rfi_mode=rfi_mode, taps=4, rows_per_seg=Rr, max_seg=S, nsets=nsets) as h:
        for k in range(nb):
            h.select_set(k % nsets)
            for a in range(nant):
                d = np.roll(data, a, axis=0)[k * S:(k + 1) * S]
                for s in range(S):
                    h.submit_planar(a, s, d[s, 0], d[s, 1])
            h.process(S)
            if k >= nsets - 1:
                collect(h, k - (nsets - 1))
        for j in range(max(0, nb - (nsets - 1)), nb):
            collect(h, j)
    return [tuple(np.concatenate(x) if x else None for x in o) for o in out]


@pytest.mark.parametrize("Rr,S,nb,rfi_mode,nant", [(8, 1, 9, 2, 1), (64, 2, 3, 1, 1), (24, 2, 3, 2, 2)])
def test_pfb_buffer_set_counts_agree(Rr, S, nb, rfi_mode, nant):
    """The same batches through one, two and three buffer sets (no pipelining; detect beside the next channeliser;
    the kurtosis pass beside the previous channeliser as well) give the same bytes and weights: RFI mode 1 (no raw
    transform), two antennas in one handle, a row count that is no multiple of 32 (8-row detect chunks), rows with
    code 0 (dropped frames).  The one-set results are the ones the tests above pin to the oracle."""
    lp = libpb()
    data = make_input(66, Rr, S * nb)
    got = {n: _run_many(lp, data, Rr, S, rfi_mode, nant, nsets=n) for n in (1, 2, 3)}
    for n in (2, 3):
        for a in range(nant):
            for i, what in enumerate(("raw", "kur", "weights")):
                x, y = got[1][a][i], got[n][a][i]
                if x is None:
                    assert y is None
                    continue
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (n, a, what)
    assert (got[1][0][1] != 0).any()
    if rfi_mode == 2:
        assert (got[1][0][1] != got[1][0][0]).any()       # something was excised
