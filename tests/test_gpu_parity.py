"""Parity of the HIP path (through the C ABI) against the CPU oracle on seeded inputs.

Bar: bit-exact for flags, weights and every 8/4/2-bit code with the in-library LDS FFT
(same operation order as the oracle's FFT).  With the hipFFT back end the FFT differs from
the oracle's by ~1e-7 relative, so codes may differ by one step where the pre-quantiser value
sits on a step edge; everything upstream of the FFT is still exact."""
import numpy as np
import pytest

from helpers import NCHAN, compact_ave, libpb, make_input, oracle_run

pytestmark = pytest.mark.gpu

R = 16          # rows per segment in these tests (200 000 samples per pol)
NSEG = 3


def _run_gpu(lp, data, backend, rfi_mode=2, npol=1, nbit=8, debug_keep=True, **kw):
    """debug_keep=True keeps the per-block statistics, which also selects the two-kernel path (kurtosis pass, then
    channeliser); False is the production path of the in-library FFT: the channeliser flags its own rows
    (k_channelize_kur)."""
    nseg = data.shape[0]
    h = lp.PbHandle(nant=1, nbit=nbit, npol=npol, rfi_mode=rfi_mode, fft_backend=backend,
                    rows_per_seg=R, max_seg=nseg, keep_ave=True, debug_keep=debug_keep, **kw)
    for s in range(nseg):
        h.submit_planar(0, s, data[s, 0], data[s, 1])
    h.process(nseg)
    out = h.fetch(0, 0, nseg, weights=True, ave=True)
    out["flags"] = [h.debug_fetch(lp.DBG_FLAGS, 0, s) for s in range(nseg)]
    keep = rfi_mode and debug_keep
    out["st_pow"] = [h.debug_fetch(lp.DBG_POW, 0, s) for s in range(nseg)] if keep else None
    out["st_kur"] = [h.debug_fetch(lp.DBG_KUR, 0, s) for s in range(nseg)] if keep else None
    out["st_dag"] = [h.debug_fetch(lp.DBG_DAG, 0, s) for s in range(nseg)] if keep else None
    out["bp"] = h.get_bandpass(0)
    out["trim"], out["ave_per_seg"] = h.trim, h.ave_per_seg
    h.close()
    return out


def _same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    return np.array_equal(a, b)


@pytest.fixture(scope="module")
def data():
    return make_input(3, R, NSEG)


@pytest.fixture(scope="module")
def oracle_mode2(oracle, data):
    return oracle_run(oracle, data, R, rfi_mode=2, npol=1, nbit=8)


def test_prefft_statistics_and_flags_exact(oracle, data, oracle_mode2):
    lp = libpb()
    g = _run_gpu(lp, data, lp.FFT_HIPFFT)
    res, _, _ = oracle_mode2
    nflag = 0
    for s in range(NSEG):
        nb = g["flags"][s].size
        assert _same_bits(g["st_pow"][s].ravel(), res[s].pow), "kurtosis pow differs"
        assert _same_bits(g["st_kur"][s].ravel(), res[s].kur), "kurtosis kur differs (NaN path included)"
        assert _same_bits(g["st_dag"][s].ravel(), res[s].dag), "D'Agostino score differs"
        flags = (res[s].dag[:nb] > 3.0).astype(np.uint8)
        assert np.array_equal(g["flags"][s], flags)
        nflag += int(flags.sum())
        assert _same_bits(g["weights"][s * R:(s + 1) * R], res[s].weights[:R])
    assert nflag > 20            # the RFI and dropped-frame paths were really exercised
    w = g["weights"]
    assert (w == 0).any() and ((w > 0) & (w < 1)).any()


@pytest.mark.parametrize("backend", ["hipfft", "lds"])
def test_flags_exact_without_the_located_crossings(oracle, data, oracle_mode2, monkeypatch, backend):
    """PB_DAG_BANDS=0 forces the fallback pb_create takes when it cannot bracket the score's crossings (host and device
    disagreeing): every flag is then decided by the score itself.  Flags, weights and codes are still the oracle's --
    in the kurtosis kernel (statistics kept) and in the channeliser that flags its own rows."""
    lp = libpb()
    monkeypatch.setenv("PB_DAG_BANDS", "0")
    res, _, _ = oracle_mode2
    if backend == "hipfft":
        g = _run_gpu(lp, data, lp.FFT_HIPFFT)
        for s in range(NSEG):
            nb = g["flags"][s].size
            assert np.array_equal(g["flags"][s], (res[s].dag[:nb] > 3.0).astype(np.uint8))
            assert _same_bits(g["weights"][s * R:(s + 1) * R], res[s].weights[:R])
    with lp.PbHandle(nant=1, nbit=8, rfi_mode=2, rows_per_seg=R, max_seg=NSEG) as h:
        _, _, bands = h.dag_check(2.0, 2.001)
        assert bands[0] == 0 and np.isinf(bands[1]) and bands[2] == 0 and np.isinf(bands[3])     # no crossings in use
        for s in range(NSEG):
            h.submit_planar(0, s, data[s, 0], data[s, 1])
        h.process(NSEG)
        out = h.fetch(0, 0, NSEG, weights=True)
    assert np.array_equal(out["kur"], np.concatenate([r.codes_kur for r in res]))
    assert np.array_equal(out["raw"], np.concatenate([r.codes_raw for r in res]))
    monkeypatch.delenv("PB_DAG_BANDS")
    with lp.PbHandle(nant=1, rows_per_seg=8, max_seg=1) as h:
        _, _, bands = h.dag_check(2.0, 2.001)
        assert 0 < bands[0] < bands[1] < bands[2] < bands[3] < np.inf


def test_block_kurtosis_row_statistic(oracle, data, oracle_mode2):
    """K4: block_kurtosis + compute_dagostino2 (src/pb_kernels.cu:140-241), the per-FFT-row statistic the
    reference computes and no output uses; kept with debug_keep and bit-exact against the oracle."""
    lp = libpb()
    res, _, _ = oracle_mode2
    nseg = data.shape[0]
    with lp.PbHandle(nant=1, nbit=8, rfi_mode=2, rows_per_seg=R, max_seg=nseg, debug_keep=True) as h:
        for s in range(nseg):
            h.submit_planar(0, s, data[s, 0], data[s, 1])
        h.process(nseg)
        for s in range(nseg):
            pow_fb, kur_fb = oracle.block_kurtosis(res[s].pow, res[s].kur, res[s].dag)
            dag_fb = oracle.compute_dagostino(kur_fb, which=1)
            assert _same_bits(h.debug_fetch(lp.DBG_POW_FB, 0, s).ravel(), pow_fb)
            assert _same_bits(h.debug_fetch(lp.DBG_KUR_FB, 0, s).ravel(), kur_fb)
            assert _same_bits(h.debug_fetch(lp.DBG_DAG_FB, 0, s), dag_fb[:R])
            assert np.isfinite(pow_fb).all() and (pow_fb > 0).any()


@pytest.mark.parametrize("nbit", [8, 4, 2])
def test_hipfft_backend_codes(oracle, data, nbit):
    lp = libpb()
    g = _run_gpu(lp, data, lp.FFT_HIPFFT, nbit=nbit)
    res, bp_raw, bp_kur = oracle_run(oracle, data, R, rfi_mode=2, npol=1, nbit=nbit)
    per = 8 // nbit
    for name in ("raw", "kur"):
        got = g[name]
        ref = np.concatenate([getattr(r, "codes_" + name) for r in res])
        gs = np.stack([(got >> (nbit * j)) & ((1 << nbit) - 1) for j in range(per)], -1).astype(int)
        rs = np.stack([(ref >> (nbit * j)) & ((1 << nbit) - 1) for j in range(per)], -1).astype(int)
        d = np.abs(gs - rs)
        assert d.max() <= 1, "%s codes differ by more than one step" % name
        frac = (d != 0).mean()
        assert frac < 2e-3, "%s: %.2e of codes differ (FFT rounding should touch only step edges)" % (name, frac)
    for name, ref in (("ave_raw", [r.ave_raw for r in res]), ("ave_kur", [r.ave_kur for r in res])):
        ref = np.concatenate([compact_ave(a, R, 1) for a in ref])
        assert np.abs(g[name] - ref).max() < 2e-4


@pytest.mark.parametrize("rfi_mode,npol", [(0, 1), (1, 1), (2, 2)])
def test_hipfft_backend_modes(oracle, data, rfi_mode, npol):
    lp = libpb()
    g = _run_gpu(lp, data, lp.FFT_HIPFFT, rfi_mode=rfi_mode, npol=npol)
    res, _, _ = oracle_run(oracle, data, R, rfi_mode=rfi_mode, npol=npol, nbit=8)
    names = {0: ("raw",), 1: ("kur",), 2: ("raw", "kur")}[rfi_mode]
    for name in names:
        ref = np.concatenate([getattr(r, "codes_" + name) for r in res]).astype(int)
        d = np.abs(g[name].astype(int) - ref)
        assert d.max() <= 1 and (d != 0).mean() < 2e-3


# ---------------------------------------------------------------------------
# LDS FFT back end: same FFT operation order as the oracle -> everything bit-exact

@pytest.mark.parametrize("keep", [True, False])
@pytest.mark.parametrize("rfi_mode,npol,nbit", [(2, 1, 8), (2, 1, 4), (2, 1, 2), (0, 1, 8), (1, 1, 8),
                                                 (2, 2, 8), (2, 2, 2)])
def test_lds_backend_bit_exact(oracle, data, rfi_mode, npol, nbit, keep):
    """keep=True: kurtosis kernel + channeliser (statistics kept); keep=False: the channeliser that flags its own rows"""
    lp = libpb()
    g = _run_gpu(lp, data, lp.FFT_LDS, rfi_mode=rfi_mode, npol=npol, nbit=nbit, debug_keep=keep)
    if rfi_mode:
        res0, _, _ = oracle_run(oracle, data, R, rfi_mode=rfi_mode, npol=npol, nbit=nbit)
        for s in range(NSEG):          # flags and weights of whichever kernel computed them
            nb = g["flags"][s].size
            assert np.array_equal(g["flags"][s], (res0[s].dag[:nb] > 3.0).astype(np.uint8))
    res, bp_raw, bp_kur = oracle_run(oracle, data, R, rfi_mode=rfi_mode, npol=npol, nbit=nbit)
    names = {0: ("raw",), 1: ("kur",), 2: ("raw", "kur")}[rfi_mode]
    for name in names:
        ref = np.concatenate([getattr(r, "codes_" + name) for r in res])
        assert np.array_equal(g[name], ref), "%s codes differ from the oracle" % name
        refa = np.concatenate([compact_ave(getattr(r, "ave_" + name), R, npol) for r in res])
        assert _same_bits(g["ave_" + name], refa), "%s fp32 plane differs" % name
    # persistent bandpass state (compact 4096 channels of the reference's 6251)
    gr, gk = g["bp"]
    if rfi_mode != 1:
        assert _same_bits(gr, bp_raw.reshape(2, NCHAN)[:, 2155:])
    if rfi_mode == 2:
        assert _same_bits(gk, bp_kur.reshape(2, NCHAN)[:, 2155:])
    if rfi_mode == 1:       # reference aliases bp_kur_dev = bp_dev; we keep it in the kur slot
        assert _same_bits(gk, bp_raw.reshape(2, NCHAN)[:, 2155:])


@pytest.mark.parametrize("keep", [True, False])
@pytest.mark.parametrize("case", ["dropped_segment", "constant_128", "saturated", "one_pol_dead"])
def test_lds_backend_degenerate_inputs(oracle, case, keep):
    """Inputs the live system does meet: a whole segment of dropped frames (zeros -> kurtosis NaN,
    zero power, bandpass initialised from a zero mean), a dead digitiser (constant mid-scale or rail),
    one polarisation missing.  Codes, fp32 planes (NaN bit patterns included) and bandpass state must
    still equal the oracle's."""
    lp = libpb()
    d = make_input(11, R, NSEG, rfi=False, dropped=False)
    if case == "dropped_segment":
        d[0] = 0
        d[2, 0, :] = 0
    elif case == "constant_128":
        d[1] = 128
    elif case == "saturated":
        d[0, 1] = 255
        d[1, 0, 12500 * 3:12500 * 9] = 255
    else:
        d[:, 1] = 0
    g = _run_gpu(lp, d, lp.FFT_LDS, rfi_mode=2, npol=1, nbit=8, debug_keep=keep)
    res, bp_raw, bp_kur = oracle_run(oracle, d, R, rfi_mode=2, npol=1, nbit=8)
    for name in ("raw", "kur"):
        refa = np.concatenate([compact_ave(getattr(r, "ave_" + name), R, 1) for r in res])
        ga = g["ave_" + name]
        # NaNs may differ in payload/sign between a CPU and a GPU divide; they must sit in the same places
        assert np.array_equal(np.isnan(ga), np.isnan(refa)), "%s: NaN positions differ" % name
        ok = ~np.isnan(refa)
        assert _same_bits(ga[ok], refa[ok]), "%s fp32 plane differs" % name
        # ... and the quantiser sends a NaN to the same code on both sides (0 for 8 bits)
        ref = np.concatenate([getattr(r, "codes_" + name) for r in res])
        assert np.array_equal(g[name], ref), "%s codes differ from the oracle" % name
    gr, gk = g["bp"]
    for got, ref in ((gr, bp_raw), (gk, bp_kur)):
        ref = ref.reshape(2, NCHAN)[:, 2155:]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        m = ~np.isnan(ref)
        assert _same_bits(got[m], ref[m])


@pytest.mark.parametrize("keep", [True, False])
@pytest.mark.parametrize("case", ["dropped_segment", "saturated", "odd_rows_dead"])
def test_lds_backend_degenerate_inputs_excised_only(oracle, case, keep):
    """RFI mode 1 (the excised stream alone): rows whose every block is flagged get no transform at all, and
    the workgroup then has to request its next row itself -- at an even row (first of a workgroup's two), at
    an odd one, and for whole segments."""
    lp = libpb()
    d = make_input(11, R, NSEG, rfi=False, dropped=False)
    if case == "dropped_segment":
        d[0] = 0
        d[2, 0, :] = 0
    elif case == "saturated":
        d[0, 1] = 255
        d[1, 0, 12500 * 2:12500 * 3] = 255           # row 2 (even) of segment 1, pol 0
    else:
        for r in (1, 3, 5):
            d[1, :, 12500 * r:12500 * (r + 1)] = 255
    g = _run_gpu(lp, d, lp.FFT_LDS, rfi_mode=1, npol=1, nbit=8, debug_keep=keep)
    res, bp_raw, _ = oracle_run(oracle, d, R, rfi_mode=1, npol=1, nbit=8)
    refa = np.concatenate([compact_ave(r.ave_kur, R, 1) for r in res])
    ga = g["ave_kur"]
    assert np.array_equal(np.isnan(ga), np.isnan(refa))
    ok = ~np.isnan(refa)
    assert _same_bits(ga[ok], refa[ok])
    assert np.array_equal(g["kur"], np.concatenate([r.codes_kur for r in res]))
    ref = bp_raw.reshape(2, NCHAN)[:, 2155:]
    got = g["bp"][1]
    m = ~np.isnan(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and _same_bits(got[m], ref[m])


def test_channelizer_fft_matches_oracle_fft_bitwise(oracle):
    import synth
    lp = libpb()
    x = synth.gauss(77, 5 * 12500).astype(np.float32)
    with lp.PbHandle(rows_per_seg=8, max_seg=1) as h:
        X = h.channelize_f32(x, 5, taps=1)
    ref = oracle.rfft(x)
    assert np.array_equal(X.view(np.uint32), ref.view(np.uint32))


def test_pfb_taps4_matches_reference_polyphase_filterbank(golden, oracle):
    """taps=4 channeliser against the reference's analysis/baseband.py:polyphase_filterbank
    golden (float64 there, fp32 here: tolerance 2e-6 of the spectrum's peak)."""
    import synth
    lp = libpb()
    xs = synth.gauss(int(golden["p2_seed"]), 8 * 50000).astype(np.float32)
    nrows = 28
    with lp.PbHandle(rows_per_seg=8, max_seg=1) as h:
        X = h.channelize_f32(xs[:(nrows + 3) * 12500], nrows, taps=4)
    ref = golden["p2_slice"]
    got = X[:, ::25]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err


# ---------------------------------------------------------------------------
# antenna batching on one GPU + incoherent coadd

@pytest.mark.parametrize("nsets,own_stream", [(1, False), (2, False), (2, True)])
def test_antenna_batch_and_coadd(oracle, nsets, own_stream):
    """nsets=2: detect runs on the library's second stream; the local sum queued right behind
    pb_process (no fetch, no sync in between, as bench.py --gpus N does) must still see its planes."""
    lp = libpb()
    A, nseg = 3, 2
    datas = [make_input(20 + a, R, nseg) for a in range(A)]
    import torch
    with lp.PbHandle(nant=A, nbit=8, rows_per_seg=R, max_seg=nseg, keep_ave=True, nsets=nsets) as h:
        d_sum = torch.zeros(nseg * h.ave_per_seg, dtype=torch.float32, device="cuda")
        cs = torch.cuda.Stream() if own_stream else None
        torch.cuda.synchronize()
        if own_stream:                       # the sum (and the collective) on a stream of its own
            h.set_coadd_stream(cs.cuda_stream)
        for a in range(A):
            for s in range(nseg):
                h.submit_planar(a, s, datas[a][s, 0], datas[a][s, 1])
        h.process(nseg)
        h.coadd_local(nseg, d_sum.data_ptr())
        h.sync()
        summed = d_sum.cpu().numpy()
        outs = [h.fetch(a, 0, nseg, ave=True) for a in range(A)]
        codes = h.coadd_finish(nseg, d_sum.data_ptr(), A)
    planes = []
    for a in range(A):
        res, _, _ = oracle_run(oracle, datas[a], R)
        assert np.array_equal(outs[a]["raw"], np.concatenate([r.codes_raw for r in res]))
        assert np.array_equal(outs[a]["kur"], np.concatenate([r.codes_kur for r in res]))
        planes.append(np.concatenate([compact_ave(r.ave_kur, R, 1) for r in res]))
        assert _same_bits(outs[a]["ave_kur"], planes[a])
    ref = np.zeros_like(planes[0])
    for a in range(A):
        ref = (ref + planes[a]).astype(np.float32)
    assert _same_bits(summed, ref)
    scale = np.float32(1.0 / np.sqrt(float(A)))
    v = (ref * scale).astype(np.float32)
    tmp = (v.astype(np.float64) / 0.02957 + 127.5).astype(np.float32)
    exp = np.where(tmp <= 0, 0, np.where(tmp >= 255, 255, np.floor(tmp))).astype(np.uint8)
    assert np.array_equal(codes, exp)


@pytest.mark.parametrize("nsets,nb,lag", [(2, 3, 1), (3, 7, 2), (2, 6, 0)])
def test_pipelined_buffer_sets_match_serial(oracle, nsets, nb, lag):
    """Batches rotate through nsets buffer sets and are collected `lag` batches later (0: at once);
    kurtosis, channeliser, detect and copy-out of neighbouring batches overlap on four streams, the
    bandpass state still advances in process order, so the concatenated output equals the oracle's
    serial run."""
    lp = libpb()
    nseg = 2
    data = make_input(31, R, nseg * nb)
    got_raw, got_kur = [], []

    def collect(h, b):
        h.select_set(b % nsets)
        got_raw.append(h.fetch_view(0, 0, nseg).copy())
        got_kur.append(h.fetch_view(0, 1, nseg).copy())

    with lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=nseg, nsets=nsets) as h:
        for b in range(nb):
            h.select_set(b % nsets)
            for s in range(nseg):
                h.submit_planar(0, s, data[b * nseg + s, 0], data[b * nseg + s, 1])
            h.process(nseg)
            if b >= lag:
                collect(h, b - lag)
        for b in range(max(0, nb - lag), nb):
            collect(h, b)
    res, _, _ = oracle_run(oracle, data, R)
    assert np.array_equal(np.concatenate(got_raw), np.concatenate([r.codes_raw for r in res]))
    assert np.array_equal(np.concatenate(got_kur), np.concatenate([r.codes_kur for r in res]))


def test_pipelined_buffer_sets_hipfft_backend(oracle):
    """The hipFFT back end over two buffer sets: its kurtosis pass runs on the main stream, so staging into a reused
    set must be ordered behind the latest FFT stage (submit_stream).  Codes within one step of the oracle's on
    < 0.2 % of samples, as for the serial hipFFT runs."""
    lp = libpb()
    nseg, nb, nsets = 2, 5, 2
    data = make_input(33, R, nseg * nb)
    got_raw, got_kur = [], []
    with lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=nseg, nsets=nsets, fft_backend=lp.FFT_HIPFFT) as h:
        for b in range(nb):
            h.select_set(b % nsets)
            for s in range(nseg):
                h.submit_planar(0, s, data[b * nseg + s, 0], data[b * nseg + s, 1])
            h.process(nseg)
            if b >= 1:
                h.select_set((b - 1) % nsets)
                got_raw.append(h.fetch_view(0, 0, nseg).copy())
                got_kur.append(h.fetch_view(0, 1, nseg).copy())
        h.select_set((nb - 1) % nsets)
        got_raw.append(h.fetch_view(0, 0, nseg).copy())
        got_kur.append(h.fetch_view(0, 1, nseg).copy())
    res, _, _ = oracle_run(oracle, data, R)
    for got, name in ((got_raw, "codes_raw"), (got_kur, "codes_kur")):
        ref = np.concatenate([getattr(r, name) for r in res]).astype(int)
        d = np.abs(np.concatenate(got).astype(int) - ref)
        assert d.max() <= 1 and (d != 0).mean() < 2e-3


@pytest.mark.parametrize("keep", [False, True])
@pytest.mark.parametrize("nsets,rfi_mode", [(2, 2), (3, 2), (2, 1)])
def test_pipelined_sets_with_32_row_chunks_bit_exact(oracle, nsets, rfi_mode, keep):
    """R = 64: detect's 32-row chunks, and -- pipelined, with the channeliser that flags its own rows -- its
    three-chunks-in-flight ring (launch_detect_pow picks DEPTH 3 there, 2 everywhere else); five batches through
    reused buffer sets, collected nsets - 1 batches late, against the oracle's serial run.  keep: the two-kernel path
    (with three sets its kurtosis pass runs beside the previous batch's channeliser)."""
    lp = libpb()
    Rr, nseg, nb = 64, 2, 5
    data = make_input(37, Rr, nseg * nb)
    got = {"raw": [], "kur": []}

    def collect(h, b):
        h.select_set(b % nsets)
        o = h.fetch(0, 0, nseg, raw=rfi_mode != 1, kur=rfi_mode != 0)
        for k in got:
            got[k].append(o[k])

    with lp.PbHandle(nbit=8, rfi_mode=rfi_mode, rows_per_seg=Rr, max_seg=nseg, nsets=nsets, debug_keep=keep) as h:
        for b in range(nb):
            h.select_set(b % nsets)
            for s in range(nseg):
                h.submit_planar(0, s, data[b * nseg + s, 0], data[b * nseg + s, 1])
            h.process(nseg)
            if b >= nsets - 1:
                collect(h, b - (nsets - 1))
        for b in range(max(0, nb - (nsets - 1)), nb):
            collect(h, b)
    res, _, _ = oracle_run(oracle, data, Rr, rfi_mode=rfi_mode)
    if rfi_mode != 1:
        assert np.array_equal(np.concatenate(got["raw"]), np.concatenate([r.codes_raw for r in res]))
    assert np.array_equal(np.concatenate(got["kur"]), np.concatenate([r.codes_kur for r in res]))


def test_flag_decision_without_the_cube_root_equals_the_score_for_every_float(oracle):
    """The kernels decide "D'Agostino score > 3" from the cube root's argument against crossings located at
    pb_create (the score itself only in the bands around them).  Exhaustive where it matters: every binary32
    kurtosis in [1, 16] (all crossings of the N = 500 score lie there: 3 +- ~0.66) -- 33.5 million values -- plus
    ranges far out on both sides, decision against score on the device: no mismatch.  And the crossings the
    library found agree with the oracle's score to the float."""
    lp = libpb()
    with lp.PbHandle(rows_per_seg=8, max_seg=1) as h:
        total = 0
        for lo, hi in ((1.0, 16.0), (0.01, 0.0101), (50.0, 50.5), (1e6, 1.001e6), (1e-30, 1.001e-30)):
            n, bad, bands = h.dag_check(lo, hi)
            assert bad == 0, (lo, hi, bad)
            total += n
        assert total > 3.3e7
    t_lo_sure, t_lo_clear, t_hi_clear, t_hi_sure = [np.float32(v) for v in bands]
    assert 0 < t_lo_sure < t_lo_clear < t_hi_clear < t_hi_sure
    # the bands are a few floats wide at most
    assert int(t_lo_clear.view(np.uint32)) - int(t_lo_sure.view(np.uint32)) <= 8
    assert int(t_hi_sure.view(np.uint32)) - int(t_hi_clear.view(np.uint32)) <= 8
    # against the oracle: kurtosis values whose t lands just outside the bands
    kur = np.linspace(1.5, 6.0, 200001).astype(np.float32)
    dag = oracle.compute_dagostino(np.concatenate([kur, kur]))[:kur.size]
    fl = dag > 3.0
    # crossings in kurtosis: flagged below ~2.45 and above ~3.78, clear in between
    i0, i1 = np.argmax(~fl), kur.size - np.argmax(~fl[::-1])
    assert fl[:i0].all() and (~fl[i0:i1]).all() and fl[i1:].all() and 2.2 < kur[i0] < 2.7 and 3.5 < kur[i1 - 1] < 4.1
