"""An independent float64 NumPy model of K7-K11 (src/pb_kernels.cu:393-735) against the C oracle.

The C oracle (oracle/pb_oracle.c) restates the reference's CUDA kernels statement by statement in binary32; nothing
in the reference pins it (no fixture, no runnable binary).  This test is a second reading of the same source,
written the other way round -- vectorised over channels, explicit over time, float64 throughout -- so that a
misread index, weight convention or branch in one of the two shows up as a gross difference.  It is evidence, not
a pin: agreement is to float tolerance (1e-4 of the unit-variance planes; 8-bit codes equal except where the
pre-quantiser value sits within that tolerance of a step edge)."""
import ctypes as C

import numpy as np

NCHAN, CHANMIN, NCHANOUT, NSCRUNCH, MIN_WEIGHT = 6251, 2155, 4096, 8, 0.2


def _model_raw(X, bp, scale):
    """detect_and_normalize2 :393-429, pscrunch :514-524, tscrunch :564-589.  X complex [2][R][NCHAN]."""
    R = X.shape[1]
    out = np.empty((2, R, NCHAN))
    for pol in range(2):
        pw = X[pol].real.astype(np.float64) ** 2 + X[pol].imag.astype(np.float64) ** 2
        b = bp[pol].astype(np.float64).copy()
        init = b == 0
        b[init] = pw[:, init].sum(axis=0) / R
        for t in range(R):
            b = scale * pw[t] + (1 - scale) * b
            out[pol, t] = pw[t] / b - 1
        bp[pol] = b
    ps = np.sqrt(0.5) * (out[0] + out[1])
    return ps.reshape(R // NSCRUNCH, NSCRUNCH, NCHAN).sum(axis=1) * np.sqrt(1.0 / NSCRUNCH)


def _model_kur(X, w, bp, scale):
    """detect_and_normalize3 :431-511, pscrunch_weights :527-560 (pre-kernel weights), tscrunch_weights :591-630.
    w: row weights [R] (both pols share them on this path)."""
    R = X.shape[1]
    out = np.zeros((2, R, NCHAN))
    good = w != 0
    for pol in range(2):
        s = X[pol].real.astype(np.float64) ** 2 + X[pol].imag.astype(np.float64) ** 2
        b = bp[pol].astype(np.float64).copy()
        init = b == 0
        if good.any():
            b[init] = (s[good][:, init] / w[good, None]).sum(axis=0) / good.sum()
        else:
            b[init] = 1.0
        for t in range(R):
            if not good[t]:
                continue                       # x = 0, bandpass untouched
            pw = s[t] / w[t]
            clip = pw > b * 11
            bn = scale * pw + (1 - scale) * b
            b = np.where(clip, b, bn)
            out[pol, t] = np.where(clip, 10.0, pw / b - 1)
        bp[pol] = b
    ok = w >= MIN_WEIGHT                       # both pols' weights are equal: cases 2 and 0 of the switch only
    ps = np.where(ok[:, None], np.sqrt(0.5) * (out[0] + out[1]), 0.0)
    w2 = np.where(ok, w, 0.0)                  # 0.5 (w + w) = w
    ave = np.zeros((R // NSCRUNCH, NCHAN))
    for g in range(R // NSCRUNCH):
        rows = slice(g * NSCRUNCH, (g + 1) * NSCRUNCH)
        use = w2[rows] >= MIN_WEIGHT
        if w2[rows][use].sum() / NSCRUNCH >= MIN_WEIGHT:
            ave[g] = (w2[rows][use, None] * ps[rows][use]).sum(axis=0) / np.sqrt(use.sum())
    return ave


def _codes8(ave):
    """sel_and_dig_8b :711-735 on the 4096 output channels"""
    tmp = ave[:, CHANMIN:CHANMIN + NCHANOUT] / 0.02957 + 127.5
    return np.where(tmp <= 0, 0, np.where(tmp >= 255, 255, np.floor(np.clip(tmp, 0, 255)))).astype(np.uint8)


def test_numpy_model_of_detect_scrunch_digitise_agrees_with_the_c_oracle(oracle):
    L = oracle.lib()
    fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    rng = np.random.default_rng(7)
    R, nseg = 32, 3
    scale = np.float32((12500.0 / 128000000 * 8) / 1.0)
    bp_c = [np.zeros(2 * NCHAN, np.float32), np.zeros(2 * NCHAN, np.float32)]
    bp_m = [np.zeros((2, NCHAN)), np.zeros((2, NCHAN))]
    nbad_codes = 0
    for s in range(nseg):
        X = (rng.standard_normal((2, R, NCHAN)) + 1j * rng.standard_normal((2, R, NCHAN))).astype(np.complex64)
        X[:, 5, 3000:3010] *= 9.0                      # a burst that the 11x clip catches
        w = np.ones(R, np.float32)
        w[3] = 0.0                                     # no data
        w[7] = 0.12                                    # below MIN_WEIGHT
        w[12] = 0.6
        if s == 1:
            w[16:24] = [0.0, 0.0, 0.16, 0.0, 0.0, 0.0, 0.0, 0.0]       # a time sample with too little weight
        # ---- C oracle, raw stream
        f = np.ascontiguousarray(X).view(np.float32).copy()
        L.orc_detect_and_normalize2(f.ctypes.data_as(fp), bp_c[0].ctypes.data_as(fp), C.c_float(scale), R)
        L.orc_pscrunch(f.ctypes.data_as(fp), R * NCHAN)
        ave_c = np.zeros(R // 8 * NCHAN, np.float32)
        L.orc_tscrunch(f.ctypes.data_as(fp), ave_c.ctypes.data_as(fp), ave_c.size)
        codes_c = np.zeros(R // 8 * NCHANOUT, np.uint8)
        L.orc_sel_and_dig_8b(ave_c.ctypes.data_as(fp), codes_c.ctypes.data_as(u8p), codes_c.size, 1, R // 8)
        # ---- model, raw stream
        ave_m = _model_raw(X, bp_m[0], float(scale))
        assert np.abs(ave_c.reshape(-1, NCHAN) - ave_m).max() < 2e-4
        assert np.allclose(bp_c[0].reshape(2, NCHAN), bp_m[0], rtol=2e-5)
        d = np.abs(codes_c.reshape(-1, NCHANOUT).astype(int) - _codes8(ave_m).astype(int))
        assert d.max() <= 1
        nbad_codes += int((d != 0).sum())
        # ---- C oracle, excised stream
        fk = np.ascontiguousarray(X).view(np.float32).copy()
        kw = np.ascontiguousarray(np.concatenate([w, w]))
        L.orc_detect_and_normalize3(fk.ctypes.data_as(fp), kw.ctypes.data_as(fp), bp_c[1].ctypes.data_as(fp), C.c_float(scale), R)
        L.orc_pscrunch_weights(fk.ctypes.data_as(fp), kw.ctypes.data_as(fp), R * NCHAN)
        avek_c = np.zeros(R // 8 * NCHAN, np.float32)
        L.orc_tscrunch_weights(fk.ctypes.data_as(fp), avek_c.ctypes.data_as(fp), kw.ctypes.data_as(fp), avek_c.size)
        codesk_c = np.zeros(R // 8 * NCHANOUT, np.uint8)
        L.orc_sel_and_dig_8b(avek_c.ctypes.data_as(fp), codesk_c.ctypes.data_as(u8p), codesk_c.size, 1, R // 8)
        # ---- model, excised stream
        avek_m = _model_kur(X, w.astype(np.float64), bp_m[1], float(scale))
        assert np.abs(avek_c.reshape(-1, NCHAN) - avek_m).max() < 5e-4
        assert np.allclose(bp_c[1].reshape(2, NCHAN), bp_m[1], rtol=2e-5)
        dk = np.abs(codesk_c.reshape(-1, NCHANOUT).astype(int) - _codes8(avek_m).astype(int))
        assert dk.max() <= 1
        nbad_codes += int((dk != 0).sum())
        if s == 1:
            assert (avek_c.reshape(-1, NCHAN)[2] == 0).all()            # the under-weighted time sample is zeroed
        assert (avek_m != ave_m).any()
    assert nbad_codes < 1e-3 * 2 * nseg * (R // 8) * NCHANOUT            # only values on a step edge may differ


def test_numpy_model_of_kurtosis_flags_and_weights_agrees_with_the_c_oracle(oracle):
    """K1-K3 + K5 (src/pb_kernels.cu:23-134, :243-295): convertarray, per-block power and (non-central) kurtosis,
    the D'Agostino score as scipy computes it from the kurtosis, max over pols, flags above 3, row weights =
    unflagged fraction -- float64 NumPy against the binary32 oracle on a segment with RFI and dropped frames."""
    import scipy.stats as st
    from helpers import make_input
    R = 16
    data = make_input(21, R, 2)
    bp0, bp1 = np.zeros(2 * NCHAN, np.float32), np.zeros(2 * NCHAN, np.float32)
    nflag = 0
    for s in range(2):
        r = oracle.segment(data[s], R, bp0, bp1)
        u = data[s].astype(np.float64)
        x = np.where(u == 0, 0.0, u / 128.0 - 1.0).reshape(2, R * 25, 500)
        pw = (x ** 2).mean(axis=2)
        with np.errstate(invalid="ignore", divide="ignore"):
            kur = (x ** 4).mean(axis=2) / pw ** 2
        nb = R * 25
        assert np.allclose(r.pow.reshape(2, nb), pw, rtol=2e-5)
        ok = np.isfinite(kur)
        assert np.array_equal(np.isnan(r.kur.reshape(2, nb)), ~ok)                # all-zero blocks: 0/0
        assert np.allclose(r.kur.reshape(2, nb)[ok], kur[ok], rtol=5e-5)
        # Anscombe-Glynn z-score of a kurtosis b2 for n = 500 (what scipy.stats.kurtosistest evaluates)
        n = 500.0
        E = 3.0 * (n - 1) / (n + 1)
        var = 24.0 * n * (n - 2) * (n - 3) / ((n + 1) ** 2 * (n + 3) * (n + 5))
        xs = (kur - E) / np.sqrt(var)
        sb = 6.0 * (n * n - 5 * n + 2) / ((n + 7) * (n + 9)) * np.sqrt(6.0 * (n + 3) * (n + 5) / (n * (n - 2) * (n - 3)))
        A = 6.0 + 8.0 / sb * (2.0 / sb + np.sqrt(1 + 4.0 / sb ** 2))
        with np.errstate(invalid="ignore"):
            term = (1 - 2.0 / A) / (1 + xs * np.sqrt(2.0 / (A - 4.0)))
            z = np.abs((1 - 2.0 / (9 * A) - np.cbrt(term)) / np.sqrt(2.0 / (9 * A)))
        z = np.where(ok & (term > 0), z, 9.0)                                       # DAG_INF where the score is undefined
        dag = np.maximum(z[0], z[1])
        got = r.dag[:nb]
        close = np.abs(got - dag) < 1e-3 * np.maximum(1.0, dag)
        assert close.mean() > 0.999 and np.abs(got - dag)[~close].max(initial=0) < 0.05
        flags = dag > 3.0
        edge = np.abs(dag - 3.0) < 1e-3
        assert np.array_equal((got > 3.0)[~edge], flags[~edge])
        nflag += int(flags.sum())
        wm = (~(got > 3.0)).reshape(R, 25).sum(axis=1) * 0.04
        # the oracle returns kur_weights as tscrunch_weights sees them: pscrunch_weights (:527-560) has rewritten
        # rows 0..R-1 (zero below MIN_WEIGHT) and left the second pol's copy, rows R..2R-1, as apply_kurtosis wrote it
        assert np.allclose(r.weights[R:2 * R], wm, atol=1e-6)
        assert np.allclose(r.weights[:R], np.where(wm >= MIN_WEIGHT - 1e-9, wm, 0.0), atol=1e-6)
        # spot check of the formula against scipy itself on one clean block (zero-mean copy, see test_oracle_golden)
        h = x[0, 7, :250] - x[0, 7, :250].mean()
        blk = np.concatenate([h, -h])
        zk = st.kurtosistest(blk).statistic
        b2 = (blk ** 4).mean() / (blk ** 2).mean() ** 2
        xs1 = (b2 - E) / np.sqrt(var)
        t1 = (1 - 2.0 / A) / (1 + xs1 * np.sqrt(2.0 / (A - 4.0)))
        assert abs((1 - 2.0 / (9 * A) - np.cbrt(t1)) / np.sqrt(2.0 / (9 * A)) - zk) < 1e-9
    assert nflag > 20


def test_numpy_model_of_the_4_and_2_bit_packers_agrees_with_the_c_oracle(oracle):
    """sel_and_dig_4b :672-708 (two samples per byte, the first in the low nibble) and sel_and_dig_2b :633-669 (four
    per byte, the first in the low bits; thresholds -0.6109 / 0.3970 / 1.4050), one and two polarisations
    ([time][pol][channel] byte order for npol 2): vectorised NumPy against the oracle, values kept away from the
    decision levels by construction so that the comparison is exact."""
    L = oracle.lib()
    fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    rng = np.random.default_rng(11)
    ntime = 4
    for npol in (1, 2):
        ave = (rng.standard_normal((npol, ntime, NCHAN)) * 1.3).astype(np.float32)
        # keep clear of the 2-bit levels and of the 4-bit step edges
        for lvl in (-0.6109, 0.3970, 1.4050):
            ave[np.abs(ave - lvl) < 1e-3] += 0.01
        t4 = ave.astype(np.float64) / 0.3188 + 7.5
        ave[np.abs(t4 - np.round(t4)) < 1e-3] += 0.002
        sel = ave[:, :, CHANMIN:CHANMIN + NCHANOUT].transpose(1, 0, 2).reshape(-1)       # [time][pol][channel]
        # 4 bits
        q4 = np.clip(np.floor(sel.astype(np.float64) / 0.3188 + 7.5), 0, 15).astype(np.uint8)
        exp4 = (q4[0::2] | (q4[1::2] << 4)).astype(np.uint8)
        got4 = np.zeros(exp4.size, np.uint8)
        L.orc_sel_and_dig_4b(ave.ctypes.data_as(fp), got4.ctypes.data_as(u8p), got4.size, npol, ntime)
        assert np.array_equal(got4, exp4)
        # 2 bits
        q2 = np.digitize(sel.astype(np.float64), [-0.6109, 0.3970, 1.4050]).astype(np.uint8)
        exp2 = (q2[0::4] | (q2[1::4] << 2) | (q2[2::4] << 4) | (q2[3::4] << 6)).astype(np.uint8)
        got2 = np.zeros(exp2.size, np.uint8)
        L.orc_sel_and_dig_2b(ave.ctypes.data_as(fp), got2.ctypes.data_as(u8p), got2.size, npol, ntime)
        assert np.array_equal(got2, exp2)
        assert len(np.unique(q2)) == 4 and q4.min() == 0 and q4.max() == 15
