"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (vlite-fast_amd/) never does.

Two halves:

* ctypes bindings to oracle/liboracle.so (pb_oracle.c), the C restatement of
  the reference's CUDA kernels src/pb_kernels.cu (K1..K11) and the segment
  driver src/process_baseband.cu:1108-1376.
* NumPy restatements of the reference's Python analysis functions
  (/root/reference/analysis):
    filterbank             baseband.py:960-989      (P1, the CPU baseline)
    polyphase_filterbank   baseband.py:1207-1237    (P2, defines taps=4)
    vdif_get_data          baseband.py:221-298      (P3)
    VDIFHeader fields      baseband.py:17-28        (H8)
    dedisperse / optimize_pulse / chan_mask   loc_step0.py:44-66,111-147   (S1)
    inplace_roll / tophat_smooth / qn         utils.py:4-17,74-122,187-195 (S1)
  These are pinned by golden vectors produced by importing the reference itself
  (tests/golden/make_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NFFT = 12500
NCHAN = 6251
NSCRUNCH = 8
NKURTO = 500
CHANMIN = 2155
CHANMAX = 6250
NCHANOUT = CHANMAX - CHANMIN + 1
VLITE_RATE = 128000000
FRAMESPERSEC = 25600
VD_FRM = 5032
VD_DAT = 5000


def build():
    """Compile oracle/liboracle.so (building the checker is not using it)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


_VARIANT = "strict"
_LIBS = {}


def set_variant(name):
    """"strict" (default; the oracle the HIP path is held to, bit for bit) or "fmad" (a*b+c fused
    where nvcc's default may fuse it -- sensitivity measurements only, pb_oracle.c header)."""
    global _VARIANT, _LIB
    if name not in ("strict", "fmad"):
        raise ValueError(name)
    _VARIANT = name
    _LIB = _LIBS.get(name)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so" if _VARIANT == "strict" else "liboracle_fmad.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_is_fmad.restype = C.c_int
        assert L.orc_is_fmad() == (1 if _VARIANT == "fmad" else 0)
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_uint8)
        L.orc_powf_third.restype = C.c_float
        L.orc_powf_third.argtypes = [C.c_float]
        L.orc_dag_constants.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.orc_convertarray.argtypes = [fp, up, C.c_size_t]
        L.orc_kurtosis.argtypes = [fp, fp, fp, C.c_size_t]
        L.orc_compute_dagostino.argtypes = [fp, fp, C.c_size_t, C.c_int]
        L.orc_block_kurtosis.argtypes = [fp, fp, fp, fp, fp, C.c_size_t]
        L.orc_apply_kurtosis.argtypes = [fp, fp, fp, fp, C.c_size_t]
        L.orc_rfft.argtypes = [fp, fp, C.c_size_t]
        L.orc_set_frb_delays.argtypes = [fp, C.c_float, C.c_int]
        L.orc_inject_frb.argtypes = [fp, fp, C.c_int, C.c_float, C.c_float, C.c_int]
        L.orc_detect_and_normalize2.argtypes = [fp, fp, C.c_float, C.c_int]
        L.orc_detect_and_normalize3.argtypes = [fp, fp, fp, C.c_float, C.c_int]
        L.orc_pscrunch.argtypes = [fp, C.c_size_t]
        L.orc_pscrunch_weights.argtypes = [fp, fp, C.c_size_t]
        L.orc_tscrunch.argtypes = [fp, fp, C.c_size_t]
        L.orc_tscrunch_weights.argtypes = [fp, fp, fp, C.c_size_t]
        for n in ("orc_sel_and_dig_8b", "orc_sel_and_dig_4b", "orc_sel_and_dig_2b"):
            getattr(L, n).argtypes = [fp, up, C.c_size_t, C.c_int, C.c_int]
        L.orc_fft_table_ptr.restype = C.c_void_p
        L.orc_fft_table_ptr.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
        L.orc_segment.restype = C.c_int
        L.orc_segment.argtypes = [up, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.c_int,
                                  up, up, fp, fp, fp, fp, fp, fp]
        _LIB = L
        _LIBS[_VARIANT] = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def powf_third(t):
    return float(lib().orc_powf_third(C.c_float(t)))


def dag_constants(which=0):
    out = (C.c_double * 7)()
    lib().orc_dag_constants(which, out)
    return dict(zip(("mu1", "mu2", "g1", "A", "Z1", "Z2", "Z3"), list(out)))


def convertarray(u):
    u = np.ascontiguousarray(u, dtype=np.uint8)
    out = np.empty(u.size, dtype=np.float32)
    lib().orc_convertarray(_f(out), _u(u), u.size)
    return out.reshape(u.shape)


def kurtosis(time):
    time = np.ascontiguousarray(time, dtype=np.float32).ravel()
    nb = time.size // NKURTO
    pow_ = np.empty(nb, np.float32)
    kur = np.empty(nb, np.float32)
    lib().orc_kurtosis(_f(time), _f(pow_), _f(kur), nb)
    return pow_, kur


def compute_dagostino(kur, which=0):
    kur = np.ascontiguousarray(kur, dtype=np.float32)
    dag = np.empty_like(kur)
    lib().orc_compute_dagostino(_f(kur), _f(dag), kur.size // 2, which)
    return dag


def block_kurtosis(pow_, kur, dag):
    """K4 (src/pb_kernels.cu:140-212): per-FFT-row pow / kur from the per-block statistics of both
    pols; returns (pow_block, kur_block), 2 * nrows entries each."""
    pow_ = np.ascontiguousarray(pow_, np.float32)
    kur = np.ascontiguousarray(kur, np.float32)
    dag = np.ascontiguousarray(dag, np.float32)
    nfft = pow_.size // (NFFT // NKURTO)
    pb = np.zeros(nfft, np.float32)
    kb = np.zeros(nfft, np.float32)
    lib().orc_block_kurtosis(_f(pow_), _f(kur), _f(dag), _f(pb), _f(kb), nfft)
    return pb, kb


def apply_kurtosis(time, dag):
    time = np.ascontiguousarray(time, dtype=np.float32).ravel()
    out = np.empty_like(time)
    nb = time.size // NKURTO
    norms = np.zeros(time.size // NFFT, np.float32)
    lib().orc_apply_kurtosis(_f(time), _f(out), _f(np.ascontiguousarray(dag, np.float32)),
                             _f(norms), nb)
    return out, norms


def rfft(x):
    """x: float32 [nffts*12500] -> complex64 [nffts, 6251] (oracle's fixed-order FFT)."""
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    n = x.size // NFFT
    out = np.empty((n, NCHAN), dtype=np.complex64)
    lib().orc_rfft(_f(x), out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out


def fft_table(which):
    nbytes = C.c_size_t()
    p = lib().orc_fft_table_ptr(which, C.byref(nbytes))
    buf = (C.c_char * nbytes.value).from_address(p)
    return np.frombuffer(buf, dtype=np.float32).copy()


def set_frb_delays(dm, nrows=1024):
    d = np.empty(NCHAN, np.float32)
    lib().orc_set_frb_delays(_f(d), C.c_float(dm), nrows)
    return d


class SegmentResult(object):
    pass


def trim_bytes(nrows, npol, nbit):
    polfac = 2 if npol == 1 else 1
    return 2 * nrows * NCHANOUT // (polfac * NSCRUNCH) // (8 // nbit)


def scrunch_len(nrows, npol):
    polfac = 2 if npol == 1 else 1
    return 2 * nrows * NCHAN // (polfac * NSCRUNCH)


def segment(udat, nrows, bp_raw, bp_kur, rfi_mode=2, npol=1, nbit=8, frb_delays=None,
            inject_now=0, want_stats=True):
    """One segment through the reference chain.  udat: u8 [2, nrows*12500].
    bp_raw / bp_kur: float32 [2*6251], updated in place (persistent state)."""
    udat = np.ascontiguousarray(udat, dtype=np.uint8)
    assert udat.size == 2 * nrows * NFFT
    assert bp_raw.dtype == np.float32 and bp_raw.size == 2 * NCHAN
    assert bp_kur.dtype == np.float32 and bp_kur.size == 2 * NCHAN
    r = SegmentResult()
    tb = trim_bytes(nrows, npol, nbit)
    sl = scrunch_len(nrows, npol)
    nblk = 2 * nrows * NFFT // NKURTO
    r.codes_raw = np.zeros(tb, np.uint8)
    r.codes_kur = np.zeros(tb, np.uint8)
    r.ave_raw = np.zeros(sl, np.float32)
    r.ave_kur = np.zeros(sl, np.float32)
    r.weights = np.zeros(2 * nrows, np.float32)
    r.dag = np.zeros(nblk, np.float32) if want_stats else None
    r.pow = np.zeros(nblk, np.float32) if want_stats else None
    r.kur = np.zeros(nblk, np.float32) if want_stats else None
    rc = lib().orc_segment(_u(udat), nrows, rfi_mode, npol, nbit, _f(bp_raw), _f(bp_kur),
                           _f(frb_delays), inject_now, _u(r.codes_raw), _u(r.codes_kur),
                           _f(r.ave_raw), _f(r.ave_kur), _f(r.weights), _f(r.dag), _f(r.pow),
                           _f(r.kur))
    if rc != 0:
        raise ValueError("orc_segment rejected its arguments")
    return r


def sel_and_dig(fft_ave, nrows, npol=1, nbit=8):
    """K11 alone: the full [(pol)][nrows/8][6251] fp32 plane -> filterbank bytes (orc_sel_and_dig_8b/4b/2b,
    src/pb_kernels.cu:711-735, :672-708, :633-669).  Used by the coadd tests: the root requantises the summed plane."""
    fft_ave = np.ascontiguousarray(fft_ave, dtype=np.float32).ravel()
    assert fft_ave.size == scrunch_len(nrows, npol)
    out = np.zeros(trim_bytes(nrows, npol, nbit), np.uint8)
    fn = {8: "orc_sel_and_dig_8b", 4: "orc_sel_and_dig_4b", 2: "orc_sel_and_dig_2b"}[nbit]
    getattr(lib(), fn)(_f(fft_ave), _u(out), out.size, npol, nrows // 8)
    return out


# --------------------------------------------------------------------------
# NumPy restatements of the reference's Python analysis code

def filterbank(samples, nfft=NFFT):
    """P1: |rfft|^2 per block, float64.  analysis/baseband.py:960-989 (detect=True,
    favg=0, tavg=False)."""
    samples = np.asarray(samples)
    nblk = samples.shape[0] // nfft
    out = np.empty((nblk, nfft // 2 + 1), dtype=np.float64)
    for i in range(nblk):
        out[i] = np.abs(np.fft.rfft(samples[i * nfft:(i + 1) * nfft])) ** 2
    return out


def hamming_sym(n):
    """scipy.signal.hamming(n) (symmetric): 0.54 - 0.46 cos(2 pi k/(n-1))."""
    k = np.arange(n, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * k / (n - 1))


def pfb_coefficients(nchan=NFFT // 2, nwindow=4):
    """FIR taps of polyphase_filterbank, analysis/baseband.py:1212-1218,1230-1232:
    tap j carries window[j*ns:(j+1)*ns] * norms[0] * (norms[j] if j else 1)."""
    ns = 2 * nchan
    window = hamming_sym(nwindow * ns)
    norms = [1. / np.sum(window[j * ns:(j + 1) * ns] ** 2) for j in range(nwindow)]
    taps = np.empty((nwindow, ns), dtype=np.float64)
    for j in range(nwindow):
        taps[j] = window[j * ns:(j + 1) * ns] * norms[0] * (norms[j] if j else 1.0)
    return taps


def polyphase_filterbank(samples, nchan=64, nwindow=4):
    """P2: real-input WOLA PFB.  analysis/baseband.py:1207-1237."""
    samples = np.asarray(samples)
    ns = 2 * nchan
    window = hamming_sym(nwindow * ns)
    norms = [1. / np.sum(window[j * ns:(j + 1) * ns] ** 2) for j in range(nwindow)]
    nwindows = len(samples) // (nwindow * ns) - 1
    nspectra = nwindows * nwindow
    out = np.empty((nspectra, nchan + 1), dtype=np.complex64)
    for i in range(nspectra):
        i0 = i * ns
        tmp = window * samples[i0:i0 + nwindow * ns] * norms[0]
        for j in range(1, nwindow):
            tmp[:ns] += tmp[j * ns:(j + 1) * ns] * norms[j]
        out[i] = np.fft.rfft(tmp[:ns])
    return out


def vdif_header_fields(words):
    """H8: analysis/baseband.py:19-28.  words: 8 little-endian uint32."""
    d = np.asarray(words, dtype=np.uint32)
    frame_length = int(d[2] & (2 ** 24 - 1)) * 8
    threadid = int((d[3] & (2 ** 26 - 2 ** 16)) >> 16)
    return dict(second=int(d[0] & (2 ** 30 - 1)),
                epoch=int((d[1] & (2 ** 30 - 2 ** 24)) >> 24),
                frame=int(d[1] & (2 ** 24 - 1)),
                frame_length=frame_length, frame_nsamp=frame_length - 32,
                station=int(d[3] & (2 ** 16 - 1)), threadid=threadid,
                thread=int(threadid != 0))


def vdif_get_data(raw, first_thread=0):
    """P3: BasebandFragment.get_data(thread=-1), analysis/baseband.py:269-298:
    strip 32-B headers, de-interleave even/odd frames, float32 minus 127.5."""
    raw = np.asarray(raw, dtype=np.uint8)
    nframe = raw.size // VD_FRM
    by_frames = raw[:nframe * VD_FRM].reshape(nframe, VD_FRM)[:, 32:]
    t0 = by_frames[::2].reshape(-1)
    t1 = by_frames[1::2].reshape(-1)
    if first_thread == 1:
        t0, t1 = t1, t0
    n = min(t0.size, t1.size)
    buff = np.empty((2, n), dtype=np.float32)
    buff[0] = t0[:n]
    buff[1] = t1[:n]
    buff -= 127.5
    return buff


def inplace_roll(a, shift):
    """analysis/utils.py:4-17"""
    if shift == 0:
        return a
    mshift = a.shape[-1] - abs(shift)
    if shift < 0:
        shift = -shift
        shift, mshift = mshift, shift
    tmp = a[..., mshift:].copy()
    a[..., shift:] = a[..., :mshift]
    a[..., :shift] = tmp
    return a


def get_vlite_chan_freqs(nchan):
    """analysis/loc_step0.py:40-42"""
    return ((np.arange(nchan)) * 64. / nchan + 320)[::-1]


def dedisperse(array, dm, tsamp, ref_freq=320):
    """analysis/loc_step0.py:44-66 (antenna x channel x time, in place)."""
    nchan = array.shape[1]
    chan_freqs = get_vlite_chan_freqs(nchan)
    sample_delays = -np.round(
        dm * 4.15e-3 * ((chan_freqs * 1e-3) ** -2 - (ref_freq * 1e-3) ** -2) / tsamp).astype(int)
    for ichan in range(nchan):
        inplace_roll(array[:, ichan], sample_delays[ichan])


def chan_mask():
    """analysis/loc_step0.py:111-118"""
    mask = np.zeros(6251)
    mask[2350:6200] = 1
    mask[3123:3128] = 0
    mask[4297] = 0
    mask[4988:4992] = 0
    return mask


def tophat_smooth(a, n):
    """analysis/utils.py:74-122, 1-D branch."""
    if n % 2 == 0:
        n += 1
    lena = a.shape[0]
    if lena < n:
        return a
    c = np.cumsum(a, axis=0)
    out = np.empty_like(a)
    out[n // 2 + 1:len(a) - n // 2] = (c[n:] - c[:len(a) - n]) * (1. / n)
    out[:n // 2 + 1] = c[:n:2] / np.arange(1, n + 1, 2)
    out[len(a) - n // 2:] = (c[-1] - c[len(a) - n + 1::2]) / np.arange(1, n, 2)[::-1]
    return out


def qn(s):
    """analysis/utils.py:187-195: 2.2219 x first quartile of pairwise |differences|."""
    s = np.asarray(s)
    iu = np.triu_indices(len(s), k=1)
    diffs = np.abs(s[iu[0]] - s[iu[1]])
    return 2.2219 * np.percentile(diffs, 25)


def optimize_pulse(ts, i0, i1, wmax=32):
    """analysis/loc_step0.py:120-147"""
    widths = np.arange(1, wmax + 1, 2)
    nsamp = i1 - i0
    s = np.append(ts[i0:i0 + int(nsamp * 0.25)], ts[i0 + int(nsamp * 0.75):i1])
    mean = np.median(s)
    std = qn(s)
    sns = np.zeros(len(widths))
    locs = np.zeros(len(widths))
    for iw, w in enumerate(widths):
        sts = tophat_smooth(ts[i0:i1], w)[w:-w]
        a = np.argmax(sts)
        locs[iw] = a + w
        sns[iw] = (sts[a] - mean) / std * w ** 0.5
    return widths, sns, locs


# ------------------------------------------------------------------------------------------------------------------
# The downstream search's checker (BASELINE config 5).  heimdall / dedisp are third-party and absent from
# /root/reference (parity with heimdall's candidate list is UNPINNED); what the reference itself holds
# about this stage is restated here and used by tests/test_gpu_search.py:
#   * incoherent dedispersion = every channel shifted by its dispersion delay, then summed over channels
#     (analysis/loc_step0.py:44-66, the roll; the per-channel delay formula with the constant of src/candidate.py:33);
#   * the S/N estimator: off-pulse median and Qn, (running mean - median) / Qn * sqrt(w)
#     (analysis/loc_step0.py:120-147 with analysis/utils.py:74-122, 187-195);
#   * the candidate line's nine columns (src/candidate.py:8-18).

DM_CONST = 4.148808e3          # src/candidate.py:33: dm_delay = 4.148808e3 * dm * |f0^-2 - f1^-2| (MHz, seconds)


def search_delays(dms, fch1, foff, nchan, tsamp):
    """[ndm][nchan] delays in samples relative to the first (highest) channel: the formula of src/candidate.py:33 per
    channel, rounded half up (floor(x + 0.5): the delays are >= 0, where it equals analysis/loc_step0.py:62's
    np.round except on exact halves)."""
    f = fch1 + foff * np.arange(nchan)
    return np.stack([np.floor(DM_CONST * dm * (f ** -2 - f[0] ** -2) / tsamp + 0.5).astype(int) for dm in dms])


def dedisperse_series(codes, delays, zap):
    """codes [T][nchan] (SIGPROC time-major 8-bit), delays [ndm][nchan], zap [nchan] bool -> ([ndm][tout] exact integer
    sums, tout).  analysis/loc_step0.py:44-66 rolls channel c by -delay[c] and the caller sums over channels; this is
    that sum over the samples no roll wraps around (tout = T - the largest delay of a kept channel): a slice per
    channel instead of an in-place roll, the same numbers (tests/test_oracle_golden.py ties the two together)."""
    T, nchan = codes.shape
    ndm = delays.shape[0]
    maxd = max(int(delays[i][~zap].max()) for i in range(ndm))
    tout = T - maxd
    series = np.zeros((ndm, tout), np.uint32)
    x = codes.astype(np.uint32)
    for i in range(ndm):
        for c in range(nchan):
            if not zap[c]:
                series[i] += x[delays[i, c]:delays[i, c] + tout, c]
    return series, tout


def boxcar_best(x, mean, rms, nbox):
    """best S/N over the boxcar widths 2^0 .. 2^(nbox-1) starting at every sample, from a given level and rms:
    (sum of w samples - w mean) / (rms sqrt(w)) -- the running-mean form of analysis/loc_step0.py:143-146,
    (mean_w - level) / sigma * w^0.5, written with sums.  -> (best S/N per start sample, log2 of its width)"""
    x = np.asarray(x, np.float64)
    tout = x.size
    best = np.full(tout, -1e30)
    bw = np.zeros(tout, int)
    c = np.concatenate([[0], np.cumsum(x)])
    for k in range(nbox):
        w = 1 << k
        sn = np.full(tout, -1e30)
        sn[:tout - w + 1] = (c[w:] - c[:-w] - w * float(mean)) / (float(rms) * np.sqrt(w))
        upd = sn > best
        best[upd], bw[upd] = sn[upd], k
    return best, bw


def pulse_sn(ts, i0, i1, start, w):
    """The reference's S/N definition (analysis/loc_step0.py:120-147) for ONE given boxcar: level = median and sigma =
    Qn (analysis/utils.py:187-195) of the first and last quarter of ts[i0:i1], S/N = (mean of the w samples from
    `start` - level) / sigma * sqrt(w).  optimize_pulse itself only tries ODD widths, because tophat_smooth forces them
    (analysis/utils.py:84-86); this evaluates the same estimator at the width and position a search reports."""
    ts = np.asarray(ts, np.float64)
    nsamp = i1 - i0
    s = np.append(ts[i0:i0 + int(nsamp * 0.25)], ts[i0 + int(nsamp * 0.75):i1])
    return (ts[start:start + w].mean() - np.median(s)) / qn(s) * w ** 0.5


def candidate_columns(line):
    """the nine columns of a heimdall candidate line as src/candidate.py:8-18 reads them"""
    t = line.split()
    return dict(sn=float(t[0]), peak_idx=int(t[1]), peak_time=float(t[2]), tfilt=int(t[3]), dmi=int(t[4]), dm=float(t[5]),
                ngiant=int(t[6]), i0=int(t[7]), i1=int(t[8]), ncol=len(t))
