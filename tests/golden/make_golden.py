#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's Python analysis code.

Run in the build container only (needs /root/reference, which never travels to the
GPU box):   python tests/golden/make_golden.py

What is pinned:
  p1   analysis/baseband.py:960   filterbank(x, nfft=12500)
  p2   analysis/baseband.py:1207  polyphase_filterbank(x, nchan, nwindow=4)
  h8   analysis/baseband.py:17    VDIFHeader field decode
  p3   analysis/baseband.py:221   BasebandFragment.get_data(thread=-1)
  s1   analysis/loc_step0.py:44,111,120 + analysis/utils.py:4,74,187
       dedisperse / chan_mask / optimize_pulse / tophat_smooth / qn

Only inputs (regenerated from tests/golden/synth.py) and the reference's OUTPUTS are
stored; no reference source text is copied.

Import shims (ordinary Python errors, not refusals): the reference imports `pyfftw`
(absent from the image) and `scipy.signal.hamming/hanning` (moved to
scipy.signal.windows in SciPy >= 1.13).  An empty module named pyfftw and two aliases
make `import baseband` succeed; neither is used by the functions called here except
hamming, which is SciPy's own.
"""
import ast
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = "/root/reference/analysis"


def import_reference():
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, "hamming"):
        scipy.signal.hamming = scipy.signal.windows.hamming
    if not hasattr(scipy.signal, "hanning"):
        scipy.signal.hanning = scipy.signal.windows.hann
    sys.modules.setdefault("pyfftw", types.ModuleType("pyfftw"))
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import baseband
    import utils
    # loc_step0.py is a script over private data paths; take only its function defs
    src = open(os.path.join(REF, "loc_step0.py")).read()
    tree = ast.parse(src)
    wanted = {"get_vlite_chan_freqs", "dedisperse", "chan_mask", "optimize_pulse"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"np": np, "utils": utils, "inplace_roll": utils.inplace_roll}
    exec(compile(mod, "loc_step0_functions", "exec"), ns)
    return baseband, utils, ns


def s1_plane(seed, nchan, ntime, dm, t0, width, amp, tsamp, fref):
    """[1][nchan][ntime] float32 noise plane with a dispersed top-hat pulse; channel c is
    at get_vlite_chan_freqs(nchan)[c] MHz; delay convention of loc_step0.dedisperse."""
    plane = synth.gauss(seed, nchan * ntime).reshape(1, nchan, ntime).astype(np.float32)
    freqs = ((np.arange(nchan)) * 64. / nchan + 320)[::-1]
    delays = np.round(dm * 4.15e-3 * ((freqs * 1e-3) ** -2 - (fref * 1e-3) ** -2) / tsamp).astype(int)
    for c in range(nchan):
        i0 = t0 + delays[c]
        plane[0, c, i0:i0 + width] += amp
    return plane


def main():
    baseband, utils, loc0 = import_reference()
    out = {}

    # ---- P1: filterbank on GPU-scaled voltages (u/128-1), 16 spectra
    u = synth.baseband_u8(11, 16 * 12500)
    x = np.where(u == 0, np.float32(0), u.astype(np.float32) / np.float32(128) - np.float32(1)).astype(np.float32)
    fb = baseband.filterbank(x, nfft=12500)
    assert fb.shape == (16, 6251) and fb.dtype == np.float64
    out["p1_seed"] = 11
    out["p1_bins"] = np.arange(0, 6251, 25)
    out["p1_slice"] = fb[:, ::25].copy()
    out["p1_band"] = fb[:4, 2155:2155 + 256].copy()
    out["p1_rowsum"] = fb.sum(axis=1)

    # ---- P2: 4-tap PFB, full size (nchan=6250) and a small one (nchan=64)
    xs = synth.gauss(12, 8 * 50000).astype(np.float32)
    pf = baseband.polyphase_filterbank(xs, nchan=6250, nwindow=4)
    assert pf.shape == (28, 6251) and pf.dtype == np.complex64
    out["p2_seed"] = 12
    out["p2_slice"] = pf[:, ::25].copy()
    out["p2_abs2_rowsum"] = (np.abs(pf.astype(np.complex128)) ** 2).sum(axis=1)
    xs2 = synth.gauss(13, 4096).astype(np.float32)
    pf2 = baseband.polyphase_filterbank(xs2, nchan=64, nwindow=4)
    out["p2s_seed"] = 13
    out["p2s_full"] = pf2.copy()

    # ---- H8: VDIF header decode
    cases = [(0, 0, 0, 0, 0), (12345678, 33, 25599, 7, 1), (987654321 & 0x3FFFFFFF, 41, 1234, 99, 0),
             (31, 63, 16777215, 65535, 1023)]
    hdr_words, hdr_fields = [], []
    for (sec, ep, fr, st, th) in cases:
        w = synth.vdif_header_words(sec, ep, fr, st, th)
        h = baseband.VDIFHeader(w)
        hdr_words.append(w)
        hdr_fields.append([h.second, h.epoch, h.frame, h.frame_length, h.frame_nsamp, h.station,
                           h.threadid, h.thread])
    out["h8_words"] = np.array(hdr_words, dtype=np.uint32)
    out["h8_fields"] = np.array(hdr_fields, dtype=np.int64)
    h = baseband.VDIFHeader(synth.vdif_header_words(3600, 33, 0, 5, 0))
    out["h8_unix_ep33_sec3600"] = h.get_unix_timestamp()
    out["h8_utc_ep33_sec3600"] = np.array(h.get_utc_str())

    # ---- P3: get_data(thread=-1) on an 8-frame file (thread 0 first, alternating)
    nfr = 8
    payload = synth.baseband_u8(14, nfr * 5000).reshape(nfr, 5000)
    raw = np.zeros((nfr, 5032), dtype=np.uint8)
    for i in range(nfr):
        w = synth.vdif_header_words(100, 33, i // 2, 5, i % 2)
        raw[i, :32] = w.view(np.uint8)
        raw[i, 32:] = payload[i]
    with tempfile.NamedTemporaryFile(suffix=".vdif", delete=False) as f:
        f.write(raw.tobytes())
        fname = f.name
    frag = baseband.BasebandFragment(fname)
    d = frag.get_data(thread=-1, nsamp=4 * 5000, offs=[0, 0])
    os.unlink(fname)
    assert d.shape == (2, 20000) and d.dtype == np.float32
    out["p3_seed"] = 14
    out["p3_plus_127p5"] = (d + 127.5).astype(np.uint8)     # values are u8 - 127.5
    assert np.array_equal(out["p3_plus_127p5"].astype(np.float32) - np.float32(127.5), d)

    # ---- S1: dedispersion + boxcar S/N
    nchan, ntime, tsamp, fref = 512, 1536, 12500 * 8 / 128e6, 361.94144882
    dm, t0, width, amp = 60.0, 500, 3, 0.35
    plane = s1_plane(15, nchan, ntime, dm, t0, width, amp, tsamp, fref)
    work = plane.copy()
    loc0["dedisperse"](work, dm, tsamp, ref_freq=fref)
    ts = work[0].sum(axis=0).astype(np.float64)
    i0, i1 = t0 - 128, t0 + 128
    widths, sns, locs = loc0["optimize_pulse"](ts, i0, i1)
    out["s1_params"] = np.array([15, nchan, ntime, dm, t0, width, amp, tsamp, fref, i0, i1])
    out["s1_ts"] = ts
    out["s1_widths"] = widths
    out["s1_sns"] = sns
    out["s1_locs"] = locs
    out["s1_qn"] = utils.qn(ts[:200])
    out["s1_tophat5"] = utils.tophat_smooth(ts[:64].copy(), 5)
    out["s1_chan_mask_idx"] = np.nonzero(loc0["chan_mask"]() == 0)[0]
    a = np.arange(20.).reshape(2, 10)
    out["s1_roll3"] = utils.inplace_roll(a.copy(), 3)
    out["s1_rollm4"] = utils.inplace_roll(a.copy(), -4)

    path = os.path.join(HERE, "reference_python.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k, v in out.items():
        print("  %-22s %s" % (k, getattr(v, "shape", v)))


if __name__ == "__main__":
    main()
