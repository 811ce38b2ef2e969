"""Dedispersion + boxcar search (BASELINE config 5) on the GPU: integer arithmetic bit-exact against
a NumPy restatement, S/N of an injected dispersed pulse consistent with the reference's own
estimator (analysis/loc_step0.py, pinned by tests/golden), candidate lines parse with the
reference's src/candidate.py column order."""
import importlib

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
search = importlib.import_module("vlite-fast_amd.search")


def _plane(seed, T, nchan, dm, t0, width, amp, fch1, foff, tsamp):
    """8-bit codes [T][nchan]: noise (mean 127.5, sigma 33.8 like sel_and_dig_8b of unit-variance
    data) + a dispersed top-hat pulse."""
    g = synth.gauss(seed, T * nchan).reshape(T, nchan)
    f = fch1 + foff * np.arange(nchan)
    delay = np.round(4.148808e3 * dm * (f ** -2 - f[0] ** -2) / tsamp).astype(int)
    for c in range(nchan):
        g[t0 + delay[c]:t0 + delay[c] + width, c] += amp
    return np.clip(np.floor(g / 0.02957 + 127.5), 0, 255).astype(np.uint8), delay


def _numpy_search(codes, nchan, delays, zap, nbox):
    T = codes.shape[0]
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


def test_dedisperse_and_boxcar_small():
    nchan, T, tsamp = 256, 2048, search.TSAMP
    fch1, foff = 361.94, -0.16
    codes, _ = _plane(71, T, nchan, dm=40.0, t0=700, width=4, amp=0.8, fch1=fch1, foff=foff, tsamp=tsamp)
    zap = ((0, 10), (250, 256))
    with search.Searcher(nchan=nchan, max_samples=T, fch1=fch1, foff=foff, tsamp=tsamp, dm_min=0.0, dm_max=80.0,
                         dm_step=4.0, boxcar_max=16, zap=zap) as s:
        r = s.run(codes, want_series=True)
        dms, nbox, maxd = s.dms, s.nbox, s.max_delay
    f = fch1 + foff * np.arange(nchan)
    delays = np.stack([np.floor(4.148808e3 * dm * (f ** -2 - f[0] ** -2) / tsamp + 0.5).astype(int) for dm in dms])
    zmask = np.zeros(nchan, bool)
    for lo, hi in zap:
        zmask[lo:hi] = True
    ref, tout = _numpy_search(codes, nchan, delays, zmask, nbox)
    assert tout == r["tout"] and maxd == int(delays[-1][~zmask].max())
    assert np.array_equal(r["series"], ref)                        # dedispersion: exact integers
    # boxcar S/N from the kernel's own (clipped) mean and rms
    i = 10                                                          # DM 40
    mean, rms = r["stats"][i]
    x = ref[i].astype(np.float64)
    best = np.full(tout, -1e30)
    bw = np.zeros(tout, int)
    for k in range(nbox):
        w = 1 << k
        c = np.concatenate([[0], np.cumsum(x)])
        sn = np.full(tout, -1e30)
        sn[:tout - w + 1] = (c[w:] - c[:-w] - w * float(mean)) / (float(rms) * np.sqrt(w))
        upd = sn > best
        best[upd], bw[upd] = sn[upd], k
    np.testing.assert_allclose(r["snr"][i], best, rtol=2e-5, atol=2e-4)
    mad = 1.4826 * np.median(np.abs(x - np.median(x)))              # robust sigma (the pulse inflates x.std())
    assert abs(float(mean) - np.median(x)) < 0.1 * mad and 0.9 < float(rms) / mad < 1.1
    # the pulse is found at the right DM, time and width
    k = np.unravel_index(np.argmax(r["snr"]), r["snr"].shape)
    assert abs(dms[k[0]] - 40.0) <= 4.0 and abs(k[1] - 700) <= 4 and (1 << r["width_log2"][k]) in (2, 4, 8)


def test_config5_full_band_candidates(oracle):
    """4096 channels, heimdall's production flags (DM 2-1000, boxcar <= 64, zapped edges), a 10-s block
    with a DM 300 pulse: one dominant candidate at the injected DM / time / width, S/N consistent with
    the reference's optimize_pulse estimator on the same dedispersed series."""
    nchan, T = 4096, 20480            # DM 1000 sweeps 11 322 samples across the band
    dm, t0, width, amp = 302.0, 3000, 4, 0.35       # on the DM grid 2, 12, 22, ...
    codes, delay = _plane(72, T, nchan, dm, t0, width, amp, search.FCH1, search.FOFF, search.TSAMP)
    with search.Searcher(max_samples=T, dm_step=10.0) as s:
        r = s.run(codes, want_series=True)
        cands = search.find_candidates(r["snr"], r["width_log2"], s.dms, s.tsamp, threshold=7.0)
        dms = s.dms
    assert len(cands) >= 1
    top = cands[0]
    assert abs(top["dm"] - dm) <= 10.0 and abs(top["peak_idx"] - t0) <= 4 and 1 <= top["tfilt"] <= 3
    assert top["snr"] > 3 * max([c["snr"] for c in cands[1:]] + [0.0]) or len(cands) == 1
    # S/N cross-check with the reference's estimator (median / Qn / top-hat) on the same series
    i = int(np.argmin(np.abs(dms - dm)))
    ts = r["series"][i].astype(np.float64)
    widths, sns, locs = oracle.optimize_pulse(ts, t0 - 128, t0 + 128, wmax=16)
    ref_sn = float(sns.max())
    assert 0.8 < top["snr"] / ref_sn < 1.25, (top["snr"], ref_sn)
    # candidate line: the reference's parser column order (src/candidate.py:8-18)
    toks = search.candidate_line(top).split()
    assert abs(float(toks[0]) - top["snr"]) < 1e-3 and int(toks[1]) == top["peak_idx"] and int(toks[4]) == top["dmi"]
    assert abs(float(toks[5]) - top["dm"]) < 1e-3 and int(toks[7]) <= top["peak_idx"] < int(toks[8]) and len(toks) == 9
