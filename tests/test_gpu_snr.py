"""Injected-pulse tests on the GPU.

* inject_frb parity: the in-band self test of the reference (src/process_baseband.cu:1231-1251,
  src/pb_kernels.cu:338-391) is bit-exact vs the oracle on small segments;
* BASELINE config 3 at full size: 8 antennas batched on one GPU, the config's 10 s of synthetic noise with a
  DM = 500 pc cm^-3, 2 ms, x1.05 pulse; the dedispersed S/N (estimator of
  analysis/loc_step0.py:optimize_pulse, pinned by tests/golden) is in the range the reference
  quotes for a single antenna ("about 25-30", src/process_baseband.cu:1239) and the fp32 coadd of 8
  antennas gains sqrt(8);
* the acceptance check of BASELINE config 3 itself: on identical seeded bytes, the S/N of the injected
  DM-500 pulse recovered from the HIP path's output is within 1 % of the S/N recovered from the oracle's
  (it is identical with the in-library FFT, which is bit-exact; the hipFFT back end is the case where
  the 1 % means something)."""
import numpy as np
import pytest

from helpers import NCHAN, compact_ave, libpb, make_input, oracle_run

pytestmark = pytest.mark.gpu


def test_inject_frb_bit_exact_vs_oracle(oracle):
    lp = libpb()
    R, nseg = 16, 4
    data = make_input(41, R, nseg, rfi=True, dropped=False)
    delays = oracle.set_frb_delays(80.0, R)
    res, _, _ = oracle_run(oracle, data, R, frb_delays=delays, inject_now=1)
    plain, _, _ = oracle_run(oracle, data, R)
    assert any(not np.array_equal(a.codes_raw, b.codes_raw) for a, b in zip(res, plain))   # it did something
    with lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=nseg, inject_frb=True, keep_ave=True) as h:
        for s in range(nseg):
            h.submit_planar(0, s, data[s, 0], data[s, 1])
        h.process(nseg, inject_now=1)
        out = h.fetch(0, 0, nseg, ave=True)
    assert np.array_equal(out["raw"], np.concatenate([r.codes_raw for r in res]))
    assert np.array_equal(out["kur"], np.concatenate([r.codes_kur for r in res]))
    ref = np.concatenate([compact_ave(r.ave_kur, R, 1) for r in res])
    assert np.array_equal(out["ave_kur"].view(np.uint32), ref.view(np.uint32))
    # hipFFT back end: same injection on complex planes, codes within one step
    with lp.PbHandle(nbit=8, rows_per_seg=R, max_seg=nseg, inject_frb=True, fft_backend=lp.FFT_HIPFFT) as h:
        for s in range(nseg):
            h.submit_planar(0, s, data[s, 0], data[s, 1])
        h.process(nseg, inject_now=1)
        o2 = h.fetch(0, 0, nseg)
    d = np.abs(o2["kur"].astype(int) - out["kur"].astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


def _snr(oracle, plane4096, t_pulse):
    """plane4096: [4096][T] fp32 (channel 0 = FFT bin 2155).  Dedisperse at DM 500 to 384 MHz with
    the reference's roll-per-channel routine and measure the boxcar S/N."""
    T = plane4096.shape[1]
    full = np.zeros((1, NCHAN, T), np.float32)
    full[0, 2154:2154 + 4096] = plane4096            # get_vlite_chan_freqs(6251)[i] = 384 - 64 (i+1)/6251
    tsamp = 12500 * 8 / 128e6
    oracle.dedisperse(full, 500.0, tsamp, ref_freq=384.0)
    mask = oracle.chan_mask()
    ts = (full[0] * mask[:, None]).sum(axis=0).astype(np.float64)
    i0, i1 = t_pulse - 128, t_pulse + 128
    widths, sns, locs = oracle.optimize_pulse(ts, i0, i1)
    k = int(np.argmax(sns))
    return float(sns[k]), int(widths[k]), int(locs[k]) + i0


def test_config3_dm500_pulse_8_antennas(oracle):
    import torch
    lp = libpb()
    A, S, NSEC, R = 8, 10, 10, 1024          # BASELINE configs[2]: 8 antennas, 10 s
    n = R * 12500
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    planes = np.zeros((A, 4096, NSEC * 1280), np.float32)
    coadd = np.zeros((4096, NSEC * 1280), np.float32)
    with lp.PbHandle(nant=A, nbit=8, rows_per_seg=R, max_seg=S, inject_frb=True, keep_ave=True) as h:
        h.set_frb_params(dm=500.0, width_rows=-1.0, amp=1.05)
        d_sum = torch.zeros(S * h.ave_per_seg, dtype=torch.float32, device=dev)
        for sec in range(NSEC):
            for a in range(A):
                g.manual_seed(1000 * (42 + a) + sec)        # independent noise per antenna (seed 42+ant)
                xs = [(torch.randn(2 * n, device=dev, generator=g) * 16.9 + 128.5).clamp_(0, 255).to(torch.uint8)
                      for s in range(S)]
                torch.cuda.synchronize()          # torch's stream and the library's are different streams
                for s in range(S):
                    h.submit_planar_dev(a, s, xs[s].data_ptr(), xs[s].data_ptr() + n, n)
                h.sync()                          # copies done before torch may recycle xs
            inject_now = 0 if sec < 1 else 1 + S * (sec - 1)      # the pulse enters the band at t = 1 s
            h.process(S, inject_now)
            for a in range(A):
                o = h.fetch(a, 0, S, raw=False, kur=False, ave=True)
                planes[a, :, sec * 1280:(sec + 1) * 1280] = o["ave_kur"].reshape(S * 128, 4096).T
            h.coadd_local(S, d_sum.data_ptr())
            h.sync()
            coadd[:, sec * 1280:(sec + 1) * 1280] = (d_sum.cpu().numpy() / np.sqrt(A)).reshape(S * 128, 4096).T
    t_pulse = 1280
    single = [_snr(oracle, planes[a], t_pulse) for a in range(A)]
    sn1 = np.array([s[0] for s in single])
    snc, wc, locc = _snr(oracle, coadd, t_pulse)
    print("single-antenna S/N", np.round(sn1, 1), "coadd", round(snc, 1), "width", wc, "loc", locc)
    assert abs(locc - t_pulse) <= 4 and wc <= 7               # 2 ms = 2.56 samples, at the injected time
    assert 15 < sn1.mean() < 45                               # reference: "about 25-30" for one antenna
    assert 0.8 * np.sqrt(A) < snc / sn1.mean() < 1.2 * np.sqrt(A)


def test_config3_snr_within_one_percent_of_oracle(oracle):
    """BASELINE config 3 acceptance: one antenna, 7 s of seeded genbase-style noise (NumPy PCG64, so that the
    oracle consumes the very same bytes), R = 1024, the reference's self-test pulse (DM 500 pc cm^-3
    referenced to 384 MHz, 2 ms, amplitude x1.05: src/pb_kernels.cu:338-391, src/process_baseband.cu:1238-1239)
    entering the band at t = 1 s; DM 500 sweeps 4.4 s across 361.94 -> 320 MHz.  Excised stream (RFI mode 1,
    what the search consumes).  S/N by the reference's estimator (analysis/loc_step0.py:120-147 restated in
    oracle.optimize_pulse, pinned by tests/golden) from the fp32 pre-quantisation planes AND from the 8-bit
    codes of (a) the oracle, (b) the HIP path with the in-library FFT, (c) the HIP path with hipFFT.

    Why ~36-40 and not the "about 25-30" of the reference's comment: x1.05 in voltage is +10.25 % in power;
    one output sample (2 pols x 8 rows, unit variance) moves by 0.1025 * 16 / sqrt(16) = 0.41, summed over
    the 3850 unmasked channels that is 25.4 per 781-us sample -- the comment's figure -- and the estimator's
    3-sample boxcar over the 2.56-sample pulse adds the rest."""
    lp = libpb()
    S, NSEC, R = 10, 7, 1024
    n = R * 12500
    T = NSEC * S * 128
    delays = oracle.set_frb_delays(500.0, R)
    bp_raw = np.zeros(2 * NCHAN, np.float32)
    bp_kur = np.zeros(2 * NCHAN, np.float32)
    planes = {k: np.zeros((4096, T), np.float32) for k in ("oracle", "lds", "hipfft")}
    codes = {k: np.zeros((4096, T), np.float32) for k in ("oracle", "lds", "hipfft")}
    rng = np.random.default_rng(20261004)
    hs = {"lds": lp.PbHandle(nbit=8, rfi_mode=1, rows_per_seg=R, max_seg=S, inject_frb=True, keep_ave=True),
          "hipfft": lp.PbHandle(nbit=8, rfi_mode=1, rows_per_seg=R, max_seg=S, inject_frb=True, keep_ave=True,
                                fft_backend=lp.FFT_HIPFFT)}
    try:
        for h in hs.values():
            h.set_frb_params(dm=500.0, width_rows=-1.0, amp=1.05)
        for sec in range(NSEC):
            data = np.empty((S, 2, n), np.uint8)
            for s in range(S):
                data[s] = np.clip(rng.standard_normal((2, n), dtype=np.float32) * np.float32(16.9) + np.float32(128.5),
                                  0, 255).astype(np.uint8)
            inject_now = 0 if sec < 1 else 1 + S * (sec - 1)          # the pulse enters the band at t = 1 s
            for k, h in hs.items():
                for s in range(S):
                    h.submit_planar(0, s, data[s, 0], data[s, 1])
                h.process(S, inject_now)
                o = h.fetch(0, 0, S, raw=False, kur=True, ave=True)
                planes[k][:, sec * S * 128:(sec + 1) * S * 128] = o["ave_kur"].reshape(S * 128, 4096).T
                codes[k][:, sec * S * 128:(sec + 1) * S * 128] = o["kur"].reshape(S * 128, 4096).T
            inj = inject_now
            for s in range(S):
                r = oracle.segment(data[s], R, bp_raw, bp_kur, rfi_mode=1, npol=1, nbit=8, frb_delays=delays,
                                   inject_now=inj, want_stats=False)
                if inj > 0:
                    inj += 1
                t0 = (sec * S + s) * 128
                planes["oracle"][:, t0:t0 + 128] = compact_ave(r.ave_kur, R, 1).reshape(128, 4096).T
                codes["oracle"][:, t0:t0 + 128] = r.codes_kur.reshape(128, 4096).T
    finally:
        for h in hs.values():
            h.close()
    t_pulse = S * 128
    sn = {k: _snr(oracle, planes[k], t_pulse) for k in planes}
    snc = {k: _snr(oracle, codes[k] - 127.5, t_pulse) for k in codes}
    print("S/N from fp32 planes:", {k: round(v[0], 3) for k, v in sn.items()},
          " from 8-bit codes:", {k: round(v[0], 3) for k, v in snc.items()}, " width / location:", sn["oracle"][1:])
    assert abs(sn["oracle"][2] - t_pulse) <= 4 and sn["oracle"][1] <= 7
    assert sn["oracle"][0] > 20                                        # the pulse is there at all
    # in-library FFT: bit-exact planes and codes, hence the same S/N to the last bit
    assert np.array_equal(planes["lds"].view(np.uint32), planes["oracle"].view(np.uint32))
    assert np.array_equal(codes["lds"], codes["oracle"])
    for k in ("lds", "hipfft"):
        assert abs(sn[k][0] - sn["oracle"][0]) / sn["oracle"][0] < 0.01, (k, sn[k], sn["oracle"])
        assert abs(snc[k][0] - snc["oracle"][0]) / snc["oracle"][0] < 0.01, (k, snc[k], snc["oracle"])
