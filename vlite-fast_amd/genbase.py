#!/usr/bin/env python3
"""genbase-style synthetic baseband: the test-side producer for the hot path.

Follows the recipe of /root/reference/src/genbase.cu (not its CUDA code):
  Gaussian noise, sigma 1                      :313,377  (cuRAND there; NumPy Philox here, so
                                                          streams are NOT bit-reproducible vs cuRAND)
  periodic pulse: x *= 1+ampl for phase < 0.03 :554-585  (every skip_period-th period)
  coherent dispersion by FFT chirp             :525-552,587-598
      arg = 2 pi DM/2.41e-10 f^2 / (f0^2 (f0+f)), f0 = 320 MHz, f = 64 i/n MHz
      with the band-pass taper (1-exp(-(f/.05)^2) - exp(-((1-f)/.1)^2)) (1+0.2 f)
      overlap-save with n_dm_samp samples of overlap :174-196,366-400
  swap_sideband: negate odd samples            :651-661
  optional RFI: uniform +-2.5 for 10 % of every 11.3 us :671-687
  digitise: trunc(x/0.02957/2 + 128.5) clamped :689-708
  VDIF framing: 5000 samples per frame, threads alternating, 25600 frames/s :445-486
  ring header                                  :330-353
Flags: -t -n -p -k -a -s -r -d -f as the reference's genbase; the output is a dump file
(`--out`, the reference's `-e`) in the FileRing layout of dada.py (header + frame stream).
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

_pkg = __package__ or "vlite-fast_amd"
if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vdif = importlib.import_module(_pkg + ".vdif")
dada = importlib.import_module(_pkg + ".dada")

VLITE_RATE = 128000000


def dm_samples(dm, rate=VLITE_RATE):
    """(n_lo, n_hi) overlap sample counts, src/genbase.cu:174-196 (including its swap)."""
    freq, freq_hi, freq_lo = 352., 384., 320.
    tsamp = 1.0 / rate
    t_lo = dm / 2.41e-10 * (1. / (freq_lo * freq_lo) - 1. / (freq * freq))   # us
    t_hi = dm / 2.41e-10 * (1. / (freq * freq) - 1. / (freq_hi * freq_hi))
    n_lo = int(int(t_lo) * 1e-6 / tsamp)      # `(unsigned long) t_dm_lo*1e-6/tsamp`: cast binds first
    n_hi = int(int(t_hi) * 1e-6 / tsamp)
    n_lo += n_lo & 1
    n_hi += n_hi & 1
    return n_hi, n_lo                          # swapped, :191-195


def dm_kernel(dm, n):
    """Chirp x band-pass taper, complex64, length n = buflen/2+1 (the 1/(2(n-1)) of the
    reference is cuFFT's missing C2R normalisation; numpy.fft.irfft normalises itself)."""
    i = np.arange(n, dtype=np.float64)
    freq = 64. * i / float(n)
    freq0 = 320.
    arg = (2 * np.pi * float(np.float32(dm)) / 2.41e-10) * freq * freq / (freq0 * freq0 * (freq0 + freq))
    ker = (np.cos(arg) + 1j * np.sin(arg))
    f = freq / 64.
    scale = 1 - np.exp(-(f * f) / (0.05 * 0.05))
    scale -= np.exp(-((1 - f) * (1 - f)) / (0.10 * 0.10))
    scale *= (1 + 0.20 * f)
    return (ker * scale).astype(np.complex64)


def set_profile(x, current_sample, period, skip_period, ampl):
    n = x.size
    sample = current_sample + np.arange(n, dtype=np.int64)
    phasei = sample // period
    phasef = ((sample - phasei * period).astype(np.float32) / np.float32(period))
    on = (phasef < np.float32(0.03)) & ((phasei % skip_period) == 0)
    x[on] *= np.float32(ampl)


def add_rfi(x, rng, current_sample, tsamp_us):
    i = np.arange(x.size, dtype=np.int64) + int(current_sample)
    phase = np.fmod((i * (tsamp_us / 11.3)).astype(np.float32), np.float32(1))
    on = phase < np.float32(0.1)
    x[on] += np.float32(5.) * (rng.random(int(on.sum()), dtype=np.float32) - np.float32(0.5))


def digitize(x):
    tmp = (x.astype(np.float32) / np.float32(0.02957) / np.float32(2) + np.float32(128.5))
    return np.where(tmp <= 0, 0, np.where(tmp >= 255, 255, tmp)).astype(np.uint8)   # cast truncates


def generate(tobs=5.0, dm=30.0, period=0.5, ampl=0.05, poln_ratio=1.0, seed=42, rfi=False, skip_period=1,
             buflen=VLITE_RATE // 4, rate=VLITE_RATE):
    """Yield (pol0, pol1) uint8 chunks of new_samps = buflen - n_dm_samp samples each until
    tobs seconds have been generated (the reference's outer loop, :366-500)."""
    n_lo, n_hi = dm_samples(dm, rate)
    n_dm = n_lo + n_hi
    period_s = int(period * rate)
    if buflen < 2 * (n_dm + period_s) and dm > 0 and buflen == VLITE_RATE // 4:
        raise ValueError("Buffer not long enough to perform dedispersion!")    # :207-211
    rng = np.random.Generator(np.random.Philox(seed))
    rng_rfi = np.random.Generator(np.random.Philox(1233456))
    ampls = (1 + ampl, 1 + ampl * poln_ratio)
    ker = dm_kernel(dm, buflen // 2 + 1)
    new = buflen - n_dm
    ovl = []
    for p in range(2):
        o = rng.standard_normal(n_dm, dtype=np.float32)
        set_profile(o, 0, period_s, skip_period, ampls[p])
        ovl.append(o)
    current = n_dm
    end = int(tobs * rate)
    tsamp_us = 1e6 / rate
    while current < end:
        outs = []
        for p in range(2):
            fdat = np.empty(buflen, np.float32)
            fdat[:n_dm] = ovl[p]
            fdat[n_dm:] = rng.standard_normal(new, dtype=np.float32)
            set_profile(fdat[n_dm:], current, period_s, skip_period, ampls[p])
            ovl[p] = fdat[buflen - n_dm:].copy()
            F = np.fft.rfft(fdat).astype(np.complex64) * ker
            x = np.fft.irfft(F, buflen).astype(np.float32)
            x[1::2] = -x[1::2]
            if rfi:
                add_rfi(x, rng_rfi, current - n_dm - n_lo, tsamp_us)
            outs.append(digitize(x[n_lo:n_lo + new]))
        current += new
        yield outs[0], outs[1]


def ring_header(t_unix, name="B0833-45", station=0):
    """What genbase puts in the ring header (:330-353) plus STATIONID for downstream naming."""
    h = {}
    vdif.ascii_header_set(h, "STATIONID", "%d" % station)
    vdif.ascii_header_set(h, "NAME", name)
    vdif.ascii_header_set(h, "NCHAN", "1")
    vdif.ascii_header_set(h, "BANDWIDTH", "%f" % 64.0)
    vdif.ascii_header_set(h, "CFREQ", "%f" % 352.0)
    vdif.ascii_header_set(h, "NPOL", "2")
    vdif.ascii_header_set(h, "NBIT", "8")
    vdif.ascii_header_set(h, "RA", "%f" % 0.87180)
    vdif.ascii_header_set(h, "DEC", "%f" % 0.72452)
    vdif.ascii_header_set(h, "UTC_START", time.strftime(vdif.DADA_TIMESTR, time.gmtime(t_unix)))
    return h


def write_observation(path, chunks, t_unix, station=0, name="B0833-45", frames_per_sec=vdif.FRAMESPERSEC):
    """Frame a stream of (pol0, pol1) chunks into a dump file.  Returns frames written per thread."""
    epoch, second = vdif.epoch_for_unix(t_unix)
    carry = [np.empty(0, np.uint8), np.empty(0, np.uint8)]
    nframes = 0
    with open(path, "wb") as f:
        f.write(vdif.ascii_header_format(ring_header(t_unix, name, station)))
        for p0, p1 in chunks:
            d = [np.concatenate([carry[0], p0]), np.concatenate([carry[1], p1])]
            nfr = d[0].size // vdif.VD_DAT
            if nfr:
                # frame numbers roll over every frames_per_sec (25600; tests use shorter seconds)
                for f0 in range(0, nfr, 4096):
                    f1 = min(nfr, f0 + 4096)
                    fnum = nframes + f0 + np.arange(f1 - f0)
                    blk = vdif.frame_block(d[0][f0 * vdif.VD_DAT:f1 * vdif.VD_DAT],
                                           d[1][f0 * vdif.VD_DAT:f1 * vdif.VD_DAT], 0, epoch, station)
                    w = blk.reshape(f1 - f0, 2, vdif.VD_FRM)[:, :, :8].view("<u4")
                    w[:, :, 0] = ((second + fnum // frames_per_sec) & 0x3FFFFFFF)[:, None].astype(np.uint32)
                    w[:, :, 1] = ((fnum % frames_per_sec).astype(np.uint32) | np.uint32((epoch & 0x3F) << 24))[:, None]
                    f.write(blk.tobytes())
            nframes += nfr
            carry = [d[0][nfr * vdif.VD_DAT:], d[1][nfr * vdif.VD_DAT:]]
    return nframes


def main(argv=None):
    ap = argparse.ArgumentParser(prog="genbase", add_help=False)
    ap.add_argument("-h", action="help")
    ap.add_argument("-t", dest="tobs", type=float, default=5)
    ap.add_argument("-n", dest="nobs", type=int, default=1)
    ap.add_argument("-p", dest="period", type=float, default=0.5)
    ap.add_argument("-k", dest="skip_period", type=int, default=1)
    ap.add_argument("-a", dest="ampl", type=float, default=0.05)
    ap.add_argument("-s", dest="poln_ratio", type=float, default=1.0)
    ap.add_argument("-r", dest="seed", type=int, default=42)
    ap.add_argument("-d", dest="dm", type=float, default=30)
    ap.add_argument("-f", dest="rfi", action="store_true")
    ap.add_argument("-e", dest="to_disk", action="store_true")   # always to disk here
    ap.add_argument("--out", default="baseband_sim.uw")
    ap.add_argument("--unix-time", type=int, default=None,
                    help="extension: start of the observation (default: now, as the reference stamps it); several "
                         "antennas' dumps that are to be coadded need the same one")
    a = ap.parse_args(argv)
    for i in range(a.nobs):
        path = a.out if a.nobs == 1 else "%s.%d" % (a.out, i)
        n = write_observation(path, generate(a.tobs, a.dm, a.period, a.ampl, a.poln_ratio, a.seed + i, a.rfi,
                                             a.skip_period), int(time.time()) if a.unix_time is None else a.unix_time)
        print("wrote %s: %d frames per thread" % (path, n))


if __name__ == "__main__":
    main()
