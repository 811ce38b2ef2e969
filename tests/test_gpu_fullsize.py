"""Full-size segments (1024 rows x 12500 samples x 2 pols = 25.6 MB each, the reference's
FFTS_PER_SEG).  The C oracle takes about a second per such segment, so two of them are compared bit for
bit (test_fullsize_two_segments_bit_exact_vs_oracle); at batch sizes beyond that, parity is checked
through size-independent properties of the path:
  * batching invariance: S segments in one pb_process call == S calls of one segment (the bandpass is
    the only state carried, in order) == the same data in a different antenna slot of a batch;
  * the two FFT back ends (in-library LDS FFT vs hipFFT) agree to one quantiser step on < 0.2 % of the
    samples, i.e. the hand-written 12500-point FFT is right at full batch size;
  * statistics of the output on genbase-style noise are what the reference's quantiser assumes
    (8-bit codes centred on 127.5 with sigma 1/0.02957 = 33.8 steps)."""
import numpy as np
import pytest

from helpers import libpb

pytestmark = pytest.mark.gpu
R, NSEG = 1024, 4


def _noise(seed, nseg):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    n = R * 12500
    x = (torch.randn(nseg * 2 * n, device="cuda", generator=g) * 16.9 + 128.5).clamp_(0, 255).to(torch.uint8)
    torch.cuda.synchronize()
    return x, n


def _submit(h, ant, x, n, nseg, seg0=0):
    for s in range(nseg):
        base = x.data_ptr() + (seg0 + s) * 2 * n
        h.submit_planar_dev(ant, s, base, base + n, n)
    h.sync()


def test_fullsize_two_segments_bit_exact_vs_oracle(oracle):
    """Two consecutive full-size segments (bandpass carried) with RFI bursts, a row of weight 0, a
    strongly flagged row and a dropped frame: both streams' codes and the bandpass state equal the
    oracle's, at 8 and at 2 bits."""
    from helpers import NCHAN, make_input, oracle_run
    lp = libpb()
    d = make_input(41, R, 2)
    for nbit in (8, 2):
        res, bp_raw, bp_kur = oracle_run(oracle, d, R, rfi_mode=2, npol=1, nbit=nbit)
        with lp.PbHandle(nant=1, nbit=nbit, rows_per_seg=R, max_seg=2) as h:
            for s in range(2):
                h.submit_planar(0, s, d[s, 0], d[s, 1])
            h.process(2)
            out = h.fetch(0, 0, 2)
            gr, gk = h.get_bandpass(0)
        assert np.array_equal(out["raw"], np.concatenate([r.codes_raw for r in res]))
        assert np.array_equal(out["kur"], np.concatenate([r.codes_kur for r in res]))
        assert np.array_equal(gr.view(np.uint32), bp_raw.reshape(2, NCHAN)[:, 2155:].view(np.uint32))
        assert np.array_equal(gk.view(np.uint32), bp_kur.reshape(2, NCHAN)[:, 2155:].view(np.uint32))


def test_fullsize_batching_invariance():
    lp = libpb()
    x, n = _noise(7, NSEG)
    with lp.PbHandle(nant=2, nbit=8, rows_per_seg=R, max_seg=NSEG) as h:
        _submit(h, 0, x, n, NSEG)
        _submit(h, 1, x, n, NSEG)
        h.process(NSEG)
        a0 = h.fetch(0, 0, NSEG)
        a1 = h.fetch(1, 0, NSEG)
    assert np.array_equal(a0["raw"], a1["raw"]) and np.array_equal(a0["kur"], a1["kur"])
    raw, kur = [], []
    with lp.PbHandle(nant=1, nbit=8, rows_per_seg=R, max_seg=1) as h:
        for s in range(NSEG):
            _submit(h, 0, x, n, 1, seg0=s)
            h.process(1)
            o = h.fetch(0, 0, 1)
            raw.append(o["raw"])
            kur.append(o["kur"])
    assert np.array_equal(np.concatenate(raw), a0["raw"]) and np.array_equal(np.concatenate(kur), a0["kur"])
    # the quantiser's design point: unit-variance normalised power -> mean 127.5, sigma 33.8 codes
    c = a0["raw"][a0["trim"] if "trim" in a0 else 524288:].astype(np.float64)      # skip the first segment (bandpass settling)
    assert abs(c.mean() - 127.0) < 1.5 and abs(c.std() - 33.8) < 2.0
    assert a0["raw"].size == NSEG * 128 * 4096


def test_fullsize_lds_fft_vs_hipfft():
    lp = libpb()
    x, n = _noise(8, 2)
    out = {}
    for name, be in (("lds", lp.FFT_LDS), ("hipfft", lp.FFT_HIPFFT)):
        with lp.PbHandle(nant=1, nbit=8, rows_per_seg=R, max_seg=2, fft_backend=be) as h:
            _submit(h, 0, x, n, 2)
            h.process(2)
            out[name] = h.fetch(0, 0, 2)
    for k in ("raw", "kur"):
        d = np.abs(out["lds"][k].astype(int) - out["hipfft"][k].astype(int))
        assert d.max() <= 1, k
        assert (d != 0).mean() < 2e-3, (k, (d != 0).mean())
