"""Dedispersion + boxcar search on a filterbank block, and heimdall-style candidate lines.

Stands where the external `heimdall_stream` stands in the reference's chain
(/root/reference/scripts/start_heimdall_single_antenna:21; candidate line columns
`S/N, peak_idx, peak_time, tfilt, dmi, dm, ngiant, i0, i1` as parsed by src/candidate.py:8-18).
The GPU work is libpb_hip.so's pb_search_* (csrc/pb_search.hip); this module only drives it and
groups the above-threshold samples into candidates on the host.  heimdall is third-party and not in
the image: its DM spacing, baseline removal and giant merging are not reproduced -- parity with its
candidate list is unpinned.
"""
import ctypes as C
import importlib

import numpy as np

_pkg = __package__ or "vlite-fast_amd"

# heimdall flags of the reference's production launch
HEIMDALL_DM = (2.0, 1000.0)
HEIMDALL_BOXCAR_MAX = 64
HEIMDALL_GULP = 30720
HEIMDALL_ZAP = ((0, 190), (3900, 4096))
FCH1 = 384 + (2155 - 0.5) * (-64. / 6251)     # MHz, src/process_baseband.cu:261
FOFF = -64. / 6251
TSAMP = 12500.0 / 128e6 * 8


def dedisp_dm_list(dm_start, dm_end, tsamp, fch1, foff, nchan, tol=1.25, pulse_width_us=40.0):
    """Trial DMs spaced so that the smearing of a pulse at one DM by the next step stays within `tol`:
    the recursion of the dedisp library's generate_dm_list that heimdall runs on (-dm_tol 1.25,
    -dm_pulse_width 40 are heimdall's defaults; dedisp/heimdall are third party, unpinned forks in the
    reference's src/INSTALL and absent here: restated from the published algorithm, Barsdell et al. 2012
    / Levin 2012 eq. 4.6; the candidate list it leads to stays parity-unpinned).
        a = 8.3 df / f^3 (us per DM unit per channel, f = band centre in GHz, df in MHz)
        dm' = (b2 dm + sqrt(-a2 b2 dm^2 + (a2 + b2) k)) / (a2 + b2),
        a2 = a^2, b2 = a2 nchan^2 / 16, k = (dt^2 + w^2)(tol^2 - 1) + tol^2 a2 dm^2"""
    dt = tsamp * 1e6
    f = (fch1 + (nchan / 2 - 0.5) * foff) * 1e-3
    tol2 = tol * tol
    a = 8.3 * abs(foff) / (f * f * f)
    a2 = a * a
    b2 = a2 * (nchan * nchan / 16.0)
    c = (dt * dt + pulse_width_us * pulse_width_us) * (tol2 - 1.0)
    dms = [float(dm_start)]
    while dms[-1] < dm_end:
        prev = dms[-1]
        k = c + tol2 * a2 * prev * prev
        dms.append((b2 * prev + np.sqrt(-a2 * b2 * prev * prev + (a2 + b2) * k)) / (a2 + b2))
    return np.asarray(dms, np.float32)


class Searcher(object):
    def __init__(self, device=0, nchan=4096, max_samples=HEIMDALL_GULP, fch1=FCH1, foff=FOFF, tsamp=TSAMP,
                 dm_min=HEIMDALL_DM[0], dm_max=HEIMDALL_DM[1], dm_step=2.0, boxcar_max=HEIMDALL_BOXCAR_MAX,
                 zap=HEIMDALL_ZAP, dm_list=None):
        """dm_list: explicit trial DMs (e.g. dedisp_dm_list(...)); default the linear grid dm_min..dm_max"""
        lp = importlib.import_module(_pkg + ".libpb")
        self._lp = lp
        self._L = lp.load()
        zr = (C.c_int * (2 * len(zap)))(*[v for pair in zap for v in pair])
        self._s = C.c_void_p()
        if dm_list is not None:
            dl = np.ascontiguousarray(dm_list, np.float32)
            rc = self._L.pb_search_create_list(device, nchan, max_samples, fch1, foff, tsamp,
                                               dl.ctypes.data_as(C.POINTER(C.c_float)), dl.size, boxcar_max, zr, len(zap),
                                               C.byref(self._s))
        else:
            rc = self._L.pb_search_create(device, nchan, max_samples, fch1, foff, tsamp, dm_min, dm_max, dm_step,
                                          boxcar_max, zr, len(zap), C.byref(self._s))
        if rc != 0:
            raise lp.PbError("pb_search_create failed (%d): %s" % (rc, self._L.pb_search_last_error(None).decode()))
        ndm, nbox, md = C.c_int(), C.c_int(), C.c_int()
        self._L.pb_search_info(self._s, C.byref(ndm), C.byref(nbox), C.byref(md))
        self.ndm, self.nbox, self.max_delay = ndm.value, nbox.value, md.value
        self.nchan, self.tsamp = nchan, tsamp
        self.dms = dm_min + dm_step * np.arange(self.ndm) if dm_list is None else np.asarray(dm_list, np.float64)

    def close(self):
        if getattr(self, "_s", None):
            self._L.pb_search_destroy(self._s)
            self._s = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_baseline(self, window_samples):
        """running baseline over this many samples (heimdall: 2 s = 2560); 0 = one clipped mean per series"""
        rc = self._L.pb_search_set_baseline(self._s, int(window_samples))
        if rc != 0:
            raise ValueError("baseline window must be >= 0")

    def timers(self):
        """device ms of the last run's stages"""
        t = (C.c_float * 6)()
        self._L.pb_search_timers(self._s, t)
        return dict(zip(("h2d", "transpose", "dedisperse", "stats", "boxcar", "d2h"), [float(v) for v in t]))

    def peaks(self, codes, threshold, nbit=8, device_ptr=None, nsamp=None, max_out=1 << 18):
        """Search one block and return only the (DM, sample) points with S/N >= threshold:
        dict(dmi, t, snr, width_log2, total).  codes: host uint8 array, or device_ptr + nsamp for codes that
        are already on the GPU (e.g. PbHandle.output_dev)."""
        if device_ptr is None:
            codes = np.ascontiguousarray(codes, np.uint8).ravel()
            nsamp = codes.size * (8 // nbit) // self.nchan
            ptr, on_dev = codes.ctypes.data_as(C.c_void_p), 0
        else:
            ptr, on_dev = C.c_void_p(device_ptr), 1
        buf = np.empty((max_out, 4), np.int32)
        n, t = C.c_int(), C.c_int()
        rc = self._L.pb_search_peaks(self._s, ptr, on_dev, nsamp, nbit, C.c_float(threshold),
                                     buf.ctypes.data_as(C.POINTER(C.c_int32)), max_out, C.byref(n), C.byref(t))
        if rc != 0:
            raise self._lp.PbError("pb_search_peaks failed (%d): %s" % (rc, self._L.pb_search_last_error(self._s).decode()))
        k = min(n.value, max_out, 1 << 20)        # (the device list holds 2^20 points)
        b = buf[:k]
        return dict(dmi=b[:, 0].copy(), t=b[:, 1].copy(), snr=b[:, 2].copy().view(np.float32), width_log2=b[:, 3].astype(np.uint8),
                    total=n.value, tout=t.value)

    def run(self, codes, nbit=8, want_series=False):
        """codes: uint8 array holding nsamp x nchan samples of nbit bits (SIGPROC order).
        Returns dict(snr [ndm][tout] f32, width_log2 [ndm][tout] u8, stats [ndm][2], series?)."""
        codes = np.ascontiguousarray(codes, np.uint8).ravel()
        nsamp = codes.size * (8 // nbit) // self.nchan
        tout = nsamp - self.max_delay
        if tout < 64:
            raise ValueError("block shorter than the largest dispersion delay (%d samples)" % self.max_delay)
        snr = np.empty((self.ndm, tout), np.float32)
        wid = np.empty((self.ndm, tout), np.uint8)
        stats = np.empty((self.ndm, 2), np.float32)
        series = np.empty((self.ndm, tout), np.uint32) if want_series else None
        t = C.c_int()
        rc = self._L.pb_search_run(self._s, codes.ctypes.data_as(C.c_void_p), 0, nsamp, nbit,
                                   snr.ctypes.data_as(C.POINTER(C.c_float)), wid.ctypes.data_as(C.POINTER(C.c_uint8)),
                                   series.ctypes.data_as(C.POINTER(C.c_uint32)) if want_series else None,
                                   stats.ctypes.data_as(C.POINTER(C.c_float)), C.byref(t))
        if rc != 0:
            raise self._lp.PbError("pb_search_run failed (%d): %s" % (rc, self._L.pb_search_last_error(self._s).decode()))
        return dict(snr=snr, width_log2=wid, stats=stats, series=series, tout=tout)


def overlap(c, o, delta_dm=0.1, delta_w=3):
    """The reference coincidencer's test for "the same event" between two candidates, src/candidate.py:49-65:
    DMs within delta_dm (fractional, relative to the other's DM), widths (i1 - i0) within a factor delta_w, and the
    sample ranges [i0, i1) overlapping.  c, o: dicts with dm, i0, i1.  delta_w None: no width test."""
    if o["dm"] == 0 or abs(c["dm"] / o["dm"] - 1) > delta_dm:
        if not (o["dm"] == 0 and c["dm"] == 0):
            return False
    if delta_w is not None:
        w1, w2 = float(c["i1"] - c["i0"]), float(o["i1"] - o["i0"])
        if min(w1, w2) <= 0 or max(w1, w2) / min(w1, w2) > delta_w:
            return False
    if c["i0"] < o["i0"]:
        return o["i0"] < c["i1"]
    return c["i0"] < o["i1"]


def group_points(idm, it, sn, w, dms, tsamp, dm_tol=0.1, delta_w=None, sample0=0):
    """Above-threshold (DM index, sample, S/N, boxcar width) points of ONE beam's search -> candidates: strongest
    first, a candidate absorbs every point that overlaps it in time and lies within dm_tol (fractional) of its DM.
    This clustering is this framework's own (heimdall's, which the reference runs at this place, is third-party and
    absent: unpinned).  It uses the time and DM tests of the reference's cross-beam coincidencer (src/candidate.py:
    53-54, 64-66) but by default NOT its width-ratio test (:57-63, delta_w = 3): the points of one series are boxcar
    trials of the same samples, a pulse shows up at every width from 1 to 64, and splitting them by width would
    report one pulse several times.  delta_w (e.g. 3) applies that test as well, for like-for-like experiments.
    Ties in S/N are broken by (DM index, sample, width): the GPU appends points in arrival order, which differs from run
    to run, and the grouping depends on the order.  Returns dicts with heimdall's nine columns."""
    idm, it, w = np.asarray(idm, np.int64), np.asarray(it, np.int64), np.asarray(w, np.int64)
    sn = np.asarray(sn)
    if idm.size == 0:
        return []
    order = np.lexsort((w, it, idm, -sn.astype(np.float64)))
    idm, it, sn, w = idm[order], it[order], sn[order], w[order]
    pdm = np.asarray(dms)[idm]
    taken = np.zeros(idm.size, bool)
    cands = []
    for k in range(idm.size):
        if taken[k]:
            continue
        i0, i1, dm = it[k], it[k] + w[k], pdm[k]
        near = (~taken) & (it < i1) & (it + w > i0) & (np.abs(pdm - dm) <= dm_tol * max(dm, 1.0) + 1e-9)
        if delta_w is not None:
            near &= (np.maximum(w, w[k]) <= delta_w * np.minimum(w, w[k]))
        taken |= near
        cands.append(dict(snr=float(sn[k]), peak_idx=int(sample0 + it[k]), peak_time=float((sample0 + it[k]) * tsamp),
                          tfilt=int(np.log2(w[k])), dmi=int(idm[k]), dm=float(dm), ngiant=int(near.sum()),
                          i0=int(sample0 + it[near].min()), i1=int(sample0 + (it[near] + w[near]).max())))
    return cands


def find_candidates(snr, width_log2, dms, tsamp, threshold=6.0, dm_tol=0.1, sample0=0, delta_w=None):
    """group_points on full S/N planes (Searcher.run): every (DM, sample) at or above the threshold"""
    idm, it = np.nonzero(snr >= threshold)
    return group_points(idm, it, snr[idm, it], 1 << width_log2[idm, it].astype(np.int64), dms, tsamp, dm_tol, delta_w, sample0)


def candidates_from_peaks(pk, dms, tsamp, dm_tol=0.1, sample0=0, delta_w=None):
    """group_points on a peak list (Searcher.peaks) instead of full S/N planes"""
    return group_points(pk["dmi"], pk["t"], pk["snr"], 1 << pk["width_log2"].astype(np.int64), dms, tsamp, dm_tol, delta_w, sample0)


class GulpSearch(object):
    """The search over a stream of filterbank samples in gulps, as heimdall runs it (-nsamps_gulp 30720,
    scripts/start_heimdall_single_antenna:21): every gulp is searched together with the last
    max_delay + widest_boxcar - 1 samples of the stream before it, and of a gulp's output samples only those for
    which EVERY boxcar width fits inside the gulp are emitted (k_boxcar stops at the end of the block: the last
    widest - 1 samples have only been tried with the narrow widths); the rest are produced by the next gulp.  So
    each output sample is produced exactly once, with all widths, and neither a pulse whose sweep straddles a gulp
    boundary nor a wide one that starts just before it is lost.  push() returns the candidates of the samples that
    became complete; finish() those of the stream's last widest - 1 samples (narrow widths only: there is no more
    data); sample indices count from the start of the stream."""

    def __init__(self, searcher, threshold=6.0, nbit=8, coincidencer=None, utc_start="1970-01-01-00:00:00", beam=1):
        self.s, self.threshold, self.nbit = searcher, threshold, nbit
        self.tail = np.zeros((0, searcher.nchan), np.uint8)
        self.done = 0                    # output samples produced so far = stream index of the next block's first output
        self.ov = (1 << (searcher.nbox - 1)) - 1        # widest boxcar - 1
        self.pending = None              # peaks of the latest gulp's last `ov` output samples (for finish())
        self.coincidencer, self.utc_start, self.beam = coincidencer, utc_start, beam

    def _send(self, cands, first, nsamps):
        if self.coincidencer:
            cmod = importlib.import_module(_pkg + ".candidates")
            host, port = cmod.parse_coincidencer(self.coincidencer)
            cmod.send_candidates(host, port, self.utc_start, self.beam, cands, first_sample=first, nsamps=nsamps)

    @staticmethod
    def _subset(pk, sel):
        return dict(dmi=pk["dmi"][sel], t=pk["t"][sel], snr=pk["snr"][sel], width_log2=pk["width_log2"][sel])

    def push(self, block):
        """block: uint8 [nsamp][nchan] (8-bit codes) of NEW samples"""
        if self.nbit != 8:
            raise ValueError("GulpSearch takes unpacked 8-bit samples")
        block = np.asarray(block, np.uint8).reshape(-1, self.s.nchan)
        data = np.concatenate([self.tail, block]) if self.tail.size else block
        if data.shape[0] - self.s.max_delay < 64 + self.ov:
            self.tail = data                          # not enough yet for one output block
            return []
        pk = self.s.peaks(data, self.threshold, nbit=8)
        nout = pk["tout"] - self.ov                   # samples with every width tried
        full = pk["t"] < nout
        cands = candidates_from_peaks(self._subset(pk, full), self.s.dms, self.s.tsamp, sample0=self.done)
        self.pending = (self._subset(pk, ~full), self.done, pk["tout"])
        first = self.done
        self.done += nout
        keep = min(self.s.max_delay + self.ov, data.shape[0])
        self.tail = data[data.shape[0] - keep:].copy()
        self._send(cands, first, nout)
        return cands

    def finish(self):
        """end of the stream: candidates of the last widest - 1 output samples"""
        if self.pending is None:
            return []
        pk, sample0, tout = self.pending
        self.pending = None
        cands = candidates_from_peaks(pk, self.s.dms, self.s.tsamp, sample0=sample0)
        self._send(cands, self.done, tout - (self.done - sample0))
        self.done = sample0 + tout
        return cands


def candidate_line(c):
    """One heimdall-format text line (columns of src/candidate.py:8-18); see candidates.py for the TCP leg."""
    import importlib
    return importlib.import_module((__package__ or "vlite-fast_amd") + ".candidates").candidate_line(c)


def main(argv=None):
    """heimdall's place in the chain for a SIGPROC file: `python -m vlite-fast_amd.search -f obs_kur.fil
    -dm 2 1000 -boxcar_max 64 -nsamps_gulp 30720 -zap_chans 0 190 -zap_chans 3900 4096 -coincidencer vlite-nrl:27555`
    (the flags of scripts/start_heimdall_single_antenna:21).  Prints heimdall-format candidate lines per gulp
    and, with -coincidencer, sends them there."""
    import argparse
    sig = importlib.import_module(_pkg + ".sigproc")
    ap = argparse.ArgumentParser(prog="search", prefix_chars="-")
    ap.add_argument("-f", dest="fil", required=True)
    ap.add_argument("-dm", nargs=2, type=float, default=list(HEIMDALL_DM))
    ap.add_argument("-dm_tol", type=float, default=1.25)
    ap.add_argument("-dm_step", type=float, default=0.0, help="> 0: linear grid instead of the tolerance-spaced list")
    ap.add_argument("-boxcar_max", type=int, default=HEIMDALL_BOXCAR_MAX)
    ap.add_argument("-nsamps_gulp", type=int, default=HEIMDALL_GULP)
    ap.add_argument("-zap_chans", nargs=2, type=int, action="append", default=None)
    ap.add_argument("-detect_thresh", type=float, default=6.0)
    ap.add_argument("-baseline_length", type=float, default=2.0, help="seconds")
    ap.add_argument("-beam", type=int, default=1)
    ap.add_argument("-coincidencer", default=None)
    ap.add_argument("-gpu_id", type=int, default=0)
    a = ap.parse_args(argv)
    with open(a.fil, "rb") as f:
        hdr, off = sig.read_header(f.read(4096))
    if hdr["nbits"] != 8 or hdr.get("nifs", 1) != 1:
        raise SystemExit("search: 8-bit single-IF filterbanks only")
    nchan, tsamp = hdr["nchans"], hdr["tsamp"]
    zap = tuple(tuple(z) for z in a.zap_chans) if a.zap_chans else ()
    dml = None if a.dm_step > 0 else dedisp_dm_list(a.dm[0], a.dm[1], tsamp, hdr["fch1"], hdr["foff"], nchan, a.dm_tol)
    kw = dict(device=a.gpu_id, nchan=nchan, fch1=hdr["fch1"], foff=hdr["foff"], tsamp=tsamp, boxcar_max=a.boxcar_max, zap=zap)
    s = Searcher(max_samples=a.nsamps_gulp + int(4.148808e3 * a.dm[1] * abs((hdr["fch1"] + (nchan - 1) * hdr["foff"]) ** -2
                                                                       - hdr["fch1"] ** -2) / tsamp) + 64,
                 dm_list=dml, dm_min=a.dm[0], dm_max=a.dm[1], dm_step=a.dm_step or 2.0, **kw)
    s.set_baseline(int(a.baseline_length / tsamp))
    import time as _time
    utc = _time.strftime("%Y-%m-%d-%H:%M:%S", _time.gmtime(round((hdr.get("tstart", 40587.0) - 40587.0) * 86400.0)))
    g = GulpSearch(s, threshold=a.detect_thresh, coincidencer=a.coincidencer, utc_start=utc, beam=a.beam)
    ntot = 0
    with open(a.fil, "rb") as f:
        f.seek(off)
        while True:
            blk = np.frombuffer(f.read(a.nsamps_gulp * nchan), np.uint8)
            if blk.size < nchan:
                break
            for c in g.push(blk[:blk.size // nchan * nchan].reshape(-1, nchan)):
                print(candidate_line(c))
                ntot += 1
    for c in g.finish():
        print(candidate_line(c))
        ntot += 1
    s.close()
    return ntot


if __name__ == "__main__":
    main()
