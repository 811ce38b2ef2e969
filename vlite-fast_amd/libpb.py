"""ctypes binding of libpb_hip.so (include/pb_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is visible the
constructor raises.  Build with `make -C vlite-fast_amd/csrc` (or __graft_entry__.build()).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PB_LIBPATH") or os.path.join(_HERE, "csrc", "libpb_hip.so")

NFFT = 12500
NCHAN = 6251
NSCRUNCH = 8
NKURTO = 500
CHANMIN = 2155
CHANMAX = 6250
NCHANOUT = 4096
VLITE_RATE = 128000000
FFT_LDS = 0
FFT_HIPFFT = 1

STAGES = ("kurtosis", "channelize", "fft", "inject", "detect", "deframe", "coadd", "h2d")
DBG_POW, DBG_KUR, DBG_DAG, DBG_FLAGS, DBG_ROWWEIGHT, DBG_POW_FB, DBG_KUR_FB, DBG_DAG_FB = range(8)


class PbError(RuntimeError):
    pass


class PbConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("nant", C.c_int32),
                ("nbit", C.c_int32), ("npol", C.c_int32), ("rfi_mode", C.c_int32),
                ("taps", C.c_int32), ("fft_backend", C.c_int32), ("rows_per_seg", C.c_int32),
                ("max_seg", C.c_int32), ("inject_frb", C.c_int32), ("keep_ave", C.c_int32),
                ("debug_keep", C.c_int32), ("nsets", C.c_int32)]


class PbSizes(C.Structure):
    _fields_ = [("seg_samples_per_pol", C.c_uint64), ("input_bytes_per_seg", C.c_uint64),
                ("code_bytes_per_seg", C.c_uint64), ("ave_floats_per_seg", C.c_uint64),
                ("rows_per_seg", C.c_uint64), ("blocks_per_seg_pol", C.c_uint64),
                ("device_bytes", C.c_uint64)]


class PbTimers(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8)]


EXPORTS = ["pb_config_default", "pb_create", "pb_destroy", "pb_last_error", "pb_query",
           "pb_set_stream", "pb_sync", "pb_reset_bandpass", "pb_reset_history", "pb_get_bandpass", "pb_set_bandpass",
           "pb_submit_planar", "pb_submit_planar_dev", "pb_submit_vdif", "pb_submit_vdif_at", "pb_input_dev", "pb_process", "pb_set_frb_params", "pb_select_set", "pb_fetch", "pb_fetch_ptr",
           "pb_output_dev", "pb_coadd_local", "pb_coadd_local_codes", "pb_coadd_local_tree", "pb_coadd_tree", "pb_coadd_tree_order", "pb_coadd_digitise", "pb_coadd_publish", "pb_set_coadd_target", "pb_coadd_release", "pb_set_coadd_stream", "pb_coadd_finish", "pb_coadd_fetch_ptr", "pb_profile", "pb_get_timers", "pb_host_alloc", "pb_host_free",
           "pb_debug_fetch", "pb_debug_dag_check", "pb_channelize_f32", "pb_version", "pb_search_create", "pb_search_create_list", "pb_search_destroy",
           "pb_search_last_error", "pb_search_info", "pb_search_run", "pb_search_set_baseline", "pb_search_peaks",
           "pb_search_timers"]

_lib = None


def load():
    """dlopen libpb_hip.so and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PbError("%s not found: build it with `make -C vlite-fast_amd/csrc`; "
                      "there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 (soname
    # libamdhip64.so.7) but link it by file name.  If torch is imported AFTER this library has
    # pulled in /opt/rocm's copy, the loader maps a second runtime and torch then sees no GPU.
    # Importing torch first makes both resolve to the same (torch's) runtime.
    if os.environ.get("PB_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u8p, fp = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    L.pb_config_default.argtypes = [C.POINTER(PbConfig)]
    L.pb_config_default.restype = None
    L.pb_create.argtypes = [C.POINTER(PbConfig), C.POINTER(vp)]
    L.pb_destroy.argtypes = [vp]
    L.pb_destroy.restype = None
    L.pb_last_error.argtypes = [vp]
    L.pb_last_error.restype = C.c_char_p
    L.pb_query.argtypes = [vp, C.POINTER(PbSizes)]
    L.pb_set_stream.argtypes = [vp, vp]
    L.pb_sync.argtypes = [vp]
    L.pb_reset_bandpass.argtypes = [vp, C.c_int]
    L.pb_reset_history.argtypes = [vp, C.c_int]
    L.pb_get_bandpass.argtypes = [vp, C.c_int, fp, fp]
    L.pb_set_bandpass.argtypes = [vp, C.c_int, fp, fp]
    L.pb_submit_planar.argtypes = [vp, C.c_int, C.c_int, u8p, u8p, C.c_size_t]
    L.pb_submit_planar_dev.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_size_t]
    L.pb_submit_vdif.argtypes = [vp, C.c_int, C.c_int, u8p, C.c_size_t]
    L.pb_submit_vdif_at.argtypes = [vp, C.c_int, C.c_int, u8p, C.c_size_t, C.c_int64, C.c_int64]
    L.pb_input_dev.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.pb_process.argtypes = [vp, C.c_int, C.c_int]
    L.pb_fetch.argtypes = [vp, C.c_int, C.c_int, C.c_int, u8p, u8p, fp, fp, fp]
    L.pb_set_frb_params.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    L.pb_select_set.argtypes = [vp, C.c_int]
    L.pb_fetch_ptr.argtypes = [vp, C.c_int, C.c_int, C.POINTER(u8p)]
    L.pb_output_dev.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    L.pb_coadd_local.argtypes = [vp, C.c_int, vp, C.c_int]
    L.pb_coadd_local_codes.argtypes = [vp, C.c_int, vp, C.c_int]
    L.pb_coadd_local_tree.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.c_int, vp]
    L.pb_coadd_tree.argtypes = [vp, C.POINTER(vp), C.c_int, vp, C.c_size_t]
    L.pb_coadd_tree_order.argtypes = [C.c_int, C.POINTER(C.c_int32)]
    L.pb_coadd_digitise.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    L.pb_coadd_publish.argtypes = [vp, vp, C.c_size_t]
    L.pb_set_coadd_target.argtypes = [vp, vp]
    L.pb_coadd_release.argtypes = [vp]
    L.pb_set_coadd_stream.argtypes = [vp, vp]
    L.pb_coadd_finish.argtypes = [vp, C.c_int, vp, C.c_int, u8p]
    L.pb_coadd_fetch_ptr.argtypes = [vp, C.c_int, C.POINTER(u8p)]
    L.pb_profile.argtypes = [vp, C.c_int]
    L.pb_get_timers.argtypes = [vp, C.POINTER(PbTimers), C.c_int]
    L.pb_debug_fetch.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    L.pb_channelize_f32.argtypes = [vp, fp, C.c_int, C.c_int, fp]
    L.pb_debug_dag_check.argtypes = [vp, C.c_float, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), fp]
    L.pb_version.restype = C.c_char_p
    L.pb_host_alloc.argtypes = [C.c_size_t]
    L.pb_host_alloc.restype = vp              # (a pointer: the default int restype would truncate it)
    L.pb_host_free.argtypes = [vp]
    L.pb_host_free.restype = None
    ip = C.POINTER(C.c_int)
    L.pb_search_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int, ip, C.c_int, C.POINTER(vp)]
    L.pb_search_create_list.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, fp, C.c_int, C.c_int,
                                        ip, C.c_int, C.POINTER(vp)]
    L.pb_search_destroy.argtypes = [vp]
    L.pb_search_destroy.restype = None
    L.pb_search_last_error.argtypes = [vp]
    L.pb_search_last_error.restype = C.c_char_p
    L.pb_search_info.argtypes = [vp, ip, ip, ip]
    L.pb_search_run.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, fp, u8p, C.POINTER(C.c_uint32), fp, ip]
    L.pb_search_set_baseline.argtypes = [vp, C.c_int]
    L.pb_search_peaks.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int32), C.c_int, ip, ip]
    L.pb_search_timers.argtypes = [vp, fp]
    for n in EXPORTS:
        f = getattr(L, n)
        if f.restype is C.c_int or n in ("pb_create", "pb_query", "pb_sync"):
            f.restype = C.c_int
    _lib = L
    return L


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class PbHandle(object):
    """One GPU's baseband->filterbank state: what process_baseband allocates once per run
    (/root/reference/src/process_baseband.cu:578-709) plus the per-segment device work."""

    def __init__(self, device=0, nant=1, nbit=8, npol=1, rfi_mode=2, taps=1, fft_backend=FFT_LDS,
                 rows_per_seg=1024, max_seg=10, inject_frb=False, keep_ave=False, debug_keep=False,
                 nsets=1):
        L = load()
        cfg = PbConfig()
        L.pb_config_default(C.byref(cfg))
        cfg.device, cfg.nant, cfg.nbit, cfg.npol = device, nant, nbit, npol
        cfg.rfi_mode, cfg.taps, cfg.fft_backend = rfi_mode, taps, fft_backend
        cfg.rows_per_seg, cfg.max_seg = rows_per_seg, max_seg
        cfg.inject_frb, cfg.keep_ave, cfg.debug_keep = int(inject_frb), int(keep_ave), int(debug_keep)
        cfg.nsets = nsets
        self._L = L
        self._h = C.c_void_p()
        rc = L.pb_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = L.pb_last_error(None).decode()
            self._h = None
            if rc == -22:
                raise ValueError(msg)
            raise PbError("pb_create failed (%d): %s" % (rc, msg))
        self.cfg = cfg
        s = PbSizes()
        self._chk(L.pb_query(self._h, C.byref(s)))
        self.sizes = s
        self.seg_samples = int(s.seg_samples_per_pol)
        self.trim = int(s.code_bytes_per_seg)
        self.ave_per_seg = int(s.ave_floats_per_seg)
        self.rows = int(s.rows_per_seg)
        self.nblk = int(s.blocks_per_seg_pol)
        self.nant, self.max_seg, self.nsets = nant, max_seg, nsets
        self.cur_set = 0

    def _chk(self, rc):
        if rc != 0:
            msg = self._L.pb_last_error(self._h).decode()
            if rc == -22:
                raise ValueError(msg)
            raise PbError("libpb_hip error %d: %s" % (rc, msg))

    def close(self):
        if getattr(self, "_h", None):
            self._L.pb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- input
    def submit_planar(self, ant, seg, pol0, pol1):
        pol0 = np.ascontiguousarray(pol0, np.uint8)
        pol1 = np.ascontiguousarray(pol1, np.uint8)
        if pol0.size != pol1.size:
            raise ValueError("polarisations differ in length")
        self._chk(self._L.pb_submit_planar(self._h, ant, seg, _u8(pol0), _u8(pol1), pol0.size))

    def submit_planar_dev(self, ant, seg, d_pol0, d_pol1, nsamp):
        """d_pol0 / d_pol1: device addresses (e.g. torch tensor .data_ptr())."""
        self._chk(self._L.pb_submit_planar_dev(self._h, ant, seg, C.c_void_p(d_pol0), C.c_void_p(d_pol1), nsamp))

    def submit_vdif(self, ant, seg0, block, second=None, frame0=0):
        """Queue one block of VDIF frames (asynchronous: keep `block` untouched until the batch's output has
        been fetched).  second / frame0: time origin of the block; default = its first frame's header."""
        block = np.ascontiguousarray(block, np.uint8)
        if second is None:
            self._chk(self._L.pb_submit_vdif(self._h, ant, seg0, _u8(block), block.size))
        else:
            self._chk(self._L.pb_submit_vdif_at(self._h, ant, seg0, _u8(block), block.size, int(second), int(frame0)))

    def input_dev(self, ant):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.pb_input_dev(self._h, ant, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- compute
    def process(self, nseg, inject_now=0):
        self._chk(self._L.pb_process(self._h, nseg, inject_now))

    def set_frb_params(self, dm=80.0, width_rows=-1.0, amp=1.05):
        self._chk(self._L.pb_set_frb_params(self._h, dm, width_rows, amp))

    def select_set(self, i):
        self._chk(self._L.pb_select_set(self._h, i))
        self.cur_set = i

    def fetch_view(self, ant, stream, nseg):
        """Zero-copy numpy view of the selected set's codes in pinned host memory."""
        p = C.POINTER(C.c_uint8)()
        self._chk(self._L.pb_fetch_ptr(self._h, ant, stream, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(nseg * self.trim,))

    def sync(self):
        self._chk(self._L.pb_sync(self._h))

    def set_stream(self, stream_ptr):
        self._chk(self._L.pb_set_stream(self._h, C.c_void_p(stream_ptr)))

    # ---- output
    def fetch(self, ant, seg0, nseg, raw=True, kur=True, weights=False, ave=False):
        out = {}
        r = np.empty(nseg * self.trim, np.uint8) if raw else None
        k = np.empty(nseg * self.trim, np.uint8) if kur else None
        w = np.empty(nseg * self.rows, np.float32) if weights else None
        ar = np.empty(nseg * self.ave_per_seg, np.float32) if ave else None
        ak = np.empty(nseg * self.ave_per_seg, np.float32) if ave else None
        self._chk(self._L.pb_fetch(self._h, ant, seg0, nseg, _u8(r), _u8(k), _f32(w), _f32(ar), _f32(ak)))
        out.update(raw=r, kur=k, weights=w, ave_raw=ar, ave_kur=ak)
        return out

    def output_dev(self, ant, stream):
        c, a = C.c_void_p(), C.c_void_p()
        self._chk(self._L.pb_output_dev(self._h, ant, stream, C.byref(c), C.byref(a)))
        return c.value, a.value

    def reset_bandpass(self, ant):
        self._chk(self._L.pb_reset_bandpass(self._h, ant))

    def reset_history(self, ant):
        self._chk(self._L.pb_reset_history(self._h, ant))

    def get_bandpass(self, ant):
        r = np.empty((2, NCHANOUT), np.float32)
        k = np.empty((2, NCHANOUT), np.float32)
        self._chk(self._L.pb_get_bandpass(self._h, ant, _f32(r), _f32(k)))
        return r, k

    def set_bandpass(self, ant, raw, kur):
        raw = np.ascontiguousarray(raw, np.float32)
        kur = np.ascontiguousarray(kur, np.float32)
        assert raw.size == 2 * NCHANOUT and kur.size == 2 * NCHANOUT
        self._chk(self._L.pb_set_bandpass(self._h, ant, _f32(raw), _f32(kur)))

    def set_coadd_stream(self, stream_ptr):
        """hipStream_t (int) on which coadd_local / coadd_finish run; 0 = the main stream."""
        self._chk(self._L.pb_set_coadd_stream(self._h, C.c_void_p(stream_ptr or None)))

    def coadd_local(self, nseg, d_sum_ptr, accumulate=False):
        self._chk(self._L.pb_coadd_local(self._h, nseg, C.c_void_p(d_sum_ptr), int(accumulate)))

    def coadd_local_codes(self, nseg, d_sum_ptr, accumulate=False):
        """the local sum from the batch's quantised codes (cell centres) instead of the fp32 planes"""
        self._chk(self._L.pb_coadd_local_codes(self._h, nseg, C.c_void_p(d_sum_ptr), int(accumulate)))

    def coadd_local_tree(self, nseg, ant_order, d_dst_ptr):
        """d_dst = T_n over the fp32 planes of this handle's antennas ant_order[0..n): one rank's node of the
        fixed-order incoherent sum (include/pb_hip.h: pb_coadd_local_tree; the order is coadd.tree_order's)"""
        o = (C.c_int32 * len(ant_order))(*[int(a) for a in ant_order])
        self._chk(self._L.pb_coadd_local_tree(self._h, nseg, o, len(ant_order), C.c_void_p(d_dst_ptr)))

    def coadd_tree(self, leaf_ptrs, d_dst_ptr, nfloat):
        """d_dst[0..nfloat) = T_n over the caller's device planes leaf_ptrs (the root: the gathered partial sums)"""
        lv = (C.c_void_p * len(leaf_ptrs))(*[int(p) for p in leaf_ptrs])
        self._chk(self._L.pb_coadd_tree(self._h, lv, len(leaf_ptrs), C.c_void_p(d_dst_ptr), int(nfloat)))

    def coadd_digitise(self, d_sum_ptr, nfloat, nant_total, d_codes_ptr):
        """sel_and_dig of a flat range of the npol = 1 plane / sqrt(nant_total), device to device (a rank's slice)"""
        self._chk(self._L.pb_coadd_digitise(self._h, C.c_void_p(d_sum_ptr), int(nfloat), int(nant_total), C.c_void_p(d_codes_ptr)))

    def coadd_publish(self, d_codes_ptr, nbytes):
        """the batch's coadded bytes (assembled on the device) -> the pinned buffer coadd_view hands out"""
        self._chk(self._L.pb_coadd_publish(self._h, C.c_void_p(d_codes_ptr), int(nbytes)))

    def set_coadd_target(self, d_sum_ptr):
        """nant = 1: the selected set's detect writes the plane to be reduced straight into d_sum (0 / None: off);
        coadd_local(nseg, d_sum_ptr) then launches nothing, and coadd_release() follows the collective."""
        self._chk(self._L.pb_set_coadd_target(self._h, C.c_void_p(d_sum_ptr or None)))

    def coadd_release(self):
        self._chk(self._L.pb_coadd_release(self._h))

    def coadd_finish(self, nseg, d_sum_ptr, nant_total, blocking=True):
        if not blocking:
            self._chk(self._L.pb_coadd_finish(self._h, nseg, C.c_void_p(d_sum_ptr), nant_total, None))
            return None
        codes = np.empty(nseg * self.trim, np.uint8)
        self._chk(self._L.pb_coadd_finish(self._h, nseg, C.c_void_p(d_sum_ptr), nant_total, _u8(codes)))
        return codes

    def coadd_view(self, nseg, age=0):
        """Zero-copy view of the coadded bytes of the latest (age 0) / previous (age 1) finish."""
        p = C.POINTER(C.c_uint8)()
        self._chk(self._L.pb_coadd_fetch_ptr(self._h, age, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(nseg * self.trim,))

    def profile(self, enable=True):
        self._chk(self._L.pb_profile(self._h, int(enable)))

    def timers(self, reset=False):
        t = PbTimers()
        self._chk(self._L.pb_get_timers(self._h, C.byref(t), int(reset)))
        return {n: (t.ms[i], int(t.launches[i])) for i, n in enumerate(STAGES)}

    def debug_fetch(self, what, ant=0, seg=0):
        if what in (DBG_POW, DBG_KUR, DBG_DAG):
            a = np.empty((2, self.nblk), np.float32)
        elif what == DBG_FLAGS:
            a = np.empty(self.nblk, np.uint8)
        elif what in (DBG_POW_FB, DBG_KUR_FB):
            a = np.empty((2, self.rows), np.float32)
        else:
            a = np.empty(self.rows, np.float32)
        self._chk(self._L.pb_debug_fetch(self._h, what, ant, seg, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def dag_check(self, kur_lo, kur_hi):
        """(floats checked, mismatches, bands): the flag decision without the cube root against the score itself for
        every binary32 kurtosis in [kur_lo, kur_hi]"""
        n, bad = C.c_uint64(), C.c_uint64()
        bands = np.zeros(4, np.float32)
        self._chk(self._L.pb_debug_dag_check(self._h, kur_lo, kur_hi, C.byref(n), C.byref(bad), _f32(bands)))
        return n.value, bad.value, bands

    def channelize_f32(self, x, nrows, taps=1):
        x = np.ascontiguousarray(x, np.float32).ravel()
        assert x.size == (nrows + taps - 1) * NFFT
        out = np.empty((nrows, NCHAN), np.complex64)
        self._chk(self._L.pb_channelize_f32(self._h, _f32(x), nrows, taps, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out
