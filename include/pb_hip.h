/*
 * pb_hip.h -- C ABI of libpb_hip.so: the MI355X baseband -> filterbank hot path.
 *
 * The reference (kerrm/vlite-fast) has no plugin / FFI interface for this path: it is the
 * body of one executable, src/process_baseband.cu main() (:334-1614), calling the kernels
 * of src/pb_kernels.cu.  The drop-in boundary is therefore the process boundary
 * (scripts/start_process:50 -> `process_baseband -k -K -w -b -g -o -C`), and this header
 * is what a host process -- ours in Python, or the reference's own main() with its CUDA
 * calls replaced -- binds in place of the per-segment device code at
 * src/process_baseband.cu:1108-1376.  Each entry point names the reference lines it
 * replaces.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * PB_E* code and leaves a message for pb_last_error(); nothing throws across the
 * boundary (the reference is fail-stop: cudacheck -> throw 20, src/cuda_util.cu:4-12).
 * A handle owns one GPU's state and is not thread-safe; different handles are independent.
 *
 * Geometry (compile-time in the reference, src/process_baseband.h:16-55):
 *   NFFT 12500, NCHAN 6251, NSCRUNCH 8, NKURTO 500, CHANMIN 2155, CHANMAX 6250
 *   -> 4096 output channels; a *segment* is rows_per_seg FFT rows of one antenna, both
 *   polarisations (the reference fixes rows_per_seg = FFTS_PER_SEG = 1024 = 100 ms).
 * Only the 4096 output channels (FFT bins 2155..6250) are carried past the FFT: the other
 * 2155 bins never reach an output of the reference, so planes named "compact" below are
 * [..][4096] where the reference's are [..][6251] offset by CHANMIN.
 */
#ifndef PB_HIP_H
#define PB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PB_NFFT 12500
#define PB_NCHAN 6251
#define PB_NSCRUNCH 8
#define PB_NKURTO 500
#define PB_CHANMIN 2155
#define PB_CHANMAX 6250
#define PB_NCHANOUT 4096
#define PB_BLK_PER_FFT 25
#define PB_VDIF_FRAME 5032
#define PB_VDIF_DATA 5000
#define PB_FRAMES_PER_SEC 25600

#define PB_OK 0
#define PB_EINVAL (-22)
#define PB_ENOMEM (-12)
#define PB_EHIP (-5)      /* a HIP / hipFFT call failed; see pb_last_error */
#define PB_ESTATE (-1)

#define PB_FFT_LDS 0      /* in-library fixed-order radix 25x25x10 FFT in LDS (bit-exact vs oracle) */
#define PB_FFT_HIPFFT 1   /* hipFFT/rocFFT R2C plan, as the reference uses cuFFT */

typedef struct pb_handle pb_handle;

typedef struct pb_config {
    uint32_t struct_size;   /* sizeof(pb_config), for ABI growth */
    int32_t device;         /* -g   src/process_baseband.cu:458-472 */
    int32_t nant;           /* antennas batched on this GPU (reference: one process each) */
    int32_t nbit;           /* -b   2 | 4 | 8              :416-425 */
    int32_t npol;           /* -P   1 = AA+BB, 2 = AA,BB    :439-448 */
    int32_t rfi_mode;       /* -r   0 | 1 | 2              :427-437 */
    int32_t taps;           /* 1 = rectangular window (reference GPU path),
                               4 = 4-tap Hamming WOLA of analysis/baseband.py:1207 */
    int32_t fft_backend;    /* PB_FFT_LDS | PB_FFT_HIPFFT */
    int32_t rows_per_seg;   /* FFTS_PER_SEG (1024); any positive multiple of 8 */
    int32_t max_seg;        /* segments staged per pb_process call (>= 1) */
    int32_t inject_frb;     /* -i   :399-401, :711-718 */
    int32_t keep_ave;       /* also keep the fp32 pre-quantisation planes (coadd input) */
    int32_t debug_keep;     /* keep kurtosis statistics for pb_debug_fetch (selects the kurtosis kernel + channeliser
                             * pair; without it the in-library-FFT channeliser flags its own rows: same results) */
    int32_t nsets;          /* buffer sets for batch pipelining, 1..8 (1 = none, 2 = double-buffered, 3 = the host
                             * collects the batch queued two calls ago) */
} pb_config;

typedef struct pb_sizes {
    uint64_t seg_samples_per_pol;  /* rows_per_seg * 12500 */
    uint64_t input_bytes_per_seg;  /* 2 * seg_samples_per_pol */
    uint64_t code_bytes_per_seg;   /* "trim", src/process_baseband.cu:667-675 */
    uint64_t ave_floats_per_seg;   /* compact: (npol==1?1:2) * rows_per_seg/8 * 4096 */
    uint64_t rows_per_seg;
    uint64_t blocks_per_seg_pol;   /* rows_per_seg * 25 kurtosis blocks */
    uint64_t device_bytes;         /* HBM held by the handle */
} pb_sizes;

/* per-stage device time, hipEvent pairs on the handle's stream (reference: the PROFILE
 * stopwatch, src/process_baseband.h:9-13, src/process_baseband.cu:1538-1558) */
#define PB_NSTAGE 8
enum { PB_ST_KURTOSIS = 0, PB_ST_CHANNELIZE = 1, PB_ST_FFT = 2, PB_ST_INJECT = 3,
       PB_ST_DETECT = 4, PB_ST_DEFRAME = 5, PB_ST_COADD = 6, PB_ST_H2D = 7 };
typedef struct pb_timers {
    double ms[PB_NSTAGE];
    uint64_t launches[PB_NSTAGE];
} pb_timers;

/* what pb_debug_fetch can return (needs debug_keep=1 for the first three) */
enum { PB_DBG_POW = 0,      /* float [2][nblk]   kurtosis `pow`  src/pb_kernels.cu:104 */
       PB_DBG_KUR = 1,      /* float [2][nblk]   `kur`           :105 */
       PB_DBG_DAG = 2,      /* float [2][nblk]   `dag`           :132 */
       PB_DBG_FLAGS = 3,    /* uint8 [nblk]      dag > DAG_THRESH (:256), shared by both pols */
       PB_DBG_ROWWEIGHT = 4, /* float [rows]     kur_weights after apply_kurtosis (:292) */
       PB_DBG_POW_FB = 5,   /* float [2][rows]   block_kurtosis `pow_block` :140-212 (diagnostic, K4) */
       PB_DBG_KUR_FB = 6,   /* float [2][rows]   block_kurtosis `kur_block` */
       PB_DBG_DAG_FB = 7    /* float [rows]      compute_dagostino2 over FFT rows :219-241 */ };

void pb_config_default(pb_config *cfg);

/* replaces the one-time allocations + cufftPlan1d, src/process_baseband.cu:472-709 */
int pb_create(const pb_config *cfg, pb_handle **out);
/* replaces :1572-1602 */
void pb_destroy(pb_handle *h);
const char *pb_last_error(const pb_handle *h);   /* h may be NULL: last pb_create failure */
int pb_query(const pb_handle *h, pb_sizes *out);

/* run on a caller-supplied hipStream_t (e.g. torch's current stream); NULL = own stream */
int pb_set_stream(pb_handle *h, void *hip_stream);
int pb_sync(pb_handle *h);

/* bandpass state bp_dev / bp_kur_dev (:700-709): zero = "initialise from the next
 * segment's mean" (src/pb_kernels.cu:406-411,444-461).  Compact [2 pols][4096]. */
int pb_reset_bandpass(pb_handle *h, int ant);
/* taps=4 only: forget the three rows the FIR window carries over from the previous call (start of
 * a new observation).  No-op for taps=1. */
int pb_reset_history(pb_handle *h, int ant);
int pb_get_bandpass(pb_handle *h, int ant, float *bp_raw, float *bp_kur);
int pb_set_bandpass(pb_handle *h, int ant, const float *bp_raw, const float *bp_kur);

/* Stage one segment of one antenna from host memory: the two blocking H2D copies at
 * src/process_baseband.cu:1116-1122 (pol-planar 8-bit samples). */
int pb_submit_planar(pb_handle *h, int ant, int seg, const uint8_t *pol0, const uint8_t *pol1,
                     size_t nsamp_per_pol);
/* Same from device memory (HBM-resident producer), device-to-device on the handle's stream. */
int pb_submit_planar_dev(pb_handle *h, int ant, int seg, const void *d_pol0, const void *d_pol1,
                         size_t nsamp_per_pol);
/* Stage from a raw 1-s ring block of 5032-B VDIF frames (the layout writer/genbase put in
 * ring 0x40): replaces the host frame-demux loop :1015-1067 AND the H2D copies; the
 * headers are indexed on the host, payloads are gathered on the GPU.  seg0 = first of the
 * rows_per_seg-sized segments the block fills (10 per second at rows_per_seg = 1024).
 * Missing frames are zero-filled (the writer's gap fill, src/writer.c:674-688). */
int pb_submit_vdif(pb_handle *h, int ant, int seg0, const uint8_t *block, size_t nbytes);
/* The same with the block's time origin given explicitly (VDIF seconds-from-epoch of its first sample
 * and the frame number within that second), as the reference places every frame by its own header
 * (:1032-1045: frame number and thread id decide the slot, not arrival order): a frame that is absent --
 * the very first one included -- leaves zeros, frames of other seconds in the block are ignored.
 * pb_submit_vdif takes the origin from the block's first frame.  Both return as soon as the copies and
 * the gather kernel are queued; `block` must stay untouched until the batch's output has been fetched
 * (or pb_sync): page-locked memory makes the copy truly asynchronous. */
int pb_submit_vdif_at(pb_handle *h, int ant, int seg0, const uint8_t *block, size_t nbytes,
                      int64_t second, int64_t frame0);
/* HBM-resident producers write here directly: [max_seg][2 pols][seg_samples_per_pol] u8 */
int pb_input_dev(pb_handle *h, int ant, void **dptr, size_t *nbytes);

/* All device work of the segment loop body, src/process_baseband.cu:1152-1354, for
 * segments [0, nseg) of every antenna, asynchronously on the handle's stream.
 * inject_now: the reference's inject_frb_now counter (:1231-1251) for segment 0 of this
 * call (0 = none); it is advanced per segment internally. */
int pb_process(pb_handle *h, int nseg, int inject_now);

/* Parameters of the injected test FRB (needs inject_frb=1).  Reference: DM 80, 2 ms, x1.05,
 * compile-time (src/process_baseband.cu:717,1238-1239); generalised so that BASELINE config 3
 * (DM 500) can be synthesised.  width_rows < 0 selects the reference's 2 ms. */
int pb_set_frb_params(pb_handle *h, float dm, float width_rows, float amp);

/* The D2H copies at :1370-1375.  Any pointer may be NULL.  raw_codes / kur_codes:
 * nseg * code_bytes_per_seg; ave_*: nseg * ave_floats_per_seg (needs keep_ave);
 * weights: nseg * rows_per_seg, kur_weights as tscrunch_weights sees them. */
int pb_fetch(pb_handle *h, int ant, int seg0, int nseg, uint8_t *raw_codes, uint8_t *kur_codes,
             float *weights, float *ave_raw, float *ave_kur);
/* Pipelining across batches (nsets >= 2): every submit / process / fetch / debug call acts on
 * the SELECTED buffer set.  Typical loop: select(k % n); submit; process; select((k-1) % n); fetch
 * (n = 3: fetch batch k-2, whose bytes have arrived, so that the wait never delays queuing batch k+1).
 * Detect + D2H of one set run on streams of their own while the next set's channeliser (and, with
 * taps = 4 / the hipFFT back end / debug_keep, its kurtosis pass) runs on the first; the bandpass state
 * is shared and advances in pb_process order.  A set's input may be restaged once its previous detect
 * is done; its filterbank bytes must have been fetched before the pb_process after next on that set. */
int pb_select_set(pb_handle *h, int set);
/* zero-copy view of the selected set's filterbank bytes in pinned host memory (valid until the
 * next pb_process on that set): [max_seg][code_bytes_per_seg] of antenna ant, stream 0/1 */
int pb_fetch_ptr(pb_handle *h, int ant, int stream, const uint8_t **codes);
/* device addresses of the same outputs, [max_seg][..] each (stream 0 = raw, 1 = kur) */
int pb_output_dev(pb_handle *h, int ant, int stream, void **codes, void **ave);

/* Incoherent sum (replaces the external MPI coadder scripts/start_coadd:16; arithmetic
 * unpinned): d_sum[seg][ave_floats] (+)= sum over this handle's antennas
 * of their fp32 kur-stream planes (raw stream if rfi_mode 0).  The cross-GPU step is an
 * RCCL reduce of d_sum done by the host (torch.distributed); then on the root:
 * codes = sel_and_dig(d_sum / sqrt(nant_total)). */
int pb_coadd_local(pb_handle *h, int nseg, float *d_sum, int accumulate);
/* The same local sum from the batch's QUANTISED codes instead of the fp32 planes: d_sum (+)= sum over antennas of
 * the centre of each code's quantiser cell, in the fp32 planes' layout (so the reduce and pb_coadd_finish are
 * unchanged).  This is what a coadder fed from the co rings (writer.c:343-352) can do; offered for like-for-like
 * comparisons with the fp32 mode, which is the default.  Works with keep_ave = 0. */
int pb_coadd_local_codes(pb_handle *h, int nseg, float *d_sum, int accumulate);
/* One antenna per handle (nant = 1, the sharding of BASELINE configs[3] at one antenna per GPU): the plane to
 * be reduced IS the antenna's fp32 plane, so detect can write it straight into the caller's buffer and the local
 * sum needs no kernel at all (a 21-MB copy per second of data that cost 0.085 ms of a 0.68-ms step beside the
 * channeliser).  pb_set_coadd_target gives the SELECTED buffer set its own d_sum (NULL: back to the internal
 * plane); pb_coadd_local(h, nseg, that pointer, 0) then only orders the coadd stream behind detect, and
 * pb_coadd_release -- called after the host has queued the collective (and pb_coadd_finish on the root) on the
 * coadd stream -- tells the library that the buffer may be overwritten by the set's next batch. */
/* The incoherent sum in a DEFINED order, so that the coadded bytes do not depend on how the antennas are spread over
 * GPUs or on a collective's internal order (the reference's coadder, scripts/start_coadd:16,20-58,
 * is an external MPI program whose arithmetic is not in the repository).  The order: antennas split by the parity of
 * their index, recursively -- S(o, s) = S(o, 2s) + S(o + s, 2s), a single antenna's plane at the leaves, coadded =
 * S(0, 1).  With the leaves listed left to right that is T_1(x) = x, T_n(x_0..x_{n-1}) = T_ceil(n/2)(x_0..) +
 * T_floor(n/2)(x_ceil(n/2)..), which is what these two calls evaluate, fp32, element by element:
 *   pb_coadd_local_tree: d_dst[seg][ave_floats] = T_n over the fp32 planes (kur stream; raw if rfi_mode 0) of this
 *     handle's antennas ant_order[0..n) of the selected set's batch -- one rank's node of the tree.  Ordered against
 *     detect and the set's next batch like pb_coadd_local.  Needs keep_ave.
 *   pb_coadd_tree: d_dst[0..nfloat) = T_n over n caller-owned device planes d_leaves[0..n) (the root: the ranks'
 *     partial sums as gathered), on the coadd stream.  d_dst must not be one of the leaves unless n = 1.
 * n <= PB_COADD_MAX_LEAVES; planes 16-byte aligned, nfloat a multiple of 4.  Which plane is which leaf is the host's
 * business (vlite-fast_amd/coadd.py: tree_order); pb_coadd_finish(d_dst) follows on the root. */
#define PB_COADD_MAX_LEAVES 32
/* the leaf order of that tree over 0 .. n-1 (even positions first, recursively: 0 | 0 1 | 0 2 1 | 0 2 1 3 | 0 4 2 1 3 |
 * ... ; bit reversal when n is a power of two): order[0..n) for 1 <= n <= PB_COADD_MAX_LEAVES.  Host only, no handle. */
int pb_coadd_tree_order(int n, int32_t *order);
int pb_coadd_local_tree(pb_handle *h, int nseg, const int32_t *ant_order, int n, float *d_dst);
int pb_coadd_tree(pb_handle *h, const float *const *d_leaves, int n, float *d_dst, size_t nfloat);
/* The root's work spread over the ranks ("sliced" layout of the ordered sum, coadd.py): every rank sums and
 * requantises 1/W of the plane, and only code bytes travel to rank 0.
 *   pb_coadd_digitise: d_codes[0 .. nfloat * nbit / 8) = sel_and_dig(d_sum[0 .. nfloat) / sqrt(nant_total)) for a FLAT
 *     range of the npol = 1 plane (sample i of the plane is sample i of the code stream; with two polarisations the code
 *     stream interleaves them, src/pb_kernels.cu:723-727, and a flat range is not a code range: PB_ESTATE).  Device to
 *     device, on the coadd stream; nfloat a multiple of 8.
 *   pb_coadd_publish: the coadded code bytes of one batch, assembled on the device by the caller (the gathered
 *     slices), go to the pinned host buffer that pb_coadd_fetch_ptr hands out -- what pb_coadd_finish does with the
 *     bytes it computes itself.  nbytes <= max_seg * code_bytes_per_seg. */
int pb_coadd_digitise(pb_handle *h, const float *d_sum, size_t nfloat, int nant_total, uint8_t *d_codes);
int pb_coadd_publish(pb_handle *h, const uint8_t *d_codes, size_t nbytes);
int pb_set_coadd_target(pb_handle *h, float *d_sum);
int pb_coadd_release(pb_handle *h);
/* Run pb_coadd_local / pb_coadd_finish (and so the collective the host queues between them) on
 * `stream` (a hipStream_t; NULL = the handle's main stream, the default).  On a stream of its own the
 * sum of batch k, its RCCL reduce and the root's requantisation overlap the kernels of batch k+1;
 * the library orders them against detect of the same buffer set with events. */
int pb_set_coadd_stream(pb_handle *h, void *stream);
int pb_coadd_finish(pb_handle *h, int nseg, const float *d_sum, int nant_total,
                    uint8_t *codes_host);   /* codes_host NULL: asynchronous, see pb_coadd_fetch_ptr */
/* pinned-host view of the coadded bytes of the latest (age 0) or previous (age 1) pb_coadd_finish */
int pb_coadd_fetch_ptr(pb_handle *h, int age, const uint8_t **codes);

int pb_profile(pb_handle *h, int enable);
int pb_get_timers(pb_handle *h, pb_timers *out, int reset);
int pb_debug_fetch(pb_handle *h, int what, int ant, int seg, void *dst, size_t nbytes);

/* Test hook for the flag decision (src/pb_kernels.cu:109-134, DAG_THRESH): the kernels decide "score > 3" from the
 * cube root's ARGUMENT against crossings located at pb_create and evaluate the score itself only in the few-float
 * bands around them.  For every binary32 kurtosis in [kur_lo, kur_hi] the device compares that decision with the
 * score; *nmismatch must come back 0.  bands4 (may be NULL): t_lo_sure, t_lo_clear, t_hi_clear, t_hi_sure. */
int pb_debug_dag_check(pb_handle *h, float kur_lo, float kur_hi, uint64_t *nchecked, uint64_t *nmismatch,
                       float *bands4);

/* channeliser alone, for the taps=4 parity test against polyphase_filterbank
 * (analysis/baseband.py:1207): x is nrows+taps-1 rows of 12500 float32 on the host,
 * out is nrows x 6251 complex64 (interleaved re,im). */
int pb_channelize_f32(pb_handle *h, const float *x, int nrows, int taps, float *out);

/* ---- downstream search (replaces the external heimdall for BASELINE config 5) ----
 * Brute-force incoherent dedispersion over a linear DM grid + boxcar matched filter on a block of
 * filterbank codes (SIGPROC order [time][channel], 8/4/2 bit).  Delay constant 4.148808e3 MHz^2 s
 * (src/candidate.py:33); reference = top of the band.  zap_ranges: nzap pairs [lo, hi) of channels
 * to ignore (heimdall's -zap_chans).  Parity with heimdall: unpinned (third-party, absent).
 * The band, the sample time and the DM grid are binary32 in this interface (SIGPROC headers carry doubles: the caller
 * rounds); the delays are floor(x + 0.5) of the formula evaluated in DOUBLE from those rounded values, DM i of the grid
 * being (double)dm_min + i * (double)dm_step -- a checker must round its parameters the same way, or a delay within
 * 1e-7 of a half sample lands on the other side (tests/test_gpu_search.py: random geometries). */
typedef struct pb_search pb_search;
int pb_search_create(int device, int nchan, int max_samples, float fch1_mhz, float foff_mhz, float tsamp_s,
                     float dm_min, float dm_max, float dm_step, int boxcar_max, const int *zap_ranges,
                     int nzap, pb_search **out);
/* The same over an explicit, strictly ascending list of trial DMs (e.g. search.dedisp_dm_list). */
int pb_search_create_list(int device, int nchan, int max_samples, float fch1_mhz, float foff_mhz, float tsamp_s,
                          const float *dm_list, int ndm, int boxcar_max, const int *zap_ranges, int nzap,
                          pb_search **out);
void pb_search_destroy(pb_search *s);
const char *pb_search_last_error(const pb_search *s);
int pb_search_info(const pb_search *s, int *ndm, int *nbox, int *max_delay);
/* outputs (host; any may be NULL): snr / width_log2 / series are [ndm][tout], tout = nsamp - max_delay;
 * stats is [ndm][2] = clipped mean and rms of each dedispersed series */
int pb_search_run(pb_search *s, const void *codes, int codes_on_device, int nsamp, int nbit, float *snr,
                  uint8_t *width_log2, uint32_t *series, float *stats, int *tout);
/* Running baseline: the mean over `window_samples` samples around each sample is removed from every
 * dedispersed series before it is normalised (heimdall smooths its baseline over 2 s); 0 (default) = one
 * clipped mean per series. */
int pb_search_set_baseline(pb_search *s, int window_samples);
/* The same search returning only the points at or above `threshold`: peaks[4 i ..] = DM index, sample,
 * S/N (float bits), log2 width, at most max_out of them; *npeaks = how many there were. */
int pb_search_peaks(pb_search *s, const void *codes, int codes_on_device, int nsamp, int nbit, float threshold,
                    int32_t *peaks, int max_out, int *npeaks, int *tout);
/* device time of the last run's stages in ms: H2D, transpose, dedisperse, prefix + statistics, boxcar, D2H */
int pb_search_timers(const pb_search *s, float *ms6);

/* Page-locked host memory for the blocks handed to pb_submit_vdif(_at) (H2D copies from pageable memory are
 * staged and synchronous); the reference allocates its 1-s host buffer the same way (cudaMallocHost,
 * src/process_baseband.cu:579).  A host program above this ABI needs no HIP headers.  NULL on failure. */
void *pb_host_alloc(size_t nbytes);
void pb_host_free(void *p);

const char *pb_version(void);

#ifdef __cplusplus
}
#endif
#endif
