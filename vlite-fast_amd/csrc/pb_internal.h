// Internal declarations shared by the HIP translation units of libpb_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "pb_hip.h"

// D'Agostino constants (src/pb_kernels.cu:3-12 of the reference), evaluated on the host
// with the reference's float/double mix and handed to the kernel by value.
struct DagConsts {
    double one_m_2_over_A;  // (1 - 2./A)
    double mu1;
    double Z1, Z2, Z3;
    // Where the score crosses DAG_THRESH, in terms of t = (1 - 2/A) / (1 + (kur - 3 - mu1) Z3) (the argument of the
    // cube root): the block is flagged for 0 < t <= t_lo_sure and for t >= t_hi_sure, not flagged for
    // t_lo_clear <= t <= t_hi_clear; in the (few floats wide, normally empty) bands between them the score itself is
    // evaluated.  Found on the device at pb_create by evaluating the score on every float around the crossings
    // (pb_api.hip: find_dag_bands), so that a flag never needs the cube root and its eight Newton steps.
    float t_lo_sure, t_lo_clear, t_hi_clear, t_hi_sure;
};

struct FrbParams {
    const float *delays;  // [6251] or nullptr
    int since;            // nfft_since_frb for the first row of the launch
    float width;
    float amp;
};

// twiddle tables of the LDS FFT (all float2, built on the host in double); the radix-5/25/10 twiddles
// are compile-time literals (fft_consts.h)
struct FftTables {
    float2 *tw2;    // [12 pairs (r, r+1)][25 k][2]   pass-2 twiddles exp(-2 pi i r k / 625), r = 1..24 (fft_lds.h: load_t2)
    float2 *tw3;    // [625 j][10]    pass-3 twiddles exp(-2 pi i r j / 6250), r = 1..9, one pad
    float2 *post;   // [6251]
    float2 *postc;  // [4096] = post[2155..6250], its own 16-byte aligned allocation
    float *taps;    // [4][12500] FIR taps (taps=4) or nullptr
    float2 *taps_n; // the same interleaved for the PFB channeliser: [6250 n][4 taps] pairs (samples 2n, 2n+1)
};

// Scheduling choices read from the environment ONCE PER HANDLE, at pb_create (INTEGRATION.md lists them).  None of
// them changes a result: tests/test_gpu_schedules.py runs one full-size second through their combinations and
// requires identical bytes.  (Per handle, not per process: a test -- or a host -- can hold handles of several modes.)
struct PbSched {
    int overlap_detect;   // PB_OVERLAP_DETECT (1): detect + copy-out on streams of their own beside the next batch
    int kur_early;        // PB_KUR_EARLY (1): >= 3 sets: the kurtosis pass does not wait for the previous channeliser
    int detect_depth;     // PB_DETECT_DEPTH (0 = by configuration): 2 / 3 chunks in flight in detect's ring
    // read from the environment by the experiments build only (`make exp`); the shipped library uses the defaults
    int copy_dma;         // PB_COPY_DMA (0): 1 = hipMemcpyAsync instead of the copy-out kernel
    int copy_wgs;         // PB_COPY_WGS (8): workgroups of the copy-out kernel
    int det_cus;          // PB_DET_CUS (0): CU mask of the detect stream
    int det_prio;         // PB_DET_PRIO (1): detect's stream at the highest priority
};
PbSched pb_read_sched();

struct pb_handle {
    pb_config cfg;
    PbSched sched;
    int R;                 // rows per segment
    int S;                 // max segments
    int A;                 // antennas
    size_t seg_samples;    // R * 12500
    size_t nblk_seg;       // R * 25 (per pol)
    size_t trim;           // code bytes per segment
    size_t ave_per_seg;    // compact floats per segment
    hipStream_t stream;
    bool own_stream;
    size_t device_bytes;

    uint8_t *d_in;         // [A][S][2][seg_samples]
    uint8_t *d_vdif;       // staging for one raw 1-s block
    int32_t *d_frame_idx;  // [2][frames] frame -> slot in block, -1 = missing
    size_t vdif_cap;
    int32_t *h_frame_idx[8];   // page-locked frame index per buffer set (the H2D copy of it is asynchronous)
    size_t h_idx_cap[8];
    hipEvent_t ev_idx[8];      // that copy has completed: the host may rewrite the index
    uint8_t *d_flags;      // [A][S*R*25]
    float *d_wrow;         // [A][S*R] row weights, then [A][S*R] uint32 row flag masks (pb_rowmask)
    float *d_stats;        // debug: [A][3][2][S*R*25] pow,kur,dag
    float *d_fraw, *d_fkur;      // hipFFT path: f32 voltages, same indexing as d_in
    float2 *d_Xraw, *d_Xkur;     // hipFFT path: [A][S][2][R][6251]
    float *d_Praw, *d_Pkur;      // LDS path: power [A][S][2][R][4096]
    float *d_bp;           // [A][2 streams][2 pols][4096]
    uint8_t *d_codes;      // [A][2 streams][S][trim]
    float *d_ave;          // [A][2 streams][S][ave_per_seg]
    float *d_coadd_target; // nant = 1: the coadd stream's plane of antenna 0 goes here instead (pb_set_coadd_target)
    float *d_frb_delays;   // [6251]
    uint8_t *d_hist_in;    // taps=4: [2 slots][A][2][3][12512] last three rows of the previous batch
    int hist_rd;           // the slot the current batch reads (its history kernel fills the other)
    uint8_t *d_hist_flags; // taps=4: [2][A][3][25] their kurtosis flags (1 = flagged / no data)
    uint8_t *d_hist_valid; // taps=4: [2][A][3] slot holds data
    float *d_tapE;         // taps=4: [4][25] window energy per (tap, block) + total at [100]
    float frb_width, frb_amp;   // inject_frb parameters (rows, amplitude factor)
    // --- pipeline slots: the d_* buffers above (except d_bp, d_vdif, tables) exist once per
    // set; the members above always alias the SELECTED set (pb_select_set)
    uint8_t *h_codes;      // pinned mirror of d_codes, filled asynchronously after detect
    hipEvent_t ev_chan;    // kernels of this set done (its D2H may start)
    hipEvent_t ev_det;     // D2H of this set done (set may be refilled / fetched)
    hipEvent_t ev_cl;      // pb_coadd_local has read this set's fp32 planes (next detect may overwrite)
    int processed;         // segments of the last pb_process on this set
    struct BufSet {
        uint8_t *d_in, *d_flags, *d_codes, *h_codes;
        float *d_wrow, *d_stats, *d_fraw, *d_fkur, *d_Praw, *d_Pkur, *d_ave, *d_coadd_target;
        float2 *d_Xraw, *d_Xkur;
        hipEvent_t ev_chan, ev_det, ev_cl;
        int processed;
    };
    int fuse;                                          // PB_FUSE_KURTOSIS at pb_create (pb_fused_kurtosis)
    std::vector<BufSet> sets;
    int cur_set;
    hipStream_t s_det;     // detect of the previous batch (pipelined mode); copy-out in the single-set mode
    hipStream_t s_copy;    // copy-out of the filterbank bytes in pipelined mode, so that it does not hold up the next detect
    hipStream_t s_kur;     // kurtosis of the next batch, beside detect of the previous one
    hipEvent_t ev_fftdone, ev_kur, ev_alldone;
    hipEvent_t ev_hist;    // taps = 4: the batch's last rows and flags have been kept for the next one
    int last_set;          // buffer set of the previous pb_process (-1: none)
    bool staged;           // input has been queued on s_kur since the last pb_process
    uint8_t *d_coadd_codes, *h_coadd_codes;   // [2][S*trim] coadded bytes (device / pinned), lazily
    hipEvent_t ev_coadd[2];
    hipStream_t s_coadd;   // stream of pb_coadd_local / pb_coadd_finish (nullptr: the main stream)
    int coadd_slot, coadd_last;
    FftTables ft;
    DagConsts dag, dag_fb;   // D'Agostino constants for N = 500 (blocks) and N = 12500 (FFT rows, K4)
    int dag_bands;           // 1: crossings located (flags decided without the cube root); -1: PB_DAG_BANDS=0; -2: search failed, fallback
    DagConsts *d_dag;        // dag in device memory
    std::map<long, hipfftHandle> plans;

    bool profile;
    pb_timers timers;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;        // recorded, not yet read back
    std::vector<hipEvent_t> ev_pool;     // free events
    std::string err;
};

// The channeliser computes the kurtosis flags of its own rows (k_channelize_kur) instead of a kurtosis pass in
// front of it: in-library FFT, rectangular window, an RFI mode that flags, no statistics kept.  PB_FUSE_KURTOSIS=0
// keeps the two kernels (timing experiments).
bool pb_fused_kurtosis(const pb_handle *h);
// PB_EXPERIMENTS: the experiments build (libpb_hip_exp.so, `make exp`) also reads PB_SKIP, which leaves kernels out
// (results invalid; energy and upper-bound measurements).  Never in the shipped library.
#ifndef PB_EXPERIMENTS
#define PB_EXPERIMENTS 0
#endif
// row flag masks (bit r = kurtosis block r of the row is flagged), written by k_kurtosis_row behind the weights
static inline uint32_t *pb_rowmask(pb_handle *h) { return (uint32_t *)(h->d_wrow + (size_t)h->A * h->S * h->R); }

// ---- launchers (each enqueues on h->stream and returns a hipError_t) ----
hipError_t launch_kurtosis_flag(pb_handle *h, int nseg, bool write_f32);
hipError_t launch_deframe(pb_handle *h, int ant, int seg0, size_t nframes_per_thread);
hipError_t launch_dag_scan(pb_handle *h, const DagConsts &c, uint32_t bits0, int n, uint8_t *d_out);
hipError_t launch_dag_check(pb_handle *h, uint32_t bits_lo, uint64_t n, unsigned long long *d_mismatches);
hipError_t launch_inject_c64(pb_handle *h, int nseg, int inject_now);
hipError_t launch_detect(pb_handle *h, int nseg, int inject_now);
hipError_t launch_detect_pow(pb_handle *h, int nseg);
// device -> pinned host copy done by a kernel (see k_detect.hip: hipMemcpyAsync blocks the host now and then)
hipError_t launch_copy_out(const PbSched &sched, uint8_t *host_pinned, const uint8_t *dev, size_t nbytes, hipStream_t st);
hipError_t launch_channelize(pb_handle *h, int nseg, int inject_now);
hipError_t launch_channelize_pfb(pb_handle *h, int nseg, int inject_now);
hipError_t launch_pfb_weights(pb_handle *h, int nseg);
hipError_t launch_pfb_history(pb_handle *h, int nseg);   // taps = 4: the batch's last three rows and flags into the history slot the NEXT batch reads
hipError_t launch_channelize_f32(pb_handle *h, const float *d_x, int nrows, int taps, float2 *d_out);
hipError_t launch_coadd_local(pb_handle *h, int nseg, float *d_sum, int accumulate, hipStream_t st);
hipError_t launch_coadd_digitise_flat(pb_handle *h, const float *d_sum, size_t nfloat, float scale, uint8_t *d_codes,
                                      hipStream_t st);
hipError_t launch_coadd_tree(const float *const *leaves, int n, float *d_dst, size_t nfloat, hipStream_t st);
hipError_t launch_coadd_digitise(pb_handle *h, int nseg, const float *d_sum, float scale,
                                 uint8_t *d_codes, hipStream_t st);
