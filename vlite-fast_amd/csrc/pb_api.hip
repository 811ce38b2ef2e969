// C ABI of libpb_hip.so (include/pb_hip.h): handle lifetime, staging, the per-call kernel
// sequence that replaces the segment loop body of src/process_baseband.cu:1108-1376, and
// the D2H side.  No CPU fallback exists: every entry point needs a HIP device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kurtosis_dev.h"
#include "pb_internal.h"

static std::string g_create_err;
static void drain_timers(pb_handle *h);

static hipError_t sync_all(pb_handle *h)
{
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && h->s_det) e = hipStreamSynchronize(h->s_det);
    if (e == hipSuccess && h->s_copy) e = hipStreamSynchronize(h->s_copy);
    if (e == hipSuccess && h->s_kur) e = hipStreamSynchronize(h->s_kur);
    if (e == hipSuccess && h->s_coadd) e = hipStreamSynchronize(h->s_coadd);
    return e;
}

#define HIPCHK(h, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return PB_EHIP;                                                               \
        }                                                                                 \
    } while (0)

static int fail(pb_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    else g_create_err = msg;
    return code;
}

extern "C" const char *pb_version(void) { return "pb_hip 0.1 (gfx950)"; }

// Launches left out on request -- RESULTS INVALID -- for the energy / upper-bound measurements of tools/energy_probe.py,
// tools/taps1_unfused_bound.sh and tools/pfb_fused_ab.sh: bit 0 the channeliser, bit 1 detect, bit 2 the kurtosis pass.
// Compiled only into the experiments build (`make exp` -> libpb_hip_exp.so, -DPB_EXPERIMENTS=1, selected with
// PB_LIBPATH): the shipped library does not read the variable and cannot be talked into skipping a kernel.
static inline int pb_skip_mask()
{
#if PB_EXPERIMENTS
    static const int skip = getenv("PB_SKIP") ? atoi(getenv("PB_SKIP")) : 0;
    return skip;
#else
    return 0;
#endif
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

PbSched pb_read_sched()
{
    PbSched s;
    s.overlap_detect = env_int("PB_OVERLAP_DETECT", 1);
    s.kur_early = env_int("PB_KUR_EARLY", 1);
    s.detect_depth = env_int("PB_DETECT_DEPTH", 0);
    // switches that only ever served one timing experiment each are read by the experiments build alone
    s.copy_dma = 0;
    s.copy_wgs = 8;
    s.det_cus = 0;
    s.det_prio = 1;
#if PB_EXPERIMENTS
    s.copy_dma = env_int("PB_COPY_DMA", 0);
    s.copy_wgs = env_int("PB_COPY_WGS", 8);
    s.det_cus = env_int("PB_DET_CUS", 0);
    s.det_prio = env_int("PB_DET_PRIO", 1);
#endif
    return s;
}

bool pb_fused_kurtosis(const pb_handle *h)
{
    // h->fuse (PB_FUSE_KURTOSIS when the handle was created): 0 never, 1 (default) the rectangular window.  taps = 4
    // keeps kurtosis pass + weights + channeliser: a channeliser that flags its own rows was built in round 4, was
    // bit-exact and 17 % SLOWER (profiles/r04_notes.md section 10) and left the library in round 5
    // (tools/experiments/k_channelize_pfb_kur.patch).
    if (h->fuse < 1 || h->cfg.taps != 1) return false;
    return h->cfg.fft_backend == PB_FFT_LDS && h->cfg.rfi_mode != 0 && !h->cfg.debug_keep;
}

extern "C" void *pb_host_alloc(size_t nbytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, nbytes ? nbytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void pb_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" void pb_config_default(pb_config *c)
{
    memset(c, 0, sizeof *c);
    c->struct_size = sizeof *c;
    c->device = 0;
    c->nant = 1;
    c->nbit = 2;        // reference default NBIT, src/process_baseband.cu:31,54
    c->npol = 1;
    c->rfi_mode = 2;    // :351
    c->taps = 1;
    c->fft_backend = PB_FFT_LDS;
    c->rows_per_seg = 1024;
    c->max_seg = 10;
    c->inject_frb = 0;
    c->keep_ave = 0;
    c->debug_keep = 0;
    c->nsets = 1;
}

// D'Agostino constants with the reference's float/double evaluation (src/pb_kernels.cu:4-11)
static DagConsts make_dag(float NK)
{
    const double mu1 = -6. / (NK + 1);
    const float den = (NK + 1) * (NK + 1) * (NK + 3) * (NK + 5);
    const double mu2 = (24. * NK * (NK - 2) * (NK - 3)) / den;
    const float q = NK * NK - 5 * NK + 2;
    const float d2 = (NK + 7) * (NK + 9);
    const float d3 = NK * (NK - 2) * (NK - 3);
    const double g1 = 6. * q / d2 * sqrt((6. * (NK + 3) * (NK + 5)) / d3);
    const double A = 6. + (8. / g1) * (2. / g1 + sqrt(1. + 4. / (g1 * g1)));
    DagConsts c;
    c.t_lo_sure = c.t_hi_clear = 0.f;                                  // (no crossings known: every t is in a
    c.t_lo_clear = c.t_hi_sure = __builtin_inff();                     //  "band", the score itself decides)
    c.one_m_2_over_A = (1 - 2. / A);
    c.mu1 = mu1;
    c.Z1 = sqrt(4.5 * A);
    c.Z2 = 1 - 2. / (9 * A);
    c.Z3 = sqrt(2. / (mu2 * (A - 4)));
    return c;
}

// Where the D'Agostino score crosses DAG_THRESH as a function of t (DagConsts): the crossings are located on the host
// (the score is the same IEEE double arithmetic there) by bisection over the positive floats, then the DEVICE
// evaluates the score on every float within DAG_WIN of each crossing and the bands are read off its answers: below /
// above a crossing by more than the window the score is monotonic beyond doubt (one float step of t moves it by
// 4e-7, the cube root's last-ulp wobble by at most 1.2e-6: three steps).
#define DAG_WIN 256
static int find_dag_bands(pb_handle *h, DagConsts &c)
{
    auto flagged = [&](uint32_t bits) { return dag_of_t(__builtin_bit_cast(float, bits), c) > 3.0f; };
    const float t_mid = (float)(c.Z2 * c.Z2 * c.Z2);                 // score 0
    const uint32_t b_mid = __builtin_bit_cast(uint32_t, t_mid);
    uint32_t lo = __builtin_bit_cast(uint32_t, 1e-6f), hi = b_mid;   // flagged at lo, clear at hi
    if (!flagged(lo) || flagged(hi)) return fail(h, PB_ESTATE, "D'Agostino score: no lower crossing");
    while (hi - lo > 1) {
        const uint32_t m = lo + (hi - lo) / 2;
        if (flagged(m)) lo = m; else hi = m;
    }
    const uint32_t cross_lo = hi;
    lo = b_mid;
    hi = __builtin_bit_cast(uint32_t, 100.0f);                        // clear at lo, flagged at hi
    if (flagged(lo) || !flagged(hi)) return fail(h, PB_ESTATE, "D'Agostino score: no upper crossing");
    while (hi - lo > 1) {
        const uint32_t m = lo + (hi - lo) / 2;
        if (flagged(m)) hi = m; else lo = m;
    }
    const uint32_t cross_hi = hi;
    const int n = 2 * DAG_WIN + 1;
    uint8_t *d_f = nullptr;
    std::vector<uint8_t> f(2 * n);
    HIPCHK(h, hipMalloc((void **)&d_f, 2 * n));
    hipError_t e = launch_dag_scan(h, c, cross_lo - DAG_WIN, n, d_f);
    if (e == hipSuccess) e = launch_dag_scan(h, c, cross_hi - DAG_WIN, n, d_f + n);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(f.data(), d_f, 2 * n, hipMemcpyDeviceToHost);
    (void)hipFree(d_f);
    HIPCHK(h, e);
    const uint8_t *a = f.data(), *b = f.data() + n;
    // lower crossing: flagged ... flagged | band | clear ... clear
    int first_clear = 0, last_flag = n - 1;
    while (first_clear < n && a[first_clear]) ++first_clear;
    while (last_flag >= 0 && !a[last_flag]) --last_flag;
    // upper crossing: clear ... clear | band | flagged ... flagged
    int first_flag = 0, last_clear = n - 1;
    while (first_flag < n && !b[first_flag]) ++first_flag;
    while (last_clear >= 0 && b[last_clear]) --last_clear;
    if (first_clear < 8 || last_flag > n - 9 || last_flag < 0 || first_clear >= n || first_flag < 8 || last_clear > n - 9 ||
        last_clear < 0 || first_flag >= n)
        return fail(h, PB_ESTATE, "D'Agostino score: crossing not inside the scanned window");
    c.t_lo_sure = __builtin_bit_cast(float, cross_lo - DAG_WIN + (uint32_t)(first_clear - 1));
    c.t_lo_clear = __builtin_bit_cast(float, cross_lo - DAG_WIN + (uint32_t)(last_flag + 1));
    c.t_hi_clear = __builtin_bit_cast(float, cross_hi - DAG_WIN + (uint32_t)(first_flag - 1));
    c.t_hi_sure = __builtin_bit_cast(float, cross_hi - DAG_WIN + (uint32_t)(last_clear + 1));
    return PB_OK;
}

template <class T>
static hipError_t dmalloc(pb_handle *h, T **p, size_t n)
{
    if (n == 0) { *p = nullptr; return hipSuccess; }
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e == hipSuccess) h->device_bytes += n * sizeof(T);
    return e;
}

static float2 cis_neg(double num, double den)
{
    const double a = 2.0 * M_PI * num / den;
    return make_float2((float)cos(a), (float)(-sin(a)));
}

static int build_fft_tables(pb_handle *h)
{
    std::vector<float2> tw2(600), tw3(6250), post(PB_NCHAN);
    // fft_lds.h's layouts.  tw2[pair = (r-1)/2][k][2], r = 1..24: twiddles (r, r + 1) as one 16-byte entry with the lane's
    // k fastest, so a wave's load touches 4 cache lines, not one 192-byte row per k.  tw3[j][r-1], r = 1..9 (+ 1 pad):
    // per-thread contiguous (the same re-layout of tw3 gains as much again but costs the two-kernel channeliser a
    // register spill, profiles/r05_notes.md)
    for (int k = 0; k < 25; ++k)
        for (int r = 1; r < 25; ++r) tw2[(((r - 1) / 2) * 25 + k) * 2 + ((r - 1) & 1)] = cis_neg((double)(r * k), 625.0);
    for (int j = 0; j < 625; ++j) {
        for (int r = 1; r < 10; ++r) tw3[j * 10 + (r - 1)] = cis_neg((double)(r * j), 6250.0);
        tw3[j * 10 + 9] = make_float2(0.f, 0.f);
    }
    for (int k = 0; k < PB_NCHAN; ++k) {
        const double a = 2.0 * M_PI * (double)k / (double)PB_NFFT;
        post[k] = make_float2((float)(-sin(a)), (float)(-cos(a)));
    }
    FftTables &t = h->ft;
    HIPCHK(h, dmalloc(h, &t.tw2, 600));
    HIPCHK(h, dmalloc(h, &t.tw3, 6250));
    HIPCHK(h, dmalloc(h, &t.post, PB_NCHAN));
    HIPCHK(h, hipMemcpy(t.tw2, tw2.data(), 600 * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(t.tw3, tw3.data(), 6250 * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(t.post, post.data(), PB_NCHAN * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(h, &t.postc, (size_t)PB_NCHANOUT));
    HIPCHK(h, hipMemcpy(t.postc, post.data() + PB_CHANMIN, PB_NCHANOUT * sizeof(float2), hipMemcpyHostToDevice));
    // 4-tap Hamming WOLA taps of analysis/baseband.py:1207-1232 (always built: 200 KB):
    // tap j = window[j ns : (j+1) ns] * norms[0] * (j ? norms[j] : 1), evaluated in double
    {
        const int ns = PB_NFFT, nw = 4;
        std::vector<double> win((size_t)nw * ns);
        for (size_t k = 0; k < win.size(); ++k)
            win[k] = 0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(win.size() - 1));
        double norms[4];
        for (int j = 0; j < nw; ++j) {
            double s = 0;
            for (int k = 0; k < ns; ++k) s += win[(size_t)j * ns + k] * win[(size_t)j * ns + k];
            norms[j] = 1. / s;
        }
        std::vector<float> taps((size_t)nw * ns);
        for (int j = 0; j < nw; ++j)
            for (int k = 0; k < ns; ++k)
                taps[(size_t)j * ns + k] = (float)(win[(size_t)j * ns + k] * norms[0] * (j ? norms[j] : 1.0));
        HIPCHK(h, dmalloc(h, &t.taps, taps.size()));
        HIPCHK(h, hipMemcpy(t.taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice));
        std::vector<float2> tn((size_t)(ns / 2) * 4);
        for (int n = 0; n < ns / 2; ++n)
            for (int j = 0; j < 4; ++j)
                tn[(size_t)n * 4 + j] = make_float2(taps[(size_t)j * ns + 2 * n], taps[(size_t)j * ns + 2 * n + 1]);
        HIPCHK(h, dmalloc(h, &t.taps_n, tn.size()));
        HIPCHK(h, hipMemcpy(t.taps_n, tn.data(), tn.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    return PB_OK;
}

static void store_set(pb_handle *h, int i)
{
    pb_handle::BufSet &b = h->sets[i];
    b.d_in = h->d_in; b.d_flags = h->d_flags; b.d_codes = h->d_codes; b.h_codes = h->h_codes;
    b.d_wrow = h->d_wrow; b.d_stats = h->d_stats; b.d_fraw = h->d_fraw; b.d_fkur = h->d_fkur;
    b.d_Praw = h->d_Praw; b.d_Pkur = h->d_Pkur; b.d_ave = h->d_ave; b.d_Xraw = h->d_Xraw; b.d_Xkur = h->d_Xkur;
    b.d_coadd_target = h->d_coadd_target;
    b.ev_chan = h->ev_chan; b.ev_det = h->ev_det; b.ev_cl = h->ev_cl; b.processed = h->processed;
}

static void load_set(pb_handle *h, int i)
{
    const pb_handle::BufSet &b = h->sets[i];
    h->d_in = b.d_in; h->d_flags = b.d_flags; h->d_codes = b.d_codes; h->h_codes = b.h_codes;
    h->d_wrow = b.d_wrow; h->d_stats = b.d_stats; h->d_fraw = b.d_fraw; h->d_fkur = b.d_fkur;
    h->d_Praw = b.d_Praw; h->d_Pkur = b.d_Pkur; h->d_ave = b.d_ave; h->d_Xraw = b.d_Xraw; h->d_Xkur = b.d_Xkur;
    h->d_coadd_target = b.d_coadd_target;
    h->ev_chan = b.ev_chan; h->ev_det = b.ev_det; h->ev_cl = b.ev_cl; h->processed = b.processed;
    h->cur_set = i;
}

// set_frb_delays, src/pb_kernels.cu:338-346; width_rows < 0 selects the reference's 2 ms
// (frb_width = 2e-3*SEG_PER_SEC*FFTS_PER_SEG with the macro expanded left to right, :1238)
static int set_frb(pb_handle *h, float dm, float width_rows, float amp)
{
    std::vector<float> d(PB_NCHAN);
    const double rate = (double)h->R * PB_NFFT * 10;
    for (int i = 0; i < PB_NCHAN; ++i) {
        const double freq = 0.384 - (i * 0.064) / PB_NCHAN;
        const double scale = 4.15e-3 * dm * 10 * rate / 10 / PB_NFFT;
        d[i] = (float)(scale / (freq * freq) - scale / (0.384 * 0.384));
    }
    if (!h->d_frb_delays) HIPCHK(h, dmalloc(h, &h->d_frb_delays, (size_t)PB_NCHAN));
    HIPCHK(h, hipMemcpy(h->d_frb_delays, d.data(), PB_NCHAN * sizeof(float), hipMemcpyHostToDevice));
    h->frb_width = width_rows < 0 ? (float)(2e-3 * 10 * rate / 10 / PB_NFFT) : width_rows;
    h->frb_amp = amp;
    return PB_OK;
}

static int create_impl(pb_handle *h)
{
    const pb_config &c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    {
        // detect is latency-bound and small (one workgroup per CU): give its stream the highest
        // priority so that its workgroups slot in between the channeliser's as CUs free up
        int lo = 0, hi = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        // timing experiments: PB_DET_CUS=n confines detect to n CUs of every 32 (a CU mask on its stream), so that
        // its workgroups pack three to a CU there instead of taking one channeliser slot on every CU
        const int det_cus = h->sched.det_cus;
        if (det_cus > 0 && det_cus < 32) {
            uint32_t mask[8];
            for (int i = 0; i < 8; ++i) mask[i] = (1u << det_cus) - 1u;
            HIPCHK(h, hipExtStreamCreateWithCUMask(&h->s_det, 8, mask));
        } else {
            // (PB_DET_PRIO=0: detect's stream at the default priority; timing experiments)
            HIPCHK(h, hipStreamCreateWithPriority(&h->s_det, hipStreamNonBlocking, h->sched.det_prio ? hi : lo));
        }
    }
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_kur, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_copy, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fftdone, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_kur, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_hist, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_alldone, hipEventDisableTiming));
    const size_t A = h->A, S = h->S, R = h->R;
    const size_t in_elems = A * S * 2 * h->seg_samples;
    HIPCHK(h, dmalloc(h, &h->d_bp, A * 2 * 2 * PB_NCHANOUT));
    HIPCHK(h, hipMemset(h->d_bp, 0, A * 2 * 2 * PB_NCHANOUT * sizeof(float)));  // :702,708
    // One buffer set per pipeline slot (cfg.nsets): while detect + D2H of one batch run on the
    // second stream, kurtosis + channeliser of the next batch fill the other set.
    h->sets.resize(c.nsets);
    for (int si = 0; si < c.nsets; ++si) {
    h->d_in = h->d_flags = h->d_codes = h->h_codes = nullptr;
    h->d_wrow = h->d_stats = h->d_fraw = h->d_fkur = h->d_Praw = h->d_Pkur = h->d_ave = nullptr;
    h->d_Xraw = h->d_Xkur = nullptr;
    h->d_coadd_target = nullptr;
    HIPCHK(h, dmalloc(h, &h->d_in, in_elems + 64));   // + overhang of the channeliser's 16-byte row loads
    HIPCHK(h, dmalloc(h, &h->d_flags, A * S * h->nblk_seg));
    HIPCHK(h, hipMemset(h->d_flags, 0, A * S * h->nblk_seg));
    HIPCHK(h, dmalloc(h, &h->d_wrow, 2 * A * S * R));   // weights, then the rows' flag masks (pb_rowmask)
    HIPCHK(h, hipMemset(h->d_wrow + A * S * R, 0, A * S * R * sizeof(uint32_t)));
    {
        // rfi_mode 0 never computes weights: every row counts fully
        std::vector<float> ones(A * S * R, 1.0f);
        HIPCHK(h, hipMemcpy(h->d_wrow, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (c.debug_keep) HIPCHK(h, dmalloc(h, &h->d_stats, A * 6 * S * h->nblk_seg + A * 5 * S * R));
    const bool need_raw = c.rfi_mode != 1, need_kur = c.rfi_mode != 0;
    if (c.fft_backend == PB_FFT_HIPFFT) {
        if (need_raw) HIPCHK(h, dmalloc(h, &h->d_fraw, in_elems));
        if (need_kur) HIPCHK(h, dmalloc(h, &h->d_fkur, in_elems));
        const size_t xn = A * S * 2 * R * PB_NCHAN;
        if (need_raw) HIPCHK(h, dmalloc(h, &h->d_Xraw, xn));
        if (need_kur) HIPCHK(h, dmalloc(h, &h->d_Xkur, xn));
    }
    {
        // power planes: written by the LDS channeliser, or from the complex spectra of hipFFT
        const size_t pn = A * S * 2 * R * PB_NCHANOUT;
        if (need_raw) HIPCHK(h, dmalloc(h, &h->d_Praw, pn));
        if (need_kur) HIPCHK(h, dmalloc(h, &h->d_Pkur, pn));
    }
    HIPCHK(h, dmalloc(h, &h->d_codes, A * 2 * S * h->trim));
    HIPCHK(h, hipMemset(h->d_codes, 0, A * 2 * S * h->trim));
    HIPCHK(h, hipHostMalloc((void **)&h->h_codes, A * 2 * S * h->trim, hipHostMallocDefault));
    memset(h->h_codes, 0, A * 2 * S * h->trim);
    if (c.keep_ave) {
        HIPCHK(h, dmalloc(h, &h->d_ave, A * 2 * S * h->ave_per_seg));
        HIPCHK(h, hipMemset(h->d_ave, 0, A * 2 * S * h->ave_per_seg * sizeof(float)));
    }
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_chan, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_det, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_cl, hipEventDisableTiming));
    h->processed = 0;
    store_set(h, si);
    }
    load_set(h, 0);
    if (c.inject_frb) {
        // DM of 80, 2 ms, amplitude 1.05: src/process_baseband.cu:717,1238-1239
        int rc = set_frb(h, 80.f, -1.f, 1.05f);
        if (rc) return rc;
    }
    int rc = build_fft_tables(h);
    if (rc) return rc;
    if (c.taps == 4) {
        // two slots each: a batch's channeliser and weights read slot hist_rd, its history kernel fills the other one
        const size_t hb = 2 * A * 2 * 3 * 12512;
        HIPCHK(h, dmalloc(h, &h->d_hist_in, hb));
        HIPCHK(h, hipMemset(h->d_hist_in, 0, hb));
        HIPCHK(h, dmalloc(h, &h->d_hist_flags, 2 * A * 3 * PB_BLK_PER_FFT));
        HIPCHK(h, hipMemset(h->d_hist_flags, 1, 2 * A * 3 * PB_BLK_PER_FFT));
        HIPCHK(h, dmalloc(h, &h->d_hist_valid, 2 * A * 3));
        HIPCHK(h, hipMemset(h->d_hist_valid, 0, 2 * A * 3));
        h->hist_rd = 0;
        // window energy per (tap, 500-sample block), from the float taps the kernel multiplies by
        std::vector<float> taps((size_t)4 * PB_NFFT), E(101);
        HIPCHK(h, hipMemcpy(taps.data(), h->ft.taps, taps.size() * sizeof(float), hipMemcpyDeviceToHost));
        double tot = 0;
        for (int j = 0; j < 4; ++j)
            for (int b = 0; b < PB_BLK_PER_FFT; ++b) {
                double e = 0;
                for (int m = 0; m < PB_NKURTO; ++m) {
                    const double t = taps[(size_t)j * PB_NFFT + b * PB_NKURTO + m];
                    e += t * t;
                }
                E[j * PB_BLK_PER_FFT + b] = (float)e;
            }
        float totf = 0.f;
        for (int i = 0; i < 100; ++i) totf = totf + E[i];      // same order as k_pfb_weights sums
        (void)tot;
        E[100] = totf;
        HIPCHK(h, dmalloc(h, &h->d_tapE, (size_t)101));
        HIPCHK(h, hipMemcpy(h->d_tapE, E.data(), 101 * sizeof(float), hipMemcpyHostToDevice));
    }
    h->dag = make_dag((float)PB_NKURTO);
    h->dag_fb = make_dag((float)PB_NFFT);
    h->dag_bands = 0;
    if (env_int("PB_DAG_BANDS", 1) == 0) {
        // (test hook: the fallback below on purpose -- tests/test_gpu_parity.py checks that the flags stay bit-exact)
        h->dag_bands = -1;
    } else if (int rcb = find_dag_bands(h, h->dag)) {
        // The bands only save work (the score is then evaluated in a few floats around the crossings instead of
        // everywhere): if the search does not bracket the crossings -- host and device disagreeing by more than the
        // scanned window -- fall back to "no bands", where the score itself decides every flag: same flags, more
        // arithmetic.  That disagreement is worth knowing about on a library pinned bit for bit: it is said once on
        // stderr and kept in the handle (pb_debug_dag_check reports the bands as 0, inf, 0, inf).  A HIP error is fatal.
        if (rcb != PB_ESTATE) return rcb;
        fprintf(stderr, "libpb_hip: warning: %s -- flag decision falls back to evaluating the D'Agostino score for "
                        "every block (results unchanged, kurtosis statistic slower)\n", h->err.c_str());
        h->dag = make_dag((float)PB_NKURTO);
        h->dag_bands = -2;
        h->err.clear();
    } else {
        h->dag_bands = 1;
    }
    // (a copy in device memory for the channeliser that flags its own rows: ten fewer scalar registers of arguments)
    HIPCHK(h, hipMalloc((void **)&h->d_dag, sizeof(DagConsts)));
    HIPCHK(h, hipMemcpy(h->d_dag, &h->dag, sizeof(DagConsts), hipMemcpyHostToDevice));
    return PB_OK;
}

extern "C" int pb_create(const pb_config *cfg, pb_handle **out)
{
    if (!cfg || !out) return fail(nullptr, PB_EINVAL, "pb_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(pb_config)) return fail(nullptr, PB_EINVAL, "pb_create: pb_config size mismatch");
    const pb_config &c = *cfg;
    if (!(c.nbit == 2 || c.nbit == 4 || c.nbit == 8)) return fail(nullptr, PB_EINVAL, "Unsupported NBIT!");
    if (!(c.npol == 1 || c.npol == 2)) return fail(nullptr, PB_EINVAL, "Unsupported npol!");
    if (c.rfi_mode < 0 || c.rfi_mode > 2) return fail(nullptr, PB_EINVAL, "Unsupported RFI mode!");
    if (!(c.taps == 1 || c.taps == 4)) return fail(nullptr, PB_EINVAL, "taps must be 1 or 4");
    if (c.taps == 4 && c.fft_backend != PB_FFT_LDS) return fail(nullptr, PB_EINVAL, "taps=4 needs the LDS FFT back end");
    if (!(c.fft_backend == PB_FFT_LDS || c.fft_backend == PB_FFT_HIPFFT)) return fail(nullptr, PB_EINVAL, "bad fft_backend");
    if (c.nant < 1 || c.max_seg < 1) return fail(nullptr, PB_EINVAL, "nant and max_seg must be >= 1");
    if (c.nsets < 1 || c.nsets > 8) return fail(nullptr, PB_EINVAL, "nsets must be 1..8");
    if (c.rows_per_seg < 8 || c.rows_per_seg % PB_NSCRUNCH) return fail(nullptr, PB_EINVAL, "rows_per_seg must be a positive multiple of 8");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PB_EHIP, "pb_create: no HIP device (libpb_hip has no CPU fallback)");
    if (c.device < 0 || c.device >= ndev) return fail(nullptr, PB_EINVAL, "pb_create: bad device id");

    pb_handle *h = new pb_handle();
    h->cfg = c;
    h->fuse = env_int("PB_FUSE_KURTOSIS", 1);
    h->sched = pb_read_sched();
    h->R = c.rows_per_seg;
    h->S = c.max_seg;
    h->A = c.nant;
    h->seg_samples = (size_t)h->R * PB_NFFT;
    h->nblk_seg = (size_t)h->R * PB_BLK_PER_FFT;
    const int polfac = c.npol == 1 ? 2 : 1;
    h->trim = (size_t)2 * h->R * PB_NCHANOUT / (polfac * PB_NSCRUNCH) / (8 / c.nbit);
    h->ave_per_seg = (size_t)2 * h->R * PB_NCHANOUT / (polfac * PB_NSCRUNCH);
    h->stream = nullptr;
    h->own_stream = false;
    h->device_bytes = 0;
    h->d_in = h->d_vdif = h->d_flags = h->d_codes = nullptr;
    h->d_frame_idx = nullptr;
    h->vdif_cap = 0;
    for (int i = 0; i < 8; ++i) { h->h_frame_idx[i] = nullptr; h->h_idx_cap[i] = 0; h->ev_idx[i] = nullptr; }
    h->d_wrow = h->d_stats = h->d_fraw = h->d_fkur = h->d_Praw = h->d_Pkur = h->d_bp = h->d_ave = nullptr;
    h->d_frb_delays = nullptr;
    h->d_hist_in = h->d_hist_flags = h->d_hist_valid = nullptr;
    h->hist_rd = 0;
    h->d_tapE = nullptr;
    h->d_dag = nullptr;
    h->frb_width = 0.f;
    h->frb_amp = 1.f;
    h->d_Xraw = h->d_Xkur = nullptr;
    h->h_codes = nullptr;
    h->ev_chan = h->ev_det = h->ev_cl = nullptr;
    h->s_coadd = nullptr;
    h->processed = 0;
    h->cur_set = 0;
    h->s_det = nullptr;
    h->s_copy = nullptr;
    h->s_kur = nullptr;
    h->ev_fftdone = h->ev_kur = h->ev_alldone = h->ev_hist = nullptr;
    h->last_set = -1;
    h->staged = false;
    h->d_coadd_target = nullptr;
    h->d_coadd_codes = h->h_coadd_codes = nullptr;
    h->ev_coadd[0] = h->ev_coadd[1] = nullptr;
    h->coadd_slot = h->coadd_last = 0;
    memset(&h->ft, 0, sizeof h->ft);
    h->profile = false;
    memset(&h->timers, 0, sizeof h->timers);
    int rc = create_impl(h);
    if (rc != PB_OK) {
        g_create_err = h->err;
        pb_destroy(h);
        return rc;
    }
    *out = h;
    return PB_OK;
}

extern "C" void pb_destroy(pb_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->s_det) (void)hipStreamSynchronize(h->s_det);
    if (h->s_copy) (void)hipStreamSynchronize(h->s_copy);
    for (auto &kv : h->plans) hipfftDestroy(kv.second);
    bool stored = false;
    for (auto &b : h->sets) {
        if (b.d_in == h->d_in) stored = true;
        void *sp[] = {b.d_in, b.d_flags, b.d_codes, b.d_wrow, b.d_stats, b.d_fraw, b.d_fkur,
                      b.d_Praw, b.d_Pkur, b.d_ave, b.d_Xraw, b.d_Xkur};
        for (void *p : sp)
            if (p) (void)hipFree(p);
        if (b.h_codes) (void)hipHostFree(b.h_codes);
        if (b.ev_chan) (void)hipEventDestroy(b.ev_chan);
        if (b.ev_det) (void)hipEventDestroy(b.ev_det);
        if (b.ev_cl) (void)hipEventDestroy(b.ev_cl);
    }
    if (!stored) {  // pb_create failed half-way through a set: free the loose members
        void *sp[] = {h->d_in, h->d_flags, h->d_codes, h->d_wrow, h->d_stats, h->d_fraw, h->d_fkur,
                      h->d_Praw, h->d_Pkur, h->d_ave, h->d_Xraw, h->d_Xkur};
        for (void *p : sp)
            if (p) (void)hipFree(p);
        if (h->h_codes) (void)hipHostFree(h->h_codes);
    }
    void *ptrs[] = {h->d_vdif, h->d_frame_idx, h->d_bp, h->d_frb_delays, h->d_hist_in, h->d_hist_flags,
                    h->d_hist_valid, h->d_tapE, h->d_dag, h->ft.tw2,
                    h->ft.tw3, h->ft.post, h->ft.postc, h->ft.taps, h->ft.taps_n};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (int i = 0; i < 8; ++i) {
        if (h->h_frame_idx[i]) (void)hipHostFree(h->h_frame_idx[i]);
        if (h->ev_idx[i]) (void)hipEventDestroy(h->ev_idx[i]);
    }
    if (h->d_coadd_codes) (void)hipFree(h->d_coadd_codes);
    if (h->h_coadd_codes) (void)hipHostFree(h->h_coadd_codes);
    for (int i = 0; i < 2; ++i)
        if (h->ev_coadd[i]) (void)hipEventDestroy(h->ev_coadd[i]);
    if (h->s_det) (void)hipStreamDestroy(h->s_det);
    if (h->s_copy) (void)hipStreamDestroy(h->s_copy);
    if (h->s_kur) { (void)hipStreamSynchronize(h->s_kur); (void)hipStreamDestroy(h->s_kur); }
    if (h->ev_fftdone) (void)hipEventDestroy(h->ev_fftdone);
    if (h->ev_hist) (void)hipEventDestroy(h->ev_hist);
    if (h->ev_kur) (void)hipEventDestroy(h->ev_kur);
    if (h->ev_alldone) (void)hipEventDestroy(h->ev_alldone);
    drain_timers(h);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" const char *pb_last_error(const pb_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int pb_query(const pb_handle *h, pb_sizes *o)
{
    if (!h || !o) return PB_EINVAL;
    o->seg_samples_per_pol = h->seg_samples;
    o->input_bytes_per_seg = 2 * h->seg_samples;
    o->code_bytes_per_seg = h->trim;
    o->ave_floats_per_seg = h->ave_per_seg;
    o->rows_per_seg = h->R;
    o->blocks_per_seg_pol = h->nblk_seg;
    o->device_bytes = h->device_bytes;
    return PB_OK;
}

extern "C" int pb_set_stream(pb_handle *h, void *s)
{
    if (!h) return PB_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    if (h->own_stream) {
        HIPCHK(h, hipStreamDestroy(h->stream));
        h->own_stream = false;
    }
    if (s) {
        h->stream = (hipStream_t)s;
    } else {
        HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    for (auto &kv : h->plans) hipfftSetStream(kv.second, h->stream);
    return PB_OK;
}

extern "C" int pb_sync(pb_handle *h)
{
    if (!h) return PB_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    return PB_OK;
}

// Stream that input staging goes to.  With >= 2 buffer sets the next batch is staged while the
// previous one computes: staging and the kurtosis pass share s_kur (ordered), the channeliser on the
// main stream waits for the kurtosis event.  Refilling the set that was processed last waits for
// that whole batch.
static hipError_t submit_stream(pb_handle *h, hipStream_t *out)
{
    if (h->sets.size() < 2) { *out = h->stream; return hipSuccess; }
    *out = h->s_kur;
    h->staged = true;          // the next pb_process orders its first kernel behind s_kur
    if (h->last_set == h->cur_set) {
        // refilling the set that was processed last: behind that whole batch -- detect / copy-out (ev_alldone); with
        // taps = 4 the history kernel, which reads the last three rows of d_in, ran on this very stream or before
        // the channeliser that detect followed
        return hipStreamWaitEvent(h->s_kur, h->ev_alldone, 0);
    }
    // hipFFT back end: its kurtosis pass (the last reader of d_in) runs on the MAIN stream, which s_kur is
    // not ordered behind: staging batch k+2 into this set must wait for the latest FFT stage, which is
    // queued behind every earlier reader of this set's input.
    if (h->cfg.fft_backend == PB_FFT_HIPFFT && h->last_set >= 0) return hipStreamWaitEvent(h->s_kur, h->ev_fftdone, 0);
    // In-library FFT: s_kur is not in general ordered behind the channeliser that read this set's input last, on
    // the main stream (the channeliser that flags its own rows leaves s_kur empty; with three or more sets the
    // kurtosis pass of the two-kernel path no longer waits for the previous channeliser either).  This set's
    // ev_chan -- its detect is done, queued behind that channeliser and, with taps = 4, behind the history kernel
    // that read the set's last rows -- does, and it completed while the batches in between were being channelised.
    if (h->processed > 0) return hipStreamWaitEvent(h->s_kur, h->ev_chan, 0);
    return hipSuccess;
}

static int check_ant(pb_handle *h, int ant)
{
    if (ant < 0 || ant >= h->A) return fail(h, PB_EINVAL, "antenna index out of range");
    return PB_OK;
}

extern "C" int pb_reset_bandpass(pb_handle *h, int ant)
{
    if (!h) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // the last detect (which may be on the second stream) still owns the bandpass; the next one is
    // ordered behind this stream by its channeliser's event
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_alldone, 0));
    HIPCHK(h, hipMemsetAsync(h->d_bp + (size_t)ant * 4 * PB_NCHANOUT, 0, 4 * PB_NCHANOUT * sizeof(float), h->stream));
    return PB_OK;
}

extern "C" int pb_reset_history(pb_handle *h, int ant)
{
    if (!h) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (h->cfg.taps != 4) return PB_OK;           // only the PFB window carries rows across calls
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    for (int slot = 0; slot < 2; ++slot) {
        const size_t a = (size_t)slot * h->A + ant;
        HIPCHK(h, hipMemset(h->d_hist_in + a * 2 * 3 * 12512, 0, (size_t)2 * 3 * 12512));
        HIPCHK(h, hipMemset(h->d_hist_flags + a * 3 * PB_BLK_PER_FFT, 1, 3 * PB_BLK_PER_FFT));
        HIPCHK(h, hipMemset(h->d_hist_valid + a * 3, 0, 3));
    }
    HIPCHK(h, hipStreamSynchronize(nullptr));     // (the handle's streams do not wait for the null stream by themselves)
    return PB_OK;
}

extern "C" int pb_get_bandpass(pb_handle *h, int ant, float *raw, float *kur)
{
    if (!h) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    const float *b = h->d_bp + (size_t)ant * 4 * PB_NCHANOUT;
    if (raw) HIPCHK(h, hipMemcpy(raw, b, 2 * PB_NCHANOUT * sizeof(float), hipMemcpyDeviceToHost));
    if (kur) HIPCHK(h, hipMemcpy(kur, b + 2 * PB_NCHANOUT, 2 * PB_NCHANOUT * sizeof(float), hipMemcpyDeviceToHost));
    return PB_OK;
}

extern "C" int pb_set_bandpass(pb_handle *h, int ant, const float *raw, const float *kur)
{
    if (!h) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    float *b = h->d_bp + (size_t)ant * 4 * PB_NCHANOUT;
    if (raw) HIPCHK(h, hipMemcpy(b, raw, 2 * PB_NCHANOUT * sizeof(float), hipMemcpyHostToDevice));
    if (kur) HIPCHK(h, hipMemcpy(b + 2 * PB_NCHANOUT, kur, 2 * PB_NCHANOUT * sizeof(float), hipMemcpyHostToDevice));
    return PB_OK;
}

extern "C" int pb_submit_planar(pb_handle *h, int ant, int seg, const uint8_t *pol0, const uint8_t *pol1, size_t nsamp)
{
    if (!h || !pol0 || !pol1) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (seg < 0 || seg >= h->S) return fail(h, PB_EINVAL, "segment slot out of range");
    if (nsamp != h->seg_samples) return fail(h, PB_EINVAL, "pb_submit_planar: nsamp must equal seg_samples_per_pol");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    uint8_t *dst = h->d_in + (((size_t)ant * h->S + seg) * 2) * h->seg_samples;
    hipStream_t ss;
    HIPCHK(h, submit_stream(h, &ss));
    HIPCHK(h, hipMemcpyAsync(dst, pol0, nsamp, hipMemcpyHostToDevice, ss));
    HIPCHK(h, hipMemcpyAsync(dst + h->seg_samples, pol1, nsamp, hipMemcpyHostToDevice, ss));
    return PB_OK;
}

extern "C" int pb_submit_planar_dev(pb_handle *h, int ant, int seg, const void *pol0, const void *pol1, size_t nsamp)
{
    if (!h || !pol0 || !pol1) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (seg < 0 || seg >= h->S) return fail(h, PB_EINVAL, "segment slot out of range");
    if (nsamp != h->seg_samples) return fail(h, PB_EINVAL, "pb_submit_planar_dev: nsamp must equal seg_samples_per_pol");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    uint8_t *dst = h->d_in + (((size_t)ant * h->S + seg) * 2) * h->seg_samples;
    hipStream_t ss;
    HIPCHK(h, submit_stream(h, &ss));
    HIPCHK(h, hipMemcpyAsync(dst, pol0, nsamp, hipMemcpyDeviceToDevice, ss));
    HIPCHK(h, hipMemcpyAsync(dst + h->seg_samples, pol1, nsamp, hipMemcpyDeviceToDevice, ss));
    return PB_OK;
}

extern "C" int pb_submit_vdif_at(pb_handle *h, int ant, int seg0, const uint8_t *block, size_t nbytes,
                                 int64_t sec0, int64_t fr0)
{
    if (!h || !block) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (nbytes == 0 || nbytes % (2 * PB_VDIF_FRAME)) return fail(h, PB_EINVAL, "pb_submit_vdif: block is not a whole number of frame pairs");
    const size_t nslots = nbytes / PB_VDIF_FRAME, nfr = nslots / 2;
    if ((nfr * PB_VDIF_DATA) % h->seg_samples) return fail(h, PB_EINVAL, "pb_submit_vdif: block does not fill whole segments");
    const size_t nsegs = nfr * PB_VDIF_DATA / h->seg_samples;
    if (seg0 < 0 || seg0 + nsegs > (size_t)h->S) return fail(h, PB_EINVAL, "pb_submit_vdif: segments out of range");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int set = h->cur_set;
    // The frame index lives in page-locked memory of its own per buffer set, so that its H2D copy -- like
    // the block's -- is asynchronous and this call returns without waiting for the device (it used to
    // synchronise the stream because the index was a stack vector).  The event says the previous copy out
    // of this buffer has completed.
    if (h->h_idx_cap[set] < 2 * nfr) {
        if (h->h_frame_idx[set]) {
            HIPCHK(h, hipEventSynchronize(h->ev_idx[set]));
            (void)hipHostFree(h->h_frame_idx[set]);
            h->h_frame_idx[set] = nullptr;
        }
        HIPCHK(h, hipHostMalloc((void **)&h->h_frame_idx[set], 2 * nfr * sizeof(int32_t), hipHostMallocDefault));
        h->h_idx_cap[set] = 2 * nfr;
        if (!h->ev_idx[set]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_idx[set], hipEventDisableTiming));
    } else {
        HIPCHK(h, hipEventSynchronize(h->ev_idx[set]));
    }
    // index the headers: word0 bits 0-29 seconds, word1 bits 0-23 frame, word3 bits 16-25 thread
    int32_t *idx = h->h_frame_idx[set];
    for (size_t i = 0; i < 2 * nfr; ++i) idx[i] = -1;
    uint32_t w0, w1, w3;
    for (size_t i = 0; i < nslots; ++i) {
        const uint8_t *p = block + i * PB_VDIF_FRAME;
        memcpy(&w0, p, 4);
        memcpy(&w1, p + 4, 4);
        memcpy(&w3, p + 12, 4);
        if (w0 & 0x80000000u) continue;  // invalid-data bit
        const int thread = ((w3 >> 16) & 0x3ff) != 0;
        const int64_t rel = ((int64_t)(w0 & 0x3fffffff) - sec0) * PB_FRAMES_PER_SEC + (int64_t)(w1 & 0xffffff) - fr0;
        if (rel < 0 || rel >= (int64_t)nfr) continue;
        idx[(size_t)thread * nfr + rel] = (int32_t)i;
    }
    if (h->vdif_cap < nbytes) {
        HIPCHK(h, sync_all(h));
        if (h->d_vdif) { (void)hipFree(h->d_vdif); h->device_bytes -= h->vdif_cap; }
        if (h->d_frame_idx) (void)hipFree(h->d_frame_idx);
        h->d_vdif = nullptr;
        h->d_frame_idx = nullptr;
        HIPCHK(h, dmalloc(h, &h->d_vdif, nbytes));
        HIPCHK(h, hipMalloc((void **)&h->d_frame_idx, 2 * nfr * sizeof(int32_t)));
        h->vdif_cap = nbytes;
    }
    // d_vdif / d_frame_idx are shared by the buffer sets: every use of them is on the staging stream, in order
    hipStream_t ss;
    HIPCHK(h, submit_stream(h, &ss));
    HIPCHK(h, hipMemcpyAsync(h->d_vdif, block, nbytes, hipMemcpyHostToDevice, ss));
    HIPCHK(h, hipMemcpyAsync(h->d_frame_idx, idx, 2 * nfr * sizeof(int32_t), hipMemcpyHostToDevice, ss));
    HIPCHK(h, hipEventRecord(h->ev_idx[set], ss));
    {
        hipStream_t s_main = h->stream;
        h->stream = ss;
        hipError_t e = launch_deframe(h, ant, seg0, nfr);
        h->stream = s_main;
        HIPCHK(h, e);
    }
    return PB_OK;
}

extern "C" int pb_submit_vdif(pb_handle *h, int ant, int seg0, const uint8_t *block, size_t nbytes)
{
    if (!h || !block || nbytes < PB_VDIF_FRAME) return PB_EINVAL;
    uint32_t w0, w1;
    memcpy(&w0, block, 4);
    memcpy(&w1, block + 4, 4);
    return pb_submit_vdif_at(h, ant, seg0, block, nbytes, (int64_t)(w0 & 0x3fffffff), (int64_t)(w1 & 0xffffff));
}

extern "C" int pb_input_dev(pb_handle *h, int ant, void **dptr, size_t *nbytes)
{
    if (!h || !dptr) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    *dptr = h->d_in + (size_t)ant * h->S * 2 * h->seg_samples;
    if (nbytes) *nbytes = (size_t)h->S * 2 * h->seg_samples;
    return PB_OK;
}

// Per-stage device time without disturbing the stream: each stage records a pair of events
// from a pool; pb_get_timers synchronises once and reads them all back.
static hipEvent_t take_event(pb_handle *h)
{
    if (!h->ev_pool.empty()) {
        hipEvent_t e = h->ev_pool.back();
        h->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Read back the oldest event pairs that have completed, without waiting for anything: keeps the
// pool small, so that a long profiled run creates no events (hipEventCreate costs tens of us).
static void retire_done(pb_handle *h)
{
    size_t n = 0;
    while (n < h->pending.size() && hipEventQuery(h->pending[n].b) == hipSuccess) {
        const auto &p = h->pending[n];
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->timers.ms[p.stage] += ms;
            h->timers.launches[p.stage] += 1;
        }
        h->ev_pool.push_back(p.a);
        h->ev_pool.push_back(p.b);
        ++n;
    }
    if (n) h->pending.erase(h->pending.begin(), h->pending.begin() + n);
}

struct StageTimer {
    pb_handle *h;
    int stage;
    hipEvent_t a, b;
    StageTimer(pb_handle *h_, int s) : h(h_), stage(s), a(nullptr), b(nullptr)
    {
        if (h->profile) {
            if (h->ev_pool.size() < 2) retire_done(h);
            a = take_event(h);
            b = take_event(h);
            (void)hipEventRecord(a, h->stream);
        }
    }
    void stop()
    {
        if (!h->profile) return;
        (void)hipEventRecord(b, h->stream);
        h->pending.push_back({stage, a, b});
    }
};

static void drain_timers(pb_handle *h)
{
    if (h->pending.empty()) return;
    (void)sync_all(h);
    for (auto &p : h->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->timers.ms[p.stage] += ms;
            h->timers.launches[p.stage] += 1;
        }
        h->ev_pool.push_back(p.a);
        h->ev_pool.push_back(p.b);
    }
    h->pending.clear();
}

static int exec_fft(pb_handle *h, int nseg)
{
    const long batch = (long)nseg * 2 * h->R;
    hipfftHandle plan;
    auto it = h->plans.find(batch);
    if (it == h->plans.end()) {
        int n[1] = {PB_NFFT};
        hipfftResult r = hipfftPlanMany(&plan, 1, n, nullptr, 1, PB_NFFT, nullptr, 1, PB_NCHAN, HIPFFT_R2C, (int)batch);
        if (r != HIPFFT_SUCCESS) return fail(h, PB_EHIP, "hipfftPlanMany failed: " + std::to_string((int)r));
        r = hipfftSetStream(plan, h->stream);
        if (r != HIPFFT_SUCCESS) return fail(h, PB_EHIP, "hipfftSetStream failed");
        h->plans[batch] = plan;
    } else {
        plan = it->second;
    }
    const size_t in_ant = (size_t)h->S * 2 * h->seg_samples;
    const size_t x_ant = (size_t)h->S * 2 * h->R * PB_NCHAN;
    for (int a = 0; a < h->A; ++a) {
        if (h->cfg.rfi_mode != 1) {
            hipfftResult r = hipfftExecR2C(plan, h->d_fraw + a * in_ant, (hipfftComplex *)(h->d_Xraw + a * x_ant));
            if (r != HIPFFT_SUCCESS) return fail(h, PB_EHIP, "hipfftExecR2C failed: " + std::to_string((int)r));
        }
        if (h->cfg.rfi_mode != 0) {
            hipfftResult r = hipfftExecR2C(plan, h->d_fkur + a * in_ant, (hipfftComplex *)(h->d_Xkur + a * x_ant));
            if (r != HIPFFT_SUCCESS) return fail(h, PB_EHIP, "hipfftExecR2C failed: " + std::to_string((int)r));
        }
    }
    return PB_OK;
}

// Every plane the configured RFI mode / FFT back end launches kernels on exists in the selected buffer set: a
// launcher never dereferences a stream's buffers that pb_create did not allocate (modes 0 / 1 have one stream).
static int check_planes(pb_handle *h)
{
    const bool hipfft = h->cfg.fft_backend == PB_FFT_HIPFFT;
    const bool need_raw = h->cfg.rfi_mode != 1, need_kur = h->cfg.rfi_mode != 0;
    bool ok = h->d_in && h->d_flags && h->d_wrow && h->d_codes && h->h_codes && h->d_bp;
    if (need_raw) ok = ok && h->d_Praw && (!hipfft || (h->d_fraw && h->d_Xraw));
    if (need_kur) ok = ok && h->d_Pkur && (!hipfft || (h->d_fkur && h->d_Xkur));
    if (h->cfg.keep_ave) ok = ok && h->d_ave;
    if (h->cfg.taps == 4) ok = ok && h->d_hist_in && h->d_hist_flags && h->d_hist_valid && h->d_tapE;
    return ok ? PB_OK : fail(h, PB_ESTATE, "pb_process: a buffer of the selected set is missing for this RFI mode / back end");
}

extern "C" int pb_process(pb_handle *h, int nseg, int inject_now)
{
    if (!h) return PB_EINVAL;
    if (nseg < 1 || nseg > h->S) return fail(h, PB_EINVAL, "pb_process: nseg out of range");
    if (int rc = check_planes(h)) return rc;
    if (!h->cfg.inject_frb) inject_now = 0;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const bool hipfft = h->cfg.fft_backend == PB_FFT_HIPFFT;
    // What this batch's first kernels overwrite -- the set's flags, weights and power planes -- was last read by the
    // set's previous detect (ev_chan).  The copy-out of that batch's bytes (ev_det) only has to have drained before
    // THIS batch's detect overwrites the code buffer: waiting for it here held the channeliser back by ~0.1 ms per
    // step once the kurtosis pass in front of it was gone (dispatch timeline, profiles/r03_notes.md).
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_chan, 0));
    {
        // The kurtosis pass only needs the raw bytes, and it is a light memory/LDS kernel while
        // detect of the PREVIOUS batch is a latency-bound one that leaves the CUs mostly idle: when
        // the previous batch used another buffer set, run kurtosis on its own stream, released as
        // soon as the previous channeliser has finished, i.e. beside the previous detect.
        const bool fused = pb_fused_kurtosis(h);     // the channeliser flags its own rows: nothing to launch here
        const bool overlap = !fused && !hipfft && h->sets.size() >= 2 && h->last_set >= 0 && h->last_set != h->cur_set;
        hipStream_t s_main = h->stream;
        hipError_t e = hipSuccess;
        if (overlap) {
            // Three or more buffer sets: the kurtosis pass of this batch touches nothing the previous batch's
            // channeliser uses (another set; the PFB history kernel behind it waits for that channeliser itself),
            // so it does not wait for that channeliser and runs beside it -- taps = 4: 0.990 -> 0.965 ms per second
            // of data, the two-kernel taps = 1 path 0.660 -> 0.623 (same box, alternating).  With two sets this
            // set's flags and weights are still being read by the previous batch's channeliser's successor, detect,
            // until ev_chan; the wait for the channeliser keeps the kurtosis pass beside that detect.
            // PB_KUR_EARLY=0 restores the wait.
            if (!(h->sched.kur_early && h->sets.size() >= 3)) e = hipStreamWaitEvent(h->s_kur, h->ev_fftdone, 0);
            if (e == hipSuccess) e = hipStreamWaitEvent(h->s_kur, h->ev_chan, 0);  // flags / weights of this set free (its detect is done)
            h->stream = h->s_kur;
        } else if (h->sets.size() >= 2 && (h->staged || !fused)) {
            // staging went to s_kur: the main stream must see it.  (Nothing staged since the last call -- the
            // caller writes the input buffers itself, pb_input_dev -- and nothing queued on s_kur: no event pair,
            // a cross-stream dependency costs the command processor ~10-25 us even when it is already met.)
            e = hipEventRecord(h->ev_kur, h->s_kur);
            if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, h->ev_kur, 0);
        }
        h->staged = false;
        if (e == hipSuccess && !fused) {
            {
                StageTimer t(h, PB_ST_KURTOSIS);
                // (PB_SKIP & 4, PB_EXPERIMENTS builds only: only once the set holds the flags of the same input from an
                //  earlier call -- the experiment re-processes one second of data, and without flags there would be no
                //  excised transforms either)
                if (!((pb_skip_mask() & 4) && h->processed > 0)) e = launch_kurtosis_flag(h, nseg, hipfft);
                t.stop();
            }
            // taps = 4: the weights read the flags of the previous batch's last rows (history slot hist_rd, filled by
            // that batch's history kernel earlier on this stream).
            // ... or on the main stream, when that batch was not on the overlap path (the first batch, a batch that
            // repeated its set): ev_hist, recorded behind every history kernel on whichever stream ran it, orders the
            // weights behind it either way (free when both are on this stream).
            if (e == hipSuccess && h->cfg.taps == 4) e = hipStreamWaitEvent(h->stream, h->ev_hist, 0);
            if (e == hipSuccess) e = launch_pfb_weights(h, nseg);
            if (e == hipSuccess && h->cfg.taps == 4) {
                // The channeliser only needs what has been queued up to here: release it now.  Then keep this
                // batch's last three rows and flags for the next one in the OTHER history slot -- the one the
                // previous batch's channeliser reads, hence behind it (ev_fftdone still holds its record); the next
                // batch's weights and staging follow on this stream, its channeliser behind them.  (Round 2 ran the
                // history kernel behind the channeliser on the main stream and the next weights behind THAT: two
                // cross-stream hand-overs between consecutive channelisers.)
                if (overlap) e = hipEventRecord(h->ev_kur, h->s_kur);
                if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, h->ev_fftdone, 0);
                if (e == hipSuccess) e = launch_pfb_history(h, nseg);
                if (e == hipSuccess) e = hipEventRecord(h->ev_hist, h->stream);
            }
        }
        if (overlap) {
            if (e == hipSuccess && h->cfg.taps != 4) e = hipEventRecord(h->ev_kur, h->s_kur);
            h->stream = s_main;
            if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, h->ev_kur, 0);
        }
        HIPCHK(h, e);
    }
    if (hipfft) {
        StageTimer t(h, PB_ST_FFT);
        int rc = exec_fft(h, nseg);
        if (rc) return rc;
        t.stop();
        if (inject_now > 0) {
            StageTimer ti(h, PB_ST_INJECT);
            HIPCHK(h, launch_inject_c64(h, nseg, inject_now));
            ti.stop();
        }
    } else {
        StageTimer t(h, PB_ST_CHANNELIZE);
        if (!(pb_skip_mask() & 1))
            HIPCHK(h, h->cfg.taps == 4 ? launch_channelize_pfb(h, nseg, inject_now) : launch_channelize(h, nseg, inject_now));
        t.stop();
    }
    HIPCHK(h, hipEventRecord(h->ev_fftdone, h->stream));
    if (h->cfg.taps == 4) h->hist_rd ^= 1;       // the next batch reads what this batch's history kernel has kept
    h->last_set = h->cur_set;
    // With two or more buffer sets, detect + D2H of this batch run on the second stream so that they
    // overlap the NEXT batch's kurtosis and channeliser.  Detect is latency-bound on its serial
    // bandpass wave (one 48 KB workgroup per CU), the channeliser VALU-bound: side by side the step is
    // ~9 % shorter than back to back (0.915 vs 1.007 ms per second of data; PB_OVERLAP_DETECT=0 turns
    // it off).  With one buffer set everything stays on one stream.
    if (h->sched.overlap_detect && h->sets.size() >= 2 && !hipfft) {
        hipStream_t s_main = h->stream;
        hipError_t e2 = hipStreamWaitEvent(h->s_det, h->ev_fftdone, 0);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(h->s_det, h->ev_det, 0);  // this set's previous bytes have left
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(h->s_det, h->ev_cl, 0);   // planes of this set were summed
        h->stream = h->s_det;
        if (e2 == hipSuccess) {
            StageTimer t(h, PB_ST_DETECT);
            if (!(pb_skip_mask() & 2)) e2 = launch_detect(h, nseg, inject_now);
            t.stop();
        }
        if (e2 == hipSuccess) e2 = hipEventRecord(h->ev_chan, h->s_det);   // this set's kernels are done
        // the copy-out has a stream of its own: behind detect on s_det it would hold up the next batch's
        // detect whenever it is slow (many antennas, few copy workgroups)
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(h->s_copy, h->ev_chan, 0);
        for (int a = 0; a < h->A && e2 == hipSuccess; ++a)
            for (int st = 0; st < 2 && e2 == hipSuccess; ++st) {
                if (st == 0 ? h->cfg.rfi_mode == 1 : h->cfg.rfi_mode == 0) continue;
                const size_t o = ((size_t)a * 2 + st) * h->S * h->trim;
                e2 = launch_copy_out(h->sched, h->h_codes + o, h->d_codes + o, (size_t)nseg * h->trim, h->s_copy);
            }
        if (e2 == hipSuccess) e2 = hipEventRecord(h->ev_det, h->s_copy);
        if (e2 == hipSuccess) e2 = hipEventRecord(h->ev_alldone, h->s_copy);
        h->stream = s_main;
        HIPCHK(h, e2);
        h->processed = nseg;
        h->sets[h->cur_set].processed = nseg;
        return PB_OK;
    }
    {
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_det, 0));  // this set's previous bytes have left
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_cl, 0));   // planes of this set were summed
        StageTimer t(h, PB_ST_DETECT);
        HIPCHK(h, launch_detect(h, nseg, inject_now));
        t.stop();
    }
    HIPCHK(h, hipEventRecord(h->ev_chan, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_alldone, h->stream));
    hipError_t e = hipStreamWaitEvent(h->s_det, h->ev_chan, 0);
    for (int a = 0; a < h->A && e == hipSuccess; ++a)
        for (int st = 0; st < 2 && e == hipSuccess; ++st) {
            if (st == 0 ? h->cfg.rfi_mode == 1 : h->cfg.rfi_mode == 0) continue;
            const size_t o = ((size_t)a * 2 + st) * h->S * h->trim;
            e = launch_copy_out(h->sched, h->h_codes + o, h->d_codes + o, (size_t)nseg * h->trim, h->s_det);
        }
    if (e == hipSuccess) e = hipEventRecord(h->ev_det, h->s_det);
    HIPCHK(h, e);
    h->processed = nseg;
    h->sets[h->cur_set].processed = nseg;
    return PB_OK;
}

extern "C" int pb_set_frb_params(pb_handle *h, float dm, float width_rows, float amp)
{
    if (!h) return PB_EINVAL;
    if (!h->cfg.inject_frb) return fail(h, PB_ESTATE, "pb_set_frb_params needs inject_frb=1");
    if (!(dm >= 0) || !(amp > 0)) return fail(h, PB_EINVAL, "pb_set_frb_params: bad dm or amp");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    return set_frb(h, dm, width_rows, amp);
}

extern "C" int pb_select_set(pb_handle *h, int set)
{
    if (!h) return PB_EINVAL;
    if (set < 0 || set >= (int)h->sets.size()) return fail(h, PB_EINVAL, "pb_select_set: no such buffer set");
    store_set(h, h->cur_set);
    load_set(h, set);
    return PB_OK;
}

extern "C" int pb_fetch_ptr(pb_handle *h, int ant, int stream, const uint8_t **codes)
{
    if (!h || !codes) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (stream < 0 || stream > 1) return fail(h, PB_EINVAL, "stream must be 0 or 1");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipEventSynchronize(h->ev_det));
    *codes = h->h_codes + ((size_t)ant * 2 + stream) * h->S * h->trim;
    return PB_OK;
}

extern "C" int pb_fetch(pb_handle *h, int ant, int seg0, int nseg, uint8_t *raw_codes, uint8_t *kur_codes,
                        float *weights, float *ave_raw, float *ave_kur)
{
    if (!h) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (seg0 < 0 || nseg < 1 || seg0 + nseg > h->S) return fail(h, PB_EINVAL, "pb_fetch: segment range");
    if ((ave_raw || ave_kur) && !h->cfg.keep_ave) return fail(h, PB_ESTATE, "pb_fetch: fp32 planes need keep_ave=1");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // the selected set's detect + asynchronous D2H (pinned mirror h_codes) have an event of
    // their own, so fetching one set does not wait for work in flight on the other
    HIPCHK(h, hipEventSynchronize(h->ev_det));
    const size_t S = h->S;
    const uint8_t *c0 = h->h_codes + (((size_t)ant * 2 + 0) * S + seg0) * h->trim;
    const uint8_t *c1 = h->h_codes + (((size_t)ant * 2 + 1) * S + seg0) * h->trim;
    if (raw_codes) memcpy(raw_codes, c0, (size_t)nseg * h->trim);
    if (kur_codes) memcpy(kur_codes, c1, (size_t)nseg * h->trim);
    if (ave_raw) HIPCHK(h, hipMemcpy(ave_raw, h->d_ave + (((size_t)ant * 2 + 0) * S + seg0) * h->ave_per_seg,
                                    (size_t)nseg * h->ave_per_seg * sizeof(float), hipMemcpyDeviceToHost));
    const int cstream = h->cfg.rfi_mode == 0 ? 0 : 1;     // the stream whose plane a coadd target replaces
    if (h->d_coadd_target && ant == 0 && ((cstream == 0 && ave_raw) || (cstream == 1 && ave_kur))) {
        HIPCHK(h, hipMemcpy(cstream ? ave_kur : ave_raw, h->d_coadd_target + (size_t)seg0 * h->ave_per_seg,
                            (size_t)nseg * h->ave_per_seg * sizeof(float), hipMemcpyDeviceToHost));
        if (cstream) ave_kur = nullptr; else ave_raw = nullptr;
    }
    if (ave_kur) HIPCHK(h, hipMemcpy(ave_kur, h->d_ave + (((size_t)ant * 2 + 1) * S + seg0) * h->ave_per_seg,
                                    (size_t)nseg * h->ave_per_seg * sizeof(float), hipMemcpyDeviceToHost));
    if (weights) {
        // what tscrunch_weights sees: after pscrunch_weights (npol 1) rows below MIN_WEIGHT are 0
        std::vector<float> w((size_t)nseg * h->R);
        HIPCHK(h, hipMemcpy(w.data(), h->d_wrow + (size_t)ant * S * h->R + (size_t)seg0 * h->R,
                            w.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < w.size(); ++i) {
            float v = w[i];
            if (h->cfg.npol == 1) v = ((double)v >= 0.2) ? (float)(0.5 * (double)(v + v)) : 0.f;
            weights[i] = v;
        }
    }
    return PB_OK;
}

extern "C" int pb_output_dev(pb_handle *h, int ant, int stream, void **codes, void **ave)
{
    if (!h) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (stream < 0 || stream > 1) return fail(h, PB_EINVAL, "stream must be 0 or 1");
    if (codes) *codes = h->d_codes + ((size_t)ant * 2 + stream) * h->S * h->trim;
    if (ave) {
        *ave = h->d_ave ? h->d_ave + ((size_t)ant * 2 + stream) * h->S * h->ave_per_seg : nullptr;
        if (h->d_coadd_target && ant == 0 && stream == (h->cfg.rfi_mode == 0 ? 0 : 1)) *ave = h->d_coadd_target;
    }
    return PB_OK;
}

extern "C" int pb_set_coadd_stream(pb_handle *h, void *stream)
{
    if (!h) return PB_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    h->s_coadd = (hipStream_t)stream;
    return PB_OK;
}

extern "C" int pb_coadd_local(pb_handle *h, int nseg, float *d_sum, int accumulate)
{
    if (!h || !d_sum) return PB_EINVAL;
    if (!h->cfg.keep_ave) return fail(h, PB_ESTATE, "pb_coadd_local needs keep_ave=1");
    if (nseg < 1 || nseg > h->S) return fail(h, PB_EINVAL, "pb_coadd_local: nseg out of range");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // The fp32 planes come from this set's detect, which may have run on the second stream.  On a
    // coadd stream of its own (pb_set_coadd_stream) this wait holds up neither the next batch's
    // channeliser nor its detect.
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    // The planes come from this set's detect, which runs on another stream.  When the host already knows that it
    // has finished (e.g. the batch's bytes have been fetched), no device-side wait is queued: a pending
    // cross-stream wait on that event was measured to cost the whole pipeline 0.12 ms per second of data.
    if (hipEventQuery(h->ev_chan) != hipSuccess) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_chan, 0));
    if (h->d_coadd_target && d_sum == h->d_coadd_target && !accumulate) {
        // nant = 1 with a coadd target: detect wrote the plane into d_sum itself.  Nothing to launch; the
        // "planes consumed" event is recorded by pb_coadd_release, after the caller's collective.
        if (nseg > h->processed) return fail(h, PB_EINVAL, "pb_coadd_local: more segments than the batch holds");
        return PB_OK;
    }
    hipStream_t s_main = h->stream;
    h->stream = cs;                       // (StageTimer records on h->stream)
    hipError_t e;
    {
        StageTimer t(h, PB_ST_COADD);
        e = launch_coadd_local(h, nseg, d_sum, accumulate, cs);
        t.stop();
    }
    if (e == hipSuccess) e = hipEventRecord(h->ev_cl, cs);
    h->stream = s_main;
    HIPCHK(h, e);
    return PB_OK;
}

// The same local sum taken from this batch's quantised codes (coadd_codes.hip): what a coadder fed from the co
// rings has to work with.  Needs no fp32 planes (keep_ave may be 0).
hipError_t launch_coadd_local_codes(pb_handle *h, int nseg, float *d_sum, int accumulate, hipStream_t st);
extern "C" int pb_coadd_local_codes(pb_handle *h, int nseg, float *d_sum, int accumulate)
{
    if (!h || !d_sum) return PB_EINVAL;
    if (nseg < 1 || nseg > h->S) return fail(h, PB_EINVAL, "pb_coadd_local_codes: nseg out of range");
    if (nseg > h->processed) return fail(h, PB_EINVAL, "pb_coadd_local_codes: more segments than the batch holds");
    if (h->d_coadd_target && d_sum == h->d_coadd_target)
        return fail(h, PB_ESTATE, "pb_coadd_local_codes: d_sum is the set's fp32 coadd target");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    if (hipEventQuery(h->ev_chan) != hipSuccess) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_chan, 0));
    hipError_t e = launch_coadd_local_codes(h, nseg, d_sum, accumulate, cs);
    if (e == hipSuccess) e = hipEventRecord(h->ev_cl, cs);      // (the set's next detect writes codes and planes alike)
    HIPCHK(h, e);
    return PB_OK;
}

// leaves of S over items[0..n) left to right: the even positions' subtree, then the odd positions'
static void tree_order_rec(const int32_t *items, int n, int32_t *&out)
{
    if (n <= 1) {
        if (n == 1) *out++ = items[0];
        return;
    }
    int32_t ev[PB_COADD_MAX_LEAVES], od[PB_COADD_MAX_LEAVES];
    int ne = 0, no = 0;
    for (int i = 0; i < n; ++i) (i & 1 ? od[no++] : ev[ne++]) = items[i];
    tree_order_rec(ev, ne, out);
    tree_order_rec(od, no, out);
}

extern "C" int pb_coadd_tree_order(int n, int32_t *order)
{
    if (!order || n < 1 || n > PB_COADD_MAX_LEAVES) return PB_EINVAL;
    int32_t items[PB_COADD_MAX_LEAVES];
    for (int i = 0; i < n; ++i) items[i] = i;
    int32_t *out = order;
    tree_order_rec(items, n, out);
    return PB_OK;
}

// The incoherent sum in the defined order (coadd_tree.hip): one rank's node of the tree from
// the selected set's fp32 planes, ordered against detect and the set's next batch exactly like pb_coadd_local.
extern "C" int pb_coadd_local_tree(pb_handle *h, int nseg, const int32_t *ant_order, int n, float *d_dst)
{
    if (!h || !d_dst || !ant_order) return PB_EINVAL;
    if (!h->cfg.keep_ave) return fail(h, PB_ESTATE, "pb_coadd_local_tree needs keep_ave=1");
    if (nseg < 1 || nseg > h->S) return fail(h, PB_EINVAL, "pb_coadd_local_tree: nseg out of range");
    if (nseg > h->processed) return fail(h, PB_EINVAL, "pb_coadd_local_tree: more segments than the batch holds");
    if (n < 1 || n > PB_COADD_MAX_LEAVES) return fail(h, PB_EINVAL, "pb_coadd_local_tree: 1..PB_COADD_MAX_LEAVES leaves");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int stream = h->cfg.rfi_mode == 0 ? 0 : 1;
    const float *leaves[PB_COADD_MAX_LEAVES];
    for (int i = 0; i < n; ++i) {
        const int a = ant_order[i];
        if (a < 0 || a >= h->A) return fail(h, PB_EINVAL, "pb_coadd_local_tree: antenna index out of range");
        leaves[i] = (h->d_coadd_target && a == 0) ? h->d_coadd_target
                                                  : h->d_ave + ((size_t)a * 2 + stream) * h->S * h->ave_per_seg;
        if (leaves[i] == d_dst && n > 1) return fail(h, PB_EINVAL, "pb_coadd_local_tree: d_dst is one of the leaves");
    }
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    if (hipEventQuery(h->ev_chan) != hipSuccess) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_chan, 0));
    if (n == 1 && leaves[0] == d_dst) return PB_OK;       // (a coadd target: detect wrote the leaf where it is wanted;
                                                          //  pb_coadd_release books the "planes consumed" event)
    hipStream_t s_main = h->stream;
    h->stream = cs;                       // (StageTimer records on h->stream)
    hipError_t e;
    {
        StageTimer t(h, PB_ST_COADD);
        e = launch_coadd_tree(leaves, n, d_dst, (size_t)nseg * h->ave_per_seg, cs);
        t.stop();
    }
    if (e == hipSuccess) e = hipEventRecord(h->ev_cl, cs);
    h->stream = s_main;
    HIPCHK(h, e);
    return PB_OK;
}

// The root's part: T_n over the gathered partial sums (caller-owned planes), on the coadd stream.
extern "C" int pb_coadd_tree(pb_handle *h, const float *const *d_leaves, int n, float *d_dst, size_t nfloat)
{
    if (!h || !d_leaves || !d_dst) return PB_EINVAL;
    if (n < 1 || n > PB_COADD_MAX_LEAVES) return fail(h, PB_EINVAL, "pb_coadd_tree: 1..PB_COADD_MAX_LEAVES leaves");
    if (nfloat & 3) return fail(h, PB_EINVAL, "pb_coadd_tree: nfloat must be a multiple of 4");
    for (int i = 0; i < n; ++i) {
        if (!d_leaves[i] || ((uintptr_t)d_leaves[i] & 15)) return fail(h, PB_EINVAL, "pb_coadd_tree: leaf null or not 16-byte aligned");
        if (d_leaves[i] == d_dst && n > 1) return fail(h, PB_EINVAL, "pb_coadd_tree: d_dst is one of the leaves");
    }
    if ((uintptr_t)d_dst & 15) return fail(h, PB_EINVAL, "pb_coadd_tree: d_dst not 16-byte aligned");
    if (n == 1 && d_leaves[0] == d_dst) return PB_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    hipStream_t s_main = h->stream;
    h->stream = cs;
    hipError_t e;
    {
        StageTimer t(h, PB_ST_COADD);
        e = launch_coadd_tree(d_leaves, n, d_dst, nfloat, cs);
        t.stop();
    }
    h->stream = s_main;
    HIPCHK(h, e);
    return PB_OK;
}

extern "C" int pb_set_coadd_target(pb_handle *h, float *d_sum)
{
    if (!h) return PB_EINVAL;
    if (d_sum && (h->A != 1 || !h->cfg.keep_ave || h->cfg.fft_backend != PB_FFT_LDS))
        return fail(h, PB_ESTATE, "pb_set_coadd_target needs nant=1, keep_ave=1 and the LDS back end");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));              // no detect of this handle is writing a plane
    h->d_coadd_target = d_sum;
    return PB_OK;
}

extern "C" int pb_coadd_release(pb_handle *h)
{
    if (!h) return PB_EINVAL;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    HIPCHK(h, hipEventRecord(h->ev_cl, cs));      // what the set's next detect waits for before it writes the planes
    return PB_OK;
}

// the two pinned / device slots the coadded bytes alternate between (allocated on first use)
static int coadd_slots(pb_handle *h)
{
    const size_t nb = (size_t)h->S * h->trim;
    if (!h->d_coadd_codes) {
        HIPCHK(h, dmalloc(h, &h->d_coadd_codes, 2 * nb));
        HIPCHK(h, hipHostMalloc((void **)&h->h_coadd_codes, 2 * nb, hipHostMallocDefault));
        for (int i = 0; i < 2; ++i) HIPCHK(h, hipEventCreateWithFlags(&h->ev_coadd[i], hipEventDisableTiming));
        h->coadd_slot = 0;
    }
    return PB_OK;
}

// Root side of the incoherent sum: scale by 1/sqrt(N), requantise on the GPU, bring the bytes to
// pinned host memory asynchronously.  codes_host != NULL: wait and copy out (simple, blocking).
// codes_host == NULL: return at once; the bytes of call k are read with pb_coadd_fetch_ptr after
// later work has been queued (two pinned buffers alternate), so the root does not stall per batch.
extern "C" int pb_coadd_finish(pb_handle *h, int nseg, const float *d_sum, int nant_total, uint8_t *codes_host)
{
    if (!h || !d_sum || nant_total < 1) return PB_EINVAL;
    if (nseg < 1 || nseg > h->S) return fail(h, PB_EINVAL, "pb_coadd_finish: nseg out of range");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t nb = (size_t)h->S * h->trim;
    if (int rc = coadd_slots(h)) return rc;
    const int slot = h->coadd_slot;
    h->coadd_slot ^= 1;
    h->coadd_last = slot;
    const float scale = (float)(1.0 / sqrt((double)nant_total));
    // requantise and copy out on the coadd stream, in order (the slot's previous copy-out, two calls
    // ago, went through the same stream)
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    HIPCHK(h, launch_coadd_digitise(h, nseg, d_sum, scale, h->d_coadd_codes + slot * nb, cs));
    HIPCHK(h, launch_copy_out(h->sched, h->h_coadd_codes + slot * nb, h->d_coadd_codes + slot * nb, (size_t)nseg * h->trim, cs));
    HIPCHK(h, hipEventRecord(h->ev_coadd[slot], cs));
    if (codes_host) {
        HIPCHK(h, hipEventSynchronize(h->ev_coadd[slot]));
        memcpy(codes_host, h->h_coadd_codes + slot * nb, (size_t)nseg * h->trim);
    }
    return PB_OK;
}

// One rank's slice of the ordered sum, requantised where it was summed (coadd.py, "sliced" layout).
extern "C" int pb_coadd_digitise(pb_handle *h, const float *d_sum, size_t nfloat, int nant_total, uint8_t *d_codes)
{
    if (!h || !d_sum || !d_codes || nant_total < 1) return PB_EINVAL;
    if (h->cfg.npol != 1) return fail(h, PB_ESTATE, "pb_coadd_digitise: a flat plane range is a code range only with npol = 1");
    if (nfloat == 0 || (nfloat & 7)) return fail(h, PB_EINVAL, "pb_coadd_digitise: nfloat must be a positive multiple of 8");
    if (((uintptr_t)d_sum & 15) || ((uintptr_t)d_codes & 3)) return fail(h, PB_EINVAL, "pb_coadd_digitise: misaligned buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    const float scale = (float)(1.0 / sqrt((double)nant_total));
    HIPCHK(h, launch_coadd_digitise_flat(h, d_sum, nfloat, scale, d_codes, cs));
    return PB_OK;
}

// The batch's coadded bytes, assembled on the device by the caller, to the pinned buffer pb_coadd_fetch_ptr hands out.
extern "C" int pb_coadd_publish(pb_handle *h, const uint8_t *d_codes, size_t nbytes)
{
    if (!h || !d_codes) return PB_EINVAL;
    const size_t nb = (size_t)h->S * h->trim;
    if (nbytes == 0 || nbytes > nb) return fail(h, PB_EINVAL, "pb_coadd_publish: more bytes than a batch holds");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (int rc = coadd_slots(h)) return rc;
    const int slot = h->coadd_slot;
    h->coadd_slot ^= 1;
    h->coadd_last = slot;
    hipStream_t cs = h->s_coadd ? h->s_coadd : h->stream;
    HIPCHK(h, launch_copy_out(h->sched, h->h_coadd_codes + slot * nb, d_codes, nbytes, cs));
    HIPCHK(h, hipEventRecord(h->ev_coadd[slot], cs));
    return PB_OK;
}

extern "C" int pb_coadd_fetch_ptr(pb_handle *h, int age, const uint8_t **codes)
{
    if (!h || !codes) return PB_EINVAL;
    if (!h->d_coadd_codes) return fail(h, PB_ESTATE, "pb_coadd_fetch_ptr: no pb_coadd_finish yet");
    if (age < 0 || age > 1) return fail(h, PB_EINVAL, "pb_coadd_fetch_ptr: age must be 0 (latest) or 1 (previous)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int slot = age == 0 ? h->coadd_last : (h->coadd_last ^ 1);
    HIPCHK(h, hipEventSynchronize(h->ev_coadd[slot]));
    *codes = h->h_coadd_codes + (size_t)slot * h->S * h->trim;
    return PB_OK;
}

extern "C" int pb_profile(pb_handle *h, int enable)
{
    if (!h) return PB_EINVAL;
    h->profile = enable != 0;
    if (h->profile) {
        // enough events for a few batches in flight; later ones are recycled by retire_done
        (void)hipSetDevice(h->cfg.device);
        while (h->ev_pool.size() < 64) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            h->ev_pool.push_back(e);
        }
    }
    return PB_OK;
}

extern "C" int pb_get_timers(pb_handle *h, pb_timers *out, int reset)
{
    if (!h || !out) return PB_EINVAL;
    (void)hipSetDevice(h->cfg.device);
    drain_timers(h);
    *out = h->timers;
    if (reset) memset(&h->timers, 0, sizeof h->timers);
    return PB_OK;
}

extern "C" int pb_debug_fetch(pb_handle *h, int what, int ant, int seg, void *dst, size_t nbytes)
{
    if (!h || !dst) return PB_EINVAL;
    if (check_ant(h, ant)) return PB_EINVAL;
    if (seg < 0 || seg >= h->S) return fail(h, PB_EINVAL, "segment out of range");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, sync_all(h));
    const size_t nb = h->nblk_seg, cap = (size_t)h->S * nb;
    if (what == PB_DBG_POW || what == PB_DBG_KUR || what == PB_DBG_DAG) {
        if (!h->d_stats) return fail(h, PB_ESTATE, "pb_debug_fetch: statistics need debug_keep=1");
        if (nbytes != 2 * nb * sizeof(float)) return fail(h, PB_EINVAL, "pb_debug_fetch: size");
        for (int pol = 0; pol < 2; ++pol) {
            const float *src = h->d_stats + (size_t)ant * 6 * cap + ((size_t)what * 2 + pol) * cap + (size_t)seg * nb;
            HIPCHK(h, hipMemcpy((float *)dst + pol * nb, src, nb * sizeof(float), hipMemcpyDeviceToHost));
        }
        return PB_OK;
    }
    if (what == PB_DBG_POW_FB || what == PB_DBG_KUR_FB || what == PB_DBG_DAG_FB) {
        if (!h->d_stats) return fail(h, PB_ESTATE, "pb_debug_fetch: statistics need debug_keep=1");
        const size_t rcap = (size_t)h->S * h->R;
        const float *fb = h->d_stats + (size_t)h->A * 6 * cap + (size_t)ant * 5 * rcap;
        const int npl = what == PB_DBG_DAG_FB ? 1 : 2;
        const int first = what == PB_DBG_POW_FB ? 0 : (what == PB_DBG_KUR_FB ? 2 : 4);
        if (nbytes != (size_t)npl * h->R * sizeof(float)) return fail(h, PB_EINVAL, "pb_debug_fetch: size");
        for (int pl = 0; pl < npl; ++pl)
            HIPCHK(h, hipMemcpy((float *)dst + (size_t)pl * h->R, fb + (size_t)(first + pl) * rcap + (size_t)seg * h->R,
                                (size_t)h->R * sizeof(float), hipMemcpyDeviceToHost));
        return PB_OK;
    }
    if (what == PB_DBG_FLAGS) {
        if (nbytes != nb) return fail(h, PB_EINVAL, "pb_debug_fetch: size");
        HIPCHK(h, hipMemcpy(dst, h->d_flags + (size_t)ant * cap + (size_t)seg * nb, nb, hipMemcpyDeviceToHost));
        return PB_OK;
    }
    if (what == PB_DBG_ROWWEIGHT) {
        if (nbytes != (size_t)h->R * sizeof(float)) return fail(h, PB_EINVAL, "pb_debug_fetch: size");
        HIPCHK(h, hipMemcpy(dst, h->d_wrow + (size_t)ant * h->S * h->R + (size_t)seg * h->R, nbytes, hipMemcpyDeviceToHost));
        return PB_OK;
    }
    return fail(h, PB_EINVAL, "pb_debug_fetch: unknown item");
}

extern "C" int pb_debug_dag_check(pb_handle *h, float kur_lo, float kur_hi, uint64_t *nchecked, uint64_t *nmismatch,
                                  float *bands4)
{
    if (!h || !nchecked || !nmismatch) return PB_EINVAL;
    if (!(kur_lo > 0) || !(kur_hi >= kur_lo)) return fail(h, PB_EINVAL, "pb_debug_dag_check: 0 < kur_lo <= kur_hi");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const uint32_t b0 = __builtin_bit_cast(uint32_t, kur_lo), b1 = __builtin_bit_cast(uint32_t, kur_hi);
    const uint64_t n = (uint64_t)(b1 - b0) + 1;
    unsigned long long *d_bad = nullptr, bad = 0;
    HIPCHK(h, hipMalloc((void **)&d_bad, sizeof bad));
    hipError_t e = hipMemset(d_bad, 0, sizeof bad);
    if (e == hipSuccess) e = launch_dag_check(h, b0, n, d_bad);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    HIPCHK(h, e);
    *nchecked = n;
    *nmismatch = bad;
    if (bands4) {
        bands4[0] = h->dag.t_lo_sure;
        bands4[1] = h->dag.t_lo_clear;
        bands4[2] = h->dag.t_hi_clear;
        bands4[3] = h->dag.t_hi_sure;
    }
    return PB_OK;
}

extern "C" int pb_channelize_f32(pb_handle *h, const float *x, int nrows, int taps, float *out)
{
    if (!h || !x || !out || nrows < 1) return PB_EINVAL;
    if (!(taps == 1 || taps == 4)) return fail(h, PB_EINVAL, "taps must be 1 or 4");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    float *d_x = nullptr;
    float2 *d_o = nullptr;
    const size_t nin = (size_t)(nrows + taps - 1) * PB_NFFT;
    hipError_t e = hipMalloc((void **)&d_x, nin * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&d_o, (size_t)nrows * PB_NCHAN * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(d_x, x, nin * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_channelize_f32(h, d_x, nrows, taps, d_o);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_o, (size_t)nrows * PB_NCHAN * sizeof(float2), hipMemcpyDeviceToHost);
    if (d_x) (void)hipFree(d_x);
    if (d_o) (void)hipFree(d_o);
    HIPCHK(h, e);
    return PB_OK;
}
