// On-GPU incoherent dedispersion + boxcar matched filter over a (coadded) filterbank block:
// the stage downstream of process_baseband in the reference's production chain
// (external `heimdall_stream -dm 2 1000 -boxcar_max 64 -nsamps_gulp 30720 -zap_chans 0 190
//  -zap_chans 3900 4096`, /root/reference/scripts/start_heimdall_single_antenna:21).
// heimdall and its dedisp library are third-party and absent, so parity with heimdall's candidate
// list is UNPINNED; what is pinned is the arithmetic below against a NumPy restatement (tests) and
// the recovered S/N against the reference's own estimator (analysis/loc_step0.py:optimize_pulse).
//
//   1. transpose the time-major codes [T][nchan] (SIGPROC order, 8/4/2-bit) to channel-major u8
//   2. brute-force dedispersion: D[dm][t] = sum over unzapped channels of x[c][t + delay(dm, c)],
//      delay = round(4.148808e3 * dm * (f_c^-2 - f_top^-2) / tsamp)  (MHz; src/candidate.py:33)
//   3. per-DM mean / rms with one 3-sigma clip, boxcars 2^0..2^k (k = log2 boxcar_max):
//      S/N(t, w) = (sum_{i<w} D[t+i] - w mean) / (rms sqrt(w)); keep the best width per (dm, t)
// All integer up to the normalisation, so 1-2 are bit-exact vs the NumPy restatement.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "pb_hip.h"

struct pb_search {
    int device, nchan, max_samples, ndm, nbox;
    float fch1, foff, tsamp, dm_min, dm_step;
    int max_delay;
    uint8_t *d_codes, *d_xt;       // raw block; channel-major [nchan][tpitch]
    int32_t *d_delay;              // [ndm][nact] delays of the channels that are not zapped
    int32_t *d_chans;              // [nact] their indices
    int nact;
    uint32_t *d_D;                 // [ndm][max_samples]
    float *d_stats;                // [ndm][2] mean, rms
    float *d_snr;                  // [ndm][max_samples]
    uint8_t *d_wid;                // [ndm][max_samples] log2 width
    uint64_t *d_P;                 // [ndm][tpitch + 1] prefix sums of D (P[0] = 0)
    int32_t *d_peaks;              // [1 + 4 * max_peaks]: count, then (dm index, sample, S/N bits, log2 width)
    int max_peaks;
    int baseline;                  // running-baseline window in samples (0: one clipped mean per DM series)
    int last_tout;
    float ms[6];                   // last run: H2D, transpose, dedisperse, prefix + statistics, boxcar, D2H
    hipEvent_t ev[7];
    size_t tpitch;
    hipStream_t stream;
    std::string err;
};

static std::string g_search_err;

#define SCHK(s, call)                                                         \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            (s)->err = std::string(#call) + ": " + hipGetErrorString(e_);     \
            return PB_EHIP;                                                   \
        }                                                                     \
    } while (0)

// codes: [T][nchan] samples of nbit bits, low bits first within a byte -> xt[c][t] u8
__global__ __launch_bounds__(256) void k_transpose_codes(const uint8_t *__restrict__ codes, uint8_t *__restrict__ xt,
                                                         int T, int nchan, int nbit, size_t tpitch)
{
    __shared__ uint8_t tile[64][65];
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int per = 8 / nbit, maskv = (1 << nbit) - 1;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int tt = i / 64, cc = i % 64;
        const int t = t0 + tt, c = c0 + cc;
        uint8_t v = 0;
        if (t < T && c < nchan) {
            const size_t n = (size_t)t * nchan + c;
            v = (codes[n / per] >> (nbit * (n % per))) & maskv;
        }
        tile[tt][cc] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i / 64, tt = i % 64;
        const int t = t0 + tt, c = c0 + cc;
        if (t < T && c < nchan) xt[(size_t)c * tpitch + t] = tile[tt][cc];
    }
}

// Brute-force dedispersion, tiled over (time, DM): one workgroup = 2048 output samples x DDM_DB
// neighbouring trial DMs, a thread = 8 consecutive samples (two packed dwords) of every DM of the block.
// For each channel the 8 samples at t + delay(dm, c) are 12 bytes fetched with ONE 16-byte-or-less load
// per lane (the delay is uniform over the wave: its aligned part goes into the scalar offset of the
// load, its low two bits into v_alignbyte) instead of eight 1-byte loads, and neighbouring DMs of the
// block hit the same cache lines.  The bytes are summed as packed 16-bit pairs (v_pk_add_u16), widened to
// 32 bits every 256 channels (256 x 255 < 2^16).  Integer arithmetic: the sums are exact whatever the order.
#ifndef DDM_DB
#define DDM_DB 4
#endif
#ifndef DDM_SAMPLES
#define DDM_SAMPLES 8
#endif
#ifndef DDM_CU
#define DDM_CU 4          // channels whose loads are in flight together
#endif
#define DDM_TILE (256 * DDM_SAMPLES)

// chans: the nact channels that are not zapped; delay: [ndm][nact] over those channels
__global__ __launch_bounds__(256) void k_dedisperse(const uint8_t *__restrict__ xt, const int32_t *__restrict__ chans,
                                                    const int32_t *__restrict__ delay, uint32_t *__restrict__ D,
                                                    int nact, int ndm, int tout, size_t tpitch, size_t dpitch)
{
    const int dm0 = blockIdx.y * DDM_DB;
    const int t = blockIdx.x * DDM_TILE + threadIdx.x * DDM_SAMPLES;      // 4-byte aligned (tpitch is a multiple of 64)
    // Lanes past the last output sample of the last tile have nothing to do, and their loads (up to a tile past
    // tout, plus the delay) would leave the last channel's row -- and, for the last channel, the allocation
    // (tpitch * nchan + 64 bytes).  No barrier in this kernel: a whole-lane exit is safe.  A live lane reads at
    // most t + d + 11 <= tout - 1 + max_delay + 11 = nsamp + 10, inside the row's pitch or the 64-byte pad.
    if (t >= tout) return;
    uint32_t lo[DDM_DB][DDM_SAMPLES / 4], hi[DDM_DB][DDM_SAMPLES / 4];   // packed u16 partial sums: bytes 0,2 / 1,3
    uint32_t acc[DDM_DB][DDM_SAMPLES];
    const int32_t *dl[DDM_DB];
#pragma unroll
    for (int j = 0; j < DDM_DB; ++j) {
        dl[j] = delay + (size_t)(dm0 + j < ndm ? dm0 + j : ndm - 1) * nact;
#pragma unroll
        for (int w = 0; w < DDM_SAMPLES / 4; ++w) lo[j][w] = hi[j][w] = 0u;
#pragma unroll
        for (int i = 0; i < DDM_SAMPLES; ++i) acc[j][i] = 0u;
    }
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < DDM_DB; ++j) {
#pragma unroll
            for (int w = 0; w < DDM_SAMPLES / 4; ++w) {
                acc[j][4 * w + 0] += lo[j][w] & 0xffffu;
                acc[j][4 * w + 1] += hi[j][w] & 0xffffu;
                acc[j][4 * w + 2] += lo[j][w] >> 16;
                acc[j][4 * w + 3] += hi[j][w] >> 16;
                lo[j][w] = hi[j][w] = 0u;
            }
        }
    };
    // one channel of one DM: samples t+d .. t+d+DDM_SAMPLES-1 = DDM_SAMPLES/4 + 1 aligned dwords from (t + d) & ~3
    auto fetch = [&](int c, int j, uint32_t (&w)[DDM_SAMPLES / 4 + 1], unsigned &sh) {
        const int32_t d = dl[j][c];                                 // uniform -> scalar load
        const uint32_t *p = (const uint32_t *)(xt + (size_t)chans[c] * tpitch + t + (d & ~3));
#pragma unroll
        for (int i = 0; i <= DDM_SAMPLES / 4; ++i) w[i] = p[i];
        sh = (unsigned)(d & 3);
    };
    auto add = [&](int j, const uint32_t (&w)[DDM_SAMPLES / 4 + 1], unsigned sh) {
#pragma unroll
        for (int i = 0; i < DDM_SAMPLES / 4; ++i) {
            const uint32_t a = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);      // bytes sh.. of w[i+1]:w[i]
            lo[j][i] += a & 0x00ff00ffu;
            hi[j][i] += (a >> 8) & 0x00ff00ffu;
        }
    };
    // DDM_CU channels at a time: all their loads are issued before the first is used (the loop is bound by
    // load latency, not by anything it computes)
    int c = 0, since = 0;
    for (; c + DDM_CU <= nact; c += DDM_CU) {
        uint32_t w[DDM_CU][DDM_DB][DDM_SAMPLES / 4 + 1];
        unsigned sh[DDM_CU][DDM_DB];
#pragma unroll
        for (int u = 0; u < DDM_CU; ++u)
#pragma unroll
            for (int j = 0; j < DDM_DB; ++j) fetch(c + u, j, w[u][j], sh[u][j]);
#pragma unroll
        for (int u = 0; u < DDM_CU; ++u)
#pragma unroll
            for (int j = 0; j < DDM_DB; ++j) add(j, w[u][j], sh[u][j]);
        since += DDM_CU;
        if (since >= 240) {                                     // 16-bit sums: at most 257 channels of 255 between flushes
            flush();
            since = 0;
        }
    }
    for (; c < nact; ++c) {
#pragma unroll
        for (int j = 0; j < DDM_DB; ++j) {
            uint32_t w[DDM_SAMPLES / 4 + 1];
            unsigned sh;
            fetch(c, j, w, sh);
            add(j, w, sh);
        }
    }
    flush();
#pragma unroll
    for (int j = 0; j < DDM_DB; ++j) {
        if (dm0 + j >= ndm) break;
        uint32_t *out = D + (size_t)(dm0 + j) * dpitch + t;
#pragma unroll
        for (int i = 0; i < DDM_SAMPLES; i += 4) {
            if (t + i + 4 <= tout) {
                *(uint4 *)(out + i) = make_uint4(acc[j][i], acc[j][i + 1], acc[j][i + 2], acc[j][i + 3]);
            } else {
                for (int q = 0; q < 4; ++q)
                    if (t + i + q < tout) out[i + q] = acc[j][i + q];
            }
        }
    }
}

__device__ __forceinline__ double block_sum(double v, double *sh)
{
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_down(v, s);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double r = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;
}

// Inclusive prefix sums of each DM series (64-bit: a series sums to ~2e10), P[0] = 0: boxcar sums and the
// running baseline become differences of two entries.  One workgroup per DM.
__global__ __launch_bounds__(256) void k_prefix(const uint32_t *__restrict__ D, uint64_t *__restrict__ P, int tout,
                                                size_t dpitch, size_t ppitch)
{
    __shared__ uint64_t sh[256];
    const uint32_t *x = D + (size_t)blockIdx.x * dpitch;
    uint64_t *p = P + (size_t)blockIdx.x * ppitch;
    const int per = (tout + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(lo + per, tout);
    uint64_t s = 0;
    for (int t = lo; t < hi; ++t) s += x[t];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 256; ++i) {
            const uint64_t v = sh[i];
            sh[i] = run;
            run += v;
        }
        p[0] = 0;
    }
    __syncthreads();
    uint64_t run = sh[threadIdx.x];
    for (int t = lo; t < hi; ++t) {
        run += x[t];
        p[t + 1] = run;
    }
}

// baseline under sample t: the mean of the series over the W samples around it (window clamped to the
// series), or the per-series constant when W == 0
__device__ __forceinline__ float baseline_at(const uint64_t *__restrict__ p, int t, int W, int tout, float gmean)
{
    if (W <= 0) return gmean;
    int lo = t - W / 2, hi = lo + W;
    if (lo < 0) { lo = 0; hi = min(W, tout); }
    if (hi > tout) { hi = tout; lo = max(0, tout - W); }
    return (float)((double)(p[hi] - p[lo]) / (double)(hi - lo));
}

// mean and rms of each DM series about its baseline, then once more over the samples within 3 rms
__global__ __launch_bounds__(256) void k_series_stats(const uint32_t *__restrict__ D, const uint64_t *__restrict__ P,
                                                      float *__restrict__ stats, int tout, int W, size_t dpitch,
                                                      size_t ppitch)
{
    __shared__ double sh[4];
    const uint32_t *x = D + (size_t)blockIdx.x * dpitch;
    const uint64_t *p = P + (size_t)blockIdx.x * ppitch;
    const float gmean0 = (float)((double)p[tout] / (double)tout);
    double s1 = 0, s2 = 0;
    for (int t = threadIdx.x; t < tout; t += 256) {
        const double v = (double)x[t] - (double)baseline_at(p, t, W, tout, gmean0);
        s1 += v;
        s2 += v * v;
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    double mean = s1 / tout, var = s2 / tout - mean * mean;
    double rms = sqrt(var > 0 ? var : 0);
    double c1 = 0, c2 = 0, cn = 0;
    for (int t = threadIdx.x; t < tout; t += 256) {
        const double v = (double)x[t] - (double)baseline_at(p, t, W, tout, gmean0);
        if (fabs(v - mean) <= 3 * rms) {
            c1 += v;
            c2 += v * v;
            cn += 1;
        }
    }
    c1 = block_sum(c1, sh);
    c2 = block_sum(c2, sh);
    cn = block_sum(cn, sh);
    if (cn > 0) {
        mean = c1 / cn;
        var = c2 / cn - mean * mean;
        // a Gaussian clipped at 3 sigma has 0.9733 of its variance left
        rms = sqrt((var > 0 ? var : 0) / 0.97330);
    }
    if (threadIdx.x == 0) {
        // [0]: level to subtract on top of the running baseline (W > 0) or the clipped mean itself (W == 0)
        stats[2 * blockIdx.x] = (float)(W > 0 ? mean : mean + (double)gmean0);
        stats[2 * blockIdx.x + 1] = (float)rms;
    }
}

// best boxcar per (DM, sample) from prefix sums; samples at or above `thresh` also go to the peak list
__global__ __launch_bounds__(256) void k_boxcar(const uint64_t *__restrict__ P, const float *__restrict__ stats,
                                                float *__restrict__ snr, uint8_t *__restrict__ wid, int tout,
                                                int nbox, int W, size_t dpitch, size_t ppitch, float thresh,
                                                int32_t *__restrict__ peaks, int max_peaks)
{
    const int dm = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tout) return;
    const uint64_t *p = P + (size_t)dm * ppitch;
    const float lvl = stats[2 * dm], rms = stats[2 * dm + 1];
    float best = -1e30f;
    int bw = 0;
    for (int k = 0; k < nbox; ++k) {
        const int w = 1 << k;
        if (t + w > tout) break;
        const float sum = (float)(p[t + w] - p[t]);
        const float base = W > 0 ? baseline_at(p, t + w / 2, W, tout, 0.f) + lvl : lvl;
        const float sn = (sum - (float)w * base) / (rms * sqrtf((float)w));
        if (sn > best) {
            best = sn;
            bw = k;
        }
    }
    if (!(rms > 0)) best = 0.f;
    snr[(size_t)dm * dpitch + t] = best;
    wid[(size_t)dm * dpitch + t] = (uint8_t)bw;
    if (peaks && best >= thresh) {
        const int i = atomicAdd(peaks, 1);
        if (i < max_peaks) {
            peaks[1 + 4 * i] = dm;
            peaks[2 + 4 * i] = t;
            peaks[3 + 4 * i] = __float_as_int(best);
            peaks[4 + 4 * i] = bw;
        }
    }
}

extern "C" const char *pb_search_last_error(const pb_search *s) { return s ? s->err.c_str() : g_search_err.c_str(); }

static int search_create(int device, int nchan, int max_samples, float fch1_mhz, float foff_mhz, float tsamp_s,
                         const std::vector<double> &dms, int boxcar_max, const int *zap_ranges, int nzap,
                         pb_search **out);

extern "C" int pb_search_create(int device, int nchan, int max_samples, float fch1_mhz, float foff_mhz, float tsamp_s,
                                float dm_min, float dm_max, float dm_step, int boxcar_max, const int *zap_ranges,
                                int nzap, pb_search **out)
{
    if (!out || !(dm_step > 0) || dm_max < dm_min) {
        g_search_err = "pb_search_create: bad argument";
        return PB_EINVAL;
    }
    const int ndm = (int)floor((dm_max - dm_min) / dm_step + 1e-6) + 1;
    std::vector<double> dms(ndm);
    for (int i = 0; i < ndm; ++i) dms[i] = (double)dm_min + (double)i * (double)dm_step;
    return search_create(device, nchan, max_samples, fch1_mhz, foff_mhz, tsamp_s, dms, boxcar_max, zap_ranges, nzap, out);
}

// The same over an explicit list of trial DMs (ascending), e.g. a tolerance-spaced one
extern "C" int pb_search_create_list(int device, int nchan, int max_samples, float fch1_mhz, float foff_mhz,
                                     float tsamp_s, const float *dm_list, int ndm, int boxcar_max,
                                     const int *zap_ranges, int nzap, pb_search **out)
{
    if (!out || !dm_list || ndm < 1) {
        g_search_err = "pb_search_create_list: bad argument";
        return PB_EINVAL;
    }
    std::vector<double> dms(dm_list, dm_list + ndm);
    for (int i = 1; i < ndm; ++i)
        if (!(dms[i] > dms[i - 1])) {
            g_search_err = "pb_search_create_list: the DM list must be strictly ascending";
            return PB_EINVAL;
        }
    return search_create(device, nchan, max_samples, fch1_mhz, foff_mhz, tsamp_s, dms, boxcar_max, zap_ranges, nzap, out);
}

static int search_create(int device, int nchan, int max_samples, float fch1_mhz, float foff_mhz, float tsamp_s,
                         const std::vector<double> &dms, int boxcar_max, const int *zap_ranges, int nzap,
                         pb_search **out)
{
    const float dm_min = (float)dms.front();
    if (!out || nchan < 1 || max_samples < 64 || boxcar_max < 1) {
        g_search_err = "pb_search_create: bad argument";
        return PB_EINVAL;
    }
    if (!(dm_min >= 0) || device < 0) {
        // a negative DM would give negative delays (reads before the block); a negative device goes to hipSetDevice
        g_search_err = "pb_search_create: dm_min and device must be >= 0";
        return PB_EINVAL;
    }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
        g_search_err = "pb_search_create: no HIP device (no CPU fallback)";
        return PB_EHIP;
    }
    pb_search *s = new pb_search();
    s->device = device;
    s->nchan = nchan;
    s->max_samples = max_samples;
    s->fch1 = fch1_mhz;
    s->foff = foff_mhz;
    s->tsamp = tsamp_s;
    s->dm_min = dm_min;
    s->dm_step = dms.size() > 1 ? (float)(dms[1] - dms[0]) : 0.f;
    s->ndm = (int)dms.size();
    s->nbox = 0;
    while ((1 << s->nbox) <= boxcar_max) s->nbox++;
    s->d_codes = s->d_xt = s->d_wid = nullptr;
    s->d_delay = nullptr;
    s->d_chans = nullptr;
    s->d_D = nullptr;
    s->d_stats = s->d_snr = nullptr;
    s->stream = nullptr;
    s->d_P = nullptr;
    s->d_peaks = nullptr;
    s->max_peaks = 1 << 20;
    s->baseline = 0;
    s->last_tout = 0;
    for (int i = 0; i < 6; ++i) s->ms[i] = 0.f;
    for (int i = 0; i < 7; ++i) s->ev[i] = nullptr;
    // delay table in double on the host; the top of the band (highest frequency) is the reference
    std::vector<int32_t> chans;
    for (int c = 0; c < nchan; ++c) {
        bool zap = false;
        for (int z = 0; z < nzap; ++z)
            if (c >= zap_ranges[2 * z] && c < zap_ranges[2 * z + 1]) zap = true;
        if (!zap) chans.push_back(c);
    }
    s->nact = (int)chans.size();
    if (s->nact == 0) {
        g_search_err = "pb_search_create: every channel is zapped";
        delete s;
        return PB_EINVAL;
    }
    std::vector<int32_t> delay((size_t)s->ndm * s->nact);
    const double ftop = foff_mhz < 0 ? fch1_mhz : fch1_mhz + (nchan - 1) * foff_mhz;
    int maxd = 0;
    for (int i = 0; i < s->ndm; ++i) {
        const double dm = dms[i];
        for (int k = 0; k < s->nact; ++k) {
            const double f = (double)fch1_mhz + (double)chans[k] * (double)foff_mhz;
            const double d = 4.148808e3 * dm * (1.0 / (f * f) - 1.0 / (ftop * ftop)) / (double)tsamp_s;
            const int32_t di = (int32_t)floor(d + 0.5);
            delay[(size_t)i * s->nact + k] = di;
            if (di > maxd) maxd = di;
        }
    }
    s->max_delay = maxd;
    s->tpitch = ((size_t)max_samples + 63) / 64 * 64;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_codes, (size_t)max_samples * nchan);
    // + 64: the dedispersion kernel fetches whole dwords around its 8 samples (up to 11 bytes past the last one)
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_xt, s->tpitch * nchan + 64);
    if (e == hipSuccess) e = hipMemset(s->d_xt, 0, s->tpitch * nchan + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_delay, delay.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_chans, chans.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(s->d_chans, chans.data(), chans.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_D, (size_t)s->ndm * s->tpitch * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_stats, (size_t)s->ndm * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_snr, (size_t)s->ndm * s->tpitch * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_wid, (size_t)s->ndm * s->tpitch);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_P, (size_t)s->ndm * (s->tpitch + 8) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_peaks, (size_t)(1 + 4 * s->max_peaks) * sizeof(int32_t));
    for (int i = 0; i < 7 && e == hipSuccess; ++i) e = hipEventCreate(&s->ev[i]);
    if (e == hipSuccess) e = hipMemcpy(s->d_delay, delay.data(), delay.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        g_search_err = std::string("pb_search_create: ") + hipGetErrorString(e);
        pb_search_destroy(s);
        return PB_EHIP;
    }
    *out = s;
    return PB_OK;
}

extern "C" void pb_search_destroy(pb_search *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    void *p[] = {s->d_codes, s->d_xt, s->d_delay, s->d_D, s->d_stats, s->d_snr, s->d_wid, s->d_P, s->d_peaks, s->d_chans};
    for (int i = 0; i < 7; ++i)
        if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
    for (void *q : p)
        if (q) (void)hipFree(q);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" int pb_search_info(const pb_search *s, int *ndm, int *nbox, int *max_delay)
{
    if (!s) return PB_EINVAL;
    if (ndm) *ndm = s->ndm;
    if (nbox) *nbox = s->nbox;
    if (max_delay) *max_delay = s->max_delay;
    return PB_OK;
}

extern "C" int pb_search_set_baseline(pb_search *s, int window_samples)
{
    if (!s || window_samples < 0) return PB_EINVAL;
    s->baseline = window_samples;
    return PB_OK;
}

static int search_run(pb_search *s, const void *codes, int codes_on_device, int nsamp, int nbit, float thresh,
                      bool want_peaks)
{
    if (!s || !codes) return PB_EINVAL;
    if (nsamp > s->max_samples || !(nbit == 8 || nbit == 4 || nbit == 2)) {
        s->err = "pb_search_run: block too long or bad nbit";
        return PB_EINVAL;
    }
    const int tout = nsamp - s->max_delay;
    if (tout < 64) {
        s->err = "pb_search_run: block shorter than the largest dispersion delay";
        return PB_EINVAL;
    }
    SCHK(s, hipSetDevice(s->device));
    const size_t nbytes = (size_t)nsamp * s->nchan * nbit / 8;
    const uint8_t *d_codes = (const uint8_t *)codes;
    SCHK(s, hipEventRecord(s->ev[0], s->stream));
    if (!codes_on_device) {
        SCHK(s, hipMemcpyAsync(s->d_codes, codes, nbytes, hipMemcpyHostToDevice, s->stream));
        d_codes = s->d_codes;
    }
    SCHK(s, hipEventRecord(s->ev[1], s->stream));
    dim3 gt((nsamp + 63) / 64, (s->nchan + 63) / 64);
    k_transpose_codes<<<gt, 256, 0, s->stream>>>(d_codes, s->d_xt, nsamp, s->nchan, nbit, s->tpitch);
    SCHK(s, hipEventRecord(s->ev[2], s->stream));
    dim3 gd((tout + 255) / 256, s->ndm);
    dim3 gdd((tout + DDM_TILE - 1) / DDM_TILE, (s->ndm + DDM_DB - 1) / DDM_DB);
    k_dedisperse<<<gdd, 256, 0, s->stream>>>(s->d_xt, s->d_chans, s->d_delay, s->d_D, s->nact, s->ndm, tout, s->tpitch, s->tpitch);
    SCHK(s, hipEventRecord(s->ev[3], s->stream));
    const size_t ppitch = s->tpitch + 8;
    k_prefix<<<s->ndm, 256, 0, s->stream>>>(s->d_D, s->d_P, tout, s->tpitch, ppitch);
    k_series_stats<<<s->ndm, 256, 0, s->stream>>>(s->d_D, s->d_P, s->d_stats, tout, s->baseline, s->tpitch, ppitch);
    SCHK(s, hipEventRecord(s->ev[4], s->stream));
    if (want_peaks) SCHK(s, hipMemsetAsync(s->d_peaks, 0, sizeof(int32_t), s->stream));
    k_boxcar<<<gd, 256, 0, s->stream>>>(s->d_P, s->d_stats, s->d_snr, s->d_wid, tout, s->nbox, s->baseline, s->tpitch,
                                        ppitch, thresh, want_peaks ? s->d_peaks : nullptr, s->max_peaks);
    SCHK(s, hipEventRecord(s->ev[5], s->stream));
    SCHK(s, hipGetLastError());
    s->last_tout = tout;
    return PB_OK;
}

static void search_times(pb_search *s)
{
    for (int i = 0; i < 6; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]) == hipSuccess) s->ms[i] = ms;
    }
}

// codes: nsamp x nchan samples of nbit bits in SIGPROC order (host, or device if codes_on_device).
// Outputs (host, any may be NULL): snr [ndm][tout] float, width_log2 [ndm][tout] u8,
// series [ndm][tout] u32 (the dedispersed sums), stats [ndm][2]; tout = nsamp - max_delay.
extern "C" int pb_search_run(pb_search *s, const void *codes, int codes_on_device, int nsamp, int nbit, float *snr,
                             uint8_t *width_log2, uint32_t *series, float *stats, int *tout_out)
{
    int rc = search_run(s, codes, codes_on_device, nsamp, nbit, 0.f, false);
    if (rc) return rc;
    const int tout = s->last_tout;
    SCHK(s, hipStreamSynchronize(s->stream));
    if (snr) SCHK(s, hipMemcpy2D(snr, tout * sizeof(float), s->d_snr, s->tpitch * sizeof(float), tout * sizeof(float), s->ndm, hipMemcpyDeviceToHost));
    if (width_log2) SCHK(s, hipMemcpy2D(width_log2, tout, s->d_wid, s->tpitch, tout, s->ndm, hipMemcpyDeviceToHost));
    if (series) SCHK(s, hipMemcpy2D(series, tout * sizeof(uint32_t), s->d_D, s->tpitch * sizeof(uint32_t), tout * sizeof(uint32_t), s->ndm, hipMemcpyDeviceToHost));
    if (stats) SCHK(s, hipMemcpy(stats, s->d_stats, (size_t)s->ndm * 2 * sizeof(float), hipMemcpyDeviceToHost));
    SCHK(s, hipEventRecord(s->ev[6], s->stream));
    SCHK(s, hipEventSynchronize(s->ev[6]));
    search_times(s);
    if (tout_out) *tout_out = tout;
    return PB_OK;
}

// The search as the production chain uses it: only the (DM, sample) points at or above `threshold` come back
// (what heimdall's giant finder starts from), not the S/N planes: peaks[4 i ..] = DM index, sample, S/N
// (float bits), log2 width; *npeaks may exceed max_out when the list was truncated.
extern "C" int pb_search_peaks(pb_search *s, const void *codes, int codes_on_device, int nsamp, int nbit,
                               float threshold, int32_t *peaks, int max_out, int *npeaks, int *tout_out)
{
    if (!peaks || !npeaks || max_out < 1) return PB_EINVAL;
    int rc = search_run(s, codes, codes_on_device, nsamp, nbit, threshold, true);
    if (rc) return rc;
    int32_t n = 0;
    SCHK(s, hipMemcpyAsync(&n, s->d_peaks, sizeof n, hipMemcpyDeviceToHost, s->stream));
    SCHK(s, hipStreamSynchronize(s->stream));
    const int ncopy = std::min(std::min((int)n, max_out), s->max_peaks);      // never past the device list
    if (ncopy > 0) SCHK(s, hipMemcpy(peaks, s->d_peaks + 1, (size_t)ncopy * 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    SCHK(s, hipEventRecord(s->ev[6], s->stream));
    SCHK(s, hipEventSynchronize(s->ev[6]));
    search_times(s);
    *npeaks = n;
    if (tout_out) *tout_out = s->last_tout;
    return PB_OK;
}

extern "C" int pb_search_timers(const pb_search *s, float *ms6)
{
    if (!s || !ms6) return PB_EINVAL;
    for (int i = 0; i < 6; ++i) ms6[i] = s->ms[i];
    return PB_OK;
}
