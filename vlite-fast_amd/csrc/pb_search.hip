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
    int32_t *d_delay;              // [ndm][nchan], -1 = zapped
    uint32_t *d_D;                 // [ndm][max_samples]
    float *d_stats;                // [ndm][2] mean, rms
    float *d_snr;                  // [ndm][max_samples]
    uint8_t *d_wid;                // [ndm][max_samples] log2 width
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

__global__ __launch_bounds__(256) void k_dedisperse(const uint8_t *__restrict__ xt, const int32_t *__restrict__ delay,
                                                    uint32_t *__restrict__ D, int nchan, int tout, size_t tpitch,
                                                    size_t dpitch)
{
    const int dm = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int32_t *dl = delay + (size_t)dm * nchan;
    if (t >= tout) return;
    uint32_t acc = 0;
    for (int c = 0; c < nchan; ++c) {
        const int32_t d = dl[c];        // uniform -> scalar load
        if (d < 0) continue;
        acc += xt[(size_t)c * tpitch + t + d];
    }
    D[(size_t)dm * dpitch + t] = acc;
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

// mean and rms of each DM series, then once more over the samples within 3 rms of the mean
__global__ __launch_bounds__(256) void k_series_stats(const uint32_t *__restrict__ D, float *__restrict__ stats,
                                                      int tout, size_t dpitch)
{
    __shared__ double sh[4];
    const uint32_t *x = D + (size_t)blockIdx.x * dpitch;
    double s1 = 0, s2 = 0;
    for (int t = threadIdx.x; t < tout; t += 256) {
        const double v = x[t];
        s1 += v;
        s2 += v * v;
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    double mean = s1 / tout, var = s2 / tout - mean * mean;
    double rms = sqrt(var > 0 ? var : 0);
    double c1 = 0, c2 = 0, cn = 0;
    for (int t = threadIdx.x; t < tout; t += 256) {
        const double v = x[t];
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
        stats[2 * blockIdx.x] = (float)mean;
        stats[2 * blockIdx.x + 1] = (float)rms;
    }
}

__global__ __launch_bounds__(256) void k_boxcar(const uint32_t *__restrict__ D, const float *__restrict__ stats,
                                                float *__restrict__ snr, uint8_t *__restrict__ wid, int tout,
                                                int nbox, size_t dpitch)
{
    const int dm = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tout) return;
    const uint32_t *x = D + (size_t)dm * dpitch;
    const float mean = stats[2 * dm], rms = stats[2 * dm + 1];
    float best = -1e30f;
    int bw = 0;
    uint32_t acc = 0;
    int have = 0;
    for (int k = 0; k < nbox; ++k) {
        const int w = 1 << k;
        if (t + w > tout) break;
        for (; have < w; ++have) acc += x[t + have];
        const float s = ((float)acc - (float)w * mean) / (rms * sqrtf((float)w));
        if (s > best) {
            best = s;
            bw = k;
        }
    }
    snr[(size_t)dm * dpitch + t] = rms > 0 ? best : 0.f;
    wid[(size_t)dm * dpitch + t] = (uint8_t)bw;
}

extern "C" const char *pb_search_last_error(const pb_search *s) { return s ? s->err.c_str() : g_search_err.c_str(); }

extern "C" int pb_search_create(int device, int nchan, int max_samples, float fch1_mhz, float foff_mhz, float tsamp_s,
                                float dm_min, float dm_max, float dm_step, int boxcar_max, const int *zap_ranges,
                                int nzap, pb_search **out)
{
    if (!out || nchan < 1 || max_samples < 64 || !(dm_step > 0) || dm_max < dm_min || boxcar_max < 1) {
        g_search_err = "pb_search_create: bad argument";
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
    s->dm_step = dm_step;
    s->ndm = (int)floor((dm_max - dm_min) / dm_step + 1e-6) + 1;
    s->nbox = 0;
    while ((1 << s->nbox) <= boxcar_max) s->nbox++;
    s->d_codes = s->d_xt = s->d_wid = nullptr;
    s->d_delay = nullptr;
    s->d_D = nullptr;
    s->d_stats = s->d_snr = nullptr;
    s->stream = nullptr;
    // delay table in double on the host; the top of the band (highest frequency) is the reference
    std::vector<int32_t> delay((size_t)s->ndm * nchan);
    const double ftop = foff_mhz < 0 ? fch1_mhz : fch1_mhz + (nchan - 1) * foff_mhz;
    int maxd = 0;
    for (int i = 0; i < s->ndm; ++i) {
        const double dm = (double)dm_min + (double)i * (double)dm_step;
        for (int c = 0; c < nchan; ++c) {
            bool zap = false;
            for (int z = 0; z < nzap; ++z)
                if (c >= zap_ranges[2 * z] && c < zap_ranges[2 * z + 1]) zap = true;
            const double f = (double)fch1_mhz + (double)c * (double)foff_mhz;
            const double d = 4.148808e3 * dm * (1.0 / (f * f) - 1.0 / (ftop * ftop)) / (double)tsamp_s;
            const int32_t di = (int32_t)floor(d + 0.5);
            delay[(size_t)i * nchan + c] = zap ? -1 : di;
            if (!zap && di > maxd) maxd = di;
        }
    }
    s->max_delay = maxd;
    s->tpitch = ((size_t)max_samples + 63) / 64 * 64;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_codes, (size_t)max_samples * nchan);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_xt, s->tpitch * nchan);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_delay, delay.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_D, (size_t)s->ndm * s->tpitch * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_stats, (size_t)s->ndm * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_snr, (size_t)s->ndm * s->tpitch * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_wid, (size_t)s->ndm * s->tpitch);
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
    void *p[] = {s->d_codes, s->d_xt, s->d_delay, s->d_D, s->d_stats, s->d_snr, s->d_wid};
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

// codes: nsamp x nchan samples of nbit bits in SIGPROC order (host, or device if codes_on_device).
// Outputs (host, any may be NULL): snr [ndm][tout] float, width_log2 [ndm][tout] u8,
// series [ndm][tout] u32 (the dedispersed sums), stats [ndm][2]; tout = nsamp - max_delay.
extern "C" int pb_search_run(pb_search *s, const void *codes, int codes_on_device, int nsamp, int nbit, float *snr,
                             uint8_t *width_log2, uint32_t *series, float *stats, int *tout_out)
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
    if (!codes_on_device) {
        SCHK(s, hipMemcpyAsync(s->d_codes, codes, nbytes, hipMemcpyHostToDevice, s->stream));
        d_codes = s->d_codes;
    }
    dim3 gt((nsamp + 63) / 64, (s->nchan + 63) / 64);
    k_transpose_codes<<<gt, 256, 0, s->stream>>>(d_codes, s->d_xt, nsamp, s->nchan, nbit, s->tpitch);
    dim3 gd((tout + 255) / 256, s->ndm);
    k_dedisperse<<<gd, 256, 0, s->stream>>>(s->d_xt, s->d_delay, s->d_D, s->nchan, tout, s->tpitch, s->tpitch);
    k_series_stats<<<s->ndm, 256, 0, s->stream>>>(s->d_D, s->d_stats, tout, s->tpitch);
    k_boxcar<<<gd, 256, 0, s->stream>>>(s->d_D, s->d_stats, s->d_snr, s->d_wid, tout, s->nbox, s->tpitch);
    SCHK(s, hipGetLastError());
    SCHK(s, hipStreamSynchronize(s->stream));
    if (snr) SCHK(s, hipMemcpy2D(snr, tout * sizeof(float), s->d_snr, s->tpitch * sizeof(float), tout * sizeof(float), s->ndm, hipMemcpyDeviceToHost));
    if (width_log2) SCHK(s, hipMemcpy2D(width_log2, tout, s->d_wid, s->tpitch, tout, s->ndm, hipMemcpyDeviceToHost));
    if (series) SCHK(s, hipMemcpy2D(series, tout * sizeof(uint32_t), s->d_D, s->tpitch * sizeof(uint32_t), tout * sizeof(uint32_t), s->ndm, hipMemcpyDeviceToHost));
    if (stats) SCHK(s, hipMemcpy(stats, s->d_stats, (size_t)s->ndm * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (tout_out) *tout_out = tout;
    return PB_OK;
}
