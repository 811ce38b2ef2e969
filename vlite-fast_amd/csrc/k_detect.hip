// Square-law detect + running-bandpass normalise + pol scrunch + time scrunch + requantise.
//
// One pass over the channelised data replaces, for each of the two output streams,
//   detect_and_normalize2 / 3   src/pb_kernels.cu:393-429 / :431-511
//   pscrunch / pscrunch_weights :514-524 / :527-560
//   tscrunch / tscrunch_weights :564-589 / :591-630
//   sel_and_dig_8b / 4b / 2b    :711-735 / :672-708 / :633-669
// restricted to the 4096 output channels (bins 2155..6250).
//
// The bandpass is a serial-in-time recurrence per (channel, pol, stream) with a clip
// branch in the excised stream (:493), so time cannot be parallelised without changing
// results.  Mapping: one wave = 32 channels x 2 pols of one stream; the partner pol's
// sample is one cross-lane exchange away, rows are streamed through registers CH rows
// ahead, and the GPU is filled by (channel groups) x (streams) x (antennas).
// HBM traffic: 4 B (power plane) or 8 B (complex plane) per (row, channel, pol, stream)
// read once; outputs are 1/64 of that.
#include "pb_internal.h"

struct DetectArgs {
    const void *in[2];       // per stream; float power [..][4096] or float2 spectra [..][6251]
    size_t ant_stride;       // elements
    size_t seg_stride;
    size_t pol_stride;
    size_t row_stride;
    int chan_off;            // 2155 for complex planes, 0 for compact power planes
    const float *wrow;       // [A][wrow_ant_stride]
    size_t wrow_ant_stride;
    float *bp;               // [A][2][2][4096]
    uint8_t *codes;          // [A][2][S][trim]
    float *ave;              // [A][2][S][ave_per_seg] or nullptr
    size_t trim, ave_per_seg;
    int S, R, nseg, npol, nbit, first_stream;
    float scale, oms, tscale;
};

template <bool IN_C64> struct InT { typedef float type; };
template <> struct InT<true> { typedef float2 type; };

template <bool IN_C64>
__device__ __forceinline__ float power_of(typename InT<IN_C64>::type v);
template <> __device__ __forceinline__ float power_of<true>(float2 v)
{
    const float xx = v.x * v.x;
    const float yy = v.y * v.y;
    return xx + yy;
}
template <> __device__ __forceinline__ float power_of<false>(float v) { return v; }

template <bool IN_C64, int CH>
__global__ __launch_bounds__(64) void k_detect(DetectArgs a)
{
    typedef typename InT<IN_C64>::type T;
    const int lane = threadIdx.x;
    const int pol = lane >> 5;
    const int c = blockIdx.x * 32 + (lane & 31);
    const int stream = a.first_stream + blockIdx.y;  // 0 = raw (dn2), 1 = excised (dn3)
    const int ant = blockIdx.z;
    const bool kur = stream == 1;
    const int R = a.R;
    const int cps = R / CH;  // chunks per segment

    const T *in = (const T *)a.in[stream] + (size_t)ant * a.ant_stride + (size_t)pol * a.pol_stride +
                  a.chan_off + c;
    const float *wrow = a.wrow + (size_t)ant * a.wrow_ant_stride;
    float *bpp = a.bp + (((size_t)ant * 2 + stream) * 2 + pol) * PB_NCHANOUT + c;
    uint8_t *codes = a.codes + ((size_t)ant * 2 + stream) * a.S * a.trim;
    float *ave = a.ave ? a.ave + ((size_t)ant * 2 + stream) * a.S * a.ave_per_seg : nullptr;
    const int ntime = R / PB_NSCRUNCH;

    float bp = *bpp;
    float acc = 0.f;
    int wt_sum = 0;
    float wt_sumf = 0.f;

    T cur[CH], nxt[CH];
    const int nchunk = a.nseg * cps;
#pragma unroll
    for (int j = 0; j < CH; ++j) nxt[j] = in[(size_t)j * a.row_stride];

    for (int k = 0; k < nchunk; ++k) {
        const int seg = k / cps, row0 = (k % cps) * CH;
#pragma unroll
        for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
        if (k + 1 < nchunk) {
            const int k1 = k + 1;
            const T *p = in + (size_t)(k1 / cps) * a.seg_stride + (size_t)((k1 % cps) * CH) * a.row_stride;
#pragma unroll
            for (int j = 0; j < CH; ++j) nxt[j] = p[(size_t)j * a.row_stride];
        }
        const float *wseg = wrow + (size_t)seg * R;

        if (row0 == 0 && bp == 0.f) {
            // initialise the bandpass from this segment's mean (:406-411, :444-461)
            const T *p = in + (size_t)seg * a.seg_stride;
            if (!kur) {
                for (int t = 0; t < R; ++t) bp += power_of<IN_C64>(p[(size_t)t * a.row_stride]);
                bp /= (float)R;
            } else {
                int good = 0;
                for (int t = 0; t < R; ++t) {
                    const float w = wseg[t];
                    if (w == 0.f) continue;
                    good++;
                    bp += power_of<IN_C64>(p[(size_t)t * a.row_stride]) / w;
                }
                if (good == 0) bp = 1.f;
                else bp /= (float)good;
            }
        }

#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int row = row0 + j;
            const float pw = power_of<IN_C64>(cur[j]);
            float w = 1.f;
            float x;
            if (!kur) {
                const float t1 = a.scale * pw;
                const float t2 = a.oms * bp;
                bp = t1 + t2;
                x = pw / bp - 1.f;
            } else {
                w = wseg[row];
                if (w == 0.f) {
                    x = 0.f;
                } else {
                    const float pk = (w == 1.f) ? pw : pw / w;
                    if (pk > bp * 11.f) {
                        x = 10.f;
                    } else {
                        const float t1 = a.scale * pk;
                        const float t2 = a.oms * bp;
                        bp = t1 + t2;
                        x = pk / bp - 1.f;
                    }
                }
            }
            // polarisation scrunch
            float p = x;
            float wt = w;  // weight tscrunch_weights sees for this row
            if (a.npol == 1) {
                const float xo = __shfl_xor(x, 32);
                const float s = x + xo;
                if (!kur) {
                    p = (float)(M_SQRT1_2 * (double)s);
                } else if ((double)w >= 0.2) {  // both pols carry the same row weight
                    p = (float)(M_SQRT1_2 * (double)s);
                    wt = (float)(0.5 * (double)(w + w));
                } else {
                    p = 0.f;
                    wt = 0.f;
                }
            }
            // time scrunch
            if (!kur) {
                acc += p;
            } else if (!((double)wt < 0.2)) {
                wt_sum++;
                wt_sumf += wt;
                const float prod = wt * p;
                acc += prod;
            }
            if ((row & (PB_NSCRUNCH - 1)) == PB_NSCRUNCH - 1) {
                if (!kur) {
                    acc *= a.tscale;
                } else {
                    if ((double)(wt_sumf / PB_NSCRUNCH) >= 0.2) acc /= sqrtf((float)wt_sum);
                    else acc = 0.f;
                }
                const int trow = row >> 3;
                // sample index within the segment, as sel_and_dig_* lays it out
                const size_t n = (a.npol == 1) ? (size_t)trow * PB_NCHANOUT + c
                                               : ((size_t)trow * 2 + pol) * PB_NCHANOUT + c;
                const bool writer = (a.npol == 2) || (pol == 0);
                if (ave && writer) {
                    const size_t ai = (a.npol == 1) ? n : ((size_t)pol * ntime + trow) * PB_NCHANOUT + c;
                    ave[(size_t)seg * a.ave_per_seg + ai] = acc;
                }
                uint8_t *cseg = codes + (size_t)seg * a.trim;
                if (a.nbit == 8) {
                    const float tmp = (float)((double)acc / 0.02957 + 127.5);
                    const uint8_t q = tmp <= 0 ? 0 : (tmp >= 255 ? 255 : (uint8_t)tmp);
                    if (writer) cseg[n] = q;
                } else if (a.nbit == 4) {
                    const float tmp = (float)((double)acc / 0.3188 + 7.5);
                    const unsigned q = tmp <= 0 ? 0u : (tmp >= 15 ? 15u : (unsigned)(uint8_t)tmp);
                    const unsigned hi = __shfl_down(q, 1);
                    if (writer && !(lane & 1)) cseg[n >> 1] = (uint8_t)(q | (hi << 4));
                } else {
                    const double t = (double)acc;
                    const unsigned q = t < -0.6109 ? 0u : (t < 0.3970 ? 1u : (t < 1.4050 ? 2u : 3u));
                    const unsigned q1 = __shfl_down(q, 1);
                    const unsigned q2 = __shfl_down(q, 2);
                    const unsigned q3 = __shfl_down(q, 3);
                    if (writer && !(lane & 3)) cseg[n >> 2] = (uint8_t)(q | (q1 << 2) | (q2 << 4) | (q3 << 6));
                }
                acc = 0.f;
                wt_sum = 0;
                wt_sumf = 0.f;
            }
        }
    }
    *bpp = bp;
}

// inject_frb on complex planes, src/pb_kernels.cu:348-391 (hipFFT back end; the LDS back
// end applies the same multiplications inside the channeliser before detection).
__global__ void k_inject_c64(float2 *__restrict__ X, size_t ant_stride, size_t seg_stride,
                             size_t pol_stride, const float *__restrict__ delays, int inject_now,
                             float width, float amp, int R)
{
    const int i = PB_CHANMIN + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PB_NCHAN) return;
    const int seg = blockIdx.y, ant = blockIdx.z;
    const int since = (inject_now - 1 + seg) * R;
    int lo = (int)(delays[i] + 0.5) - since;
    int hi = (int)(delays[i] + width + 0.5) - since;
    if (lo >= R) return;
    if (hi < 0) return;
    if (lo < 0) lo = 0;
    if (hi >= R) hi = R - 1;
    for (int pol = 0; pol < 2; ++pol) {
        float2 *p = X + (size_t)ant * ant_stride + (size_t)seg * seg_stride + (size_t)pol * pol_stride + i;
        for (int t = lo; t <= hi; ++t) {
            float2 v = p[(size_t)t * PB_NCHAN];
            v.x *= amp;
            v.y *= amp;
            p[(size_t)t * PB_NCHAN] = v;
        }
    }
}

hipError_t launch_inject_c64(pb_handle *h, int nseg, int inject_now)
{
    if (inject_now <= 0 || !h->d_frb_delays) return hipSuccess;
    const size_t pol_stride = (size_t)h->R * PB_NCHAN;
    const size_t seg_stride = 2 * pol_stride;
    const size_t ant_stride = (size_t)h->S * seg_stride;
    const float width = h->frb_width;
    dim3 grid((PB_NCHANOUT + 255) / 256, nseg, h->A);
    if (h->cfg.rfi_mode != 1)
        k_inject_c64<<<grid, 256, 0, h->stream>>>(h->d_Xraw, ant_stride, seg_stride, pol_stride,
                                                  h->d_frb_delays, inject_now, width, h->frb_amp, h->R);
    if (h->cfg.rfi_mode != 0)
        k_inject_c64<<<grid, 256, 0, h->stream>>>(h->d_Xkur, ant_stride, seg_stride, pol_stride,
                                                  h->d_frb_delays, inject_now, width, h->frb_amp, h->R);
    return hipGetLastError();
}

// Filterbank bytes to the pinned host mirror.  A kernel storing straight into the mapped host buffer
// instead of hipMemcpyAsync: on this stack an asynchronous D2H copy blocks the calling host thread for
// 5-7 ms every twenty-odd calls (measured with PB_TRACE-style timers around each runtime call), which
// is six batches' worth of GPU time.  10 MB per second of data per antenna is nothing for PCIe.
__global__ __launch_bounds__(256) void k_copy_out(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

hipError_t launch_copy_out(uint8_t *host_pinned, const uint8_t *dev, size_t nbytes, hipStream_t st)
{
    if (nbytes == 0) return hipSuccess;
    if ((nbytes & 15) || ((uintptr_t)host_pinned & 15) || ((uintptr_t)dev & 15))
        return hipMemcpyAsync(host_pinned, dev, nbytes, hipMemcpyDeviceToHost, st);
    void *dptr = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dptr, host_pinned, 0);
    if (e != hipSuccess) return e;
    const size_t n16 = nbytes >> 4;
    // 8 workgroups saturate PCIe (10 MB in ~0.2 ms); more only hold store queues of more CUs full,
    // which slows the channeliser running beside them (32: +10 % on the step).  PB_COPY_WGS overrides.
    static const int maxb = getenv("PB_COPY_WGS") ? atoi(getenv("PB_COPY_WGS")) : 8;
    const unsigned nb = (unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)maxb);
    k_copy_out<<<nb, 256, 0, st>>>((uint4 *)dptr, (const uint4 *)dev, n16);
    return hipGetLastError();
}

hipError_t launch_detect(pb_handle *h, int nseg, int)
{
    if (h->cfg.fft_backend != PB_FFT_HIPFFT) return launch_detect_pow(h, nseg);
    DetectArgs a;
    a.in[0] = h->d_Xraw;
    a.in[1] = h->d_Xkur;
    a.row_stride = PB_NCHAN;
    a.chan_off = PB_CHANMIN;
    a.pol_stride = (size_t)h->R * a.row_stride;
    a.seg_stride = 2 * a.pol_stride;
    a.ant_stride = (size_t)h->S * a.seg_stride;
    a.wrow = h->d_wrow;
    a.wrow_ant_stride = (size_t)h->S * h->R;
    a.bp = h->d_bp;
    a.codes = h->d_codes;
    a.ave = h->cfg.keep_ave ? h->d_ave : nullptr;
    a.trim = h->trim;
    a.ave_per_seg = h->ave_per_seg;
    a.S = h->S;
    a.R = h->R;
    a.nseg = nseg;
    a.npol = h->cfg.npol;
    a.nbit = h->cfg.nbit;
    a.first_stream = h->cfg.rfi_mode == 1 ? 1 : 0;
    const double tsamp = (double)PB_NFFT / 128000000 * PB_NSCRUNCH;  // src/process_baseband.cu:739-741
    a.scale = (float)(tsamp / 1.0);
    a.oms = 1 - a.scale;
    a.tscale = (float)sqrt(1. / PB_NSCRUNCH);
    const int nstreams = h->cfg.rfi_mode == 2 ? 2 : 1;
    dim3 grid(PB_NCHANOUT / 32, nstreams, h->A);
    if ((h->R % 32) == 0) k_detect<true, 32><<<grid, 64, 0, h->stream>>>(a);
    else k_detect<true, 8><<<grid, 64, 0, h->stream>>>(a);
    return hipGetLastError();
}

// ---- incoherent coadd helpers (pb_coadd_local / pb_coadd_finish) ----
__global__ void k_coadd_local(const float *__restrict__ ave, size_t ant_stride, int A,
                              float *__restrict__ sum, size_t n, int accumulate)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        float s = accumulate ? sum[i] : 0.f;
        for (int a = 0; a < A; ++a) s += ave[(size_t)a * ant_stride + i];
        sum[i] = s;
    }
}

hipError_t launch_coadd_local(pb_handle *h, int nseg, float *d_sum, int accumulate, hipStream_t st)
{
    const int stream = h->cfg.rfi_mode == 0 ? 0 : 1;
    const size_t n = (size_t)nseg * h->ave_per_seg;
    const float *ave = h->d_ave + (size_t)stream * h->S * h->ave_per_seg;
    k_coadd_local<<<1024, 256, 0, st>>>(ave, (size_t)2 * h->S * h->ave_per_seg, h->A, d_sum, n,
                                               accumulate);
    return hipGetLastError();
}

// sel_and_dig on a compact [time][4096] (npol 1) or [pol][time][4096] (npol 2) plane
__global__ void k_coadd_digitise(const float *__restrict__ sum, float scale, uint8_t *__restrict__ codes,
                                 size_t nsamp_seg, size_t trim, int nseg, int npol, int nbit, int ntime)
{
    const size_t per = 8 / nbit;
    const size_t nbytes = (size_t)nseg * trim;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nbytes;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t seg = i / trim, ib = i % trim;
        unsigned byte = 0;
        for (size_t j = 0; j < per; ++j) {
            const size_t n = ib * per + j;  // sample index in sel_and_dig order
            size_t src;
            if (npol == 1) {
                src = n;
            } else {
                const size_t trow = n / (2 * PB_NCHANOUT), rem = n % (2 * PB_NCHANOUT);
                const size_t pol = rem / PB_NCHANOUT, c = rem % PB_NCHANOUT;
                src = (pol * ntime + trow) * PB_NCHANOUT + c;
            }
            const float v = sum[seg * nsamp_seg + src] * scale;
            unsigned q;
            if (nbit == 8) {
                const float tmp = (float)((double)v / 0.02957 + 127.5);
                q = tmp <= 0 ? 0u : (tmp >= 255 ? 255u : (unsigned)(uint8_t)tmp);
            } else if (nbit == 4) {
                const float tmp = (float)((double)v / 0.3188 + 7.5);
                q = tmp <= 0 ? 0u : (tmp >= 15 ? 15u : (unsigned)(uint8_t)tmp);
            } else {
                const double t = (double)v;
                q = t < -0.6109 ? 0u : (t < 0.3970 ? 1u : (t < 1.4050 ? 2u : 3u));
            }
            byte |= q << (nbit * j);
        }
        codes[i] = (uint8_t)byte;
    }
}

hipError_t launch_coadd_digitise(pb_handle *h, int nseg, const float *d_sum, float scale, uint8_t *d_codes,
                                 hipStream_t st)
{
    k_coadd_digitise<<<512, 256, 0, st>>>(d_sum, scale, d_codes, h->ave_per_seg, h->trim, nseg,
                                                 h->cfg.npol, h->cfg.nbit, h->R / PB_NSCRUNCH);
    return hipGetLastError();
}
