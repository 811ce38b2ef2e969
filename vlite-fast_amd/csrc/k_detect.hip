// hipFFT back end: square-law detect of the complex spectra into the compact power planes that
// k_detect2 consumes (the same planes the LDS channeliser writes), FRB injection on complex planes,
// the device-to-pinned-host copy kernel and the incoherent-sum helpers.
//
//   detect_and_normalize2 / 3 (detect part)  src/pb_kernels.cu:393-429 / :431-511
//   inject_frb                                :347-391
#include "pb_internal.h"

// P[row][c] = |X[row][2155 + c]|^2 (raw stream) or that over the row weight (excised stream; +inf
// for rows of weight 0, the convention of k_channelize.hip / k_detect2.hip).  One thread = four
// channels: 8-byte loads (rows of 6251 complex are only 8-byte aligned), one 16-byte store.
__global__ __launch_bounds__(256) void k_power_c64(const float2 *__restrict__ X, float *__restrict__ P,
                                                   const float *__restrict__ wrow, size_t nrowpol, int R, int kur)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // (rowpol, quad of channels)
    const size_t rp = i / (PB_NCHANOUT / 4);
    if (rp >= nrowpol) return;
    const int c4 = (int)(i % (PB_NCHANOUT / 4)) * 4;
    const float2 *x = X + rp * PB_NCHAN + PB_CHANMIN + c4;
    float pw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2 v = x[q];
        const float xx = v.x * v.x;
        const float yy = v.y * v.y;
        pw[q] = xx + yy;
    }
    if (kur) {
        // rp = ((ant * S + seg) * 2 + pol) * R + row  ->  weight index (ant * S + seg) * R + row
        const size_t as = rp / ((size_t)2 * R);
        const float w = wrow[as * R + rp % R];
#pragma unroll
        for (int q = 0; q < 4; ++q) pw[q] = w == 0.f ? __builtin_inff() : pw[q] / w;
    }
    *(float4 *)(P + rp * PB_NCHANOUT + c4) = make_float4(pw[0], pw[1], pw[2], pw[3]);
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

hipError_t launch_copy_out(const PbSched &sched, uint8_t *host_pinned, const uint8_t *dev, size_t nbytes, hipStream_t st)
{
    if (nbytes == 0) return hipSuccess;
    if (sched.copy_dma || (nbytes & 15) || ((uintptr_t)host_pinned & 15) || ((uintptr_t)dev & 15))
        return hipMemcpyAsync(host_pinned, dev, nbytes, hipMemcpyDeviceToHost, st);
    void *dptr = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dptr, host_pinned, 0);
    if (e != hipSuccess) return e;
    const size_t n16 = nbytes >> 4;
    // 8 workgroups saturate PCIe (10 MB in ~0.2 ms); more only hold store queues of more CUs full,
    // which slows the channeliser running beside them (32: +10 % on the step).  PB_COPY_WGS overrides.
    const unsigned nb = (unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)(sched.copy_wgs > 0 ? sched.copy_wgs : 8));
    k_copy_out<<<nb, 256, 0, st>>>((uint4 *)dptr, (const uint4 *)dev, n16);
    return hipGetLastError();
}

hipError_t launch_detect(pb_handle *h, int nseg, int)
{
    if (h->cfg.fft_backend == PB_FFT_HIPFFT) {
        // complex spectra -> power planes of the segments just transformed (all antennas at once when
        // the whole handle is processed, else antenna by antenna: planes are [A][S][2][R][...])
        const size_t per_ant = (size_t)nseg * 2 * h->R;
        for (int st = 0; st < 2; ++st) {
            if (st == 0 ? h->cfg.rfi_mode == 1 : h->cfg.rfi_mode == 0) continue;
            const float2 *X = st ? h->d_Xkur : h->d_Xraw;
            float *P = st ? h->d_Pkur : h->d_Praw;
            for (int ant = 0; ant < h->A; ++ant) {
                const size_t rp0 = (size_t)ant * h->S * 2 * h->R;
                const size_t nthreads = per_ant * (PB_NCHANOUT / 4);
                k_power_c64<<<(unsigned)((nthreads + 255) / 256), 256, 0, h->stream>>>(
                    X + rp0 * PB_NCHAN, P + rp0 * PB_NCHANOUT, h->d_wrow + (size_t)ant * h->S * h->R, per_ant, h->R, st);
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return launch_detect_pow(h, nseg);
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

// the common case (8 bits, one pol: codes[i] = q(sum[i] * scale)), four samples per thread and step: one 16-byte
// load, one 4-byte store.  (Not straight into the page-locked host mirror: 256 workgroups waiting on PCIe writes
// would each hold a CU slot that a channeliser workgroup wants; the 8-workgroup copy kernel does that.)
__global__ __launch_bounds__(256) void k_coadd_digitise8(const float4 *__restrict__ sum, float scale,
                                                         uint32_t *__restrict__ codes, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = sum[i];
        const float x[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
        unsigned word = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float tmp = (float)((double)x[j] / 0.02957 + 127.5);
            const unsigned q = tmp <= 0 ? 0u : (tmp >= 255 ? 255u : (unsigned)(uint8_t)tmp);
            word |= q << (8 * j);
        }
        codes[i] = word;
    }
}

// a flat range of the npol = 1 plane: sample i of the plane is sample i of the code stream
hipError_t launch_coadd_digitise_flat(pb_handle *h, const float *d_sum, size_t nfloat, float scale, uint8_t *d_codes,
                                      hipStream_t st)
{
    const size_t nbytes = nfloat * h->cfg.nbit / 8;
    if (h->cfg.nbit == 8 && (nfloat & 3) == 0) {
        k_coadd_digitise8<<<256, 256, 0, st>>>((const float4 *)d_sum, scale, (uint32_t *)d_codes, nfloat / 4);
        return hipGetLastError();
    }
    k_coadd_digitise<<<512, 256, 0, st>>>(d_sum, scale, d_codes, nfloat, nbytes, 1, 1, h->cfg.nbit, 0);
    return hipGetLastError();
}

hipError_t launch_coadd_digitise(pb_handle *h, int nseg, const float *d_sum, float scale, uint8_t *d_codes,
                                 hipStream_t st)
{
    if (h->cfg.nbit == 8 && h->cfg.npol == 1 && h->ave_per_seg == h->trim && (h->trim & 3) == 0) {
        const size_t n4 = (size_t)nseg * h->trim / 4;
        k_coadd_digitise8<<<256, 256, 0, st>>>((const float4 *)d_sum, scale, (uint32_t *)d_codes, n4);
        return hipGetLastError();
    }
    k_coadd_digitise<<<512, 256, 0, st>>>(d_sum, scale, d_codes, h->ave_per_seg, h->trim, nseg,
                                                 h->cfg.npol, h->cfg.nbit, h->R / PB_NSCRUNCH);
    return hipGetLastError();
}
