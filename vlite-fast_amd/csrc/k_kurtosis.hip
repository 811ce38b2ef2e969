// Spectral-kurtosis RFI flagging over 500-sample blocks, fused with the 8-bit unpack.
//
// One workgroup = one 500-sample block index, two waves = the two polarisations, because
// the flag is the max over both pols of the D'Agostino score.  A single pass over the raw
// bytes replaces four reference kernels:
//   convertarray       src/pb_kernels.cu:23-33      (u==0 -> 0, else u/128-1)
//   kurtosis           :35-107                      (fixed halving tree, restated for wave64)
//   compute_dagostino  :109-134                     (double/float mix, max over pols)
//   apply_kurtosis     :243-295                     (zero flagged blocks; weights)
// HBM traffic: 1 B/sample read; +1 B per 500 samples of flags written; in the hipFFT
// back end additionally 4 (+4) B/sample of fp32 voltages written.
#include "pb_internal.h"

__device__ __forceinline__ float cvt_sample(unsigned u)
{
    return u == 0 ? 0.0f : (float)u / 128 - 1;
}

// t^(float(1/3)) by Newton cube root in IEEE double: the same operation sequence as
// orc_powf_third in the oracle, so that flags agree bit for bit (deviation 1).
__device__ float dev_powf_third(float t)
{
    double d = (double)t;
    unsigned long long bits = (unsigned long long)__double_as_longlong(d);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    int q = (e >= 0) ? e / 3 : -((-e + 2) / 3);
    int r = e - 3 * q;
    bits = (bits & 0x000fffffffffffffULL) | ((unsigned long long)(1023 + r) << 52);
    double m = __longlong_as_double((long long)bits);
    double y = 1.0 + (m - 1.0) * (1.0 / 7.0);
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        double y2 = y * y;
        y = y - (y2 * y - m) / (3.0 * y2);
    }
    double s = __longlong_as_double((long long)((unsigned long long)(1023 + q) << 52));
    double z = (y - 1.0) / (y + 1.0);
    double z2 = z * z;
    double lny = 2.0 * z * (1.0 + z2 * (1.0 / 3.0 + z2 * (1.0 / 5.0 + z2 * (1.0 / 7.0))));
    double lnt = (double)(3 * q) * 0.69314718055994531 + 3.0 * lny;
    const double dexp = (double)(float)(1. / 3) - 1. / 3;
    return (float)(y * s * (1.0 + dexp * lnt));
}

__device__ float dag_one(float kur, const DagConsts &c)
{
    float dag = 9.0f;  // DAG_INF = DAG_THRESH + DAG_FB_THRESH + 1
    if (kur != 0.f) {  // true for NaN (all-zero block): t is NaN, t > 0 false, stays DAG_INF
        float t = (float)(c.one_m_2_over_A / (1. + ((double)kur - 3. - c.mu1) * c.Z3));
        if (t > 0) dag = fabsf((float)(c.Z1 * (c.Z2 - (double)dev_powf_third(t))));
    }
    return dag;
}

__device__ __forceinline__ float4 cvt4(uint32_t w)
{
    float4 f;
    f.x = cvt_sample(w & 0xff);
    f.y = cvt_sample((w >> 8) & 0xff);
    f.z = cvt_sample((w >> 16) & 0xff);
    f.w = cvt_sample(w >> 24);
    return f;
}

template <bool WRITE_F32>
__global__ __launch_bounds__(128) void k_kurtosis_flag(
    const uint8_t *__restrict__ in, size_t in_ant_stride, size_t seg_samples, int blk_per_seg,
    uint8_t *__restrict__ flags, size_t flags_ant_stride,
    float *__restrict__ stats, size_t nblk_cap,
    float *__restrict__ fraw, float *__restrict__ fkur, int write_raw, DagConsts dc)
{
    __shared__ uint32_t sbytes[2][128];
    __shared__ float sdag[2];
    const int wave = threadIdx.x >> 6;  // = polarisation
    const int lane = threadIdx.x & 63;
    const size_t b = blockIdx.x;
    const int ant = blockIdx.y;
    const size_t seg = b / (size_t)blk_per_seg, bi = b % (size_t)blk_per_seg;
    const size_t off = (seg * 2 + wave) * seg_samples + bi * PB_NKURTO;  // sample index in the antenna
    const uint32_t *src32 = (const uint32_t *)(in + (size_t)ant * in_ant_stride + off);

    const uint32_t w0 = src32[lane];
    const uint32_t w1 = (lane + 64 < 125) ? src32[lane + 64] : 0u;
    sbytes[wave][lane] = w0;
    sbytes[wave][lane + 64] = w1;
    __syncthreads();

    // leaves t = lane + 64 i hold (x[t]^2, x[t+250]^2); slots 250..255 are zero
    const uint8_t *sb = (const uint8_t *)sbytes[wave];
    float d2[4], d4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = lane + 64 * i;
        if (t < 250) {
            const float x0 = cvt_sample(sb[t]);
            const float x1 = cvt_sample(sb[t + 250]);
            const float a = x0 * x0;
            const float tm = x1 * x1;
            const float a2 = a * a;
            const float t2 = tm * tm;
            d4[i] = a2 + t2;
            d2[i] = a + tm;
        } else {
            d2[i] = 0.f;
            d4[i] = 0.f;
        }
    }
    // halving tree 128, 64 in registers, then 32..1 across the wave: d[t] += d[t+s]
    float s2 = (d2[0] + d2[2]) + (d2[1] + d2[3]);
    float s4 = (d4[0] + d4[2]) + (d4[1] + d4[3]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        s2 = s2 + __shfl_down(s2, s);
        s4 = s4 + __shfl_down(s4, s);
    }
    if (lane == 0) {
        const float p = s2 / PB_NKURTO;
        const float k = s4 / PB_NKURTO / (p * p);
        const float dg = dag_one(k, dc);
        sdag[wave] = dg;
        if (stats) {
            const size_t ab = (size_t)ant * 6 * nblk_cap;
            stats[ab + (0 * 2 + wave) * nblk_cap + b] = p;
            stats[ab + (1 * 2 + wave) * nblk_cap + b] = k;
        }
    }
    __syncthreads();
    const float dmax = fmaxf(sdag[0], sdag[1]);
    const bool bad = dmax > 3.0f;  // DAG_THRESH
    if (lane == 0) {
        if (stats) stats[(size_t)ant * 6 * nblk_cap + (2 * 2 + wave) * nblk_cap + b] = dmax;
        if (wave == 0) flags[(size_t)ant * flags_ant_stride + b] = bad ? 1 : 0;
    }
    if (WRITE_F32) {
        const size_t fo = (size_t)ant * in_ant_stride + off;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v0 = cvt4(w0);
        if (write_raw) ((float4 *)(fraw + fo))[lane] = v0;
        ((float4 *)(fkur + fo))[lane] = bad ? z : v0;
        if (lane + 64 < 125) {
            const float4 v1 = cvt4(w1);
            if (write_raw) ((float4 *)(fraw + fo))[lane + 64] = v1;
            ((float4 *)(fkur + fo))[lane + 64] = bad ? z : v1;
        }
    }
}

// plain unpack for rfi_mode 0 (convertarray only)
__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ in,
                                                float *__restrict__ out, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x)
        ((float4 *)out)[i] = cvt4(((const uint32_t *)in)[i]);
}

hipError_t launch_kurtosis_flag(pb_handle *h, int nseg, bool write_f32)
{
    const size_t in_ant_stride = (size_t)h->S * 2 * h->seg_samples;
    const size_t nblk_cap = (size_t)h->S * h->nblk_seg;
    if (h->cfg.rfi_mode == 0) {
        if (!write_f32) return hipSuccess;
        for (int a = 0; a < h->A; ++a) {
            const size_t n4 = (size_t)nseg * 2 * h->seg_samples / 4;
            k_unpack<<<2048, 256, 0, h->stream>>>(h->d_in + a * in_ant_stride,
                                                  h->d_fraw + a * in_ant_stride, n4);
        }
        return hipGetLastError();
    }
    dim3 grid((unsigned)((size_t)nseg * h->nblk_seg), (unsigned)h->A);
    if (write_f32)
        k_kurtosis_flag<true><<<grid, 128, 0, h->stream>>>(
            h->d_in, in_ant_stride, h->seg_samples, (int)h->nblk_seg, h->d_flags, nblk_cap,
            h->d_stats, nblk_cap, h->d_fraw, h->d_fkur, h->cfg.rfi_mode == 2 ? 1 : 0, h->dag);
    else
        k_kurtosis_flag<false><<<grid, 128, 0, h->stream>>>(
            h->d_in, in_ant_stride, h->seg_samples, (int)h->nblk_seg, h->d_flags, nblk_cap,
            h->d_stats, nblk_cap, nullptr, nullptr, 0, h->dag);
    return hipGetLastError();
}

// kur_weights after apply_kurtosis (src/pb_kernels.cu:292: one atomicAdd of 500/12500 per
// unflagged block): k identical float additions give the same sum in any order, so the
// row weight is a table lookup on the unflagged count.  The flag is shared by both pols,
// hence kur_weights[t] == kur_weights[t + FFTS_PER_SEG].
__constant__ float c_wtab[26];

__global__ void k_row_weights(const uint8_t *__restrict__ flags, float *__restrict__ wrow,
                              size_t rows, size_t ant_stride_rows)
{
    const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int ant = blockIdx.y;
    if (r >= rows) return;
    const uint8_t *f = flags + (size_t)ant * ant_stride_rows * PB_BLK_PER_FFT + r * PB_BLK_PER_FFT;
    int good = 0;
#pragma unroll
    for (int k = 0; k < PB_BLK_PER_FFT; ++k) good += f[k] ? 0 : 1;
    wrow[(size_t)ant * ant_stride_rows + r] = c_wtab[good];
}

hipError_t launch_row_weights(pb_handle *h, int nseg)
{
    static bool tab_ready = false;
    if (!tab_ready) {
        float tab[26];
        const float inc = (float)PB_NKURTO / PB_NFFT;
        float w = 0.f;
        tab[0] = 0.f;
        for (int k = 1; k <= 25; ++k) {
            w = w + inc;
            tab[k] = w;
        }
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_wtab), tab, sizeof tab);
        if (e != hipSuccess) return e;
        tab_ready = true;
    }
    const size_t rows = (size_t)nseg * h->R;
    dim3 grid((unsigned)((rows + 255) / 256), (unsigned)h->A);
    k_row_weights<<<grid, 256, 0, h->stream>>>(h->d_flags, h->d_wrow, rows, (size_t)h->S * h->R);
    return hipGetLastError();
}

// Gather 5000-byte VDIF payloads of one raw ring block into the pol-planar segment layout
// (replaces the host loop src/process_baseband.cu:1015-1067).  idx[thread][frame] = slot of
// that frame in the block, or -1 (zero fill).
__global__ __launch_bounds__(256) void k_deframe(const uint8_t *__restrict__ block,
                                                 const int32_t *__restrict__ idx,
                                                 uint8_t *__restrict__ dst_ant, size_t nframes,
                                                 size_t seg_samples, int seg0)
{
    const size_t f = blockIdx.x;
    const int pol = blockIdx.y;
    const int32_t slot = idx[(size_t)pol * nframes + f];
    const size_t sample = f * PB_VDIF_DATA;
    const size_t seg = sample / seg_samples, within = sample % seg_samples;
    uint2 *dst = (uint2 *)(dst_ant + ((size_t)(seg0 + seg) * 2 + pol) * seg_samples + within);
    if (slot < 0) {
        for (int i = threadIdx.x; i < PB_VDIF_DATA / 8; i += blockDim.x) dst[i] = make_uint2(0, 0);
    } else {
        const uint2 *src = (const uint2 *)(block + (size_t)slot * PB_VDIF_FRAME + 32);
        for (int i = threadIdx.x; i < PB_VDIF_DATA / 8; i += blockDim.x) dst[i] = src[i];
    }
}

hipError_t launch_deframe(pb_handle *h, int ant, int seg0, size_t nframes_per_thread)
{
    dim3 grid((unsigned)nframes_per_thread, 2);
    uint8_t *dst = h->d_in + (size_t)ant * h->S * 2 * h->seg_samples;
    k_deframe<<<grid, 256, 0, h->stream>>>(h->d_vdif, h->d_frame_idx, dst, nframes_per_thread,
                                           h->seg_samples, seg0);
    return hipGetLastError();
}
