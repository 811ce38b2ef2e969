// Spectral-kurtosis RFI flagging over 500-sample blocks, fused with the 8-bit unpack.
//
// One workgroup = one FFT row (12500 samples) of BOTH polarisations = 50 kurtosis blocks,
// because the flag of a block is the max over the two pols of the D'Agostino score and the
// weight of a row counts its unflagged blocks.  A single pass over the raw bytes replaces
//   convertarray       src/pb_kernels.cu:23-33      (u==0 -> 0, else u/128-1)
//   kurtosis           :35-107                      (fixed halving tree, restated for wave64)
//   compute_dagostino  :109-134                     (double/float mix, max over pols)
//   apply_kurtosis     :243-295                     (flags; weights; zeroing in the hipFFT path)
// Structure: the two 12.5 KB rows are staged in LDS with 16-byte loads; each wave reduces
// blocks (4 leaves per lane + the reference's halving tree across the wave); then 50 lanes
// evaluate pow, kur and the double-precision D'Agostino score side by side instead of one
// lane per block.
// HBM traffic: 1 B/sample read; +1 B per 500 samples of flags written; in the hipFFT
// back end additionally 4 (+4) B/sample of fp32 voltages written.
#include "pb_internal.h"

typedef float f2k __attribute__((ext_vector_type(2)));

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

__device__ __forceinline__ unsigned fix0(unsigned w)
{
    const unsigned t = ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w;   // bit 7 of a byte set <=> byte != 0
    return w | (~t & 0x80808080u);
}
__device__ __forceinline__ uint4 fix0(uint4 q) { return make_uint4(fix0(q.x), fix0(q.y), fix0(q.z), fix0(q.w)); }

// r[lane] += r[lane + s] for s = 8, 4, 2, 1 inside a row of 16 lanes (DPP row_shl: lane i reads lane
// i + s of its row); only lanes < s of the row hold meaningful sums afterwards, lane 0 the total --
// the same additions in the same order as the reference's halving tree (kurtosis :60-94).
template <int S> __device__ __forceinline__ float add_row_shl(float r)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x100 + S, 0xf, 0xf, true);
    return r + __int_as_float(t);
}

#define ROW_CHUNKS 784   // 16-byte chunks covering a 12500-byte row at any 4-byte alignment

template <bool WRITE_F32>
__global__ __launch_bounds__(256) void k_kurtosis_row(
    const uint8_t *__restrict__ in, size_t in_ant_stride, size_t seg_samples, int R,
    uint8_t *__restrict__ flags, size_t flags_ant_stride, float *__restrict__ wrow_out,
    uint32_t *__restrict__ rowmask_out, size_t wrow_ant_stride, float *__restrict__ stats, size_t nblk_cap,
    float *__restrict__ fraw, float *__restrict__ fkur, int write_raw, DagConsts dc, DagConsts dc_fb,
    float *__restrict__ stats_fb, size_t nrow_cap)
{
    __shared__ uint4 sraw[2][ROW_CHUNKS];
    __shared__ float s2[50], s4[50], sdag[50];
    __shared__ unsigned sflag[25];
    __shared__ float spow[50], skur[50], sfb[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: block indices stay in scalar registers
    const int grow = blockIdx.x;  // seg * R + row
    const int ant = blockIdx.y;
    const int seg = grow / R, row = grow % R;

    // (no runtime-indexed private arrays: they would live in scratch memory)
    const size_t rbyte0 = (size_t)ant * in_ant_stride + ((size_t)seg * 2 + 0) * seg_samples + (size_t)row * PB_NFFT;
    const size_t rbyte1 = rbyte0 + seg_samples;
    const unsigned off0 = (unsigned)(rbyte0 & 15), off1 = (unsigned)(rbyte1 & 15);
    {
        // 782 or 783 aligned 16-byte chunks cover a row: tid, tid+256, tid+512 always exist
        const uint4 *s0 = (const uint4 *)(in + (rbyte0 - off0));
        const uint4 *s1 = (const uint4 *)(in + (rbyte1 - off1));
        const bool h0 = tid + 768 < (int)((off0 + PB_NFFT + 15) >> 4);
        const bool h1 = tid + 768 < (int)((off1 + PB_NFFT + 15) >> 4);
        const uint4 a0 = s0[tid], a1 = s0[tid + 256], a2 = s0[tid + 512];
        const uint4 b0 = s1[tid], b1 = s1[tid + 256], b2 = s1[tid + 512];
        uint4 a3 = make_uint4(0u, 0u, 0u, 0u), b3 = a3;
        if (h0) a3 = s0[tid + 768];
        if (h1) b3 = s1[tid + 768];
        // code 0 ("no sample" -> 0.0, convertarray :23-33) is rewritten to code 128 (= 0.0) four bytes
        // per instruction, so that the conversion below is one fma per pair: u/128 - 1 is exact
        sraw[0][tid] = fix0(a0); sraw[0][tid + 256] = fix0(a1); sraw[0][tid + 512] = fix0(a2);
        sraw[1][tid] = fix0(b0); sraw[1][tid + 256] = fix0(b1); sraw[1][tid + 512] = fix0(b2);
        if (h0) sraw[0][tid + 768] = fix0(a3);
        if (h1) sraw[1][tid + 768] = fix0(b3);
    }
    __syncthreads();

    // each wave reduces blocks wave, wave+4, ...: leaves t = lane + 64 i hold (x[t]^2, x[t+250]^2),
    // the pair side by side in packed-f32 instructions
    for (int bi = wave; bi < 50; bi += 4) {
        const int pol = bi / 25, blk = bi % 25;
        const uint8_t *sb = (const uint8_t *)sraw[pol] + (pol ? off1 : off0) + blk * PB_NKURTO;
        float d2[4], d4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = lane + 64 * i;
            const bool in = t < 250;
            const int tt = in ? t : 0;
            f2k u;
            u.x = (float)sb[tt];
            u.y = (float)sb[tt + 250];
            const f2k k128 = {0.0078125f, 0.0078125f}, m1 = {-1.0f, -1.0f};
            const f2k x = __builtin_elementwise_fma(u, k128, m1);
            const f2k a = x * x;
            const f2k a2 = a * a;
            const float e4 = a2.x + a2.y, e2 = a.x + a.y;
            d4[i] = in ? e4 : 0.f;
            d2[i] = in ? e2 : 0.f;
        }
        // halving tree 128, 64 in registers, then 32..1 across the wave: d[t] += d[t+s]
        float r2 = (d2[0] + d2[2]) + (d2[1] + d2[3]);
        float r4 = (d4[0] + d4[2]) + (d4[1] + d4[3]);
        r2 = r2 + __shfl_down(r2, 32);
        r4 = r4 + __shfl_down(r4, 32);
        r2 = r2 + __shfl_down(r2, 16);
        r4 = r4 + __shfl_down(r4, 16);
        r2 = add_row_shl<8>(r2);
        r4 = add_row_shl<8>(r4);
        r2 = add_row_shl<4>(r2);
        r4 = add_row_shl<4>(r4);
        r2 = add_row_shl<2>(r2);
        r4 = add_row_shl<2>(r4);
        r2 = add_row_shl<1>(r2);
        r4 = add_row_shl<1>(r4);
        if (lane == 0) {
            s2[bi] = r2;
            s4[bi] = r4;
        }
    }
    __syncthreads();
    const size_t b0 = (size_t)grow * PB_BLK_PER_FFT;   // first block index of this row (per pol)
    if (tid < 50) {
        const int pol = tid / 25, blk = tid % 25;
        const float p = s2[tid] / PB_NKURTO;
        const float k = s4[tid] / PB_NKURTO / (p * p);
        sdag[tid] = dag_one(k, dc);
        if (stats_fb) {
            spow[tid] = p;
            skur[tid] = k;
        }
        if (stats) {
            const size_t ab = (size_t)ant * 6 * nblk_cap;
            stats[ab + (0 * 2 + pol) * nblk_cap + b0 + blk] = p;
            stats[ab + (1 * 2 + pol) * nblk_cap + b0 + blk] = k;
        }
    }
    __syncthreads();
    if (tid < 25) {
        const float dmax = fmaxf(sdag[tid], sdag[25 + tid]);
        const bool bad = dmax > 3.0f;  // DAG_THRESH
        sflag[tid] = bad ? 1u : 0u;
        flags[(size_t)ant * flags_ant_stride + b0 + tid] = bad ? 1 : 0;
        if (stats) {
            const size_t ab = (size_t)ant * 6 * nblk_cap;
            stats[ab + (2 * 2 + 0) * nblk_cap + b0 + tid] = dmax;
            stats[ab + (2 * 2 + 1) * nblk_cap + b0 + tid] = dmax;
        }
    }
    __syncthreads();
    if (stats_fb) {
        // K4: block_kurtosis + compute_dagostino2 (src/pb_kernels.cu:140-241), the statistic over the whole
        // FFT row.  No output depends on it (apply_kurtosis ignores dag_fb, :255-256), so it is only
        // computed when the statistics are kept (debug_keep): one lane per pol walks the reference's
        // 32-slot halving tree as written (25 live slots, lanes 0..15 stay live).
        if (tid < 2) {
            float d2[32], d4[32];
            unsigned wt[32];
            for (int l = 0; l < 32; ++l) {
                if (l > 24) {
                    d2[l] = 0.f; d4[l] = 0.f; wt[l] = 0;
                    continue;
                }
                const float dmax = fmaxf(sdag[l], sdag[25 + l]);
                const float pw = spow[tid * 25 + l], ku = skur[tid * 25 + l];
                wt[l] = dmax < 3.0f ? 1u : 0u;
                const float wf = (float)wt[l];
                d2[l] = wf * pw;
                d4[l] = wf * ku * pw * pw;
            }
            for (int sft = 16; sft >= 1; sft >>= 1)
                for (int l = 0; l < 16; ++l) {
                    d2[l] = d2[l] + d2[l + sft];
                    d4[l] = d4[l] + d4[l + sft];
                    wt[l] = (wt[l] + wt[l + sft]) & 0xffu;
                }
            float pb = 0.f, kb = 0.f;
            if (wt[0] > 0) {
                pb = d2[0] / (float)wt[0];
                kb = d4[0] / (float)wt[0] / (pb * pb);
            }
            sfb[tid] = pb;
            sfb[2 + tid] = kb;
        }
        __syncthreads();
        if (tid < 2) {
            const size_t ab = (size_t)ant * 5 * nrow_cap;
            stats_fb[ab + (0 + tid) * nrow_cap + grow] = sfb[tid];
            stats_fb[ab + (2 + tid) * nrow_cap + grow] = sfb[2 + tid];
            if (tid == 0)
                stats_fb[ab + 4 * nrow_cap + grow] = fmaxf(dag_one(sfb[2], dc_fb), dag_one(sfb[3], dc_fb));
        }
    }
    if (tid == 0) {
        // kur_weights after apply_kurtosis (:292): one atomicAdd of 500/12500 per unflagged block;
        // identical addends sum to the same float in any order.  Both pols share the flag, hence
        // kur_weights[t] == kur_weights[t + FFTS_PER_SEG].
        const float inc = (float)PB_NKURTO / PB_NFFT;
        float w = 0.f;
        uint32_t m = 0;
        for (int k = 0; k < PB_BLK_PER_FFT; ++k) {
            if (!sflag[k]) w = w + inc;
            m |= (sflag[k] ? 1u : 0u) << k;
        }
        wrow_out[(size_t)ant * wrow_ant_stride + grow] = w;
        // the row's 25 flags as one word: the channeliser fetches it (and the weight) with one scalar load
        rowmask_out[(size_t)ant * wrow_ant_stride + grow] = m;
    }
    if (WRITE_F32) {
#pragma unroll
        for (int pol = 0; pol < 2; ++pol) {
            const uint32_t *sw = (const uint32_t *)((const uint8_t *)sraw[pol] + (pol ? off1 : off0));
            const size_t fo = pol ? rbyte1 : rbyte0;  // float index = sample index
            for (int i = tid; i < PB_NFFT / 4; i += 256) {
                const float4 v = cvt4(sw[i]);
                if (write_raw) ((float4 *)(fraw + fo))[i] = v;
                ((float4 *)(fkur + fo))[i] = sflag[i / 125] ? make_float4(0.f, 0.f, 0.f, 0.f) : v;
            }
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
    dim3 grid((unsigned)(nseg * h->R), (unsigned)h->A);
    const size_t wrow_ant = (size_t)h->S * h->R;
    // row statistics (K4) live behind the six per-block planes of every antenna
    const size_t nrow_cap = (size_t)h->S * h->R;
    float *stats_fb = h->d_stats ? h->d_stats + (size_t)h->A * 6 * nblk_cap : nullptr;
    if (write_f32)
        k_kurtosis_row<true><<<grid, 256, 0, h->stream>>>(
            h->d_in, in_ant_stride, h->seg_samples, h->R, h->d_flags, nblk_cap, h->d_wrow, pb_rowmask(h), wrow_ant,
            h->d_stats, nblk_cap, h->d_fraw, h->d_fkur, h->cfg.rfi_mode == 2 ? 1 : 0, h->dag, h->dag_fb, stats_fb,
            nrow_cap);
    else
        k_kurtosis_row<false><<<grid, 256, 0, h->stream>>>(
            h->d_in, in_ant_stride, h->seg_samples, h->R, h->d_flags, nblk_cap, h->d_wrow, pb_rowmask(h), wrow_ant,
            h->d_stats, nblk_cap, nullptr, nullptr, 0, h->dag, h->dag_fb, stats_fb, nrow_cap);
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
