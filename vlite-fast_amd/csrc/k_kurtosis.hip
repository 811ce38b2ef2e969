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

#include "kurtosis_dev.h"

#ifdef KU_STAMP_ON
// timing experiments (variant builds only): the clock at the phase boundaries of every workgroup
__shared__ unsigned long long ku_ts[6];
__device__ unsigned long long g_ku_stamp[20480][6];
extern "C" int pb_internal_ku_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ku_stamp), sizeof(g_ku_stamp));
}
#define KU_STAMP(i) do { if (threadIdx.x == 0) ku_ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define KU_STAMP_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 20480 && blockIdx.y == 0) \
    for (int i_ = 0; i_ < 6; ++i_) g_ku_stamp[blockIdx.x][i_] = ku_ts[i_]; } while (0)
#else
#define KU_STAMP(i)
#define KU_STAMP_FLUSH()
#endif
#define ROW_CHUNKS 784   // 16-byte chunks covering a 12500-byte row at any 4-byte alignment

// ROW_STATS: also the whole-row statistic K4 (debug_keep only; kept out of the production kernel, whose
// registers it would otherwise dictate)
template <bool WRITE_F32, bool ROW_STATS>
__global__ __launch_bounds__(256, 6) void k_kurtosis_row(
    uint8_t *in, size_t in_ant_stride, size_t seg_samples, int R,
    uint8_t *__restrict__ flags, size_t flags_ant_stride, float *__restrict__ wrow_out,
    uint32_t *__restrict__ rowmask_out, size_t wrow_ant_stride, float *__restrict__ stats, size_t nblk_cap,
    float *__restrict__ fraw, float *__restrict__ fkur, int write_raw, DagConsts dc, DagConsts dc_fb,
    float *__restrict__ stats_fb, size_t nrow_cap)
{
    __shared__ uint4 sraw[2][ROW_CHUNKS];
    KU_STAMP(0);
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
        // per instruction, so that the conversion below is one fma per pair: u/128 - 1 is exact.  A chunk
        // that held such a code also goes back to the input buffer patched (rare: dropped frames only), so
        // that the channelisers -- which read every byte again, the PFB one four times -- need not repeat
        // this.  Neighbouring rows share their boundary chunks: both write the same bytes.
        auto stage = [&](uint4 *dst, const uint4 *src, int i, uint4 q) __attribute__((always_inline)) {
            const uint4 f = fix0(q);
            dst[i] = f;
            if ((f.x ^ q.x) | (f.y ^ q.y) | (f.z ^ q.z) | (f.w ^ q.w)) ((uint4 *)src)[i] = f;
        };
        stage(sraw[0], s0, tid, a0); stage(sraw[0], s0, tid + 256, a1); stage(sraw[0], s0, tid + 512, a2);
        stage(sraw[1], s1, tid, b0); stage(sraw[1], s1, tid + 256, b1); stage(sraw[1], s1, tid + 512, b2);
        if (h0) stage(sraw[0], s0, tid + 768, a3);
        if (h1) stage(sraw[1], s1, tid + 768, b3);
    }
    __syncthreads();
    KU_STAMP(1);

    // moments of the 2 x 25 blocks: 13, 13, 12, 12 consecutive blocks per wave (kurtosis_dev.h)
    row_block_moments((const uint8_t *)sraw[0] + off0, (const uint8_t *)sraw[1] + off1, wave, lane, s2, s4);
    __syncthreads();
    KU_STAMP(2);
    const size_t b0 = (size_t)grow * PB_BLK_PER_FFT;   // first block index of this row (per pol)
    if (tid < 50) {
        const int pol = tid / 25, blk = tid % 25;
        const float p = s2[tid] / PB_NKURTO;
        const float k = s4[tid] / PB_NKURTO / (p * p);
        // the score itself only where it is kept (statistics); a flag needs no cube root (dag_flag): 9 = "above"
        sdag[tid] = (stats || (ROW_STATS && stats_fb)) ? dag_one(k, dc) : (dag_flag(k, dc) ? 9.0f : 0.0f);
        if (ROW_STATS && stats_fb) {
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
    KU_STAMP(3);
    if (tid < 25) {
        const float dmax = fmaxf(sdag[tid], sdag[25 + tid]);
        const bool bad = dmax > 3.0f;  // DAG_THRESH
        sflag[tid] = bad ? 1u : 0u;
        flags[(size_t)ant * flags_ant_stride + b0 + tid] = bad ? 1 : 0;
        // the row's 25 flags as one word (lanes 0..24 of wave 0 are the only active ones here): the channeliser
        // fetches it, and the weight, with one scalar load each
        const uint32_t m = (uint32_t)__ballot(bad);
        if (tid == 0) {
            // kur_weights after apply_kurtosis (:292): one atomicAdd of 500/12500 per unflagged block;
            // identical addends sum to the same float in any order.  Both pols share the flag, hence
            // kur_weights[t] == kur_weights[t + FFTS_PER_SEG].
            const float inc = (float)PB_NKURTO / PB_NFFT;
            float w = 0.f;
            for (int k = PB_BLK_PER_FFT - __popc(m); k > 0; --k) w = w + inc;
            wrow_out[(size_t)ant * wrow_ant_stride + grow] = w;
            rowmask_out[(size_t)ant * wrow_ant_stride + grow] = m;
        }
        if (stats) {
            const size_t ab = (size_t)ant * 6 * nblk_cap;
            stats[ab + (2 * 2 + 0) * nblk_cap + b0 + tid] = dmax;
            stats[ab + (2 * 2 + 1) * nblk_cap + b0 + tid] = dmax;
        }
    }
    __syncthreads();
    KU_STAMP(4);
    if (ROW_STATS && stats_fb) {
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
    KU_STAMP(5);
    KU_STAMP_FLUSH();
}

// score > DAG_THRESH for n consecutive floats t from bit pattern bits0 on (pb_create: where the score crosses the
// threshold, evaluated by the device that will evaluate the flags)
__global__ void k_dag_scan(DagConsts c, uint32_t bits0, int n, uint8_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dag_of_t(__uint_as_float(bits0 + (uint32_t)i), c) > 3.0f ? 1 : 0;
}

hipError_t launch_dag_scan(pb_handle *h, const DagConsts &c, uint32_t bits0, int n, uint8_t *d_out)
{
    k_dag_scan<<<(n + 255) / 256, 256, 0, h->stream>>>(c, bits0, n, d_out);
    return hipGetLastError();
}

// the flag without the cube root (dag_flag) against the score itself for every float kurtosis in [bits_lo, bits_hi]
__global__ void k_dag_check(DagConsts c, uint32_t bits_lo, uint64_t n, unsigned long long *__restrict__ mismatches)
{
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float kur = __uint_as_float(bits_lo + (uint32_t)i);
        bad += dag_flag(kur, c) != (dag_one(kur, c) > 3.0f);
    }
    if (bad) atomicAdd(mismatches, bad);
}

hipError_t launch_dag_check(pb_handle *h, uint32_t bits_lo, uint64_t n, unsigned long long *d_mismatches)
{
    k_dag_check<<<2048, 256, 0, h->stream>>>(h->dag, bits_lo, n, d_mismatches);
    return hipGetLastError();
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
#define PB_KURT_LAUNCH(F32, FB)                                                                                   \
    k_kurtosis_row<F32, FB><<<grid, 256, 0, h->stream>>>(                                                          \
        h->d_in, in_ant_stride, h->seg_samples, h->R, h->d_flags, nblk_cap, h->d_wrow, pb_rowmask(h), wrow_ant,    \
        h->d_stats, nblk_cap, F32 ? h->d_fraw : nullptr, F32 ? h->d_fkur : nullptr,                                \
        F32 && h->cfg.rfi_mode == 2 ? 1 : 0, h->dag, h->dag_fb, stats_fb, nrow_cap)
    if (write_f32) {
        if (stats_fb) PB_KURT_LAUNCH(true, true); else PB_KURT_LAUNCH(true, false);
    } else {
        if (stats_fb) PB_KURT_LAUNCH(false, true); else PB_KURT_LAUNCH(false, false);
    }
#undef PB_KURT_LAUNCH
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
