// Detect / normalise / scrunch / requantise for the LDS-FFT back end (power planes in,
// filterbank bytes out).  Same reference kernels as k_detect.hip
//   detect_and_normalize2 / 3, pscrunch(_weights), tscrunch(_weights), sel_and_dig_*
//   (src/pb_kernels.cu:393-735)
// but split so that only what is truly serial in time runs serially:
//
//   loader   (1 wave): streams the power plane into an LDS ring D2_DEPTH chunks ahead with
//            global_load_lds_dwordx4 (LDS-DMA): the 67 MB per segment are latency-bound
//            unless ~4 chunks per workgroup are in flight, and DMA costs no registers;
//   phase A  (1 wave, lane = 32 channels x 2 pols of one stream): the running-bandpass
//            recurrence  bp = s*p + (1-s)*bp  with the 11x clip of the excised stream -- a
//            handful of instructions per row -- leaving the bp used by each row in LDS;
//   phase B  (2 waves, lane = one channel x one group of 8 rows): everything that only needs
//            (p, bp) of its own row -- the normalising division, pol scrunch in double, the
//            weighted 8-row time scrunch and the 8/4/2-bit quantiser.
// Phase B of chunk k-1 overlaps phase A of chunk k (double-buffered LDS, one barrier per
// chunk).  The excised stream's power plane already carries pow/w (the channeliser divides
// by the row weight, see k_channelize.hip; +inf for rows of weight 0), so neither a division
// nor the row weight sits on the serial path.
// Stream, npol and nbit are template parameters and every data-dependent choice is a select,
// so each phase is straight-line code the scheduler can interleave across rows.
//
// Exactness notes (all selects reproduce the reference's branches bit for bit):
//   (double)w >= 0.2  <=>  w >= 0.2f   (0.2f is the smallest float above 0.2);
//   0.5*(w+w) == w exactly; adding +0.0f to a sum that started at +0.0f is the identity.
//
// HBM traffic: the power planes (4 B per row x channel x pol x stream) once; outputs 1/64 of it.
#include "pb_internal.h"

struct Detect2Args {
    const float *P[2];       // per stream [A][S][2][R][4096]; stream 1 holds pow / w
    const float *wrow;       // [A][S*R] row weights after apply_kurtosis
    float *bp;               // [A][2][2][4096]
    uint8_t *codes;          // [A][2][S][trim]
    float *ave;              // [A][2][S][ave_per_seg] or nullptr
    size_t trim, ave_per_seg;
    int S, R, nseg;
    float scale, oms, tscale;
};

#define D2_DEPTH 2                 // chunks of power loads in flight per workgroup: 4 slots x 8 KB + 16 KB
                                   // = 48 KB of LDS, so that a detect workgroup fits beside two channeliser
                                   // workgroups of the next batch (deeper rings measured no faster)
#define D2_NSLOT (D2_DEPTH + 2)    // LDS ring slots (a slot is re-filled two barriers after its last reader)

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int NBIT> __device__ __forceinline__ unsigned quantise(float acc)
{
    if (NBIT == 8) {
        const float tmp = (float)((double)acc / 0.02957 + 127.5);
        return tmp <= 0 ? 0u : (tmp >= 255 ? 255u : (unsigned)(uint8_t)tmp);
    } else if (NBIT == 4) {
        const float tmp = (float)((double)acc / 0.3188 + 7.5);
        return tmp <= 0 ? 0u : (tmp >= 15 ? 15u : (unsigned)(uint8_t)tmp);
    } else {
        const double t = (double)acc;
        return t < -0.6109 ? 0u : (t < 0.3970 ? 1u : (t < 1.4050 ? 2u : 3u));
    }
}

template <int T, bool KUR, int NPOL, int NBIT>
__device__ __forceinline__ void detect2_body(const Detect2Args &a, float (*s_p)[T][64], float (*s_u)[T][64],
                                             float (*s_w)[T])
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cg = blockIdx.x;
    constexpr int stream = KUR ? 1 : 0;
    const int ant = blockIdx.z;
    const int R = a.R, cps = R / T, nchunk = a.nseg * cps;
    const int ntime = R / PB_NSCRUNCH;
    const float *wrow = a.wrow + (size_t)ant * a.S * R;
    const size_t pol_stride = (size_t)R * PB_NCHANOUT, seg_stride = 2 * pol_stride;
    const float *Pant = a.P[stream] + (size_t)ant * a.S * seg_stride;
    constexpr int LPC = T / 4;        // DMA instructions per chunk (4 rows of 64 floats each)

    // ---- loader (wave 3): lane -> (row in group of 4, pol, 4 channels)
    const int ld_row = lane >> 4, ld_pol = (lane >> 3) & 1, ld_c = cg * 32 + (lane & 7) * 4;
    auto issue = [&](int kk) {
        const int seg = kk / cps, row0 = (kk % cps) * T;
        const float *src = Pant + (size_t)seg * seg_stride + (size_t)ld_pol * pol_stride +
                           (size_t)(row0 + ld_row) * PB_NCHANOUT + ld_c;
#pragma unroll
        for (int i = 0; i < LPC; ++i)
            __builtin_amdgcn_global_load_lds(
                (const void __attribute__((address_space(1))) *)(src + (size_t)(4 * i) * PB_NCHANOUT),
                (void __attribute__((address_space(3))) *)&s_p[kk % D2_NSLOT][4 * i][0], 16, 0, 0);
    };
    if (wave == 3) {
        for (int kk = 0; kk < D2_DEPTH && kk < nchunk; ++kk) issue(kk);
        if (nchunk > D2_DEPTH - 1) wait_vmcnt<(D2_DEPTH - 1) * LPC>();
        else wait_vmcnt<0>();
    }

    // ---- phase A state (wave 0)
    const int polA = lane >> 5, cA = cg * 32 + (lane & 31);
    const float *inA = Pant + (size_t)polA * pol_stride + cA;
    float *bpp = a.bp + (((size_t)ant * 2 + stream) * 2 + polA) * PB_NCHANOUT + cA;
    float bp = (wave == 0) ? *bpp : 0.f;
    // row weights of the next chunk are requested one iteration ahead (rows of chunk k are the
    // contiguous wrow[k*T .. k*T+T-1]); a load issued in the iteration that uses it would put a full
    // memory latency on every chunk
    float wv_next = (KUR && wave == 0 && lane < T && nchunk > 0) ? wrow[lane] : 1.f;

    // ---- phase B state (waves 1, 2)
    const int idxB = (wave - 1) * 64 + lane;      // group-in-chunk * 32 + channel
    const int gB = idxB >> 5, chB = idxB & 31;
    const bool activeB = (wave == 1 || wave == 2) && gB < T / PB_NSCRUNCH;
    const int cB = cg * 32 + chB;
    uint8_t *codes = a.codes + ((size_t)ant * 2 + stream) * a.S * a.trim;
    float *ave = a.ave ? a.ave + ((size_t)ant * 2 + stream) * a.S * a.ave_per_seg : nullptr;
    const float scale = a.scale, oms = a.oms, tscale = a.tscale;
    __syncthreads();

    for (int k = 0; k <= nchunk; ++k) {
        const int buf = k & 1;
        if (wave == 3) {
            if (k + D2_DEPTH < nchunk) {
                issue(k + D2_DEPTH);
                wait_vmcnt<(D2_DEPTH - 1) * LPC>();   // chunk k+1 has landed
            } else {
                wait_vmcnt<0>();
            }
        } else if (wave == 0 && k < nchunk) {
            const int seg = k / cps, row0 = (k % cps) * T;
            const int slot = k % D2_NSLOT;
            const float *wseg = wrow + (size_t)seg * R;
            const float wv = wv_next;
            if (KUR) {
                if (k + 1 < nchunk && lane < T) wv_next = wrow[(size_t)(k + 1) * T + lane];
                if (lane < T) s_w[buf][lane] = wv;
            }
            if (row0 == 0 && bp == 0.f) {
                // initialise the bandpass from this segment's mean (:406-411, :444-461)
                const float *p = inA + (size_t)seg * seg_stride;
                if (!KUR) {
                    for (int t = 0; t < R; ++t) bp += p[(size_t)t * PB_NCHANOUT];
                    bp /= (float)R;
                } else {
                    int good = 0;
                    for (int t = 0; t < R; ++t) {
                        if (wseg[t] == 0.f) continue;
                        good++;
                        bp += p[(size_t)t * PB_NCHANOUT];
                    }
                    if (good == 0) bp = 1.f;
                    else bp /= (float)good;
                }
            }
            float pk[T];
#pragma unroll
            for (int j = 0; j < T; ++j) pk[j] = s_p[slot][j][lane];
#pragma unroll
            for (int j = 0; j < T; ++j) {
                float u;
                const float t1 = scale * pk[j];
                const float t2 = oms * bp;
                const float bpn = t1 + t2;
                if (!KUR) {
                    bp = bpn;
                    u = bp;
                } else {
                    // rows with weight 0 arrive as +inf from the channeliser, so the 11x clip test
                    // alone keeps the bandpass unchanged for them (:474-476) as for clipped samples
                    // (:493-494); phase B turns their x into 0 from the row weight
                    const bool clip = pk[j] > bp * 11.f;
                    bp = clip ? bp : bpn;
                    u = clip ? -1.f : bp;
                }
                s_u[buf][j][lane] = u;
            }
        } else if (activeB && k >= 1) {
            const int kb = k - 1, pb = kb & 1, slot = kb % D2_NSLOT;
            const int seg = kb / cps, row0 = (kb % cps) * T;
            float acc0 = 0.f, acc1 = 0.f, wt_sumf = 0.f;
            int wt_sum = 0;
            float p0[PB_NSCRUNCH], p1[PB_NSCRUNCH], u0[PB_NSCRUNCH], u1[PB_NSCRUNCH], wr[PB_NSCRUNCH];
#pragma unroll
            for (int j = 0; j < PB_NSCRUNCH; ++j) {
                const int rl = gB * PB_NSCRUNCH + j;
                p0[j] = s_p[slot][rl][chB];
                p1[j] = s_p[slot][rl][32 + chB];
                u0[j] = s_u[pb][rl][chB];
                u1[j] = s_u[pb][rl][32 + chB];
                wr[j] = KUR ? s_w[pb][rl] : 1.f;
            }
#pragma unroll
            for (int j = 0; j < PB_NSCRUNCH; ++j) {
                float x0 = p0[j] / u0[j] - 1.f;
                float x1 = p1[j] / u1[j] - 1.f;
                const float w = wr[j];
                if (KUR) {
                    x0 = u0[j] < 0.f ? 10.f : x0;
                    x1 = u1[j] < 0.f ? 10.f : x1;
                    x0 = w == 0.f ? 0.f : x0;
                    x1 = w == 0.f ? 0.f : x1;
                }
                if (NPOL == 1) {
                    const float s = x0 + x1;
                    const float p = (float)(M_SQRT1_2 * (double)s);
                    if (!KUR) {
                        acc0 += p;
                    } else {
                        const bool ok = w >= 0.2f;           // MIN_WEIGHT, both pols share the row weight
                        wt_sum += ok ? 1 : 0;
                        wt_sumf += ok ? w : 0.f;
                        const float prod = w * p;
                        acc0 += ok ? prod : 0.f;
                    }
                } else {
                    if (!KUR) {
                        acc0 += x0;
                        acc1 += x1;
                    } else {
                        const bool ok = !(w < 0.2f);
                        wt_sum += ok ? 1 : 0;
                        wt_sumf += ok ? w : 0.f;
                        const float pr0 = w * x0, pr1 = w * x1;
                        acc0 += ok ? pr0 : 0.f;
                        acc1 += ok ? pr1 : 0.f;
                    }
                }
            }
            if (!KUR) {
                acc0 *= tscale;
                acc1 *= tscale;
            } else {
                const bool ok = (wt_sumf / PB_NSCRUNCH) >= 0.2f;
                const float d = sqrtf((float)wt_sum);
                const float q0 = acc0 / d, q1 = acc1 / d;
                acc0 = ok ? q0 : 0.f;
                acc1 = ok ? q1 : 0.f;
            }
            const int trow = (row0 >> 3) + gB;
            uint8_t *cseg = codes + (size_t)seg * a.trim;
            float *aseg = ave ? ave + (size_t)seg * a.ave_per_seg : nullptr;
#pragma unroll
            for (int pol = 0; pol < NPOL; ++pol) {
                const float acc = pol ? acc1 : acc0;
                const size_t n = (NPOL == 1) ? (size_t)trow * PB_NCHANOUT + cB
                                             : ((size_t)trow * 2 + pol) * PB_NCHANOUT + cB;
                if (aseg) aseg[(NPOL == 1) ? n : ((size_t)pol * ntime + trow) * PB_NCHANOUT + cB] = acc;
                const unsigned q = quantise<NBIT>(acc);
                if (NBIT == 8) {
                    cseg[n] = (uint8_t)q;
                } else if (NBIT == 4) {
                    const unsigned hi = __shfl_down(q, 1);
                    if (!(lane & 1)) cseg[n >> 1] = (uint8_t)(q | (hi << 4));
                } else {
                    const unsigned q1 = __shfl_down(q, 1);
                    const unsigned q2 = __shfl_down(q, 2);
                    const unsigned q3 = __shfl_down(q, 3);
                    if (!(lane & 3)) cseg[n >> 2] = (uint8_t)(q | (q1 << 2) | (q2 << 4) | (q3 << 6));
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) *bpp = bp;
}

// MODE = rfi_mode: 0 raw stream only, 1 excised only, 2 both (blockIdx.y picks the stream)
template <int T, int NPOL, int NBIT, int MODE>
__global__ __launch_bounds__(256) void k_detect2(Detect2Args a)
{
    // ring of power chunks filled by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU)
    __shared__ __attribute__((aligned(16))) float s_p[D2_NSLOT][T][64];
    __shared__ float s_u[2][T][64];   // bp used for the row; < 0 marks a clipped sample
    __shared__ float s_w[2][T];
    if (MODE == 0 || (MODE == 2 && blockIdx.y == 0)) detect2_body<T, false, NPOL, NBIT>(a, s_p, s_u, s_w);
    else detect2_body<T, true, NPOL, NBIT>(a, s_p, s_u, s_w);
}

template <int T, int NPOL, int NBIT>
static void launch_mode(const Detect2Args &a, int mode, dim3 grid, hipStream_t st)
{
    if (mode == 0) k_detect2<T, NPOL, NBIT, 0><<<grid, 256, 0, st>>>(a);
    else if (mode == 1) k_detect2<T, NPOL, NBIT, 1><<<grid, 256, 0, st>>>(a);
    else k_detect2<T, NPOL, NBIT, 2><<<grid, 256, 0, st>>>(a);
}

template <int T>
static void launch_all(const Detect2Args &a, int mode, int npol, int nbit, dim3 grid, hipStream_t st)
{
    if (npol == 1) {
        if (nbit == 8) launch_mode<T, 1, 8>(a, mode, grid, st);
        else if (nbit == 4) launch_mode<T, 1, 4>(a, mode, grid, st);
        else launch_mode<T, 1, 2>(a, mode, grid, st);
    } else {
        if (nbit == 8) launch_mode<T, 2, 8>(a, mode, grid, st);
        else if (nbit == 4) launch_mode<T, 2, 4>(a, mode, grid, st);
        else launch_mode<T, 2, 2>(a, mode, grid, st);
    }
}

hipError_t launch_detect_pow(pb_handle *h, int nseg)
{
    Detect2Args a;
    a.P[0] = h->d_Praw;
    a.P[1] = h->d_Pkur;
    a.wrow = h->d_wrow;
    a.bp = h->d_bp;
    a.codes = h->d_codes;
    a.ave = h->cfg.keep_ave ? h->d_ave : nullptr;
    a.trim = h->trim;
    a.ave_per_seg = h->ave_per_seg;
    a.S = h->S;
    a.R = h->R;
    a.nseg = nseg;
    const double tsamp = (double)PB_NFFT / 128000000 * PB_NSCRUNCH;  // src/process_baseband.cu:739-741
    a.scale = (float)(tsamp / 1.0);
    a.oms = 1 - a.scale;
    a.tscale = (float)sqrt(1. / PB_NSCRUNCH);
    dim3 grid(PB_NCHANOUT / 32, h->cfg.rfi_mode == 2 ? 2 : 1, h->A);
    if (h->R % 32 == 0) launch_all<32>(a, h->cfg.rfi_mode, h->cfg.npol, h->cfg.nbit, grid, h->stream);
    else launch_all<8>(a, h->cfg.rfi_mode, h->cfg.npol, h->cfg.nbit, grid, h->stream);
    return hipGetLastError();
}
