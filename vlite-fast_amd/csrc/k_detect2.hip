// Detect / normalise / scrunch / requantise for the LDS-FFT back end (power planes in,
// filterbank bytes out).  Same reference kernels as k_detect.hip
//   detect_and_normalize2 / 3, pscrunch(_weights), tscrunch(_weights), sel_and_dig_*
//   (src/pb_kernels.cu:393-735)
// but split so that only what is truly serial in time runs serially.  One 384-thread workgroup = 32
// channels x 2 pols of one stream, chunks of T rows, one barrier per chunk step:
//
//   loader (wave 4): streams the power plane into an LDS ring DEPTH chunks ahead with
//            global_load_lds_dwordx4 (LDS-DMA, no registers);
//   A  (wave 0, lane = (pol, channel)): the running-bandpass recurrence  bp = s*p + (1-s)*bp  with the 11x
//            clip of the excised stream, for chunk k; leaves bp after every row in LDS;
//   B  (waves 1, 2, 3, 5; lane = (8-row group, channel), both pols): everything that only needs (p, bp) of
//            its own row -- the clip test again (same operands -> same result), the normalising division,
//            pol scrunch in double, the weighted 8-row time scrunch and the 8/4/2-bit quantiser.  A B wave
//            takes chunk c into registers at step c+1 and spends TWO steps on it (rows 0-3 of its groups,
//            then rows 4-7 and the quantiser) while the wave of the other parity takes chunk c+1.
//
// What the round-2 measurements (tools/ubench_valu.hip, profiles/r02_notes.md) say about such a
// kernel, and what this version does about it:
//   * a wave issues ONE instruction -- vector, scalar or s_nop alike -- per ~4.5 cycles whatever the
//     dependences; an LDS store costs the issuing wave ~16 cycles (13 per row as ds_write_b128 of four
//     rows), an LDS load ~9.  The recurrence wave is therefore bound by its instruction COUNT per row:
//     the bp it exports goes out as one 16-byte store per four rows, the select's two wait states
//     after v_cmp (gfx950) carry the next row's s*p, "clipped" is not exported at all (phase B
//     repeats the comparison), and the chunk bookkeeping is adds and compares, not divisions.
//   * phase B was as long as the recurrence per chunk (2 waves x ~2400 cycles per 32 rows); four waves
//     in two-step turns halve its per-step instruction stream.
// The excised stream's power plane already carries pow/w (the channeliser divides by the row weight,
// see k_channelize.hip; +inf for rows of weight 0), so neither a division nor the row weight sits on
// the serial path.  Stream, npol and nbit are template parameters and every data-dependent choice is a
// select, so each phase is straight-line code.
//
// Exactness notes (all selects reproduce the reference's branches bit for bit):
//   (double)w >= 0.2  <=>  w >= 0.2f   (0.2f is the smallest float above 0.2);
//   0.5*(w+w) == w exactly; adding +0.0f to a sum that started at +0.0f is the identity;
//   rows of weight 0 arrive as +inf: inf > bp*11 keeps the bandpass (:474-476), phase B zeroes x from w.
//
// HBM traffic: the power planes (4 B per row x channel x pol x stream) once; outputs 1/64 of it.
#include <cstdlib>

#include "pb_internal.h"

struct Detect2Args {
    const float *P[2];       // per stream [A][S][2][R][4096]; stream 1 holds pow / w
    const float *wrow;       // [A][S*R] row weights after apply_kurtosis
    float *bp;               // [A][2][2][4096]
    uint8_t *codes;          // [A][2][S][trim]
    float *ave;              // [A][2][S][ave_per_seg] or nullptr
    float *ave_target;       // antenna 0's plane of stream `target_stream` goes here instead (pb_set_coadd_target)
    int target_stream;
    size_t trim, ave_per_seg;
    int S, R, nseg;
    float scale, oms, tscale;
};

#ifndef D2_LOAD_AUX
#define D2_LOAD_AUX 0              // cache-policy bits of the plane loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef D2_PRIO_A
#define D2_PRIO_A 3                // wave priority of the recurrence wave
#endif
#ifndef D2_PRIO_B
#define D2_PRIO_B 0                // wave priority of the loader and phase-B waves (the recurrence wave runs at 3)
#endif
// DEPTH (template parameter): chunks of power loads in flight per workgroup; LDS ring slots = DEPTH + 2: B takes chunk
// k-1 into registers during step k, A reads k, k+1 .. k+DEPTH landed / in flight.  DEPTH 2: 4 x 8 KB + 16 KB of bp =
// 49 KB; DEPTH 3: 57 KB -- either way a detect workgroup takes the place of exactly one 50-KB channeliser workgroup
// of the next batch.  Beside the channeliser that flags its own rows (detect runs wholly beside it) three chunks in
// flight make the step 1 - 3 % shorter with one antenna per GPU (0.634 -> 0.617 ms on one box, 0.636 -> 0.628 on
// another, alternating runs); with two antennas per GPU, and beside the kurtosis pass and the PFB channeliser, two
// are 1 - 1.5 % better (0.986 against 1.002 ms): launch_detect_pow picks.
#ifndef D2_DEPTH_OVERLAPPED
#define D2_DEPTH_OVERLAPPED 3
#endif
#define D2_NSLOT (DEPTH + 2)       // (4 x 8 KB + 16 KB of bp = 49 KB at DEPTH 2:
                                   // a detect workgroup fits beside two channeliser workgroups of the next batch)
#define D2_THREADS 384
#define D2_WAVE_A 0
#ifndef D2_WAVE_L
#define D2_WAVE_L 5               // waves i and i+4 share a SIMD
#endif

#ifdef D2_STAMP
// timing experiments (variant builds only): per wave of one workgroup, cycles spent working / waiting at the
// step barrier; read back with pb_internal_d2_stamps
__device__ unsigned long long g_d2_stamp[8][4];
extern "C" int pb_internal_d2_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_d2_stamp), sizeof(g_d2_stamp));
}
// start / end (s_memrealtime, 100 MHz), XCC_ID | HW_ID << 8 and launch number of every workgroup of the last 64
// launches: [launch % 64][wg][4]
__device__ unsigned long long g_d2_wg[64][256][4];
__device__ unsigned g_d2_wgcnt[256];
extern "C" int pb_internal_d2_wg(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_d2_wg), sizeof(g_d2_wg));
}
#define D2_NOW() ((long long)__builtin_amdgcn_s_memtime())
#else
#define D2_NOW() 0ll
#endif

namespace {

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS operations of this wave done, then the workgroup barrier.  (Not __syncthreads(): its release fence
// may also drain vmcnt, i.e. the loader's DMA that must stay in flight across steps.)
__device__ __forceinline__ void step_barrier_raw()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef D2_STAMP
struct StepClock {
    long long work = 0, wait = 0, t = 0, extra = 0;
};
__device__ __forceinline__ void step_barrier_clk(StepClock &c)
{
    const long long t1 = D2_NOW();
    step_barrier_raw();
    const long long t2 = D2_NOW();
    if (c.t) { c.work += t1 - c.t; c.wait += t2 - t1; }
    c.t = t2;
}
#define step_barrier() step_barrier_clk(clk)
#else
#define step_barrier() step_barrier_raw()
#endif

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

// which phase-B wave (0..3) a wave is, or -1: digit = parity * 2 + half of the (group, channel) lane tasks
__device__ __forceinline__ int b_index(int wave)
{
    return wave == 1 ? 0 : (wave == 2 ? 1 : (wave == 3 ? 2 : (wave == 9 - D2_WAVE_L ? 3 : -1)));
}

// What a phase-B lane holds of its chunk between the two steps it spends on it
template <int NPOL> struct BState {
    float p[2][8];         // power, per pol, the 8 rows of the group
    float4 u[2][2];        // bp after each row, per pol, rows 0-3 / 4-7
    float uprev[2];        // bp before row 0
    float w[8];            // row weights (excised stream)
    float acc[2], wt_sumf;
    int wt_sum;
    int trow, seg;
};

// rows 4 h .. 4 h + 3 of the group
template <int NPOL, bool KUR>
__device__ __forceinline__ void phase_b_rows(BState<NPOL> &s, int h)
{
    if (h == 0) {
        s.acc[0] = s.acc[1] = 0.f;
        s.wt_sumf = 0.f;
        s.wt_sum = 0;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        float x[2];
        const float w = KUR ? (h ? s.w[4 + jj] : s.w[jj]) : 1.f;
#pragma unroll
        for (int pol = 0; pol < 2; ++pol) {
            const float4 u4 = h ? s.u[pol][1] : s.u[pol][0];
            const float ub4 = h ? s.u[pol][0].w : s.uprev[pol];          // bp before this quad's first row
            const float un = jj == 0 ? u4.x : (jj == 1 ? u4.y : (jj == 2 ? u4.z : u4.w));     // bp after the row
            const float ub = jj == 0 ? ub4 : (jj == 1 ? u4.x : (jj == 2 ? u4.y : u4.z));      // bp before the row
            const float p = h ? s.p[pol][4 + jj] : s.p[pol][jj];
#if defined(D2_ABL) && (D2_ABL & 1)
            float v = p * un - 1.f;          // (energy experiment, results invalid: what the IEEE divisions cost)
#else
            float v = p / un - 1.f;
#endif
            if (KUR) {
                v = p > ub * 11.f ? 10.f : v;          // the recurrence wave's own test (:490-491): clipped -> 10
                v = w == 0.f ? 0.f : v;                // :474-476
            }
            x[pol] = v;
        }
        if (NPOL == 1) {
            const float sum = x[0] + x[1];
            const float y = (float)(M_SQRT1_2 * (double)sum);
            if (!KUR) {
                s.acc[0] += y;
            } else {
                const bool ok = w >= 0.2f;           // MIN_WEIGHT, both pols share the row weight
                s.wt_sum += ok ? 1 : 0;
                s.wt_sumf += ok ? w : 0.f;
                const float prod = w * y;
                s.acc[0] += ok ? prod : 0.f;
            }
        } else {
            if (!KUR) {
                s.acc[0] += x[0];
                s.acc[1] += x[1];
            } else {
                const bool ok = !(w < 0.2f);
                s.wt_sum += ok ? 1 : 0;
                s.wt_sumf += ok ? w : 0.f;
                const float pr0 = w * x[0], pr1 = w * x[1];
                s.acc[0] += ok ? pr0 : 0.f;
                s.acc[1] += ok ? pr1 : 0.f;
            }
        }
    }
}

template <int NPOL, int NBIT, bool KUR>
__device__ __forceinline__ void phase_b_finish(const Detect2Args &a, const BState<NPOL> &s, int lane, int cB, int ntime,
                                               uint8_t *codes, float *ave)
{
    float acc0 = s.acc[0], acc1 = s.acc[1];
    if (!KUR) {
        acc0 *= a.tscale;
        acc1 *= a.tscale;
    } else {
        const bool ok = (s.wt_sumf / PB_NSCRUNCH) >= 0.2f;
        const float d = sqrtf((float)s.wt_sum);
        const float q0 = acc0 / d, q1 = acc1 / d;
        acc0 = ok ? q0 : 0.f;
        acc1 = ok ? q1 : 0.f;
    }
    uint8_t *cseg = codes + (size_t)s.seg * a.trim;
    float *aseg = ave ? ave + (size_t)s.seg * a.ave_per_seg : nullptr;
    const int trow = s.trow;
#pragma unroll
    for (int pol = 0; pol < NPOL; ++pol) {
        const float acc = pol ? acc1 : acc0;
        const size_t n = (NPOL == 1) ? (size_t)trow * PB_NCHANOUT + cB : ((size_t)trow * 2 + pol) * PB_NCHANOUT + cB;
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

// Position of one pipeline stage: chunk index and what is derived from it, advanced with adds and compares
// (a scalar division or remainder costs the wave tens of issue slots per step)
template <int DEPTH> struct Cursor {
    int c, slot, seg, rb;      // chunk, ring slot (c mod NSLOT), segment, chunk within the segment
    __device__ __forceinline__ void init(int c0, int cps)
    {
        c = c0;
        const int m = c0 >= 0 ? c0 : 0;        // stages start at or before chunk 0: only c >= 0 is ever used
        slot = m % D2_NSLOT;
        seg = m / cps;
        rb = m % cps;
    }
    __device__ __forceinline__ void next(int cps)
    {
        if (c >= 0) {
            slot = slot + 1 == D2_NSLOT ? 0 : slot + 1;
            rb = rb + 1;
            if (rb == cps) { rb = 0; seg = seg + 1; }
        }
        c = c + 1;
    }
};

}  // namespace

template <int T, bool KUR, int NPOL, int NBIT, int DEPTH>
__device__ __forceinline__ void detect2_body(const Detect2Args &a, float (*s_p)[T][64], float4 (*s_u)[T / 4][64],
                                             float (*s_u0)[64], float (*s_w)[T])
{
    // the wave index in a scalar register: every role test is a scalar branch, the step loops run on scalar counters
    // (Two detect workgroups per CU -- two antennas per handle -- already have their recurrence waves on different
    //  SIMDs: swapping waves 0 <-> 2 in the odd antennas' workgroups made detect 12 % slower alone, r05_notes.md.)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cg = blockIdx.x;
    constexpr int stream = KUR ? 1 : 0;
    constexpr int NQ = T / 4, NG = T / PB_NSCRUNCH;
    const int ant = blockIdx.z;
    const int R = a.R, cps = R / T, nchunk = a.nseg * cps;
    const int ntime = R / PB_NSCRUNCH;
    const float *wrow = a.wrow + (size_t)ant * a.S * R;
    const size_t pol_stride = (size_t)R * PB_NCHANOUT, seg_stride = 2 * pol_stride;
    const float *Pant = a.P[stream] + (size_t)ant * a.S * seg_stride;
    constexpr int LPC = T / 4;        // DMA instructions per chunk (4 rows of 64 floats each)
    const int nstep = nchunk + 2;     // step k: loader k + D, A chunk k, B chunks k-1 (first half) and k-2 (second half)
    const int bi = b_index(wave);
#ifdef D2_STAMP
    StepClock clk;
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    if (wave == D2_WAVE_L) {
        // ---- loader: lane -> (row in group of 4, pol, 4 channels)
#if D2_PRIO_B
        __builtin_amdgcn_s_setprio(D2_PRIO_B);
#endif
        const int ld_row = lane >> 4, ld_pol = (lane >> 3) & 1, ld_c = cg * 32 + (lane & 7) * 4;
        const size_t loff = (size_t)ld_pol * pol_stride + (size_t)ld_row * PB_NCHANOUT + ld_c;
        auto issue = [&](const Cursor<DEPTH> &cu) {
            const float *src = Pant + (size_t)cu.seg * seg_stride + (size_t)(cu.rb * T) * PB_NCHANOUT + loff;
#pragma unroll
            for (int i = 0; i < LPC; ++i)
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1))) *)(src + (size_t)(4 * i) * PB_NCHANOUT),
                    (void __attribute__((address_space(3))) *)&s_p[cu.slot][4 * i][0], 16, 0, D2_LOAD_AUX);
        };
        // at most `n` chunks' loads still in flight (n <= DEPTH - 1, LPC loads each)
        auto wait_chunks = [&](int n) {
            if (n <= 0) wait_vmcnt<0>();
            else if (n == 1) wait_vmcnt<LPC>();
            else wait_vmcnt<(DEPTH - 1) * LPC>();           // (DEPTH <= 3: n == 2)
        };
        // chunks are requested DEPTH ahead of the step that consumes them: chunk k + DEPTH in step k
        Cursor<DEPTH> cu;
        cu.init(0, cps);
        auto fill = [&](int upto) {       // request chunks < upto
            while (cu.c < nchunk && cu.c < upto) {
                issue(cu);
                cu.next(cps);
            }
        };
        fill(DEPTH);
        wait_chunks(cu.c - 1);                       // chunk 0 has landed
        step_barrier();
        for (int k = 0; k < nstep; ++k) {
            fill(k + 1 + DEPTH);
#ifdef D2_STAMP
            const long long tw0 = D2_NOW();
#endif
            wait_chunks(cu.c - (k + 2));             // chunk k + 1 has landed
#ifdef D2_STAMP
            clk.extra += D2_NOW() - tw0;
#endif
            step_barrier();
        }
    } else if (wave == D2_WAVE_A) {
        // ---- A: the recurrence.  lane -> (pol, channel)
        __builtin_amdgcn_s_setprio(D2_PRIO_A);
        const int polA = lane >> 5, cA = cg * 32 + (lane & 31);
        const float *inA = Pant + (size_t)polA * pol_stride + cA;
        float *bpp = a.bp + (((size_t)ant * 2 + stream) * 2 + polA) * PB_NCHANOUT + cA;
        float bp = *bpp;
        const float scale = a.scale, oms = a.oms;
        // row weights of the next chunk are requested one step ahead (rows of chunk k are the contiguous
        // wrow[k*T .. k*T+T-1]) and staged in LDS for phase B
        auto ldw = [&](size_t i) { return wrow[i]; };
        Cursor<DEPTH> cu;
        cu.init(0, cps);
        step_barrier();
        float wv_next = (KUR && lane < T && nchunk > 0) ? ldw(lane) : 1.f;
        for (int k = 0; k < nstep; ++k) {
            if (k < nchunk) {
                const int slot = cu.slot, buf = k & 1;
                if (KUR) {
                    const float wv = wv_next;
                    if (k + 1 < nchunk && lane < T) wv_next = ldw((size_t)(k + 1) * T + lane);
                    if (lane < T) s_w[k % 3][lane] = wv;
                }
                if (cu.rb == 0 && bp == 0.f) {
                    // initialise the bandpass from this segment's mean (:406-411, :444-461): needs the WHOLE segment
                    const float *p = inA + (size_t)cu.seg * seg_stride;
                    const size_t wseg = (size_t)cu.seg * R;
                    auto ldp = [&](int t) { return p[(size_t)t * PB_NCHANOUT]; };
                    if (!KUR) {
                        for (int t = 0; t < R; ++t) bp += ldp(t);
                        bp /= (float)R;
                    } else {
                        int good = 0;
                        for (int t = 0; t < R; ++t) {
                            if (ldw(wseg + t) == 0.f) continue;
                            good++;
                            bp += ldp(t);
                        }
                        if (good == 0) bp = 1.f;
                        else bp /= (float)good;
                    }
                }
                float pk[T];
#pragma unroll
                for (int j = 0; j < T; ++j) pk[j] = s_p[slot][j][lane];
                s_u0[buf][lane] = bp;
                float t1 = scale * pk[0];
                // (The quads are EARLY-CLOBBER outputs: a row writes its bp while later rows of the block still read their
                //  inputs, so no input may be allocated inside the quad.)
                // Four rows per asm block, in FIXED registers: the bp after each row goes straight into the quad that the
                // 16-byte LDS store takes (v[56:59] and v[60:63] alternate; a row reads its predecessor's bp where that
                // row left it), so no copy per row; the excised stream's two products with bp -- (1-s) bp and 11 bp --
                // are ONE packed multiply by the pair (1-s, 11), whose source half op_sel picks (a 64-bit operand must be
                // an even-aligned pair on gfx950).  Same IEEE operations on the same operands as before: 6 issue slots
                // per row instead of 8 (+ the copy).  The recurrence wave is bound by its instruction count
                // (header comment), and the whole kernel by the recurrence wave.
                typedef float q4 __attribute__((ext_vector_type(4)));
                typedef float c2 __attribute__((ext_vector_type(2)));
                const c2 coef = {oms, 11.f};
#define D2_ROW_KUR(PAIR, HALF, BPREG, DST, SP, P, SPN, PN)                                              \
    "v_pk_mul_f32 v[54:55], " PAIR ", %[cf] op_sel:[" HALF ",0] op_sel_hi:[" HALF ",1]\n\t"          \
    "v_add_f32 v54, " SP ", v54\n\t"                                                                  \
    "v_cmp_gt_f32 vcc, " P ", v55\n\t"                                                                \
    "v_mul_f32 " SPN ", %[sc], " PN "\n\t"                                                            \
    "s_nop 0\n\t"                                                                                     \
    "v_cndmask_b32 " DST ", v54, " BPREG ", vcc\n\t"
#define D2_ROW_RAW(BPREG, DST, SP, SPN, PN)                                                             \
    "v_mul_f32 v54, %[om], " BPREG "\n\t"                                                             \
    "v_mul_f32 " SPN ", %[sc], " PN "\n\t"                                                            \
    "v_add_f32 " DST ", " SP ", v54\n\t"
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float p0 = pk[4 * q], p1 = pk[4 * q + 1], p2 = pk[4 * q + 2], p3 = pk[4 * q + 3];
                    const float p4 = pk[4 * q + 4 < T ? 4 * q + 4 : 4 * q + 3];
                    float ta, tb, t1n;
                    q4 o;
                    if (!KUR) {
                        // bp = s p + (1-s) bp (:419); the next row's s*p rides along
                        if ((q & 1) == 0)
                            asm volatile(D2_ROW_RAW("v63", "v56", "%[sp0]", "%[ta]", "%[p1]")
                                         D2_ROW_RAW("v56", "v57", "%[ta]", "%[tb]", "%[p2]")
                                         D2_ROW_RAW("v57", "v58", "%[tb]", "%[ta]", "%[p3]")
                                         D2_ROW_RAW("v58", "v59", "%[ta]", "%[tn]", "%[p4]")
                                         : "=&{v[56:59]}"(o), [ta] "=&v"(ta), [tb] "=&v"(tb), [tn] "=&v"(t1n)
                                         : "{v63}"(bp), [sp0] "v"(t1), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4),
                                           [sc] "v"(scale), [om] "v"(oms)
                                         : "v54");
                        else
                            asm volatile(D2_ROW_RAW("v59", "v60", "%[sp0]", "%[ta]", "%[p1]")
                                         D2_ROW_RAW("v60", "v61", "%[ta]", "%[tb]", "%[p2]")
                                         D2_ROW_RAW("v61", "v62", "%[tb]", "%[ta]", "%[p3]")
                                         D2_ROW_RAW("v62", "v63", "%[ta]", "%[tn]", "%[p4]")
                                         : "=&{v[60:63]}"(o), [ta] "=&v"(ta), [tb] "=&v"(tb), [tn] "=&v"(t1n)
                                         : "{v59}"(bp), [sp0] "v"(t1), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4),
                                           [sc] "v"(scale), [om] "v"(oms)
                                         : "v54");
                    } else {
                        // t2 = (1-s) bp; lim = 11 bp; bpn = s p + t2; clip = p > lim (:490); bp = clip ? bp : bpn.
                        // The select needs two wait states after the compare on gfx950: the next row's s*p
                        // (the only independent work there is) and one s_nop.  A v_cmpx that masks the add instead
                        // is slower (exec hazards).
                        if ((q & 1) == 0)
                            asm volatile(D2_ROW_KUR("v[62:63]", "1", "v63", "v56", "%[sp0]", "%[p0]", "%[ta]", "%[p1]")
                                         D2_ROW_KUR("v[56:57]", "0", "v56", "v57", "%[ta]", "%[p1]", "%[tb]", "%[p2]")
                                         D2_ROW_KUR("v[56:57]", "1", "v57", "v58", "%[tb]", "%[p2]", "%[ta]", "%[p3]")
                                         D2_ROW_KUR("v[58:59]", "0", "v58", "v59", "%[ta]", "%[p3]", "%[tn]", "%[p4]")
                                         : "=&{v[56:59]}"(o), [ta] "=&v"(ta), [tb] "=&v"(tb), [tn] "=&v"(t1n)
                                         : "{v63}"(bp), [sp0] "v"(t1), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3),
                                           [p4] "v"(p4), [sc] "v"(scale), [cf] "v"(coef)
                                         : "vcc", "v54", "v55", "v62");
                        else
                            asm volatile(D2_ROW_KUR("v[58:59]", "1", "v59", "v60", "%[sp0]", "%[p0]", "%[ta]", "%[p1]")
                                         D2_ROW_KUR("v[60:61]", "0", "v60", "v61", "%[ta]", "%[p1]", "%[tb]", "%[p2]")
                                         D2_ROW_KUR("v[60:61]", "1", "v61", "v62", "%[tb]", "%[p2]", "%[ta]", "%[p3]")
                                         D2_ROW_KUR("v[62:63]", "0", "v62", "v63", "%[ta]", "%[p3]", "%[tn]", "%[p4]")
                                         : "=&{v[60:63]}"(o), [ta] "=&v"(ta), [tb] "=&v"(tb), [tn] "=&v"(t1n)
                                         : "{v59}"(bp), [sp0] "v"(t1), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3),
                                           [p4] "v"(p4), [sc] "v"(scale), [cf] "v"(coef)
                                         : "vcc", "v54", "v55", "v58");
                    }
                    t1 = t1n;
                    bp = o.w;
                    *(q4 *)&s_u[buf][q][lane] = o;
                }
#undef D2_ROW_KUR
#undef D2_ROW_RAW
                cu.next(cps);
            }
            step_barrier();
        }
        *bpp = bp;
    } else if (bi >= 0) {
        // ---- B.  wave -> (chunk parity, half of the lane tasks); lane task -> (8-row group, channel), both pols
#if D2_PRIO_B
        __builtin_amdgcn_s_setprio(D2_PRIO_B);
#endif
        const int par = bi >> 1, idxB = (bi & 1) * 64 + lane;
        const int g = idxB >> 5, ch = idxB & 31;
        const bool active = g < NG;
        const int cB = cg * 32 + ch;
        uint8_t *codes = a.codes + ((size_t)ant * 2 + stream) * a.S * a.trim;
        float *ave = a.ave ? a.ave + ((size_t)ant * 2 + stream) * a.S * a.ave_per_seg : nullptr;
        if (a.ave_target && ant == 0 && stream == a.target_stream) ave = a.ave_target;
        BState<NPOL> bs;
        Cursor<DEPTH> cu;                // this wave's next chunk: those of its parity
        cu.init(0, cps);
        if (par) cu.next(cps);
        step_barrier();
        for (int k = 0; k < nstep; ++k) {
            // step k: chunk k-1 was finished by A in step k-1.  If it is ours: take it into registers and do
            // rows 0-3; otherwise finish the chunk taken in the previous step (rows 4-7, quantiser)
            const int c1 = k - 1;
            if (c1 >= 0 && (c1 & 1) == par) {
                if (c1 < nchunk && active) {
                    const int slot = cu.slot, ub = c1 & 1;
#pragma unroll
                    for (int pol = 0; pol < 2; ++pol) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bs.p[pol][j] = s_p[slot][g * 8 + j][pol * 32 + ch];
                        bs.u[pol][0] = s_u[ub][2 * g][pol * 32 + ch];
                        bs.u[pol][1] = s_u[ub][2 * g + 1][pol * 32 + ch];
                        // bp before the group's first row: the last row of the quad before it, or of the chunk before
                        bs.uprev[pol] = g == 0 ? s_u0[ub][pol * 32 + ch] : s_u[ub][(g ? 2 * g : 1) - 1][pol * 32 + ch].w;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) bs.w[j] = KUR ? s_w[c1 % 3][g * 8 + j] : 1.f;
                    bs.trow = cu.rb * NG + g;
                    bs.seg = cu.seg;
                    phase_b_rows<NPOL, KUR>(bs, 0);
                }
            } else if (c1 >= 1) {
                if (c1 - 1 < nchunk && active) {
                    phase_b_rows<NPOL, KUR>(bs, 1);
                    phase_b_finish<NPOL, NBIT, KUR>(a, bs, lane, cB, ntime, codes, ave);
                }
                cu.next(cps);
                cu.next(cps);
            }
            step_barrier();
        }
    } else {
        step_barrier();
        for (int k = 0; k < nstep; ++k) step_barrier();
    }
#ifdef D2_STAMP
    if (blockIdx.x == 17 && blockIdx.y == 1 && blockIdx.z == 0 && lane == 0) {
        // accumulated over launches (launches of one stream follow one another): the reader takes differences
        g_d2_stamp[wave & 7][0] += (unsigned long long)clk.work;
        g_d2_stamp[wave & 7][1] += (unsigned long long)clk.wait;
        g_d2_stamp[wave & 7][2] += (unsigned long long)clk.extra;
        g_d2_stamp[wave & 7][3] += 1ull;
    }
    if (wave == 0 && lane == 0 && blockIdx.z == 0) {
        const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
        if (wg < 256) {
            const unsigned n = g_d2_wgcnt[wg]++;
            g_d2_wg[n & 63][wg][0] = wg_t0;
            g_d2_wg[n & 63][wg][1] = __builtin_amdgcn_s_memrealtime();
            g_d2_wg[n & 63][wg][2] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11)) |
                                     ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 8);   // XCC_ID, HW_ID
            g_d2_wg[n & 63][wg][3] = n;
        }
    }
#endif
}

// MODE = rfi_mode: 0 raw stream only, 1 excised only, 2 both (blockIdx.y picks the stream)
template <int T, int NPOL, int NBIT, int MODE, int DEPTH>
__global__ __launch_bounds__(D2_THREADS) void k_detect2(Detect2Args a)
{
    // ring of power chunks filled by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU)
    __shared__ __attribute__((aligned(16))) float s_p[D2_NSLOT][T][64];
    // bp after each row, double-buffered per chunk: [row quad][column = (pol, channel)], four rows to a 16-byte
    // entry; s_u0 = bp before the chunk's first row
    __shared__ __attribute__((aligned(16))) float4 s_u[2][T / 4][64];
    __shared__ float s_u0[2][64];
    __shared__ float s_w[3][T];
    if (MODE == 0 || (MODE == 2 && blockIdx.y == 0)) detect2_body<T, false, NPOL, NBIT, DEPTH>(a, s_p, s_u, s_u0, s_w);
    else detect2_body<T, true, NPOL, NBIT, DEPTH>(a, s_p, s_u, s_u0, s_w);
}

template <int T, int NPOL, int NBIT, int DEPTH>
static void launch_mode(const Detect2Args &a, int mode, dim3 grid, hipStream_t st)
{
    if (mode == 0) k_detect2<T, NPOL, NBIT, 0, DEPTH><<<grid, D2_THREADS, 0, st>>>(a);
    else if (mode == 1) k_detect2<T, NPOL, NBIT, 1, DEPTH><<<grid, D2_THREADS, 0, st>>>(a);
    else k_detect2<T, NPOL, NBIT, 2, DEPTH><<<grid, D2_THREADS, 0, st>>>(a);
}

template <int T, int DEPTH>
static void launch_all(const Detect2Args &a, int mode, int npol, int nbit, dim3 grid, hipStream_t st)
{
    if (npol == 1) {
        if (nbit == 8) launch_mode<T, 1, 8, DEPTH>(a, mode, grid, st);
        else if (nbit == 4) launch_mode<T, 1, 4, DEPTH>(a, mode, grid, st);
        else launch_mode<T, 1, 2, DEPTH>(a, mode, grid, st);
    } else {
        if (nbit == 8) launch_mode<T, 2, 8, DEPTH>(a, mode, grid, st);
        else if (nbit == 4) launch_mode<T, 2, 4, DEPTH>(a, mode, grid, st);
        else launch_mode<T, 2, 2, DEPTH>(a, mode, grid, st);
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
    a.ave_target = h->cfg.keep_ave ? h->d_coadd_target : nullptr;
    a.target_stream = h->cfg.rfi_mode == 0 ? 0 : 1;
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
#if defined(D2_ABL) && (D2_ABL & 2)
    grid.y = 1;      // (timing experiment, RESULTS INVALID: only the raw stream's 128 workgroups -- what the step would be
                     //  with half of detect's workgroups at their present lifetime, the bound of a two-stream workgroup)
#endif
    // three chunks in flight where detect runs wholly beside the next batch's channeliser (it flags its own rows and
    // starts straight behind the previous one), two otherwise (measured both ways, see the comment on DEPTH)
    const int depth_env = h->sched.detect_depth;     // PB_DETECT_DEPTH 2 / 3: timing experiments
    const bool deep = depth_env ? depth_env == 3
                                : (D2_DEPTH_OVERLAPPED == 3 && pb_fused_kurtosis(h) && h->cfg.taps == 1 && h->sets.size() >= 2 && h->A == 1);
    if (h->R % 32 == 0) {
        if (deep) launch_all<32, 3>(a, h->cfg.rfi_mode, h->cfg.npol, h->cfg.nbit, grid, h->stream);
        else launch_all<32, 2>(a, h->cfg.rfi_mode, h->cfg.npol, h->cfg.nbit, grid, h->stream);
    } else {
        launch_all<8, 2>(a, h->cfg.rfi_mode, h->cfg.npol, h->cfg.nbit, grid, h->stream);
    }
    return hipGetLastError();
}
