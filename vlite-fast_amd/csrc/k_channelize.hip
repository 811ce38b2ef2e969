// Channeliser: 8-bit unpack + 12500-point real FFT + square-law detect of the 4096 output
// channels, one workgroup per (row, pol), everything between the raw bytes and the
// power spectrum staying in registers and LDS.
//
// Replaces, per FFT row:  convertarray (src/pb_kernels.cu:23-33), the copy/zero half of
// apply_kurtosis (:267-287), cufftExecR2C (src/process_baseband.cu:1222-1224), inject_frb
// (src/pb_kernels.cu:348-391), the |X|^2 of detect_and_normalize2/3 (:416, :481) and the
// division by the row weight of detect_and_normalize3 (:452, :481).
//
// FFT: the 12500 real samples are packed as 6250 complex, transformed by three Stockham
// passes of radix 25, 25, 10 and split back into the real-input spectrum.  The operation
// order (butterfly formulas, where fmaf is used, twiddle tables) is the specification in
// oracle/pb_oracle.c "K6"; results agree with the oracle bit for bit.  Pass-1 butterfly input
// r comes from 500-sample kurtosis block r, so a flagged block is simply a zeroed input.
//
// Passes (rfi_mode 2): every (row, pol) workgroup transforms the unflagged data and writes the raw
// power plane -- and, when the row has no flagged block, the excised plane as well (same spectrum,
// divided by the row weight).  Only when some block is flagged does it run a second pass on the
// zeroed data (or write +inf, "no data", if every block is flagged).
//
// LDS: one 6250 x float2 buffer (50 000 B) used in place: every pass reads its inputs into
// registers, barriers, then writes.  3 workgroups per CU.
// HBM traffic per (row, pol): 12.5 KB of samples read, 16 KB + 16 KB of power written;
// twiddles (tw2 4.9 KB, tw3 49 KB, post 49 KB) stay in L2.
#include <cstdlib>
#include <cstring>

#include "pb_internal.h"

#ifdef CH_STAMP
// timing experiments (variant builds only): the clock at the phase boundaries of every workgroup's first
// transform (+ its XCC id and flag mask); read back with pb_internal_ch_stamps
#define CH_STAMP_MAXWG 40960
__shared__ unsigned long long ch_ts[9];
__device__ unsigned long long g_ch_stamp[CH_STAMP_MAXWG][13];
extern "C" int pb_internal_ch_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ch_stamp), sizeof(g_ch_stamp));
}
#ifndef CH_STAMP_TID
#define CH_STAMP_TID 0
#endif
#define FFT_STAMP(i) do { if (threadIdx.x == CH_STAMP_TID) ch_ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "fft_lds.h"
#include "kurtosis_dev.h"

#ifndef CH_ABL
#define CH_ABL 0      // energy / timing experiments (variant builds only, results invalid): 2 no plane stores, 4 no
                      // spectrum step, 8 no FFT passes, 16 no staging / unpack of the transforms  (1, no moments of the
                      // statistic, and 32, dword reads for the moments, went with the shared row_block_moments;
                      // their results are in profiles/r04_notes.md section 8)
#endif
struct ChanArgs {
    const uint8_t *in;      // [A][S][2][seg_samples]
    size_t in_ant_stride, seg_samples;
    const float *wrow;        // [A][S*R] row weights and
    const uint32_t *rowmask;  // flag masks from k_kurtosis_row (bit r = block r flagged)
    size_t wrow_ant_stride;
    float *Praw, *Pkur;     // [A][S][2][R][4096]
    size_t p_ant_stride;
    const float2 *tw2, *tw3, *post;
    const float2 *postc;    // post[2155..6250], 16-byte aligned copy
    FrbParams frb;          // delays == nullptr: no injection
    int R, rfi_mode, inject_now;
    // k_channelize_kur only: where the row's flags, weight and mask go, and the D'Agostino constants
    uint8_t *flags;         // [A][S*R*25]
    size_t flags_ant_stride;
    float *wrow_out;
    uint32_t *rowmask_out;
    const DagConsts *dag;
};

// The row's 12500 bytes go to LDS through 16-byte loads (narrow per-lane loads are bound by the address unit,
// not by HBM: 2-byte loads move 128 B per wave instruction, these 1 KiB).  Rows are only 4-byte aligned, so the
// aligned 16-byte chunks that cover the row are fetched (d_in is padded for the overhang): chunks tid, tid+256,
// tid+512 always exist (782 or 783 in all), tid+768 for tid < 15.
struct RowStage {
    uint4 t0, t1, t2, t3;
};
__device__ __forceinline__ size_t row_byte(const ChanArgs &a, int seg, int row, int pol, int ant)
{
    return (size_t)ant * a.in_ant_stride + ((size_t)seg * 2 + pol) * a.seg_samples + (size_t)row * PB_NFFT;
}
__device__ __forceinline__ void stage_request(const ChanArgs &a, int tid, int seg, int row, int pol, int ant, RowStage &st)
{
    const size_t rbyte = row_byte(a, seg, row, pol, ant);
    const unsigned o = (unsigned)(rbyte & 15);   // offset of the row's first byte in its first chunk
    const uint4 *src16 = (const uint4 *)(a.in + (rbyte - o));
    const int nch = (int)((o + PB_NFFT + 15) >> 4);   // <= 783
    st.t0 = src16[tid];
    st.t1 = src16[tid + 256];
    st.t2 = src16[tid + 512];
    st.t3 = make_uint4(0u, 0u, 0u, 0u);
    if (tid + 768 < nch) st.t3 = src16[tid + 768];
}

// x / (1 + 2^-23), the division of a row without flags (87 % of rows on clean noise), on the bits: the quotient
// is the first or the second float below x -- the second when x's mantissa field m is 0 or >= 0x400002
// (m 2^-23 >= 1.5 + ...: the exact quotient m (1 - 2^-23 + 2^-46 - ...) lies below the midpoint) -- checked
// against the IEEE division for every positive normal binary32 from 0x00800002 up (tools/check_div_full_weight.c);
// smaller values, zeros and non-finite ones take the division.  4 instructions instead of 11.
__device__ __forceinline__ float div_full_weight(float x)
{
    const unsigned b = __builtin_bit_cast(unsigned, x);
    const unsigned m1 = (b & 0x7fffffu) - 1u;
    const float fast = __builtin_bit_cast(float, b - (m1 >= 0x400001u ? 2u : 1u));
    if (__builtin_expect(b - 0x00800002u >= 0x7f800000u - 0x00800002u, 0)) return x / 1.00000011920928955078125f;
    return fast;
}

// One transform of one (row, pol): ROLE 0 = raw spectrum (also fills the excised plane when the row
// has no flagged block), ROLE 1 = excised spectrum (flagged blocks zeroed).  The row's bytes have been
// requested by stage_request; mask and wrow may still be in flight (ROLE 0 first needs them after the FFT).
// next_row >= 0: the bytes of that row (of the same segment and pol) are requested into st as soon as the FFT is
// done, so that their latency runs under this transform's spectrum step.
// FIX: code 0 -> 128 on every staging (the input buffer has not been patched by a kurtosis pass); next_pol: the
// polarisation of the next transform's row.
// prestaged: the row's (patched) bytes already lie at the start of buf -- the statistic of k_channelize_kur staged
// them there -- so nothing is staged and no barrier precedes the unpack.
// DEFER (k_channelize_kur, ROLE 0): the row's flag mask and weight are not known yet when the transform starts (one
// wave works them out while the others unpack); they are read from `smw` after the FFT, where ROLE 0 first needs
// them, and what to request next (the same row again for its excised transform, else `after_row` of pol 1) is decided
// there too.
template <int ROLE, bool FIX = false, bool DEFER = false>
__device__ __forceinline__ void channelize_pass(const ChanArgs &a, f2 *buf, int tid, int seg, int row, int pol, int ant,
                                                RowStage &st, unsigned mask, float wrow, size_t prow, int next_row,
                                                int next_pol = -1, bool prestaged = false,
                                                const unsigned *smw = nullptr, int after_row = -1)
{
    if (next_pol < 0) next_pol = pol;
    FFT_STAMP(0);
    const unsigned o = (unsigned)(row_byte(a, seg, row, pol, ant) & 15);   // recomputed, not carried from the request
    if (!prestaged) {
        // a dropped-frame byte (0) means "no sample" = 0.0 = code 128 (convertarray :23-33), so that the
        // conversion below is one fma per sample pair.  The kurtosis kernel has already patched such codes in
        // the input buffer; only RFI mode 0, which has no kurtosis pass, does it here (four per instruction).
        uint4 *stage = (uint4 *)buf;
        const bool last = tid + 768 < (int)((o + PB_NFFT + 15) >> 4);
        if (FIX || a.rfi_mode == 0) {
            stage[tid] = fix_zero_codes(st.t0);
            stage[tid + 256] = fix_zero_codes(st.t1);
            stage[tid + 512] = fix_zero_codes(st.t2);
            if (last) stage[tid + 768] = fix_zero_codes(st.t3);
        } else {
            stage[tid] = st.t0;
            stage[tid + 256] = st.t1;
            stage[tid + 512] = st.t2;
            if (last) stage[tid + 768] = st.t3;
        }
        __syncthreads();
    }
    FFT_STAMP(1);
    f2 v[25];
    if (tid < 250) {
        const unsigned zmask = ROLE == 1 ? mask : 0u;
        const uint16_t *sb = (const uint16_t *)((const uint8_t *)buf + o);
#pragma unroll
        for (int r = 0; r < 25; ++r) {
            unsigned w = sb[tid + 250 * r];
            if ((zmask >> r) & 1u) w = 0x8080u;   // flagged block -> zeros (apply_kurtosis :243-295)
            v[r] = cvt_pair_c(w);
        }
    }
    __syncthreads();   // all samples are in registers before pass 1 overwrites buf
    FFT_STAMP(2);
    // The spectrum step's eight twiddle loads are requested from inside pass 3: one L2 latency for the
    // whole step instead of one per 1024-channel slice, and most of it hidden under pass 3 (the step was
    // 25-35 % of a workgroup's life, almost all of it waiting).
    float4 tq[4][2];
    auto load_tq = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tq[i][0] = *(const float4 *)(a.postc + tid * 4 + 1024 * i);
            tq[i][1] = *(const float4 *)(a.postc + tid * 4 + 1024 * i + 2);
        }
    };
#if CH_ABL & 8
    load_tq();
    if (tid < 250)
        for (int r = 0; r < 25; ++r) buf[tid * 25 + r] = v[r];
    __syncthreads();
#else
    fft6250(v, buf, (const f2 *)a.tw2, (const f2 *)a.tw3, tid, load_tq);
#endif
    if (DEFER) {
        // (written by wave 0 before the barrier in front of pass 1)
        mask = __builtin_amdgcn_readfirstlane(smw[0]);
        wrow = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(smw[1]));
        const bool again = (a.rfi_mode == 1 || (a.rfi_mode == 2 && mask != 0)) && mask != 0x1ffffffu;
        next_row = again ? row : after_row;
        next_pol = again ? pol : 1;
    }
#ifndef FFT_LEAN
    if (next_row >= 0) stage_request(a, tid, seg, next_row, next_pol, ant, st);
#endif

    // FRB injection window of this row, per channel (inject_frb :361-380)
    const bool inject = a.frb.delays != nullptr && a.inject_now > 0;
    const int since = inject ? (a.inject_now - 1 + seg) * a.R : 0;
    const bool also_kur = a.rfi_mode == 2 && ROLE == 0 && mask == 0;
    // the weight of a row without flags: 25 x 0.04f summed left to right = 1 + 2^-23
    const bool full_w = __builtin_bit_cast(unsigned, wrow) == 0x3f800001u;
    float *P0 = (ROLE == 1 ? a.Pkur : a.Praw) + prow;
    float *P1 = a.Pkur + prow;
    // four consecutive channels per thread: 16-byte twiddle loads and 16-byte power stores (one
    // channel per lane would spare the 8-way LDS bank conflicts of these reads, but its 4-byte stores
    // measured 9 % slower overall)
#pragma unroll
    for (int i = 0; i < ((CH_ABL & 4) ? 0 : 4); ++i) {
        const int c4 = tid * 4 + 1024 * i;
        const float4 t01 = tq[i][0], t23 = tq[i][1];
        const f2 tw[4] = {mk2(t01.x, t01.y), mk2(t01.z, t01.w), mk2(t23.x, t23.y), mk2(t23.z, t23.w)};
        float pw[4];
        const int k0 = PB_CHANMIN + c4;
        f2 zas[4], zbs[4];
        read_z_pairs(buf, k0, zas, zbs);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q;
            const f2 za = zas[q];
            const f2 zb = zbs[q];
            f2 E, O;
            addsub_conj(za, zb, E, O);
            const f2 Pq = cmul(O, tw[q]);
            f2 X = mk2(0.5f, 0.5f) * (E + Pq);
            if (inject) {
                const float d = a.frb.delays[k];
                const int lo = (int)(d + 0.5) - since;
                const int hi = (int)(d + a.frb.width + 0.5) - since;
                if (row >= lo && row <= hi) X = X * mk2(a.frb.amp, a.frb.amp);
            }
            const f2 sq = X * X;
            pw[q] = sq.x + sq.y;
        }
        // the excised plane carries pow / w (detect_and_normalize3 :452,:481), so that the
        // serial bandpass recurrence downstream has no division in it
        if (ROLE == 0) store_plane4(P0, c4, pw[0], pw[1], pw[2], pw[3]);
        if (ROLE == 1)
            store_plane4(P0, c4, pw[0] / wrow, pw[1] / wrow, pw[2] / wrow, pw[3] / wrow);
        else if (also_kur)
        {
            if (full_w)
                store_plane4(P1, c4, div_full_weight(pw[0]), div_full_weight(pw[1]), div_full_weight(pw[2]),
                             div_full_weight(pw[3]));
            else
                store_plane4(P1, c4, pw[0] / wrow, pw[1] / wrow, pw[2] / wrow, pw[3] / wrow);
        }
    }
#ifdef FFT_LEAN
    if (next_row >= 0) stage_request(a, tid, seg, next_row, next_pol, ant, st);
#endif
#ifdef CH_STAMP
    if (ROLE == 0 && threadIdx.x == CH_STAMP_TID) {
        ch_ts[7] = __builtin_amdgcn_s_memtime();
        const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
        if (wg < CH_STAMP_MAXWG && blockIdx.z == 0) {
            for (int i = 0; i < 8; ++i) g_ch_stamp[wg][i] = ch_ts[i];
            g_ch_stamp[wg][8] = __builtin_amdgcn_s_getreg(20 | (3 << 11));   // HW_REG_XCC_ID[3:0]
            g_ch_stamp[wg][9] = mask;
            g_ch_stamp[wg][10] = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID
            g_ch_stamp[wg][11] = __builtin_amdgcn_s_memrealtime();
            g_ch_stamp[wg][12] = ch_ts[8];   // kernel entry
        }
    }
#endif
}

// Three workgroups per CU (LDS 3 x 50 000 B, <= 168 VGPRs).  A workgroup takes CH_ROWS consecutive rows of one
// (segment, pol) and requests the bytes of its next transform (the same row again when it has flagged blocks,
// else the next row) while the current one is in its spectrum step.  Measured (profiles/r02_notes.md, per-workgroup
// clock stamps): the request-to-LDS latency is 3000 cycles of a 22 000-cycle transform and a workgroup hand-over
// ~1500, yet 2 rows per workgroup are only 1-2 % faster than 1, and 8 or 16 are slower: the three co-resident
// workgroups fill each other's waits, and longer runs of rows bring the workgroups of a CU into step.
// Measured earlier (profiles/r01 notes):
//  * capping residency at two to let a detect workgroup of the previous batch co-reside does not
//    pay -- both kernels are bound by VALU issue, so side by side they only stretch each other;
//  * fully persistent workgroups (768 for the whole launch) were 12 % SLOWER than one workgroup per (row, pol):
//    the dispatcher's staggered starts overlap the workgroups' load, FFT and store phases better than a
//    lock-step loop does.  Short runs of rows keep that staggering.
#ifndef CH_ROWS
#define CH_ROWS 2
#endif
#ifdef FFT_LEAN
// (dynamic LDS: with a static 50-KB buffer the compiler sees three workgroups per CU and takes 168 registers)
__global__ __launch_bounds__(256, 4) void k_channelize(ChanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) f2 buf[];
#else
__global__ __launch_bounds__(256, 3) void k_channelize(ChanArgs a)
{
    __shared__ __attribute__((aligned(16))) f2 buf[M_HALF];
#endif
    FFT_STAMP(8);
    int tid = threadIdx.x;
    // grid (ceil(R / CH_ROWS), nseg * 2, A): no division to find the rows
    const int row0 = blockIdx.x * CH_ROWS, seg = blockIdx.y >> 1, pol = blockIdx.y & 1, ant = blockIdx.z;
    const int nrow = min(CH_ROWS, a.R - row0);

    // The first row's bytes are requested first; the rows' flag masks and weights (written by the kurtosis kernel:
    // lane i of every wave fetches the mask of row i, lane CH_ROWS + i its weight) follow, so that neither latency
    // precedes the other (25 flag bytes fetched before any sample was requested cost 1.7 us per workgroup).
    RowStage st;
    stage_request(a, tid, seg, row0, pol, ant, st);
    unsigned vmw = 0;
    if (a.rfi_mode) {
        const int l = tid & 63, i = l % CH_ROWS;
        const size_t wi = (size_t)ant * a.wrow_ant_stride + (size_t)seg * a.R + row0 + i;
        if (l < 2 * CH_ROWS && i < nrow) vmw = l < CH_ROWS ? a.rowmask[wi] : __builtin_bit_cast(unsigned, a.wrow[wi]);
    }
#pragma unroll 1
    for (int i = 0; i < nrow; ++i) {
        const int row = row0 + i;
        const int after = i + 1 < nrow ? row + 1 : -1;
        // row weight exactly as apply_kurtosis accumulates it: one 500/12500 per unflagged block
        const unsigned mask = __builtin_amdgcn_readlane(vmw, i);
        const float wrow = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vmw, CH_ROWS + i));
        const size_t prow = (size_t)ant * a.p_ant_stride + (((size_t)seg * 2 + pol) * a.R + row) * PB_NCHANOUT;
        const bool all_bad = mask == 0x1ffffffu;
        // transforms of this row: role 0 = raw spectrum (also fills the excised plane when the row has no
        // flagged block), role 1 = excised spectrum (only when some block is flagged: 13 % of rows on clean
        // noise), instead of launching a second grid of workgroups of which 87 % would exit at once.
        const bool second = a.rfi_mode == 1 || (a.rfi_mode == 2 && mask != 0);
        if (i) __syncthreads();   // the previous transform has finished reading buf
        if (a.rfi_mode != 1) {
            channelize_pass<0>(a, buf, tid, seg, row, pol, ant, st, mask, wrow, prow, second && !all_bad ? row : after);
            if (!second) continue;
        }
        if (all_bad) {
            // weight 0: the row's excised power is never used for its value (detect_and_normalize3
            // :474-476 writes 0 and leaves the bandpass alone); +inf makes the detect kernel's clip test
            // do exactly that without having to look at the weight
            for (int c = tid; c < PB_NCHANOUT; c += 256) a.Pkur[prow + c] = __builtin_inff();
            if (a.rfi_mode == 1 && after >= 0) stage_request(a, tid, seg, after, pol, ant, st);
            continue;
        }
        if (a.rfi_mode == 2) {
            __syncthreads();   // the raw pass has finished reading buf
            asm volatile("" : "+v"(tid));   // no sharing of tid-derived addresses across the two passes
        }
        channelize_pass<1>(a, buf, tid, seg, row, pol, ant, st, mask, wrow, prow, after);
    }
}

// ---- the channeliser that computes the flags of its own row (RFI modes 1 and 2, taps = 1) ----
// One workgroup = one FFT row of BOTH polarisations: the flag of a 500-sample block is the max over the two pols of
// the D'Agostino score, so the row's 2 x 25 blocks are reduced first (k_kurtosis_row's phase: the two rows staged in
// the FFT buffer, the reference's halving tree per block, 50 lanes of double-precision scores, ballot -> mask and
// weight, also written out for detect and the debug readers), then the transforms of pol 0 and pol 1 follow as in
// k_channelize, the next transform's bytes requested while the current one is in its spectrum step.  What it saves
// against kurtosis kernel + channeliser: the second read of every byte (256 MB per second of data), a kernel whose
// only overlap partner was the previous batch's detect (0.06 ms alone, 0.13 there), and a cross-stream hand-over;
// the next batch's channeliser then starts straight behind this one and detect runs wholly beside it.
// The input buffer is NOT patched for code 0 here (nobody reads it again but this workgroup): every staging fixes
// the codes on the way (FIX).
#ifdef KUR_STAMP
// timing experiments (variant builds only, tools/kur_stamps.py): the clock at the phase boundaries of every workgroup
__device__ unsigned long long g_kur_stamp[4][10240][10];   // the last four launches (a.frb.since counts them); [8], [9]: s_memrealtime (100 MHz, one clock for the chip) at entry / exit
extern "C" int pb_internal_kur_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kur_stamp), sizeof(g_kur_stamp));
}
#define KSTAMP(i) do { if (threadIdx.x == 0) kts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KSTAMP(i)
#endif
__global__ __launch_bounds__(256, 3) void k_channelize_kur(ChanArgs a)
{
    __shared__ __attribute__((aligned(16))) f2 buf[M_HALF];
    __shared__ unsigned smw[2];     // the row's flag mask and weight bits (outside buf: the transforms overwrite that)
#ifdef KUR_STAMP
    __shared__ unsigned long long kts[8];
#endif
    KSTAMP(0);
#ifdef KUR_STAMP
    if (threadIdx.x == 0) kts[6] = __builtin_amdgcn_s_memrealtime();
#endif
    int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.x, seg = blockIdx.y, ant = blockIdx.z;
    const int grow = seg * a.R + row;

    // both rows' bytes go to LDS, patched for code 0, for the statistics; pol 0's stay where its first transform
    // expects them (the start of buf), so that transform stages nothing
    RowStage st;
    unsigned off0, off1;
    DagConsts dc;
    uint4 *sraw0 = (uint4 *)buf, *sraw1 = (uint4 *)buf + 784;     // 2 x 12 544 B
    float *s2 = (float *)((uint4 *)buf + 2 * 784), *s4 = s2 + 50;
    {
        RowStage sb;
        stage_request(a, tid, seg, row, 0, ant, st);
        stage_request(a, tid, seg, row, 1, ant, sb);
#ifndef KUR_NO_DAG_PREFETCH
        // the D'Agostino constants are requested with the rows: their first use is on the critical path of the wave
        // that decides the flags, and a scalar load from memory is a microsecond there
        dc = *a.dag;
#endif
        off0 = (unsigned)(row_byte(a, seg, row, 0, ant) & 15);
        off1 = (unsigned)(row_byte(a, seg, row, 1, ant) & 15);
        const bool l0 = tid + 768 < (int)((off0 + PB_NFFT + 15) >> 4), l1 = tid + 768 < (int)((off1 + PB_NFFT + 15) >> 4);
        sraw0[tid] = fix_zero_codes(st.t0);
        sraw0[tid + 256] = fix_zero_codes(st.t1);
        sraw0[tid + 512] = fix_zero_codes(st.t2);
        if (l0) sraw0[tid + 768] = fix_zero_codes(st.t3);
        sraw1[tid] = fix_zero_codes(sb.t0);
        sraw1[tid + 256] = fix_zero_codes(sb.t1);
        sraw1[tid + 512] = fix_zero_codes(sb.t2);
        if (l1) sraw1[tid + 768] = fix_zero_codes(sb.t3);
    }
    __syncthreads();
    KSTAMP(1);
    // moments of the 50 blocks: k_kurtosis_row's reduction (kurtosis_dev.h)
    row_block_moments((const uint8_t *)sraw0 + off0, (const uint8_t *)sraw1 + off1, wave, lane, s2, s4);
    __syncthreads();
    KSTAMP(2);
    // The flags are the business of ONE wave: lanes 0..49 decide their block's flag (no cube root, kurtosis_dev.h), a
    // ballot brings the two pols together (bit b of the mask = block b flagged in pol 0 or pol 1: compute_dagostino's
    // max over pols, :109-134), lane 0 books the weight.  The other waves go straight on to unpacking pol 0, whose
    // bytes are where the transform expects them; the mask and the weight are first needed after that transform's FFT
    // (RFI mode 2), and they reach the other waves through `smw` across the barrier that precedes pass 1 -- which
    // also keeps pass 1 from overwriting the moments before this wave has read them.
    if (wave == 0) {
        bool f = false;
        if (lane < 50) {
            const float p = s2[lane] / PB_NKURTO;
            const float k = s4[lane] / PB_NKURTO / (p * p);
#ifndef KUR_NO_DAG_PREFETCH
            f = dag_flag(k, dc);
#else
            f = dag_flag(k, *a.dag);
#endif
        }
        const unsigned long long b = __ballot(f);
        const uint32_t m = (uint32_t)((b | (b >> 25)) & 0x1ffffffull);
        if (lane < 25) a.flags[(size_t)ant * a.flags_ant_stride + (size_t)grow * PB_BLK_PER_FFT + lane] = (m >> lane) & 1u;
        if (lane == 0) {
            // kur_weights after apply_kurtosis (:292): one 500/12500 per unflagged block, summed left to right
            const float inc = (float)PB_NKURTO / PB_NFFT;
            float w = 0.f;
            for (int k = PB_BLK_PER_FFT - __popc(m); k > 0; --k) w = w + inc;
            a.wrow_out[(size_t)ant * a.wrow_ant_stride + grow] = w;
            a.rowmask_out[(size_t)ant * a.wrow_ant_stride + grow] = m;
            smw[0] = m;
            smw[1] = __builtin_bit_cast(unsigned, w);
        }
    }
    KSTAMP(3);
    unsigned mask = 0;
    float wrow = 0.f;
    if (a.rfi_mode == 1) {
        // the excised transform zeroes flagged blocks while it unpacks: it needs the mask first
        __syncthreads();
        mask = __builtin_amdgcn_readfirstlane(smw[0]);
        wrow = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(smw[1]));
    }
    KSTAMP(4);

#pragma unroll 1
    for (int pol = 0; pol < 2; ++pol) {
        const size_t prow = (size_t)ant * a.p_ant_stride + (((size_t)seg * 2 + pol) * a.R + row) * PB_NCHANOUT;
        const int after_row = pol == 0 ? row : -1;         // what follows this pol's transforms: pol 1 of the same row
        if (pol) __syncthreads();   // the previous transform has finished reading buf
        if (a.rfi_mode != 1) {
            // what is requested while this transform is in its spectrum step -- the same row again for its excised
            // transform, else pol 1's row -- is decided inside, once the mask is known
            channelize_pass<0, true, true>(a, buf, tid, seg, row, pol, ant, st, 0u, 0.f, prow, -1, 1, pol == 0, smw, after_row);
            mask = __builtin_amdgcn_readfirstlane(smw[0]);
            wrow = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(smw[1]));
        }
        const bool all_bad = mask == 0x1ffffffu;
        const bool second = a.rfi_mode == 1 || (a.rfi_mode == 2 && mask != 0);
        if (!second) continue;
        if (all_bad) {
            for (int c = tid; c < PB_NCHANOUT; c += 256) a.Pkur[prow + c] = __builtin_inff();
            if (a.rfi_mode == 1 && after_row >= 0) stage_request(a, tid, seg, after_row, 1, ant, st);
            continue;
        }
        if (a.rfi_mode == 2) {
            __syncthreads();   // the raw pass has finished reading buf
            asm volatile("" : "+v"(tid));   // no sharing of tid-derived addresses across the two passes
        }
        // (RFI mode 1, pol 0: the bytes the statistic staged are still where this transform expects them)
        channelize_pass<1, true>(a, buf, tid, seg, row, pol, ant, st, mask, wrow, prow, after_row, 1,
                                 a.rfi_mode == 1 && pol == 0);
    }
#ifdef KUR_STAMP
    if (threadIdx.x == 0 && blockIdx.z == 0) {
        const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
        if (wg < 10240) {
            unsigned long long(*g)[10] = g_kur_stamp[a.frb.since & 3];
            for (int i = 0; i < 5; ++i) g[wg][i] = kts[i];
            g[wg][5] = __builtin_amdgcn_s_memtime();
            g[wg][6] = mask;
            g[wg][7] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11)) |       // HW_REG_XCC_ID
                       ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 8);  // HW_REG_HW_ID
            g[wg][8] = kts[6];
            g[wg][9] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

static bool check_consts(std::string &why)
{
    // the literals of fft_consts.h must be what libm gives here and in the oracle
    auto same = [](float a, double b) { float c = (float)b; return memcmp(&a, &c, 4) == 0; };
    bool ok = same(FC_C1, cos(2.0 * M_PI / 5.0)) && same(FC_C2, cos(4.0 * M_PI / 5.0)) &&
              same(FC_S1, sin(2.0 * M_PI / 5.0)) && same(FC_S2, sin(4.0 * M_PI / 5.0));
    for (int x = 1; x < 5 && ok; ++x)
        for (int y = 1; y < 5 && ok; ++y) {
            const double ang = 2.0 * M_PI * (double)(x * y) / 25.0;
            ok = same(kW25[x * 5 + y][0], cos(ang)) && same(kW25[x * 5 + y][1], -sin(ang));
        }
    for (int k = 1; k < 5 && ok; ++k) {
        const double ang = 2.0 * M_PI * (double)k / 10.0;
        ok = same(kW10[k][0], cos(ang)) && same(kW10[k][1], -sin(ang));
    }
    if (!ok) why = "fft_consts.h does not match this host's libm; rerun tools/gen_fft_consts.py";
    return ok;
}

hipError_t launch_channelize(pb_handle *h, int nseg, int inject_now)
{
    static int consts_ok = -1;
    if (consts_ok < 0) {
        std::string why;
        consts_ok = check_consts(why) ? 1 : 0;
        if (!consts_ok) h->err = why;
    }
    if (!consts_ok) return hipErrorInvalidValue;
    if (nseg <= 0) return hipSuccess;
    ChanArgs a;
    a.in = h->d_in;
    a.in_ant_stride = (size_t)h->S * 2 * h->seg_samples;
    a.seg_samples = h->seg_samples;
    a.wrow = h->d_wrow;
    a.rowmask = pb_rowmask(h);
    a.wrow_ant_stride = (size_t)h->S * h->R;
    a.Praw = h->d_Praw;
    a.Pkur = h->d_Pkur;
    a.p_ant_stride = (size_t)h->S * 2 * h->R * PB_NCHANOUT;
    a.tw2 = h->ft.tw2;
    a.tw3 = h->ft.tw3;
    a.post = h->ft.post;
    a.postc = h->ft.postc;
    a.frb.delays = (inject_now > 0) ? h->d_frb_delays : nullptr;
    a.frb.width = h->frb_width;
    a.frb.amp = h->frb_amp;
#ifdef KUR_STAMP
    static int kur_launches = 0;
    a.frb.since = kur_launches++;     // (unused by the kernels: which of the four stamp buffers)
#else
    a.frb.since = 0;
#endif
    a.R = h->R;
    a.rfi_mode = h->cfg.rfi_mode;
    a.inject_now = inject_now;
    a.flags = h->d_flags;
    a.flags_ant_stride = (size_t)h->S * h->nblk_seg;
    a.wrow_out = h->d_wrow;
    a.rowmask_out = pb_rowmask(h);
    a.dag = h->d_dag;
    if (pb_fused_kurtosis(h)) {
        // one workgroup per row (both pols): it computes the row's flags itself
        dim3 gk((unsigned)h->R, (unsigned)nseg, (unsigned)h->A);
        k_channelize_kur<<<gk, 256, 0, h->stream>>>(a);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((h->R + CH_ROWS - 1) / CH_ROWS), (unsigned)(nseg * 2), (unsigned)h->A);
#ifdef FFT_LEAN
    static const int lean_lds = getenv("PB_LEAN_LDS") ? atoi(getenv("PB_LEAN_LDS")) : M_HALF * 8;
    k_channelize<<<grid, 256, lean_lds, h->stream>>>(a);
#else
    k_channelize<<<grid, 256, 0, h->stream>>>(a);
#endif
    return hipGetLastError();
}

// ---- float-input channeliser with optional 4-tap FIR (pb_channelize_f32) ----
// x: (nrows + taps - 1) rows of 12500 floats; out row t = rfft(sum_j taps[j] * x[row t + j]),
// the WOLA form of analysis/baseband.py:1226-1233.
__global__ __launch_bounds__(256) void k_channelize_f32(const float *__restrict__ x, int taps,
                                                        const float *__restrict__ fir, float2 *__restrict__ out,
                                                        const float2 *tw2, const float2 *tw3, const float2 *post)
{
    __shared__ f2 buf[M_HALF];
    const int tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const float *xr = x + row * PB_NFFT;
    f2 v[25];
    if (tid < 250) {
#pragma unroll
        for (int r = 0; r < 25; ++r) {
            const int n = 2 * (tid + 250 * r);
            if (taps == 1) {
                v[r] = *(const f2 *)(xr + n);
            } else {
                float2 acc = make_float2(0.f, 0.f);
                for (int j = 0; j < 4; ++j) {
                    const float2 s = *(const float2 *)(xr + (size_t)j * PB_NFFT + n);
                    const float2 t = *(const float2 *)(fir + (size_t)j * PB_NFFT + n);
                    const float px = t.x * s.x, py = t.y * s.y;
                    acc.x = j ? acc.x + px : px;
                    acc.y = j ? acc.y + py : py;
                }
                v[r] = mk2(acc.x, acc.y);
            }
        }
    }
    fft6250(v, buf, (const f2 *)tw2, (const f2 *)tw3, tid);
    for (int k = tid; k < PB_NCHAN; k += 256) {
        const f2 X = rsplit(buf, (const f2 *)post, k);
        out[row * PB_NCHAN + k] = make_float2(X.x, X.y);
    }
}

hipError_t launch_channelize_f32(pb_handle *h, const float *d_x, int nrows, int taps, float2 *d_out)
{
    hipError_t e = launch_channelize(h, 0, 0);  // validates the compile-time constants
    if (e != hipSuccess) return e;
    k_channelize_f32<<<nrows, 256, 0, h->stream>>>(d_x, taps, h->ft.taps, d_out, h->ft.tw2, h->ft.tw3, h->ft.post);
    return hipGetLastError();
}
