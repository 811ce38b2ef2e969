// taps = 4: the 4-tap Hamming WOLA polyphase window of the reference's NumPy channeliser
// (analysis/baseband.py:1207-1237) in the streaming 8-bit path.
//
// The reference's GPU path has no PFB (rectangular window only); this mode is the north_star's
// "4-tap polyphase FIR window" and is defined here as the causal form of polyphase_filterbank:
//     output row g  =  rfft( sum_{j=0..3} taps[j] (.) v_{g-3+j} ),   v = unpacked voltages,
// i.e. output row g is the reference function's spectrum i = g - 3 of the same sample stream; rows
// before the start of the stream are zeros.  The three most recent rows (and their kurtosis flags)
// are kept per antenna between pb_process calls.  Excision zeroes flagged 500-sample blocks of each
// contributing row before the window is applied; the row weight generalises apply_kurtosis'
// "fraction of unflagged samples" to the window's energy:
//     w(g) = sum_{j,b unflagged and present} E[j][b] / sum_{j,b} E[j][b],  E[j][b] = sum_{m in b} taps[j][m]^2.
//
// FIR arithmetic (the build's own definition -- the reference's is float64 NumPy, matched to 2e-6 of the spectrum
// peak; restated for the parity tests as orc_pfb_fir in oracle/pb_oracle.c): one multiply, then three fused
// multiply-adds in tap order,
//     y = fma(t3, x3, fma(t2, x2, fma(t1, x1, t0 * x0))),          x_j = sample of row g - 3 + j.
// The kernel evaluates the same numbers on scaled operands: the samples stay 8-bit integers s = 128 x = code - 128
// (one SDWA conversion each, straight from the packed bytes) and the table holds t / 128 -- powers of two commute
// with every rounding above.  A flagged block contributes zeros, i.e. its fma is skipped (fma(t, 0, acc) = acc).
//
// Structure (round 3).  A workgroup walks PFB_RUN consecutive output rows of one (segment, pol).  Each thread keeps
// the bytes of ITS 25 sample pairs of the three previous rows in registers (39 VGPRs: a thread's pairs n = tid +
// 250 r are the same for every row), so that an output row stages ONE new row through LDS -- 16-byte loads, then
// 25 two-byte LDS reads per thread -- instead of four rows per transform (the first version: 5.7 row stagings and
// 143 two-byte LDS reads per output row at 1.43 transforms per row), and the excised transform of a row with flags
// reuses the registers.  The FFT is the register-lean build of fft_lds.h (FFT_LEAN: twiddles requested where they
// are used, pass 3 one butterfly at a time) so that FFT + history still fit 168 VGPRs = three workgroups per CU.
// Parity: spectra vs the polyphase_filterbank golden (2e-6); the whole chain bit-exact vs the oracle's kernels
// composed around orc_pfb_fir, RFI modes 0 and 2, pipelined buffer sets, R = 1024 (tests/test_gpu_pfb.py).
#ifndef PFB_WGS
#define PFB_WGS 2            // workgroups per CU the register budget is set for
#endif
#ifndef PFB_LEAN
#define PFB_LEAN (PFB_WGS >= 3)
#endif
#if PFB_LEAN
#define FFT_LEAN             // the register-lean FFT, nothing requested ahead
#endif
#ifndef PFB_TQ
#define PFB_TQ 0             // 1: the spectrum step's twiddles are requested from inside pass 3 (32 VGPRs)
#endif
#include "fft_lds.h"

#ifndef PFB_RUN
#define PFB_RUN 8            // output rows per workgroup (R is a multiple of 8)
#endif
#define PFB_HIST_STRIDE 12512
#define PFB_NW 13            // packed words per row and thread: 25 pairs of signed bytes, two pairs per word

struct PfbArgs {
    const uint8_t *in;       // [A][S][2][seg_samples]
    size_t in_ant_stride, seg_samples;
    const uint8_t *hist;     // [A][2][3][PFB_HIST_STRIDE]
    const uint8_t *hflags;   // [A][3][25]
    const uint8_t *hvalid;   // [A][3] history slot holds data
    const float *wrow;       // [A][S*R]  (already the PFB weights)
    const uint32_t *rowmask; // [A][S*R]  flag masks of the rows (k_kurtosis_row)
    size_t wrow_ant_stride;
    const float2 *fir;       // [6250 n][4 taps] coefficient pairs of samples (2n, 2n+1), scaled by 1/128 (FftTables::taps_n)
    float *Praw, *Pkur;
    size_t p_ant_stride;
    const float2 *tw2, *tw3, *postc;
    FrbParams frb;
    int R, rfi_mode, inject_now;
};

// flags of the 25 blocks of row rr (rr < 0: history slot 3 + rr; a slot that holds no data yet -- start of the
// stream -- is all zeros: nothing to excise there, its missing weight is booked by k_pfb_weights)
__device__ __forceinline__ unsigned row_mask(const PfbArgs &a, int ant, int rr)
{
    if (rr >= 0) return __builtin_amdgcn_readfirstlane(a.rowmask[(size_t)ant * a.wrow_ant_stride + rr]);
    if (a.hvalid[ant * 3 + (3 + rr)] == 0) return 0u;
    const uint8_t *f = a.hflags + ((size_t)ant * 3 + (3 + rr)) * PB_BLK_PER_FFT;
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < PB_BLK_PER_FFT; ++r) m |= (f[r] ? 1u : 0u) << r;
    return __builtin_amdgcn_readfirstlane(m);
}

struct RowRegs {
    unsigned w[PFB_NW];      // word q: pairs r = 2q (bytes 0, 1) and r = 2q + 1 (bytes 2, 3), bytes = code - 128 (signed)
};

// Row rr (global row index of the batch; rr < 0: history) of (ant, pol) -> this thread's 25 sample pairs.
// The row goes through LDS with 16-byte loads of the aligned chunks that cover it (narrow per-lane global loads are
// bound by the address unit): row_request asks for the chunks (782 or 783: tid, tid + 256, tid + 512 always exist),
// row_place puts them in LDS and takes the thread's pairs; two barriers: bytes staged / bytes taken (lds is the FFT
// buffer).  Request and placement are separate so that the next row's bytes travel while this row is transformed.
struct RowStage {
    uint4 t0, t1, t2, t3;
    unsigned o;       // offset of the row's first byte in its first chunk
    bool fix;         // code 0 -> 128 on the way
};

__device__ __forceinline__ void row_request(const PfbArgs &a, int tid, int ant, int pol, int rr, RowStage &st)
{
    const uint8_t *base;
    size_t rbyte;
    if (rr >= 0) {
        const int sj = rr / a.R, rj = rr - sj * a.R;
        rbyte = (size_t)ant * a.in_ant_stride + ((size_t)sj * 2 + pol) * a.seg_samples + (size_t)rj * PB_NFFT;
        base = a.in;
        // code 0 ("no sample") is code 128 = 0.0: the kurtosis kernel has patched the input buffer; RFI mode 0 has
        // no kurtosis pass
        st.fix = a.rfi_mode == 0;
    } else {
        rbyte = (((size_t)ant * 2 + pol) * 3 + (3 + rr)) * PFB_HIST_STRIDE;
        base = a.hist;
        st.fix = true;           // history rows: zero-filled before the stream starts
    }
    st.o = (unsigned)(rbyte & 15);
    const uint4 *src16 = (const uint4 *)(base + (rbyte - st.o));
    const int nch = (int)((st.o + PB_NFFT + 15) >> 4);      // <= 783
    st.t0 = src16[tid];
    st.t1 = src16[tid + 256];
    st.t2 = src16[tid + 512];
    st.t3 = make_uint4(0u, 0u, 0u, 0u);
    if (tid + 768 < nch) st.t3 = src16[tid + 768];
}

__device__ __forceinline__ void row_place(const RowStage &st, uint8_t *lds, int tid, RowRegs &out)
{
    uint4 *dst = (uint4 *)lds;
    const unsigned o = st.o;
    const bool last = tid + 768 < (int)((o + PB_NFFT + 15) >> 4);
    if (st.fix) {
        dst[tid] = fix_zero_codes(st.t0);
        dst[tid + 256] = fix_zero_codes(st.t1);
        dst[tid + 512] = fix_zero_codes(st.t2);
        if (last) dst[tid + 768] = fix_zero_codes(st.t3);
    } else {
        dst[tid] = st.t0;
        dst[tid + 256] = st.t1;
        dst[tid + 512] = st.t2;
        if (last) dst[tid + 768] = st.t3;
    }
    __syncthreads();
    if (tid < 250) {
        const uint16_t *s = (const uint16_t *)(lds + o);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const unsigned lo = s[tid + 250 * (2 * q)], hi = s[tid + 250 * (2 * q + 1)];
            out.w[q] = (lo | (hi << 16)) ^ 0x80808080u;
        }
        out.w[12] = (unsigned)s[tid + 250 * 24] ^ 0x00008080u;
    }
    __syncthreads();
}

__device__ __forceinline__ f2 pair_of(unsigned w, int half)
{
    // (float)(signed char): one v_cvt_f32_i32_sdwa sext(...) src0_sel:BYTE_k each
    return half == 0 ? mk2((float)(int)(signed char)(w), (float)(int)(signed char)(w >> 8))
                     : mk2((float)(int)(signed char)(w >> 16), (float)((int)w >> 24));
}

// The window: v[r] = FIR of sample pair n = tid + 250 r.  KUR: blocks flagged in mask[j] are skipped for row j.
template <bool KUR>
__device__ __forceinline__ void pfb_fir(const PfbArgs &a, int tid, const RowRegs &r0, const RowRegs &r1, const RowRegs &r2,
                                        const RowRegs &r3, const unsigned (&mask)[4], f2 (&v)[25])
{
    // window coefficients of samples (2n, 2n+1), n = tid + 250 r, tap j: through a buffer descriptor with the
    // lane part (32 tid) in the vector offset and r in the scalar offset.  The four taps of sample pair n are 32
    // contiguous bytes and consecutive lanes take consecutive n: two coalesced 16-byte loads per block r.
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)a.fir, 0, 6250 * 4 * 8, 0x00020000);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const unsigned any = KUR ? (mask[0] | mask[1] | mask[2] | mask[3]) : 0u;
#pragma unroll
    for (int r = 0; r < 25; ++r) {
        const int q = r >> 1, h = r & 1;
        const f4 c01 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsF, tid * 32, 250 * r * 32, 0));
        const f4 c23 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsF, tid * 32 + 16, 250 * r * 32, 0));
        const f2 x0 = pair_of(r0.w[q], h), x1 = pair_of(r1.w[q], h), x2 = pair_of(r2.w[q], h), x3 = pair_of(r3.w[q], h);
        const f2 t0 = mk2(c01.x, c01.y), t1 = mk2(c01.z, c01.w), t2 = mk2(c23.x, c23.y), t3 = mk2(c23.z, c23.w);
        f2 acc;
        if (KUR && ((any >> r) & 1u)) {
            // block r is flagged in at least one contributing row (wave-uniform, rare): that row's term is a zero
            acc = mk2(0.f, 0.f);
            if (!((mask[0] >> r) & 1u)) acc = pkfma(t0, x0, acc);
            if (!((mask[1] >> r) & 1u)) acc = pkfma(t1, x1, acc);
            if (!((mask[2] >> r) & 1u)) acc = pkfma(t2, x2, acc);
            if (!((mask[3] >> r) & 1u)) acc = pkfma(t3, x3, acc);
        } else {
            acc = t0 * x0;
            acc = pkfma(t1, x1, acc);
            acc = pkfma(t2, x2, acc);
            acc = pkfma(t3, x3, acc);
        }
        v[r] = acc;
        // five blocks at a time: letting the scheduler hoist all 50 coefficient loads spills
        if (r % 5 == 4) __builtin_amdgcn_sched_barrier(0);
    }
}

// FFT of the windowed row in v and the spectrum step: ROLE 0 = raw spectrum (also fills the excised plane when
// none of the four contributing rows has a flagged block), ROLE 1 = excised spectrum.
template <int ROLE>
__device__ __forceinline__ void pfb_spectrum(const PfbArgs &a, f2 (&v)[25], f2 *buf, int tid, int seg, int row,
                                             unsigned differ, float w, size_t prow)
{
#if !PFB_TQ || defined(FFT_LEAN)
    fft6250(v, buf, (const f2 *)a.tw2, (const f2 *)a.tw3, tid);
#else
    // the spectrum step's twiddles are requested from inside pass 3 (as in k_channelize.hip)
    float4 tq[4][2];
    auto load_tq = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tq[i][0] = *(const float4 *)(a.postc + tid * 4 + 1024 * i);
            tq[i][1] = *(const float4 *)(a.postc + tid * 4 + 1024 * i + 2);
        }
    };
    fft6250(v, buf, (const f2 *)a.tw2, (const f2 *)a.tw3, tid, load_tq);
#endif

    const bool inject = a.frb.delays != nullptr && a.inject_now > 0;
    const int since = inject ? (a.inject_now - 1 + seg) * a.R : 0;
    const bool also_kur = a.rfi_mode == 2 && ROLE == 0 && differ == 0;
    float *P0 = (ROLE == 1 ? a.Pkur : a.Praw) + prow;
    float *P1 = a.Pkur + prow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = tid * 4 + 1024 * i;
        // (the slice's twiddles are requested here, not ahead of the transform: registers are what this kernel is
        // short of -- the thread's history bytes live through the FFT)
#if !PFB_TQ || defined(FFT_LEAN)
        const float4 t01 = *(const float4 *)(a.postc + c4), t23 = *(const float4 *)(a.postc + c4 + 2);
#else
        const float4 t01 = tq[i][0], t23 = tq[i][1];
#endif
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
        if (ROLE == 0) *(float4 *)(P0 + c4) = make_float4(pw[0], pw[1], pw[2], pw[3]);
        if (ROLE == 1 || (also_kur && w != 1.0f))
            *(float4 *)((ROLE == 1 ? P0 : P1) + c4) = make_float4(pw[0] / w, pw[1] / w, pw[2] / w, pw[3] / w);
        else if (also_kur)      // x / 1 = x: no division for a row whose window is complete and unflagged
            *(float4 *)(P1 + c4) = make_float4(pw[0], pw[1], pw[2], pw[3]);
    }
}

// One workgroup = PFB_RUN consecutive output rows of one (segment, pol); both transforms of a row (the excised one
// only when some contributing block is flagged: 43 % of rows on clean noise) from the same registers.
__global__ __launch_bounds__(256, PFB_WGS) void k_channelize_pfb(PfbArgs a)
{
    __shared__ __attribute__((aligned(16))) f2 buf[M_HALF];      // 50 000 B: row staging, then the FFT buffer
    uint8_t *lds = (uint8_t *)buf;
    int tid = threadIdx.x;
    // grid (R / PFB_RUN, nseg * 2, A).  Workgroups go to the 8 XCDs in turn and neighbouring runs share three
    // input rows: every XCD takes a contiguous eighth of the segment's runs when they divide evenly.
    const int nruns = a.R / PFB_RUN;
    const int run = (nruns & 7) == 0 ? (int)(blockIdx.x & 7) * (nruns >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int row0 = run * PFB_RUN;
    const int seg = blockIdx.y >> 1, pol = blockIdx.y & 1, ant = blockIdx.z;
    const int grow0 = seg * a.R + row0;

    // the three rows before the run (of this segment, the one before, or the previous batch's last rows)
    RowRegs h0, h1, h2, cur;
    RowStage st;
    {
        // all four requests first: one memory latency for the run's head, not four
        RowStage s0, s1, s2;
        row_request(a, tid, ant, pol, grow0 - 3, s0);
        row_request(a, tid, ant, pol, grow0 - 2, s1);
        row_request(a, tid, ant, pol, grow0 - 1, s2);
        row_request(a, tid, ant, pol, grow0, st);
        row_place(s0, lds, tid, h0);
        row_place(s1, lds, tid, h1);
        row_place(s2, lds, tid, h2);
    }
    unsigned mask[4] = {0u, 0u, 0u, 0u};
    if (a.rfi_mode) {
        mask[1] = row_mask(a, ant, grow0 - 3);
        mask[2] = row_mask(a, ant, grow0 - 2);
        mask[3] = row_mask(a, ant, grow0 - 1);
    }
#pragma unroll 1
    for (int i = 0; i < PFB_RUN; ++i) {
        const int row = row0 + i, grow = grow0 + i;
        row_place(st, lds, tid, cur);        // (its second barrier: the staged bytes are in registers)
        if (i + 1 < PFB_RUN) row_request(a, tid, ant, pol, grow + 1, st);   // travels under this row's transforms
        mask[0] = mask[1];
        mask[1] = mask[2];
        mask[2] = mask[3];
        mask[3] = a.rfi_mode ? row_mask(a, ant, grow) : 0u;
        const unsigned differ = mask[0] | mask[1] | mask[2] | mask[3];
        const float w = a.rfi_mode ? a.wrow[(size_t)ant * a.wrow_ant_stride + grow] : 1.f;
        const size_t prow = (size_t)ant * a.p_ant_stride + (((size_t)seg * 2 + pol) * a.R + row) * PB_NCHANOUT;
        f2 v[25];
        if (a.rfi_mode != 1) {
            if (tid < 250) pfb_fir<false>(a, tid, h0, h1, h2, cur, mask, v);
            pfb_spectrum<0>(a, v, buf, tid, seg, row, differ, w, prow);
        }
        const bool second = a.rfi_mode == 1 || (a.rfi_mode == 2 && differ != 0);
        if (second) {
            if (w == 0.f) {
                for (int c = tid; c < PB_NCHANOUT; c += 256) a.Pkur[prow + c] = __builtin_inff();
            } else {
                if (a.rfi_mode == 2) {
                    __syncthreads();   // the raw pass has finished reading the FFT buffer
                    asm volatile("" : "+v"(tid));   // no sharing of tid-derived addresses across the two passes
                }
                if (tid < 250) pfb_fir<true>(a, tid, h0, h1, h2, cur, mask, v);
                pfb_spectrum<1>(a, v, buf, tid, seg, row, differ, w, prow);
            }
        }
        h0 = h1;
        h1 = h2;
        h2 = cur;
        __syncthreads();               // the spectrum step has finished reading buf: the next row may be staged
    }
}

// row weights of the PFB mode (see the header comment); overwrites wrow[g]
__global__ void k_pfb_weights(const uint32_t *__restrict__ rowmask, size_t wrow_ant_stride,
                              const uint8_t *__restrict__ hflags, const float *__restrict__ tapE,
                              float *__restrict__ wrow, int nrows)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int ant = blockIdx.y;
    if (g >= nrows) return;
    // flag masks of the four contributing rows: the kurtosis kernel's mask words (one load each; reading the 100
    // flag bytes one after the other made this 40-workgroup kernel take 57 us on the critical path), the
    // history slots' bytes for the first three rows of a batch
    unsigned m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = g - 3 + j;
        if (rr >= 0) {
            m[j] = rowmask[(size_t)ant * wrow_ant_stride + rr];
        } else {
            const uint8_t *f = hflags + ((size_t)ant * 3 + (3 + rr)) * PB_BLK_PER_FFT;
            unsigned mm = 0;
            for (int b = 0; b < PB_BLK_PER_FFT; ++b) mm |= (f[b] ? 1u : 0u) << b;
            m[j] = mm;
        }
    }
    // sum of the unflagged (tap, block) energies, taps then blocks ascending (tapE[100] is that sum with no flag
    // at all, accumulated in the same order: an unflagged row gets exactly 1)
    float s = tapE[100];
    if (m[0] | m[1] | m[2] | m[3]) {
        s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int b = 0; b < PB_BLK_PER_FFT; ++b)
                if (!((m[j] >> b) & 1u)) s = s + tapE[j * PB_BLK_PER_FFT + b];
    }
    wrow[(size_t)ant * wrow_ant_stride + g] = s / tapE[100];
}

// keep the last three rows (and flags) of the batch for the next call
__global__ void k_pfb_history(const uint8_t *__restrict__ in, size_t in_ant_stride, size_t seg_samples,
                              const uint8_t *__restrict__ flags, size_t flags_ant_stride,
                              uint8_t *__restrict__ hist, uint8_t *__restrict__ hflags,
                              uint8_t *__restrict__ hvalid, int R, int nrows)
{
    const int j = blockIdx.x;          // history slot 0..2 <- row nrows-3+j
    const int pol = blockIdx.y, ant = blockIdx.z;
    const int rr = nrows - 3 + j;
    const uint8_t *src = in + (size_t)ant * in_ant_stride + ((size_t)(rr / R) * 2 + pol) * seg_samples +
                         (size_t)(rr % R) * PB_NFFT;
    uint8_t *dst = hist + (((size_t)ant * 2 + pol) * 3 + j) * PFB_HIST_STRIDE;
    for (int i = threadIdx.x; i < PB_NFFT / 4; i += blockDim.x) ((uint32_t *)dst)[i] = ((const uint32_t *)src)[i];
    if (pol == 0 && threadIdx.x < PB_BLK_PER_FFT)
        hflags[((size_t)ant * 3 + j) * PB_BLK_PER_FFT + threadIdx.x] =
            flags[(size_t)ant * flags_ant_stride + (size_t)rr * PB_BLK_PER_FFT + threadIdx.x];
    if (pol == 0 && threadIdx.x == 0) hvalid[ant * 3 + j] = 1;
}

// row weights of the batch, from the kurtosis flags: queued right behind the kurtosis pass, on its stream
hipError_t launch_pfb_weights(pb_handle *h, int nseg)
{
    if (h->cfg.taps != 4 || !h->cfg.rfi_mode) return hipSuccess;
    const int nrows = nseg * h->R;
    dim3 g((nrows + 255) / 256, h->A);
    k_pfb_weights<<<g, 256, 0, h->stream>>>(pb_rowmask(h), (size_t)h->S * h->R, h->d_hist_flags, h->d_tapE, h->d_wrow, nrows);
    return hipGetLastError();
}

hipError_t launch_channelize_pfb(pb_handle *h, int nseg, int inject_now)
{
    if (h->R % PFB_RUN) return hipErrorInvalidValue;       // (pb_create: rows_per_seg is a multiple of 8)
    PfbArgs a;
    a.in = h->d_in;
    a.in_ant_stride = (size_t)h->S * 2 * h->seg_samples;
    a.seg_samples = h->seg_samples;
    a.hist = h->d_hist_in;
    a.hflags = h->d_hist_flags;
    a.hvalid = h->d_hist_valid;
    a.wrow = h->d_wrow;
    a.rowmask = pb_rowmask(h);
    a.wrow_ant_stride = (size_t)h->S * h->R;
    a.fir = h->ft.taps_n;
    a.Praw = h->d_Praw;
    a.Pkur = h->d_Pkur;
    a.p_ant_stride = (size_t)h->S * 2 * h->R * PB_NCHANOUT;
    a.tw2 = h->ft.tw2;
    a.tw3 = h->ft.tw3;
    a.postc = h->ft.postc;
    a.frb.delays = (inject_now > 0) ? h->d_frb_delays : nullptr;
    a.frb.width = h->frb_width;
    a.frb.amp = h->frb_amp;
    a.frb.since = 0;
    a.R = h->R;
    a.rfi_mode = h->cfg.rfi_mode;
    a.inject_now = inject_now;
    dim3 grid((unsigned)(h->R / PFB_RUN), (unsigned)(nseg * 2), (unsigned)h->A);
    k_channelize_pfb<<<grid, 256, 0, h->stream>>>(a);
    return hipGetLastError();
}

// keep the batch's last three rows and their flags for the next call (queued behind the channeliser, after the
// event that releases detect: it is not on the path to the output)
hipError_t launch_pfb_history(pb_handle *h, int nseg)
{
    const int nrows = nseg * h->R;
    dim3 gh(3, 2, h->A);
    k_pfb_history<<<gh, 256, 0, h->stream>>>(h->d_in, (size_t)h->S * 2 * h->seg_samples, h->seg_samples, h->d_flags,
                                             (size_t)h->S * h->nblk_seg, h->d_hist_in, h->d_hist_flags, h->d_hist_valid,
                                             h->R, nrows);
    return hipGetLastError();
}
