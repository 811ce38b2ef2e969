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
// Parity: spectra vs polyphase_filterbank golden (tests, 2e-6), whole chain vs the oracle's kernels
// composed around the same fp32 FIR (bit-exact, RFI mode 0); the weight definition has no reference.
// ordinary (L2-allocating) stores of the power planes here: with non-temporal ones this kernel is 6 % faster alone
// (0.68 against 0.73 ms per launch) and the pipeline 3 % slower (1.02 against 0.995 ms per step: detect's reads of
// the planes then come from further away), and the step is what counts (profiles/r03_notes.md)
#ifndef PB_NT_STORES
#define PB_NT_STORES 0
#endif
#if PB_NT_STORES == 3       // (variant builds: the header's default)
#undef PB_NT_STORES
#endif
#include "fft_lds.h"
#include "kurtosis_dev.h"

#ifndef PFB_DBG
#define PFB_DBG 0     // timing experiments only: 1 no coefficient loads, 2 one tap only, 4 stage one row only
#endif
#ifndef PFB_ROW_LDS
#define PFB_ROW_LDS 12528    // 12500 bytes + up to 12 of alignment slack, padded to 16
#endif
#define PFB_HIST_STRIDE 12512
#ifndef PFB_STAGE_DMA
#define PFB_STAGE_DMA 1      // rows go to LDS by global_load_lds_dwordx4 (0: through registers; timing experiments)
#endif
#ifndef PFB_COEF_PRE
#define PFB_COEF_PRE 0      // blocks whose window coefficients are requested before the rows
#endif
#ifndef PFB_WIN_GROUP
#define PFB_WIN_GROUP 5      // blocks of the window loop between scheduling barriers
#endif

struct PfbArgs {
    const uint8_t *in;       // [A][S][2][seg_samples]
    size_t in_ant_stride, seg_samples;
    const uint8_t *hist;     // [A][2][3][PFB_HIST_STRIDE]
    const uint8_t *flags;    // [A][S*R*25]
    size_t flags_ant_stride;
    const uint8_t *hflags;   // [A][3][25]
    const uint8_t *hvalid;   // [A][3] history slot holds data
    const float *wrow;       // [A][S*R]  (already the PFB weights)
    const uint32_t *rowmask; // [A][S*R]  flag masks of the rows (k_kurtosis_row), one scalar load instead of 25 bytes
    size_t wrow_ant_stride;
    const float2 *fir;       // [6250 n][4 taps] coefficient pairs of samples (2n, 2n+1) (FftTables::taps_n)
    float *Praw, *Pkur;
    size_t p_ant_stride;
    const float2 *tw2, *tw3, *postc;
    FrbParams frb;
    int R, rfi_mode, inject_now;
};

__device__ __forceinline__ unsigned row_mask(const PfbArgs &a, int ant, int rr)
{
    // flags of the 25 blocks of row rr (rr < 0: history slot 3 + rr, the first three rows of a batch only)
    if (rr >= 0) return __builtin_amdgcn_readfirstlane(a.rowmask[(size_t)ant * a.wrow_ant_stride + rr]);
    const uint8_t *f = a.hflags + ((size_t)ant * 3 + (3 + rr)) * PB_BLK_PER_FFT;
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < PB_BLK_PER_FFT; ++r) m |= (f[r] ? 1u : 0u) << r;
    return __builtin_amdgcn_readfirstlane(m);
}

// One transform of one (row, pol): ROLE 0 = raw spectrum (also fills the excised plane when none of the four
// contributing rows has a flagged block), ROLE 1 = excised spectrum.  The masks and the weight are scalar loads
// requested at the top of the kernel; ROLE 0 first needs them after the FFT, so their latency and that of the
// rows' bytes run side by side.
template <int role>
__device__ __forceinline__ void pfb_pass(const PfbArgs &a, uint8_t *lds, int tid, int grow, int seg, int row, int pol,
                                         int ant, const unsigned (&mask)[4], unsigned differ, float w, size_t prow)
{
    f2 *buf = (f2 *)lds;
    // window coefficients of samples (2n, 2n+1), n = tid + 250 r, tap j: through a buffer descriptor
    // with the lane part (8 tid) in the vector offset and (j, r) in the scalar offset
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)a.fir, 0, 6250 * 4 * 8, 0x00020000);
    typedef float f4 __attribute__((ext_vector_type(4)));
    // the four taps of sample pair n are 32 contiguous bytes, and consecutive lanes take consecutive
    // n: two coalesced 16-byte loads per block r (taps 0,1 and 2,3) instead of four 8-byte ones
    auto coef2 = [&](int jj, int r) __attribute__((always_inline)) {
        if (PFB_DBG & 1) { f4 one = {1.f, 1.f, 1.f, 1.f}; return one; }
        return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsF, tid * 32, (250 * r * 4 + jj) * 8, 0));
    };
#if PFB_COEF_PRE
    // the first blocks' coefficients are requested before the rows: they arrive while the rows are on their way
    f4 pre[2 * PFB_COEF_PRE];
    if (tid < 250) {
#pragma unroll
        for (int r = 0; r < PFB_COEF_PRE; ++r) {
            pre[2 * r] = coef2(0, r);
            pre[2 * r + 1] = coef2(2, r);
        }
    }
#endif
    // stage the four rows (16-byte loads of the aligned chunks that cover each row)
    unsigned off[4];
#pragma unroll
    for (int j = (PFB_DBG & 4) ? 3 : 0; j < 4; ++j) {
        const int rr = grow - 3 + j;
        const uint8_t *base;
        size_t rbyte;
        if (rr >= 0) {
            int sj = seg, rj = row - 3 + j;   // (rr / R, rr % R) without the divisions
            while (rj < 0) {
                rj += a.R;
                --sj;
            }
            rbyte = (size_t)ant * a.in_ant_stride + ((size_t)sj * 2 + pol) * a.seg_samples + (size_t)rj * PB_NFFT;
            base = a.in;
        } else {
            rbyte = (((size_t)ant * 2 + pol) * 3 + (3 + rr)) * PFB_HIST_STRIDE;
            base = a.hist;
        }
        const unsigned o = (unsigned)(rbyte & 15);
        off[j] = o;
        const uint4 *src16 = (const uint4 *)(base + (rbyte - o));
        const int nch = (int)((o + PB_NFFT + 15) >> 4);
        uint4 *dst = (uint4 *)(lds + j * PFB_ROW_LDS);
        // code 0 ("no sample") is code 128 = 0.0: the kurtosis kernel has patched the input buffer; history rows
        // (zero-filled before the stream starts) and RFI mode 0 (no kurtosis pass) are patched here, four bytes
        // per instruction (fft_lds.h)
        if (a.rfi_mode == 0 || rr < 0) {
            // (all four requests first: a loop over a run-time count is not unrolled, and each of its loads would
            // wait for the one before)
            const uint4 q0 = src16[tid], q1 = src16[tid + 256], q2 = src16[tid + 512];
            uint4 q3 = make_uint4(0u, 0u, 0u, 0u);
            const bool last = tid + 768 < nch;
            if (last) q3 = src16[tid + 768];
            dst[tid] = fix_zero_codes(q0);
            dst[tid + 256] = fix_zero_codes(q1);
            dst[tid + 512] = fix_zero_codes(q2);
            if (last) dst[tid + 768] = fix_zero_codes(q3);
        } else {
#if PFB_STAGE_DMA
            // straight into LDS (global_load_lds_dwordx4): no registers, no ds_write; a wave's 64 lanes fill 1 KB
            // from the wave-uniform LDS address up
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = t * 256 + tid;
                if (t < 3 || i < nch)              // a row is 782 or 783 chunks
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src16 + i),
                                                     (void __attribute__((address_space(3))) *)(dst + t * 256 + wv * 64),
                                                     16, 0, 0);
            }
#else
            for (int i = tid; i < nch; i += 256) dst[i] = src16[i];
#endif
        }
    }
#if PFB_STAGE_DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();

    f2 v[25];
    if (tid < 250) {
        constexpr bool kur = role == 1;
        const uint16_t *s0 = (const uint16_t *)(lds + 0 * PFB_ROW_LDS + off[0]);
        const uint16_t *s1 = (const uint16_t *)(lds + 1 * PFB_ROW_LDS + off[1]);
        const uint16_t *s2 = (const uint16_t *)(lds + 2 * PFB_ROW_LDS + off[2]);
        const uint16_t *s3 = (const uint16_t *)(lds + 3 * PFB_ROW_LDS + off[3]);
        const unsigned m0 = kur ? mask[0] : 0u, m1 = kur ? mask[1] : 0u, m2 = kur ? mask[2] : 0u,
                       m3 = kur ? mask[3] : 0u;
#pragma unroll
        for (int r = 0; r < 25; ++r) {
            const int n = tid + 250 * r;
            // a flagged block contributes zeros = code 128 (apply_kurtosis :243-295)
            // (bit arithmetic, not a select: the compiler turns a select on this wave-uniform condition
            // into a branch around each LDS read)
            auto pick = [&](unsigned m, unsigned raw) __attribute__((always_inline)) {
                const unsigned z = 0u - ((m >> r) & 1u);          // all ones when block r is flagged
                return (raw & ~z) | (0x8080u & z);
            };
            const unsigned w0 = pick(m0, s0[n]);
            const unsigned w1 = (PFB_DBG & 2) ? w0 : pick(m1, s1[n]), w2 = (PFB_DBG & 2) ? w0 : pick(m2, s2[n]), w3 = (PFB_DBG & 2) ? w0 : pick(m3, s3[n]);
            // sum_j taps[j] * x_j, products then left-to-right adds (the order of k_channelize_f32),
            // re and im side by side in packed instructions
#if PFB_COEF_PRE
            const f4 c01 = r < PFB_COEF_PRE ? pre[2 * (r < PFB_COEF_PRE ? r : 0)] : coef2(0, r);
            const f4 c23 = r < PFB_COEF_PRE ? pre[2 * (r < PFB_COEF_PRE ? r : 0) + 1] : coef2(2, r);
#else
            const f4 c01 = coef2(0, r), c23 = coef2(2, r);
#endif
            f2 acc = mk2(c01.x, c01.y) * cvt_pair_c(w0);
            acc = acc + mk2(c01.z, c01.w) * cvt_pair_c(w1);
            acc = acc + mk2(c23.x, c23.y) * cvt_pair_c(w2);
            acc = acc + mk2(c23.z, c23.w) * cvt_pair_c(w3);
            v[r] = acc;
            // five blocks at a time: letting the scheduler hoist all 100 coefficient loads spills
            if (r % PFB_WIN_GROUP == PFB_WIN_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
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

    const bool inject = a.frb.delays != nullptr && a.inject_now > 0;
    const int since = inject ? (a.inject_now - 1 + seg) * a.R : 0;
    const bool also_kur = a.rfi_mode == 2 && role == 0 && differ == 0;
    float *P0 = (role == 1 ? a.Pkur : a.Praw) + prow;
    float *P1 = a.Pkur + prow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
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
        if (role == 0) store_plane4(P0, c4, pw[0], pw[1], pw[2], pw[3]);
        if (role == 1 || (also_kur && w != 1.0f))
            store_plane4(role == 1 ? P0 : P1, c4, pw[0] / w, pw[1] / w, pw[2] / w, pw[3] / w);
        else if (also_kur)      // x / 1 = x: no division for a row whose window is complete and unflagged
            store_plane4(P1, c4, pw[0], pw[1], pw[2], pw[3]);
    }
}

// One workgroup per (row, pol) does both transforms (the second only when some contributing block is flagged: 43 %
// of rows on clean noise), as k_channelize does: a second grid of workgroups for the excised spectra spent a load
// latency each on finding out that 57 % of them had nothing to do.
__global__ __launch_bounds__(256, 3) void k_channelize_pfb(PfbArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * PFB_ROW_LDS];   // 50 112 B, reused as the FFT buffer
    int tid = threadIdx.x;
    // grid (R, nseg * 2, A): no division to find the row.  Workgroups go to the 8 XCDs in turn, and an input row
    // is read by the workgroups of four consecutive output rows: give every XCD a contiguous eighth of the
    // segment's rows, so that three of those four reads hit its own L2 (R is a multiple of 8).
#ifndef PFB_NO_XCD_MAP
    const int row = (int)(blockIdx.x & 7) * (a.R >> 3) + (int)(blockIdx.x >> 3);
#else
    const int row = blockIdx.x;
#endif
    const int seg = blockIdx.y >> 1, pol = blockIdx.y & 1, ant = blockIdx.z;
    const int grow = seg * a.R + row;

    // zeroing masks of the four contributing rows.  A history slot that holds no data yet (start of
    // the stream) is all zeros: nothing to excise there (its missing weight is booked by k_pfb_weights).
    unsigned mask[4] = {0, 0, 0, 0};
    unsigned differ = 0;
    const float w = a.rfi_mode ? a.wrow[(size_t)ant * a.wrow_ant_stride + grow] : 1.f;
    if (a.rfi_mode) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = grow - 3 + j;
            const bool present = rr >= 0 || a.hvalid[ant * 3 + (3 + rr)] != 0;
            mask[j] = present ? row_mask(a, ant, rr) : 0u;
            differ |= mask[j];
        }
    }
    const size_t prow = (size_t)ant * a.p_ant_stride + (((size_t)seg * 2 + pol) * a.R + row) * PB_NCHANOUT;
    if (a.rfi_mode != 1) pfb_pass<0>(a, lds, tid, grow, seg, row, pol, ant, mask, differ, w, prow);
    if (a.rfi_mode == 0 || (a.rfi_mode == 2 && differ == 0)) return;
    if (w == 0.f) {
        for (int c = tid; c < PB_NCHANOUT; c += 256) a.Pkur[prow + c] = __builtin_inff();
        return;
    }
    if (a.rfi_mode == 2) {
        __syncthreads();   // the raw pass has finished reading the FFT buffer
        asm volatile("" : "+v"(tid));   // no sharing of tid-derived addresses across the two passes
    }
    pfb_pass<1>(a, lds, tid, grow, seg, row, pol, ant, mask, differ, w, prow);
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
    k_pfb_weights<<<g, 256, 0, h->stream>>>(pb_rowmask(h), (size_t)h->S * h->R,
                                            h->d_hist_flags + (size_t)h->hist_rd * h->A * 3 * PB_BLK_PER_FFT, h->d_tapE, h->d_wrow, nrows);
    return hipGetLastError();
}

hipError_t launch_channelize_pfb(pb_handle *h, int nseg, int inject_now)
{
    PfbArgs a;
    a.in = h->d_in;
    a.in_ant_stride = (size_t)h->S * 2 * h->seg_samples;
    a.seg_samples = h->seg_samples;
    a.hist = h->d_hist_in + (size_t)h->hist_rd * h->A * 2 * 3 * PFB_HIST_STRIDE;      // the slot this batch reads
    a.flags = h->d_flags;
    a.flags_ant_stride = (size_t)h->S * h->nblk_seg;
    a.hflags = h->d_hist_flags + (size_t)h->hist_rd * h->A * 3 * PB_BLK_PER_FFT;
    a.hvalid = h->d_hist_valid + (size_t)h->hist_rd * h->A * 3;
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
    dim3 grid((unsigned)h->R, (unsigned)(nseg * 2), (unsigned)h->A);
    k_channelize_pfb<<<grid, 256, 0, h->stream>>>(a);
    return hipGetLastError();
}

// keep the batch's last three rows and their flags for the next call, in the history slot this batch does NOT read
// (queued behind the weights on the kurtosis stream, behind the previous batch's channeliser, which read that slot)
hipError_t launch_pfb_history(pb_handle *h, int nseg)
{
    const int nrows = nseg * h->R;
    const int wr = h->hist_rd ^ 1;       // the slot the NEXT batch will read
    dim3 gh(3, 2, h->A);
    k_pfb_history<<<gh, 256, 0, h->stream>>>(h->d_in, (size_t)h->S * 2 * h->seg_samples, h->seg_samples, h->d_flags,
                                             (size_t)h->S * h->nblk_seg,
                                             h->d_hist_in + (size_t)wr * h->A * 2 * 3 * PFB_HIST_STRIDE,
                                             h->d_hist_flags + (size_t)wr * h->A * 3 * PB_BLK_PER_FFT,
                                             h->d_hist_valid + (size_t)wr * h->A * 3, h->R, nrows);
    return hipGetLastError();
}
