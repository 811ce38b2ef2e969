// LDS FFT building blocks shared by the channeliser kernels: radix-5/25/10 butterflies with
// compile-time twiddles, the 6250-point in-place Stockham FFT and the real-input split.  The
// operation order is the specification in oracle/pb_oracle.c "K6" (bit-exact).
#pragma once
#include "fft_consts.h"
#include "pb_internal.h"

#define M_HALF 6250

#ifdef FFT_LEAN
#define FFT_PREFETCH 0
#endif
#ifndef FFT_PREFETCH
#define FFT_PREFETCH 5   // bit 0: pass-2 twiddles requested before pass 1; pass-3 twiddles requested before the
                         // pass-2 arithmetic (bit 1) or between that arithmetic and its LDS stores (bit 2)
#endif

#ifndef FFT_ABL
#define FFT_ABL 0       // energy experiments (variant builds only, results invalid): 1 no butterfly arithmetic, 2 no LDS exchange
#endif
#ifndef FFT_STAMP
#define FFT_STAMP(i)   // timing experiments: a variant build records the wave's clock at phase boundary i
#endif

namespace {
constexpr float kW25[25][2] = FC_W25_INIT;
constexpr float kW10[5][2] = FC_W10_INIT;
}

// Complex values live in aligned VGPR pairs (re, im) so that the butterflies are packed-f32
// instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per lane per issue,
// each rounding exactly like its scalar form).  Swaps and sign flips of halves fold into the
// instructions' op_sel / neg modifiers.  The arithmetic is element for element the oracle's.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// The shuffled forms below are written as instructions because the compiler spends two extra moves
// on each instead of using the source-half selectors:
//   op_sel[i] / op_sel_hi[i]: which half of source i feeds the low / high result (0 = low half)
//   neg_lo[i] / neg_hi[i]:    negate source i in the low / high result
// Independent operations share one asm block so that no instruction consumes its predecessor's
// result (the compiler pads a dependent pair of asm statements with s_nop, 4 cycles each).

// y_minus = m - i n = m + (n.y, -n.x),  y_plus = m + i n = m - (n.y, -n.x), for two (m, n) pairs
__device__ __forceinline__ void rot4(f2 m1, f2 n1, f2 m2, f2 n2, f2 &y1, f2 &y4, f2 &y2, f2 &y3)
{
    asm("v_pk_add_f32 %0, %4, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %4, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %2, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=&v"(y1), "=&v"(y4), "=&v"(y2), "=&v"(y3)
        : "v"(m1), "v"(n1), "v"(m2), "v"(n2));
}

// E = a + conj(b), O = a - conj(b)
__device__ __forceinline__ void addsub_conj(f2 a, f2 b, f2 &E, f2 &O)
{
    asm("v_pk_add_f32 %0, %2, %3 neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %2, %3 neg_lo:[0,1]"
        : "=&v"(E), "=&v"(O)
        : "v"(a), "v"(b));
}

// a *= w, complex: (a.x w.x - a.y w.y, a.x w.y + a.y w.x) evaluated as
// t = a.y * (w.y, w.x);  a = fma(a.x, (w.x, w.y), (-t.x, t.y))      [oracle cmul]
#define PB_CMUL_MUL(t, a, w) "v_pk_mul_f32 " t ", " a ", " w " op_sel:[1,1] op_sel_hi:[1,0]\n\t"
#define PB_CMUL_FMA(t, a, w) "v_pk_fma_f32 " a ", " a ", " w ", " t " op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
#define PB_CMUL4_BODY                                                                               \
    PB_CMUL_MUL("%4", "%0", "%8") PB_CMUL_MUL("%5", "%1", "%9") PB_CMUL_MUL("%6", "%2", "%10")      \
    PB_CMUL_MUL("%7", "%3", "%11") PB_CMUL_FMA("%4", "%0", "%8") PB_CMUL_FMA("%5", "%1", "%9")      \
    PB_CMUL_FMA("%6", "%2", "%10") PB_CMUL_FMA("%7", "%3", "%11")
// four at a time, twiddles in vector registers
__device__ __forceinline__ void cmul4(f2 &a0, f2 &a1, f2 &a2, f2 &a3, f2 w0, f2 w1, f2 w2, f2 w3)
{
    f2 t0, t1, t2, t3;
    asm(PB_CMUL4_BODY
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}
// four at a time, compile-time twiddles in scalar register pairs
__device__ __forceinline__ void cmul4_k(f2 &a0, f2 &a1, f2 &a2, f2 &a3, f2 w0, f2 w1, f2 w2, f2 w3)
{
    f2 t0, t1, t2, t3;
    asm(PB_CMUL4_BODY
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "s"(w0), "s"(w1), "s"(w2), "s"(w3));
}
// one, twiddle in vector registers (the compiler schedules other work between the two halves)
__device__ __forceinline__ f2 cmul(f2 a, f2 w)
{
    f2 t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(t) : "v"(a), "v"(w), "0"(t));
    return t;
}

__device__ __forceinline__ void dft5(f2 v0, f2 v1, f2 v2, f2 v3, f2 v4, f2 &y0, f2 &y1, f2 &y2, f2 &y3, f2 &y4)
{
    constexpr float C1 = FC_C1, C2 = FC_C2, S1 = FC_S1, S2 = FC_S2;
    const f2 t1 = v1 + v4, t2 = v2 + v3, t3 = v1 - v4, t4 = v2 - v3;
    y0 = (v0 + t1) + t2;
    const f2 m1 = pkfma(mk2(C2, C2), t2, pkfma(mk2(C1, C1), t1, v0));
    const f2 m2 = pkfma(mk2(C1, C1), t2, pkfma(mk2(C2, C2), t1, v0));
    const f2 n1 = pkfma(mk2(S2, S2), t4, mk2(S1, S1) * t3);
    const f2 n2 = pkfma(mk2(-S1, -S1), t4, mk2(S2, S2) * t3);
    // y1 = m1 - i n1, y4 = m1 + i n1, y2 = m2 - i n2, y3 = m2 + i n2
    rot4(m1, n1, m2, n2, y1, y4, y2, y3);
}

__device__ __forceinline__ void dft25(f2 (&v)[25])
{
    // stage 1 in place: A[n2][k1] lives in v[5*k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        f2 a0, a1, a2, a3, a4;
        dft5(v[n2], v[5 + n2], v[10 + n2], v[15 + n2], v[20 + n2], a0, a1, a2, a3, a4);
        if (n2)
            cmul4_k(a1, a2, a3, a4, mk2(kW25[n2 * 5 + 1][0], kW25[n2 * 5 + 1][1]),
                    mk2(kW25[n2 * 5 + 2][0], kW25[n2 * 5 + 2][1]), mk2(kW25[n2 * 5 + 3][0], kW25[n2 * 5 + 3][1]),
                    mk2(kW25[n2 * 5 + 4][0], kW25[n2 * 5 + 4][1]));
        v[n2] = a0;
        v[5 + n2] = a1;
        v[10 + n2] = a2;
        v[15 + n2] = a3;
        v[20 + n2] = a4;
    }
    // stage 2: for each k1 a DFT5 over n2; output k1 + 5 k2
    f2 o[25];
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1)
        dft5(v[5 * k1], v[5 * k1 + 1], v[5 * k1 + 2], v[5 * k1 + 3], v[5 * k1 + 4], o[k1], o[k1 + 5], o[k1 + 10],
             o[k1 + 15], o[k1 + 20]);
#pragma unroll
    for (int i = 0; i < 25; ++i) v[i] = o[i];
}

__device__ __forceinline__ void dft10(f2 (&v)[10])
{
    f2 A0[5], A1[5];
    dft5(v[0], v[2], v[4], v[6], v[8], A0[0], A0[1], A0[2], A0[3], A0[4]);
    dft5(v[1], v[3], v[5], v[7], v[9], A1[0], A1[1], A1[2], A1[3], A1[4]);
    cmul4_k(A1[1], A1[2], A1[3], A1[4], mk2(kW10[1][0], kW10[1][1]), mk2(kW10[2][0], kW10[2][1]),
            mk2(kW10[3][0], kW10[3][1]), mk2(kW10[4][0], kW10[4][1]));
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
        v[k1] = A0[k1] + A1[k1];
        v[k1 + 5] = A0[k1] - A1[k1];
    }
}

// Complex FFT of length 6250 of the sequence whose pass-1 butterfly inputs are already in
// v (thread tid < 250 holds z[tid + 250 r], r = 0..24).  Result Z[0..6249] in buf (natural order).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

template <class Hook = NoHook>
__device__ __forceinline__ void fft6250(f2 (&v)[25], f2 *buf, const f2 *__restrict__ tw2,
                                        const f2 *__restrict__ tw3, int tid, Hook in_pass3 = Hook())
{
    // Twiddles of the next pass are requested BEFORE the barriers that precede their use, so
    // that their L2 latency hides under this pass's arithmetic and LDS traffic.
    const int k = tid % 25;
    f2 t2[24];
    f2 t3[3][9];
    // twiddle tables through buffer descriptors: the per-r row offset goes in the scalar offset, so
    // the loads need no 64-bit vector address arithmetic
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)tw2, 0, 25 * 24 * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc((void *)tw3, 0, 625 * 10 * 8, 0x00020000);
    auto load_t2 = [&]() {
        if (tid < 250) {
#pragma unroll
            // table layout [pair (r, r+1) = 12][k = 25] of 16 bytes (round 5; it was [k][r]): two twiddles per 16-byte
            // load, and the 25 distinct k of a wave's lanes read 400 contiguous bytes = 4 cache lines per instruction
            // instead of 25 (one 192-byte row per k).  The L1 works through a wave's load line by line: with three
            // workgroups per CU the twiddle loads kept it busy for more than half of the time (profiles/r05_notes.md)
            for (int r = 1; r < 25; r += 2) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs2, k * 16, ((r - 1) / 2) * (25 * 16), 0));
                t2[r - 1] = mk2(q.x, q.y);
                t2[r] = mk2(q.z, q.w);
            }
        }
    };
    auto load_t3 = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = tid + 256 * i;
            if (j < 625) {
#pragma unroll
                // table layout [j][r = 1..9, one pad]: 80 contiguous bytes per butterfly, five 16-byte loads
                for (int r = 1; r < 10; r += 2) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs3, j * 80, (r - 1) * 8, 0));
                    t3[i][r - 1] = mk2(q.x, q.y);
                    if (r < 9) t3[i][r] = mk2(q.z, q.w);
                }
            }
        }
    };
#if FFT_PREFETCH & 1
    load_t2();
#endif
    // pass 1: R = 25, Ns = 1
    if (tid < 250) {
        if (!(FFT_ABL & 1)) dft25(v);
        if (!(FFT_ABL & 2)) {
#pragma unroll
            for (int r = 0; r < 25; ++r) buf[tid * 25 + r] = v[r];
        }
    }
    __syncthreads();
    FFT_STAMP(3);
    // pass 2: R = 25, Ns = 25
    if (tid < 250 && !(FFT_ABL & 2)) {
#pragma unroll
        for (int r = 0; r < 25; ++r) v[r] = buf[tid + 250 * r];
    }
    __syncthreads();
    FFT_STAMP(4);
#if !(FFT_PREFETCH & 1)
    load_t2();
#endif
#if FFT_PREFETCH & 2
    load_t3();
#endif
    if (tid < 250 && !(FFT_ABL & 1)) {
#pragma unroll
        for (int r = 1; r < 25; r += 4)
            cmul4(v[r], v[r + 1], v[r + 2], v[r + 3], t2[r - 1], t2[r], t2[r + 1], t2[r + 2]);
        dft25(v);
    }
    if (FFT_ABL & 1) { v[0] = v[0] + t2[0] + t2[23]; }
#if FFT_PREFETCH & 4
    load_t3();   // after the pass-2 arithmetic (the register peak), before its LDS stores and the barrier
#endif
    if (tid < 250 && !(FFT_ABL & 2)) {
        const int j0 = (tid / 25) * 625 + k;
#pragma unroll
        for (int r = 0; r < 25; ++r) buf[j0 + 25 * r] = v[r];
    }
    __syncthreads();
    FFT_STAMP(5);
#ifdef FFT_LEAN
    // pass 3, register-lean form (timing experiments: a 128-VGPR channeliser): one butterfly at a time, its
    // twiddles requested just before it
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = tid + 256 * i;
        if (j < 625) {
            f2 u[10];
            f2 t[9];
#pragma unroll
            for (int r = 1; r < 10; r += 2) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs3, j * 80, (r - 1) * 8, 0));
                t[r - 1] = mk2(q.x, q.y);
                if (r < 9) t[r] = mk2(q.z, q.w);
            }
#pragma unroll
            for (int r = 0; r < 10; ++r) u[r] = buf[j + 625 * r];
            cmul4(u[1], u[2], u[3], u[4], t[0], t[1], t[2], t[3]);
            cmul4(u[5], u[6], u[7], u[8], t[4], t[5], t[6], t[7]);
            u[9] = cmul(u[9], t[8]);
            dft10(u);
#pragma unroll
            for (int r = 0; r < 10; ++r) buf[j + 625 * r] = u[r];
        }
    }
    in_pass3();
#else
    // pass 3: R = 10, Ns = 625; butterflies j = tid, tid + 256, tid + 512, each in place on
    // buf[j + 625 r] (no barrier between its loads and its stores)
    f2 u[3][10];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = tid + 256 * i;
        if (j < 625) {
#pragma unroll
            for (int r = 0; r < 10; ++r) u[i][r] = (FFT_ABL & 2) ? v[(r + i) % 25] : buf[j + 625 * r];
        }
    }
#if !(FFT_PREFETCH & 6) && !defined(FFT_LEAN)
    load_t3();
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = tid + 256 * i;
        if (j < 625) {
            if (!(FFT_ABL & 1)) {
                cmul4(u[i][1], u[i][2], u[i][3], u[i][4], t3[i][0], t3[i][1], t3[i][2], t3[i][3]);
                cmul4(u[i][5], u[i][6], u[i][7], u[i][8], t3[i][4], t3[i][5], t3[i][6], t3[i][7]);
                u[i][9] = cmul(u[i][9], t3[i][8]);
            } else {
                u[i][0] = u[i][0] + t3[i][0] + t3[i][8];
            }
            if (i == 0) in_pass3();
            if (!(FFT_ABL & 1)) dft10(u[i]);
            if (!(FFT_ABL & 2) || i == 0) {
#pragma unroll
                for (int r = 0; r < 10; ++r) buf[j + 625 * r] = u[i][r];
            }
        }
    }
#endif
    __syncthreads();
    FFT_STAMP(6);
}

// real-input split: X[k] = 0.5 (E + T[k] O), E = Z[k] + conj Z[M-k], O = Z[k] - conj Z[M-k]
__device__ __forceinline__ f2 rsplit(const f2 *buf, const f2 *__restrict__ post, int k)
{
    const f2 a = buf[k == M_HALF ? 0 : k];
    f2 b = buf[k == 0 ? 0 : M_HALF - k];
    b.y = -b.y;
    const f2 E = a + b, O = a - b;
    const f2 P = cmul(O, post[k]);
    return mk2(0.5f, 0.5f) * (E + P);
}

// Z[k] and Z[M - k] for k = k0 .. k0 + 3, k0 odd (the spectrum step's four channels of a thread; Z[M] = Z[0]),
// through 16-byte LDS reads where the pair is aligned -- (k0+1, k0+2), (M-k0-1, M-k0), (M-k0-3, M-k0-2): 5 reads
// instead of 8, and at the 32-byte lane stride of this step a 16-byte read meets half the bank conflicts of two
// 8-byte ones.  buf must be 16-byte aligned.
__device__ __forceinline__ void read_z_pairs(const f2 *buf, int k0, f2 (&za)[4], f2 (&zb)[4])
{
    typedef float f4v __attribute__((ext_vector_type(4)));
#ifdef FFT_ZCONF_ABL
    // timing experiment (variant builds only, RESULTS INVALID): the same five reads at conflict-free addresses (lane
    // stride 8 / 16 bytes) -- what the step's 4-way (8-byte) and 2-way (16-byte) bank conflicts cost
    const int l = threadIdx.x;
    const f2 za0 = buf[l + (k0 & 1024)];
    const f4v za12 = *(const f4v *)(buf + 2 * l + 512);
    const f2 za3 = buf[l + 256 + (k0 & 1024)];
    const f4v zb10 = *(const f4v *)(buf + 2 * l + 2048);
    const f4v zb32 = *(const f4v *)(buf + 2 * l + 3072);
#else
    const f2 za0 = buf[k0];
    const f4v za12 = *(const f4v *)(buf + k0 + 1);
    const f2 za3 = buf[k0 + 3 == M_HALF ? 0 : k0 + 3];
    const f4v zb10 = *(const f4v *)(buf + (M_HALF - k0 - 1));
    const f4v zb32 = *(const f4v *)(buf + (M_HALF - k0 - 3));
#endif
    za[0] = za0;
    za[1] = mk2(za12.x, za12.y);
    za[2] = mk2(za12.z, za12.w);
    za[3] = za3;
    zb[0] = mk2(zb10.z, zb10.w);
    zb[1] = mk2(zb10.x, zb10.y);
    zb[2] = mk2(zb32.z, zb32.w);
    zb[3] = mk2(zb32.x, zb32.y);
}

// Power-plane stores.  PB_NT_STORES: as streaming (non-temporal) stores -- 671 MB per launch of planes that
// nobody on this die reads again pass through a 4-MB L2 and push out the rows' bytes that a workgroup requests a
// second time 20 us later.
#ifndef PB_NT_STORES      // 0 ordinary, 1 non-temporal
#define PB_NT_STORES 1
#endif
// four consecutive channels c4 .. c4 + 3 of the row whose plane starts at `row` (wave-uniform)
__device__ __forceinline__ void store_plane4(float *row, int c4, float a, float b, float c, float d)
{
    typedef float f4s __attribute__((ext_vector_type(4)));
    const f4s v = {a, b, c, d};
#ifdef CH_ABL
#if CH_ABL & 2
    if (a == 123.456f) *(f4s *)(row + c4) = v;      // (experiments: the stores are compiled but never executed)
    return;
#endif
#endif
#if PB_NT_STORES
    __builtin_nontemporal_store(v, (f4s *)(row + c4));
#else
    *(f4s *)(row + c4) = v;
#endif
}

__device__ __forceinline__ float cvt_sample_c(unsigned u) { return u == 0 ? 0.0f : (float)u / 128 - 1; }

// convertarray (src/pb_kernels.cu:23-33) on packed codes.  u / 128 and the subtraction of 1 are both
// exact in binary32, so fma(u, 1/128, -1) is the same number; code 0 ("no sample" -> 0.0) is first
// rewritten as code 128, whose value is 0.0.
__device__ __forceinline__ unsigned fix_zero_codes(unsigned w)
{
    const unsigned t = ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w;   // bit 7 of a byte set <=> byte != 0
    return w | (~t & 0x80808080u);
}
__device__ __forceinline__ uint4 fix_zero_codes(uint4 q)
{
    return make_uint4(fix_zero_codes(q.x), fix_zero_codes(q.y), fix_zero_codes(q.z), fix_zero_codes(q.w));
}
// two fixed codes (low byte = even sample = re, next byte = odd sample = im) -> (re, im)
__device__ __forceinline__ f2 cvt_pair_c(unsigned w)
{
    const f2 u = mk2((float)(w & 0xffu), (float)((w >> 8) & 0xffu));
    return pkfma(u, mk2(0.0078125f, 0.0078125f), mk2(-1.0f, -1.0f));
}

