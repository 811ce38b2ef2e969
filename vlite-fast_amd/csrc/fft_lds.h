// LDS FFT building blocks shared by the channeliser kernels: radix-5/25/10 butterflies with
// compile-time twiddles, the 6250-point in-place Stockham FFT and the real-input split.  The
// operation order is the specification in oracle/pb_oracle.c "K6" (bit-exact).
#pragma once
#include "fft_consts.h"
#include "pb_internal.h"

#define M_HALF 6250
#ifndef FFT_PREFETCH
#define FFT_PREFETCH 1   // bit 0: pass-2 twiddles, bit 1: pass-3 twiddles requested one barrier early
#endif

namespace {
constexpr float kW25[25][2] = FC_W25_INIT;
constexpr float kW10[5][2] = FC_W10_INIT;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 w)
{
    const float t1 = a.y * w.y;
    const float re = __builtin_fmaf(a.x, w.x, -t1);
    const float t2 = a.y * w.x;
    const float im = __builtin_fmaf(a.x, w.y, t2);
    return make_float2(re, im);
}

__device__ __forceinline__ void dft5(float2 v0, float2 v1, float2 v2, float2 v3, float2 v4,
                                     float2 &y0, float2 &y1, float2 &y2, float2 &y3, float2 &y4)
{
    constexpr float C1 = FC_C1, C2 = FC_C2, S1 = FC_S1, S2 = FC_S2;
    const float2 t1 = make_float2(v1.x + v4.x, v1.y + v4.y);
    const float2 t2 = make_float2(v2.x + v3.x, v2.y + v3.y);
    const float2 t3 = make_float2(v1.x - v4.x, v1.y - v4.y);
    const float2 t4 = make_float2(v2.x - v3.x, v2.y - v3.y);
    y0.x = (v0.x + t1.x) + t2.x;
    y0.y = (v0.y + t1.y) + t2.y;
    float2 m1, m2, n1, n2;
    m1.x = __builtin_fmaf(C2, t2.x, __builtin_fmaf(C1, t1.x, v0.x));
    m1.y = __builtin_fmaf(C2, t2.y, __builtin_fmaf(C1, t1.y, v0.y));
    m2.x = __builtin_fmaf(C1, t2.x, __builtin_fmaf(C2, t1.x, v0.x));
    m2.y = __builtin_fmaf(C1, t2.y, __builtin_fmaf(C2, t1.y, v0.y));
    n1.x = __builtin_fmaf(S2, t4.x, S1 * t3.x);
    n1.y = __builtin_fmaf(S2, t4.y, S1 * t3.y);
    n2.x = __builtin_fmaf(-S1, t4.x, S2 * t3.x);
    n2.y = __builtin_fmaf(-S1, t4.y, S2 * t3.y);
    y1 = make_float2(m1.x + n1.y, m1.y - n1.x);
    y4 = make_float2(m1.x - n1.y, m1.y + n1.x);
    y2 = make_float2(m2.x + n2.y, m2.y - n2.x);
    y3 = make_float2(m2.x - n2.y, m2.y + n2.x);
}

__device__ __forceinline__ void dft25(float2 (&v)[25])
{
    // stage 1 in place: A[n2][k1] lives in v[5*k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        float2 a0, a1, a2, a3, a4;
        dft5(v[n2], v[5 + n2], v[10 + n2], v[15 + n2], v[20 + n2], a0, a1, a2, a3, a4);
        if (n2) {
            a1 = cmul(a1, make_float2(kW25[n2 * 5 + 1][0], kW25[n2 * 5 + 1][1]));
            a2 = cmul(a2, make_float2(kW25[n2 * 5 + 2][0], kW25[n2 * 5 + 2][1]));
            a3 = cmul(a3, make_float2(kW25[n2 * 5 + 3][0], kW25[n2 * 5 + 3][1]));
            a4 = cmul(a4, make_float2(kW25[n2 * 5 + 4][0], kW25[n2 * 5 + 4][1]));
        }
        v[n2] = a0;
        v[5 + n2] = a1;
        v[10 + n2] = a2;
        v[15 + n2] = a3;
        v[20 + n2] = a4;
    }
    // stage 2: for each k1 a DFT5 over n2; output k1 + 5 k2
    float2 o[25];
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1)
        dft5(v[5 * k1], v[5 * k1 + 1], v[5 * k1 + 2], v[5 * k1 + 3], v[5 * k1 + 4], o[k1], o[k1 + 5], o[k1 + 10],
             o[k1 + 15], o[k1 + 20]);
#pragma unroll
    for (int i = 0; i < 25; ++i) v[i] = o[i];
}

__device__ __forceinline__ void dft10(float2 (&v)[10])
{
    float2 A0[5], A1[5];
    dft5(v[0], v[2], v[4], v[6], v[8], A0[0], A0[1], A0[2], A0[3], A0[4]);
    dft5(v[1], v[3], v[5], v[7], v[9], A1[0], A1[1], A1[2], A1[3], A1[4]);
#pragma unroll
    for (int k1 = 1; k1 < 5; ++k1) A1[k1] = cmul(A1[k1], make_float2(kW10[k1][0], kW10[k1][1]));
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
        v[k1] = make_float2(A0[k1].x + A1[k1].x, A0[k1].y + A1[k1].y);
        v[k1 + 5] = make_float2(A0[k1].x - A1[k1].x, A0[k1].y - A1[k1].y);
    }
}

// Complex FFT of length 6250 of the sequence whose pass-1 butterfly inputs are already in
// v (thread tid < 250 holds z[tid + 250 r], r = 0..24).  Result Z[0..6249] in buf (natural order).
__device__ __forceinline__ void fft6250(float2 (&v)[25], float2 *buf, const float2 *__restrict__ tw2,
                                        const float2 *__restrict__ tw3, int tid)
{
    // Twiddles of the next pass are requested BEFORE the barriers that precede their use, so
    // that their L2 latency hides under this pass's arithmetic and LDS traffic.
    const int k = tid % 25;
    float2 t2[24];
    float2 t3[3][9];
    auto load_t2 = [&]() {
        if (tid < 250) {
#pragma unroll
            for (int r = 1; r < 25; ++r) t2[r - 1] = tw2[r * 25 + k];
        }
    };
    auto load_t3 = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = tid + 256 * i;
            if (j < 625) {
#pragma unroll
                for (int r = 1; r < 10; ++r) t3[i][r - 1] = tw3[r * 625 + j];
            }
        }
    };
#if FFT_PREFETCH & 1
    load_t2();
#endif
    // pass 1: R = 25, Ns = 1
    if (tid < 250) {
        dft25(v);
#pragma unroll
        for (int r = 0; r < 25; ++r) buf[tid * 25 + r] = v[r];
    }
    __syncthreads();
    // pass 2: R = 25, Ns = 25
    if (tid < 250) {
#pragma unroll
        for (int r = 0; r < 25; ++r) v[r] = buf[tid + 250 * r];
    }
    __syncthreads();
#if !(FFT_PREFETCH & 1)
    load_t2();
#endif
#if FFT_PREFETCH & 2
    load_t3();
#endif
    if (tid < 250) {
#pragma unroll
        for (int r = 1; r < 25; ++r) v[r] = cmul(v[r], t2[r - 1]);
        dft25(v);
        const int j0 = (tid / 25) * 625 + k;
#pragma unroll
        for (int r = 0; r < 25; ++r) buf[j0 + 25 * r] = v[r];
    }
    __syncthreads();
    // pass 3: R = 10, Ns = 625; butterflies j = tid, tid + 256, tid + 512
    float2 u[3][10];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = tid + 256 * i;
        if (j < 625) {
#pragma unroll
            for (int r = 0; r < 10; ++r) u[i][r] = buf[j + 625 * r];
        }
    }
    __syncthreads();
#if !(FFT_PREFETCH & 2)
    load_t3();
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = tid + 256 * i;
        if (j < 625) {
#pragma unroll
            for (int r = 1; r < 10; ++r) u[i][r] = cmul(u[i][r], t3[i][r - 1]);
            dft10(u[i]);
#pragma unroll
            for (int r = 0; r < 10; ++r) buf[j + 625 * r] = u[i][r];
        }
    }
    __syncthreads();
}

// real-input split: X[k] = 0.5 (E + T[k] O), E = Z[k] + conj Z[M-k], O = Z[k] - conj Z[M-k]
__device__ __forceinline__ float2 rsplit(const float2 *buf, const float2 *__restrict__ post, int k)
{
    const float2 a = buf[k == M_HALF ? 0 : k];
    float2 b = buf[k == 0 ? 0 : M_HALF - k];
    b.y = -b.y;
    const float2 E = make_float2(a.x + b.x, a.y + b.y);
    const float2 O = make_float2(a.x - b.x, a.y - b.y);
    const float2 P = cmul(O, post[k]);
    return make_float2(0.5f * (E.x + P.x), 0.5f * (E.y + P.y));
}

__device__ __forceinline__ float cvt_sample_c(unsigned u) { return u == 0 ? 0.0f : (float)u / 128 - 1; }

