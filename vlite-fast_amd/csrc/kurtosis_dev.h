// Device pieces of the spectral-kurtosis statistic shared by k_kurtosis_row (k_kurtosis.hip) and the channeliser
// that computes the flags of its own row (k_channelize.hip): the reference's halving tree restated for wave64
//   kurtosis           src/pb_kernels.cu:35-107
//   compute_dagostino  :109-134
// (see k_kurtosis.hip for the structure and the exactness argument).
#pragma once
#include "pb_internal.h"

typedef float f2k __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float cvt_sample(unsigned u)
{
    return u == 0 ? 0.0f : (float)u / 128 - 1;
}

// t^(float(1/3)) by Newton cube root in IEEE double: the same operation sequence as
// orc_powf_third in the oracle, so that flags agree bit for bit (deviation 1).
__host__ __device__ inline float dev_powf_third(float t)
{
    double d = (double)t;
    unsigned long long bits = __builtin_bit_cast(unsigned long long, d);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    int q = (e >= 0) ? e / 3 : -((-e + 2) / 3);
    int r = e - 3 * q;
    bits = (bits & 0x000fffffffffffffULL) | ((unsigned long long)(1023 + r) << 52);
    double m = __builtin_bit_cast(double, bits);
    double y = 1.0 + (m - 1.0) * (1.0 / 7.0);
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        double y2 = y * y;
        y = y - (y2 * y - m) / (3.0 * y2);
    }
    double s = __builtin_bit_cast(double, (unsigned long long)(1023 + q) << 52);
    double z = (y - 1.0) / (y + 1.0);
    double z2 = z * z;
    double lny = 2.0 * z * (1.0 + z2 * (1.0 / 3.0 + z2 * (1.0 / 5.0 + z2 * (1.0 / 7.0))));
    double lnt = (double)(3 * q) * 0.69314718055994531 + 3.0 * lny;
    const double dexp = (double)(float)(1. / 3) - 1. / 3;
    return (float)(y * s * (1.0 + dexp * lnt));
}

// the score as a function of t, the argument of the cube root (t > 0)
__host__ __device__ inline float dag_of_t(float t, const DagConsts &c)
{
    const float v = (float)(c.Z1 * (c.Z2 - (double)dev_powf_third(t)));
    return __builtin_fabsf(v);
}
__host__ __device__ inline float dag_t(float kur, const DagConsts &c)
{
    return (float)(c.one_m_2_over_A / (1. + ((double)kur - 3. - c.mu1) * c.Z3));
}

__device__ inline float dag_one(float kur, const DagConsts &c)
{
    float dag = 9.0f;  // DAG_INF = DAG_THRESH + DAG_FB_THRESH + 1
    if (kur != 0.f) {  // true for NaN (all-zero block): t is NaN, t > 0 false, stays DAG_INF
        const float t = dag_t(kur, c);
        if (t > 0) dag = dag_of_t(t, c);
    }
    return dag;
}

// dag_one(kur) > DAG_THRESH without the cube root: t against the crossings found at pb_create (DagConsts), the
// score itself only in the bands around them.  Same decision as the score, float for float (tests: every flag of
// every parity test, and pb_debug_dag_check over every float kurtosis around both crossings).
__device__ inline bool dag_flag(float kur, const DagConsts &c)
{
    if (!(kur != 0.f)) return true;            // kur == 0: the score stays DAG_INF
    const float t = dag_t(kur, c);
    if (!(t > 0)) return true;                 // NaN (all-zero block) or a negative argument: DAG_INF
    if (t <= c.t_lo_sure || t >= c.t_hi_sure) return true;
    if (t >= c.t_lo_clear && t <= c.t_hi_clear) return false;
    return dag_of_t(t, c) > 3.0f;              // inside a band (rare)
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
// r[lane] += r[lane + 32] (lanes 0..31) and r[lane] += r[lane + 16] (lanes 0..15) with gfx950's lane-swap
// instructions: one VALU issue each where __shfl_down goes through the LDS crossbar (ds_bpermute, ~100 cycles
// of latency on a chain of 12-13 dependent block reductions per wave).  v_permlane32_swap a, b exchanges lanes
// 32..63 of a with lanes 0..31 of b; v_permlane16_swap the odd 16-lane rows of a with the even rows of b.
__device__ __forceinline__ float add_down32(float r)
{
    const unsigned u = __float_as_uint(r);
    return r + __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[1]);
}
__device__ __forceinline__ float add_down16(float r)
{
    const unsigned u = __float_as_uint(r);
    return r + __uint_as_float(__builtin_amdgcn_permlane16_swap(u, u, false, false)[1]);
}
// two blocks at once: lanes 0..31 get x[l] + x[l + 32], lanes 32..63 get y[l - 32] + y[l]
__device__ __forceinline__ float fold32(float x, float y)
{
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
// four blocks at once (x rows: A A B B, y rows: C C E E): rows (A, C, B, E), each row l < 16: v[l] + v[l + 16]
__device__ __forceinline__ float fold16(float x, float y)
{
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
template <int S> __device__ __forceinline__ float add_row_shl(float r)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x100 + S, 0xf, 0xf, true);
    return r + __int_as_float(t);
}


// Moments of ONE 500-sample block at sb (patched codes, in LDS): this lane's share of sum x^2 and sum x^4 before the
// cross-lane levels.  Leaves t = lane + 64 i (i = 0..3; i = 3 only for lane < 58) and their partners t + 250: two
// LEAVES share a packed register -- (t0, t1) and (t2, t3), partners likewise -- so that the pair sums
// x[t]^2 + x[t+250]^2 of two leaves are one packed add, and level 128 of the halving tree, (d0 + d2, d1 + d3),
// another; level 64 adds the halves (kurtosis, src/pb_kernels.cu:60-94: the same additions on the same operands).
__device__ __forceinline__ void block_leaves(const uint8_t *sb, int lane, float &r2, float &r4)
{
    const bool in3 = lane < 250 - 192;
    const int t3 = in3 ? lane + 192 : 0;
    f2k uA, uB, uC, uD;
    uA.x = (float)sb[lane];
    uA.y = (float)sb[lane + 64];
    uB.x = (float)sb[lane + 250];
    uB.y = (float)sb[lane + 314];
    uC.x = (float)sb[lane + 128];
    uC.y = (float)sb[t3];
    uD.x = (float)sb[lane + 378];
    uD.y = (float)sb[t3 + 250];
    const f2k k128 = {0.0078125f, 0.0078125f}, m1 = {-1.0f, -1.0f};
    const f2k xA = __builtin_elementwise_fma(uA, k128, m1), xB = __builtin_elementwise_fma(uB, k128, m1);
    const f2k xC = __builtin_elementwise_fma(uC, k128, m1), xD = __builtin_elementwise_fma(uD, k128, m1);
    const f2k aA = xA * xA, aB = xB * xB, aC = xC * xC, aD = xD * xD;
    const f2k qA = aA * aA, qB = aB * aB, qC = aC * aC, qD = aD * aD;
    const f2k e2ab = aA + aB, e4ab = qA + qB;        // (d[0], d[1])
    f2k e2cd = aC + aD, e4cd = qC + qD;              // (d[2], d[3])
    e2cd.y = in3 ? e2cd.y : 0.f;
    e4cd.y = in3 ? e4cd.y : 0.f;
    const f2k l2 = e2ab + e2cd, l4 = e4ab + e4cd;    // (d0 + d2, d1 + d3)
    r2 = l2.x + l2.y;
    r4 = l4.x + l4.y;
}

// Moments of a row's 2 x 25 blocks (pol 0's 12500 patched bytes at p0, pol 1's at p1, both in LDS) by the four
// waves of a workgroup: s2[bi], s4[bi] = sum x^2, sum x^4 of block bi (bi < 25: pol 0).  Each wave reduces 13, 13,
// 12, 12 consecutive blocks; four blocks share the cross-lane levels of the tree (one v_permlane32_swap + add folds
// the upper halves of two blocks at once, one v_permlane16_swap + add does level 16 of four, the DPP levels 8..1
// work inside 16-lane rows anyway): 20 cross-lane instructions per four blocks instead of 120.  The caller puts a
// barrier before reading s2 / s4.
__device__ __forceinline__ void row_block_moments(const uint8_t *p0, const uint8_t *p1, int wave, int lane,
                                                  float *s2, float *s4)
{
    auto at = [&](int bi) __attribute__((always_inline)) {
        return bi >= 25 ? p1 + (bi - 25) * PB_NKURTO : p0 + bi * PB_NKURTO;
    };
    const int bi0 = wave * 12 + min(wave, 2), bi1 = bi0 + (wave < 2 ? 13 : 12);
    int bi = bi0;
    for (; bi + 4 <= bi1; bi += 4) {
        float a2, a4, b2, b4, c2, c4, e2, e4;
        block_leaves(at(bi), lane, a2, a4);
        block_leaves(at(bi + 1), lane, b2, b4);
        block_leaves(at(bi + 2), lane, c2, c4);
        block_leaves(at(bi + 3), lane, e2, e4);
        // level 32: x' = (x lanes 0..31, y lanes 0..31), y' = (x lanes 32..63, y lanes 32..63)
        float ab2 = fold32(a2, b2), ab4 = fold32(a4, b4), ce2 = fold32(c2, e2), ce4 = fold32(c4, e4);
        // level 16: rows (A, C, B, E)
        float q2 = fold16(ab2, ce2), q4 = fold16(ab4, ce4);
        q2 = add_row_shl<8>(q2);
        q4 = add_row_shl<8>(q4);
        q2 = add_row_shl<4>(q2);
        q4 = add_row_shl<4>(q4);
        q2 = add_row_shl<2>(q2);
        q4 = add_row_shl<2>(q4);
        q2 = add_row_shl<1>(q2);
        q4 = add_row_shl<1>(q4);
        if ((lane & 15) == 0) {
            const int rowi = lane >> 4;                            // 0: A, 1: C, 2: B, 3: E
            const int dst = bi + ((rowi & 1) << 1) + (rowi >> 1);
            s2[dst] = q2;
            s4[dst] = q4;
        }
    }
    for (; bi < bi1; ++bi) {
        float r2, r4;
        block_leaves(at(bi), lane, r2, r4);
        r2 = add_down32(r2);
        r4 = add_down32(r4);
        r2 = add_down16(r2);
        r4 = add_down16(r4);
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
}
