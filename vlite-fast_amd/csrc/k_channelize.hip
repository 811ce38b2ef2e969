// Channeliser: 8-bit unpack (+ optional 4-tap FIR window) + 12500-point real FFT + square-law
// detect of the 4096 output channels, one workgroup per (row, pol), everything between the raw
// bytes and the power spectrum staying in registers and LDS.
//
// Replaces, per FFT row:  convertarray (src/pb_kernels.cu:23-33), the copy/zero half of
// apply_kurtosis (:267-287), cufftExecR2C (src/process_baseband.cu:1222-1224), inject_frb
// (src/pb_kernels.cu:348-391) and the |X|^2 of detect_and_normalize2/3 (:416, :481).
//
// FFT: the 12500 real samples are packed as 6250 complex, transformed by three Stockham
// passes of radix 25, 25, 10 and split back into the real-input spectrum.  The operation
// order (butterfly formulas, where fmaf is used, twiddle tables) is the specification in
// oracle/pb_oracle.c "K6"; results agree with the oracle bit for bit.  Pass r of radix 25
// reads input r from 500-sample block r, so a flagged kurtosis block is simply a zeroed
// butterfly input.
//
// LDS: one 6250 x float2 buffer (50 000 B) used in place: every pass reads its inputs into
// registers, barriers, then writes.  3 workgroups per CU.
// HBM traffic per (row, pol): 12.5 KB of samples read, 16 KB (+16 KB excised stream) of
// power written (the excised stream's already divided by the row weight); twiddles stay in L2.
#include "pb_internal.h"

#define M_HALF 6250

__constant__ float2 c_w25[25];   // w25^(n2*k1), [n2][k1]
__constant__ float2 c_w10[5];    // w10^k1
__constant__ float c_r5[4];      // cos(2pi/5), cos(4pi/5), sin(2pi/5), sin(4pi/5)

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
    const float C1 = c_r5[0], C2 = c_r5[1], S1 = c_r5[2], S2 = c_r5[3];
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
    float2 A[5][5];  // [n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        dft5(v[n2], v[5 + n2], v[10 + n2], v[15 + n2], v[20 + n2], A[n2][0], A[n2][1], A[n2][2], A[n2][3],
             A[n2][4]);
        if (n2) {
#pragma unroll
            for (int k1 = 1; k1 < 5; ++k1) A[n2][k1] = cmul(A[n2][k1], c_w25[n2 * 5 + k1]);
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1)
        dft5(A[0][k1], A[1][k1], A[2][k1], A[3][k1], A[4][k1], v[k1], v[k1 + 5], v[k1 + 10], v[k1 + 15],
             v[k1 + 20]);
}

__device__ __forceinline__ void dft10(float2 (&v)[10])
{
    float2 A0[5], A1[5];
    dft5(v[0], v[2], v[4], v[6], v[8], A0[0], A0[1], A0[2], A0[3], A0[4]);
    dft5(v[1], v[3], v[5], v[7], v[9], A1[0], A1[1], A1[2], A1[3], A1[4]);
#pragma unroll
    for (int k1 = 1; k1 < 5; ++k1) A1[k1] = cmul(A1[k1], c_w10[k1]);
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
    // pass 1: R = 25, Ns = 1
    if (tid < 250) {
        dft25(v);
#pragma unroll
        for (int r = 0; r < 25; ++r) buf[tid * 25 + r] = v[r];
    }
    __syncthreads();
    // pass 2: R = 25, Ns = 25
    const int k = tid % 25;
    if (tid < 250) {
#pragma unroll
        for (int r = 0; r < 25; ++r) v[r] = buf[tid + 250 * r];
    }
    __syncthreads();
    if (tid < 250) {
#pragma unroll
        for (int r = 1; r < 25; ++r) v[r] = cmul(v[r], tw2[r * 25 + k]);
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
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = tid + 256 * i;
        if (j < 625) {
#pragma unroll
            for (int r = 1; r < 10; ++r) u[i][r] = cmul(u[i][r], tw3[r * 625 + j]);
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

struct ChanArgs {
    const uint8_t *in;      // [A][S][2][seg_samples]
    size_t in_ant_stride, seg_samples;
    const uint8_t *flags;   // [A][S*R*25]
    size_t flags_ant_stride;
    float *Praw, *Pkur;     // [A][S][2][R][4096]
    size_t p_ant_stride;
    const float2 *tw2, *tw3, *post;
    FrbParams frb;          // delays == nullptr: no injection
    int R, rfi_mode, inject_now;
};

__global__ __launch_bounds__(256) void k_channelize(ChanArgs a)
{
    __shared__ float2 buf[M_HALF];
    __shared__ unsigned smask;
    const int tid = threadIdx.x;
    const int grow = blockIdx.x;  // seg * R + row
    const int pol = blockIdx.y, ant = blockIdx.z;
    const int seg = grow / a.R, row = grow % a.R;

    if (a.rfi_mode) {
        if (tid < 64) {
            const uint8_t f = tid < PB_BLK_PER_FFT
                                  ? a.flags[(size_t)ant * a.flags_ant_stride + (size_t)grow * PB_BLK_PER_FFT + tid]
                                  : 0;
            const unsigned long long m = __ballot(f != 0);
            if (tid == 0) smask = (unsigned)m;
        }
    } else if (tid == 0) {
        smask = 0;
    }

    const uint16_t *src = (const uint16_t *)(a.in + (size_t)ant * a.in_ant_stride +
                                             ((size_t)seg * 2 + pol) * a.seg_samples + (size_t)row * PB_NFFT);
    // 25 sample pairs per thread, kept packed (two 16-bit pairs per register) so that the
    // excised pass can rebuild its inputs without a second trip to memory
    unsigned zp[13];
    if (tid < 250) {
#pragma unroll
        for (int r = 0; r < 25; r += 2) {
            const unsigned lo = src[tid + 250 * r];
            const unsigned hi = (r + 1 < 25) ? src[tid + 250 * (r + 1)] : 0u;
            zp[r >> 1] = lo | (hi << 16);
        }
    }
    __syncthreads();
    const unsigned mask = smask;
    const size_t prow = (size_t)ant * a.p_ant_stride + (((size_t)seg * 2 + pol) * a.R + row) * PB_NCHANOUT;

    // FRB injection window of this row, per channel (inject_frb :361-380)
    const bool inject = a.frb.delays != nullptr && a.inject_now > 0;
    const int since = inject ? (a.inject_now - 1 + seg) * a.R : 0;

    const bool all_bad = mask == 0x1ffffffu;
    // row weight exactly as apply_kurtosis accumulates it: one 500/12500 per unflagged block
    float wrow = 0.f;
    {
        const float inc = (float)PB_NKURTO / PB_NFFT;
        const int good = PB_BLK_PER_FFT - __popc(mask);
        for (int i = 0; i < good; ++i) wrow = wrow + inc;
    }
    const bool do_raw = a.rfi_mode != 1;
    const bool do_kur = a.rfi_mode != 0;
    // stream order: raw first
    for (int pass = 0; pass < 2; ++pass) {
        const bool is_kur = pass == 1;
        if (is_kur ? !do_kur : !do_raw) continue;
        float *P = (is_kur ? a.Pkur : a.Praw) + prow;
        if (is_kur && mask == 0 && do_raw) {
            // no flagged block in this row: the excised spectrum IS the raw spectrum
            // (buf still holds Z of the raw pass)
        } else if (is_kur && all_bad) {
            for (int c = tid; c < PB_NCHANOUT; c += 256) P[c] = 0.f;
            continue;
        } else {
            float2 v[25];
#pragma unroll
            for (int r = 0; r < 25; ++r) {
                const unsigned w = (zp[r >> 1] >> ((r & 1) * 16)) & 0xffffu;
                v[r] = (is_kur && ((mask >> r) & 1u)) ? make_float2(0.f, 0.f)
                                                      : make_float2(cvt_sample_c(w & 0xff), cvt_sample_c(w >> 8));
            }
            if (pass == 1 && do_raw) __syncthreads();  // raw pass's readers of buf are done
            fft6250(v, buf, a.tw2, a.tw3, tid);
        }
        for (int c = tid; c < PB_NCHANOUT; c += 256) {
            const int k = PB_CHANMIN + c;
            float2 X = rsplit(buf, a.post, k);
            if (inject) {
                const float d = a.frb.delays[k];
                const int lo = (int)(d + 0.5) - since;
                const int hi = (int)(d + a.frb.width + 0.5) - since;
                if (row >= lo && row <= hi) {
                    X.x *= a.frb.amp;
                    X.y *= a.frb.amp;
                }
            }
            const float xx = X.x * X.x;
            const float yy = X.y * X.y;
            const float pw = xx + yy;
            // the excised plane carries pow / w (detect_and_normalize3 :452,:481), so that the
            // serial bandpass recurrence downstream has no division in it
            P[c] = is_kur ? pw / wrow : pw;
        }
    }
}

hipError_t launch_channelize(pb_handle *h, int nseg, int inject_now)
{
    static bool consts_ready = false;
    if (!consts_ready) {
        float2 w25[25], w10[5];
        for (int a = 0; a < 5; ++a)
            for (int b = 0; b < 5; ++b) {
                const double ang = 2.0 * M_PI * (double)(a * b) / 25.0;
                w25[a * 5 + b] = make_float2((float)cos(ang), (float)(-sin(ang)));
            }
        for (int k = 0; k < 5; ++k) {
            const double ang = 2.0 * M_PI * (double)k / 10.0;
            w10[k] = make_float2((float)cos(ang), (float)(-sin(ang)));
        }
        float r5[4] = {(float)cos(2.0 * M_PI / 5.0), (float)cos(4.0 * M_PI / 5.0), (float)sin(2.0 * M_PI / 5.0),
                       (float)sin(4.0 * M_PI / 5.0)};
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_w25), w25, sizeof w25);
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_w10), w10, sizeof w10);
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_r5), r5, sizeof r5);
        if (e != hipSuccess) return e;
        consts_ready = true;
    }
    ChanArgs a;
    a.in = h->d_in;
    a.in_ant_stride = (size_t)h->S * 2 * h->seg_samples;
    a.seg_samples = h->seg_samples;
    a.flags = h->d_flags;
    a.flags_ant_stride = (size_t)h->S * h->nblk_seg;
    a.Praw = h->d_Praw;
    a.Pkur = h->d_Pkur;
    a.p_ant_stride = (size_t)h->S * 2 * h->R * PB_NCHANOUT;
    a.tw2 = h->ft.tw2;
    a.tw3 = h->ft.tw3;
    a.post = h->ft.post;
    a.frb.delays = (inject_now > 0) ? h->d_frb_delays : nullptr;
    const double rate = (double)h->R * PB_NFFT * 10;
    a.frb.width = (float)(2e-3 * 10 * rate / 10 / PB_NFFT);
    a.frb.amp = 1.05f;
    a.frb.since = 0;
    a.R = h->R;
    a.rfi_mode = h->cfg.rfi_mode;
    a.inject_now = inject_now;
    dim3 grid((unsigned)(nseg * h->R), 2, (unsigned)h->A);
    k_channelize<<<grid, 256, 0, h->stream>>>(a);
    return hipGetLastError();
}

// ---- float-input channeliser with optional 4-tap FIR (pb_channelize_f32) ----
// x: (nrows + taps - 1) rows of 12500 floats; out row t = rfft(sum_j taps[j] * x[row t + j]),
// the WOLA form of analysis/baseband.py:1226-1233.
__global__ __launch_bounds__(256) void k_channelize_f32(const float *__restrict__ x, int taps,
                                                        const float *__restrict__ fir, float2 *__restrict__ out,
                                                        const float2 *tw2, const float2 *tw3, const float2 *post)
{
    __shared__ float2 buf[M_HALF];
    const int tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const float *xr = x + row * PB_NFFT;
    float2 v[25];
    if (tid < 250) {
#pragma unroll
        for (int r = 0; r < 25; ++r) {
            const int n = 2 * (tid + 250 * r);
            if (taps == 1) {
                v[r] = *(const float2 *)(xr + n);
            } else {
                float2 acc = make_float2(0.f, 0.f);
                for (int j = 0; j < 4; ++j) {
                    const float2 s = *(const float2 *)(xr + (size_t)j * PB_NFFT + n);
                    const float2 t = *(const float2 *)(fir + (size_t)j * PB_NFFT + n);
                    const float px = t.x * s.x, py = t.y * s.y;
                    acc.x = j ? acc.x + px : px;
                    acc.y = j ? acc.y + py : py;
                }
                v[r] = acc;
            }
        }
    }
    fft6250(v, buf, tw2, tw3, tid);
    for (int k = tid; k < PB_NCHAN; k += 256) out[row * PB_NCHAN + k] = rsplit(buf, post, k);
}

hipError_t launch_channelize_f32(pb_handle *h, const float *d_x, int nrows, int taps, float2 *d_out)
{
    hipError_t e = launch_channelize(h, 0, 0);  // make sure the constant tables are loaded
    (void)e;
    (void)hipGetLastError();
    k_channelize_f32<<<nrows, 256, 0, h->stream>>>(d_x, taps, h->ft.taps, d_out, h->ft.tw2, h->ft.tw3, h->ft.post);
    return hipGetLastError();
}
