// Can a SMALL LDS ring keep up with detect's plane stream?  (profiles/r06_notes.md 4a: a detect workgroup that fits beside
// three channeliser workgroups has ~8 KB of LDS for its ring; at HBM latency that is too few bytes in flight -- unless the
// lines are already in L2 when the LDS-DMA asks for them.)  One workgroup per (32 channels, stream) like k_detect2, the
// same addresses (128-byte pieces, one per row and pol, 16 KB apart), chunks of T rows through a ring of NSLOT slots by
// global_load_lds_dwordx4, a consumer that takes `step_cycles` per chunk -- with and without a second wave that TOUCHES
// the pieces `ahead` chunks in front of the loader with ordinary loads (its own vmcnt; the values are never used).
//   hipcc -O2 --offload-arch=gfx950 -o build/ring_probe tools/ring_probe.hip && build/ring_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define NCH 4096
template <int T, int NSLOT>
__global__ __launch_bounds__(256) void k_ring(const float *__restrict__ plane, int R, int nrows_total, int step_cycles, int ahead,
                                              unsigned long long *out, float *sink)
{
    __shared__ __attribute__((aligned(16))) float s_p[NSLOT][T][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cg = blockIdx.x, stream = blockIdx.y;
    const size_t pol_stride = (size_t)R * NCH;
    const float *P = plane + (size_t)stream * 2 * pol_stride * (nrows_total / R);      // [seg][pol][row][ch] per stream
    const int nchunk = nrows_total / T, cps = R / T;
    constexpr int LPC = T / 4, DEPTH = NSLOT - 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (wave == 0) {
        // loader: lane -> (row in group of 4, pol, 4 channels)
        const int ld_row = lane >> 4, ld_pol = (lane >> 3) & 1, ld_c = cg * 32 + (lane & 7) * 4;
        int c = 0;
        auto issue = [&]() {
            const int seg = c / cps, rb = c % cps, slot = c % NSLOT;
            const float *src = P + (size_t)seg * 2 * pol_stride + (size_t)ld_pol * pol_stride + (size_t)(rb * T + ld_row) * NCH + ld_c;
#pragma unroll
            for (int i = 0; i < LPC; ++i)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + (size_t)(4 * i) * NCH),
                                                 (void __attribute__((address_space(3))) *)&s_p[slot][4 * i][0], 16, 0, 0);
            ++c;
        };
        for (int i = 0; i < DEPTH && c < nchunk; ++i) issue();
        for (int k = 0; k < nchunk; ++k) {
            // chunk k must have landed: at most (c - k - 1) chunks may still be in flight
            const int inflight = c - k - 1;
            if (inflight <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (inflight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
            else if (inflight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPC) : "memory");
            else if (inflight == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPC < 63 ? 4 * LPC : 63) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // chunk k is in LDS: consumers may read
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // consumers done with chunk k: its slot is free
            if (c < nchunk) issue();
        }
    } else if (wave == 1 && ahead > 0) {
        // toucher: one 4-byte load per 128-byte piece of the chunk `ahead` chunks in front of the one being consumed
        const int t_row = lane >> 1, t_pol = lane & 1;          // lanes 0 .. 2T-1
        for (int k = 0; k < nchunk; ++k) {
            const int c = k + ahead;
            if (c < nchunk && lane < 2 * T) {
                const int seg = c / cps, rb = c % cps;
                const float *q = P + (size_t)seg * 2 * pol_stride + (size_t)t_pol * pol_stride + (size_t)(rb * T + t_row) * NCH + cg * 32;
                // fire and forget: the value lands in a register nothing else uses and is never waited for (a load the
                // compiler knows about would be waited for at its first use, one memory latency per chunk, at a barrier
                // the whole workgroup shares)
                asm volatile("global_load_dword v100, %0, off" ::"v"(q) : "v100", "memory");
            }
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // consumers: read the chunk, spend the step
        float acc = 0.f;
        for (int k = 0; k < nchunk; ++k) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            acc += s_p[k % NSLOT][lane & (T - 1)][lane];
            const unsigned long long ts = __builtin_amdgcn_s_memtime();
            while ((long long)(__builtin_amdgcn_s_memtime() - ts) < step_cycles) __builtin_amdgcn_s_sleep(2);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (acc == 123.456f) sink[1] = acc;
    }
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

template <int T, int NSLOT>
static void run(const float *d_plane, int R, int nrows, int step_cycles, int ahead, unsigned long long *d_out, float *d_sink)
{
    dim3 grid(NCH / 32, 2);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        k_ring<T, NSLOT><<<grid, 256>>>(d_plane, R, nrows, step_cycles, ahead, d_out, d_sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double bytes = 2.0 * 2 * (double)nrows * NCH * 4;
    printf("T %2d ring %d slots (%5d B)  consumer %4d cycles/chunk  touch ahead %3d : %.3f ms per launch = %.2f TB/s\n", T, NSLOT,
           NSLOT * T * 256, step_cycles, ahead, best, bytes / best / 1e9);
}

int main()
{
    const int R = 1024, nseg = 10, nrows = R * nseg;
    const size_t nfl = (size_t)2 * 2 * nrows * NCH;        // two streams x two pols
    float *d_plane, *d_sink;
    unsigned long long *d_out;
    hipMalloc(&d_plane, nfl * 4);
    hipMemset(d_plane, 0, nfl * 4);
    hipMalloc(&d_sink, 64);
    hipMalloc(&d_out, 256 * 8);
    printf("detect's plane stream (671 MB per launch) through a per-workgroup LDS ring, 256 workgroups of 4 waves:\n");
    run<32, 5>(d_plane, R, nrows, 1700, 0, d_out, d_sink);      // k_detect2's shape
    run<32, 4>(d_plane, R, nrows, 1700, 0, d_out, d_sink);
    for (int step : {600, 300}) {
        run<8, 4>(d_plane, R, nrows, step, 0, d_out, d_sink);
        for (int ahead : {8, 16, 32}) run<8, 4>(d_plane, R, nrows, step, ahead, d_out, d_sink);
        run<8, 3>(d_plane, R, nrows, step, 0, d_out, d_sink);
        run<8, 3>(d_plane, R, nrows, step, 16, d_out, d_sink);
    }
    run<16, 3>(d_plane, R, nrows, 1000, 0, d_out, d_sink);
    run<16, 3>(d_plane, R, nrows, 1000, 8, d_out, d_sink);
    return 0;
}
