// Micro-benchmark (diagnostic, not product): vector-issue cost on gfx950 of the instructions the
// channeliser and the detect kernel are made of, at 1..4 waves per SIMD.
//   build: hipcc -O3 --offload-arch=gfx950 -o gpurun_out/ubench_valu tools/ubench_valu.hip
//   run  : ./ubench_valu            (prints cycles per wave-instruction, s_memtime ticks)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(1024) void k_issue(float *out, long long *cyc, int iters)
{
    f2 a0 = {1.f + threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
       a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // independent v_pk_fma_f32
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                              "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                              "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b), "v"(c));
        } else if (KIND == 1) {   // independent v_fma_f32
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                              "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                              "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                         : "v"(b.x), "v"(c.x));
        } else if (KIND == 2) {   // dependent v_fma_f32 chain
            asm volatile(REP8(REP8("v_fma_f32 %0, %0, %1, %2\n")) : "+v"(a0.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 3) {   // dependent v_pk_fma_f32 chain
            asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %1, %2\n")) : "+v"(a0) : "v"(b), "v"(c));
        } else if (KIND == 4) {   // dependent mul -> add (the bandpass recurrence without contraction)
            asm volatile(REP8(REP8("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n")) : "+v"(a0.x) : "v"(b.x), "v"(c.x));
        } else if (KIND == 5) {   // recurrence with the 11x clip: mul, mul, add, cmp, cndmask
            asm volatile(REP8(REP8("v_mul_f32 %1, %0, %3\n v_mul_f32 %2, 0x41300000, %0\n v_add_f32 %1, %1, %4\n"
                                   "v_cmp_gt_f32 vcc, %4, %2\n v_cndmask_b32 %0, %1, %0, vcc\n"))
                         : "+v"(a0.x), "=&v"(a1.x), "=&v"(a2.x)
                         : "v"(b.x), "v"(c.x)
                         : "vcc");
        } else if (KIND == 6) {   // independent v_pk_mul_f32 / v_pk_add_f32 alternating
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n"
                              "v_pk_add_f32 %3, %3, %9\n v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n"
                              "v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b), "v"(c));
        } else if (KIND == 7) {   // independent v_add_f32 (no fma)
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n"
                              "v_mul_f32 %3, %3, %9\n v_add_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n"
                              "v_add_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9\n")
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                         : "v"(c.x), "v"(b.x));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND> static void run(const char *name, int per_iter)
{
    float *out;
    long long *cyc;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    hipMalloc(&cyc, 16 * 256 * sizeof(long long));
    const int iters = 200;
    printf("%-44s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        for (int rep = 0; rep < 2; ++rep) k_issue<KIND><<<256, 256 * wps>>>(out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(16 * 256);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        const int nw = 256 * 4 * wps;
        for (int i = 0; i < nw; ++i) s += (double)h[i];
        // cycles of wave life per wave-instruction, and SIMD cycles per wave-instruction (= that / waves per SIMD)
        const double per = s / nw / ((double)iters * per_iter);
        printf("  %dw/SIMD: %6.2f (%5.2f/SIMD)", wps, per, per / wps);
    }
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    printf("cycles (s_memtime ticks) of wave life per wave-instruction; in brackets SIMD cycles per instruction\n");
    run<0>("v_pk_fma_f32 independent x8", 64);
    run<1>("v_fma_f32 independent x8", 64);
    run<6>("v_pk_mul/v_pk_add independent x8", 64);
    run<7>("v_add/v_mul independent x8", 64);
    run<2>("v_fma_f32 dependent chain", 64);
    run<3>("v_pk_fma_f32 dependent chain", 64);
    run<4>("v_mul -> v_add dependent (per pair)", 64);
    run<5>("clip recurrence: 5 instr (per row)", 64);
    return 0;
}
