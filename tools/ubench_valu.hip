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
    __shared__ float dummy_lds[4096];
    dummy_lds[threadIdx.x] = 0.f;
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
        } else if (KIND == 8) {   // the detect recurrence, select form as the compiler emits it (wait states after v_cmp)
            asm volatile(REP8(REP8("v_mul_f32 %3, %5, %4\n v_mul_f32 %1, %0, %5\n v_mul_f32 %2, 0x41300000, %0\n v_add_f32 %1, %1, %3\n"
                                   "v_cmp_gt_f32 vcc, %4, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc\n"))
                         : "+v"(a0.x), "=&v"(a1.x), "=&v"(a2.x), "=&v"(a3.x)
                         : "v"(b.x), "v"(c.x)
                         : "vcc");
        } else if (KIND == 9) {   // execution-mask form: v_cmpx, add under the mask, restore exec
            asm volatile(REP8(REP8("v_mul_f32 %3, %5, %4\n v_mul_f32 %1, %0, %5\n v_mul_f32 %2, 0x41300000, %0\n"
                                   "v_cmpx_ngt_f32 %4, %2\n v_add_f32 %0, %1, %3\n s_mov_b64 exec, -1\n"))
                         : "+v"(a0.x), "=&v"(a1.x), "=&v"(a2.x), "=&v"(a3.x)
                         : "v"(b.x), "v"(c.x)
                         : "vcc");
        } else if (KIND == 10) {  // sign-mask form: no VCC, no exec
            asm volatile(REP8(REP8("v_mul_f32 %3, %5, %4\n v_mul_f32 %1, %0, %5\n v_mul_f32 %2, 0x41300000, %0\n v_add_f32 %1, %1, %3\n"
                                   "v_sub_f32 %2, %2, %4\n v_ashrrev_i32 %2, 31, %2\n v_bfi_b32 %0, %2, %0, %1\n"))
                         : "+v"(a0.x), "=&v"(a1.x), "=&v"(a2.x), "=&v"(a3.x)
                         : "v"(b.x), "v"(c.x));
        } else if (KIND == 11) {  // select form with the wait states filled by the next row's independent product
            asm volatile(REP8(REP8("v_mul_f32 %1, %0, %5\n v_mul_f32 %2, 0x41300000, %0\n v_add_f32 %1, %1, %3\n"
                                   "v_cmp_gt_f32 vcc, %4, %2\n v_mul_f32 %3, %5, %4\n s_nop 0\n v_cndmask_b32 %0, %1, %0, vcc\n"))
                         : "+v"(a0.x), "=&v"(a1.x), "=&v"(a2.x), "+v"(a3.x)
                         : "v"(b.x), "v"(c.x)
                         : "vcc");
        } else if (KIND == 12) {  // select form + an LDS store of the result every row
            asm volatile(REP8(REP8("v_mul_f32 %1, %0, %5\n v_mul_f32 %2, 0x41300000, %0\n v_add_f32 %1, %1, %3\n"
                                   "v_cmp_gt_f32 vcc, %4, %2\n v_mul_f32 %3, %5, %4\n s_nop 0\n v_cndmask_b32 %0, %1, %0, vcc\n ds_write_b32 %6, %0\n"))
                         : "+v"(a0.x), "=&v"(a1.x), "=&v"(a2.x), "+v"(a3.x)
                         : "v"(b.x), "v"(c.x), "v"(threadIdx.x * 4)
                         : "vcc");
        } else if (KIND == 13) {  // pol scrunch: cvt f32->f64, mul f64, cvt f64->f32 (dependent triple)
            double d;
            asm volatile(REP8(REP8("v_cvt_f64_f32 %1, %0\n v_mul_f64 %1, %1, %2\n v_cvt_f32_f64 %0, %1\n"))
                         : "+v"(a0.x), "=&v"(d) : "v"(0.70710678118654752440));
        } else if (KIND == 14) {  // IEEE f32 division as the compiler expands it (10 instructions)
            asm volatile(REP8(REP8("v_div_scale_f32 %1, s[10:11], %4, %4, %0\n v_rcp_f32 %2, %1\n v_div_scale_f32 %3, vcc, %0, %4, %0\n"
                                   "v_fma_f32 %5, -%1, %2, 1.0\n v_fmac_f32 %2, %5, %2\n v_mul_f32 %5, %3, %2\n v_fma_f32 %6, -%1, %5, %3\n"
                                   "v_fmac_f32 %5, %6, %2\n v_fma_f32 %1, -%1, %5, %3\n v_div_fmas_f32 %1, %1, %2, %5\n v_div_fixup_f32 %0, %1, %4, %0\n"))
                         : "+v"(a0.x), "=&v"(a1.x), "=&v"(a2.x), "=&v"(a3.x)
                         : "v"(b.x), "v"(a4.x), "v"(a5.x)
                         : "vcc", "s10", "s11");
        } else if (KIND == 15) {  // v_rcp_f32 alone (independent)
            asm volatile(REP8(REP8("v_rcp_f32 %0, %1\n")) : "=v"(a0.x) : "v"(b.x));
        } else if (KIND == 16) {  // v_cvt_f64_f32 alone
            double d;
            asm volatile(REP8(REP8("v_cvt_f64_f32 %0, %1\n")) : "=v"(d) : "v"(b.x));
        } else if (KIND == 17) {  // v_mul_f64 alone
            double d = 1.0;
            asm volatile(REP8(REP8("v_mul_f64 %0, %0, %1\n")) : "+v"(d) : "v"(0.999999));
        } else if (KIND == 18) {  // ds_read_b32 alone
            asm volatile(REP8(REP8("ds_read_b32 %0, %1\n")) "s_waitcnt lgkmcnt(0)" : "=v"(a0.x) : "v"(threadIdx.x * 4));
        } else if (KIND == 19) {  // ds_read_b128 alone
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 q;
            asm volatile(REP8(REP8("ds_read_b128 %0, %1\n")) "s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"((threadIdx.x & 63) * 16));
            a0.x += q.x;
        } else if (KIND == 20) {  // ds_write_b128 alone
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 q = {a0.x, a1.x, a2.x, a3.x};
            asm volatile(REP8(REP8("ds_write_b128 %1, %0\n")) "s_waitcnt lgkmcnt(0)" : : "v"(q), "v"((threadIdx.x & 63) * 16));
        } else if (KIND == 21) {  // ds_write_b32 alone
            asm volatile(REP8(REP8("ds_write_b32 %1, %0\n")) "s_waitcnt lgkmcnt(0)" : : "v"(a0.x), "v"(threadIdx.x * 4));
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
    run<8>("detect row, select + s_nop 1 (7 slots)", 64);
    run<9>("detect row, v_cmpx form (6 slots)", 64);
    run<10>("detect row, sign-mask form (7 slots)", 64);
    run<11>("detect row, select, hazard filled (7)", 64);
    run<12>("detect row, select filled + ds_write (8)", 64);
    run<13>("cvt f64<-f32, mul f64, cvt f32<-f64 (triple)", 64);
    run<14>("IEEE f32 division expansion (11 instr)", 64);
    run<15>("v_rcp_f32", 64);
    run<16>("v_cvt_f64_f32", 64);
    run<17>("v_mul_f64 dependent", 64);
    run<18>("ds_read_b32", 64);
    run<19>("ds_read_b128", 64);
    run<20>("ds_write_b128", 64);
    run<21>("ds_write_b32", 64);
    return 0;
}
