// Measurement tool (not part of the library): synthetic co-runners that each take ONE kind of resource from a kernel
// running beside them, for a fixed time -- LDS capacity (workgroup slots), vector issue slots, LDS bandwidth, HBM read
// bandwidth.  tools/corun_probe.py runs the channeliser (experiments build, detect's launch left out) beside each of
// them and reads the channeliser's per-launch time: what of detect's presence is it that costs the channeliser its
// 0.39 -> 0.58 ms?   Build: hipcc -O3 -shared -fPIC --offload-arch=gfx950 -o build/libcorun.so tools/corun.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t now100() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz
// `n` x 64 cycles of sleep (the instruction takes an immediate: units of 8)
__device__ __forceinline__ void nap(int n)
{
    for (; n >= 8; n -= 8) __builtin_amdgcn_s_sleep(8);
    for (; n > 0; --n) __builtin_amdgcn_s_sleep(1);
}

// hold `lds` bytes of LDS and nothing else: sleeps until t_end
__global__ void k_hold(uint64_t ticks, unsigned *sink)
{
    extern __shared__ unsigned smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    const uint64_t t_end = now100() + ticks;
    while (now100() < t_end) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0 && smem[0] == 12345) sink[0] = 1;
}

// vector instructions at a given duty: `burst` dependent-free packed FMAs, then s_sleep(idle)
__global__ void k_valu(uint64_t ticks, int burst, int idle, float *sink)
{
    extern __shared__ unsigned vsm[];          // (optional dynamic LDS: the footprint of a small workgroup that also computes)
    if (threadIdx.x == 0) vsm[0] = 1;
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
    const float m = 1.0000001f, c = 1e-9f;
    const uint64_t t_end = now100() + ticks;
    while (now100() < t_end) {
        for (int i = 0; i < burst; ++i) {
            a0 = __builtin_fmaf(a0, m, c);
            a1 = __builtin_fmaf(a1, m, c);
            a2 = __builtin_fmaf(a2, m, c);
            a3 = __builtin_fmaf(a3, m, c);
        }
        if (idle) nap(idle);
    }
    if (a0 + a1 + a2 + a3 == 12345.f) sink[0] = a0;
}

// LDS traffic: every lane reads and writes 16 bytes per iteration, `idle` sleep between bursts of 16
__global__ void k_ldsbw(uint64_t ticks, int idle, float *sink)
{
    __shared__ float4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    const uint64_t t_end = now100() + ticks;
    int j = threadIdx.x;
    while (now100() < t_end) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 v = buf[(j + 64 * i) & 1023];
            acc.x += v.x;
            buf[(j + 64 * i + 512) & 1023] = acc;
        }
        if (idle) nap(idle);
    }
    if (acc.x == 12345.f) sink[0] = acc.x;
}

// HBM reads: grid-stride float4 loads over `n4` elements, round and round until t_end; `idle` throttles
__global__ void k_hbm(uint64_t ticks, const float4 *src, size_t n4, int idle, float *sink, unsigned long long *bytes)
{
    float acc = 0.f;
    const uint64_t t_end = now100() + ticks;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long n = 0;
    while (now100() < t_end) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = src[i];
            acc += v.x + v.w;
            i += (size_t)gridDim.x * blockDim.x;
            if (i >= n4) i -= n4;
        }
        n += 8 * 16;
        if (idle) nap(idle);
    }
    if (acc == 12345.f) sink[0] = acc;
    if (bytes) atomicAdd(bytes, n);
}

extern "C" int corun_hold(void *stream, int nwg, int threads, int lds_bytes, double ms, void *sink)
{
    (void)hipFuncSetAttribute((const void *)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    k_hold<<<nwg, threads, lds_bytes, (hipStream_t)stream>>>((uint64_t)(ms * 1e5), (unsigned *)sink);
    return (int)hipGetLastError();
}
extern "C" int corun_valu(void *stream, int nwg, int threads, int burst, int idle, double ms, void *sink)
{
    k_valu<<<nwg, threads, 16, (hipStream_t)stream>>>((uint64_t)(ms * 1e5), burst, idle, (float *)sink);
    return (int)hipGetLastError();
}
// the same with `lds_bytes` of LDS held
extern "C" int corun_valu_lds(void *stream, int nwg, int threads, int burst, int idle, int lds_bytes, double ms, void *sink)
{
    (void)hipFuncSetAttribute((const void *)k_valu, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    k_valu<<<nwg, threads, lds_bytes, (hipStream_t)stream>>>((uint64_t)(ms * 1e5), burst, idle, (float *)sink);
    return (int)hipGetLastError();
}
extern "C" int corun_ldsbw(void *stream, int nwg, int threads, int idle, double ms, void *sink)
{
    k_ldsbw<<<nwg, threads, 0, (hipStream_t)stream>>>((uint64_t)(ms * 1e5), idle, (float *)sink);
    return (int)hipGetLastError();
}
extern "C" int corun_hbm(void *stream, int nwg, int threads, const void *src, size_t nbytes, int idle, double ms, void *sink,
                         void *bytes)
{
    k_hbm<<<nwg, threads, 0, (hipStream_t)stream>>>((uint64_t)(ms * 1e5), (const float4 *)src, nbytes / 16, idle, (float *)sink,
                                                    (unsigned long long *)bytes);
    return (int)hipGetLastError();
}
