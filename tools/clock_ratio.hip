// Timing experiment: s_memtime (shader clock) ticks per microsecond of s_memrealtime (100 MHz) on a lightly and on a
// fully loaded chip: one workgroup spinning alone, then 4096 workgroups of packed-FMA + LDS work around it.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(unsigned long long *out, int iters, int heavy)
{
    __shared__ float lds[4096];
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fmaf(a, b, c);
        d = __builtin_fmaf(d, b, a);
        if (heavy) {
            lds[(threadIdx.x * 17 + i) & 4095] = a;
            c = __builtin_fmaf(c, b, lds[(threadIdx.x * 5 + i) & 4095]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a + d + c == 12345.678f) out[2] = 1;
}
int main()
{
    unsigned long long *d, h[3];
    hipMalloc((void **)&d, 24);
    for (int rep = 0; rep < 3; ++rep) {
        k_spin<<<1, 64>>>(d, 400000, 0);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("one wave alone:       %llu ticks in %.1f us -> %.0f ticks per us\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
    }
    for (int rep = 0; rep < 4; ++rep) {
        k_spin<<<8192, 768>>>(d, 60000, 1);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("chip full of FMA+LDS: %llu ticks in %.1f us -> %.0f ticks per us\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
    }
    return 0;
}
