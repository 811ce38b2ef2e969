// How much LDS does a workgroup really take?  The dispatcher allocates LDS in granules; with three 50-KB channeliser
// workgroups per CU the granule decides how many bytes a workgroup may ADD (a twiddle table) before only two fit, and
// whether two of them still fit beside one detect workgroup.  Measured, not looked up: workgroups that hold `lds` bytes
// and sleep record (XCC, CU, start, end); the host reports the largest number that overlapped on one CU.
//   hipcc -O2 --offload-arch=gfx950 -o build/lds_granule tools/lds_granule.hip && build/lds_granule
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

__global__ void k_hold(unsigned long long *out, int sleep_us)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    if (threadIdx.x == 0) lds[0] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)sleep_us * 100ull) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t0;
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(20 | (3 << 11));    // HW_REG_XCC_ID[3:0]
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
    }
}

static int resident(int lds, int threads, int nwg)
{
    unsigned long long *d;
    hipMalloc(&d, (size_t)nwg * 32);
    hipFuncSetAttribute((const void *)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k_hold<<<nwg, threads, lds>>>(d, 300);
    if (hipDeviceSynchronize() != hipSuccess) { hipFree(d); return -1; }
    std::vector<unsigned long long> h((size_t)nwg * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int i = 0; i < nwg; ++i) {
        const unsigned long long hw = h[i * 4 + 3];
        const unsigned long long cu = (h[i * 4 + 2] << 16) | ((hw >> 8) & 0xff);      // XCC, then HW_ID[15:8] = SE_ID, SH_ID, CU_ID
        ev[cu].push_back({h[i * 4 + 0], +1});
        ev[cu].push_back({h[i * 4 + 1], -1});
    }
    int best = 0;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0;
        for (auto &e : kv.second) { cur += e.second; best = std::max(best, cur); }
    }
    return best;
}

int main()
{
    printf("CUs seen and workgroups of 256 threads resident per CU by dynamic LDS bytes:\n");
    int prev = -1;
    for (int lds = 48 * 1024; lds <= 84 * 1024; lds += 128) {
        const int r = resident(lds, 256, 256 * 6);
        if (r != prev) printf("  lds %6d B -> %d per CU\n", lds, r);
        prev = r;
    }
    // two channeliser-sized holders beside one detect-sized: largest channeliser LDS for which 2 x chan + 58368 fit is
    // read off the granule above; report the granule candidates directly
    for (int g : {128, 256, 512, 1024, 1280, 2048}) {
        auto up = [g](int x) { return (x + g - 1) / g * g; };
        printf("  if granule %4d: 3 x %d = %d (<= 163840: %s)\n", g, up(50008), 3 * up(50008), 3 * up(50008) <= 163840 ? "yes" : "no");
    }
    return 0;
}
