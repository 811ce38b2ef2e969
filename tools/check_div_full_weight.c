/* Exhaustive check of div_full_weight (k_channelize.hip): for every positive normal binary32 x >= 0x00800002,
 * x / (1 + 2^-23) computed by IEEE division equals the float whose bits are bits(x) - (m == 0 || m >= 0x400002 ? 2 : 1),
 * m = mantissa field of x.   gcc -O2 -ffp-contract=off -o /tmp/chk tools/check_div_full_weight.c && /tmp/chk */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline uint32_t fb(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void)
{
    const float w = bf(0x3f800001u);
    unsigned long long bad = 0, n = 0;
    for (uint32_t b = 0x00800002u; b < 0x7f800000u; ++b) {
        const uint32_t m1 = (b & 0x7fffffu) - 1u;
        const uint32_t fast = b - (m1 >= 0x400001u ? 2u : 1u);
        if (fb(bf(b) / w) != fast) ++bad;
        ++n;
    }
    printf("%llu values, %llu mismatches\n", n, bad);
    return bad != 0;
}
