/* dev tool: exhaustive check of pointrcnn_amd/csrc/ref_trig.h against the HOST libm (glibc) it restates.
 *   gcc -O2 -ffp-contract=off -o /tmp/ref_trig_check tools/ref_trig_check.c -lm && /tmp/ref_trig_check
 * sinf / cosf: every float with |x| < 120 (2.25e9 values, ~25 s); atanf: every positive float; atan2f: 4e8 random pairs.
 * Expected on an FMA-capable x86-64 host with glibc 2.35: 0 mismatches everywhere (recorded in DESIGN.md section 2). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../pointrcnn_amd/csrc/ref_trig.h"

int main(void) {
    long bad_s = 0, bad_c = 0, n = 0, bad_a = 0, bad_a2 = 0;
    const uint32_t lim = rt_asuint(120.0f);
    for (uint32_t u = 0; u < lim; u++)
        for (int sg = 0; sg < 2; sg++) {
            const float x = rt_asfloat(u | ((uint32_t)sg << 31));
            float a = sinf(x), b = prcnn_ref_sinf(x);
            bad_s += memcmp(&a, &b, 4) != 0;
            a = cosf(x); b = prcnn_ref_cosf(x);
            bad_c += memcmp(&a, &b, 4) != 0;
            n++;
        }
    printf("sinf / cosf: %ld floats with |x| < 120: %ld / %ld mismatches\n", n, bad_s, bad_c);
    for (uint32_t u = 0; u < 0x7f800000u; u++) { const float x = rt_asfloat(u); float a = atanf(x), b = rt_atanf(x); bad_a += memcmp(&a, &b, 4) != 0; }
    printf("atanf: all positive floats: %ld mismatches\n", bad_a);
    srand(1);
    for (long i = 0; i < 400000000L; i++) {
        const uint32_t r1 = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand(), r2 = (uint32_t)rand() * 40503u ^ ((uint32_t)rand() << 11);
        float y = ((int32_t)r1) * (1.0f / 2147483648.0f) * 8.0f, x = ((int32_t)r2) * (1.0f / 2147483648.0f) * 8.0f;
        if (i & 1) y *= 1e-3f;
        if (i & 2) x *= 1e-2f;
        float a = atan2f(y, x), b = prcnn_ref_atan2f(y, x);
        bad_a2 += memcmp(&a, &b, 4) != 0;
    }
    printf("atan2f: 400000000 random pairs: %ld mismatches\n", bad_a2);
    return (bad_s || bad_c || bad_a || bad_a2) ? 1 : 0;
}
