/* Brute-force check that pt_div_by_reciprocal(a, b, 1/b) and pt_div_by_reciprocal2(a, b, 1/b, low part) == a / b bit for bit for every
 * (a, b) the kernels may feed it (predicates of csrc/pt_exact_div.h).  Built and run by
 * tests/test_exact_division.py.  Prints "<cases> <mismatches>". */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../path-tracing_amd/csrc/pt_exact_div.h"

static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint64_t rng(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }

int main(int argc, char **argv) {
    long n_random = argc > 1 ? atol(argv[1]) : 100000000L;
    unsigned long long total = 0, bad = 0;
#pragma omp parallel reduction(+ : total, bad)
    {
        uint64_t s = 0x9E3779B97F4A7C15ull;
#ifdef _OPENMP
        extern int omp_get_thread_num(void);
        s *= (uint64_t)(omp_get_thread_num() + 1);
#endif
#pragma omp for
        for (long i = 0; i < n_random; i++) {
            uint64_t r1 = rng(&s), r2 = rng(&s);
            uint32_t mb = r1 & 0x7fffff, ma = r2 & 0x7fffff;
            int eb = (int)((r1 >> 23) % 41) - 20, ea = (int)((r2 >> 23) % 181) - 80;      /* |a| in [2^-80, 2^101) */
            if ((i & 15) == 0) mb = 0x7ffeff - (uint32_t)((r1 >> 41) & 0xff);   /* just below the excluded top */
            if ((i & 15) == 1) mb = (uint32_t)((r1 >> 41) & 0xff);
            if ((i & 15) == 2) ma = ((r2 >> 41) & 1) ? 0x7fffff : 0;
            float b = asf((uint32_t)(((r1 >> 40) & 1) << 31) | ((uint32_t)(eb + 127) << 23) | mb);
            float a = asf((uint32_t)(((r2 >> 40) & 1) << 31) | ((uint32_t)(ea + 127) << 23) | ma);
            if ((i & 1023) == 3) a = 0.0f;
            if (!pt_div_safe_denominator(b)) continue;
            float y = 1.0f / b, want = a / b, got = pt_div_by_reciprocal(a, b, y);
            float got2 = pt_div_by_reciprocal2(a, b, y, pt_reciprocal_low(b, y));
            total++;
            if (asu(got) != asu(want) && !(got == 0.0f && want == 0.0f)) bad++;
            if (asu(got2) != asu(want) && !(got2 == 0.0f && want == 0.0f)) bad++;
        }
        /* adversarial: for every significand of b, quotients next to a rounding boundary */
#pragma omp for schedule(dynamic, 4096)
        for (uint32_t mb = 0; mb < 0x7fff00u; mb += 1) {
            float b = asf(0x3f800000u | mb), y = 1.0f / b, yl = pt_reciprocal_low(b, y);
            uint64_t t = 0xD1B54A32D192ED03ull ^ ((uint64_t)mb * 0x9E3779B97F4A7C15ull);
            for (int k = 0; k < 8; k++) {
                uint32_t mq = (uint32_t)(rng(&t) & 0x7fffff);
                double qmid = (double)asf(0x3f800000u | mq) + 0x1p-24;
                float a0 = (float)(qmid * (double)b);
                for (int dl = -1; dl <= 1; dl++) {
                    float a = asf(asu(a0) + (uint32_t)dl);
                    float want = a / b, got = pt_div_by_reciprocal(a, b, y), got2 = pt_div_by_reciprocal2(a, b, y, yl);
                    total++;
                    if (asu(got) != asu(want)) bad++;
                    if (asu(got2) != asu(want)) bad++;
                }
            }
        }
    }
    printf("%llu %llu\n", total, bad);
    return bad != 0;
}
