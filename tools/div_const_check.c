// exactness of the FMA-corrected reciprocal division by the constants 3 and 6 (k1_tstat): fp32 exhaustive, fp64 random + edge sweep
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
static inline float divf_c(float a, float d, float r) { float q = a * r; float rem = fmaf(-q, d, a); return fmaf(rem, r, q); }
static inline double divd_c(double a, double d, double r) { double q = a * r; double rem = fma(-q, d, a); return fma(rem, r, q); }
static uint64_t sm64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main(void) {
    const float dfs[2] = {3.0f, 6.0f};
    for (int k = 0; k < 2; k++) {
        const float d = dfs[k], r = 1.0f / d;
        uint64_t bad = 0, badsub = 0;
#pragma omp parallel for reduction(+:bad,badsub)
        for (int64_t u = 0; u < (1ll << 32); u++) {
            uint32_t b = (uint32_t)u; float a; memcpy(&a, &b, 4);
            if (!isfinite(a)) continue;
            const float want = a / d, got = divf_c(a, d, r);
            if (memcmp(&want, &got, 4) != 0) { if (fabsf(want) < 1.1754944e-38f * 4) badsub++; else bad++; }
        }
        printf("fp32 / %g: exhaustive over finite inputs: %llu mismatches in the normal range, %llu with |a/d| < 4*FLT_MIN\n", d, (unsigned long long)bad, (unsigned long long)badsub);
    }
    const double dds[2] = {3.0, 6.0};
    for (int k = 0; k < 2; k++) {
        const double d = dds[k], r = 1.0 / d;
        uint64_t bad = 0;
#pragma omp parallel reduction(+:bad)
        {
            uint64_t s = 12345 + 977 * omp_get_thread_num();
            for (int64_t i = 0; i < 400000000ll; i++) {
                uint64_t b = sm64(&s);
                // exponents within +-200 of 1.0 (the sums of k1_tstat are far inside), random sign and mantissa
                uint64_t e = 1023 - 200 + (b >> 52) % 400;
                b = (b & 0x800fffffffffffffull) | (e << 52);
                double a; memcpy(&a, &b, 8);
                const double want = a / d, got = divd_c(a, d, r);
                if (memcmp(&want, &got, 8) != 0) bad++;
            }
        }
        // mantissa edge sweep: all-ones / all-zeros neighbourhoods
        for (uint64_t m = 0; m < 2000000; m++) for (int side = 0; side < 2; side++) {
            uint64_t b = (1023ull << 52) | (side ? (0xfffffffffffffull - m) : m);
            double a; memcpy(&a, &b, 8);
            const double want = a / d, got = divd_c(a, d, r);
            if (memcmp(&want, &got, 8) != 0) bad++;
        }
        printf("fp64 / %g: %llu mismatches in 8 x 4e8 random + 4e6 edge cases\n", d, (unsigned long long)bad);
    }
    return 0;
}
