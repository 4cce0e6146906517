/* The quotient the single-launch kernels compute without the hardware divide (csrc/cnnq_qdq.hip.h, qdq1_fast), restated
 * in C and brute-forced against the C divide: rs = 1 / s (correctly rounded), q0 = x rs, r0 = x - s q0, q1 = q0 + r0 rs,
 * r1 = x - s q1, q = q1 + r1 rs with fmaf.  Inside the domain the kernels check (s in [1e-8, 2^30], |x| <= 2^70):
 *   - 2^-70 <= |x|: the QUOTIENT must be bit-identical to x / s;
 *   - |x| < 2^-70 (zeros, denormals): the CODE rint(clamp(q + zp, 0, qmax)) and y = (code - zp) s must be.
 * usage: fastdiv_check <scales> <dividends per scale> <seed>; prints the counts, exit status 1 on any mismatch. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static uint64_t st = 88172645463325252ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

static float quotient(float x, float s, float rs) {
    float q = x * rs;
    float r = fmaf(-s, q, x);
    q = fmaf(r, rs, q);
    r = fmaf(-s, q, x);
    q = fmaf(r, rs, q);
    return q;
}

static float code_of(float q, float zp, float qmax) {
    q = q + zp;
    q = (q > qmax) ? qmax : q;
    q = (q < 0.f) ? 0.f : q;
    return rintf(q);
}

int main(int argc, char** argv) {
    const long ns = argc > 1 ? atol(argv[1]) : 2000, nx = argc > 2 ? atol(argv[2]) : 20000;
    st ^= (uint64_t)(argc > 3 ? atol(argv[3]) : 1) * 0x9E3779B97F4A7C15ull;
    long badq = 0, badc = 0, nq = 0, nc = 0;
    for (long i = 0; i < ns; ++i) {
        uint32_t man = rnd() & 0x7fffff;
        const int k = (int)(i & 7);
        if (k == 0) man = 0x7fffff;                        /* all-ones significand */
        else if (k == 1) man = 0;                          /* power of two */
        else if (k == 2) man = 0x7fffff - (rnd() & 15);
        else if (k == 3) man = rnd() & 15;
        const int ex = (int)(rnd() % 57) - 26;             /* 2^-26 .. 2^30 */
        float s = u2f(((uint32_t)(ex + 127) << 23) | man);
        if (s < 1e-8f) s = 1e-8f;                          /* the scale floor (iq.py:566) */
        if (s > 0x1p30f) s = 0x1p30f;
        const float rs = 1.0f / s;
        const float qmax = (i & 16) ? 255.f : 15.f;
        const float zp = (i & 32) ? 0.f : rintf((float)((int64_t)(rnd() % 4001) - 2000));
        for (long j = 0; j < nx; ++j) {
            float x;
            const int m = (int)(j % 6);
            if (m == 0) {                                  /* random bits, 2^-70 .. 2^70 */
                x = u2f(((uint32_t)(rnd() & 1) << 31) | ((uint32_t)((int)(rnd() % 140) - 70 + 127) << 23) | (uint32_t)(rnd() & 0x7fffff));
            } else if (m == 1) {                           /* a few ulps around s * (q +- half an ulp of q) */
                const float q = u2f((127u << 23) | (uint32_t)(rnd() & 0x7fffff)) * (float)(1 << (rnd() % 12));
                const double xd = (double)s * ((double)q + (double)q * 5.9604645e-8 * ((rnd() & 1) ? 1 : -1));
                x = u2f(f2u((float)xd) + (uint32_t)((int)(rnd() % 5) - 2));
            } else if (m == 2) {                           /* a few ulps around s * (k or k + 1/2): the rounding ties of the code */
                const float q = (float)(rnd() % 4096) + 0.5f * (float)(rnd() & 1);
                x = u2f(f2u((float)((double)s * q)) + (uint32_t)((int)(rnd() % 7) - 3));
                if (rnd() & 1) x = -x;
            } else if (m == 3) {                           /* dividends near the scale's own binade */
                x = u2f(((uint32_t)(ex + (int)(rnd() % 24) - 4 + 127) << 23) | (uint32_t)(rnd() & 0x7fffff));
            } else if (m == 4) {                           /* the underflow side: zeros, denormals, tiny numbers */
                const int e = (int)(rnd() % 60);
                x = e == 0 ? 0.f : e < 12 ? u2f((uint32_t)(rnd() & 0x7fffff)) : u2f(((uint32_t)(127 - 70 - e) << 23) | (uint32_t)(rnd() & 0x7fffff));
                if (rnd() & 1) x = -x;
            } else {                                       /* significands near all-ones on both sides */
                x = u2f(((uint32_t)(ex + (int)(rnd() % 3) - 1 + 127) << 23) | (0x7fffff - (uint32_t)(rnd() & 63)));
            }
            if (!(fabsf(x) <= 0x1p70f)) continue;
            const float ref = x / s, got = quotient(x, s, rs);
            if (fabsf(x) >= 0x1p-70f) {
                ++nq;
                if (f2u(ref) != f2u(got)) { if (badq < 5) printf("quotient x=%a s=%a ieee=%a got=%a\n", x, s, ref, got); ++badq; }
            }
            ++nc;
            const float cr = code_of(ref, zp, qmax), cg = code_of(got, zp, qmax);
            const float yr = (cr - zp) * s, yg = (cg - zp) * s;
            if (f2u(cr) != f2u(cg) || f2u(yr) != f2u(yg)) { if (badc < 5) printf("code x=%a s=%a zp=%g ieee=%g got=%g\n", x, s, zp, cr, cg); ++badc; }
        }
    }
    printf("quotients %ld mismatches %ld   codes %ld mismatches %ld\n", nq, badq, nc, badc);
    return (badq || badc) ? 1 : 0;
}
