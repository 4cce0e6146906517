/* oracle/qdq_core.c - plain-C restatement of the two elementwise cores of the hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/quant_oracle.py): used by tests/ to cross-check the
 * torch-based oracle's IEEE semantics (true division, separate roundings, rint vs roundf).
 * (bench.py's cpu_baseline times the torch-based oracle, not this file.)  Never linked into the product.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/build_oracle.py).
 *
 *  oracle_pc_qdq   follows pytorch_quantizer/quantization/qtypes/int_quantizer.py:573-592
 *                  (div, add, clamp/where, round half-even, sub, mul) on native NCHW with the
 *                  per-channel scale / zero_point / qmax of :559-572.
 *  oracle_pt_qdq   follows kernels/gemmlowp.cu:8-25 (fminf/fmaxf/roundf) given scale/shift.
 */
#include <math.h>
#include <stdint.h>

void oracle_pc_qdq(const float* x, float* y, uint8_t* codes, int64_t N, int64_t C, int64_t HW,
                   const float* scale, const float* zp, const float* qmax) {
    for (int64_t n = 0; n < N; ++n)
        for (int64_t c = 0; c < C; ++c) {
            const float s = scale[c], z = zp[c], qm = qmax[c];
            const float* xi = x + (n * C + c) * HW;
            float* yi = y + (n * C + c) * HW;
            for (int64_t i = 0; i < HW; ++i) {
                volatile float q = xi[i] / s;
                q = q + z;
                float t = q;
                t = (t > qm) ? qm : t;
                t = (t < 0.f) ? 0.f : t;
                t = rintf(t); /* round-half-even in the default rounding mode */
                if (codes) codes[(n * C + c) * HW + i] = (uint8_t)t;
                volatile float d = t - z;
                yi[i] = d * s;
            }
        }
}

void oracle_pt_qdq(const float* x, float* y, int64_t n, float scale, float shift, float qmax, int true_zero) {
    for (int64_t i = 0; i < n; ++i) {
        volatile float t = true_zero ? (x[i] / scale) : (x[i] + shift);
        t = true_zero ? (t + shift) : (t / scale);
        t = t + 0.f;
        float u = fminf(t, qmax);
        u = fmaxf(u, 0.f);
        u = roundf(u);
        volatile float d = true_zero ? (u - shift) : (u * scale);
        y[i] = true_zero ? (d * scale) : (d - shift);
    }
}
