/* tests/c/cabi_demo.c - the drop-in boundary used from plain C (test infrastructure): include/cnnq_hip.h is a
 * C header, libcnnq_hip.so takes device pointers and a stream and nothing else.  Quantizes a small NCHW tensor
 * per channel to 4 bits with dynamic min/max (cnnq_pc_minmax_qdq) and checks the result against the same
 * arithmetic written out in scalar C (iq.py:559-592: scale = (max-min)/15, floor 1e-8, zp = rint(-min/scale),
 * q = clamp(rint(x/scale + zp)), y = (q - zp)*scale).
 *
 * build: gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/cabi_demo.c \
 *            -Lcnn_quantization_amd -lcnnq_hip -L/opt/rocm/lib -lamdhip64 -lm -o cabi_demo */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "cnnq_hip.h"

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)_e, __LINE__); return 2; } } while (0)

int main(void) {
    const int64_t N = 3, C = 5, HW = 35;
    const size_t n = (size_t)(N * C * HW);
    float* hx = (float*)malloc(n * sizeof(float));
    float* hy = (float*)malloc(n * sizeof(float));
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        hx[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * (float)(1 + (i / HW) % C);
    }
    float *dx, *dy, *pmm, *qp;
    const int G = cnnq_pc_groups(N, C, HW, 1);
    if (G <= 0) { fprintf(stderr, "cnnq_pc_groups -> %d\n", G); return 2; }
    CHECK(hipMalloc((void**)&dx, n * sizeof(float)));
    CHECK(hipMalloc((void**)&dy, n * sizeof(float)));
    CHECK(hipMalloc((void**)&pmm, (size_t)G * 2 * C * sizeof(float)));
    CHECK(hipMalloc((void**)&qp, (size_t)CNNQ_NQP * C * sizeof(float)));
    CHECK(hipMemcpy(dx, hx, n * sizeof(float), hipMemcpyHostToDevice));
    const int rc = cnnq_pc_minmax_qdq(dx, dy, N, C, HW, 4, 0, pmm, qp, NULL, NULL, NULL);
    if (rc != 0) { fprintf(stderr, "cnnq_pc_minmax_qdq -> %d\n", rc); return 2; }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hy, dy, n * sizeof(float), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int64_t c = 0; c < C; ++c) {
        float mn = INFINITY, mx = -INFINITY;
        for (int64_t b = 0; b < N; ++b)
            for (int64_t i = 0; i < HW; ++i) {
                const float v = hx[(b * C + c) * HW + i];
                mn = fminf(mn, v);
                mx = fmaxf(mx, v);
            }
        volatile float scale = (mx - mn) / 15.0f;
        if (scale < 1e-8f) scale = 1e-8f;
        const volatile float zp = rintf(0.0f - mn / scale);
        for (int64_t b = 0; b < N; ++b)
            for (int64_t i = 0; i < HW; ++i) {
                const size_t k = (size_t)((b * C + c) * HW + i);
                volatile float q = hx[k] / scale;
                q = q + zp;
                q = q > 15.0f ? 15.0f : q;
                q = q < 0.0f ? 0.0f : q;
                q = rintf(q);
                volatile float d = q - zp;
                const float ref = d * scale;
                if (memcmp(&ref, &hy[k], sizeof(float)) != 0) ++bad;
            }
    }
    printf("%s: %s, %zu of %zu elements differ\n", cnnq_version(), bad ? "MISMATCH" : "bit-exact", bad, n);
    hipFree(dx); hipFree(dy); hipFree(pmm); hipFree(qp);
    free(hx); free(hy);
    return bad ? 1 : 0;
}
