/* tests/c/cabi_demo.c - the drop-in boundary used from plain C (test infrastructure): include/cnnq_hip.h is a
 * C header, libcnnq_hip.so takes device pointers and a stream and nothing else.  Quantizes a small NCHW tensor
 * per channel to 4 bits with dynamic min/max (cnnq_pc_minmax_qdq) and checks the result against the same
 * arithmetic written out in scalar C (iq.py:559-592: scale = (max-min)/15, floor 1e-8, zp = rint(-min/scale),
 * q = clamp(rint(x/scale + zp)), y = (q - zp)*scale).  The same tensor then goes through the one-call route the Python
 * host uses (cnnq_pc_minmax_qdq_auto with a group workspace from cnnq_group_ws_alloc: the single-launch kernels) on a
 * second, float4-friendly geometry, and through the two halves of the multi-GPU form with a world of one
 * (cnnq_pc_minmax_local_auto -> cnnq_pc_gathered_qdq); every route must return the same bits.
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

/* [40, 6, 56, 56]: rows of whole float4s, channels wider than a workgroup - the chain, the one-call route (single
 * launch, workgroups exchanging their extrema) and the two exchange halves with W = 1 must agree bit for bit */
static int routes_agree(void) {
    const int64_t N = 40, C = 6, HW = 56 * 56;
    const size_t n = (size_t)(N * C * HW);
    float* hx = (float*)malloc(n * sizeof(float));
    float* h0 = (float*)malloc(n * sizeof(float));
    float* h1 = (float*)malloc(n * sizeof(float));
    unsigned s = 777u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        hx[i] = ((float)(s >> 8) / 16777216.0f - 0.3f) * (float)(1 + (i / (size_t)HW) % (size_t)C);
    }
    float *dx, *y0, *y1, *y2, *pmm, *qp, *ws, *local;
    void* gws = NULL;
    /* the exchange workspace of the single-launch routes is sized by the library for the shape at hand (include/cnnq_hip.h:
     * cnnq_pc_group_workspace; the Python host allocates one 18 MB block that covers every layer of the BASELINE configs) */
    const size_t gws_bytes = cnnq_pc_group_workspace(N, C, HW);
    const int G = cnnq_pc_groups(N, C, HW, 1);
    const size_t wsb = cnnq_pc_minmax_qdq_workspace(N, C, HW);
    if (G <= 0 || wsb == 0 || gws_bytes == 0) { fprintf(stderr, "plan failed\n"); return 2; }
    CHECK(hipMalloc((void**)&dx, n * sizeof(float)));
    CHECK(hipMalloc((void**)&y0, n * sizeof(float)));
    CHECK(hipMalloc((void**)&y1, n * sizeof(float)));
    CHECK(hipMalloc((void**)&y2, n * sizeof(float)));
    CHECK(hipMalloc((void**)&pmm, (size_t)G * 2 * C * sizeof(float)));
    CHECK(hipMalloc((void**)&qp, (size_t)CNNQ_NQP * C * sizeof(float)));
    CHECK(hipMalloc((void**)&ws, wsb));
    CHECK(hipMalloc((void**)&local, (size_t)2 * C * sizeof(float)));
    CHECK(hipMemcpy(dx, hx, n * sizeof(float), hipMemcpyHostToDevice));
    int rc = cnnq_group_ws_alloc(gws_bytes, &gws);
    if (rc != 0) { fprintf(stderr, "cnnq_group_ws_alloc -> %d\n", rc); return 2; }
    rc = cnnq_pc_minmax_qdq(dx, y0, N, C, HW, 4, 0, pmm, qp, NULL, NULL, NULL);
    if (rc == 0) rc = cnnq_pc_minmax_qdq_auto(dx, y1, N, C, HW, 4, 0, ws, gws, gws_bytes, 1, NULL);
    if (rc == 0) rc = cnnq_pc_minmax_local_auto(dx, N, C, HW, pmm, gws, gws_bytes, local, NULL);
    if (rc == 0) rc = cnnq_pc_gathered_qdq(dx, y2, N, C, HW, local, 1, 4, 0, qp, NULL);
    if (rc != 0) { fprintf(stderr, "route call -> %d\n", rc); return 2; }
    CHECK(hipDeviceSynchronize());
    uint32_t status = 99;
    rc = cnnq_group_ws_status(gws, &status);
    if (rc != 0 || status != 0) { fprintf(stderr, "group workspace status %u (rc %d)\n", status, rc); return 2; }
    size_t bad = 0;
    CHECK(hipMemcpy(h0, y0, n * sizeof(float), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h1, y1, n * sizeof(float), hipMemcpyDeviceToHost));
    bad += memcmp(h0, h1, n * sizeof(float)) != 0;
    CHECK(hipMemcpy(h1, y2, n * sizeof(float), hipMemcpyDeviceToHost));
    bad += memcmp(h0, h1, n * sizeof(float)) != 0;
    printf("routes: chain / one-call single launch / exchange halves %s\n", bad ? "DIFFER" : "agree bit for bit");
    cnnq_group_ws_free(gws);
    hipFree(dx); hipFree(y0); hipFree(y1); hipFree(y2); hipFree(pmm); hipFree(qp); hipFree(ws); hipFree(local);
    free(hx); free(h0); free(h1);
    return bad ? 1 : 0;
}

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
    if (bad) return 1;
    return routes_agree();
}
