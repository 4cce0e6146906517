// tools/ubench_stpol.hip - round 4: the cache-policy bits of the STORES (sc0 / sc1 / nt) in the sample-strided order the
// register-tile kernels are forced into: write only and copy, 32-row tiles of 4 KB rows.  (development aid)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ void st(f4* p, f4 v) {
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int R, int MODE, int POL>
__global__ void __launch_bounds__(256, 3) k_sp(const f4* __restrict__ x, f4* __restrict__ y, int N, int P4, int order) {
    const int S = N / R, ncb = P4 / 256;
    const int q = order == 0 ? (int)blockIdx.x % ncb : (int)blockIdx.x / S;
    const int s = order == 0 ? (int)blockIdx.x / ncb : (int)blockIdx.x % S;
    const size_t col = (size_t)q * 256 + threadIdx.x;
    f4 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const size_t e = (size_t)(s * R + j) * P4 + col;
        if (MODE == 2) { const float t = (float)(int)(e & 1023) * 1e-3f; v[j] = f4{t, t + 1.f, t + 2.f, t + 3.f}; }
        else v[j] = __builtin_nontemporal_load(x + e);
    }
    __builtin_amdgcn_sched_barrier(0);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < R; ++j) mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
    const float scale = mx > 1e30f ? 2.f : 1.0001f;
#pragma unroll
    for (int j = 0; j < R; ++j) st<POL>(y + (size_t)(s * R + j) * P4 + col, v[j] * scale);
}

template <int MODE, int POL>
static float run(const void* x, void* y, int N, int P4, int order, int reps) {
    const unsigned grid = (unsigned)((P4 / 256) * (N / 32));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_sp<32, MODE, POL>), dim3(grid), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, order);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_sp<32, MODE, POL>), dim3(grid), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, order);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / reps;
}

extern "C" float ustpol(const void* x, void* y, int N, int P4, int order, int mode, int pol, int reps) {
#define C_(P) if (pol == P) return mode == 2 ? run<2, P>(x, y, N, P4, order, reps) : run<0, P>(x, y, N, P4, order, reps);
    C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7)
    return -1.f;
}
