// tools/ubench_width.hip - round 4: does it cost to start a workgroup's 4 KB-ish row pieces off the 64 / 128-byte grid?
// (development aid)  The whole-channel tiles of the 14x14 layers are 5 channels = 245 float4 = 3920 bytes wide: every row piece
// begins and ends inside a 64-byte sector that the neighbouring workgroup writes the rest of.  Register tiles of 32 rows
// (samples) x w float4 (w lanes of 256 active), sample-strided order (row blocks fastest), non-temporal; MODE 0 copy, 2 write only.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int MODE>
__global__ void __launch_bounds__(256, 3) k_w(const f4* __restrict__ x, f4* __restrict__ y, int N, int P4, int w, int nrb) {
    const int cb = (int)blockIdx.x / nrb, rb = (int)blockIdx.x % nrb;
    const int t = (int)threadIdx.x;
    const bool on = t < w && cb * w + t < P4;
    const int c = on ? cb * w + t : cb * w;
    f4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = min(rb * R + r, N - 1);
        if (MODE == 2) { const float f = (float)(n + c) * 1e-3f; v[r] = f4{f, f + 1.f, f + 2.f, f + 3.f}; }
        else v[r] = __builtin_nontemporal_load(x + (size_t)n * P4 + c);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, fmaxf(fmaxf(v[r].x, v[r].y), fmaxf(v[r].z, v[r].w)));
    const float scale = mx > 1e30f ? 2.f : 1.0001f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = rb * R + r;
        if (on && n < N) __builtin_nontemporal_store(v[r] * scale, y + (size_t)n * P4 + c);
    }
}

extern "C" float uwidth(int mode, const void* x, void* y, int N, int P4, int w, int reps) {
    const int nrb = (N + 31) / 32, ncb = (P4 + w - 1) / w;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto go = [&] {
        if (mode == 0) hipLaunchKernelGGL((k_w<32, 0>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, w, nrb);
        else hipLaunchKernelGGL((k_w<32, 2>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, w, nrb);
    };
    go();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) go();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / reps;
}
