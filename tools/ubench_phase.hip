// tools/ubench_phase.hip - round 4: do the co-resident workgroups of the sample-strided order hurt each other by writing the
// SAME column offsets of different samples at the same time (addresses that differ by multiples of the 3.2 MB sample stride:
// equal low 16 bits)?  Register tiles of 32 rows x 256 float4, row blocks fastest (the order a per-channel exchange forces);
// ROT 1: workgroup rb starts its 32 row stores at row 8 * (rb % 4) instead of 0 (four phases), ROT 2: the column block a lane
// stores first is staggered too (cb % 4).  MODE 0 copy, 2 write only.  (development aid)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R0, int R>
__device__ __forceinline__ void store_rot(const f4 (&v)[R], f4* y, int rb, int c, int N, int P4, float scale) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        constexpr int dummy = 0; (void)dummy;
        const int r = (i + R0) % R;
        const int n = rb * R + r;
        if (n < N && c < P4) __builtin_nontemporal_store(v[r] * scale, y + (size_t)n * P4 + c);
    }
}

template <int MODE, int ROT>
__global__ void __launch_bounds__(256, 3) k_phase(const f4* __restrict__ x, f4* __restrict__ y, int N, int P4, int nrb) {
    constexpr int R = 32;
    const int cb = (int)blockIdx.x / nrb, rb = (int)blockIdx.x % nrb;
    const int c = cb * 256 + (int)threadIdx.x;
    f4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = min(rb * R + r, N - 1);
        if (MODE == 2) { const float f = (float)(n + c) * 1e-3f; v[r] = f4{f, f + 1.f, f + 2.f, f + 3.f}; }
        else v[r] = __builtin_nontemporal_load(x + (size_t)n * P4 + min(c, P4 - 1));
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, fmaxf(fmaxf(v[r].x, v[r].y), fmaxf(v[r].z, v[r].w)));
    const float scale = mx > 1e30f ? 2.f : 1.0001f;
    const int ph = ROT == 0 ? 0 : (ROT == 1 ? (rb & 3) : ((rb + cb) & 3));
    switch (ph) {
        case 0: store_rot<0, R>(v, y, rb, c, N, P4, scale); break;
        case 1: store_rot<8, R>(v, y, rb, c, N, P4, scale); break;
        case 2: store_rot<16, R>(v, y, rb, c, N, P4, scale); break;
        default: store_rot<24, R>(v, y, rb, c, N, P4, scale); break;
    }
}

extern "C" float uphase(int mode, int rot, const void* x, void* y, int N, int P4, int reps) {
    const int nrb = (N + 31) / 32, ncb = (P4 + 255) / 256;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto go = [&] {
#define L_(M, T) if (mode == M && rot == T) hipLaunchKernelGGL((k_phase<M, T>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, nrb);
        L_(0, 0) L_(0, 1) L_(0, 2) L_(2, 0) L_(2, 1) L_(2, 2)
#undef L_
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
