// tools/ubench_tile.hip - how fast can a workgroup-resident TILE of an NCHW tensor be copied (load all, then store
// all: the memory structure of the single-launch kernels), as a function of the tile shape?  (development aid)
//   tile = R rows (samples, stride P floats) x J*256 float4 columns (contiguous); blockIdx -> (column block, row block)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int J, int R, int ORDER, int DEP>
__global__ void __launch_bounds__(256) k_tile(const f4* __restrict__ x, f4* __restrict__ y, int N, int P4, int ncb) {
    // ORDER 0: consecutive blocks = consecutive column blocks of the same rows; 1: consecutive blocks = consecutive row
    // blocks of the same columns
    const int nrb = (N + R - 1) / R;
    const int cb = ORDER == 0 ? (int)blockIdx.x % ncb : (int)blockIdx.x / nrb;
    const int rb = ORDER == 0 ? (int)blockIdx.x / ncb : (int)blockIdx.x % nrb;
    f4 v[R][J];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = min(rb * R + r, N - 1);
            const int c = min((cb * J + j) * 256 + (int)threadIdx.x, P4 - 1);
            v[r][j] = __builtin_nontemporal_load(x + (size_t)n * P4 + c);
        }
    float scale = 1.0001f;
    if (DEP) {   // the single-launch kernels' dependency: no store before EVERY load of the tile has landed
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                mn = fminf(fminf(mn, v[r][j].x), fminf(v[r][j].y, fminf(v[r][j].z, v[r][j].w)));
                mx = fmaxf(fmaxf(mx, v[r][j].x), fmaxf(v[r][j].y, fmaxf(v[r][j].z, v[r][j].w)));
            }
        if (mx - mn > 1e30f) scale = 2.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = rb * R + r;
            const int c = (cb * J + j) * 256 + (int)threadIdx.x;
            if (n < N && c < P4) __builtin_nontemporal_store(v[r][j] * scale, y + (size_t)n * P4 + c);
        }
}

template <int J, int R, int ORDER, int DEP>
static float run(const void* x, void* y, int N, int P4, int reps) {
    const int ncb = (P4 + J * 256 - 1) / (J * 256), nrb = (N + R - 1) / R;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_tile<J, R, ORDER, DEP>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, ncb);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_tile<J, R, ORDER, DEP>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, ncb);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

extern "C" float utile(int J, int R, int order, int dep, const void* x, void* y, int N, int P4, int reps) {
#define CASE(j, r) if (J == j && R == r) return dep ? (order ? run<j, r, 1, 1>(x, y, N, P4, reps) : run<j, r, 0, 1>(x, y, N, P4, reps)) : (order ? run<j, r, 1, 0>(x, y, N, P4, reps) : run<j, r, 0, 0>(x, y, N, P4, reps));
    CASE(1, 32) CASE(1, 16) CASE(1, 8) CASE(1, 4) CASE(1, 1)
    CASE(2, 16) CASE(2, 8) CASE(2, 4) CASE(2, 1)
    CASE(4, 8) CASE(4, 4) CASE(4, 2) CASE(4, 1)
    CASE(8, 4) CASE(8, 2) CASE(8, 1) CASE(16, 2) CASE(16, 1)
    return -1.f;
}
