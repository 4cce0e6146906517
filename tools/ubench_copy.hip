// tools/ubench_copy.hip - what is the best read+write streaming structure on MI355X? (development aid)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// each thread moves B float4 per iteration: loads first, then stores; CONTIG=1: a wave's B loads are
// adjacent 1 KB rows (wave owns a B KB chunk); CONTIG=0: loads are a whole-grid stride apart
template <int TPB, int B, int NT, int CONTIG>
__global__ void __launch_bounds__(TPB) k_copy(const f4* __restrict__ x, f4* __restrict__ y, size_t n4) {
    const size_t nthreads = (size_t)gridDim.x * TPB;
    if (CONTIG) {
        const size_t wave = ((size_t)blockIdx.x * TPB + threadIdx.x) >> 6, lane = threadIdx.x & 63;
        const size_t nwaves = nthreads >> 6;
        for (size_t base = wave * 64 * B; base < n4; base += nwaves * 64 * B) {
            f4 v[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const size_t i = base + b * 64 + lane;
                if (i < n4) v[b] = (NT & 1) ? __builtin_nontemporal_load(x + i) : x[i];
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const size_t i = base + b * 64 + lane;
                if (i < n4) { if (NT & 2) __builtin_nontemporal_store(v[b], y + i); else y[i] = v[b]; }
            }
        }
    } else {
        for (size_t i0 = (size_t)blockIdx.x * TPB + threadIdx.x; i0 < n4; i0 += nthreads * B) {
            f4 v[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const size_t i = i0 + b * nthreads;
                if (i < n4) v[b] = (NT & 1) ? __builtin_nontemporal_load(x + i) : x[i];
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const size_t i = i0 + b * nthreads;
                if (i < n4) { if (NT & 2) __builtin_nontemporal_store(v[b], y + i); else y[i] = v[b]; }
            }
        }
    }
}

template <int TPB, int B, int NT, int CONTIG>
static float run(const float* x, float* y, size_t n4, int grid, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_copy<TPB, B, NT, CONTIG>), dim3(grid), dim3(TPB), 0, 0, (const f4*)x, (f4*)y, n4);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_copy<TPB, B, NT, CONTIG>), dim3(grid), dim3(TPB), 0, 0, (const f4*)x, (f4*)y, n4);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

extern "C" float ucopy(int tpb, int B, int nt, int contig, const float* x, float* y, size_t n4, int grid, int reps) {
#define CASE(T, BB, N, C) if (tpb == T && B == BB && nt == N && contig == C) return run<T, BB, N, C>(x, y, n4, grid, reps);
#define ALLNT(T, BB, C) CASE(T, BB, 0, C) CASE(T, BB, 2, C) CASE(T, BB, 3, C)
#define ALLB(T, C) ALLNT(T, 1, C) ALLNT(T, 2, C) ALLNT(T, 4, C) ALLNT(T, 8, C)
    ALLB(256, 0) ALLB(256, 1) ALLB(512, 0) ALLB(512, 1) ALLB(1024, 0) ALLB(1024, 1)
    return -1.f;
}
extern "C" float umemcpy(const float* x, float* y, size_t bytes, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

// read-only streaming: one-shot (each thread B float4) vs grid-stride; result folded to defeat DCE
template <int TPB, int B, int NT>
__global__ void __launch_bounds__(TPB) k_read(const f4* __restrict__ x, float* __restrict__ out, size_t n4) {
    const size_t nthreads = (size_t)gridDim.x * TPB;
    float m = -1e30f;
    for (size_t i0 = (size_t)blockIdx.x * TPB * B + threadIdx.x; i0 < n4; i0 += nthreads * B) {
        f4 v[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const size_t i = i0 + (size_t)b * TPB;
            v[b] = (i < n4) ? ((NT & 1) ? __builtin_nontemporal_load(x + i) : x[i]) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int b = 0; b < B; ++b) m = fmaxf(m, fmaxf(fmaxf(v[b].x, v[b].y), fmaxf(v[b].z, v[b].w)));
    }
    if (m == 12345.678f) out[blockIdx.x] = m;
}
extern "C" float uread(int B, int nt, const float* x, float* out, size_t n4, int grid, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() {
        if (B == 1 && nt == 0) hipLaunchKernelGGL((k_read<256, 1, 0>), dim3(grid), dim3(256), 0, 0, (const f4*)x, out, n4);
        if (B == 4 && nt == 0) hipLaunchKernelGGL((k_read<256, 4, 0>), dim3(grid), dim3(256), 0, 0, (const f4*)x, out, n4);
        if (B == 8 && nt == 0) hipLaunchKernelGGL((k_read<256, 8, 0>), dim3(grid), dim3(256), 0, 0, (const f4*)x, out, n4);
        if (B == 1 && nt == 1) hipLaunchKernelGGL((k_read<256, 1, 1>), dim3(grid), dim3(256), 0, 0, (const f4*)x, out, n4);
        if (B == 4 && nt == 1) hipLaunchKernelGGL((k_read<256, 4, 1>), dim3(grid), dim3(256), 0, 0, (const f4*)x, out, n4);
        if (B == 8 && nt == 1) hipLaunchKernelGGL((k_read<256, 8, 1>), dim3(grid), dim3(256), 0, 0, (const f4*)x, out, n4);
    };
    go(); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
