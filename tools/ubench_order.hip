// tools/ubench_order.hip - round 4: which WORKGROUP ORDER and ROW OWNERSHIP inside a co-resident block stores fastest?
// (development aid)  x[N][P4] float4; a workgroup's tile = R rows x 256 float4 (4 KB per row); a "block" = G adjacent
// column blocks (the channels that are resident together) x all N rows = G * N / R workgroups.
//   order 0: inside a block the column block is fastest (consecutive workgroups write adjacent 4 KB of the same rows);
//   order 1: the row block is fastest (consecutive workgroups = the members of one column block: what k_mmq_group does)
//   rows  0: a workgroup owns R consecutive rows; 1: rows s, s + S, s + 2S ... (interleaved)
//   mode  0: copy (load all, then store all), 2: write only
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int MODE, int NT>
__global__ void __launch_bounds__(256, 3) k_ord(const f4* __restrict__ x, f4* __restrict__ y, int N, int P4, int G, int order, int rows) {
    const int S = N / R;
    const int per = G * S;
    const int blk = (int)blockIdx.x / per, r = (int)blockIdx.x - blk * per;
    const int q = order == 0 ? r % G : r / S;
    const int s = order == 0 ? r / G : r % S;
    const size_t col = (size_t)(blk * G + q) * 256 + threadIdx.x;
    const int n0 = rows ? s : s * R, dn = rows ? S : 1;
    f4 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const size_t e = (size_t)(n0 + j * dn) * P4 + col;
        if (MODE == 2) { const float t = (float)(int)(e & 1023) * 1e-3f; v[j] = f4{t, t + 1.f, t + 2.f, t + 3.f}; }
        else v[j] = __builtin_nontemporal_load(x + e);
    }
    __builtin_amdgcn_sched_barrier(0);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < R; ++j) mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
    const float scale = mx > 1e30f ? 2.f : 1.0001f;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const size_t e = (size_t)(n0 + j * dn) * P4 + col;
        if (NT) __builtin_nontemporal_store(v[j] * scale, y + e);
        else y[e] = v[j] * scale;
    }
}

extern "C" float uord(const void* x, void* y, int N, int P4, int G, int order, int rows, int mode, int nt, int reps) {
    const int R = 32;
    const unsigned grid = (unsigned)((P4 / 256) * (N / R));
    auto go = [&] {
#define L_(M, T) hipLaunchKernelGGL((k_ord<32, M, T>), dim3(grid), dim3(256), 0, 0, (const f4*)x, (f4*)y, N, P4, G, order, rows)
        if (mode == 0) { if (nt) L_(0, 1); else L_(0, 0); } else { if (nt) L_(2, 1); else L_(2, 0); }
#undef L_
    };
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
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
