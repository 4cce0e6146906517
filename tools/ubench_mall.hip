// tools/ubench_mall.hip - can the SECOND read of x come out of the 256 MB Infinity Cache when statistics pass and Q/DQ
// pass run chunk by chunk inside ONE launch?  (development aid; nothing of the product links against it)
//
//   umall_read : read-only sweep of a buffer of S bytes, `reps` times inside one launch (what a cache-resident
//                re-read costs, S = 64 ... 1024 MB)
//   umall_two  : x[N][P] (P floats per sample plane) is cut into channel chunks of `cf` floats per sample (a chunk = N
//                strided runs of cf*4 bytes); persistent workgroups walk the sequence
//                    A(0), { A(k+1) interleaved with B(k) }, B(last)
//                where an A tile reads 16 KB and folds it to min / max (plain loads: they allocate in the caches) and a
//                B tile reads the same 16 KB again (non-temporal: last use), runs the Q/DQ arithmetic and stores 16 KB.
//                No dependency between the two is enforced (timing only).  cf = P: two full passes in one launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float qdq(float v, float sc, float zp, float qm) {
    float q = v / sc + zp;
    q = fminf(fmaxf(q, 0.f), qm);
    q = rintf(q);
    return (q - zp) * sc;
}

template <int NT>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ x, float* __restrict__ out, long long n4, int reps) {
    float m = -INFINITY;
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (int r = 0; r < reps; ++r)
        for (long long i = (long long)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
            f4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long k = i + j * 256 < n4 ? i + j * 256 : n4 - 1;
                v[j] = NT ? __builtin_nontemporal_load(x + k) : x[k];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
        }
    if (m > 1e30f) out[0] = m;
}

struct Two {
    const f4* x;
    f4* y;
    float* out;
    int N;
    long long P4;     // float4 per sample plane
    int c4;           // float4 per run (chunk width per sample), a multiple of 1024
    int nchunks;
    int tpr;          // tiles (1024 float4 = 16 KB) per run
    long long tpc;    // tiles per chunk = N * tpr
    float sc, zp, qm;
    int order;        // 0: samples outer (runs one after the other), 1: tiles of a run spread over consecutive samples
};

template <int ANT>
__global__ void __launch_bounds__(256) k_two(const Two a) {
    const long long steps = (long long)(a.nchunks + 1) * 2 * a.tpc;
    float m = -INFINITY;
    for (long long i = blockIdx.x; i < steps; i += gridDim.x) {
        const long long s = i / (2 * a.tpc), r = i - s * 2 * a.tpc;
        const int kind = (int)(r & 1);
        const long long t = r >> 1;
        const long long chunk = kind ? s - 1 : s;
        if (chunk < 0 || chunk >= a.nchunks) continue;
        long long n, j;
        if (a.order == 0) { n = t / a.tpr; j = t - n * a.tpr; }
        else { j = t / a.N; n = t - j * a.N; }
        const long long base = n * a.P4 + chunk * a.c4 + j * 1024 + threadIdx.x;
        f4 v[4];
        if (kind == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ANT ? __builtin_nontemporal_load(a.x + base + q * 256) : a.x[base + q * 256];
#pragma unroll
            for (int q = 0; q < 4; ++q) m = fmaxf(m, fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w)));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = __builtin_nontemporal_load(a.x + base + q * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 o;
                o.x = qdq(v[q].x, a.sc, a.zp, a.qm); o.y = qdq(v[q].y, a.sc, a.zp, a.qm);
                o.z = qdq(v[q].z, a.sc, a.zp, a.qm); o.w = qdq(v[q].w, a.sc, a.zp, a.qm);
                __builtin_nontemporal_store(o, a.y + base + q * 256);
            }
        }
    }
    if (m > 1e30f) a.out[0] = m;
}

template <typename F>
static float timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}

extern "C" float umall_read(const void* x, void* out, long long bytes, int inner, int nt, int grid, int reps) {
    const long long n4 = bytes / 16;
    if (nt) return timeit([&] { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, (const f4*)x, (float*)out, n4, inner); }, reps);
    return timeit([&] { hipLaunchKernelGGL(k_read<0>, dim3(grid), dim3(256), 0, 0, (const f4*)x, (float*)out, n4, inner); }, reps);
}

// cf: floats per run (a multiple of 4096); P % cf == 0
extern "C" float umall_two(const void* x, void* y, void* out, int N, long long P, long long cf, int order, int a_nt, int grid,
                           int reps) {
    Two a;
    a.x = (const f4*)x; a.y = (f4*)y; a.out = (float*)out; a.N = N; a.P4 = P / 4; a.c4 = (int)(cf / 4);
    a.nchunks = (int)(P / cf); a.tpr = a.c4 / 1024; a.tpc = (long long)N * a.tpr; a.sc = 0.37f; a.zp = 7.f; a.qm = 15.f;
    a.order = order;
    if (a.tpr < 1 || P % cf) return -1.f;
    if (a_nt) return timeit([&] { hipLaunchKernelGGL(k_two<1>, dim3(grid), dim3(256), 0, 0, a); }, reps);
    return timeit([&] { hipLaunchKernelGGL(k_two<0>, dim3(grid), dim3(256), 0, 0, a); }, reps);
}
