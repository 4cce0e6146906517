// tools/ubench_mall3.hip - round 4: can the SECOND read of x come out of the 256 MB Infinity Cache when both passes are
// ONE-SHOT short workgroups dispatched in address order (the form that streams at 6.5 TB/s), not the persistent loop of
// ubench_mall.hip?  (development aid; nothing of the product links against it)
//
//   x[N][P4] float4.  A chunk = the same c4 float4 of every sample (N strided runs), cut into tiles of L*256 float4.
//   A tile: read, fold to a maximum, publish one partial + bump the chunk's counter (dep = 1).
//   B tile: (dep = 1: lane 0 polls the chunk's counter), read the same bytes again, Q/DQ, store.
//   um3_one : ONE launch.  blockIdx -> phase s = b / (2 tpc), kind = b & 1, tile t: A of chunk s / B of chunk s - lag.
//   um3_sep : separate launches per chunk: A(k + lag), B(k).
//   um3_pass: one full pass A or B over the tensor (the no-cache reference).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float qdq(float v, float sc, float zp, float qm) {
    float q = v / sc + zp;
    q = fminf(fmaxf(q, 0.f), qm);
    q = rintf(q);
    return (q - zp) * sc;
}

struct Par {
    const f4* x;
    f4* y;
    float* out;
    unsigned* done;   // [nchunks] tiles of pass A finished (uncached memory, zero before the launch)
    float* part;      // [nchunks * tpc] partial maxima
    int N;
    long long P4;     // float4 per sample
    int c4;           // float4 per run
    int nchunks, tpr, tpc, lag, dep;
    int chunk0;       // um3_sep: the chunk this launch works on
    float sc, zp, qm;
};

template <int L, int NT>
__device__ __forceinline__ void load_tile(const f4* p, f4 (&v)[L]) {
#pragma unroll
    for (int q = 0; q < L; ++q) v[q] = NT ? __builtin_nontemporal_load(p + q * 256) : p[q * 256];
}

template <int L, int ANT>
__device__ __forceinline__ void tile_a(const Par& a, int chunk, int t) {
    const int n = t / a.tpr, j = t - n * a.tpr;
    const long long base = (long long)n * a.P4 + (long long)chunk * a.c4 + (long long)j * (L * 256) + threadIdx.x;
    f4 v[L];
    load_tile<L, ANT>(a.x + base, v);
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < L; ++q) m = fmaxf(m, fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w)));
    if (a.dep) {
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        __shared__ float sm[4];
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
            __hip_atomic_store(a.part + (size_t)chunk * a.tpc + t, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);   // the partial has left before the counter moves
            __hip_atomic_fetch_add(a.done + chunk * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (m > 1e30f) a.out[0] = m;
}

template <int L, int LNT, int SNT>
__device__ __forceinline__ void tile_b(const Par& a, int chunk, int t) {
    const int n = t / a.tpr, j = t - n * a.tpr;
    const long long base = (long long)n * a.P4 + (long long)chunk * a.c4 + (long long)j * (L * 256) + threadIdx.x;
    float sc = a.sc;
    if (a.dep) {
        __shared__ float smx;
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(a.done + chunk * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.tpc && ++spins < (1 << 22))
                __builtin_amdgcn_s_sleep(8);
            smx = __hip_atomic_load(a.part + (size_t)chunk * a.tpc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        sc = a.sc + 1e-30f * smx;
    }
    f4 v[L];
    load_tile<L, LNT>(a.x + base, v);
#pragma unroll
    for (int q = 0; q < L; ++q) {
        f4 o;
        o.x = qdq(v[q].x, sc, a.zp, a.qm); o.y = qdq(v[q].y, sc, a.zp, a.qm);
        o.z = qdq(v[q].z, sc, a.zp, a.qm); o.w = qdq(v[q].w, sc, a.zp, a.qm);
        if (SNT) __builtin_nontemporal_store(o, a.y + base + q * 256);
        else a.y[base + q * 256] = o;
    }
}

// ONE launch: phases of 2 * tpc workgroups; even = A tile of chunk s, odd = B tile of chunk s - lag
template <int L, int ANT, int LNT, int SNT>
__global__ void __launch_bounds__(256) k_one(const Par a) {
    const unsigned b = blockIdx.x;
    const unsigned per = 2u * (unsigned)a.tpc;
    const int s = (int)(b / per);
    const unsigned r = b - (unsigned)s * per;
    const int t = (int)(r >> 1);
    if (r & 1) {
        const int chunk = s - a.lag;
        if (chunk < 0 || chunk >= a.nchunks) return;
        tile_b<L, LNT, SNT>(a, chunk, t);
    } else {
        if (s >= a.nchunks) return;
        tile_a<L, ANT>(a, s, t);
    }
}

template <int L, int ANT>
__global__ void __launch_bounds__(256) k_a(const Par a) { tile_a<L, ANT>(a, a.chunk0, (int)blockIdx.x); }
template <int L, int LNT, int SNT>
__global__ void __launch_bounds__(256) k_b(const Par a) { tile_b<L, LNT, SNT>(a, a.chunk0, (int)blockIdx.x); }

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

static int fill(Par& a, const void* x, void* y, void* out, void* done, void* part, int N, long long P, long long cf, int L, int lag,
                int dep) {
    a.x = (const f4*)x; a.y = (f4*)y; a.out = (float*)out; a.done = (unsigned*)done; a.part = (float*)part;
    a.N = N; a.P4 = P / 4; a.c4 = (int)(cf / 4);
    if (P % cf || a.c4 % (L * 256)) return -1;
    a.nchunks = (int)(P / cf); a.tpr = a.c4 / (L * 256); a.tpc = N * a.tpr; a.lag = lag; a.dep = dep; a.chunk0 = 0;
    a.sc = 0.37f; a.zp = 7.f; a.qm = 15.f;
    return 0;
}

#define DISPATCH_POL(KERN, L, grid, a)                                                                            \
    switch (pol) {                                                                                                \
    case 0: hipLaunchKernelGGL((KERN<L, 0, 0, 0>), dim3(grid), dim3(256), 0, 0, a); break;                        \
    case 1: hipLaunchKernelGGL((KERN<L, 0, 1, 1>), dim3(grid), dim3(256), 0, 0, a); break;                        \
    case 2: hipLaunchKernelGGL((KERN<L, 1, 1, 1>), dim3(grid), dim3(256), 0, 0, a); break;                        \
    case 3: hipLaunchKernelGGL((KERN<L, 0, 0, 1>), dim3(grid), dim3(256), 0, 0, a); break;                        \
    default: hipLaunchKernelGGL((KERN<L, 0, 1, 0>), dim3(grid), dim3(256), 0, 0, a); break;                       \
    }

// pol: 0 all plain; 1 A plain, B nt load + nt store; 2 everything nt; 3 A plain, B plain load + nt store; 4 A plain, B nt load + plain store
extern "C" float um3_one(const void* x, void* y, void* out, void* done, void* part, int N, long long P, long long cf, int L, int lag,
                         int dep, int pol, int reps) {
    Par a;
    if (fill(a, x, y, out, done, part, N, P, cf, L, lag, dep)) return -1.f;
    const long long grid = (long long)(a.nchunks + lag) * 2 * a.tpc;
    if (grid >= (1ll << 31)) return -1.f;
    auto go = [&] {
        if (dep) (void)hipMemsetAsync(done, 0, (size_t)a.nchunks * 256, 0);
        if (L == 4) { DISPATCH_POL(k_one, 4, (unsigned)grid, a) } else { DISPATCH_POL(k_one, 8, (unsigned)grid, a) }
    };
    return timeit(go, reps);
}

extern "C" float um3_sep(const void* x, void* y, void* out, void* done, void* part, int N, long long P, long long cf, int L, int lag,
                         int pol, int reps) {
    Par a;
    if (fill(a, x, y, out, done, part, N, P, cf, L, lag, 0) || L != 4) return -1.f;
    auto go = [&] {
        for (int s = 0; s < a.nchunks + lag; ++s) {
            Par p = a;
            if (s < a.nchunks) {
                p.chunk0 = s;
                if (pol == 2) hipLaunchKernelGGL((k_a<4, 1>), dim3(a.tpc), dim3(256), 0, 0, p);
                else hipLaunchKernelGGL((k_a<4, 0>), dim3(a.tpc), dim3(256), 0, 0, p);
            }
            if (s - lag >= 0) {
                p.chunk0 = s - lag;
                switch (pol) {
                case 0: hipLaunchKernelGGL((k_b<4, 0, 0>), dim3(a.tpc), dim3(256), 0, 0, p); break;
                case 3: hipLaunchKernelGGL((k_b<4, 0, 1>), dim3(a.tpc), dim3(256), 0, 0, p); break;
                case 4: hipLaunchKernelGGL((k_b<4, 1, 0>), dim3(a.tpc), dim3(256), 0, 0, p); break;
                default: hipLaunchKernelGGL((k_b<4, 1, 1>), dim3(a.tpc), dim3(256), 0, 0, p); break;
                }
            }
        }
    };
    return timeit(go, reps);
}

// kind 0: pass A alone over the whole tensor, 1: pass B alone (cf = P: one chunk)
extern "C" float um3_pass(const void* x, void* y, void* out, int N, long long P, int kind, int nt, int reps) {
    Par a;
    if (fill(a, x, y, out, nullptr, nullptr, N, P, P, 4, 0, 0)) return -1.f;
    auto go = [&] {
        if (kind == 0) {
            if (nt) hipLaunchKernelGGL((k_a<4, 1>), dim3(a.tpc), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL((k_a<4, 0>), dim3(a.tpc), dim3(256), 0, 0, a);
        } else {
            if (nt) hipLaunchKernelGGL((k_b<4, 1, 1>), dim3(a.tpc), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL((k_b<4, 0, 0>), dim3(a.tpc), dim3(256), 0, 0, a);
        }
    };
    return timeit(go, reps);
}

extern "C" int um3_alloc(size_t bytes, void** p) {
    hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(*p, 0, bytes);
}
