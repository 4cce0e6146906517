// tools/ubench_pipe.hip - which WORKGROUP STRUCTURE streams a register-resident tile fastest when a dependent round
// trip (the in-launch exchange of the per-channel extrema) sits between a tile's loads and its stores?
// (development aid for cnnq_group.hip.h / cnnq_pipe.hip.h; nothing of the product links against it)
//
//   rows  : the round-2 structure - one workgroup per tile, tile = K samples x one <= 256-lane piece of a channel row
//           (strided pieces), load all -> [delay] -> ALU + store all
//   flat  : one workgroup per tile, tile = 256*K consecutive float4 of the channel's flattened [N][H*W/4] space
//           (every lane busy, contiguous runs of a whole channel row)
//   pipe  : persistent workgroups, flat tiles of KH loads per lane, two register buffers: the loads of tile i+1 are
//           issued BEFORE the delay of tile i, so a workgroup always has something in flight
// delay = ticks of the 100 MHz clock one lane spends polling while the others sit at the barrier; alu = 1 runs the
// real Q/DQ arithmetic (IEEE divide, clamp, rint) instead of one multiply.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
    const float* x;
    float* y;
    int N, C, cpc;       // samples, channels, float4 per channel row
    long long P4;        // float4 per sample plane
    int ntiles, Tc;      // tiles in the tensor, tiles per channel
    long long total;     // N * cpc: float4 per channel
    long long delay;
    float sc, zp, qm;
    int alu;
    int pk;              // 1: the packed regime - one float4 stored per 8 loaded (the arithmetic of all 8 feeds it)
    int w, nb, S;        // rows structure: lanes per piece, pieces per channel row, batch splits
};

__device__ __forceinline__ float qdq(float v, float sc, float zp, float qm) {
    float q = v / sc + zp;
    q = fminf(fmaxf(q, 0.f), qm);
    q = rintf(q);
    return (q - zp) * sc;
}

__device__ __forceinline__ void spin(long long ticks) {
    if (ticks < 0) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}

// flat tile: lane t's j-th float4 is element f = m*256*K + j*256 + t of channel c's flattened [N][cpc] space
template <int K>
struct Flat {
    int n, col, q256, r256;
    long long base;
    // branch-free: a conditional around a load makes the compiler wait for the previous load first
    __device__ __forceinline__ void init(const Args& a, int tile) {
        const int c = tile / a.Tc, m = tile - c * a.Tc;
        const long long f = (long long)m * 256 * K + threadIdx.x;
        n = (int)(f / a.cpc);
        col = (int)(f - (long long)n * a.cpc);
        q256 = 256 / a.cpc;
        r256 = 256 - q256 * a.cpc;
        base = (long long)c * a.cpc;
    }
    __device__ __forceinline__ bool valid(const Args& a) const { return n < a.N; }
    __device__ __forceinline__ long long at(const Args& a) const {
        const int nn = n < a.N ? n : a.N - 1;      // past the end: re-read the last sample's element
        return (long long)nn * a.P4 + base + col;
    }
    __device__ __forceinline__ void step(const Args& a) {
        col += r256;
        n += q256;
        const bool wrap = col >= a.cpc;
        col -= wrap ? a.cpc : 0;
        n += wrap ? 1 : 0;
    }
};

template <int K>
__device__ __forceinline__ void load_flat(const Args& a, int tile, f4 (&v)[K]) {
    Flat<K> it;
    it.init(a, tile);
    const f4* x4 = reinterpret_cast<const f4*>(a.x);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        v[j] = __builtin_nontemporal_load(x4 + it.at(a));
        it.step(a);
    }
}

template <int K>
__device__ __forceinline__ float fold(const f4 (&v)[K]) {
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        mn = fminf(fminf(mn, v[j].x), fminf(v[j].y, fminf(v[j].z, v[j].w)));
        mx = fmaxf(fmaxf(mx, v[j].x), fmaxf(v[j].y, fmaxf(v[j].z, v[j].w)));
    }
    return (mx - mn) > 1e30f ? 1.f : 0.f;   // a dependence on every load that is (almost) never 1
}

template <int K, int ALU>
__device__ __forceinline__ void store_flat(const Args& a, int tile, const f4 (&v)[K], float dep) {
    Flat<K> it;
    it.init(a, tile);
    f4* y4 = reinterpret_cast<f4*>(a.y);
    const float sc = a.sc + dep, zp = a.zp, qm = a.qm;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < K; ++j) {
        f4 o;
        if (ALU) { o.x = qdq(v[j].x, sc, zp, qm); o.y = qdq(v[j].y, sc, zp, qm); o.z = qdq(v[j].z, sc, zp, qm); o.w = qdq(v[j].w, sc, zp, qm); }
        else o = v[j] * (1.0001f + dep);
        if (a.pk) {
            acc = (j & 7) ? acc + o : o;
            if ((j & 7) == 7 && it.valid(a)) __builtin_nontemporal_store(acc, y4 + it.at(a));
        } else if (it.valid(a)) __builtin_nontemporal_store(o, y4 + it.at(a));
        it.step(a);
    }
}

template <int K, int OCC, int ALU>
__global__ void __launch_bounds__(256, OCC) k_flat(const Args a) {
    f4 v[K];
    load_flat<K>(a, blockIdx.x, v);
    const float d = fold<K>(v);
    spin(a.delay + (long long)d);   // depends on the fold: the delay starts when the tile has landed
    store_flat<K, ALU>(a, blockIdx.x, v, d);
}

template <int K, int OCC, int ALU>
__global__ void __launch_bounds__(256, OCC) k_rows(const Args a) {
    // tile id -> (channel, piece, batch split); members of a channel consecutive, as rblk_of() orders them
    const int per_c = a.nb * a.S;
    const int c = blockIdx.x / per_c, r = blockIdx.x - c * per_c;
    const int s = r / a.nb, bb = r - s * a.nb;
    const int n0 = s * K;
    const bool ok = (int)threadIdx.x < a.w && bb * a.w + (int)threadIdx.x < a.cpc;
    const long long col = (long long)c * a.cpc + bb * a.w + (ok ? threadIdx.x : 0);
    const f4* x4 = reinterpret_cast<const f4*>(a.x);
    f4* y4 = reinterpret_cast<f4*>(a.y);
    f4 v[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int n = n0 + j < a.N ? n0 + j : a.N - 1;
        v[j] = __builtin_nontemporal_load(x4 + (long long)n * a.P4 + col);
    }
    const float d = fold<K>(v);
    spin(a.delay + (long long)d);   // depends on the fold: the delay starts when the tile has landed
    const float sc = a.sc + d, zp = a.zp, qm = a.qm;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        f4 o;
        if (ALU) { o.x = qdq(v[j].x, sc, zp, qm); o.y = qdq(v[j].y, sc, zp, qm); o.z = qdq(v[j].z, sc, zp, qm); o.w = qdq(v[j].w, sc, zp, qm); }
        else o = v[j] * (1.0001f + d);
        if (ok && n0 + j < a.N) __builtin_nontemporal_store(o, y4 + (long long)(n0 + j) * a.P4 + col);
    }
}

template <int KH, int OCC, int ALU>
__global__ void __launch_bounds__(256, OCC) k_pipe(const Args a) {
    f4 A[KH], B[KH];
    const int G = gridDim.x;
    const int last = a.ntiles - 1;
    int tile = blockIdx.x;      // grid <= ntiles
    load_flat<KH>(a, tile, A);
    for (;;) {
        // the prefetch is UNCONDITIONAL (past the end it re-reads the last tile): a conditional issue makes the
        // compiler's waitcnt insertion assume the worst at the join and wait for the prefetch too
        int nt = tile + G;
        float d = fold<KH>(A);
        load_flat<KH>(a, nt < last ? nt : last, B);
        spin(a.delay + (long long)d);
        store_flat<KH, ALU>(a, tile, A, d);
        tile = nt;
        if (tile > last) break;
        nt = tile + G;
        d = fold<KH>(B);
        load_flat<KH>(a, nt < last ? nt : last, A);
        spin(a.delay + (long long)d);
        store_flat<KH, ALU>(a, tile, B, d);
        tile = nt;
        if (tile > last) break;
    }
}

template <typename F>
static float timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

// kind 0 rows, 1 flat, 2 pipe; K loads per lane and tile; occ = workgroups per CU the kernel is compiled for
static int g_pk = 0;
extern "C" void upipe_set_pk(int pk) { g_pk = pk; }

extern "C" float upipe(int kind, int K, int occ, const void* x, void* y, int N, int C, int HW, int delay_ticks, int alu,
                       int reps) {
    Args a;
    a.x = (const float*)x; a.y = (float*)y; a.N = N; a.C = C; a.cpc = HW / 4; a.P4 = (long long)C * a.cpc;
    a.total = (long long)N * a.cpc; a.delay = delay_ticks; a.sc = 0.37f; a.zp = 7.f; a.qm = 15.f; a.alu = alu; a.pk = g_pk;
    a.Tc = (int)((a.total + 256LL * K - 1) / (256LL * K));
    a.ntiles = a.Tc * C;
    a.nb = (a.cpc + 255) / 256; a.w = (a.cpc + a.nb - 1) / a.nb; a.S = (N + K - 1) / K;
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
#define RUN(KERNEL, GRID) return timeit([&] { hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(256), 0, 0, a); }, reps)
#define RUN2(KN, K_, OCC_, GRID) do { if (alu) RUN((KN<K_, OCC_, 1>), GRID); else RUN((KN<K_, OCC_, 0>), GRID); } while (0)
    if (kind == 0) {
        const int grid = C * a.nb * a.S;
        if (K == 32 && occ == 3) RUN2(k_rows, 32, 3, grid);
        if (K == 16 && occ == 6) RUN2(k_rows, 16, 6, grid);
    } else if (kind == 1) {
        if (K == 32 && occ == 3) RUN2(k_flat, 32, 3, a.ntiles);
        if (K == 16 && occ == 6) RUN2(k_flat, 16, 6, a.ntiles);
        if (K == 8 && occ == 8) RUN2(k_flat, 8, 8, a.ntiles);
    } else {
        const int grid = cus * occ < a.ntiles ? cus * occ : a.ntiles;
        if (K == 16 && occ == 3) RUN2(k_pipe, 16, 3, grid);
        if (K == 20 && occ == 3) RUN2(k_pipe, 20, 3, grid);
        if (K == 16 && occ == 2) RUN2(k_pipe, 16, 2, grid);
        if (K == 8 && occ == 6) RUN2(k_pipe, 8, 6, grid);
        if (K == 8 && occ == 4) RUN2(k_pipe, 8, 4, grid);
        if (K == 24 && occ == 2) RUN2(k_pipe, 24, 2, grid);
        if (K == 4 && occ == 8) RUN2(k_pipe, 4, 8, grid);
    }
    return -1.f;
}
