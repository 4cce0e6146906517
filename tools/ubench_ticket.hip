// tools/ubench_ticket.hip - round 4: do the STORES of the register-tile kernels speed up when the workgroups release them in
// dispatch (address-sweep) order through a ticket counter instead of all at once after their channel's meeting?
// (development aid)   geometry of k_mmq_flat: a tile = 8192 consecutive float4 of one channel's flattened [N][cpc] space,
// dispatch in blocks of cb adjacent channels (channel fastest).
//   mode bit 0: load the tile (else synthetic values); bit 1: meet the channel's other members (counter) before storing;
//   W > 0: a workgroup stores only once ticket >= blockIdx - W; every workgroup bumps the ticket after issuing its stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

struct TP {
    const f4* x; f4* y; float* out;
    unsigned* cnt;      // [C] channel arrival counters (stride 64 words), then the ticket at cnt[C * 64]
    int N, C, cpc, Gs, cb, W, mode, nt;
};

template <int K>
__global__ void __launch_bounds__(256, 3) k_tick(const TP a) {
    const int tid = threadIdx.x;
    const int per = a.cb * a.Gs, blk = (int)blockIdx.x / per, r = (int)blockIdx.x - blk * per;
    const int c0 = blk * a.cb, cbl = min(a.cb, a.C - c0);
    const int member = r / cbl, c = c0 + (r - member * cbl);
    const long long total = (long long)a.N * a.cpc;
    f4 v[K];
    long long adr[1];
    (void)adr;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        long long f = (long long)member * (256 * K) + j * 256 + tid;
        if (f >= total) f = total - 1;
        const long long n = f / a.cpc, col = f - n * a.cpc;
        const long long e = (n * a.C + c) * a.cpc + col;
        if (a.mode & 1) v[j] = __builtin_nontemporal_load(a.x + e);
        else { const float t = (float)(int)(e & 1023) * 1e-3f; v[j] = f4{t, t + 1.f, t + 2.f, t + 3.f}; }
    }
    __builtin_amdgcn_sched_barrier(0);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < K; ++j) mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __shared__ float sm[4];
    if ((tid & 63) == 0) sm[tid >> 6] = mx;
    __syncthreads();
    unsigned* ticket = a.cnt + (size_t)a.C * 64;
    if (tid == 0) {
        if (a.mode & 2) {
            unsigned* cw = a.cnt + (size_t)c * 64;
            __hip_atomic_fetch_add(cw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.Gs && ++spins < (1 << 15))
                __builtin_amdgcn_s_sleep(32);
        }
        if (a.W > 0) {
            const long long need = (long long)blockIdx.x - a.W;
            int spins = 0;
            while (need > 0 && ++spins < (1 << 15)) {
                const long long have = (long long)__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (have >= need) break;
                // about 40 ns per ticket at full write rate: sleep in proportion to the distance (units of 64 clocks ~ 30 ns)
                long long d = need - have;
                d = d > 100 ? 100 : d;
                for (long long q = 0; q < d; ++q) __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    const float scale = (fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])) > 1e30f) ? 2.f : 1.0001f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const long long f = (long long)member * (256 * K) + j * 256 + tid;
        if (f < total) {
            const long long n = f / a.cpc, col = f - n * a.cpc;
            const long long e = (n * a.C + c) * a.cpc + col;
            if (a.nt) __builtin_nontemporal_store(v[j] * scale, a.y + e);
            else a.y[e] = v[j] * scale;
        }
    }
    if (tid == 0) __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

extern "C" float utick(const void* x, void* y, void* out, void* cnt, int N, int C, int HW, int cb, int W, int mode, int nt, int reps) {
    TP a;
    a.x = (const f4*)x; a.y = (f4*)y; a.out = (float*)out; a.cnt = (unsigned*)cnt;
    a.N = N; a.C = C; a.cpc = HW / 4; a.W = W; a.mode = mode; a.nt = nt;
    const long long total = (long long)N * a.cpc;
    a.Gs = (int)((total + 8191) / 8192);
    a.cb = cb;
    const unsigned grid = (unsigned)(C * a.Gs);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float tot = 0;
    for (int i = 0; i <= reps; ++i) {
        (void)hipMemsetAsync(cnt, 0, ((size_t)C * 64 + 64) * 4, 0);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_tick<32>), dim3(grid), dim3(256), 0, 0, a);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (i) tot += ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return tot / reps;
}

extern "C" int utick_alloc(size_t bytes, void** p) {
    hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(*p, 0, bytes);
}
