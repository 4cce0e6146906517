// tools/ubench_policy.hip - gfx950 cache-policy bits (sc0 / sc1 / nt) on a one-shot float4 copy and a
// read-only pass (development aid).  Policies are spelled in inline asm because the compiler only
// exposes plain and `nt`.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

#define LOADP(NAME, POL)                                                                   \
    __device__ __forceinline__ f4 NAME(const f4* p) {                                      \
        f4 v;                                                                              \
        asm volatile("global_load_dwordx4 %0, %1, off " POL "\n\ts_waitcnt vmcnt(0)"      \
                     : "=v"(v) : "v"(p) : "memory");                                       \
        return v;                                                                          \
    }
#define STOREP(NAME, POL)                                                                  \
    __device__ __forceinline__ void NAME(f4* p, f4 v) {                                    \
        asm volatile("global_store_dwordx4 %0, %1, off " POL :: "v"(p), "v"(v) : "memory"); \
    }
LOADP(ld0, "") LOADP(ld1, "nt") LOADP(ld2, "sc1") LOADP(ld3, "sc0 sc1") LOADP(ld4, "sc1 nt") LOADP(ld5, "sc0 sc1 nt")
LOADP(ld6, "sc0") LOADP(ld7, "sc0 nt")
STOREP(st0, "") STOREP(st1, "nt") STOREP(st2, "sc1") STOREP(st3, "sc0 sc1") STOREP(st4, "sc1 nt") STOREP(st5, "sc0 sc1 nt")
STOREP(st6, "sc0") STOREP(st7, "sc0 nt")

template <int LP, int SP>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ x, f4* __restrict__ y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f4 v;
    switch (LP) { case 0: v = ld0(x + i); break; case 1: v = ld1(x + i); break; case 2: v = ld2(x + i); break;
                  case 3: v = ld3(x + i); break; case 4: v = ld4(x + i); break; case 5: v = ld5(x + i); break;
                  case 6: v = ld6(x + i); break; default: v = ld7(x + i); }
    switch (SP) { case 0: st0(y + i, v); break; case 1: st1(y + i, v); break; case 2: st2(y + i, v); break;
                  case 3: st3(y + i, v); break; case 4: st4(y + i, v); break; case 5: st5(y + i, v); break;
                  case 6: st6(y + i, v); break; default: st7(y + i, v); }
}

typedef int i4 __attribute__((ext_vector_type(4)));
// the same one-shot copy with compiler-visible accesses: nt load + raw buffer store carrying the policy bits
// (gfx940+: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1); one resource per workgroup
template <int AUX>
__global__ void __launch_bounds__(256) k_copy_buf(const f4* __restrict__ x, float* __restrict__ y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f4 v = __builtin_nontemporal_load(x + i);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)blockIdx.x * 1024, 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), r, threadIdx.x * 16, 0, AUX);
}

template <int LP>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ x, float* __restrict__ out, size_t n4) {
    // 8 float4 per thread, one-shot blocks of 32 KB
    const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    float m = -1e30f;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const size_t i = base + (size_t)b * 256;
        if (i < n4) {
            f4 v;
            switch (LP) { case 0: v = ld0(x + i); break; case 1: v = ld1(x + i); break; case 2: v = ld2(x + i); break;
                          case 3: v = ld3(x + i); break; case 4: v = ld4(x + i); break; case 5: v = ld5(x + i); break;
                          case 6: v = ld6(x + i); break; default: v = ld7(x + i); }
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    }
    if (m == 12345.678f) out[0] = m;
}

template <typename F>
static float timeit(F launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

extern "C" float pcopy(int lp, int sp, const float* x, float* y, size_t n4, int reps) {
    const unsigned grid = (unsigned)((n4 + 255) / 256);
#define C2(L, S) if (lp == L && sp == S) return timeit([&] { hipLaunchKernelGGL((k_copy<L, S>), dim3(grid), dim3(256), 0, 0, (const f4*)x, (f4*)y, n4); }, reps);
#define C1(L) C2(L, 0) C2(L, 1) C2(L, 2) C2(L, 3) C2(L, 4) C2(L, 5) C2(L, 6) C2(L, 7)
    C1(0) C1(1) C1(2) C1(3) C1(4) C1(5) C1(6) C1(7)
    return -1.f;
}
extern "C" float pcopy_buf(int aux, const float* x, float* y, size_t n4, int reps) {
    const unsigned grid = (unsigned)((n4 + 255) / 256);
#define B1(A) if (aux == A) return timeit([&] { hipLaunchKernelGGL((k_copy_buf<A>), dim3(grid), dim3(256), 0, 0, (const f4*)x, y, n4); }, reps);
    B1(0) B1(2) B1(0x10) B1(0x11) B1(0x12) B1(0x13)
    return -1.f;
}
extern "C" float pread(int lp, const float* x, float* out, size_t n4, int reps) {
    const unsigned grid = (unsigned)((n4 + 2047) / 2048);
#define R1(L) if (lp == L) return timeit([&] { hipLaunchKernelGGL((k_read<L>), dim3(grid), dim3(256), 0, 0, (const f4*)x, out, n4); }, reps);
    R1(0) R1(1) R1(2) R1(3) R1(4) R1(5) R1(6) R1(7)
    return -1.f;
}
