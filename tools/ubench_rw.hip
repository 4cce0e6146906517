// tools/ubench_rw.hip - round 4: the READ side and the WRITE side of the register-tile structure, separately.
// (development aid)   tile = R rows (samples, stride P floats) x J*256 float4 columns; MODE 0 copy, 1 read only, 2 write only
// ORDER 0: consecutive workgroups = consecutive column blocks of the same rows (address order); 1: consecutive row blocks
// of the same columns (the order a per-channel exchange forces).  NT: non-temporal accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int J, int R, int ORDER, int MODE, int NT>
__global__ void __launch_bounds__(256) k_rw(const f4* __restrict__ x, f4* __restrict__ y, float* __restrict__ out, int N, int P4, int ncb) {
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
            if (MODE == 2) { const float t = (float)(n + c) * 1e-3f; v[r][j] = f4{t, t + 1.f, t + 2.f, t + 3.f}; }
            else v[r][j] = NT ? __builtin_nontemporal_load(x + (size_t)n * P4 + c) : x[(size_t)n * P4 + c];
        }
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            mn = fminf(fminf(mn, v[r][j].x), fminf(v[r][j].y, fminf(v[r][j].z, v[r][j].w)));
            mx = fmaxf(fmaxf(mx, v[r][j].x), fmaxf(v[r][j].y, fmaxf(v[r][j].z, v[r][j].w)));
        }
    const float scale = (mx - mn > 1e30f) ? 2.f : 1.0001f;
    if (MODE == 1) {
        if (mx - mn > 1e30f) out[0] = mx;
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = rb * R + r;
            const int c = (cb * J + j) * 256 + (int)threadIdx.x;
            if (n < N && c < P4) {
                if (NT) __builtin_nontemporal_store(v[r][j] * scale, y + (size_t)n * P4 + c);
                else y[(size_t)n * P4 + c] = v[r][j] * scale;
            }
        }
}

template <int J, int R, int ORDER, int MODE, int NT>
static float run(const void* x, void* y, void* out, int N, int P4, int reps) {
    const int ncb = (P4 + J * 256 - 1) / (J * 256), nrb = (N + R - 1) / R;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_rw<J, R, ORDER, MODE, NT>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)x, (f4*)y, (float*)out, N, P4, ncb);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_rw<J, R, ORDER, MODE, NT>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)x, (f4*)y, (float*)out, N, P4, ncb);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

template <int J, int R>
static float pick(int order, int mode, int nt, const void* x, void* y, void* out, int N, int P4, int reps) {
#define P_(O, M, T) if (order == O && mode == M && nt == T) return run<J, R, O, M, T>(x, y, out, N, P4, reps);
    P_(0, 0, 0) P_(0, 0, 1) P_(0, 1, 0) P_(0, 1, 1) P_(0, 2, 0) P_(0, 2, 1)
    P_(1, 0, 0) P_(1, 0, 1) P_(1, 1, 0) P_(1, 1, 1) P_(1, 2, 0) P_(1, 2, 1)
#undef P_
    return -1.f;
}

extern "C" float urw(int J, int R, int order, int mode, int nt, const void* x, void* y, void* out, int N, int P4, int reps) {
#define CASE(j, r) if (J == j && R == r) return pick<j, r>(order, mode, nt, x, y, out, N, P4, reps);
    CASE(1, 32) CASE(1, 8) CASE(1, 4) CASE(1, 1) CASE(4, 8) CASE(4, 1) CASE(2, 16) CASE(8, 4) CASE(16, 2) CASE(32, 1)
    return -1.f;
}

// the same over `nb` buffer pairs round-robin (cold address translations: the rotated set exceeds what the TLBs reach)
template <int J, int R>
static float multi(int order, int mode, int nt, void* const* xs, void* const* ys, int nb, void* out, int N, int P4, int rounds) {
    const int ncb = (P4 + J * 256 - 1) / (J * 256), nrb = (N + R - 1) / R;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto go = [&](int i) {
#define M_(O, M, T) if (order == O && mode == M && nt == T) hipLaunchKernelGGL((k_rw<J, R, O, M, T>), dim3(ncb * nrb), dim3(256), 0, 0, (const f4*)xs[i], (f4*)ys[i], (float*)out, N, P4, ncb);
        M_(0, 0, 1) M_(1, 0, 1) M_(0, 1, 1) M_(1, 1, 1) M_(0, 2, 1) M_(1, 2, 1)
#undef M_
    };
    for (int i = 0; i < nb; ++i) go(i);
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < rounds; ++r)
        for (int i = 0; i < nb; ++i) go(i);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / (rounds * nb);
}

extern "C" float urw_multi(int J, int R, int order, int mode, void* const* xs, void* const* ys, int nb, void* out, int N, int P4, int rounds) {
    if (J == 1 && R == 1) return multi<1, 1>(order, mode, 1, xs, ys, nb, out, N, P4, rounds);
    if (J == 1 && R == 32) return multi<1, 32>(order, mode, 1, xs, ys, nb, out, N, P4, rounds);
    return -1.f;
}

extern "C" int urw_alloc(size_t bytes, int flags, void** p) {
    if (flags < 0) return (int)hipMalloc(p, bytes);
    return (int)hipExtMallocWithFlags(p, bytes, (unsigned)flags);
}
extern "C" int urw_free(void* p) { return (int)hipFree(p); }
extern "C" int urw_fill(void* p, size_t bytes) { return (int)hipMemset(p, 0x3c, bytes); }
