// tools/ubench_qdq.hip - development micro-benchmark (NOT part of the product library).
// Variants of the streaming Q/DQ loop to find what bounds it on MI355X.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int TPB = 256;
typedef float f4 __attribute__((ext_vector_type(4)));

enum { M_COPY = 0, M_DIV = 1, M_RCP = 2 };

template <int MATH>
__device__ __forceinline__ float op(float x, float sc, float rs, float zp, float qm) {
    if constexpr (MATH == M_COPY) return x;
    float q = (MATH == M_DIV) ? x / sc : x * rs;
    q = q + zp;
    q = (q > qm) ? qm : q;
    q = (q < 0.f) ? 0.f : q;
    q = rintf(q);
    return (q - zp) * sc;
}

// column-walk structure as in k_qdq: block owns `w` f4 columns of one channel, walks n0..n1
template <int MATH, int J, int NT, int UNR>
__global__ void __launch_bounds__(TPB) k_colwalk(const float* __restrict__ x, float* __restrict__ y, int N, int C,
                                                 int HW, int nb, int w, int S, const float* __restrict__ qp) {
    const int ncb = C * nb;
    const int cb = blockIdx.x % ncb, s = blockIdx.x / ncb;
    const int n0 = (int)(((int64_t)s * N) / S), n1 = (int)(((int64_t)(s + 1) * N) / S);
    const int cpc = HW / 4, c = cb / nb, bb = cb - c * nb;
    const int col0 = c * cpc + bb * w, col1 = min(col0 + w, (c + 1) * cpc);
    const float sc = qp[c], zp = qp[C + c], qm = qp[2 * C + c];
    const float rs = 1.0f / sc;
    int col[J];
    bool ok[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cc = col0 + j * TPB + threadIdx.x;
        ok[j] = cc < col1;
        col[j] = ok[j] ? cc : col0;
    }
    const size_t P = (size_t)C * HW;
    size_t off = (size_t)n0 * P;
#pragma unroll UNR
    for (int n = n0; n < n1; ++n, off += P) {
        f4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f4* p = reinterpret_cast<const f4*>(x + off + (size_t)col[j] * 4);
            if constexpr (NT & 1) v[j] = __builtin_nontemporal_load(p); else v[j] = *p;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f4 o;
            o.x = op<MATH>(v[j].x, sc, rs, zp, qm);
            o.y = op<MATH>(v[j].y, sc, rs, zp, qm);
            o.z = op<MATH>(v[j].z, sc, rs, zp, qm);
            o.w = op<MATH>(v[j].w, sc, rs, zp, qm);
            if (ok[j]) {
                f4* p = reinterpret_cast<f4*>(y + off + (size_t)col[j] * 4);
                if constexpr (NT & 2) __builtin_nontemporal_store(o, p); else *p = o;
            }
        }
    }
}

// classic flat grid-stride f4 stream (reference ceiling); channel = (i / cpc) % C
template <int MATH, int NT>
__global__ void __launch_bounds__(TPB) k_flat(const float* __restrict__ x, float* __restrict__ y, size_t n4, int C,
                                              int cpc, const float* __restrict__ qp) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i / cpc) % C);
        const float sc = qp[c], zp = qp[C + c], qm = qp[2 * C + c];
        const float rs = 1.0f / sc;
        const f4* p = reinterpret_cast<const f4*>(x) + i;
        f4 v;
        if constexpr (NT & 1) v = __builtin_nontemporal_load(p); else v = *p;
        f4 o;
        o.x = op<MATH>(v.x, sc, rs, zp, qm);
        o.y = op<MATH>(v.y, sc, rs, zp, qm);
        o.z = op<MATH>(v.z, sc, rs, zp, qm);
        o.w = op<MATH>(v.w, sc, rs, zp, qm);
        f4* q = reinterpret_cast<f4*>(y) + i;
        if constexpr (NT & 2) __builtin_nontemporal_store(o, q); else *q = o;
    }
}

template <int MATH, int J, int NT, int UNR>
static void launch_cw(const float* x, float* y, int N, int C, int HW, int S, const float* qp) {
    const int cpc = HW / 4, cap = TPB * J;
    int nb = (cpc + cap - 1) / cap;
    const int w = (cpc + nb - 1) / nb;
    nb = (cpc + w - 1) / w;
    hipLaunchKernelGGL((k_colwalk<MATH, J, NT, UNR>), dim3((unsigned)(C * nb * S)), dim3(TPB), 0, 0, x, y, N, C, HW, nb,
                       w, S, qp);
}


// read-only pass (stands for the statistics pass): max over the block's columns
template <int J>
__global__ void __launch_bounds__(TPB) k_readonly(const float* __restrict__ x, float* __restrict__ out, int N, int C,
                                                  int HW, int nb, int w, int S) {
    const int ncb = C * nb;
    const int cb = blockIdx.x % ncb, s = blockIdx.x / ncb;
    const int n0 = (int)(((int64_t)s * N) / S), n1 = (int)(((int64_t)(s + 1) * N) / S);
    const int cpc = HW / 4, c = cb / nb, bb = cb - c * nb;
    const int col0 = c * cpc + bb * w, col1 = min(col0 + w, (c + 1) * cpc);
    int col[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { const int cc = col0 + j * TPB + threadIdx.x; col[j] = cc < col1 ? cc : col0; }
    const size_t P = (size_t)C * HW;
    size_t off = (size_t)n0 * P;
    float m = -1e30f;
#pragma unroll 2
    for (int n = n0; n < n1; ++n, off += P) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f4 v = *reinterpret_cast<const f4*>(x + off + (size_t)col[j] * 4);
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    }
    if (m == 12345.678f) out[blockIdx.x] = m;
}

// reversed traversal: last block first, last sample first
template <int MATH, int J, int NT>
__global__ void __launch_bounds__(TPB) k_colwalk_rev(const float* __restrict__ x, float* __restrict__ y, int N, int C,
                                                     int HW, int nb, int w, int S, const float* __restrict__ qp) {
    const int ncb = C * nb;
    const int bid = gridDim.x - 1 - blockIdx.x;
    const int cb = bid % ncb, s = bid / ncb;
    const int n0 = (int)(((int64_t)s * N) / S), n1 = (int)(((int64_t)(s + 1) * N) / S);
    const int cpc = HW / 4, c = cb / nb, bb = cb - c * nb;
    const int col0 = c * cpc + bb * w, col1 = min(col0 + w, (c + 1) * cpc);
    const float sc = qp[c], zp = qp[C + c], qm = qp[2 * C + c];
    const float rs = 1.0f / sc;
    int col[J];
    bool ok[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { const int cc = col0 + j * TPB + threadIdx.x; ok[j] = cc < col1; col[j] = ok[j] ? cc : col0; }
    const size_t P = (size_t)C * HW;
#pragma unroll 2
    for (int n = n1 - 1; n >= n0; --n) {
        const size_t off = (size_t)n * P;
        f4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f4* p = reinterpret_cast<const f4*>(x + off + (size_t)col[j] * 4);
            if constexpr (NT & 1) v[j] = __builtin_nontemporal_load(p); else v[j] = *p;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f4 o;
            o.x = op<MATH>(v[j].x, sc, rs, zp, qm); o.y = op<MATH>(v[j].y, sc, rs, zp, qm);
            o.z = op<MATH>(v[j].z, sc, rs, zp, qm); o.w = op<MATH>(v[j].w, sc, rs, zp, qm);
            if (ok[j]) {
                f4* p = reinterpret_cast<f4*>(y + off + (size_t)col[j] * 4);
                if constexpr (NT & 2) __builtin_nontemporal_store(o, p); else *p = o;
            }
        }
    }
}

// sequence: read-only pass over x_i then Q/DQ x_i -> y_i, over nbuf rotating buffer pairs
extern "C" float useq(int variant, float* const* xs, float* const* ys, int nbuf, int N, int C, int HW, int S,
                      const float* qp, float* scratch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int cpc = HW / 4, cap = TPB * 4;
    int nb = (cpc + cap - 1) / cap;
    const int w = (cpc + nb - 1) / nb;
    nb = (cpc + w - 1) / w;
    const dim3 grid((unsigned)(C * nb * S));
    auto go = [&](int i) {
        const float* x = xs[i % nbuf];
        float* y = ys[i % nbuf];
        if (variant != 9) hipLaunchKernelGGL((k_readonly<4>), grid, dim3(TPB), 0, 0, x, scratch, N, C, HW, nb, w, S);
        switch (variant) {
            case 0: case 9: hipLaunchKernelGGL((k_colwalk<M_DIV, 4, 0, 2>), grid, dim3(TPB), 0, 0, x, y, N, C, HW, nb, w, S, qp); break;
            case 1: hipLaunchKernelGGL((k_colwalk_rev<M_DIV, 4, 0>), grid, dim3(TPB), 0, 0, x, y, N, C, HW, nb, w, S, qp); break;
            case 2: hipLaunchKernelGGL((k_colwalk_rev<M_DIV, 4, 2>), grid, dim3(TPB), 0, 0, x, y, N, C, HW, nb, w, S, qp); break;
            case 3: hipLaunchKernelGGL((k_colwalk_rev<M_DIV, 4, 3>), grid, dim3(TPB), 0, 0, x, y, N, C, HW, nb, w, S, qp); break;
            case 4: hipLaunchKernelGGL((k_colwalk<M_DIV, 4, 2, 2>), grid, dim3(TPB), 0, 0, x, y, N, C, HW, nb, w, S, qp); break;
            case 8: break;  // read-only pass alone
        }
    };
    go(0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) go(r + 1);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

extern "C" float ubench(int variant, const float* x, float* y, int N, int C, int HW, int S, int grid,
                        const float* qp, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t n4 = (size_t)N * C * HW / 4;
    auto go = [&]() {
        switch (variant) {
            case 0: launch_cw<M_DIV, 4, 0, 2>(x, y, N, C, HW, S, qp); break;
            case 1: launch_cw<M_COPY, 4, 0, 2>(x, y, N, C, HW, S, qp); break;
            case 2: launch_cw<M_RCP, 4, 0, 2>(x, y, N, C, HW, S, qp); break;
            case 3: launch_cw<M_DIV, 4, 2, 2>(x, y, N, C, HW, S, qp); break;
            case 4: launch_cw<M_DIV, 4, 3, 2>(x, y, N, C, HW, S, qp); break;
            case 5: launch_cw<M_DIV, 2, 0, 2>(x, y, N, C, HW, S, qp); break;
            case 6: launch_cw<M_DIV, 4, 0, 1>(x, y, N, C, HW, S, qp); break;
            case 7: launch_cw<M_DIV, 4, 0, 4>(x, y, N, C, HW, S, qp); break;
            case 8: launch_cw<M_COPY, 4, 2, 2>(x, y, N, C, HW, S, qp); break;
            case 9: launch_cw<M_DIV, 1, 0, 4>(x, y, N, C, HW, S, qp); break;
            case 10: hipLaunchKernelGGL((k_flat<M_COPY, 0>), dim3(grid), dim3(TPB), 0, 0, x, y, n4, C, HW / 4, qp); break;
            case 11: hipLaunchKernelGGL((k_flat<M_DIV, 0>), dim3(grid), dim3(TPB), 0, 0, x, y, n4, C, HW / 4, qp); break;
            case 12: hipLaunchKernelGGL((k_flat<M_COPY, 2>), dim3(grid), dim3(TPB), 0, 0, x, y, n4, C, HW / 4, qp); break;
            case 13: hipLaunchKernelGGL((k_flat<M_DIV, 2>), dim3(grid), dim3(TPB), 0, 0, x, y, n4, C, HW / 4, qp); break;
            case 14: hipLaunchKernelGGL((k_flat<M_RCP, 0>), dim3(grid), dim3(TPB), 0, 0, x, y, n4, C, HW / 4, qp); break;
            case 15: launch_cw<M_COPY, 2, 0, 2>(x, y, N, C, HW, S, qp); break;
            case 16: launch_cw<M_DIV, 2, 0, 4>(x, y, N, C, HW, S, qp); break;
            case 17: launch_cw<M_DIV, 2, 2, 4>(x, y, N, C, HW, S, qp); break;
        }
    };
    go();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) go();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / reps;
}
