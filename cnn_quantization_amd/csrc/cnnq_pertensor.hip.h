// cnnq_pertensor.hip.h - per-tensor GEMMLOWP path (the replacement of kernels/gemmlowp.cu).
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// per-tensor GEMMLOWP path (replaces kernels/gemmlowp.cu)
// ------------------------------------------------------------------------------------------
// ptp: [0] scale [1] shift [2] qmax [3] true-zero flag [4] passthrough flag [5] range [6] offset
__global__ void __launch_bounds__(64) k_pt_setup(int have_host, float h_range, float h_offset,
                                                 const float* __restrict__ stats, int64_t stride, int rows,
                                                 int rows_mode, int zero_min, int num_bits, int int_exp, int etz,
                                                 float* __restrict__ ptp) {
    const int lane = threadIdx.x;
    float range, offset;
    bool ptz;
    if (have_host) {
        range = h_range;
        offset = h_offset;
        ptz = etz != 0;
    } else {
        const float* vmin = stats + (size_t)CNNQ_STAT_MIN * stride;
        const float* vmax = stats + (size_t)CNNQ_STAT_MAX * stride;
        float mn, mx;
        if (rows_mode == 0) {  // per-sample then mean over the batch (iq.py:515-526)
            double smn = 0., smx = 0.;
            for (int r = lane; r < rows; r += 64) { smn += (double)vmin[r]; smx += (double)vmax[r]; }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { smn += shfl_xor_d(smn, m); smx += shfl_xor_d(smx, m); }
            mn = (float)(smn / (double)rows);
            mx = (float)(smx / (double)rows);
        } else {  // whole tensor
            mn = INFINITY; mx = -INFINITY;
            for (int r = lane; r < rows; r += 64) { mn = fminf(mn, vmin[r]); mx = fmaxf(mx, vmax[r]); }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, shfl_xor_f(mn, m)); mx = fmaxf(mx, shfl_xor_f(mx, m)); }
        }
        if (zero_min) mn = 0.f;
        range = mx - mn;   // iq.py:379
        offset = mn;
        ptz = etz && ((offset + range) > 0.f) && (offset < 0.f);  // iq.py:613
    }
    if (lane != 0) return;
    const float qmax = (float)((1ll << num_bits) - 1);
    float scale = range / qmax;
    if (int_exp) scale = powf(2.f, (float)(int)ceilf(log2f(scale)));
    const float zero_point = roundf(-offset / scale);
    ptp[0] = scale;
    ptp[1] = ptz ? zero_point : -offset;
    ptp[2] = qmax;
    ptp[3] = ptz ? 1.f : 0.f;
    ptp[4] = (range <= 0.f) ? 1.f : 0.f;
    ptp[5] = range;
    ptp[6] = offset;
    ptp[7] = 0.f;
}

__device__ __forceinline__ float ptq1(float v, float scale, float shift, float qmax, bool etz, float nz) {
    float t = etz ? (v / scale) + shift : (v + shift) / scale;
    t = t + nz;  // the reference always adds the noise tensor (zeros when not stochastic)
    t = fminf(t, qmax);
    t = fmaxf(t, 0.f);
    t = roundf(t);
    return etz ? (t - shift) * scale : t * scale - shift;
}

// one-shot grid in address order, non-temporal streaming (the structure that reaches the copy
// ceiling on MI355X, tools/ubench_copy.py): every lane handles exactly one VEC-wide item
template <int VEC, bool NOISE>
__global__ void __launch_bounds__(TPB) k_pt_qdq(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                const float* __restrict__ ptp, const float* __restrict__ noise) {
    const float scale = ptp[0], shift = ptp[1], qmax = ptp[2];
    const bool etz = ptp[3] != 0.f, pass = ptp[4] != 0.f;
    const int64_t nv = n / VEC;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < nv) {
        float v[VEC], z[VEC], o[VEC];
        ldv_nt<VEC>(x + i * VEC, v);
        if constexpr (NOISE) ldv_nt<VEC>(noise + i * VEC, z);
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = pass ? v[e] : ptq1(v[e], scale, shift, qmax, etz, NOISE ? z[e] : 0.f);
        stv_nt<VEC>(y + i * VEC, o);
    }
    if constexpr (VEC > 1) {  // tail (n % VEC elements), handled by the first lanes of the grid
        const int64_t t = nv * VEC + i;
        if (i < VEC && t < n) y[t] = pass ? x[t] : ptq1(x[t], scale, shift, qmax, etz, NOISE ? noise[t] : 0.f);
    }
}

}  // namespace
