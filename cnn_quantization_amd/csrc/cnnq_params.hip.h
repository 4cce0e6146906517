// cnnq_params.hip.h - statistics -> ACIQ clipping -> bit allocation -> scale / zero point / qmax, on the device.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// statistics -> scale / zero point / qmax (one workgroup, no host round trips)
// ------------------------------------------------------------------------------------------
constexpr int PTPB = 1024;

// sum over the workgroup (blockDim.x a multiple of 64, <= PTPB); every thread gets the result
__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = 0.;
    const int nw = (int)(blockDim.x >> 6);
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}

__constant__ float c_laplace[9] = {1.05f, 1.86f, 2.83f, 3.89f, 5.03f, 6.2f, 7.41f, 8.64f, 9.89f};
__constant__ float c_laplace_pos[9] = {1.86f, 2.83f, 3.89f, 5.02f, 6.2f, 7.41f, 8.64f, 9.89f, 11.16f};
__constant__ float c_gaus[9] = {0.f, 1.24f, 1.71f, 2.15f, 2.55f, 2.93f, 3.28f, 3.61f, 3.92f};
__constant__ float c_gaus_pos[9] = {0.f, 1.71f, 2.15f, 2.55f, 2.93f, 3.28f, 3.61f, 3.92f, 4.2f};

// Fixed-target bit allocation, iq.py:381-407 (fp32 tensor math, double target): `prior` [C] -> bits_ws [C].  One
// workgroup (blockDim.x a multiple of 64, <= PTPB); ends with a barrier.  Shared by k_params and k_bitalloc (cnnq_aciq.hip.h).
__device__ __forceinline__ void bit_alloc_block(const float* __restrict__ prior, int C, const cnnq_params_cfg& cfg,
                                                float* __restrict__ bits_ws, double* sh) {
    const int tid = threadIdx.x;
    const int T = (int)blockDim.x;
    const float goal = (float)cfg.target;
    double target = cfg.target;
    double delta = 1.;
    // p = prior^(2/3) and its sum do not change between iterations: the first PK per thread stay in
    // registers (C <= PK * 1024 covers every CNN layer), the rest are recomputed
    constexpr int PK = 4;
    float pc[PK];
    double psum_d = 0.;
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int c = tid + k * T;
        pc[k] = c < C ? powf(prior[c], (float)(2. / 3)) : 0.f;
        if (c < C) psum_d += (double)pc[k];
    }
    for (int c = tid + PK * T; c < C; c += T) psum_d += (double)powf(prior[c], (float)(2. / 3));
    const float psum = (float)block_sum(psum_d, sh);
    auto bits_of = [&](float B, float p) -> float {
        const float bins = (B * p) / psum;
        float bits = cfg.round_mode ? rintf(log2f(bins)) : ceilf(log2f(bins));
        if (bits < 0.f) bits = 0.f;
        if (bits > 8.f) bits = 8.f;
        return bits;
    };
    float bk[PK];
    for (int it = 0; it < 10 && fabs(2. * delta) > 0.01; ++it) {
        // C * 2**target (iq.py:383): exp2 instead of the generic pow - both are within an ulp of the
        // double result, which is then rounded to float
        const float B = (float)((double)C * exp2(target));
        double bsum = 0.;
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            if (k * T >= C) break;                         // uniform: slots no thread uses
            bk[k] = bits_of(B, pc[k]);
            if (tid + k * T < C) bsum += (double)bk[k];
        }
        for (int c = tid + PK * T; c < C; c += T) {
            const float bits = bits_of(B, powf(prior[c], (float)(2. / 3)));
            bits_ws[c] = bits;
            bsum += (double)bits;
        }
        const float mean_bits = (float)block_sum(bsum, sh) / (float)C;
        delta = (double)((goal - mean_bits) / 2.f);
        target += delta;
    }
#pragma unroll
    for (int k = 0; k < PK; ++k)
        if (tid + k * T < C) bits_ws[tid + k * T] = bk[k];
    __syncthreads();
}

// One channel's clipping range and quantisation parameters from its statistics (iq.py:227-300, 327-352, 559-572) - the
// single definition of this arithmetic: k_params runs it per table row, the single-launch ACIQ kernels (cnnq_aciq.hip.h)
// in every workgroup that holds a piece of the channel.
struct ChanParams {
    float alpha, delta, offset, qmax, scale, zp;
};
__device__ __forceinline__ ChanParams channel_params(const cnnq_params_cfg& cfg, bool ba, float bits, float vmin, float vmax,
                                                     float vmean, float vstd, float vb) {
    ChanParams r;
    r.alpha = 0.f;
    if (cfg.clip == 0) {
        r.offset = cfg.positive ? 0.f : vmin;
        r.delta = vmax - r.offset;
    } else {
        if (cfg.clip == 1) {
            const int ib = (int)bits;  // NaN bits cannot occur: clamped comparisons leave 0..8
            r.alpha = vb * (cfg.positive ? c_laplace_pos[ib] : c_laplace[ib]);
        } else if (cfg.clip == 2) {
            r.alpha = vstd * (cfg.positive ? c_gaus_pos[cfg.num_bits] : c_gaus[cfg.num_bits]);
        } else {
            r.alpha = cfg.pstd * vstd;
        }
        float range;
        if (cfg.positive) {
            range = fmaxf(vmean, 0.f) + r.alpha;
            r.offset = 0.f;
        } else {
            range = 2.f * r.alpha;
            r.offset = fmaxf(vmin, vmean - r.alpha);
        }
        const float mx = r.offset + range;                     // iq.py:351
        r.delta = cfg.direct_range ? range : mx - r.offset;    // iq.py:443 (per channel) / :357 (per tensor)
    }
    if (ba) {
        r.qmax = exp2f(bits) - 1.f;
        r.scale = (r.qmax > 0.f) ? r.delta / r.qmax : 0.f;
    } else {
        r.qmax = qmax_of(cfg.num_bits);
        r.scale = r.delta / r.qmax;
    }
    r.scale = (r.scale < 1e-8f) ? 1e-8f : r.scale;  // NaN stays NaN, as torch.max does
    r.zp = zero_point_of(r.offset, r.scale);
    return r;
}

__global__ void __launch_bounds__(PTPB) k_params(const float* __restrict__ stats, int C, const cnnq_params_cfg cfg,
                                                 float* __restrict__ qp, float* __restrict__ diag,
                                                 float* __restrict__ bits_ws) {
    __shared__ double sh[PTPB / 64];
    const int tid = threadIdx.x;
    const int T = (int)blockDim.x;   // min(1024, C rounded up to whole waves): small layers skip idle waves
    const float* vmin = stats + (size_t)CNNQ_STAT_MIN * C;
    const float* vmax = stats + (size_t)CNNQ_STAT_MAX * C;
    const float* vmean = stats + (size_t)CNNQ_STAT_MEAN * C;
    const float* vstd = stats + (size_t)CNNQ_STAT_STD * C;
    const float* vb = stats + (size_t)CNNQ_STAT_B * C;
    const bool ba = cfg.bit_alloc && cfg.num_bits <= 4;
    if (ba) bit_alloc_block(cfg.prior_is_b ? vb : vstd, C, cfg, bits_ws, sh);
    for (int c = tid; c < C; c += T) {
        const float bits = ba ? bits_ws[c] : (float)cfg.num_bits;
        const ChanParams r = channel_params(cfg, ba, bits, vmin[c], vmax[c], vmean[c], vstd[c], vb[c]);
        qp[(size_t)CNNQ_QP_SCALE * C + c] = r.scale;
        qp[(size_t)CNNQ_QP_ZP * C + c] = r.zp;
        qp[(size_t)CNNQ_QP_QMAX * C + c] = r.qmax;
        if (diag) {
            diag[(size_t)CNNQ_DIAG_BITS * C + c] = bits;
            diag[(size_t)CNNQ_DIAG_ALPHA * C + c] = r.alpha;
            diag[(size_t)CNNQ_DIAG_DELTA * C + c] = r.delta;
            diag[(size_t)CNNQ_DIAG_OFFSET * C + c] = r.offset;
        }
    }
}

}  // namespace
