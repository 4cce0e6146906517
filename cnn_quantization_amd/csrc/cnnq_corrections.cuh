// cnnq_corrections.cuh - weight bias/variance correction and activation bias correction.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.cuh"
#include "cnnq_stats.cuh"

namespace {

// ------------------------------------------------------------------------------------------
// bias / variance correction (iqm.py:180-196 activations, iqm.py:374-391 weights)
// ------------------------------------------------------------------------------------------
// per-channel affine update y = ((y - a) * m + a) - s + t  (weights): with
//   a = mean(w_q), m = std(w)/(std(w_q)+1e-8) (variance correction, optional), s = a, t = mean(w)
// evaluated with the reference's operation order so that equal constants give equal bits.
__global__ void __launch_bounds__(TPB) k_weight_correct(float* __restrict__ wq, int C, int HW,
                                                        const float* __restrict__ st_w,
                                                        const float* __restrict__ st_q, int vcorr, int bcorr) {
    const int c = blockIdx.y;
    const float bias_q = st_q[(size_t)CNNQ_STAT_MEAN * C + c];
    const float bias_o = st_w[(size_t)CNNQ_STAT_MEAN * C + c];
    const float var_corr = st_w[(size_t)CNNQ_STAT_STD * C + c] / (st_q[(size_t)CNNQ_STAT_STD * C + c] + 1e-8f);
    float* row = wq + (size_t)c * HW;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
        float v = row[i];
        if (vcorr) v = (v - bias_q) * var_corr + bias_q;   // iqm.py:387
        if (bcorr) v = v - bias_q + bias_o;                 // iqm.py:391 (bias_q is the pre-correction mean)
        row[i] = v;
    }
}

// activation bias correction, pass 1: per channel sum(x'), sum(y), count(x' > 0) with x' = relu(x)
// when the layer feeds a ReLU (iqm.py:188-193) -> part3[G][3][C] (fp64)
template <int VEC, int A, int J>
__global__ void __launch_bounds__(TPB) k_bcorr_sums(const float* __restrict__ x, const float* __restrict__ y,
                                                    const Geo g, int relu_first, double* __restrict__ part3) {
    constexpr int NE = TPB * J * A;
    __shared__ double l_sx[NE], l_sy[NE], l_cn[NE];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    int col[J];
    bool ok[J];
    double sx[J][A], sy[J][A], cn[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) { sx[j][a] = 0.; sy[j][a] = 0.; cn[j][a] = 0.; }
    }
    size_t off = (size_t)b.n0 * (size_t)g.P;
    for (int n = b.n0; n < b.n1; ++n, off += g.P) {
        float vx[J][VEC], vy[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            ldv<VEC>(x + off + (size_t)col[j] * VEC, vx[j]);
            ldv<VEC>(y + off + (size_t)col[j] * VEC, vy[j]);
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                const float xv = relu_first ? fmaxf(vx[j][e], 0.f) : vx[j][e];
                sx[j][a] += (double)xv;
                sy[j][a] += (double)vy[j][e];
                cn[j][a] += (xv > 0.f) ? 1. : 0.;
            }
    }
    auto emit = [&](int ch, double a, double bq, double c) {
        double* p = part3 + (size_t)b.grp * 3 * g.C + ch;
        p[0] = a;
        p[(size_t)g.C] = bq;
        p[(size_t)2 * g.C] = c;
    };
    const int wv = tid >> 6, lane = tid & 63;
    if (g.mode == 1) {
        double ta = 0., tb = 0., tc = 0.;
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (ok[j]) { ta += sx[j][0]; tb += sy[j][0]; tc += cn[j][0]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ta += shfl_xor_d(ta, m); tb += shfl_xor_d(tb, m); tc += shfl_xor_d(tc, m); }
        if (lane == 0) { l_sx[wv] = ta; l_sy[wv] = tb; l_cn[wv] = tc; }
        __syncthreads();
        if (tid == 0) {
            double ra = 0., rb = 0., rc = 0.;
            for (int i = 0; i < TPB / 64; ++i) { ra += l_sx[i]; rb += l_sy[i]; rc += l_cn[i]; }
            emit(b.c0, ra, rb, rc);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const int e = (j * TPB + tid) * A + a;
                l_sx[e] = sx[j][a]; l_sy[e] = sy[j][a]; l_cn[e] = cn[j][a];
            }
        }
    __syncthreads();
    const int epc = g.HW * A / VEC;
    for (int ch = b.c0 + wv; ch < b.c1; ch += TPB / 64) {
        const int lo = (ch - b.c0) * epc;
        double ra = 0., rb = 0., rc = 0.;
        for (int e = lo + lane; e < lo + epc; e += 64) { ra += l_sx[e]; rb += l_sy[e]; rc += l_cn[e]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ra += shfl_xor_d(ra, m); rb += shfl_xor_d(rb, m); rc += shfl_xor_d(rc, m); }
        if (lane == 0) emit(ch, ra, rb, rc);
    }
}

// merge G records -> q_bias[c] = (sum x' - sum y) / (count + 1e-8)   (iqm.py:192-194); sums[3][C] optional
__global__ void __launch_bounds__(TPB) k_bcorr_bias(const double* __restrict__ part3, int G, int C,
                                                    double* __restrict__ sums, float* __restrict__ bias) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= C) return;
    double a = 0., bq = 0., cn = 0.;
    for (int gi = 0; gi < G; ++gi) {
        const double* p = part3 + (size_t)gi * 3 * C + c;
        a += p[0]; bq += p[(size_t)C]; cn += p[(size_t)2 * C];
    }
    if (sums) { sums[c] = a; sums[(size_t)C + c] = bq; sums[(size_t)2 * C + c] = cn; }
    if (bias) {
        const float qb = (float)a - (float)bq;
        bias[c] = qb / ((float)cn + 1e-8f);
    }
}

// pass 2: y += (y > 0) * q_bias[c]   (iqm.py:196), in place
template <int VEC, int A, int J>
__global__ void __launch_bounds__(TPB) k_bcorr_apply(float* __restrict__ y, const Geo g,
                                                     const float* __restrict__ bias) {
    __shared__ float sh_b[MAXCH];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    for (int i = tid; i < b.c1 - b.c0; i += TPB) sh_b[i] = bias[b.c0 + i];
    __syncthreads();
    int col[J];
    bool ok[J];
    float qb[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const unsigned e = (unsigned)col[j] * VEC + (A == 1 ? 0 : a);
            qb[j][a] = sh_b[(int)(e / (unsigned)g.HW) - b.c0];
        }
    }
    size_t off = (size_t)b.n0 * (size_t)g.P;
#pragma unroll 2
    for (int n = b.n0; n < b.n1; ++n, off += g.P) {
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv<VEC>(y + off + (size_t)col[j] * VEC, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                v[j][e] = v[j][e] + ((v[j][e] > 0.f) ? 1.f : 0.f) * qb[j][a];
            }
            if (ok[j]) stv<VEC>(y + off + (size_t)col[j] * VEC, v[j]);
        }
    }
}

}  // namespace
