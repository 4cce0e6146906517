// cnnq_corrections.hip.h - weight bias/variance correction and activation bias correction.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_stats.hip.h"
#include "cnnq_qdq.hip.h"

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
// FROMX: the quantized value is recomputed from x with the parameter table qp (the same qdq1 as the
// Q/DQ pass, hence the same floats) instead of being read back: the correction then costs one
// read-only pass over x plus ONE fused quantize+correct pass (k_qdq_bias) - 12 B/elem in place of the
// 24 of quantize, re-read both, update in place.
template <int VEC, int A, int J, bool FROMX, bool NTL>
__global__ void __launch_bounds__(TPB) k_bcorr_sums(const float* __restrict__ x, const float* __restrict__ y,
                                                    const Geo g, int relu_first, const float* __restrict__ qp,
                                                    double* __restrict__ part3) {
    constexpr int NE = TPB * J * A;
    __shared__ double l_sx[NE], l_sy[NE], l_cn[NE];
    __shared__ float sh_sc[FROMX ? MAXCH : 1], sh_zp[FROMX ? MAXCH : 1], sh_qm[FROMX ? MAXCH : 1];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    if constexpr (FROMX) {
        for (int i = tid; i < b.c1 - b.c0; i += TPB) {
            sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0 + i];
            sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * g.C + b.c0 + i];
            sh_qm[i] = qp[(size_t)CNNQ_QP_QMAX * g.C + b.c0 + i];
        }
        __syncthreads();
    }
    int col[J];
    bool ok[J];
    double sx[J][A], sy[J][A];
    unsigned cn[J][A];                       // exact; < 2^32 elements per lane
    float sc[J][A], zp[J][A], qm[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            sx[j][a] = 0.; sy[j][a] = 0.; cn[j][a] = 0u;
            if constexpr (FROMX) {
                const unsigned e = (unsigned)col[j] * VEC + (A == 1 ? 0 : a);
                const int ch = (int)(e / (unsigned)g.HW) - b.c0;
                sc[j][a] = sh_sc[ch]; zp[j][a] = sh_zp[ch]; qm[j][a] = sh_qm[ch];
            }
        }
    }
    size_t off = (size_t)b.n0 * (size_t)g.P;
#pragma unroll 2
    for (int n = b.n0; n < b.n1; ++n, off += g.P) {
        float vx[J][VEC], vy[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            ldv_sel<VEC, NTL>(x + off + (size_t)col[j] * VEC, vx[j]);
            if constexpr (!FROMX) ldv_sel<VEC, NTL>(y + off + (size_t)col[j] * VEC, vy[j]);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                if constexpr (FROMX) {
                    float code;
                    vy[j][e] = qdq1(vx[j][e], sc[j][a], zp[j][a], qm[j][a], code);
                }
                if (relu_first) vx[j][e] = fmaxf(vx[j][e], 0.f);
                cn[j][a] += (vx[j][e] > 0.f) ? 1u : 0u;
            }
            if constexpr (A == 1 && VEC == 4) {
                // one channel per load: the four elements are summed in fp32 first (as Mom::add4)
                sx[j][0] += (double)((vx[j][0] + vx[j][1]) + (vx[j][2] + vx[j][3]));
                sy[j][0] += (double)((vy[j][0] + vy[j][1]) + (vy[j][2] + vy[j][3]));
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    sx[j][A == 1 ? 0 : e] += (double)vx[j][e];
                    sy[j][A == 1 ? 0 : e] += (double)vy[j][e];
                }
            }
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
            if (ok[j]) { ta += sx[j][0]; tb += sy[j][0]; tc += (double)cn[j][0]; }
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
                l_sx[e] = sx[j][a]; l_sy[e] = sy[j][a]; l_cn[e] = (double)cn[j][a];
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
    // one wave per channel, lanes stride over the G records (a thread per channel walking up to 256
    // records serially took 18 us)
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    double a = 0., bq = 0., cn = 0.;
    for (int gi = lane; gi < G; gi += 64) {
        const double* p = part3 + (size_t)gi * 3 * C + c;
        a += p[0]; bq += p[(size_t)C]; cn += p[(size_t)2 * C];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a += shfl_xor_d(a, m); bq += shfl_xor_d(bq, m); cn += shfl_xor_d(cn, m); }
    if (lane) return;
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

// fused quantize + correct: y = q + (q > 0) * q_bias[c] with q = qdq1(x) (iq.py:573-592, iqm.py:196), one
// streaming pass in the Q/DQ geometry (short workgroups, non-temporal loads and stores)
template <int VEC, int A, int J>
__global__ void __launch_bounds__(TPB) k_qdq_bias(const float* __restrict__ x, float* __restrict__ y, const Geo g,
                                                  const float* __restrict__ qp, const float* __restrict__ bias) {
    __shared__ float sh_sc[MAXCH], sh_zp[MAXCH], sh_qm[MAXCH], sh_b[MAXCH];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    for (int i = tid; i < b.c1 - b.c0; i += TPB) {
        sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0 + i];
        sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * g.C + b.c0 + i];
        sh_qm[i] = qp[(size_t)CNNQ_QP_QMAX * g.C + b.c0 + i];
        sh_b[i] = bias[b.c0 + i];
    }
    __syncthreads();
    int col[J];
    bool ok[J];
    float sc[J][A], zp[J][A], qm[J][A], qb[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const unsigned e = (unsigned)col[j] * VEC + (A == 1 ? 0 : a);
            const int ch = (int)(e / (unsigned)g.HW) - b.c0;
            sc[j][a] = sh_sc[ch]; zp[j][a] = sh_zp[ch]; qm[j][a] = sh_qm[ch]; qb[j][a] = sh_b[ch];
        }
    }
    const int nrows = b.n1 - b.n0;
    constexpr int NU = (J == 1) ? 4 : 2;
#pragma unroll NU
    for (int r = 0; r < nrows; ++r) {
        const int n = g.rev ? (b.n1 - 1 - r) : (b.n0 + r);
        const size_t off = (size_t)n * (size_t)g.P;
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv_nt<VEC>(x + off + (size_t)col[j] * VEC, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                float code;
                const float q = qdq1(v[j][e], sc[j][a], zp[j][a], qm[j][a], code);
                o[e] = q + ((q > 0.f) ? 1.f : 0.f) * qb[j][a];
            }
            if (ok[j]) stv_nt<VEC>(y + off + (size_t)col[j] * VEC, o);
        }
    }
}

}  // namespace
