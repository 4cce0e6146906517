// cnnq_pack4.hip.h - integer codes (packed int4, or one byte each) as the stored activation format.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_qdq.hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// packed int4 storage: the integer codes of a <= 4-bit quantization, two per
// byte (even element in the low nibble), as the STORED activation format - 4 B read + 0.5 B written
// per element instead of 4 + 4; k_unpack4_dq reproduces the dequantized floats of k_qdq bit for bit.
// BITS = 8: one byte per code (<= 8-bit quantization, e.g. the first layer the reference keeps at 8 bit).
// ------------------------------------------------------------------------------------------
template <int J, int BITS>
__global__ void __launch_bounds__(TPB) k_q_pack4(const float* __restrict__ x, uint8_t* __restrict__ packed,
                                                 const Geo g, const float* __restrict__ qp) {
    __shared__ float sh_sc[MAXCH], sh_zp[MAXCH], sh_qm[MAXCH];
    const Blk b = blk_of<4>(g);
    const int tid = threadIdx.x;
    for (int i = tid; i < b.c1 - b.c0; i += TPB) {
        sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0 + i];
        sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * g.C + b.c0 + i];
        sh_qm[i] = qp[(size_t)CNNQ_QP_QMAX * g.C + b.c0 + i];
    }
    __syncthreads();
    int col[J];
    bool ok[J];
    float sc[J], zp[J], qm[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
        const int ch = (int)(((unsigned)col[j] * 4u) / (unsigned)g.HW) - b.c0;
        sc[j] = sh_sc[ch]; zp[j] = sh_zp[ch]; qm[j] = sh_qm[ch];
    }
    for (int n = b.n0; n < b.n1; ++n) {
        const size_t off = (size_t)n * (size_t)g.P;
        float v[J][4];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv_nt<4>(x + off + (size_t)col[j] * 4, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float cd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) (void)qdq1(v[j][e], sc[j], zp[j], qm[j], cd[e]);
            if (ok[j]) {
                if constexpr (BITS == 4) {
                    const unsigned pk = ((unsigned)cd[0] & 15u) | (((unsigned)cd[1] & 15u) << 4) |
                                        (((unsigned)cd[2] & 15u) << 8) | (((unsigned)cd[3] & 15u) << 12);
                    *reinterpret_cast<uint16_t*>(packed + (off + (size_t)col[j] * 4) / 2) = (uint16_t)pk;
                } else {
                    const unsigned pk = ((unsigned)cd[0] & 255u) | (((unsigned)cd[1] & 255u) << 8) |
                                        (((unsigned)cd[2] & 255u) << 16) | (((unsigned)cd[3] & 255u) << 24);
                    *reinterpret_cast<uint32_t*>(packed + off + (size_t)col[j] * 4) = pk;
                }
            }
        }
    }
}

template <int J, int BITS>
__global__ void __launch_bounds__(TPB) k_unpack4_dq(const uint8_t* __restrict__ packed, float* __restrict__ y,
                                                    const Geo g, const float* __restrict__ qp) {
    __shared__ float sh_sc[MAXCH], sh_zp[MAXCH];
    const Blk b = blk_of<4>(g);
    const int tid = threadIdx.x;
    for (int i = tid; i < b.c1 - b.c0; i += TPB) {
        sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0 + i];
        sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * g.C + b.c0 + i];
    }
    __syncthreads();
    int col[J];
    bool ok[J];
    float sc[J], zp[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
        const int ch = (int)(((unsigned)col[j] * 4u) / (unsigned)g.HW) - b.c0;
        sc[j] = sh_sc[ch]; zp[j] = sh_zp[ch];
    }
    for (int n = b.n0; n < b.n1; ++n) {
        const size_t off = (size_t)n * (size_t)g.P;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const unsigned pk = BITS == 4 ? *reinterpret_cast<const uint16_t*>(packed + (off + (size_t)col[j] * 4) / 2)
                                          : *reinterpret_cast<const uint32_t*>(packed + off + (size_t)col[j] * 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = ((float)((pk >> (BITS * e)) & ((1u << BITS) - 1u)) - zp[j]) * sc[j];   // iq.py:591-592
            if (ok[j]) stv_nt<4>(y + off + (size_t)col[j] * 4, o);
        }
    }
}

}  // namespace
