// cnnq_pack4.hip.h - integer codes (packed int4, or one byte each) as the stored activation format.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_qdq.hip.h"
#include "cnnq_params.hip.h"   // PTPB

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

// ------------------------------------------------------------------------------------------
// variable-width packed storage (SURVEY.md 8 f3 with bit allocation, iq.py:563-564,582-584): channel c stores
// bits[c] in 0..8 bits per code, so an activation costs sum(bits)/8 bytes per spatial position instead of 4*C -
// the deployment format the paper's bit-rate tables assume; the reference only simulates it (codes stay fp32).
//
// Layout: row (n, c) = H*W codes of bits[c] bits each, little-endian bit stream (element i occupies bits
// [i*b, (i+1)*b)), padded to 4 bytes; row (n, c) starts at n * rowoff[C] + rowoff[c] (k_packed_layout: rowoff[c] =
// running sum of ceil(H*W*bits/32)*4, rowoff[C] = bytes per sample).  A 0-bit channel stores nothing and decodes
// to the constant (0 - zp) * scale, exactly what the fused Q/DQ returns for it.
// A lane handles 8 consecutive elements of a row = bits[c] whole bytes; a workgroup walks a flat (row, group)
// index space of one channel, so small H*W still fills the lanes.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PTPB) k_packed_layout(const float* __restrict__ bits, int C, int HW,
                                                        uint32_t* __restrict__ rowoff) {
    // one workgroup; channel counts are small (<= a few thousand): a serial scan by one thread per PTPB-chunk
    __shared__ uint32_t chunk[PTPB];
    const int tid = threadIdx.x;
    const int per = (C + PTPB - 1) / PTPB;
    uint32_t sum = 0;
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) {
        const uint32_t b = (uint32_t)bits[c];
        sum += (((uint32_t)HW * b + 31u) / 32u) * 4u;
    }
    chunk[tid] = sum;
    __syncthreads();
    uint32_t base = 0;
    for (int t = 0; t < tid; ++t) base += chunk[t];
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) {
        rowoff[c] = base;
        const uint32_t b = (uint32_t)bits[c];
        base += (((uint32_t)HW * b + 31u) / 32u) * 4u;
    }
    if (tid == PTPB - 1) rowoff[C] = base;
}

template <bool QUANT>
__global__ void __launch_bounds__(TPB) k_packed(const float* __restrict__ x, float* __restrict__ y,
                                                uint8_t* __restrict__ packed, int N, int C, int HW, int S,
                                                const float* __restrict__ qp, const float* __restrict__ bits,
                                                const uint32_t* __restrict__ rowoff) {
    const int c = (int)blockIdx.x / S, s = (int)blockIdx.x - c * S;
    const int n0 = (int)(((int64_t)s * N) / S), n1 = (int)(((int64_t)(s + 1) * N) / S);
    const int b = (int)bits[c];
    const float sc = qp[(size_t)CNNQ_QP_SCALE * C + c], zp = qp[(size_t)CNNQ_QP_ZP * C + c];
    const float qm = qp[(size_t)CNNQ_QP_QMAX * C + c];
    const uint32_t plane = rowoff[C], roff = rowoff[c];
    const uint32_t rowbytes = (((uint32_t)HW * (uint32_t)b + 31u) / 32u) * 4u;
    const int ngroups = (HW + 7) / 8;
    const int64_t total = (int64_t)(n1 - n0) * ngroups;
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)(QUANT ? (const void*)x : (const void*)y) & 15) == 0);
    for (int64_t idx = threadIdx.x; idx < total; idx += TPB) {
        const int r = (int)(idx / ngroups), gi = (int)(idx - (int64_t)r * ngroups);
        const int n = n0 + r;
        const int e0 = gi * 8, cnt = min(8, HW - e0);
        const size_t xoff = ((size_t)n * C + c) * (size_t)HW + e0;
        uint8_t* rowp = packed + (size_t)n * plane + roff;
        // bytes of this group: b whole bytes, or - last group of the row - everything up to the padded row end
        const uint32_t boff = (uint32_t)gi * (uint32_t)b;
        const uint32_t nb = (gi == ngroups - 1) ? rowbytes - boff : (uint32_t)b;
        if constexpr (QUANT) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (vec4 && cnt == 8) {
                ldv_nt<4>(x + xoff, *reinterpret_cast<float(*)[4]>(&v[0]));
                ldv_nt<4>(x + xoff + 4, *reinterpret_cast<float(*)[4]>(&v[4]));
            } else {
                for (int e = 0; e < cnt; ++e) v[e] = x[xoff + e];
            }
            unsigned long long w = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float cd;
                (void)qdq1(v[e], sc, zp, qm, cd);
                if (e < cnt) w |= (unsigned long long)(unsigned)cd << (e * b);
            }
            // widest naturally aligned stores (boff = gi * b): 8 / 4 / 2-byte pieces where b allows, bytes otherwise
            if (nb == 8u && b == 8) *reinterpret_cast<unsigned long long*>(rowp + boff) = w;
            else if (nb == 4u && b == 4) *reinterpret_cast<uint32_t*>(rowp + boff) = (uint32_t)w;
            else if ((b & 1) == 0 && (nb & 1u) == 0u)
                for (uint32_t k = 0; k < nb; k += 2) *reinterpret_cast<uint16_t*>(rowp + boff + k) = (uint16_t)(k < 8 ? (w >> (8 * k)) : 0ull);
            else
                for (uint32_t k = 0; k < nb; ++k) rowp[boff + k] = (uint8_t)(k < 8 ? (w >> (8 * k)) : 0ull);
        } else {
            unsigned long long w = 0;
            const uint32_t nr = nb < 8 ? nb : 8;
            for (uint32_t k = 0; k < nr; ++k) w |= (unsigned long long)rowp[boff + k] << (8 * k);
            const unsigned long long mask = (b >= 8) ? 0xffull : ((1ull << b) - 1ull);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ((float)((w >> (e * b)) & mask) - zp) * sc;   // iq.py:591-592
            if (vec4 && cnt == 8) {
                stv_nt<4>(y + xoff, *reinterpret_cast<float(*)[4]>(&o[0]));
                stv_nt<4>(y + xoff + 4, *reinterpret_cast<float(*)[4]>(&o[4]));
            } else {
                for (int e = 0; e < cnt; ++e) y[xoff + e] = o[e];
            }
        }
    }
}

}  // namespace
