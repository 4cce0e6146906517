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
    // one workgroup; thread t owns the channels [t * per, (t + 1) * per): its byte count, an exclusive scan of the
    // PTPB counts (wave shuffles, then the 16 wave totals), then the running offsets of its own channels
    __shared__ uint32_t wsum[PTPB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (C + PTPB - 1) / PTPB;
    uint32_t sum = 0;
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) {
        const uint32_t b = (uint32_t)bits[c];
        sum += (((uint32_t)HW * b + 31u) / 32u) * 4u;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    for (int c = tid * per; c < min(C, (tid + 1) * per); ++c) {
        rowoff[c] = base;
        const uint32_t b = (uint32_t)bits[c];
        base += (((uint32_t)HW * b + 31u) / 32u) * 4u;
    }
    if (tid == PTPB - 1) rowoff[C] = base;
}

// Workgroup = k ADJACENT channels x a range of samples: per sample that is one contiguous run of k * H*W floats in x
// and of consecutive rows in the packed stream (with one channel per workgroup the 7x7 layers read 196-byte pieces
// 400 KB apart).  The unit of work is a SLOT = half a group = 4 consecutive elements = one 16-byte access to x / y;
// rows are padded to an even number of slots, so a group (slots 2g, 2g + 1) never straddles two rows.  A wave takes
// 128 consecutive slots of the flat (sample, channel, slot) space at a time:
//   * lane L moves slots base + L and base + 64 + L of x / y - two fully coalesced 16-byte accesses (8 elements per
//     lane as ONE 32-byte piece touched every cache line twice: 3.5 TB/s);
//   * lane L owns group base / 2 + L of the stream - b whole bytes, stored / loaded in the widest aligned pieces;
//   * in between, the 4 * b code bits of a slot travel between the two views with wave shuffles (4 per 8 elements).
// Samples are the outer loop (per-sample base pointers in scalar registers, 32-bit offsets inside), the sample's
// slots the inner one; no integer division anywhere in the loops.
// q -> (channel within the block, slot within the row) without an integer division
__device__ __forceinline__ void slot_split(int q, int nslots, float inv_ns, int nch, int& ch, int& sl) {
    int c = (int)(((float)q + 0.5f) * inv_ns);
    c = min(c, nch - 1);
    int t = q - c * nslots;
    if (t < 0) { --c; t += nslots; } else if (t >= nslots) { ++c; t -= nslots; }
    ch = c;
    sl = t;
}

template <bool QUANT>
__global__ void __launch_bounds__(TPB) k_packed(const float* __restrict__ x, float* __restrict__ y,
                                                uint8_t* __restrict__ packed, int N, int C, int HW, int S, int k,
                                                const float* __restrict__ qp, const float* __restrict__ bits,
                                                const uint32_t* __restrict__ rowoff) {
    __shared__ float sh_sc[MAXCH], sh_zp[MAXCH], sh_qm[MAXCH];
    __shared__ uint32_t sh_off[MAXCH];
    __shared__ int sh_b[MAXCH];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ncb = (C + k - 1) / k;
    const int s = (int)blockIdx.x / ncb, cb = (int)blockIdx.x - s * ncb;
    const int c0 = cb * k, nch = min(k, C - c0);
    const int n0 = (int)(((int64_t)s * N) / S), n1 = (int)(((int64_t)(s + 1) * N) / S);
    for (int i = tid; i < nch; i += TPB) {
        sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * C + c0 + i];
        sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * C + c0 + i];
        sh_qm[i] = qp[(size_t)CNNQ_QP_QMAX * C + c0 + i];
        sh_off[i] = rowoff[c0 + i];
        sh_b[i] = (int)bits[c0 + i];
    }
    __syncthreads();
    const uint32_t plane = rowoff[C];
    const int ngroups = (HW + 7) / 8;
    const int nslots = 2 * ngroups;              // slots per row (the last one may be empty)
    const int W = nch * nslots;                  // slots of the block per sample (even)
    const float inv_ns = 1.f / (float)nslots;
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)(QUANT ? (const void*)x : (const void*)y) & 15) == 0);
    const int src = (2 * lane) & 63;             // lane holding slot 2 * lane of the wave's chunk (view A or B)
    // A wave's unit is a chunk of 128 slots of one sample; the workgroup's chunks (samples x chunks per sample) are
    // dealt to its waves round-robin and taken U at a time: all 2 * U loads of a lane are issued before the first
    // code is computed (a tile of ~14-28 KB per workgroup is in flight as a whole; with one chunk at a time the pass
    // was latency-bound at 4.1 TB/s).
    constexpr int U = 1;   // U = 4 / 2 (more loads in flight, 100 VGPRs) ran the quantize pass at 2.7 instead of 4.3 TB/s
    const int cps = (W + 127) / 128;             // chunks per sample
    const int nchunks = (n1 - n0) * cps;
    int it = wv, itn = 0, itc = wv;              // chunk index, its sample offset and chunk within the sample
    while (itc >= cps) { itc -= cps; ++itn; }
    const int dn = (TPB / 64) / cps, dc = (TPB / 64) % cps;
    while (it < nchunks) {
        int cha[U], chb[U], chg[U], cnta[U], cntb[U], gi[U];
        unsigned offa[U], offb[U];
        uint32_t nb[U];
        size_t sbase[U];
        uint8_t* gp[U];
        float va[U][4], vb[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = it < nchunks;
            const int n = n0 + (live ? itn : 0), base = (live ? itc : 0) * 128;
            const int Wl = live ? W : 0;          // a dead unit has no slots
            const int qa = base + lane, qb = qa + 64, qg = base + 2 * lane;
            int sla, slb, slg;
            slot_split(min(qa, W - 1), nslots, inv_ns, nch, cha[u], sla);
            slot_split(min(qb, W - 1), nslots, inv_ns, nch, chb[u], slb);
            slot_split(min(qg, W - 2), nslots, inv_ns, nch, chg[u], slg);
            cnta[u] = qa < Wl ? max(0, min(4, HW - sla * 4)) : 0;
            cntb[u] = qb < Wl ? max(0, min(4, HW - slb * 4)) : 0;
            offa[u] = (unsigned)(cha[u] * HW + sla * 4);
            offb[u] = (unsigned)(chb[u] * HW + slb * 4);
            sbase[u] = ((size_t)n * C + c0) * (size_t)HW;          // the block's first element of this sample
            const int b = sh_b[chg[u]];
            gi[u] = slg >> 1;
            const uint32_t rowbytes = (((uint32_t)HW * (uint32_t)b + 31u) / 32u) * 4u;
            // bytes of this group: b whole bytes, or - last group of the row - everything up to the padded row end
            const uint32_t boff = (uint32_t)gi[u] * (uint32_t)b;
            nb[u] = qg < Wl ? ((gi[u] == ngroups - 1) ? rowbytes - boff : (uint32_t)b) : 0u;
            gp[u] = packed + (size_t)n * plane + sh_off[chg[u]] + boff;
            if constexpr (QUANT) {
                const float* xs = x + sbase[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[u][e] = 0.f; vb[u][e] = 0.f; }
                if (cnta[u]) { if (vec4) ldv_nt<4>(xs + offa[u], va[u]); else for (int e = 0; e < cnta[u]; ++e) va[u][e] = xs[offa[u] + e]; }
                if (cntb[u]) { if (vec4) ldv_nt<4>(xs + offb[u], vb[u]); else for (int e = 0; e < cntb[u]; ++e) vb[u][e] = xs[offb[u] + e]; }
            }
            it += TPB / 64;
            itn += dn;
            itc += dc;
            if (itc >= cps) { itc -= cps; ++itn; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = sh_b[chg[u]];
            if constexpr (QUANT) {
                // 4 codes of b' <= 8 bits each (b' of the slot's own channel): one 32-bit word, full-rate shift-or
                auto half_of = [&](const float (&v)[4], int cnt, int ch) -> unsigned {
                    const int bb = sh_b[ch];
                    const float sc = sh_sc[ch], zp = sh_zp[ch], qm = sh_qm[ch];
                    unsigned cds[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float cd;
                        (void)qdq1(v[e], sc, zp, qm, cd);
                        cds[e] = (e < cnt) ? (unsigned)cd : 0u;
                    }
                    return cds[0] | (cds[1] << bb) | (cds[2] << (2 * bb)) | (cds[3] << (3 * bb));
                };
                const unsigned ha = half_of(va[u], cnta[u], cha[u]), hb = half_of(vb[u], cntb[u], chb[u]);
                // group `lane` = slots 2 * lane, 2 * lane + 1 of the chunk: in view A for lane < 32, else in view B
                const unsigned a0 = (unsigned)__shfl((int)ha, src, 64), b0 = (unsigned)__shfl((int)hb, src, 64);
                const unsigned a1 = (unsigned)__shfl((int)ha, src + 1, 64), b1 = (unsigned)__shfl((int)hb, src + 1, 64);
                const unsigned lo = lane < 32 ? a0 : b0, hi = lane < 32 ? a1 : b1;
                const unsigned long long ww = (unsigned long long)lo | ((unsigned long long)hi << (4 * b));
                // widest naturally aligned stores (the group starts at gi * b): 8 / 4 / 2-byte pieces where b allows
                uint8_t* g = gp[u];
                const uint32_t nbu = nb[u];
                if (nbu == 8u && b == 8) *reinterpret_cast<unsigned long long*>(g) = ww;
                else if (nbu == 4u && b == 4) *reinterpret_cast<uint32_t*>(g) = (uint32_t)ww;
                else if ((b & 1) == 0 && (nbu & 1u) == 0u)
                    for (uint32_t kk = 0; kk < nbu; kk += 2) *reinterpret_cast<uint16_t*>(g + kk) = (uint16_t)(kk < 8 ? (ww >> (8 * kk)) : 0ull);
                else
                    for (uint32_t kk = 0; kk < nbu; ++kk) g[kk] = (uint8_t)(kk < 8 ? (ww >> (8 * kk)) : 0ull);
            } else {
                float* ys = y + sbase[u];
                const uint8_t* g = gp[u];
                unsigned long long w = 0;
                const uint32_t nr = nb[u] < 8 ? nb[u] : 8;
                if (nr == 8u && b == 8) w = *reinterpret_cast<const unsigned long long*>(g);
                else if (nr == 4u && b == 4) w = *reinterpret_cast<const uint32_t*>(g);
                else if ((b & 1) == 0)
                    for (uint32_t kk = 0; kk < nr; kk += 2) w |= (unsigned long long)*reinterpret_cast<const uint16_t*>(g + kk) << (8 * kk);
                else
                    for (uint32_t kk = 0; kk < nr; ++kk) w |= (unsigned long long)g[kk] << (8 * kk);
                const unsigned wl = (unsigned)w, wh = (unsigned)(w >> (4 * b));   // codes 0-3 (4b <= 32 bits), 4-7
                // slot L of view A is half (L & 1) of group L >> 1; slot L of view B of group 32 + (L >> 1)
                const int ga = lane >> 1, gb = 32 + (lane >> 1);
                const unsigned al = (unsigned)__shfl((int)wl, ga, 64), ah = (unsigned)__shfl((int)wh, ga, 64);
                const unsigned bl = (unsigned)__shfl((int)wl, gb, 64), bh = (unsigned)__shfl((int)wh, gb, 64);
                const unsigned ma = (lane & 1) ? ah : al, mb = (lane & 1) ? bh : bl;
                auto emit = [&](unsigned mine, unsigned off, int cnt, int ch) {
                    if (!cnt) return;
                    const int bb = sh_b[ch];
                    const float sc = sh_sc[ch], zp = sh_zp[ch];
                    const unsigned mask = (1u << bb) - 1u;                        // bb <= 8
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ((float)((mine >> (e * bb)) & mask) - zp) * sc;   // iq.py:591-592
                    if (vec4) stv_nt<4>(ys + off, o);
                    else for (int e = 0; e < cnt; ++e) ys[off + e] = o[e];
                };
                emit(ma, offa[u], cnta[u], cha[u]);
                emit(mb, offb[u], cntb[u], chb[u]);
            }
        }
    }
}

}  // namespace
