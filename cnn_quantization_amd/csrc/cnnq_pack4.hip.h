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

struct PkUnit {            // one wave-chunk: 128 slots of one sample
    int n, base;           // sample, first slot of the chunk within the block's slots of that sample
    int cha, chb, cnta, cntb;
    unsigned offa, offb;   // element offsets of the lane's two slots from the block's first element of the sample
    bool live;
};

// ROWS: chunks are aligned to rows (a row = ceil(nslots / 128) chunks, the last one partly empty): every lane of a wave
// then works on the SAME channel - one set of parameters per chunk instead of three per lane, no per-slot index split.
// Pays for long rows (>= 256 slots, i.e. H*W >= 1024), where the per-slot bookkeeping of the flat form was half of the
// instructions of an ALU-bound kernel.
template <bool QUANT, bool ROWS>
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
    const int ngroups = (HW + 7) / 8;
    const int nslots = 2 * ngroups;              // slots per row (the last one may be empty)
    const int W = nch * nslots;                  // slots of the block per sample (even)
    const float inv_ns = 1.f / (float)nslots;
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)(QUANT ? (const void*)x : (const void*)y) & 15) == 0);
    const int src = (2 * lane) & 63;             // lane holding slot 2 * lane of the wave's chunk (view A or B)
    // A wave's unit is a chunk of 128 slots of one sample; the workgroup's chunks (samples x chunks per sample) are
    // dealt to its waves round-robin.
    const int cpr = (nslots + 127) / 128;        // ROWS: chunks per row
    const float inv_cpr = 1.f / (float)cpr;
    const int cps = ROWS ? nch * cpr : (W + 127) / 128;   // chunks per sample
    const int nchunks = (n1 - n0) * cps;
    int it = wv, itn = 0, itc = wv;              // chunk index, its sample offset and chunk within the sample
    while (itc >= cps) { itc -= cps; ++itn; }
    const int dn = (TPB / 64) / cps, dc = (TPB / 64) % cps;
    auto locate = [&]() {                        // the unit at the cursor; advances the cursor
        PkUnit u;
        u.live = it < nchunks;
        u.n = n0 + (u.live ? itn : 0);
        if constexpr (ROWS) {
            int ch, j;                           // wave-uniform: the chunk's channel and its index within the row
            slot_split(u.live ? itc : 0, cpr, inv_cpr, nch, ch, j);
            ch = __builtin_amdgcn_readfirstlane(ch);
            j = __builtin_amdgcn_readfirstlane(j);
            u.base = j * 128;                    // first slot of the chunk within its row
            u.cha = u.chb = ch;
            const int nl = u.live ? nslots : 0;
            const int sla = u.base + lane, slb = sla + 64;
            u.cnta = sla < nl ? max(0, min(4, HW - sla * 4)) : 0;
            u.cntb = slb < nl ? max(0, min(4, HW - slb * 4)) : 0;
            u.offa = (unsigned)(ch * HW + sla * 4);
            u.offb = u.offa + 256u;
        } else {
            u.base = (u.live ? itc : 0) * 128;
            const int Wl = u.live ? W : 0;       // a dead unit has no slots
            const int qa = u.base + lane, qb = qa + 64;
            int sla, slb;
            slot_split(min(qa, W - 1), nslots, inv_ns, nch, u.cha, sla);
            slot_split(min(qb, W - 1), nslots, inv_ns, nch, u.chb, slb);
            u.cnta = qa < Wl ? max(0, min(4, HW - sla * 4)) : 0;
            u.cntb = qb < Wl ? max(0, min(4, HW - slb * 4)) : 0;
            u.offa = (unsigned)(u.cha * HW + sla * 4);
            u.offb = (unsigned)(u.chb * HW + slb * 4);
        }
        it += TPB / 64;
        itn += dn;
        itc += dc;
        if (itc >= cps) { itc -= cps; ++itn; }
        return u;
    };
    auto loadx = [&](const PkUnit& u, float (&va)[4], float (&vb)[4]) {
        const float* xs = x + ((size_t)u.n * C + c0) * (size_t)HW;     // the block's first element of the sample
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[e] = 0.f; vb[e] = 0.f; }
        if (u.cnta) { if (vec4) ldv_nt<4>(xs + u.offa, va); else for (int e = 0; e < u.cnta; ++e) va[e] = xs[u.offa + e]; }
        if (u.cntb) { if (vec4) ldv_nt<4>(xs + u.offb, vb); else for (int e = 0; e < u.cntb; ++e) vb[e] = xs[u.offb + e]; }
    };
    // x does not depend on the parameter tables: the first unit's loads are in flight while the tables arrive, every
    // further unit's while the previous one is being encoded (one unit at a time, no overlap: 4.0 TB/s of reads)
    PkUnit cur = locate();
    float va[4], vb[4];
    if constexpr (QUANT) loadx(cur, va, vb);
    for (int i = tid; i < nch; i += TPB) {
        sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * C + c0 + i];
        sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * C + c0 + i];
        sh_qm[i] = qp[(size_t)CNNQ_QP_QMAX * C + c0 + i];
        sh_off[i] = rowoff[c0 + i];
        sh_b[i] = (int)bits[c0 + i];
    }
    __syncthreads();
    const uint32_t plane = rowoff[C];
    while (cur.live) {
        PkUnit nxt = locate();
        float na[4], nbv[4];
        if constexpr (QUANT) loadx(nxt, na, nbv);
        // the lane's group: slots 2 * lane, 2 * lane + 1 of the chunk
        const int qg = cur.base + 2 * lane;
        int chg, slg;
        bool ghere;
        if constexpr (ROWS) {
            chg = cur.cha;
            slg = qg;
            ghere = qg < nslots;
        } else {
            slot_split(min(qg, W - 2), nslots, inv_ns, nch, chg, slg);
            ghere = qg < W;
        }
        // ROWS: one channel per wave - its bit width lives in a scalar register and selects straight-line store code
        const int b = ROWS ? __builtin_amdgcn_readfirstlane(sh_b[chg]) : sh_b[chg];
        const int gi = slg >> 1;
        const uint32_t rowbytes = (((uint32_t)HW * (uint32_t)b + 31u) / 32u) * 4u;
        // bytes of this group: b whole bytes, or - last group of the row - everything up to the padded row end
        const uint32_t boff = (uint32_t)gi * (uint32_t)b;
        const uint32_t nb = ghere ? ((gi == ngroups - 1) ? rowbytes - boff : (uint32_t)b) : 0u;
        uint8_t* g = packed + (size_t)cur.n * plane + sh_off[chg] + boff;
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
            const unsigned ha = half_of(va, cur.cnta, cur.cha), hb = half_of(vb, cur.cntb, cur.chb);
            // group `lane` is in view A for lane < 32, else in view B
            const unsigned a0 = (unsigned)__shfl((int)ha, src, 64), b0 = (unsigned)__shfl((int)hb, src, 64);
            const unsigned a1 = (unsigned)__shfl((int)ha, src + 1, 64), b1 = (unsigned)__shfl((int)hb, src + 1, 64);
            const unsigned lo = lane < 32 ? a0 : b0, hi = lane < 32 ? a1 : b1;
            const unsigned long long ww = (unsigned long long)lo | ((unsigned long long)hi << (4 * b));
            // widest naturally aligned stores (the group starts at gi * b): 8 / 4 / 2-byte pieces where b allows
            auto store_any = [&](uint32_t nbytes) {
                if (nbytes == 8u && b == 8) *reinterpret_cast<unsigned long long*>(g) = ww;
                else if (nbytes == 4u && b == 4) *reinterpret_cast<uint32_t*>(g) = (uint32_t)ww;
                else if ((b & 1) == 0 && (nbytes & 1u) == 0u)
                    for (uint32_t kk = 0; kk < nbytes; kk += 2) *reinterpret_cast<uint16_t*>(g + kk) = (uint16_t)(kk < 8 ? (ww >> (8 * kk)) : 0ull);
                else
                    for (uint32_t kk = 0; kk < nbytes; ++kk) g[kk] = (uint8_t)(kk < 8 ? (ww >> (8 * kk)) : 0ull);
            };
            if constexpr (ROWS) {
                // every group but a row's last one is exactly b bytes: no loops, no per-lane width tests
                const bool plain = ghere && gi != ngroups - 1;
                if (plain) {
                    const unsigned wl = (unsigned)ww, wh = (unsigned)(ww >> 32);
                    switch (b) {   // uniform
                        case 8: *reinterpret_cast<unsigned long long*>(g) = ww; break;
                        case 4: *reinterpret_cast<uint32_t*>(g) = wl; break;
                        case 2: *reinterpret_cast<uint16_t*>(g) = (uint16_t)wl; break;
                        case 6:
                            *reinterpret_cast<uint16_t*>(g) = (uint16_t)wl;
                            *reinterpret_cast<uint16_t*>(g + 2) = (uint16_t)(wl >> 16);
                            *reinterpret_cast<uint16_t*>(g + 4) = (uint16_t)wh;
                            break;
                        case 7: g[6] = (uint8_t)(wh >> 16); [[fallthrough]];
                        case 5: g[4] = (uint8_t)wh; if (b == 7) g[5] = (uint8_t)(wh >> 8); g[3] = (uint8_t)(wl >> 24); [[fallthrough]];
                        case 3: g[2] = (uint8_t)(wl >> 16); g[1] = (uint8_t)(wl >> 8); [[fallthrough]];
                        case 1: g[0] = (uint8_t)wl; break;
                        default: break;   // 0 bits: the channel stores nothing
                    }
                } else {
                    store_any(nb);
                }
            } else {
                store_any(nb);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = na[e]; vb[e] = nbv[e]; }
        } else {
            float* ys = y + ((size_t)cur.n * C + c0) * (size_t)HW;
            unsigned long long w = 0;
            const uint32_t nr = nb < 8 ? nb : 8;
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
            emit(ma, cur.offa, cur.cnta, cur.cha);
            emit(mb, cur.offb, cur.cntb, cur.chb);
        }
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_pack_lean (round 3): the quantize + pack pass for the common geometry - whole float4 slots (H*W % 4 == 0), aligned x.
// k_packed<QUANT> is VALU-bound: ~350 vector instructions per 8 elements per lane, of which 8 IEEE divides + clamps +
// roundings are 130; the rest is bookkeeping - every lane splits a flat slot index into (channel, slot) twice per chunk,
// fetches three parameters per slot from LDS, tests widths per lane.  Here ONE WAVE owns one channel for a run of
// samples, so scale / zero point / qmax / width / row offset are scalars (s_load, no LDS, no barrier: the four waves of
// a workgroup take four ADJACENT channels of the same samples and never talk), and the lane -> (row, slot) map of a
// chunk is computed once per wave lifetime:
//   SHORT rows (<= 128 slots: 28x28 and smaller): a chunk = RPC whole rows of consecutive samples (lane q of the 128
//          slot positions = slot q % nsl of row q / nsl; nsl = slots per row rounded up to even);
//   long rows: a chunk = 128 consecutive slots of one row.
// Both views as in k_packed: lane L loads slots L and 64 + L (two coalesced 16-byte accesses), quantizes them to 4 * b
// code bits each, four wave shuffles hand lane G the two halves of group G (8 codes = b whole bytes), which it stores
// with straight-line code selected by the scalar width.  The next chunk's loads are issued before the current chunk is
// encoded.  Same bytes as k_packed<true, *> (tests compare the two forms).
#ifndef PACK_AHEAD
#define PACK_AHEAD 1
#endif
template <bool SHORT, bool RAG = false>
__global__ void __launch_bounds__(TPB) k_pack_lean(const float* __restrict__ x, uint8_t* __restrict__ packed, const int N,
                                                   const int C, const int HW, const int rpw, const float* __restrict__ qp,
                                                   const float* __restrict__ bits, const uint32_t* __restrict__ rowoff) {
    const int lane = threadIdx.x & 63;
    const int ncb = (C + 3) / 4;
    const int s = (int)blockIdx.x / ncb, cb = (int)blockIdx.x - s * ncb;
    const int c = __builtin_amdgcn_readfirstlane(cb * 4 + (int)(threadIdx.x >> 6));      // the wave's channel: a scalar
    if (c >= C) return;
    const int n0 = s * rpw, n1 = min(N, n0 + rpw);
    // RAG: rows of H*W % 4 != 0 elements (7x7): the row's last slot holds `tail` < 4 elements; it is loaded from the
    // row's last four elements (in bounds, 4-byte aligned like every slot of such a row) and its codes are shifted down
    const int nslots = RAG ? (HW + 3) / 4 : HW / 4, ngroups = (HW + 7) / 8, nsl = 2 * ngroups;
    const int tail = HW - 4 * (nslots - 1);
    const size_t P = (size_t)C * (size_t)HW;
    const int src = (2 * lane) & 63;
    // the lane's two slots and its group inside a chunk
    int ra = 0, rb = 0, rg = 0, sa, sb, gi;
    int rpc = 1, cpr = 1;
    if constexpr (SHORT) {
        rpc = 128 / nsl;
        ra = lane / nsl; sa = lane - ra * nsl;
        rb = (64 + lane) / nsl; sb = 64 + lane - rb * nsl;
        rg = lane / ngroups; gi = lane - rg * ngroups;
    } else {
        cpr = (nsl + 127) / 128;
        sa = lane; sb = 64 + lane; gi = lane;
    }
    const int nchunks = SHORT ? (n1 - n0 + rpc - 1) / rpc : (n1 - n0) * cpr;
    if (nchunks <= 0) return;
    // chunk i -> its first sample, the lane's slots / group in it
    auto chunk = [&](int i, int& nb, int& j) {
        if constexpr (SHORT) { nb = n0 + i * rpc; j = 0; }
        else { nb = n0 + i / cpr; j = i - (i / cpr) * cpr; }
    };
    auto load2 = [&](int i, float (&va)[4], float (&vb)[4], bool& oka, bool& okb) {
        int nb, j;
        const bool live = i < nchunks;                                       // the prefetch behind the last chunk: every lane
        chunk(live ? i : nchunks - 1, nb, j);                                // re-reads one 16-byte piece (a single request)
        const float* xs = x + ((size_t)nb * P + (size_t)c * (size_t)HW);     // uniform: the chunk's first row
        const int qa = sa + j * 128, qb = sb + j * 128;                      // slot within the row
        oka = live && qa < nslots && (SHORT ? (ra < rpc && nb + ra < n1) : true);
        okb = live && qb < nslots && (SHORT ? (rb < rpc && nb + rb < n1) : true);
        // unconditional loads (a branch around a load serialises the loads): dead lanes read the chunk's first slot
        if constexpr (RAG) {
            const unsigned ea = qa == nslots - 1 ? (unsigned)(HW - 4) : (unsigned)qa * 4u;
            const unsigned eb = qb == nslots - 1 ? (unsigned)(HW - 4) : (unsigned)qb * 4u;
            ldv4_nt_a4(xs + (oka ? (unsigned)ra * (unsigned)P + ea : 0u), va);
            ldv4_nt_a4(xs + (okb ? (unsigned)rb * (unsigned)P + eb : 0u), vb);
        } else {
            const unsigned oa = oka ? (unsigned)ra * (unsigned)P + (unsigned)qa * 4u : 0u;
            const unsigned ob = okb ? (unsigned)rb * (unsigned)P + (unsigned)qb * 4u : 0u;
            ldv_nt<4>(xs + oa, va);
            ldv_nt<4>(xs + ob, vb);
        }
    };
    float va[4], vb[4];
    bool oka, okb;
    load2(0, va, vb, oka, okb);
#if PACK_AHEAD == 2
    float ma[4], mb[4];                     // the chunk after it (the loads run two chunks ahead of the encoder)
    bool moka, mokb;
    load2(1, ma, mb, moka, mokb);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // the channel's parameters only now: the first chunk of x is already on its way while they arrive
    const int b = (int)bits[c];
    if (b == 0) return;                                                                   // a 0-bit channel stores nothing
    const float sc = qp[(size_t)CNNQ_QP_SCALE * C + c], zp = qp[(size_t)CNNQ_QP_ZP * C + c], qm = qp[(size_t)CNNQ_QP_QMAX * C + c];
    const uint32_t off_c = rowoff[c], plane = rowoff[C];
    const uint32_t rowbytes = (((uint32_t)HW * (uint32_t)b + 31u) / 32u) * 4u;
    const bool fastc = sc >= 0x1p-30f && sc <= 0x1p30f && fabsf(zp) <= 0x1p30f && qm >= 0.f && qm <= 255.f;
    const float rs = uniform_f(1.0f / sc);
    // Widths 3, 5, 6, 7: a group's b bytes are neither a power of two nor aligned, and storing them byte by byte costs
    // 3-5 store instructions of 1-byte pieces at a stride of b (PMC: 21 bytes per write request).  The chunk's groups
    // form one byte stream per row (long rows: 64 b bytes of one row; short rows: `rowbytes` per row, rpc rows), whose
    // start is 4-byte aligned: lane L stores DWORD L (and 64 + L) of it - fully coalesced 256-byte stores - after
    // fetching the two groups the dword straddles from the lanes that own them (ds_bpermute; the lane -> (row, dword,
    // source lane, shifts) map is fixed for the wave's lifetime).
    const bool xch = b == 3 || b == 5 || b == 6 || b == 7;                                // scalar
    int xsrc[2], xrow[2];
    unsigned xshA[2], xshB[2], xoff[2];
    bool xokA[2], xokB[2], xlive[2];
    if (xch) {
        const int dpr = (int)(rowbytes / 4u);                                             // dwords per row
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int d = lane + 64 * r;
            int rd = 0, dd = d;
            if constexpr (SHORT) { rd = d / dpr; dd = d - rd * dpr; }
            const int g0 = (4 * dd) / b, o = 4 * dd - g0 * b;
            xsrc[r] = (SHORT ? rd * ngroups : 0) + g0;
            xshA[r] = 8u * (unsigned)o;
            xshB[r] = 8u * (unsigned)(b - o);
            xokA[r] = SHORT ? g0 < ngroups : true;                                       // long rows: codes past the row are 0 already
            xokB[r] = (SHORT ? g0 + 1 < ngroups : g0 + 1 < 64) && b - o < 4;
            xlive[r] = SHORT ? rd < rpc : d < 16 * b;
            xrow[r] = rd;
            xoff[r] = 4u * (unsigned)dd;
        }
    }
    // The codes without the hardware divide (qdq1_fast, cnnq_qdq.hip.h) when the channel's parameters are inside its
    // domain (a scalar test) AND the chunk's eight values per lane are (a wave-uniform test per chunk: the sum of their
    // squares is at most 2^120 - every |x| <= 2^60, no NaN, no inf; one fma per element); otherwise qdq1.
    auto half_of = [&](const float (&v)[4], bool ok, bool fast, bool last) -> unsigned {
        unsigned cds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float cd;
            if (fast) (void)qdq1_fast(v[e], sc, rs, zp, qm, cd);
            else (void)qdq1(v[e], sc, zp, qm, cd);
            cds[e] = (unsigned)cd;
        }
        unsigned h = cds[0] | (cds[1] << b) | (cds[2] << (2 * b)) | (cds[3] << (3 * b));
        if constexpr (RAG) h = last ? h >> ((4 - tail) * b) : h;     // the row's last slot: its `tail` elements sit on top
        return ok ? h : 0u;
    };
    auto chunk_in_domain = [&](const float (&p)[4], const float (&q)[4]) -> bool {
        float t0 = p[0] * p[0], t1 = p[1] * p[1];
        t0 = __builtin_fmaf(p[2], p[2], t0); t1 = __builtin_fmaf(p[3], p[3], t1);
        t0 = __builtin_fmaf(q[0], q[0], t0); t1 = __builtin_fmaf(q[1], q[1], t1);
        t0 = __builtin_fmaf(q[2], q[2], t0); t1 = __builtin_fmaf(q[3], q[3], t1);
        return fastc && __builtin_amdgcn_ballot_w64(!(t0 + t1 <= 0x1p120f)) == 0ull;
    };
    for (int i = 0; i < nchunks; ++i) {
        float na[4], nbv[4];
        bool noka, nokb;
        load2(i + PACK_AHEAD, na, nbv, noka, nokb);                          // unconditional prefetch (past the end: one broadcast line)
        unsigned ha, hb;
        bool lasta = false, lastb = false;
        if constexpr (RAG) {
            int nbq, jq;
            chunk(i, nbq, jq);
            lasta = sa + jq * 128 == nslots - 1;
            lastb = sb + jq * 128 == nslots - 1;
        }
        if (chunk_in_domain(va, vb)) { ha = half_of(va, oka, true, lasta); hb = half_of(vb, okb, true, lastb); }
        else { ha = half_of(va, oka, false, lasta); hb = half_of(vb, okb, false, lastb); }
        const unsigned a0 = (unsigned)__shfl((int)ha, src, 64), b0 = (unsigned)__shfl((int)hb, src, 64);
        const unsigned a1 = (unsigned)__shfl((int)ha, src + 1, 64), b1 = (unsigned)__shfl((int)hb, src + 1, 64);
        const unsigned lo = lane < 32 ? a0 : b0, hi = lane < 32 ? a1 : b1;
        const unsigned long long ww = (unsigned long long)lo | ((unsigned long long)hi << (4 * b));
        int nb, j;
        chunk(i, nb, j);
        const int g_i = gi + j * 64;                                           // group within the row
        const bool ghere = g_i < ngroups && (SHORT ? (rg < rpc && nb + rg < n1) : true);
        if (xch) {
            const unsigned wl = (unsigned)ww, wh = (unsigned)(ww >> 32);
            const uint32_t cbase = SHORT ? 0u : (uint32_t)j * 64u * (uint32_t)b;                // the chunk's first byte in its row
            const uint32_t cbytes = SHORT ? rowbytes : min(64u * (uint32_t)b, rowbytes - cbase);   // bytes of the row from there
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && (SHORT ? rpc * (int)(rowbytes / 4u) : 16 * b) <= 64) break;          // uniform: one round is enough
                const unsigned al = (unsigned)__shfl((int)wl, xsrc[r], 64), ah = (unsigned)__shfl((int)wh, xsrc[r], 64);
                const unsigned bl = (unsigned)__shfl((int)wl, xsrc[r] + 1, 64);
                const unsigned long long A = xokA[r] ? ((unsigned long long)al | ((unsigned long long)ah << 32)) : 0ull;
                const unsigned dw = (unsigned)(A >> xshA[r]) | (xokB[r] ? bl << xshB[r] : 0u);
                if (xlive[r] && xoff[r] < cbytes && (SHORT ? nb + xrow[r] < n1 : true))
                    *reinterpret_cast<uint32_t*>(packed + (size_t)(nb + xrow[r]) * plane + off_c + cbase + xoff[r]) = dw;
            }
        } else if (ghere) {
            uint8_t* g = packed + (size_t)(nb + rg) * plane + off_c + (uint32_t)g_i * (uint32_t)b;
            if (g_i != ngroups - 1) {
                const unsigned wl = (unsigned)ww, wh = (unsigned)(ww >> 32);
                switch (b) {   // scalar
                    case 8: *reinterpret_cast<unsigned long long*>(g) = ww; break;
                    case 4: *reinterpret_cast<uint32_t*>(g) = wl; break;
                    case 2: *reinterpret_cast<uint16_t*>(g) = (uint16_t)wl; break;
                    case 6:
                        *reinterpret_cast<uint16_t*>(g) = (uint16_t)wl;
                        *reinterpret_cast<uint16_t*>(g + 2) = (uint16_t)(wl >> 16);
                        *reinterpret_cast<uint16_t*>(g + 4) = (uint16_t)wh;
                        break;
                    case 7: g[6] = (uint8_t)(wh >> 16); [[fallthrough]];
                    case 5: g[4] = (uint8_t)wh; if (b == 7) g[5] = (uint8_t)(wh >> 8); g[3] = (uint8_t)(wl >> 24); [[fallthrough]];
                    case 3: g[2] = (uint8_t)(wl >> 16); g[1] = (uint8_t)(wl >> 8); [[fallthrough]];
                    case 1: g[0] = (uint8_t)wl; break;
                    default: break;
                }
            } else {
                // the row's last group: everything up to the padded row end (<= b + 3 bytes), zeros beyond the codes
                const uint32_t nbytes = rowbytes - (uint32_t)g_i * (uint32_t)b;
                for (uint32_t kk = 0; kk < nbytes; ++kk) g[kk] = (uint8_t)(kk < 8 ? (ww >> (8 * kk)) : 0ull);
            }
        }
#pragma unroll
#if PACK_AHEAD == 2
        for (int e = 0; e < 4; ++e) { va[e] = ma[e]; vb[e] = mb[e]; ma[e] = na[e]; mb[e] = nbv[e]; }
        oka = moka; okb = mokb;
        moka = noka; mokb = nokb;
#else
        for (int e = 0; e < 4; ++e) { va[e] = na[e]; vb[e] = nbv[e]; }
        oka = noka;
        okb = nokb;
#endif
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_unpack_lean (round 3): the load direction of the same geometry as k_pack_lean - one channel per wave (scalar scale /
// zero point / width / row offset), chunks of 128 slots.  The chunk's bytes are read as DWORDS of the row's byte stream,
// lane L the dwords L and 64 + L (two coalesced loads whatever the width), parked in the wave's own 128-dword LDS strip;
// the lane that owns slot q then picks up the two dwords its 4 b code bits straddle with one ds_read2_b32, shifts
// (v_alignbit), extracts four codes (v_bfe_u32 with scalar width), dequantizes (code - zp) * scale (iq.py:591-592) and
// stores one 16-byte piece of y - fully coalesced on both sides, 5 vector operations per element, no per-lane width
// tests, no byte loads.  Waves never talk (no barrier: a wave's LDS operations execute in order).  Same floats as
// k_packed<false, *> (tests compare both with the fused Q/DQ).
template <bool SHORT, bool RAG = false>
__global__ void __launch_bounds__(TPB) k_unpack_lean(const uint8_t* __restrict__ packed, float* __restrict__ y, const int N,
                                                     const int C, const int HW, const int rpw, const float* __restrict__ qp,
                                                     const float* __restrict__ bits, const uint32_t* __restrict__ rowoff) {
    __shared__ uint32_t sh_dw[TPB / 64][132];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ncb = (C + 3) / 4;
    const int s = (int)blockIdx.x / ncb, cb = (int)blockIdx.x - s * ncb;
    const int c = cb * 4 + wv;                                                            // the wave's channel: a scalar
    if (c >= C) return;
    const int n0 = s * rpw, n1 = min(N, n0 + rpw);
    const int nslots = RAG ? (HW + 3) / 4 : HW / 4, ngroups = (HW + 7) / 8, nsl = 2 * ngroups;
    const int tail = HW - 4 * (nslots - 1);
    const size_t P = (size_t)C * (size_t)HW;
    const int b = (int)bits[c];                                                           // 0: every code is 0, nothing is read
    const float sc = qp[(size_t)CNNQ_QP_SCALE * C + c], zp = qp[(size_t)CNNQ_QP_ZP * C + c];
    const uint32_t off_c = rowoff[c], plane = rowoff[C];
    const uint32_t rowbytes = (((uint32_t)HW * (uint32_t)b + 31u) / 32u) * 4u;
    const int dpr = max(1, (int)(rowbytes / 4u));                                         // dwords per row
    const int cdw = 16 * b;                                                               // long rows: dwords per full chunk
    // the lane's two slots (as in k_pack_lean) and the two dwords it fetches, per chunk
    int ra = 0, rb = 0, sa, sb;
    int rpc = 1, cpr = 1;
    int lr[2] = {0, 0}, ld[2];                                                            // row and dword (within the row / chunk) of the lane's loads
    if constexpr (SHORT) {
        rpc = 128 / nsl;
        ra = lane / nsl; sa = lane - ra * nsl;
        rb = (64 + lane) / nsl; sb = 64 + lane - rb * nsl;
        lr[0] = lane / dpr; ld[0] = lane - lr[0] * dpr;
        lr[1] = (64 + lane) / dpr; ld[1] = 64 + lane - lr[1] * dpr;
    } else {
        cpr = (nsl + 127) / 128;
        sa = lane; sb = 64 + lane;
        ld[0] = lane; ld[1] = 64 + lane;
    }
    // the strip index of the first dword of the lane's slots and the bit offset inside it
    const int ta = min(130, (SHORT ? ra * dpr : 0) + (b * sa) / 8), tb = min(130, (SHORT ? rb * dpr : 0) + (b * sb) / 8);
    const unsigned sha = 4u * (unsigned)((b * sa) & 7), shb = 4u * (unsigned)((b * sb) & 7);
    const int nchunks = SHORT ? (n1 - n0 + rpc - 1) / rpc : (n1 - n0) * cpr;
    if (nchunks <= 0) return;
    auto chunk = [&](int i, int& nb, int& j) {
        if constexpr (SHORT) { nb = n0 + i * rpc; j = 0; }
        else { nb = n0 + i / cpr; j = i - (i / cpr) * cpr; }
    };
    auto loadd = [&](int i, uint32_t (&r)[2]) {
        int nb, j;
        const bool live = i < nchunks;
        chunk(live ? i : nchunks - 1, nb, j);
        const uint8_t* rowp = packed + (size_t)nb * plane + off_c;                        // uniform
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bool ok;
            uint32_t off;
            if constexpr (SHORT) {
                ok = live && lr[k] < rpc && nb + lr[k] < n1 && b > 0;
                off = (uint32_t)lr[k] * plane + 4u * (uint32_t)ld[k];
            } else {
                const uint32_t dd = (uint32_t)(j * cdw + ld[k]);
                ok = live && ld[k] < cdw && 4u * dd < rowbytes;
                off = 4u * dd;
            }
            // unconditional load (a branch around a load serialises the loads): dead lanes re-read the row's first dword
            r[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(rowp + (ok ? off : 0u)));
        }
    };
    uint32_t cur[2];
    if (b > 0) loadd(0, cur); else { cur[0] = 0u; cur[1] = 0u; }
    const unsigned cmask = b >= 8 ? 0xffu : ((1u << b) - 1u);
    for (int i = 0; i < nchunks; ++i) {
        uint32_t nxt[2] = {0u, 0u};
        if (b > 0) loadd(i + 1, nxt);                                                     // b is a scalar: no divergence
        sh_dw[wv][lane] = cur[0];
        sh_dw[wv][64 + lane] = cur[1];
        __builtin_amdgcn_wave_barrier();
        const unsigned alo = sh_dw[wv][ta], ahi = sh_dw[wv][ta + 1];
        const unsigned blo = sh_dw[wv][tb], bhi = sh_dw[wv][tb + 1];
        __builtin_amdgcn_wave_barrier();
        const unsigned ma = __builtin_amdgcn_alignbit(ahi, alo, sha), mb = __builtin_amdgcn_alignbit(bhi, blo, shb);
        int nb, j;
        chunk(i, nb, j);
        float* ys = y + ((size_t)nb * P + (size_t)c * (size_t)HW);                       // uniform: the chunk's first row
        const int qa = sa + j * 128, qb = sb + j * 128;
        const bool oka = qa < nslots && (SHORT ? (ra < rpc && nb + ra < n1) : true);
        const bool okb = qb < nslots && (SHORT ? (rb < rpc && nb + rb < n1) : true);
        auto emit = [&](unsigned m, bool ok, int r, int q) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((float)((m >> (e * b)) & cmask) - zp) * sc;
            if (!ok) return;
            float* dst = ys + (size_t)((unsigned)r * (unsigned)P + (unsigned)q * 4u);
            if constexpr (RAG) {
                if (q == nslots - 1) {
                    dst[0] = o[0];
                    if (tail > 1) dst[1] = o[1];
                    if (tail > 2) dst[2] = o[2];
                } else {
                    stv4_nt_a4(dst, o);
                }
            } else {
                stv_nt<4>(dst, o);
            }
        };
        emit(ma, oka, ra, qa);
        emit(mb, okb, rb, qb);
        cur[0] = nxt[0];
        cur[1] = nxt[1];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_unpack_flat (round 4): the load direction as ONE-SHOT workgroups in ADDRESS ORDER of y.  Measured this round
// (tools/ubench_rw.py): a write stream runs at 6.6-7.0 TB/s when consecutive workgroups write consecutive 4 KB and at
// 4.7-5.9 when a workgroup walks a 16-64 KB region in pieces (what one channel row per wave amounts to: four rows of
// 12.5 KB written 2 KB at a time), while the order of the reads is free.  Here workgroup g owns float4 [256 g, 256 g + 256)
// of the flattened y and every lane decodes its own float4: row (n, c) and slot q from the flat index (one scalar
// division pair per workgroup, a float reciprocal per lane: at most a few row wraps inside 256 slots), the channel's
// scale / zero point / width / row offset as per-lane loads of the small tables (L1-resident: a wave spans one or two
// channels when rows are long), the slot's 4 b bits from the one or two dwords of the row's stream that hold them
// (dword loads through the L1: 64 lanes read 32 b consecutive bytes), (code - zp) * scale (iq.py:591-592), one 16-byte
// non-temporal store.  RAG (H*W % 4 != 0, e.g. 7x7): a float4 of y may straddle rows, so everything is per ELEMENT.
// Same floats as k_unpack_lean / k_packed<false> (tests compare all forms with the fused Q/DQ).
template <int S, int U>
__global__ void __launch_bounds__(TPB) k_unpack_flat(const uint8_t* __restrict__ packed, float* __restrict__ y, const int N,
                                                     const int C, const int HW, const float* __restrict__ qp,
                                                     const float* __restrict__ bits, const uint32_t* __restrict__ rowoff,
                                                     const unsigned total4, const unsigned tail) {
    // A workgroup owns U * 256 consecutive float4 of y; lane t its float4 t, 256 + t, ... (U coalesced 4 KB stores).  The
    // three memory round trips of a float4 - channel tables, stream dwords, store - are issued U at a time: with one
    // float4 per lane the pass was latency-bound beyond the reach of the address-translation caches (3.9 TB/s on a
    // rotated set of more than 2 GB against 6 TB/s below it; U = 4: 5.2 and 7).
    // S = decode units per float4: 1 - rows are whole float4s (H*W % 4 == 0); 2 - ragged rows of at least 4 elements
    // (7x7): a float4 holds the end of one row and the start of the next, each a run of consecutive bits of its row's
    // stream; 4 - rows of fewer than 4 elements: every element is its own unit.
    static_assert(S == 1 || S == 2 || S == 4, "units per float4");
    const unsigned g0 = (unsigned)blockIdx.x * (TPB * U);
    const uint32_t plane = rowoff[C];
    constexpr int E = S * U;
    unsigned row[E], idx[E], cnt[E];            // row (n * C + c), first element inside the row, elements (0: dead unit)
    if constexpr (S == 1) {
        const unsigned cpc = (unsigned)HW / 4u;
        const unsigned row0 = g0 / cpc, q0 = g0 - row0 * cpc;
        const float rc = 1.0f / (float)cpc;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned t = q0 + (unsigned)u * TPB + threadIdx.x;            // < 2^24: the float quotient is exact +- 1
            unsigned w = (unsigned)((float)t * rc);
            w -= (w * cpc > t) ? 1u : 0u;
            w += ((w + 1u) * cpc <= t) ? 1u : 0u;
            const bool live = g0 + (unsigned)u * TPB + threadIdx.x < total4;
            row[u] = live ? row0 + w : 0u;
            idx[u] = live ? 4u * (t - w * cpc) : 0u;
            cnt[u] = live ? 4u : 0u;
        }
    } else {
        const double rh = 1.0 / (double)HW;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned g = g0 + (unsigned)u * TPB + threadIdx.x;
            const unsigned e0 = g * 4u;                                       // < 2^32 - 4096 (the launch checked)
            unsigned r0 = (unsigned)((double)e0 * rh);                        // exact +- 1 in double
            r0 -= ((unsigned long long)r0 * (unsigned)HW > e0) ? 1u : 0u;
            r0 += ((unsigned long long)(r0 + 1u) * (unsigned)HW <= e0) ? 1u : 0u;
            const unsigned i0 = e0 - r0 * (unsigned)HW;
            const unsigned have = g + 1u < total4 ? 4u : (g + 1u == total4 ? tail : 0u);      // elements of this float4 inside the tensor
            if constexpr (S == 2) {
                const unsigned ka = min(have, (unsigned)HW - i0);
                row[2 * u] = have ? r0 : 0u; idx[2 * u] = have ? i0 : 0u; cnt[2 * u] = ka;
                row[2 * u + 1] = have > ka ? r0 + 1u : 0u; idx[2 * u + 1] = 0u; cnt[2 * u + 1] = have - ka;
            } else {
                unsigned i = i0, r = r0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool live = (unsigned)e < have;
                    row[4 * u + e] = live ? r : 0u; idx[4 * u + e] = live ? i : 0u; cnt[4 * u + e] = live ? 1u : 0u;
                    ++i;
                    if (i == (unsigned)HW) { i = 0u; ++r; }
                }
            }
        }
    }
    // ---- round trip 1: the channels' tables (L1 / L2 resident)
    int b[E];
    float sc[E], zp[E];
    uint32_t roff[E];
    unsigned nn[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const unsigned n = row[k] / (unsigned)C, c = row[k] - n * (unsigned)C;
        nn[k] = n;
        b[k] = (int)bits[c];
        sc[k] = qp[(size_t)CNNQ_QP_SCALE * C + c];
        zp[k] = qp[(size_t)CNNQ_QP_ZP * C + c];
        roff[k] = rowoff[c];
    }
    // ---- round trip 2: the one or two dwords of the row's stream that hold the unit's bits
    uint32_t m[E], lo[E], hi[E];
    unsigned sh[E];
    bool anyk[E], twok[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t rowbytes = (((uint32_t)HW * (uint32_t)b[k] + 31u) / 32u) * 4u;
        const uint8_t* rowp = packed + (size_t)nn[k] * plane + roff[k];
        const unsigned bit = (unsigned)b[k] * idx[k], d = bit >> 5;
        // unconditional, clamped: a dead unit or a 0-bit channel (an empty row) reads the layout table instead of the
        // stream, a unit that ends in its row's last dword reads that dword twice
        const bool any = b[k] > 0 && cnt[k] > 0u, two = any && 4u * (d + 1u) < rowbytes;
        const uint8_t* pl = any ? rowp + 4u * d : reinterpret_cast<const uint8_t*>(rowoff);
        lo[k] = *reinterpret_cast<const uint32_t*>(pl);
        hi[k] = *reinterpret_cast<const uint32_t*>(two ? pl + 4 : pl);
        sh[k] = bit & 31u;
        anyk[k] = any;
        twok[k] = two;
    }
    // every unit's dwords in flight before the first is consumed (the scheduler otherwise waits for each pair in turn:
    // U dependent round trips instead of one)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < E; ++k) m[k] = anyk[k] ? __builtin_amdgcn_alignbit(twok[k] ? hi[k] : 0u, lo[k], sh[k]) : 0u;
    // ---- decode, (code - zp) * scale (iq.py:591-592), store
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned g = g0 + (unsigned)u * TPB + threadIdx.x;
        float o[4];
        unsigned have = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int k;
            unsigned pos;                                                     // which unit, which of its codes
            if constexpr (S == 1) { k = u; pos = (unsigned)e; }
            else if constexpr (S == 4) { k = 4 * u + e; pos = 0u; }
            else { const bool first = (unsigned)e < cnt[2 * u]; k = 2 * u; pos = first ? (unsigned)e : (unsigned)e - cnt[2 * u];
                   if (!first) {      // the second run of the float4: the next row's first codes
                       const unsigned cm = b[2 * u + 1] >= 8 ? 0xffu : ((1u << b[2 * u + 1]) - 1u);
                       o[e] = ((float)((m[2 * u + 1] >> (pos * (unsigned)b[2 * u + 1])) & cm) - zp[2 * u + 1]) * sc[2 * u + 1];
                       continue;
                   } }
            const unsigned cm = b[k] >= 8 ? 0xffu : ((1u << b[k]) - 1u);
            o[e] = ((float)((m[k] >> (pos * (unsigned)b[k])) & cm) - zp[k]) * sc[k];
        }
#pragma unroll
        for (int j = 0; j < S; ++j) have += cnt[S * u + j];
        if (have == 4u) {
            stv_nt<4>(y + (size_t)g * 4, o);
        } else if constexpr (S != 1) {            // the tensor's last, partial float4 (ragged rows only)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((unsigned)e < have) y[(size_t)g * 4 + e] = o[e];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// k_pack_flat (round 5): the store direction as ONE-SHOT workgroups in address order of x, the mirror of k_unpack_flat.
// k_pack_lean keeps one channel per wave and walks it chunk by chunk with one chunk of loads ahead (two 16-byte loads
// per lane in flight); here a workgroup owns U * 32 consecutive GROUPS (a group = 8 float4 slots of one row = 32 codes =
// bits[c] whole dwords of the row's stream, so every group starts on a dword of the output whatever the width) and issues
// all of its U loads per lane, and the gathers of the channels' tables, before the first is consumed.  Lane (g, j) = (tid / 8,
// tid % 8) quantizes slot j of its group to 4 b code bits and ORs them into the group's window of the wave's LDS strip
// (8 dwords per group; ds_or_b32, at most two per lane); lane (g, j) then takes dword j of the window back (an exchange with
// zero: the strip is clean for the next step) and stores it - groups of a row are consecutive dwords, so a wave's stores
// are one or two runs of consecutive dwords.  Waves never talk (a wave's LDS operations execute in order).  Rows of whole
// float4s only (H*W % 4 == 0); the other rows keep k_pack_lean.  Same bytes as k_packed<true> / k_pack_lean (tests compare).
// Measured (ResNet-50 b512, 53 tensors one by one, three boxes): 5.25-5.50 ms against k_pack_lean's 5.13-5.32 - the same
// rate from the opposite structure; U = 1 / 2 / 4 / 8: 6.76 / 5.74 / 5.25 / 5.63 ms.  The ablation builds (PACK_FLAT_ABL) say
// what the 4.9 TB/s is made of: without the stores 4.75 ms, with trivial codes instead of the quantization 4.59 ms, with both
// 3.60 ms (7.1 TB/s: the reads alone run at the chip's read-streaming rate) - ten percent each for the 0.5 B/elem of stores
// and for the ~10 vector operations per element of the exact quotient, neither hidden behind the other at this arithmetic
// intensity.  Kept as form 3 of cnnq_pc_quantize_packed_form; form 0 stays k_pack_lean.
#ifndef PACK_FLAT_ABL
#define PACK_FLAT_ABL 0      // development builds (timing, WRONG results): 1 - no stores, 2 - trivial codes instead of the quantization
#endif
template <int U>
__global__ void __launch_bounds__(TPB) k_pack_flat(const float* __restrict__ x, uint8_t* __restrict__ packed, const unsigned nrows,
                                                   const int C, const int HW, const unsigned ngroups, const unsigned total_groups,
                                                   const float* __restrict__ qp, const float* __restrict__ bits,
                                                   const uint32_t* __restrict__ rowoff) {
    __shared__ unsigned sh_strip[TPB];                     // [wave][group of the wave][8 dwords]
    const int tid = threadIdx.x;
    const unsigned j = (unsigned)tid & 7u;
    const unsigned nslots = (unsigned)HW / 4u;
    sh_strip[tid] = 0u;                                    // a lane's own wave reads and writes this entry only
    const unsigned G0 = (unsigned)blockIdx.x * (32u * U);  // the workgroup's first group (uniform: scalar divisions)
    const unsigned row0 = G0 / ngroups, gi0 = G0 - row0 * ngroups;
    const unsigned n0 = row0 / (unsigned)C, c0 = row0 - n0 * (unsigned)C;
    const float rg = 1.0f / (float)ngroups, rc = 1.0f / (float)C;
    const uint32_t plane = rowoff[C];
    unsigned cc[U], nn[U], gi[U];
    bool live[U], rowok[U];
    float v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned t = gi0 + (unsigned)u * 32u + ((unsigned)tid >> 3);     // < 2^24: the float quotient is exact +- 1
        unsigned w = (unsigned)((float)t * rg);
        w -= (w * ngroups > t) ? 1u : 0u;
        w += ((w + 1u) * ngroups <= t) ? 1u : 0u;
        gi[u] = t - w * ngroups;
        const unsigned cl = c0 + w;                                            // < C + 32 U
        unsigned dn = (unsigned)((float)cl * rc);
        dn -= (dn * (unsigned)C > cl) ? 1u : 0u;
        dn += ((dn + 1u) * (unsigned)C <= cl) ? 1u : 0u;
        cc[u] = cl - dn * (unsigned)C;
        nn[u] = n0 + dn;
        const unsigned slot = gi[u] * 8u + j;
        rowok[u] = row0 + w < nrows;                                           // the same for the 8 lanes of a group
        live[u] = rowok[u] && slot < nslots;
        // unconditional loads (a branch around a load serialises the loads): dead lanes read the tensor's first float4
        const size_t off = live[u] ? (size_t)(row0 + w) * (size_t)HW + (size_t)slot * 4 : (size_t)0;
        ldv_nt<4>(x + off, v[u]);
    }
    // the channels' tables (L1 / L2 resident; a wave spans one or two channels when rows are long)
    int b[U];
    float sc[U], zp[U], qm[U];
    uint32_t roff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned c = cc[u];
        b[u] = (int)bits[c];
        sc[u] = qp[(size_t)CNNQ_QP_SCALE * C + c];
        zp[u] = qp[(size_t)CNNQ_QP_ZP * C + c];
        qm[u] = qp[(size_t)CNNQ_QP_QMAX * C + c];
        roff[u] = rowoff[c];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned* strip = sh_strip + (tid & ~7);               // the group's window
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // the codes without the hardware divide (qdq1_fast) when every lane of the wave has its parameters and its four
        // values inside the domain (the test of k_pack_lean, per lane here: the sum of the squares is at most 2^120 - every
        // |x| <= 2^60, no NaN, no inf); otherwise qdq1 for the whole wave
        const float rs = 1.0f / sc[u];
        float t0 = v[u][0] * v[u][0], t1 = v[u][1] * v[u][1];
        t0 = __builtin_fmaf(v[u][2], v[u][2], t0);
        t1 = __builtin_fmaf(v[u][3], v[u][3], t1);
        const bool dom = sc[u] >= 0x1p-30f && sc[u] <= 0x1p30f && fabsf(zp[u]) <= 0x1p30f && qm[u] >= 0.f && qm[u] <= 255.f && t0 + t1 <= 0x1p120f;
        const bool fast = __builtin_amdgcn_ballot_w64(live[u] && b[u] > 0 && !dom) == 0ull;
        unsigned cds[4];
        if (PACK_FLAT_ABL & 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) cds[e] = __float_as_uint(v[u][e]) & 7u;
        } else if (fast) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float cd;
                (void)qdq1_fast(v[u][e], sc[u], rs, zp[u], qm[u], cd);
                cds[e] = (unsigned)cd;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float cd;
                (void)qdq1(v[u][e], sc[u], zp[u], qm[u], cd);
                cds[e] = (unsigned)cd;
            }
        }
        const unsigned ub = (unsigned)b[u];
        unsigned h = cds[0] | (cds[1] << ub) | (cds[2] << (2u * ub)) | (cds[3] << (3u * ub));      // 4 b <= 32 bits
        h = (live[u] && ub > 0u) ? h : 0u;
        const unsigned bit = j * 4u * ub, d0 = bit >> 5, s = bit & 31u;                            // inside the 32 b bits of the group
        const unsigned lo = h << s, hi = s ? h >> (32u - s) : 0u;
        if (lo) __hip_atomic_fetch_or(strip + d0, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (hi) __hip_atomic_fetch_or(strip + d0 + 1u, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned mine = __hip_atomic_exchange(strip + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // dword j of the group: dword gi * b + j of the row's stream, while inside the row's bytes
        const unsigned rowdw = ((unsigned)HW * ub + 31u) / 32u;
        if (rowok[u] && j < ub && gi[u] * ub + j < rowdw && (!(PACK_FLAT_ABL & 1) || mine == 0xdeadbeefu))
            *reinterpret_cast<uint32_t*>(packed + (size_t)nn[u] * plane + roff[u] + (size_t)(gi[u] * ub + j) * 4) = mine;
    }
}

}  // namespace
