// cnnq_resident.hip.h - config 2 (dynamic per-channel min/max -> scale / zero point -> Q/DQ, iq.py:409-451,557-603)
// in ONE launch that reads x ONCE: 8 bytes per element instead of the 12 of the statistics pass + Q/DQ pass chain,
// and one launch boundary instead of three.  Part of the single translation unit cnnq_kernels.hip.
//
// A workgroup owns k WHOLE channels for the WHOLE batch and keeps them in registers between the statistics and
// the Q/DQ, so no workgroup ever needs another one's result: no exchange, no barrier across workgroups, no
// workspace, nothing to re-arm.  The T lanes of the workgroup form a 2-D tile over (sample, column):
//
//     lane = rl * CL + cl      cl < CL = k*H*W/4 float4 columns (the channel block of one sample, contiguous),
//                              rl < RL = T / CL  row lanes; lane (rl, cl) holds samples rl, rl+RL, rl+2RL, ...
//
// i.e. K = ceil(N / RL) 16-byte loads per lane, all issued back to back (the whole tile is in flight at once),
// consumed by the min/max reduction in arrival order; per-channel extrema are finished in LDS, every lane picks
// up the scale / zero point of its column's channel, quantizes its registers sample by sample and streams y out
// (stores of sample j overlap the arithmetic of sample j+1).
//
// It applies when a channel block of the whole batch fits the register tile: RL*K >= N with K <= 32 (T <= 512) or
// K <= 16 (T = 1024) - at batch 64 every ResNet-50 layer with H*W <= 28*28.  Larger per-channel populations (56x56 and
// 112x112 at batch 64, everything at batch 512) take the kernels of cnnq_group.hip.h, in which the workgroups that
// share a channel exchange their {min, max} partials inside the launch (k_mmq_flat / k_mmq_group): also one launch and
// one read of x.  OUT selects the extra outputs shared by all three kernels (XOut in cnnq_qdq.hip.h).
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_qdq.hip.h"
#include "cnnq_xrank.hip.h"

namespace {

template <int A>
__device__ __forceinline__ void lane_acc(const float (&v)[4], float (&mn)[A], float (&mx)[A], bool& nan) {
    if constexpr (A == 1) {
        // (round 6: v_min3 / v_max3 on the raw registers, cnnq_common.hip.h - 6 instead of 12 instructions per float4)
        mn[0] = min3_raw(min3_raw(mn[0], v[0], v[1]), v[2], v[3]);
        mx[0] = max3_raw(max3_raw(mx[0], v[0], v[1]), v[2], v[3]);
        nan |= __builtin_isunordered(v[0], v[1]) | __builtin_isunordered(v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // a float4 straddles channels here: a NaN poisons only its own element's accumulator (sticky forms)
            mn[e] = pmin(mn[e], v[e]);
            mx[e] = pmax(mx[e], v[e]);
        }
    }
}

struct WGeo {
    int N, C, HW, P;
    int k;    // channels per workgroup
    int CL;   // float4 columns of a full channel block = k*HW/4
    int RL;   // row lanes (<= N)
};

template <int A, int T, int K, int OUT = 0, bool XR = false>
__global__ void __launch_bounds__(T) k_mmq_whole(const float* __restrict__ x, float* __restrict__ y, const WGeo g,
                                                 const int num_bits, const int positive, float* __restrict__ qp,
                                                 float* __restrict__ mm, const unsigned flags, const XOut xo = XOut{},
                                                 const XRank xr = XRank{}) {
    if constexpr (XR) xr_prologue(xr);                 // workgroup 0: the slots of the launch two back, the sequence mirror
    __shared__ float l_mn[T * A], l_mx[T * A];
    __shared__ float sh_rs[MAXCH];
    __shared__ int sh_slow;
    if (threadIdx.x == 0) sh_slow = 0;      // the barriers of the reductions come before its writers
    extern __shared__ unsigned cnnq_dyn_lds[];     // OUT == 1 with a histogram: 2^min(num_bits, 8) bins x HREP replicas, sized by the launch (xhist_lds_bytes)
    unsigned* const sh_hist = cnnq_dyn_lds;
    if constexpr (OUT == 1) {
        if (xo.hist) xhist_zero(sh_hist, 1 << (num_bits < 8 ? num_bits : 8));      // the barriers of the reductions below order it before the first count
    }
    __shared__ float sh_mn[MAXCH], sh_mx[MAXCH];   // per channel: extrema, then scale / zero point
    const int tid = threadIdx.x;
    const int c0 = (int)blockIdx.x * g.k;
    const int c1 = min(g.C, c0 + g.k);
    const int nch = c1 - c0;
    const int ncols = (int)(((int64_t)nch * g.HW) / 4);
    const int rl = tid / g.CL, cl = tid - rl * g.CL;
    const bool active = rl < g.RL && cl < ncols;
    const int rlc = active ? rl : 0, clc = active ? cl : 0;   // idle lanes shadow lane (0, 0); results discarded
    const size_t colbase = ((size_t)c0 * g.HW / 4 + clc) * 4;

    // ---- the tile: K 16-byte loads per lane, issued back to back; slots past the batch (N is rarely RL * K) issue no
    //      load and shadow the lane's first sample, which is neutral for min / max
    float v[K][4];
    ldv_nt<4>(x + (size_t)rlc * (size_t)g.P + colbase, v[0]);
#pragma unroll
    for (int j = 1; j < K; ++j) {
        const int n = rlc + j * g.RL;
        if (n < g.N) {
            ldv_nt<4>(x + (size_t)n * (size_t)g.P + colbase, v[j]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = v[0][e];
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // the whole tile in flight before the first value is consumed
    float mn[A], mx[A];
    bool nan = false;
#pragma unroll
    for (int a = 0; a < A; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
#pragma unroll
    for (int j = 0; j < K; ++j) lane_acc<A>(v[j], mn, mx, nan);
    if (A == 1 && nan) { mn[0] = NAN; mx[0] = NAN; }

    // ---- per-channel extrema: lanes -> LDS -> fold the row lanes onto row 0 -> one wave (or lane) per channel
#pragma unroll
    for (int a = 0; a < A; ++a) {
        l_mn[tid * A + a] = active ? mn[a] : INFINITY;
        l_mx[tid * A + a] = active ? mx[a] : -INFINITY;
    }
    __syncthreads();
    if (tid < ncols) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float tn = l_mn[tid * A + a], tx = l_mx[tid * A + a];
            for (int r = 1; r < g.RL; ++r) {
                tn = pmin(tn, l_mn[(r * g.CL + tid) * A + a]);
                tx = pmax(tx, l_mx[(r * g.CL + tid) * A + a]);
            }
            l_mn[tid * A + a] = tn;
            l_mx[tid * A + a] = tx;
        }
    }
    __syncthreads();
    const int epc = g.HW * A / 4;   // LDS entries per channel in row 0
    const int wv = tid >> 6, lane = tid & 63;
    if (epc <= 16) {
        for (int ch = tid; ch < nch; ch += T) {
            float tn = INFINITY, tx = -INFINITY;
            for (int e = ch * epc; e < (ch + 1) * epc; ++e) { tn = pmin(tn, l_mn[e]); tx = pmax(tx, l_mx[e]); }
            sh_mn[ch] = tn;
            sh_mx[ch] = tx;
        }
    } else {
        for (int ch = wv; ch < nch; ch += T / 64) {
            float tn = INFINITY, tx = -INFINITY;
            for (int e = ch * epc + lane; e < (ch + 1) * epc; e += 64) { tn = pmin(tn, l_mn[e]); tx = pmax(tx, l_mx[e]); }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { tn = pmin(tn, shfl_xor_f(tn, m)); tx = pmax(tx, shfl_xor_f(tx, m)); }
            if (lane == 0) { sh_mn[ch] = tn; sh_mx[ch] = tx; }
        }
    }
    __syncthreads();

    // ---- scale / zero point of the owned channels (iq.py:559-572)
    const float qm = qmax_of(num_bits);
    float p_sc = 0.f, p_zp = 0.f;
    if (tid < nch) {
        float cmn = sh_mn[tid], cmx = sh_mx[tid];
        if constexpr (XR) {
            (void)xr_merge(xr, c0 + tid, true, cmn, cmx);      // the batch is sharded: every rank's extrema (cnnq_xrank.hip.h)
            sh_mn[tid] = cmn;                                  // own entry only; read again by the domain test below
            sh_mx[tid] = cmx;
        }
        const float offset = positive ? 0.f : cmn;
        const float delta = cmx - offset;
        p_sc = delta / qm;
        p_sc = (p_sc < 1e-8f) ? 1e-8f : p_sc;
        p_zp = zero_point_of(offset, p_sc);
        const int c = c0 + tid;
        qp[(size_t)CNNQ_QP_SCALE * g.C + c] = p_sc;
        qp[(size_t)CNNQ_QP_ZP * g.C + c] = p_zp;
        qp[(size_t)CNNQ_QP_QMAX * g.C + c] = qm;
        if (mm) { mm[c] = cmn; mm[g.C + c] = cmx; }
    }
    if (tid < nch) {
        if (!qdq_fast_domain(sh_mn[tid], sh_mx[tid], p_sc) || (flags & MMQ_FLAG_IEEE_DIVIDE)) sh_slow = 1;   // any writer, same value
        sh_mn[tid] = p_sc;   // own entry only: no hazard with the reads above
        sh_mx[tid] = p_zp;
        sh_rs[tid] = 1.0f / p_sc;
    }
    __syncthreads();
    float sc[A], zp[A], rs[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const int ch = (int)(((unsigned)clc * 4u + (unsigned)a) / (unsigned)g.HW);
        sc[a] = sh_mn[ch];
        zp[a] = sh_mx[ch];
        rs[a] = sh_rs[ch];
    }
    const bool fast = !__builtin_amdgcn_readfirstlane(sh_slow);   // every channel of the workgroup inside qdq_fast_domain

    // ---- Q/DQ out of the registers, sample by sample
    unsigned nzp[A];
#pragma unroll
    for (int a = 0; a < A; ++a) nzp[a] = 0u;
    if (fast) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int n = rlc + j * g.RL;
            float o[4], cd[4];
            if constexpr (A == 1) {
                qdq4_fast(v[j], sc[0], rs[0], zp[0], qm, o, cd);              // two elements per instruction, the same bits (as k_mmq_flat)
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = qdq1_fast(v[j][e], sc[e], rs[e], zp[e], qm, cd[e]);
            }
            if (active && n < g.N)
                xstore<OUT, A>(xo, reinterpret_cast<char*>(y), xo.codes, xo.packed, ((size_t)n * (size_t)g.P + colbase) * 4, o, cd,
                               sh_hist, zp, nzp);
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int n = rlc + j * g.RL;
            float o[4], cd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = (A == 1 ? 0 : e);
                o[e] = qdq1(v[j][e], sc[a], zp[a], qm, cd[e]);
            }
            if (active && n < g.N)
                xstore<OUT, A>(xo, reinterpret_cast<char*>(y), xo.codes, xo.packed, ((size_t)n * (size_t)g.P + colbase) * 4, o, cd,
                               sh_hist, zp, nzp);
        }
    }
    if constexpr (OUT == 1) {
        if (xo.hist) xhist_flush<A>(sh_hist, xo.hist, 1 << (num_bits < 8 ? num_bits : 8), zp, nzp);
    }
}

}  // namespace
