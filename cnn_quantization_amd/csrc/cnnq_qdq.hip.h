// cnnq_qdq.hip.h - config-2 pipeline: exact min/max partials, parameter table, the fused per-channel Q/DQ, code entropy.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_params.hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// the core: fused per-channel quantize -> clamp -> round -> dequantize on native NCHW
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float qdq1(float x, float scale, float zp, float qmax, float& code) {
    float q = x / scale;         // IEEE divide (v_div_scale / v_rcp / v_fma.. / v_div_fixup)
    q = q + zp;                  // separately rounded (-ffp-contract=off)
    q = (q > qmax) ? qmax : q;   // compare+select keeps NaN like torch.clamp / torch.where
    q = (q < 0.f) ? 0.f : q;
    q = rintf(q);                // v_rndne_f32: half to even, as torch.round
    code = q;
    return (q - zp) * scale;
}

// The same function without the hardware divide, for callers that KNOW the channel's value range (the single-launch
// kernels hold the channel's exact extrema before they quantize).  IEEE x / s on gfx950 is v_div_scale x2, v_rcp (quarter
// rate), five fma, v_div_fmas, v_div_fixup; with rs = RN(1 / s) computed once per channel by the real divide, the quotient
// is the same two-correction sequence the hardware macro runs (q0 = x rs; r0 = x - s q0; q1 = q0 + r0 rs; r1 = x - s q1;
// q = q1 + r1 rs - every step one rounding, the remainders exact), minus the range scaling and the fix-up, which are what
// the domain below makes unnecessary:
//   * s in [1e-8, 2^30] and every |x| of the channel <= 2^70, no NaN (qdq_fast_domain on the channel's extrema);
//   * for 2^-70 <= |x| every intermediate is a normal number (|q| in [2^-100, 2^100], remainders multiples of
//     2^(ex-46) >= 2^-116), so q1 is a faithful and q the correctly rounded quotient (Markstein's theorem: rs is the
//     correctly rounded reciprocal) - bit-identical to x / s;
//   * for |x| < 2^-70 (zeros, denormals) the exact quotient and this one are both below 2^-39 in magnitude: q + zp rounds
//     to zp when zp != 0, and for zp == 0 (zp is never -0) both clamp / round to +0 - the same code and the same y.
// No NaN can occur inside the domain, so the clamp is one v_med3_f32 instead of two compare+select pairs.
// 10 VALU operations per element against 19 (of which one quarter-rate).  tests/test_fastdiv_cpu.py brute-forces the
// quotient against the C divide; the -m gpu parity tests compare whole tensors with the CPU restatement bit for bit.
constexpr unsigned MMQ_FLAG_TEST_HOOK = 1u;     // group kernels: skip the wait, recompute (tests)
constexpr unsigned MMQ_FLAG_IEEE_DIVIDE = 2u;   // every channel through the hardware divide (tests, A/B: CNNQ_IEEE_DIVIDE=1)
constexpr unsigned MMQ_FLAG_SLOTS = 16u;        // internal: the slot meeting (cnnq_group.hip.h), set by launch_group
constexpr unsigned MMQ_FLAG_COUNTERS = 32u;     // public (tests, A/B): the counter meeting of round 2 instead of the slot meeting
constexpr unsigned MMQ_FLAG_PK_PLAIN = 8u;      // OUT = 2 (development, CNNQ_PK_PLAIN=1): plain instead of non-temporal wide stores
constexpr unsigned MMQ_FLAG_PK_NARROW = 4u;     // OUT = 2: the packed buffer is not 16-byte aligned (or CNNQ_PK_NARROW=1): one 2-byte store per float4

__device__ __forceinline__ bool qdq_fast_domain(float cmn, float cmx, float scale) {
    return fabsf(cmn) <= 0x1p70f && fabsf(cmx) <= 0x1p70f && scale <= 0x1p30f;   // false for NaN / Inf extrema
}

__device__ __forceinline__ float qdq1_fast(float x, float scale, float rs, float zp, float qmax, float& code) {
    float q = x * rs;
    float r = __builtin_fmaf(-scale, q, x);
    q = __builtin_fmaf(r, rs, q);
    r = __builtin_fmaf(-scale, q, x);
    q = __builtin_fmaf(r, rs, q);
    q = q + zp;
    q = __builtin_amdgcn_fmed3f(q, 0.f, qmax);
    q = rintf(q);
    code = q;
    return (q - zp) * scale;
}

// Two elements per instruction (round 4): gfx950 issues v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 at the rate of their
// scalar forms, each half an IEEE fp32 operation with one rounding - the same bits as qdq1_fast, 6 of its 10 operations
// halved (the clamp and the rounding have no packed form).  s / rs / zp: the channel's parameters per element.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v qdq2_fast(f2v x, f2v s, f2v rs, f2v zp, float qmax, f2v& code) {
    f2v q = x * rs;
    f2v r = __builtin_elementwise_fma(-s, q, x);
    q = __builtin_elementwise_fma(r, rs, q);
    r = __builtin_elementwise_fma(-s, q, x);
    q = __builtin_elementwise_fma(r, rs, q);
    q = q + zp;
    q.x = rintf(__builtin_amdgcn_fmed3f(q.x, 0.f, qmax));
    q.y = rintf(__builtin_amdgcn_fmed3f(q.y, 0.f, qmax));
    code = q;
    return (q - zp) * s;
}

// a float4 of one channel: o / cd as qdq1_fast's four calls would leave them
__device__ __forceinline__ void qdq4_fast(const float (&x)[4], float scale, float rs, float zp, float qmax, float (&o)[4],
                                          float (&cd)[4]) {
    const f2v s2 = {scale, scale}, r2 = {rs, rs}, z2 = {zp, zp};
    f2v c0, c1;
    const f2v o0 = qdq2_fast(f2v{x[0], x[1]}, s2, r2, z2, qmax, c0);
    const f2v o1 = qdq2_fast(f2v{x[2], x[3]}, s2, r2, z2, qmax, c1);
    o[0] = o0.x; o[1] = o0.y; o[2] = o1.x; o[3] = o1.y;
    cd[0] = c0.x; cd[1] = c0.y; cd[2] = c1.x; cd[3] = c1.y;
}

__device__ __forceinline__ float uniform_f(float v) {   // a value every lane of the wave holds: into a scalar register
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Exact per-channel min / max for config 2 and the per-tensor paths.  Each workgroup writes ONE
// {min, max} pair per channel it owns into pmm[G][2][C] (plain stores, every (group, channel) entry
// written exactly once: no atomics, no initialisation, deterministic); k_minmax_params /
// k_minmax_reduce merge the G pairs with one wave per channel.  (Device-scope atomics into a shared
// table were tried first: ~160 K contended atomics per small layer cost ~50 us - see DESIGN.md.)
template <int VEC, int A, int J, bool NTL>
__global__ void __launch_bounds__(TPB) k_minmax(const float* __restrict__ x, const Geo g,
                                                float* __restrict__ pmm) {
    constexpr int NE = TPB * J * A;
    __shared__ float l_mn[NE], l_mx[NE];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    int col[J];
    bool ok[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
    }
    float mn[J][A], mx[J][A];
    bool nan[J];   // A == 1: a NaN seen by this lane's column j (v_min / v_max drop it; torch.min / max propagate it)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        nan[j] = false;
#pragma unroll
        for (int a = 0; a < A; ++a) { mn[j][a] = INFINITY; mx[j][a] = -INFINITY; }
    }
    const float* row = x + (size_t)b.n0 * (size_t)g.P;
    constexpr int NU = (J == 1) ? 4 : 2;  // samples in flight per lane
#pragma unroll NU
    for (int n = b.n0; n < b.n1; ++n, row += g.P) {
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv_sel<VEC, NTL>(row + (size_t)col[j] * VEC, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if constexpr (A == 1 && VEC == 4) {
                mn[j][0] = fminf(fminf(mn[j][0], fminf(v[j][0], v[j][1])), fminf(v[j][2], v[j][3]));
                mx[j][0] = fmaxf(fmaxf(mx[j][0], fmaxf(v[j][0], v[j][1])), fmaxf(v[j][2], v[j][3]));
                nan[j] |= __builtin_isunordered(v[j][0], v[j][1]) | __builtin_isunordered(v[j][2], v[j][3]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    mn[j][A == 1 ? 0 : e] = pmin(mn[j][A == 1 ? 0 : e], v[j][e]);
                    mx[j][A == 1 ? 0 : e] = pmax(mx[j][A == 1 ? 0 : e], v[j][e]);
                }
            }
        }
    }
    if constexpr (A == 1 && VEC == 4) {
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (nan[j]) { mn[j][0] = NAN; mx[j][0] = NAN; }
    }
    float* pn = pmm + (size_t)(2 * b.grp) * g.C;
    float* px = pn + g.C;
    const int wv = tid >> 6, lane = tid & 63;
    if (g.mode == 1) {
        float tn = INFINITY, tx = -INFINITY;
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (ok[j]) { tn = pmin(tn, mn[j][0]); tx = pmax(tx, mx[j][0]); }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { tn = pmin(tn, shfl_xor_f(tn, m)); tx = pmax(tx, shfl_xor_f(tx, m)); }
        if (lane == 0) { l_mn[wv] = tn; l_mx[wv] = tx; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < TPB / 64; ++i) { tn = pmin(tn, l_mn[i]); tx = pmax(tx, l_mx[i]); }
            pn[b.c0] = tn;
            px[b.c0] = tx;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const int e = (j * TPB + tid) * A + a;
                l_mn[e] = mn[j][a];
                l_mx[e] = mx[j][a];
            }
        }
    __syncthreads();
    const int epc = g.HW * A / VEC;
    if (epc <= 16) {
        for (int ch = b.c0 + tid; ch < b.c1; ch += TPB) {
            const int lo = (ch - b.c0) * epc;
            float tn = INFINITY, tx = -INFINITY;
            for (int e = lo; e < lo + epc; ++e) { tn = pmin(tn, l_mn[e]); tx = pmax(tx, l_mx[e]); }
            pn[ch] = tn;
            px[ch] = tx;
        }
        return;
    }
    for (int ch = b.c0 + wv; ch < b.c1; ch += TPB / 64) {
        const int lo = (ch - b.c0) * epc;
        float tn = INFINITY, tx = -INFINITY;
        for (int e = lo + lane; e < lo + epc; e += 64) { tn = pmin(tn, l_mn[e]); tx = pmax(tx, l_mx[e]); }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { tn = pmin(tn, shfl_xor_f(tn, m)); tx = pmax(tx, shfl_xor_f(tx, m)); }
        if (lane == 0) { pn[ch] = tn; px[ch] = tx; }
    }
}

// one wave64 per channel reduces the G {min, max} pairs of that channel: lanes stride over the
// groups (independent loads in flight), then a shuffle reduction - a serial loop over G in one
// thread cost 10-26 us per call (measured), this form ~3 us
__device__ __forceinline__ void reduce_pairs(const float* __restrict__ pmm, int G, int C, int c, float& mn, float& mx) {
    const int lane = threadIdx.x & 63;
    mn = INFINITY;
    mx = -INFINITY;
    for (int gi = lane; gi < G; gi += 64) {
        mn = pmin(mn, pmm[(size_t)(2 * gi) * C + c]);
        mx = pmax(mx, pmm[(size_t)(2 * gi + 1) * C + c]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { mn = pmin(mn, shfl_xor_f(mn, m)); mx = pmax(mx, shfl_xor_f(mx, m)); }
}

// pmm[G][2][C] -> out[2][C]: the rank-local extrema that ranks exchange (all_gather)
__global__ void __launch_bounds__(TPB) k_minmax_reduce(const float* __restrict__ pmm, int G, int C,
                                                       float* __restrict__ out) {
    const int c = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    float mn, mx;
    reduce_pairs(pmm, G, C, c, mn, mx);
    if ((threadIdx.x & 63) == 0) { out[c] = mn; out[C + c] = mx; }
}

// Code histogram (for the Shannon entropy of utils/entropy.py:6-17): 256 bins x 32 replicas in
// LDS (32 KB), replica = lane & 31: address % 32 == lane % 32, so the 32 lanes of an LDS service
// group always hit 32 different banks however skewed the codes are (measured with 8 replicas: 90 %
// of the LDS cycles were bank conflicts, ~18 cycles per atomic); flushed once per workgroup.
constexpr int HREP = 32;

// ---- extra outputs of the single-launch kernels (k_mmq_whole / k_mmq_group / k_mmq_flat) --------------------------
// OUT = 0: y only.  OUT = 1: y, plus the uint8 codes and / or the code histogram (iq.py:586-587: the codes exist in the
// reference only as the argument of shannon_entropy).  OUT = 2: the packed 4-bit codes INSTEAD of y (two per byte, the
// layout of k_q_pack4): 4.5 bytes per element in one launch - the stored format of SURVEY 8 f3 straight from x.
struct XOut {
    uint8_t* codes;               // OUT 1, may be null
    unsigned long long* hist;     // OUT 1, may be null: XHIST_REPLICAS tables of 256 bins; the workgroup flushes its LDS
                                  // table into table (blockIdx & 63) - same-address chains of a few hundred atomics
                                  // where one shared table would see 10^5; k_entropy_replicas sums and re-zeroes them
    uint8_t* packed;              // OUT 2
};
constexpr int XHIST_REPLICAS = 64;

// nbins = 2^num_bits (<= 256): only the bins the codes can reach are zeroed and flushed - 16 instead of 256 for int4,
// i.e. 2 KB of LDS traffic per workgroup instead of 64
// dynamic LDS of a launch that counts codes: the table the kernel zeroes and flushes, nothing when there is no histogram - a
// static 256-bin table (32 KB) cost the small-tile instances their occupancy ([512,512,7,7] with -me: 45 instead of 25 us)
inline size_t xhist_lds_bytes(int out, const void* hist, int nbins) { return (out == 1 && hist) ? (size_t)nbins * HREP * sizeof(unsigned) : 0; }
__device__ __forceinline__ void xhist_zero(unsigned* sh_hist, int nbins) {
    for (int i = threadIdx.x; i < nbins * HREP; i += blockDim.x) sh_hist[i] = 0u;
}
// the zero point (the code of x == 0, about half of a post-ReLU layer) is counted in a register
__device__ __forceinline__ void xhist_add(unsigned* sh_hist, float cd, float zp, unsigned& nzp) {
    if (cd == zp) ++nzp;
    else atomicAdd(&sh_hist[((unsigned)(int)cd & 255u) * HREP + (threadIdx.x & (HREP - 1))], 1u);
}
template <int A>
__device__ __forceinline__ void xhist_flush(unsigned* sh_hist, unsigned long long* hist, int nbins, const float (&zp)[A],
                                            const unsigned (&nzp)[A]) {
#pragma unroll
    for (int a = 0; a < A; ++a)
        if (nzp[a]) atomicAdd(&sh_hist[((unsigned)(int)zp[a] & 255u) * HREP + (threadIdx.x & (HREP - 1))], nzp[a]);
    __syncthreads();
    if ((int)threadIdx.x < nbins) {
        const int tid = threadIdx.x;
        unsigned tot = 0;
#pragma unroll 8
        for (int r = 0; r < HREP; ++r) tot += sh_hist[tid * HREP + ((r + tid) & (HREP - 1))];
        if (tot) atomicAdd(&hist[(size_t)(blockIdx.x & (XHIST_REPLICAS - 1)) * 256 + tid], (unsigned long long)tot);
    }
}
// four 4-bit codes of one float4 -> the 16-bit word of the packed stream (element 0 in the low nibble)
__device__ __forceinline__ uint16_t pack4_of(const float (&cd)[4]) {
    return (uint16_t)(((unsigned)cd[0] & 15u) | (((unsigned)cd[1] & 15u) << 4) | (((unsigned)cd[2] & 15u) << 8) |
                      (((unsigned)cd[3] & 15u) << 12));
}
// ... in the divide-free domain (qdq_fast_domain: the codes are integers 0 .. 15, no NaN): the word built in fp32 - c0 + 16 c1 +
// 256 c2 + 4096 c3 is exact (< 2^16) - three fmas and ONE conversion instead of four conversions, masks, shifts and ors: the
// same word in 4 instead of ~12 instructions (the packed output is a 4.5 B/elem launch: like config 4's statistics kernel it
// is short of issue slots, DESIGN.md section 5 note 6)
__device__ __forceinline__ uint16_t pack4_fast(const float (&cd)[4]) {
    const float p = __builtin_fmaf(cd[3], 4096.f, __builtin_fmaf(cd[2], 256.f, __builtin_fmaf(cd[1], 16.f, cd[0])));
    return (uint16_t)(unsigned)p;
}
// one float4 of results at byte offset `boff` (of the fp32 tensor) from the three bases: y, the codes (one byte per
// element: boff / 4) and the packed nibbles (boff / 8).  OFF is size_t, or unsigned when the bases are per-workgroup
// (uniform base + 32-bit lane offset is the addressing form the flat tiles load with)
template <int OUT, int A, typename OFF>
__device__ __forceinline__ void xstore(const XOut& xo, char* __restrict__ yb, uint8_t* __restrict__ cb, uint8_t* __restrict__ pb,
                                       OFF boff, const float (&o)[4], const float (&cd)[4], unsigned* sh_hist,
                                       const float (&zp)[A], unsigned (&nzp)[A]) {
    if constexpr (OUT != 2) stv_nt<4>(reinterpret_cast<float*>(yb + boff), o);
    if constexpr (OUT == 1) {
        if (xo.codes) {
            const uint32_t pk = ((uint32_t)cd[0] & 255u) | (((uint32_t)cd[1] & 255u) << 8) | (((uint32_t)cd[2] & 255u) << 16) |
                                (((uint32_t)cd[3] & 255u) << 24);
            __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(cb + boff / 4));
        }
        if (xo.hist) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xhist_add(sh_hist, cd[e], zp[A == 1 ? 0 : e], nzp[A == 1 ? 0 : e]);
        }
    }
    if constexpr (OUT == 2) {
        const unsigned pk = ((unsigned)cd[0] & 15u) | (((unsigned)cd[1] & 15u) << 4) | (((unsigned)cd[2] & 15u) << 8) |
                            (((unsigned)cd[3] & 15u) << 12);
        *reinterpret_cast<uint16_t*>(pb + boff / 8) = (uint16_t)pk;
    }
}

#ifndef QDQ_NT
#define QDQ_NT 3  // bit 0: non-temporal loads of x, bit 1: non-temporal stores of y
#endif

// pmm[G][2][C] -> qp[3][C] for config 2 (iq.py:409-424,559-572): delta = max - min (or max with a
// zero minimum), scale = max(delta / qmax, 1e-8), zero_point = round(0 - offset/scale).  One
// wave per channel; G is the groups of one tensor or the world size after the all_gather.
__global__ void __launch_bounds__(TPB) k_minmax_params(const float* __restrict__ pmm, int G, int C, int num_bits,
                                                       int positive, float* __restrict__ qp) {
    const int c = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    float mn, mx;
    reduce_pairs(pmm, G, C, c, mn, mx);
    if ((threadIdx.x & 63) != 0) return;
    const float offset = positive ? 0.f : mn;
    const float delta = mx - offset;
    const float qm = qmax_of(num_bits);
    float sc = delta / qm;
    sc = (sc < 1e-8f) ? 1e-8f : sc;
    qp[(size_t)CNNQ_QP_SCALE * C + c] = sc;
    qp[(size_t)CNNQ_QP_ZP * C + c] = zero_point_of(offset, sc);
    qp[(size_t)CNNQ_QP_QMAX * C + c] = qm;
}

// The fused Q/DQ.  Launched with MANY short workgroups in address order (about 14 KB of x each,
// see make_geo `fine`): measured on MI355X, read+write streaming runs at 6.1-6.7 TB/s this way
// against 5.4 TB/s when a workgroup walks 30+ samples (tools/ubench_copy.py, tools/split_probe.py).
// GATH (the multi-GPU form of config 2): there is no parameter table yet - `qp_in` holds the W gathered {min, max}
// records [W][2][C] of the ranks, and every workgroup derives scale / zero point of its own channels in its
// prologue with the arithmetic of k_minmax_params (min / max over the ranks are exact, so every workgroup and every
// rank gets the same bits); the first batch split also writes them to `qp_out`.  One launch boundary less behind
// every exchange.
struct GathArgs {
    int W, num_bits, positive;
    float* qp_out;
};
__device__ __forceinline__ void gathered_params(const float* __restrict__ rec, const GathArgs& ga, int C, int c, float& sc,
                                                float& zp, float& qm) {
    float mn = INFINITY, mx = -INFINITY;
    for (int r = 0; r < ga.W; ++r) {
        mn = pmin(mn, rec[(size_t)(2 * r) * C + c]);
        mx = pmax(mx, rec[(size_t)(2 * r + 1) * C + c]);
    }
    const float offset = ga.positive ? 0.f : mn;
    const float delta = mx - offset;
    qm = qmax_of(ga.num_bits);
    sc = delta / qm;
    sc = (sc < 1e-8f) ? 1e-8f : sc;
    zp = zero_point_of(offset, sc);
}

template <int VEC, int A, int J, bool CODES, bool HIST, bool GATH = false>
__global__ void __launch_bounds__(TPB) k_qdq(const float* __restrict__ x, float* __restrict__ y, const Geo g,
                                             const float* __restrict__ qp, uint8_t* __restrict__ codes,
                                             unsigned long long* __restrict__ hist, const GathArgs ga = GathArgs{}) {
    __shared__ float sh_sc[MAXCH], sh_zp[MAXCH], sh_qm[MAXCH];
    __shared__ unsigned sh_hist[HIST ? 256 * HREP : 1];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    if constexpr (HIST) {
        for (int i = tid; i < 256 * HREP; i += TPB) sh_hist[i] = 0u;
    }
    // stage this workgroup's channels once (coalesced), then every lane keeps its own in registers;
    // a workgroup that owns a slice of ONE channel reads its three parameters directly (uniform
    // address -> scalar loads) and needs neither LDS nor a barrier before it starts streaming
    const bool single = (g.mode == 1) && !HIST;
    float u_sc = 0.f, u_zp = 0.f, u_qm = 0.f;
    if constexpr (GATH) {
        const bool writer = b.n0 == 0 && ga.qp_out != nullptr && (g.mode != 1 || b.col0 == b.c0 * (g.HW / VEC));
        auto publish = [&](int c, float sc_, float zp_, float qm_) {
            ga.qp_out[(size_t)CNNQ_QP_SCALE * g.C + c] = sc_;
            ga.qp_out[(size_t)CNNQ_QP_ZP * g.C + c] = zp_;
            ga.qp_out[(size_t)CNNQ_QP_QMAX * g.C + c] = qm_;
        };
        if (single) {
            gathered_params(qp, ga, g.C, b.c0, u_sc, u_zp, u_qm);     // uniform: every lane computes the same
            if (writer && tid == 0) publish(b.c0, u_sc, u_zp, u_qm);
        } else {
            for (int i = tid; i < b.c1 - b.c0; i += TPB) {
                float sc_, zp_, qm_;
                gathered_params(qp, ga, g.C, b.c0 + i, sc_, zp_, qm_);
                sh_sc[i] = sc_; sh_zp[i] = zp_; sh_qm[i] = qm_;
                if (writer) publish(b.c0 + i, sc_, zp_, qm_);
            }
            __syncthreads();
        }
    } else {
        if (!single) {
            for (int i = tid; i < b.c1 - b.c0; i += TPB) {
                sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0 + i];
                sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * g.C + b.c0 + i];
                sh_qm[i] = qp[(size_t)CNNQ_QP_QMAX * g.C + b.c0 + i];
            }
            __syncthreads();
        } else {
            u_sc = qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0];
            u_zp = qp[(size_t)CNNQ_QP_ZP * g.C + b.c0];
            u_qm = qp[(size_t)CNNQ_QP_QMAX * g.C + b.c0];
        }
    }
    int col[J];
    bool ok[J];
    float sc[J][A], zp[J][A], qm[J][A];
    // histogram: the code of x == 0 (the zero point) is by far the most frequent one (about half
    // of a half-range layer); counting it in a register per lane instead of an LDS atomic removes
    // the same-address serialisation that otherwise doubles the kernel time
    unsigned nzp[HIST ? J : 1][HIST ? A : 1];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if (single) {
                sc[j][a] = u_sc; zp[j][a] = u_zp; qm[j][a] = u_qm;
            } else {
                const unsigned e = (unsigned)col[j] * VEC + (A == 1 ? 0 : a);
                const int ch = (int)(e / (unsigned)g.HW) - b.c0;
                sc[j][a] = sh_sc[ch];
                zp[j][a] = sh_zp[ch];
                qm[j][a] = sh_qm[ch];
            }
            if constexpr (HIST) nzp[j][a] = 0u;
        }
    }
    const int nrows = b.n1 - b.n0;
    constexpr int NU = (J == 1) ? 4 : 2;  // samples in flight per lane
#pragma unroll NU
    for (int r = 0; r < nrows; ++r) {
        const int n = g.rev ? (b.n1 - 1 - r) : (b.n0 + r);
        const size_t off = (size_t)n * (size_t)g.P;
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if constexpr ((QDQ_NT & 1) != 0) ldv_nt<VEC>(x + off + (size_t)col[j] * VEC, v[j]);
            else ldv<VEC>(x + off + (size_t)col[j] * VEC, v[j]);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float o[VEC], cd[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                o[e] = qdq1(v[j][e], sc[j][a], zp[j][a], qm[j][a], cd[e]);
            }
            if (ok[j]) {
                if constexpr ((QDQ_NT & 2) != 0) stv_nt<VEC>(y + off + (size_t)col[j] * VEC, o);
                else stv<VEC>(y + off + (size_t)col[j] * VEC, o);
                if constexpr (HIST) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int a = (A == 1 ? 0 : e);
                        if (cd[e] == zp[j][a]) ++nzp[j][a];
                        else atomicAdd(&sh_hist[((unsigned)(int)cd[e] & 255u) * HREP + (tid & (HREP - 1))], 1u);
                    }
                }
                if constexpr (CODES) {
                    uint8_t* cp = codes + off + (size_t)col[j] * VEC;
                    if constexpr (VEC == 4) {
                        const uint32_t pk = (uint32_t)cd[0] | ((uint32_t)cd[1] << 8) | ((uint32_t)cd[2] << 16) |
                                            ((uint32_t)cd[3] << 24);
                        *reinterpret_cast<uint32_t*>(cp) = pk;
                    } else {
                        *cp = (uint8_t)cd[0];
                    }
                }
            }
        }
    }
    if constexpr (HIST) {
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int a = 0; a < A; ++a)
                if (nzp[j][a]) atomicAdd(&sh_hist[((unsigned)(int)zp[j][a] & 255u) * HREP + (tid & (HREP - 1))], nzp[j][a]);
        __syncthreads();
        unsigned tot = 0;
#pragma unroll 8
        for (int r = 0; r < HREP; ++r) tot += sh_hist[tid * HREP + ((r + tid) & (HREP - 1))];
        if (tot) atomicAdd(&hist[tid], (unsigned long long)tot);
    }
}

// Shannon entropy (bits) of a histogram: -sum p log2 p over the non-empty bins
__global__ void __launch_bounds__(TPB) k_entropy(const unsigned long long* __restrict__ hist, int nbins,
                                                 float* __restrict__ out) {
    __shared__ double sh[TPB / 64];
    __shared__ double sh_total;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    double t = 0.;
    for (int i = tid; i < nbins; i += TPB) t += (double)hist[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += shfl_xor_d(t, m);
    if (lane == 0) sh[wv] = t;
    __syncthreads();
    if (tid == 0) sh_total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    const float total = (float)sh_total;
    double e = 0.;
    for (int i = tid; i < nbins; i += TPB) {
        const unsigned long long c = hist[i];
        if (c) {
            const float pr = (float)c / total;
            e += (double)(-pr * log2f(pr));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) e += shfl_xor_d(e, m);
    __syncthreads();
    if (lane == 0) sh[wv] = e;
    __syncthreads();
    if (tid == 0) out[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// entropy of XHIST_REPLICAS replica tables (the histogram output of the single-launch kernels): fold them, hand the
// folded table to the same arithmetic as k_entropy, and leave the tables zero for the next launch
// One workgroup per histogram (round 6: the tables of a whole forward's tensors in ONE launch at its end - nothing consumes an
// entropy mid-forward, iq.py:445 logs it - instead of a dependent 8-25 us launch behind every tensor): workgroup b takes the
// tables at rep + b * XHIST_REPLICAS * 256 and writes out[b].
__global__ void __launch_bounds__(TPB) k_entropy_replicas(unsigned long long* __restrict__ rep, float* __restrict__ out) {
    rep += (size_t)blockIdx.x * XHIST_REPLICAS * 256;
    out += blockIdx.x;
    __shared__ unsigned long long bins[256];
    __shared__ double sh[TPB / 64];
    __shared__ double sh_total;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    unsigned long long c = 0;
    for (int r = 0; r < XHIST_REPLICAS; ++r) {
        c += rep[(size_t)r * 256 + tid];
        rep[(size_t)r * 256 + tid] = 0ull;
    }
    bins[tid] = c;
    double t = (double)c;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += shfl_xor_d(t, m);
    if (lane == 0) sh[wv] = t;
    __syncthreads();
    if (tid == 0) sh_total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    const float total = (float)sh_total;
    double e = 0.;
    if (c) {
        const float pr = (float)c / total;
        e = (double)(-pr * log2f(pr));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) e += shfl_xor_d(e, m);
    __syncthreads();
    if (lane == 0) sh[wv] = e;
    __syncthreads();
    if (tid == 0) out[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// the replica tables folded into one plain 256-bin table (added to it), the replicas left zero
__global__ void __launch_bounds__(TPB) k_hist_replicas_fold(unsigned long long* __restrict__ rep, unsigned long long* __restrict__ hist) {
    static_assert(TPB == 256, "one thread per bin");
    const int tid = threadIdx.x;
    unsigned long long c = 0;
    for (int r = 0; r < XHIST_REPLICAS; ++r) {
        c += rep[(size_t)r * 256 + tid];
        rep[(size_t)r * 256 + tid] = 0ull;
    }
    hist[tid] += c;
}

}  // namespace
