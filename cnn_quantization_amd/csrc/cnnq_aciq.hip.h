// cnnq_aciq.hip.h - the ACIQ (Laplace clipping, optional bit allocation) path of config 3 with the register-resident
// tiles of cnnq_group.hip.h: pass B (sum |x - mean|), the parameter derivation and the Q/DQ in ONE launch and ONE read of
// x - 12 instead of 16 bytes per element for iq.py:327-352 -> :227-253, 284-300, 393-407 -> :409-451, 557-603.
// Part of the single translation unit cnnq_kernels.hip.
//
// What a channel's clipping value needs is b = mean |x - mean| (iq.py:545: the mean first, hence two passes), and with
// -baa the bit allocation, which couples ALL channels through sum std^(2/3) (iq.py:381-391).  So pass A stays a launch
// of its own (k_moments: min / max / sum / sum of squares, 4 bytes per element at the read-streaming rate), a merge
// (k_combine) and - with bit allocation on the default 'gaus' prior, which needs nothing but the std of pass A - the
// one-workgroup k_bitalloc follow; then the kernels below do for pass B what k_mmq_flat / k_mmq_group do for the
// extrema of config 2: a workgroup loads its tile of x into registers, reduces it to sum |x - mean| per channel, meets
// the other workgroups that hold pieces of the same channels (the slot meeting: a member's partial sum IS its arrival),
// derives the channel's alpha / delta / offset / scale / zero point with the arithmetic of k_params
// (channel_params: the same function) and quantizes out of its registers.  A single launch for the whole of config 3
// cannot exist: no parameter of any channel is known before every channel's std is, i.e. before all of x has been read
// once, and 1.6 GB do not fit the register files.
//
// The same structure serves the mid-tread path of config 5 (MODE 1; iq.py:185-225): b is again the only statistic that needs
// the second pass, the bin allocation (omega, from the std of pass A alone: k_mt_params<GUESS>) plays the part of the bit
// allocation, and the launch derives delta / c_min / c_max (mt_channel: the function k_mt_params runs), quantizes, clamps,
// dequantizes and counts the codes into the windowed histogram of cnnq_midtread.hip.h.
//
// Channels too populous for the plain register tiles (VGG-16 b512: [512,64,224,224] holds 103 MB per channel, the 768
// resident K = 32 tiles 98 MB) take eight more rows per tile in LDS (KL = 8: LDS-DMA fills, 160 KB per workgroup, 628 members
// per channel, one channel resident at a time) - the tile shape k_mmq_flat has as a development knob, planned here whenever
// nothing smaller fits.
//
// Sums, unlike extrema, depend on the order of the additions.  Everything here adds in an order that is a function of
// the geometry alone: a lane adds its steps in order (fp64 accumulation of fp32 |x - mean|, as k_absdev), lanes fold by
// a fixed xor tree, waves in index order, and the members' partial sums are added in MEMBER order by every member -
// only after all of them have arrived, never in arrival order - so all members of a group hold the same b, bit for
// bit, run after run.  The cold path (a wait expired / the test hook) recomputes the partial sum of EVERY member with
// that member's own lane mapping and folds them the same way: the same bits again, which the min / max kernels get for
// free from exactness.
//
// Round 6 - XR = true: the batch is sharded over W GPUs (cnnq_xrank.hip.h).  Pass A's records travel through the collective as
// before (the merged table in fa.stats is the GLOBAL batch's, fa.count its element count per channel); the partial sums of
// |x - mean| of the ranks meet INSIDE this launch: after the local meeting one thread per channel pushes the rank's sum into every
// rank's window (member 0 only) and adds the W ranks' sums in rank order - every member of every rank derives the same b, hence the
// same parameters, bit for bit, and x is still read once (12 bytes per element sharded too, where the chain reads it three times
// around two collectives).
#pragma once
#include <type_traits>
#include "cnnq_common.hip.h"
#include "cnnq_group.hip.h"
#include "cnnq_midtread.hip.h"
#include "cnnq_params.hip.h"
#include "cnnq_qdq.hip.h"

namespace {

// The bit allocation alone (the fixed-target iteration of k_params): what the single-launch kernels below need from the
// whole table before they start.  (Tried and dropped: the merge of the pass-A records and this allocation in ONE launch -
// every workgroup merges its channels, the last one to take a ticket allocates, Guideline-16 counter hand-off - ran the b512
// forward in 13.28 / 13.55 / 13.56 ms against 13.39 / 13.32 / 13.31 ms for the two launches on one box: the release fence
// and the serial tail cost what the launch boundary does.)
__global__ void __launch_bounds__(PTPB) k_bitalloc(const float* __restrict__ prior, int C, const cnnq_params_cfg cfg,
                                                   float* __restrict__ bits_ws) {
    __shared__ double sh[PTPB / 64];
    bit_alloc_block(prior, C, cfg, bits_ws, sh);
}

// (slot_of_sum / sum_of_slot - a partial sum as a slot word, complemented, NaNs canonical - live in cnnq_xrank.hip.h since round 6)

// MODE 0: ACIQ clipping + GEMMLOWP Q/DQ (config 3).  MODE 1: mid-tread quantization with bin allocation (config 5).
struct FusedArgs {
    float* stats;            // [CNNQ_NSTAT][C]: rows MIN, MAX, MEAN, STD from pass A; row B is written here
    double count;            // N * H*W: elements per channel
    const double* count_dev; // XR: [C] the GLOBAL batch's elements per channel (row COUNT of the merged moment record: the shards'
                             // sizes are known to the device, not to this rank's host)
    // MODE 0
    const float* bits;       // [C] allocated widths (k_bitalloc), or null: cfg.num_bits everywhere
    float* qp;               // [CNNQ_NQP][C] out
    float* diag;             // [CNNQ_NDIAG][C] out, may be null
    cnnq_params_cfg cfg;
    // MODE 1
    float* mt;               // [CNNQ_NMT][C]: rows OMEGA, ALPHA, WSTART from k_mt_params<GUESS>; DELTA, CMIN, CMAX written here
    MtCfg mcfg;
    unsigned long long* hist;   // CNNQ_MT_HIST_WORDS(C), zeroed by the caller; may be null (OUT = 0)
};

// the four |x - mean| of one float4 of ONE channel added to the lane's sum (steps past the tile add nothing)
__device__ __forceinline__ void absdev_step(const float (&v)[4], float mean, bool valid, double& sa) {
    double t = (double)fabsf(v[0] - mean);
    t += (double)fabsf(v[1] - mean);
    t += (double)fabsf(v[2] - mean);
    t += (double)fabsf(v[3] - mean);
    sa += valid ? t : 0.;
}

// the workgroup's sum of one value per lane: xor tree per wave, the four waves in index order; every lane gets it
__device__ __forceinline__ double wg_sum1(double v, double* l_s) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    __syncthreads();                     // l_s may still be read from a previous call
    if (lane == 0) l_s[wv] = v;
    __syncthreads();
    return ((l_s[0] + l_s[1]) + l_s[2]) + l_s[3];
}

// The slot meeting of k_fused_flat (slots_meet of cnnq_group.hip.h for sums): wave 0 stores the member's partial sum and
// polls the group's slots; lane l watches members l, l + 64, ... in windows of 4 and adds a window's values in member
// order once the whole window has arrived.  Returns 0, or 1 (a wait expired) / 2 (the test hook), meaningful in thread 0;
// tsum: the lane's share (0 outside wave 0).
__device__ __forceinline__ int slots_meet_sum(unsigned long long* slots, int member, int Gs, double msum, unsigned flags,
                                              long long timeout_ticks, double& tsum) {
    const int tid = threadIdx.x;
    tsum = 0.;
    if (tid >= 64) return 0;
    if (tid == 0) __hip_atomic_store(slots + member, slot_of_sum(msum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (flags & MMQ_FLAG_TEST_HOOK) return 2;
    long long t0 = 0;
    int spins = 0;
    for (int w0 = 0; w0 * 64 < Gs; w0 += 4) {
        const unsigned long long* p = slots + tid + 64 * w0;
        unsigned pend = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) pend |= (tid + 64 * (w0 + i) < Gs) ? (1u << i) : 0u;
        const unsigned mine = pend;
        unsigned long long v[4] = {0ull, 0ull, 0ull, 0ull};
        for (;; ++spins) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((pend >> i) & 1u) {
                    v[i] = __hip_atomic_load(p + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v[i]) pend &= ~(1u << i);
                }
            if (__ballot(pend != 0u) == 0ull) break;
            int expired = 0;
            if ((spins & 31) == 31 || spins > GRP_TIMEOUT_SPINS) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                expired = (now - t0 > timeout_ticks || spins > GRP_TIMEOUT_SPINS) ? 1 : 0;
            }
            if (__builtin_amdgcn_readfirstlane(expired)) return 1;
            if (spins < 2) __builtin_amdgcn_s_sleep(8);
            else if (spins < 6) __builtin_amdgcn_s_sleep(32);
            else __builtin_amdgcn_s_sleep(64);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((mine >> i) & 1u) tsum += sum_of_slot(v[i]);      // member order, whatever the arrival order was
    }
    return 0;
}

// the same fold over plain member sums (the cold path's table, uncached memory): identical order of additions
__device__ __forceinline__ double fold_member_sums(const unsigned long long* ms, int Gs) {
    const int tid = threadIdx.x;
    double tsum = 0.;
    if (tid < 64)
        for (int w0 = 0; w0 * 64 < Gs; w0 += 4)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tid + 64 * (w0 + i) < Gs)
                    tsum += __longlong_as_double((long long)__hip_atomic_load(ms + tid + 64 * (w0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return tsum;
}

// ---- mid-tread arithmetic of the single-launch kernels (MODE 1) ---------------------------------------------------------
// qdq_fast_domain for parameters that come out of the clipping / bit-allocation arithmetic: finite extrema do not make zp and
// qmax numbers there (an Inf or NaN activation in ANOTHER channel turns the allocation's sum, and with it every channel's
// bit width, into NaN), and qdq1_fast's single-instruction clamp is the general path's compare + select only without NaN.
__device__ __forceinline__ bool aciq_fast_domain(float vmin, float vmax, const ChanParams& cp) {
    return qdq_fast_domain(vmin, vmax, cp.scale) && cp.zp == cp.zp && cp.qmax == cp.qmax;
}

// x / delta without the hardware divide, for a channel inside the domain of qdq1_fast (finite extrema up to 2^70, delta in
// [1e-8, 2^30]): the correctly rounded quotient from the channel's correctly rounded reciprocal and two fma corrections (the
// proof in cnnq_qdq.hip.h).  There is no zero point to absorb the sign of a vanishing quotient here, so it is taken from x:
// the exact quotient of a non-zero x is non-zero with x's sign (delta > 0), and -0 / delta = -0.
__device__ __forceinline__ float mt_quot_fast(float x, float d, float rd) {
    float q = x * rd;
    float r = __builtin_fmaf(-d, q, x);
    q = __builtin_fmaf(r, rd, q);
    r = __builtin_fmaf(-d, q, x);
    q = __builtin_fmaf(r, rd, q);
    return copysignf(q, x);
}
__device__ __forceinline__ bool mt_fast_domain(float vmin, float vmax, float delta) {
    return fabsf(vmin) <= 0x1p70f && fabsf(vmax) <= 0x1p70f && delta >= 1e-8f && delta <= 0x1p30f;   // false for NaN
}
// one element: code (iq.py:202-214) and dequantized value (iq.py:224)
template <bool FAST>
__device__ __forceinline__ float mt_qdq1(float x, float d, float rd, float lo, float hi, float& code) {
    float t = rintf(FAST ? mt_quot_fast(x, d, rd) : x / d);
    // torch.min(t, hi) = t < hi ? t : hi and torch.max(t, lo) = t > lo ? t : lo, NaN kept (the bound wins ties)
    t = (t < hi || t != t) ? t : hi;
    t = (t > lo || t != t) ? t : lo;
    code = t;
    return t * d;
}
// a float4 of ONE channel in the divide-free domain (no NaN, no inf): the quotient two elements per instruction (the packed fp32
// forms issue at the rate of the scalar ones, each half one IEEE operation: the same bits as mt_quot_fast), and the clamps without
// their NaN tests - 9 instead of 16 vector operations per element; the store phase of these kernels has no slack for them
__device__ __forceinline__ void mt_qdq4_fast(const float (&x)[4], float d, float rd, float lo, float hi, float (&o)[4], float (&cd)[4]) {
    const f2v d2 = {d, d}, r2 = {rd, rd};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2v xv = {x[2 * h], x[2 * h + 1]};
        f2v q = xv * r2;
        f2v r = __builtin_elementwise_fma(-d2, q, xv);
        q = __builtin_elementwise_fma(r, r2, q);
        r = __builtin_elementwise_fma(-d2, q, xv);
        q = __builtin_elementwise_fma(r, r2, q);
        float t0 = rintf(copysignf(q.x, xv.x)), t1 = rintf(copysignf(q.y, xv.y));
        t0 = t0 < hi ? t0 : hi;          // torch.min / torch.max: the bound wins ties (no NaN in this domain)
        t1 = t1 < hi ? t1 : hi;
        t0 = t0 > lo ? t0 : lo;
        t1 = t1 > lo ? t1 : lo;
        cd[2 * h] = t0;
        cd[2 * h + 1] = t1;
        const f2v y = f2v{t0, t1} * d2;
        o[2 * h] = y.x;
        o[2 * h + 1] = y.y;
    }
}
// LDS copies of a window bin in the single-launch kernels: one per lane of a 32-lane LDS service group (bank = lane % 32
// whatever the code: conflict-free by construction, as the code table of config 2) - 16 KB per workgroup, affordable because
// these workgroups are long-lived (k_mt_qdq's short tiles keep MT_REP = 8: 4 KB to zero and flush per 56 KB of x)
constexpr int MTF_REP = 32;
#ifndef MT_CNT_ABL
#define MT_CNT_ABL 0      // development builds (timing, WRONG histograms): bit 0 - no counting, bit 1 - no flush, bit 2 - no LDS atomic (the arithmetic stays)
#endif
// one code into the histogram.  Common case, branch-free: an integer code inside the window is one LDS atomic (zero included:
// with a replica per lane of the LDS service group the mode of the distribution no longer serialises on one address, so the
// register count k_mt_qdq keeps for it - and its compares - are not needed here); a code clamped to a NON-integer bound is
// counted in a register (it equals the bound only by clamping).  Everything else - an integer outside the window, inf, NaN -
// takes a branch that is rarely entered and counts into the global bins.  10 vector operations per element instead of ~25:
// the histogram cost the long-lived workgroups 9-16 % of the launch before.
__device__ __forceinline__ void mt_count(float t, float lo, float hi, bool lo_ni, bool hi_ni, int wstart, unsigned* sh_hist,
                                         unsigned* sh_nan, unsigned long long* hist, int C, unsigned& nlo, unsigned& nhi) {
    if (MT_CNT_ABL & 1) return;
    const int k = (int)t;                       // saturating; NaN -> 0
    const unsigned kk = (unsigned)(k - wstart);
    const bool inwin = ((float)k == t) && kk < (unsigned)MT_W;
    const bool at_hi = hi_ni && t == hi;
    const bool at_lo = lo_ni && t == lo && !at_hi;      // (c_min == c_max, a constant channel: ONE value, counted once)
    nhi += at_hi ? 1u : 0u;
    nlo += at_lo ? 1u : 0u;
    if (inwin) {
        if (!(MT_CNT_ABL & 4) || t == 12345.f) atomicAdd(&sh_hist[kk * MTF_REP + ((unsigned)threadIdx.x & (MTF_REP - 1))], 1u);
    } else if (!(at_hi || at_lo)) {             // rare
        if (t == rintf(t)) {                    // integer code outside the window (or inf)
            if (t >= (float)(-MT_NB / 2) && t < (float)(MT_NB / 2)) atomicAdd(&hist[(int)t + MT_NB / 2], 1ull);
            else atomicAdd(&hist[t < 0.f ? MT_NB : MT_NB + 1], 1ull);
            atomicAdd(&hist[mt_flag_word(C)], 1ull);   // the global bins are in use
        } else {
            atomicAdd(sh_nan, 1u);              // NaN: counted with the channel's lower bound, as k_mt_qdq does
        }
    }
}
// The same in the fast domain of the quantization (mt_fast_domain: no NaN, no inf) with clamp bounds below 2^24 in magnitude,
// where a code is an integer or EXACTLY one of the clamp bounds: ten vector operations less per element - the store phase
// of these kernels has no slack, the arithmetic of the count was 1.16 ms of VGG-16 b512's 17.3.  off:
// (lane & 31) - wstart * MTF_REP, so that k * MTF_REP + off is the word of the window (as unsigned: out of range when
// outside).  nni counts the codes that are not integers (clamped to a non-integer bound, either one), nhi those equal to hi
// (meaningful when hi is not an integer; the caller sorts it out: mt_counts_of).
__device__ __forceinline__ void mt_count_fast(float t, float hi, int off, unsigned* sh_hist, unsigned long long* hist, int C,
                                              unsigned& nni, unsigned& nhi) {
    if (MT_CNT_ABL & 1) return;
    const int k = (int)t;
    const bool nonint = (float)k != t;
    const unsigned addr = (unsigned)(k * MTF_REP + off);
    nni += nonint ? 1u : 0u;
    nhi += (t == hi) ? 1u : 0u;
    if (!nonint) {
        if (addr < (unsigned)(MT_W * MTF_REP)) {
            atomicAdd(&sh_hist[addr], 1u);
        } else {                                    // an integer code outside the window: rare
            if (k >= -MT_NB / 2 && k < MT_NB / 2) atomicAdd(&hist[k + MT_NB / 2], 1ull);
            else atomicAdd(&hist[k < 0 ? MT_NB : MT_NB + 1], 1ull);
            atomicAdd(&hist[mt_flag_word(C)], 1ull);
        }
    }
}
// (nni, nhi) of mt_count_fast -> the lane's counts of "clamped to the non-integer lower / upper bound"
__device__ __forceinline__ void mt_counts_of(unsigned nni, unsigned nhi_raw, bool hi_ni, unsigned& nlo, unsigned& nhi) {
    nhi = hi_ni ? nhi_raw : 0u;                     // an integer upper bound is a code like any other (counted in the window)
    nlo = nni - nhi;                                // what is not an integer and not hi was clamped to lo
}
inline __device__ bool mt_count_fast_ok(float lo, float hi) { return fabsf(lo) < 0x1p24f && fabsf(hi) < 0x1p24f; }
// end of the workgroup: the LDS window into the replica window (blockIdx picks the replica)
__device__ __forceinline__ void mt_flush(unsigned* sh_hist, unsigned long long* hist, int C, int wstart) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (MT_CNT_ABL & 2) return;
    unsigned long long* rep = hist + MT_NB + 2 + 2 * (size_t)C + (size_t)(blockIdx.x & (MT_GR - 1)) * MT_W;
    for (int i = tid; i < MT_W; i += TPB) {
        unsigned tot = 0;
#pragma unroll 8
        for (int r = 0; r < MTF_REP; ++r) tot += sh_hist[(unsigned)i * MTF_REP + ((unsigned)(r + tid) & (MTF_REP - 1))];
        if (tot) {
            if (wstart + i < MT_NB / 2) {
                atomicAdd(&rep[i], (unsigned long long)tot);
            } else {
                atomicAdd(&hist[MT_NB + 1], (unsigned long long)tot);
                atomicAdd(&hist[mt_flag_word(C)], 1ull);
            }
        }
    }
}

// ---- flat tiles (the geometry of k_mmq_flat: a group is ONE channel, a member 256 (K + KL) consecutive float4 of it) -----
template <int K, int OUT, int KL, int MODE, bool XR = false>
__global__ void __launch_bounds__(TPB, (K == 32 ? GRP_K32_WAVES : 1)) k_fused_flat(
    const float* __restrict__ x, float* __restrict__ y, const FGeo g, const GWs ws, const FusedArgs fa, const unsigned flags,
    const XOut xo = XOut{}, const XRank xr = XRank{}) {
    static_assert(TPB == 256, "wg_sum1 folds four waves");
    if constexpr (XR) xr_prologue(xr);                 // workgroup 0: the slots of the launch two back
    __shared__ double l_s[TPB / 64];
    __shared__ double sh_tot;
    extern __shared__ unsigned cnnq_dyn_lds[];     // MODE 0, OUT == 1: nbins x HREP words, sized by the launch (xhist_lds_bytes)
    __shared__ unsigned sh_hist_mt[(OUT == 1 && MODE == 1) ? MT_W * MTF_REP : 1];
    unsigned* const sh_hist = MODE == 0 ? cnnq_dyn_lds : sh_hist_mt;
    __shared__ unsigned sh_cnt[4];            // MODE 1 histogram: NaN / clamped-low, clamped-high counts of the channel
    __shared__ __attribute__((aligned(16))) float sh_x[KL ? KL * TPB * 4 : 4];
    const cnnq_params_cfg& cfg = fa.cfg;
    const bool ba = MODE == 0 && cfg.bit_alloc && cfg.num_bits <= 4;
    const int nbins = ba ? 256 : 1 << (cfg.num_bits < 8 ? cfg.num_bits : 8);
    const bool want_hist = OUT == 1 && (MODE == 0 ? xo.hist != nullptr : fa.hist != nullptr);
    if constexpr (OUT == 1) {
        // ordered before the first count by the barriers of the exchange
        if (want_hist) {
            if constexpr (MODE == 0) xhist_zero(sh_hist, nbins);
            else for (int i = threadIdx.x; i < MT_W * MTF_REP; i += TPB) sh_hist[i] = 0u;
        }
        if (threadIdx.x < 4) sh_cnt[threadIdx.x] = 0u;
    }
    __shared__ int sh_timed_out;
    const unsigned st0 = threadIdx.x == 0 ? __hip_atomic_load(ws.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const int tid = threadIdx.x;
    int c, member;
    if (g.cb <= 1) {
        c = (int)blockIdx.x / g.Gs;
        member = (int)blockIdx.x - c * g.Gs;
    } else {
        const int per = g.cb * g.Gs, blk = (int)blockIdx.x / per, r = (int)blockIdx.x - blk * per;
        const int c0 = blk * g.cb, cbl = min(g.cb, g.C - c0);
        member = r / cbl;
        c = c0 + (r - member * cbl);
    }
    // the channel's statistics of pass A (uniform addresses: scalar loads), in flight next to the tile
    const float vmin = fa.stats[(size_t)CNNQ_STAT_MIN * g.C + c], vmax = fa.stats[(size_t)CNNQ_STAT_MAX * g.C + c];
    const float vmean = fa.stats[(size_t)CNNQ_STAT_MEAN * g.C + c], vstd = fa.stats[(size_t)CNNQ_STAT_STD * g.C + c];
    const float bits = ba ? fa.bits[c] : (float)cfg.num_bits;

    const unsigned f0 = (unsigned)member * (256u * (K + KL));   // < total
    const unsigned n_first = f0 / g.cpc;
    const unsigned u = f0 + (unsigned)tid;
    const unsigned n = u / g.cpc;
    FWalk w0;
    w0.ro = (n - n_first) * g.rs;
    w0.co = (u - n * g.cpc) * 16u;
    const unsigned long long lim64 = (unsigned long long)((unsigned)g.N - n_first) * g.rs;
    const unsigned lim = lim64 > 0xffffffffull ? 0xffffffffu : (unsigned)lim64;   // row offsets below it are inside the batch
    const size_t base = ((size_t)n_first * (size_t)g.P + (size_t)c * (size_t)g.HW) * 4;
    const char* xb = reinterpret_cast<const char*>(x) + base;
    char* yb = reinterpret_cast<char*>(y) + base;
    uint8_t* cbb = (MODE == 0 && OUT == 1 && xo.codes) ? xo.codes + base / 4 : nullptr;

    // ---- the tile: K 16-byte loads per lane, back to back, then the last KL steps straight into LDS (LDS-DMA).  Round 6: the
    //      register steps first and the LDS-DMA behind the compiler's back (lds_dma16_behind, cnnq_common.hip.h) - with the
    //      builtin in flight it waited for vmcnt(0) before the first use of any register step
    float v[K][4];
    FWalk w = w0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const unsigned off = w.ro < lim ? w.ro + w.co : 0u;
        ldv_nt<4>(reinterpret_cast<const float*>(xb + off), v[j]);
        w.step(g);
    }
    if constexpr (KL > 0) {
#pragma unroll
        for (int l = 0; l < KL; ++l) {
            const unsigned off = w.ro < lim ? w.ro + w.co : 0u;
            lds_dma16_behind(xb, off, sh_x + (l * TPB + (tid & ~63)) * 4);
            w.step(g);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // steps of this lane inside the channel: u + 256 s < total.  Two accumulators, the register steps and the LDS steps, each
    // in step order, added at the end: the order does not depend on what landed first (the cold path does the same)
    const int nvalid = u < g.total ? (int)((g.total - u + 255u) / 256u) : 0;
    double sa = 0.;
#pragma unroll
    for (int j = 0; j < K; ++j) absdev_step(v[j], vmean, j < nvalid, sa);
    if constexpr (KL > 0) {
        lds_dma_landed(sa);
        double sl = 0.;
#pragma unroll
        for (int l = 0; l < KL; ++l) {
            const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
            const float t[4] = {q.x, q.y, q.z, q.w};
            absdev_step(t, vmean, K + l < nvalid, sl);
        }
        sa = sa + sl;
    }
    const double msum = wg_sum1(sa, l_s);

    // ---- the meeting: the member's sum is its arrival
    unsigned long long* slots = ws.slots + (size_t)c * ws.gstride;   // zero at rest
    double tsum;
    {
        const long long tmo = (st0 & 1u) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS;      // lane 0's copy is the one consulted
        const int timed_out = slots_meet_sum(slots, member, g.Gs, msum, flags, tmo, tsum);
        if (tid == 0) {
            if (timed_out) atomicOr(ws.status, (unsigned)timed_out);
            sh_timed_out = timed_out;
        }
    }
    __syncthreads();
    if (sh_timed_out) {
        // cold path: every member's partial sum from x with that member's own lane mapping, into the group's block of the
        // pair region (unused by the slot meeting; every workgroup that lands here writes the same values), then the same fold
        unsigned long long* tab = ws.part + (size_t)c * ws.gstride;
        for (int m = 0; m < g.Gs; ++m) {
            const unsigned mf0 = (unsigned)m * (256u * (K + KL)), mu = mf0 + (unsigned)tid;
            const int mvalid = mu < g.total ? (int)((g.total - mu + 255u) / 256u) : 0;
            double s2 = 0., sl2 = 0.;
            for (int s = 0; s < K + KL; ++s) {
                const bool in = s < mvalid;
                const unsigned ee = in ? mu + 256u * (unsigned)s : mf0;
                const unsigned nn = ee / g.cpc;
                float t[4];
                ldv<4>(x + (size_t)nn * (size_t)g.P + (size_t)c * (size_t)g.HW + (size_t)(ee - nn * g.cpc) * 4, t);
                if (s >= K) absdev_step(t, vmean, in, sl2);
                else absdev_step(t, vmean, in, s2);
            }
            if constexpr (KL > 0) s2 = s2 + sl2;
            const double ms = wg_sum1(s2, l_s);
            if (tid == 0) __hip_atomic_store(tab + m, (unsigned long long)__double_as_longlong(ms), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tsum = fold_member_sums(tab, g.Gs);
    }
    if (tid < 64) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) tsum += shfl_xor_d(tsum, m);
        if (tid == 0) {
            // the batch is sharded: the ranks' sums, added in rank order (cnnq_xrank.hip.h); NaN when a peer never came
            if constexpr (XR) (void)xr_merge_sum(xr, xr.slot0 + c, member == 0, tsum);
            sh_tot = tsum;
        }
    }
    __syncthreads();

    // ---- b and the channel's parameters: every lane derives the same values from the same inputs
    double cnt = fa.count;
    if constexpr (XR) cnt = fa.count_dev[c];          // (uniform address: a scalar load)
    const float vb = (float)(sh_tot / cnt);
    w = w0;
    asm volatile("" : "+v"(w.ro), "+v"(w.co));      // the walk repeated and hidden from the optimiser, as in k_mmq_flat
    if constexpr (MODE == 0) {
        const ChanParams cp = channel_params(cfg, ba, bits, vmin, vmax, vmean, vstd, vb);
        const float sc = cp.scale, zp = cp.zp, qm = cp.qmax;
        const bool fast = aciq_fast_domain(vmin, vmax, cp) && !(flags & MMQ_FLAG_IEEE_DIVIDE);
        if (member == 0 && tid == 0) {
            fa.qp[(size_t)CNNQ_QP_SCALE * g.C + c] = sc;
            fa.qp[(size_t)CNNQ_QP_ZP * g.C + c] = zp;
            fa.qp[(size_t)CNNQ_QP_QMAX * g.C + c] = qm;
            fa.stats[(size_t)CNNQ_STAT_B * g.C + c] = vb;
            if (fa.diag) {
                if (!ba) fa.diag[(size_t)CNNQ_DIAG_BITS * g.C + c] = bits;      // with bit allocation the row IS fa.bits
                fa.diag[(size_t)CNNQ_DIAG_ALPHA * g.C + c] = cp.alpha;
                fa.diag[(size_t)CNNQ_DIAG_DELTA * g.C + c] = cp.delta;
                fa.diag[(size_t)CNNQ_DIAG_OFFSET * g.C + c] = cp.offset;
            }
        }
        // ---- Q/DQ out of LDS and registers
        const float zpa[1] = {zp};
        unsigned nzp[1] = {0u};
        if (__builtin_amdgcn_readfirstlane((int)fast)) {
            const float s_sc = uniform_f(sc), s_rs = uniform_f(1.0f / sc), s_zp = uniform_f(zp), s_qm = uniform_f(qm);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float o[4], cd[4];
                qdq4_fast(v[j], s_sc, s_rs, s_zp, s_qm, o, cd);
                if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, nullptr, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                w.step(g);
            }
            if constexpr (KL > 0) {
#pragma unroll
                for (int l = 0; l < KL; ++l) {
                    const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
                    const float t[4] = {q.x, q.y, q.z, q.w};
                    float o[4], cd[4];
                    qdq4_fast(t, s_sc, s_rs, s_zp, s_qm, o, cd);
                    if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, nullptr, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                    w.step(g);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float o[4], cd[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = qdq1(v[j][e], sc, zp, qm, cd[e]);
                if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, nullptr, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                w.step(g);
            }
            if constexpr (KL > 0) {
#pragma unroll
                for (int l = 0; l < KL; ++l) {
                    const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
                    const float t[4] = {q.x, q.y, q.z, q.w};
                    float o[4], cd[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = qdq1(t[e], sc, zp, qm, cd[e]);
                    if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, nullptr, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                    w.step(g);
                }
            }
        }
        if constexpr (OUT == 1) {
            if (xo.hist) xhist_flush<1>(sh_hist, xo.hist, nbins, zpa, nzp);
        }
    } else {
        const float omega = fa.mt[(size_t)CNNQ_MT_OMEGA * g.C + c], am = fa.mt[(size_t)CNNQ_MT_ALPHA * g.C + c];
        const MtChan mc = mt_channel(fa.mcfg, omega, am, vmin, vmax, vmean, vb);
        const float d = mc.delta, lo = mc.cmin, hi = mc.cmax;
        // (with the histogram the divide-free path also wants clamp bounds its lean count can take: mt_count_fast)
        const bool fast = mt_fast_domain(vmin, vmax, d) && !(flags & MMQ_FLAG_IEEE_DIVIDE) && (!want_hist || mt_count_fast_ok(lo, hi));
        if (member == 0 && tid == 0) {
            fa.mt[(size_t)CNNQ_MT_DELTA * g.C + c] = d;
            fa.mt[(size_t)CNNQ_MT_CMIN * g.C + c] = lo;
            fa.mt[(size_t)CNNQ_MT_CMAX * g.C + c] = hi;
            fa.stats[(size_t)CNNQ_STAT_B * g.C + c] = vb;
        }
        const int wstart = want_hist ? (int)fa.mt[(size_t)CNNQ_MT_WSTART * g.C] : 0;
        const bool hi_ni = hi != rintf(hi), lo_ni = lo != rintf(lo);     // a non-integer bound is a value of its own
        unsigned nlo = 0u, nhi = 0u, nni = 0u;
        const int hoff = (tid & (MTF_REP - 1)) - wstart * MTF_REP;
        auto emit = [&](const float (&t)[4], bool isfast, float rd) {
            float o[4], cd[4];
            if (isfast) {
                mt_qdq4_fast(t, d, rd, lo, hi, o, cd);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = mt_qdq1<false>(t[e], d, rd, lo, hi, cd[e]);
            }
            if (w.ro < lim) {
                stv_nt<4>(reinterpret_cast<float*>(yb + (w.ro + w.co)), o);
                if constexpr (OUT == 1) {
                    if (want_hist) {
                        if (isfast) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) mt_count_fast(cd[e], hi, hoff, sh_hist, fa.hist, g.C, nni, nhi);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) mt_count(cd[e], lo, hi, lo_ni, hi_ni, wstart, sh_hist, &sh_cnt[0], fa.hist, g.C, nlo, nhi);
                        }
                    }
                }
            }
            w.step(g);
        };
        if (__builtin_amdgcn_readfirstlane((int)fast)) {
            const float rd = uniform_f(1.0f / d);
#pragma unroll
            for (int j = 0; j < K; ++j) emit(v[j], true, rd);
            if constexpr (KL > 0) {
#pragma unroll
                for (int l = 0; l < KL; ++l) {
                    const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
                    const float t[4] = {q.x, q.y, q.z, q.w};
                    emit(t, true, rd);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) emit(v[j], false, 0.f);
            if constexpr (KL > 0) {
#pragma unroll
                for (int l = 0; l < KL; ++l) {
                    const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
                    const float t[4] = {q.x, q.y, q.z, q.w};
                    emit(t, false, 0.f);
                }
            }
        }
        if constexpr (OUT == 1) {
            if (want_hist) {
                if (__builtin_amdgcn_readfirstlane((int)fast)) mt_counts_of(nni, nhi, hi_ni, nlo, nhi);
                if (nlo) atomicAdd(&sh_cnt[0], nlo);
                if (nhi) atomicAdd(&sh_cnt[1], nhi);
                mt_flush(sh_hist, fa.hist, g.C, wstart);      // (its barrier orders the counters too)
                if (tid == 0) {
                    if (sh_cnt[0]) atomicAdd(&fa.hist[MT_NB + 2 + c], (unsigned long long)sh_cnt[0]);
                    if (sh_cnt[1]) atomicAdd(&fa.hist[MT_NB + 2 + g.C + c], (unsigned long long)sh_cnt[1]);
                }
            }
        }
    }
    // ---- leave the group; the last member out re-arms the group's slots
    __syncthreads();
    if (tid == 0) sh_timed_out = grp_depart_last(grp_lines(ws.cnt, c, g.Gs, 0, 1), member, g.Gs) ? 1 : 0;
    __syncthreads();
    if (sh_timed_out)
        for (int m = tid; m < g.Gs; m += TPB) __hip_atomic_store(slots + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- row-piece tiles (the geometry of k_mmq_group: a member is <= 256 float4 columns x <= K samples) ----------------
// (group, member) -> tile, for any workgroup (the cold path walks every member of its group)
__device__ __forceinline__ RBlk rblk_at(const Geo& g, int group, int member) {
    RBlk r;
    r.group = group;
    r.member = member;
    int s;
    if (g.mode == 1) {
        const int cpc = g.HW / 4;
        s = member / g.nb;
        const int bb = member - s * g.nb;
        const int c = g.cbeg + group;
        r.b.c0 = c;
        r.b.c1 = c + 1;
        r.b.col0 = c * cpc + bb * g.w;
        r.b.col1 = min(r.b.col0 + g.w, (c + 1) * cpc);
    } else {
        s = member;
        r.b.c0 = g.cbeg + group * g.k;
        r.b.c1 = min(g.cbeg + g.Cn, r.b.c0 + g.k);
        r.b.col0 = (int)(((int64_t)r.b.c0 * g.HW) / 4);
        r.b.col1 = (int)(((int64_t)r.b.c1 * g.HW) / 4);
    }
    r.b.n0 = (int)(((int64_t)s * g.N) / g.S);
    r.b.n1 = (int)(((int64_t)(s + 1) * g.N) / g.S);
    r.b.grp = s;
    return r;
}

// per-lane sums -> per-channel sums of the workgroup's tile in sh_sum[c1 - c0], in a fixed order (the layout of
// wg_channel_minmax: mode 1 - one channel, all lanes; mode 2 - epc LDS entries per channel)
template <int A>
__device__ __forceinline__ void wg_channel_sums(const Geo& g, const Blk& b, bool ok, const double (&sa)[A], double* l_a,
                                                double* sh_sum) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    __syncthreads();                     // l_a / sh_sum may still be read from a previous call
    if (g.mode == 1) {
        double t = ok ? sa[0] : 0.;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t += shfl_xor_d(t, m);
        if (lane == 0) l_a[wv] = t;
        __syncthreads();
        if (tid == 0) sh_sum[0] = ((l_a[0] + l_a[1]) + l_a[2]) + l_a[3];
        __syncthreads();
        return;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) l_a[tid * A + a] = ok ? sa[a] : 0.;
    __syncthreads();
    const int epc = g.HW * A / 4;   // LDS entries per channel
    if (epc <= 16) {
        for (int ch = tid; ch < b.c1 - b.c0; ch += TPB) {
            double t = 0.;
            for (int e = ch * epc; e < (ch + 1) * epc; ++e) t += l_a[e];
            sh_sum[ch] = t;
        }
    } else {
        for (int ch = wv; ch < b.c1 - b.c0; ch += TPB / 64) {
            double t = 0.;
            for (int e = ch * epc + lane; e < (ch + 1) * epc; e += 64) t += l_a[e];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) t += shfl_xor_d(t, m);
            if (lane == 0) sh_sum[ch] = t;
        }
    }
    __syncthreads();
}

// The group's sums from its members' partial sums, [member][kk] 8-byte words at `src`: slots (POLL: complemented, zero =
// not arrived) or plain doubles (the cold path's recomputed table).  Lane (ch, j) = (tid / L, tid % L) adds the values of
// channel ch of members j, j + L, ... in member order, window by window and only once a window is complete; the L lanes
// of a channel sit in one wave and fold by a fixed xor tree.  Writes sh_sum[ch] (kk > 1), or returns the lane's share in
// tsum (whole_wg: one channel, all 256 lanes take part; the caller folds).  A wave whose wait expired ORs 1 into *sh_code.
template <int W, bool POLL>
__device__ __forceinline__ void group_fold_sums(const unsigned long long* src, int Gs, int kk, int nch, bool whole_wg,
                                                long long timeout_ticks, double* sh_sum, int* sh_code, double& tsum) {
    const int tid = threadIdx.x;
    int L = 1;
    while (L < 64 && 2 * L * kk <= TPB) L <<= 1;
    if (whole_wg) L = TPB;
    const int ch = tid / L, j = tid - ch * L;
    const bool active = ch < nch;
    tsum = 0.;
    long long t0 = 0;
    int spins = 0;
    for (int w0 = 0; w0 * L < Gs; w0 += W) {
        const unsigned long long* p = src + (unsigned)((j + L * w0) * kk + ch);
        const unsigned step = (unsigned)(L * kk);
        unsigned pend = 0u;
#pragma unroll
        for (int i = 0; i < W; ++i) pend |= (active && j + L * (w0 + i) < Gs) ? (1u << i) : 0u;
        const unsigned mine = pend;
        unsigned long long v[W];
#pragma unroll
        for (int i = 0; i < W; ++i) v[i] = 0ull;
        if constexpr (POLL) {
            for (;; ++spins) {
#pragma unroll
                for (int i = 0; i < W; ++i)
                    if ((pend >> i) & 1u) {
                        v[i] = __hip_atomic_load(p + i * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v[i]) pend &= ~(1u << i);
                    }
                if (__ballot(pend != 0u) == 0ull) break;
                int expired = 0;
                if ((spins & 31) == 31 || spins > GRP_TIMEOUT_SPINS) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    expired = (now - t0 > timeout_ticks || spins > GRP_TIMEOUT_SPINS) ? 1 : 0;
                }
                if (__builtin_amdgcn_readfirstlane(expired)) {
                    if ((tid & 63) == 0) atomicOr(sh_code, 1);
                    return;
                }
                if (spins < 2) __builtin_amdgcn_s_sleep(8);
                else if (spins < 6) __builtin_amdgcn_s_sleep(32);
                else __builtin_amdgcn_s_sleep(64);
            }
#pragma unroll
            for (int i = 0; i < W; ++i)
                if ((mine >> i) & 1u) tsum += sum_of_slot(v[i]);
        } else {
#pragma unroll
            for (int i = 0; i < W; ++i)
                if ((mine >> i) & 1u)
                    tsum += __longlong_as_double((long long)__hip_atomic_load(p + i * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    if (!whole_wg) {
        for (int m = L >> 1; m >= 1; m >>= 1) tsum += shfl_xor_d(tsum, m);
        if (active && j == 0) sh_sum[ch] = tsum;
    }
}

// MODE 1 (mid-tread): sh_sc = delta, sh_zp = c_min, sh_qm = c_max, sh_rs = 1 / delta; A == 1 only (no VGG layer straddles)
// (waves per SIMD the K = 32 tile is compiled for: three - 168 registers - except the one instance that does not fit them, the
//  mid-tread form with the histogram AND the cross-rank stage, which takes two: no instance of the library uses scratch)
template <int K, int OUT, int MODE, bool XR>
constexpr int fused_group_waves() { return K != 32 ? 1 : (XR && OUT == 1 && MODE == 1) ? 2 : GRP_K32_WAVES; }
template <int A, int K, int OUT, int MODE, bool XR = false>
__global__ void __launch_bounds__(TPB, (fused_group_waves<K, OUT, MODE, XR>())) k_fused_group(
    const float* __restrict__ x, float* __restrict__ y, const Geo g, const int Gs, const GWs ws, const FusedArgs aa,
    const unsigned flags, const XOut xo = XOut{}, const XRank xr = XRank{}) {
    static_assert(MODE == 0 || A == 1, "the mid-tread form has no straddling instance");
    if constexpr (XR) xr_prologue(xr);                 // workgroup 0: the slots of the launch two back
    __shared__ double l_a[TPB * A];
    __shared__ double sh_sum[MAXCH];
    extern __shared__ unsigned cnnq_dyn_lds[];     // MODE 0, OUT == 1: nbins x HREP words, sized by the launch (xhist_lds_bytes)
    __shared__ unsigned sh_hist_mt[(OUT == 1 && MODE == 1) ? MT_W * MTF_REP : 1];
    unsigned* const sh_hist = MODE == 0 ? cnnq_dyn_lds : sh_hist_mt;
    __shared__ unsigned sh_clo[(MODE == 1 && OUT == 1) ? MAXCH : 1], sh_chi[(MODE == 1 && OUT == 1) ? MAXCH : 1];
    const cnnq_params_cfg& cfg = aa.cfg;
    const bool ba = MODE == 0 && cfg.bit_alloc && cfg.num_bits <= 4;
    const int nbins = ba ? 256 : 1 << (cfg.num_bits < 8 ? cfg.num_bits : 8);
    const bool want_hist = OUT == 1 && (MODE == 0 ? xo.hist != nullptr : aa.hist != nullptr);
    if constexpr (OUT == 1) {
        if (want_hist) {
            if constexpr (MODE == 0) xhist_zero(sh_hist, nbins);
            else {
                for (int i = threadIdx.x; i < MT_W * MTF_REP; i += TPB) sh_hist[i] = 0u;
                for (int i = threadIdx.x; i < MAXCH; i += TPB) { sh_clo[i] = 0u; sh_chi[i] = 0u; }
            }
        }
    }
    __shared__ float sh_mean[MAXCH], sh_sc[MAXCH], sh_zp[MAXCH], sh_rs[MAXCH], sh_qm[MAXCH];
    __shared__ int sh_timed_out, sh_slow;
    if (threadIdx.x == 0) sh_slow = 0;      // the barriers of the reduction and of the exchange come before its writers
    const unsigned st0 = threadIdx.x == 0 ? __hip_atomic_load(ws.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const RBlk rb = rblk_of(g, Gs);
    const Blk& b = rb.b;
    const int tid = threadIdx.x;
    const int nch = b.c1 - b.c0;
    const int col = b.col0 + tid;
    const bool ok = col < b.col1;
    const unsigned colc = (unsigned)(ok ? col : b.col0);   // idle lanes re-read the block's first column; results discarded
    const int nrows = b.n1 - b.n0;        // 1 .. K
    const size_t base = (size_t)b.n0 * (size_t)g.P + (size_t)colc * 4;
    // the means of the block's channels (one per lane, issued in front of the tile; MAXCH <= TPB)
    static_assert(MAXCH <= TPB, "one staging lane per channel");
    const float r_mean = tid < nch ? aa.stats[(size_t)CNNQ_STAT_MEAN * g.C + b.c0 + tid] : 0.f;

    // ---- the tile: K 16-byte loads per lane, issued back to back (rows past the tile re-read its last row)
    float v[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int r = j < nrows ? j : nrows - 1;
        ldv_nt<4>(x + base + (size_t)r * (size_t)g.P, v[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tid < nch) sh_mean[tid] = r_mean;
    __syncthreads();
    float mean[A];
#pragma unroll
    for (int a = 0; a < A; ++a) mean[a] = sh_mean[(int)((colc * 4u + (unsigned)a) / (unsigned)g.HW) - b.c0];
    double sa[A];
#pragma unroll
    for (int a = 0; a < A; ++a) sa[a] = 0.;
    // (a uniform branch per step, not a select: 32 lane masks held in scalar registers next to the tile made the
    //  allocator spill the tile itself)
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < nrows) {
            if constexpr (A == 1) {
                absdev_step(v[j], mean[0], true, sa[0]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) sa[e] += (double)fabsf(v[j][e] - mean[e]);
            }
        }
    }
    wg_channel_sums<A>(g, b, ok, sa, l_a, sh_sum);

    // ---- publish this workgroup's sums (a sum IS the arrival), wait for the group
    const int kk = (g.mode == 1) ? 1 : g.k;
    unsigned long long* slots = ws.slots + (size_t)rb.group * ws.gstride;    // zero at rest
    for (int ch = tid; ch < nch; ch += TPB)
        __hip_atomic_store(slots + (size_t)rb.member * kk + ch, slot_of_sum(sh_sum[ch]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) sh_timed_out = ((flags & MMQ_FLAG_TEST_HOOK) ? 2 : 0) | ((st0 & 1u) ? 4 : 0);
    __syncthreads();      // sh_sum is read above and rewritten by the meeting
    double ts = 0.;
    {
        const int c0 = sh_timed_out;
        if (!(c0 & 2))
            group_fold_sums<(K >= 32 ? 4 : 2), true>(slots, Gs, kk, nch, g.mode == 1, (c0 & 4) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS,
                                                     sh_sum, &sh_timed_out, ts);
        __syncthreads();
        if (tid == 0) {
            const int code = sh_timed_out & 3;
            if (code) atomicOr(ws.status, (unsigned)code);
        }
    }
    if (sh_timed_out & 3) {
        // cold path: every member's per-channel sums from x, with that member's own lane mapping, into the group's block
        // of the pair region (unused by the slot meeting; every workgroup that lands here writes the same values), then
        // the same fold over the table
        unsigned long long* tab = ws.part + (size_t)rb.group * ws.gstride;
        for (int m = 0; m < Gs; ++m) {
            const RBlk mb = rblk_at(g, rb.group, m);
            const int mcol = mb.b.col0 + tid;
            const bool mok = mcol < mb.b.col1;
            const int mcolc = mok ? mcol : mb.b.col0;
            const int mrows = mb.b.n1 - mb.b.n0;
            float mmean[A];
#pragma unroll
            for (int a = 0; a < A; ++a) mmean[a] = sh_mean[(int)(((unsigned)mcolc * 4u + (unsigned)a) / (unsigned)g.HW) - mb.b.c0];
            double s2[A];
#pragma unroll
            for (int a = 0; a < A; ++a) s2[a] = 0.;
            for (int j = 0; j < K; ++j) {
                const int r = j < mrows ? j : mrows - 1;
                float t[4];
                ldv<4>(x + ((size_t)(mb.b.n0 + r) * (size_t)g.P + (size_t)mcolc * 4), t);
                if (j < mrows) {
                    if constexpr (A == 1) {
                        absdev_step(t, mmean[0], true, s2[0]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) s2[e] += (double)fabsf(t[e] - mmean[e]);
                    }
                }
            }
            wg_channel_sums<A>(g, mb.b, mok, s2, l_a, sh_sum);
            for (int ch = tid; ch < nch; ch += TPB)
                __hip_atomic_store(tab + (size_t)m * kk + ch, (unsigned long long)__double_as_longlong(sh_sum[ch]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        group_fold_sums<(K >= 32 ? 4 : 2), false>(tab, Gs, kk, nch, g.mode == 1, 0, sh_sum, &sh_timed_out, ts);
        __syncthreads();
    }
    if (g.mode == 1) {
        // one channel, 256 shares: the fixed fold of wg_channel_sums
        const double one[1] = {ts};
        wg_channel_sums<1>(g, b, true, one, l_a, sh_sum);
    }

    if constexpr (XR) {
        // the batch is sharded: every rank's sum of the owned channels, added in rank order (cnnq_xrank.hip.h); a loop of its own -
        // next to the parameter arithmetic below its state cost the K = 32 tile registers it does not have.  (Lane tid reads
        // sh_sum[tid] back below: no barrier.)
        for (int ch = tid; ch < nch; ch += TPB) {
            double csum = sh_sum[ch];
            (void)xr_merge_sum(xr, xr.slot0 + b.c0 + ch, rb.member == 0, csum);
            sh_sum[ch] = csum;
        }
    }
    // ---- b and the parameters of the owned channels: identical in every member
    for (int ch = tid; ch < nch; ch += TPB) {
        const int c = b.c0 + ch;
        const float vmin = aa.stats[(size_t)CNNQ_STAT_MIN * g.C + c], vmax = aa.stats[(size_t)CNNQ_STAT_MAX * g.C + c];
        double cnt = aa.count;
        if constexpr (XR) cnt = aa.count_dev[c];
        const float vb = (float)(sh_sum[ch] / cnt);
        if constexpr (MODE == 0) {
            const float vstd = aa.stats[(size_t)CNNQ_STAT_STD * g.C + c];
            const float bits = ba ? aa.bits[c] : (float)cfg.num_bits;
            const ChanParams cp = channel_params(cfg, ba, bits, vmin, vmax, sh_mean[ch], vstd, vb);
            sh_sc[ch] = cp.scale;
            sh_zp[ch] = cp.zp;
            sh_qm[ch] = cp.qmax;
            sh_rs[ch] = 1.0f / cp.scale;
            if (!aciq_fast_domain(vmin, vmax, cp) || (flags & MMQ_FLAG_IEEE_DIVIDE)) sh_slow = 1;   // any writer, same value
            if (rb.member == 0) {
                aa.qp[(size_t)CNNQ_QP_SCALE * g.C + c] = cp.scale;
                aa.qp[(size_t)CNNQ_QP_ZP * g.C + c] = cp.zp;
                aa.qp[(size_t)CNNQ_QP_QMAX * g.C + c] = cp.qmax;
                aa.stats[(size_t)CNNQ_STAT_B * g.C + c] = vb;
                if (aa.diag) {
                    if (!ba) aa.diag[(size_t)CNNQ_DIAG_BITS * g.C + c] = bits;      // with bit allocation the row IS aa.bits
                    aa.diag[(size_t)CNNQ_DIAG_ALPHA * g.C + c] = cp.alpha;
                    aa.diag[(size_t)CNNQ_DIAG_DELTA * g.C + c] = cp.delta;
                    aa.diag[(size_t)CNNQ_DIAG_OFFSET * g.C + c] = cp.offset;
                }
            }
        } else {
            const float omega = aa.mt[(size_t)CNNQ_MT_OMEGA * g.C + c], am = aa.mt[(size_t)CNNQ_MT_ALPHA * g.C + c];
            const MtChan mc = mt_channel(aa.mcfg, omega, am, vmin, vmax, sh_mean[ch], vb);
            sh_sc[ch] = mc.delta;
            sh_zp[ch] = mc.cmin;
            sh_qm[ch] = mc.cmax;
            sh_rs[ch] = 1.0f / mc.delta;
            if (!mt_fast_domain(vmin, vmax, mc.delta) || (flags & MMQ_FLAG_IEEE_DIVIDE) || (want_hist && !mt_count_fast_ok(mc.cmin, mc.cmax))) sh_slow = 1;
            if (rb.member == 0) {
                aa.mt[(size_t)CNNQ_MT_DELTA * g.C + c] = mc.delta;
                aa.mt[(size_t)CNNQ_MT_CMIN * g.C + c] = mc.cmin;
                aa.mt[(size_t)CNNQ_MT_CMAX * g.C + c] = mc.cmax;
                aa.stats[(size_t)CNNQ_STAT_B * g.C + c] = vb;
            }
        }
    }
    __syncthreads();
    // the lane's channel indices again, hidden from the optimiser: kept alive across the meeting next to the tile they
    // cost registers the K = 32 tile does not leave (spills)
    int tidq = threadIdx.x;
    asm volatile("" : "+v"(tidq));
    // ... and the row count: the compiler otherwise keeps the K results of `j < nrows` of the first phase as 64-bit masks for this
    // one - 64 scalar registers, spilled into lanes of two vector registers the tile needs (round 6)
    int nrows_q = nrows;
    asm volatile("" : "+s"(nrows_q));
    const bool okq = b.col0 + tidq < b.col1;
    const unsigned colq = (unsigned)(okq ? b.col0 + tidq : b.col0);
    const size_t baseq = (size_t)b.n0 * (size_t)g.P + (size_t)colq * 4;
    int chl[A];
#pragma unroll
    for (int a = 0; a < A; ++a) chl[a] = (int)((colq * 4u + (unsigned)a) / (unsigned)g.HW) - b.c0;
    float sc[A], zp[A], qm[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        sc[a] = sh_sc[chl[a]];
        zp[a] = sh_zp[chl[a]];
        qm[a] = sh_qm[chl[a]];
    }
    const bool fastq = !__builtin_amdgcn_readfirstlane(sh_slow);

    if constexpr (MODE == 0) {
        // ---- Q/DQ out of the registers
        unsigned nzp[A];
#pragma unroll
        for (int a = 0; a < A; ++a) nzp[a] = 0u;
        if (fastq) {
            float rs[A];
#pragma unroll
            for (int a = 0; a < A; ++a) rs[a] = sh_rs[chl[a]];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (j < nrows_q) {
                    float o[4], cd[4];
                    if constexpr (A == 1) {
                        qdq4_fast(v[j], sc[0], rs[0], zp[0], qm[0], o, cd);      // two elements per instruction, the same bits
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = qdq1_fast(v[j][e], sc[e], rs[e], zp[e], qm[e], cd[e]);
                    }
                    if (okq)
                        xstore<OUT, A>(xo, reinterpret_cast<char*>(y), xo.codes, nullptr, (baseq + (size_t)j * (size_t)g.P) * 4, o, cd, sh_hist,
                                       zp, nzp);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (j < nrows_q) {
                    float o[4], cd[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = qdq1(v[j][e], sc[A == 1 ? 0 : e], zp[A == 1 ? 0 : e], qm[A == 1 ? 0 : e], cd[e]);
                    if (okq)
                        xstore<OUT, A>(xo, reinterpret_cast<char*>(y), xo.codes, nullptr, (baseq + (size_t)j * (size_t)g.P) * 4, o, cd, sh_hist,
                                       zp, nzp);
                }
            }
        }
        if constexpr (OUT == 1) {
            if (xo.hist) xhist_flush<A>(sh_hist, xo.hist, nbins, zp, nzp);
        }
    } else {
        // ---- mid-tread: quantize, clamp, dequantize out of the registers; count the codes
        const float d = sc[0], lo = zp[0], hi = qm[0];
        const float rd = fastq ? sh_rs[chl[0]] : 0.f;
        const int wstart = want_hist ? (int)aa.mt[(size_t)CNNQ_MT_WSTART * g.C] : 0;
        const bool hi_ni = hi != rintf(hi), lo_ni = lo != rintf(lo);
        unsigned nlo = 0u, nhi = 0u, nni = 0u;
        const int hoff = (tidq & (MTF_REP - 1)) - wstart * MTF_REP;
        // (two copies of the row loop behind ONE uniform branch: with the fast / general choice inside every unrolled row the
        //  allocator spilled the tile - 528 bytes of scratch)
        auto rows_out = [&](auto isfast_t) {
            constexpr bool ISFAST = decltype(isfast_t)::value;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (j < nrows_q) {
                    float o[4], cd[4];
                    if constexpr (ISFAST) {
                        mt_qdq4_fast(v[j], d, rd, lo, hi, o, cd);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = mt_qdq1<false>(v[j][e], d, rd, lo, hi, cd[e]);
                    }
                    if (okq) {
                        stv_nt<4>(y + baseq + (size_t)j * (size_t)g.P, o);
                        if constexpr (OUT == 1) {
                            if (want_hist) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if constexpr (ISFAST) mt_count_fast(cd[e], hi, hoff, sh_hist, aa.hist, g.C, nni, nhi);
                                    else mt_count(cd[e], lo, hi, lo_ni, hi_ni, wstart, sh_hist, &sh_clo[chl[0]], aa.hist, g.C, nlo, nhi);
                                }
                            }
                        }
                    }
                }
            }
        };
        if (fastq) rows_out(std::true_type{});
        else rows_out(std::false_type{});
        if constexpr (OUT == 1) {
            if (want_hist) {
                if (fastq) mt_counts_of(nni, nhi, hi_ni, nlo, nhi);
                if (nlo) atomicAdd(&sh_clo[chl[0]], nlo);
                if (nhi) atomicAdd(&sh_chi[chl[0]], nhi);
                mt_flush(sh_hist, aa.hist, g.C, wstart);      // (its barrier orders the clamp counters too)
                for (int i = tid; i < nch; i += TPB) {
                    if (sh_clo[i]) atomicAdd(&aa.hist[MT_NB + 2 + b.c0 + i], (unsigned long long)sh_clo[i]);
                    if (sh_chi[i]) atomicAdd(&aa.hist[MT_NB + 2 + g.C + b.c0 + i], (unsigned long long)sh_chi[i]);
                }
            }
        }
    }
    __syncthreads();
    // ---- leave the group; the last member out re-arms the group's slots
    if (tid == 0) sh_timed_out = grp_depart_last(grp_lines(ws.cnt, rb.group, Gs, 0, 1), rb.member, Gs) ? 1 : 0;
    __syncthreads();
    if (sh_timed_out)
        for (int m = tid; m < Gs * kk; m += TPB) __hip_atomic_store(slots + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace
