// cnnq_stats1.hip.h - the seven per-channel statistics of `-sm collect` (config 4; statistic_manager_perchannel.py:45-79) in
// ONE launch and ONE read of x: 4 instead of 8 bytes per element, one launch instead of three (k_moments -> k_absdev<RAW> ->
// k_combine_all).  Part of the single translation unit cnnq_kernels.hip.
//
// Round 2 built this once on the counter meeting (four dependent round trips per exchange, two exchanges per launch) and it
// lost to the two streaming passes on every layer.  With the slot meeting an exchange is one store and one polling round
// trip, and the partial sums travel the way cnnq_aciq.hip.h moves them: complemented 8-byte words that ARE the arrival,
// added in member order.  The workgroup keeps its tile of x in registers across both phases:
//
//   phase 1   per lane min / max / sum / sum of squares (and the sums of relu(x)) -> the member's five words into its line of
//             the slot region -> every member polls the group's lines and folds them in member order -> mean, std, std_pos;
//   phase 2   sum |x - mean| and sum ((x - mean) / std)^4 out of the same registers -> two more words per member -> the same
//             meeting -> b, kurtosis; member 0 writes the channel's row of the table (and the merged moment record).
//
// Flat tiles (k_stats_flat; the geometry of k_mmq_flat: a group is one channel) for the layers that hold the bytes (ResNet-50
// b512: 25 of 53 tensors, 90 % of the elements) and row-piece tiles (k_stats_group, below) for short channel rows, which pay
// only on the largest of them (cnnq_pc_stats_single has the rule and the measurements); the rest keep the three-launch chain.  The arithmetic per element and
// the final formulas are the chain's (Mom::add4, k_absdev's fp32 (x - mean) * (1 / std), mean_of / std_of, k_combine_all);
// only the order of the fp64 additions differs, as it does between any two tilings: results agree with the chain to fp64
// rounding, and are bit-identical run after run and between the meeting and the recompute path (the cold path recomputes
// EVERY member's words with that member's own lane mapping and folds them in the same order).
//
// Round 6 - XR = true (k_stats_flat, k_stats_group): the batch is sharded over W GPUs (cnnq_xrank.hip.h).  After each phase's LOCAL meeting the
// folded words of the rank - phase 1: the pair, the sums, and the rank's element count; phase 2: the two sums - are exchanged with
// the other ranks inside the launch (lane w of wave 0 takes word w: one polling loop for all of them; member 0 pushes, every
// member reads its own rank's window) and folded in rank order, so every rank writes the row of the GLOBAL batch from one read
// of its shard.  Slots: word w of channel c in slot w * C + c (phase 1: words 0..5, phase 2: 6..7).
#pragma once
#include "cnnq_aciq.hip.h"
#include "cnnq_common.hip.h"
#include "cnnq_group.hip.h"
#include "cnnq_stats.hip.h"

namespace {

constexpr int ST_W1 = 5;     // phase-1 words of a member: {min, max} pair, sum, sum of squares, relu sum, relu sum of squares
constexpr int ST_W2 = 2;     // phase-2 words: sum |x - mean|, sum z^4
constexpr int ST_LINE = 8;   // words per member line (64 bytes: one store instruction of lanes 0..4 covers a phase's words)
constexpr int ST_XW_COUNT = 5;   // cross-rank words of phase 1 (XR): 0..4 as above, 5 the rank's element count; phase 2: 6, 7
constexpr int ST_XW = 8;         // slots per channel a sharded launch uses
constexpr int ST_MAX_MEMBERS = 256;      // tiles per channel up to which the single launch is routed by default (cnnq_pc_stats_single)

struct St1Args {
    float* stats;            // [CNNQ_NSTAT][C] out: every row
    double* mom;             // [CNNQ_NMOM][C] out, may be null
    double count;            // N * H*W
    int need_relu, need_dev, need_kurt;      // relu sums; phase 2 at all; the fourth moment
};

// the member's words of one phase (every thread holds them): lane i stores word i (predicated stores of a compile-time
// indexed array: a select chain over the array made the compiler index it in scratch)
__device__ __forceinline__ void st_publish(unsigned long long* line, const unsigned long long (&w)[ST_W1], int first, int nw) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ST_W1; ++i)
        if (tid == i && i < nw) __hip_atomic_store(line + first + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Fold of the words of every member's line.  Lane (w, j) = (tid / 32, tid % 32), w < nw, takes word w of members j, j + 32,
// ... in member order (st_fold_poll: windows of four slots polled with a bounded wait; the cold path: st_fold_add member by
// member - the same per-lane sequence of additions); st_fold_finish folds the 32 lanes of a word by a fixed xor tree.  Word 0
// of phase 1 is the {min, max} pair.  Results in sh_out[w] (the pair as its two floats in sh_mm).
struct StFold {
    double acc;
    float amn, amx;
    __device__ __forceinline__ void init() { acc = 0.; amn = INFINITY; amx = -INFINITY; }
    __device__ __forceinline__ void add(unsigned long long v, bool is_pair) {
        if (is_pair) {
            float a, b;
            unpack_pair(v, a, b);
            amn = pmin(amn, a);
            amx = pmax(amx, b);
        } else {
            acc += __longlong_as_double((long long)v);
        }
    }
};
// the cold path: member m's words (every thread holds them) into the fold of the lanes that own member m
__device__ __forceinline__ void st_fold_add(StFold& f, int m, const unsigned long long (&words)[ST_W1], int nw, bool pair0) {
    const int tid = threadIdx.x;
    const int w = tid >> 5, j = tid & 31;
    if ((m & 31) == j) {
#pragma unroll
        for (int i = 0; i < ST_W1; ++i)
            if (w == i && i < nw) f.add(words[i], pair0 && i == 0);
    }
}
__device__ __forceinline__ void st_fold_finish(StFold& f, int nw, bool pair0, double* sh_out, float* sh_mm) {
    const int tid = threadIdx.x;
    const int w = tid >> 5, j = tid & 31;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        f.acc += shfl_xor_d(f.acc, m);
        f.amn = pmin(f.amn, shfl_xor_f(f.amn, m));
        f.amx = pmax(f.amx, shfl_xor_f(f.amx, m));
    }
    if (w < nw && j == 0) {
        sh_out[w] = f.acc;
        if (pair0 && w == 0) { sh_mm[0] = f.amn; sh_mm[1] = f.amx; }
    }
}
// the meeting: returns false (wave-uniformly) when a wait expired, after OR-ing 1 into *sh_code
__device__ __forceinline__ bool st_fold_poll(StFold& f, const unsigned long long* lines, int Gs, int first, int nw, bool pair0,
                                             long long tmo, int* sh_code) {
    const int tid = threadIdx.x;
    const int w = tid >> 5, j = tid & 31;
    const bool active = w < nw;
    long long t0 = 0;
    int spins = 0;
    for (int m0 = 0; m0 < Gs; m0 += 128) {        // windows of 4 members per lane
        unsigned pend = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) pend |= (active && m0 + j + 32 * i < Gs) ? (1u << i) : 0u;
        const unsigned mine = pend;
        unsigned long long v[4] = {0ull, 0ull, 0ull, 0ull};
        const unsigned long long* p = lines + (size_t)(m0 + j) * ST_LINE + first + w;
        for (;; ++spins) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((pend >> i) & 1u) {
                    v[i] = __hip_atomic_load(p + (size_t)32 * i * ST_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v[i]) pend &= ~(1u << i);
                }
            if (__ballot(pend != 0u) == 0ull) break;
            int expired = 0;
            if ((spins & 31) == 31 || spins > GRP_TIMEOUT_SPINS) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                expired = (now - t0 > tmo || spins > GRP_TIMEOUT_SPINS) ? 1 : 0;
            }
            if (__builtin_amdgcn_readfirstlane(expired)) {
                if ((tid & 63) == 0) atomicOr(sh_code, 1);
                return false;
            }
            if (spins < 2) __builtin_amdgcn_s_sleep(8);
            else if (spins < 6) __builtin_amdgcn_s_sleep(32);
            else __builtin_amdgcn_s_sleep(64);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((mine >> i) & 1u) f.add(~v[i], pair0 && w == 0);      // member order, whatever the arrival order was
    }
    return true;
}

// a member's words: the pair (NaNs canonical) as it is, the sums as plain doubles (st_publish complements them on the way out)
struct StWords {
    unsigned long long w[ST_W1];
};

// KR steps of the tile in registers, the last KL in LDS (LDS-DMA: no staging registers) - the 128 KB tile is 24 + 8: with
// all 32 steps in registers next to the accumulators of two phases the allocator spilled six of them
template <int KR, int KL, bool RELU, bool NTL, bool XR = false>
__global__ void __launch_bounds__(TPB, (KR + KL == 32 ? GRP_K32_WAVES : 1)) k_stats_flat(const float* __restrict__ x, const FGeo g, const GWs ws,
                                                                                       const St1Args sa, const unsigned flags,
                                                                                       const XRank xr = XRank{}) {
    static_assert(TPB == 256, "four waves");
    constexpr int K = KR + KL;
    if constexpr (XR) xr_prologue(xr);                 // workgroup 0: the slots of the launch two back
    __shared__ double sh_count;                        // XR: the global batch's elements per channel
    __shared__ __attribute__((aligned(16))) float sh_x[KL ? KL * TPB * 4 : 4];
    __shared__ float l_mn[TPB / 64], l_mx[TPB / 64];
    __shared__ double l_d[4][TPB / 64];
    __shared__ double sh_out[8], sh_out2[2];      // the folded words of phase 1 (kept for the final row) and of phase 2
    __shared__ float sh_mm[2];
    __shared__ int sh_code, sh_last;
    const unsigned st0 = threadIdx.x == 0 ? __hip_atomic_load(ws.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
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
    if (tid == 0) sh_code = ((flags & MMQ_FLAG_TEST_HOOK) ? 2 : 0) | ((st0 & 1u) ? 4 : 0);

    // ---- the tile: K 16-byte loads per lane, back to back (k_mmq_flat's walk)
    const unsigned f0 = (unsigned)member * (256u * K);
    const unsigned n_first = f0 / g.cpc;
    const unsigned u = f0 + (unsigned)tid;
    const unsigned n = u / g.cpc;
    FWalk w;
    w.ro = (n - n_first) * g.rs;
    w.co = (u - n * g.cpc) * 16u;
    const unsigned long long lim64 = (unsigned long long)((unsigned)g.N - n_first) * g.rs;
    const unsigned lim = lim64 > 0xffffffffull ? 0xffffffffu : (unsigned)lim64;
    const char* xb = reinterpret_cast<const char*>(x) + ((size_t)n_first * (size_t)g.P + (size_t)c * (size_t)g.HW) * 4;
    //      (round 6: the register steps FIRST - with LDS-DMA loads issued before them the compiler waits for vmcnt(0) at the first
    //      use of any register step; behind them the pairs of phase 1 start as their own loads land)
    float v[KR][4];
#pragma unroll
    for (int j = 0; j < KR; ++j) {
        const unsigned off = w.ro < lim ? w.ro + w.co : 0u;
        ldv_sel<4, NTL>(reinterpret_cast<const float*>(xb + off), v[j]);
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
    const int nvalid = u < g.total ? (int)((g.total - u + 255u) / 256u) : 0;

    // one member's phase-1 words from its lanes' accumulators: xor trees per wave, the four waves in index order (thread 0)
    auto reduce1 = [&](Mom& a, StWords& out) {
        if (a.ss != a.ss) { a.mn = NAN; a.mx = NAN; }          // a NaN element (k_moments: the sum of squares tells)
        a.template wave_reduce<RELU>();
        __syncthreads();
        if (lane == 0) {
            l_mn[wv] = a.mn; l_mx[wv] = a.mx; l_d[0][wv] = a.s; l_d[1][wv] = a.ss;
            if constexpr (RELU) { l_d[2][wv] = a.rs; l_d[3][wv] = a.rss; }
        }
        __syncthreads();
        Mom r;
        r.init();
        for (int i = 0; i < TPB / 64; ++i) {
            Mom o;
            o.mn = l_mn[i]; o.mx = l_mx[i]; o.s = l_d[0][i]; o.ss = l_d[1][i];
            o.rs = RELU ? l_d[2][i] : 0.; o.rss = RELU ? l_d[3][i] : 0.;
            r.template merge<true>(o);
        }
        const bool nn = (r.mn != r.mn) || (r.mx != r.mx);
        out.w[0] = pack_pair(nn ? NAN : r.mn, nn ? NAN : r.mx);
        out.w[1] = (unsigned long long)__double_as_longlong(r.s);
        out.w[2] = (unsigned long long)__double_as_longlong(r.ss);
        out.w[3] = (unsigned long long)__double_as_longlong(r.rs);
        out.w[4] = (unsigned long long)__double_as_longlong(r.rss);
    };
    auto reduce2 = [&](double a, double k4, StWords& out) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { a += shfl_xor_d(a, m); k4 += shfl_xor_d(k4, m); }
        __syncthreads();
        if (lane == 0) { l_d[0][wv] = a; l_d[1][wv] = k4; }
        __syncthreads();
        out.w[0] = (unsigned long long)__double_as_longlong(((l_d[0][0] + l_d[0][1]) + l_d[0][2]) + l_d[0][3]);
        out.w[1] = (unsigned long long)__double_as_longlong(((l_d[1][0] + l_d[1][1]) + l_d[1][2]) + l_d[1][3]);
        out.w[2] = out.w[3] = out.w[4] = 0ull;
    };
    // slot encoding of a member's words: the complement; a sum's NaN made canonical first (the pair's NaNs already are)
    auto encode = [&](StWords& sw, int nw, bool pair0) {
#pragma unroll
        for (int i = 0; i < ST_W1; ++i)
            if (i < nw) sw.w[i] = (pair0 && i == 0) ? ~sw.w[i] : slot_of_sum(__longlong_as_double((long long)sw.w[i]));
    };
    // a member's tile from x again, for the cold path: step s of member m for this lane
    auto cold_load = [&](int m, int s, float (&t)[4], bool& in) {
        const unsigned mf0 = (unsigned)m * (256u * K), mu = mf0 + (unsigned)tid;
        const int mvalid = mu < g.total ? (int)((g.total - mu + 255u) / 256u) : 0;
        in = s < mvalid;
        const unsigned ee = in ? mu + 256u * (unsigned)s : mf0;
        const unsigned nn = ee / g.cpc;
        ldv<4>(x + (size_t)nn * (size_t)g.P + (size_t)c * (size_t)g.HW + (size_t)(ee - nn * g.cpc) * 4, t);
    };

    // ---- phase 1: the register steps and the LDS steps in accumulators of their own, each in step order, merged at the end
    //      (the order of the additions does not depend on what landed first; the cold path does the same)
    Mom acc;
    acc.init();
    //      Round 6: the steps go in PAIRS through Mom::add8 (two elements per instruction; a lone last step through add4p)
    static_assert(KR % 2 == 0 && KL % 2 == 0, "steps in pairs");
#pragma unroll
    for (int j = 0; j < KR; j += 2) {
        if (j + 1 < nvalid) acc.template add8<RELU>(v[j], v[j + 1]);
        else if (j < nvalid) acc.template add4p<RELU>(v[j]);
    }
    if constexpr (KL > 0) {
        lds_dma_landed(acc.mn);
        Mom al;
        al.init();
#pragma unroll
        for (int l = 0; l < KL; l += 2) {
            const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
            const float4 q1 = *reinterpret_cast<const float4*>(sh_x + ((l + 1) * TPB + tid) * 4);
            const float t[4] = {q.x, q.y, q.z, q.w}, t1[4] = {q1.x, q1.y, q1.z, q1.w};
            if (KR + l + 1 < nvalid) al.template add8<RELU>(t, t1);
            else if (KR + l < nvalid) al.template add4p<RELU>(t);
        }
        acc.template merge<true>(al);
    }
    StWords sw;
    reduce1(acc, sw);
    constexpr int NW1 = RELU ? ST_W1 : 3;
    unsigned long long* lines = ws.slots + (size_t)c * ws.gstride * ST_LINE;      // zero at rest
    encode(sw, NW1, true);
    st_publish(lines + (size_t)member * ST_LINE, sw.w, 0, NW1);
    __syncthreads();                                   // sh_code is set; l_* are free again
    const long long tmo = (sh_code & 4) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS;
    StFold fold;
    fold.init();
    if (!(sh_code & 2) && st_fold_poll(fold, lines, g.Gs, 0, NW1, true, tmo, &sh_code)) st_fold_finish(fold, NW1, true, sh_out, sh_mm);
    __syncthreads();
    bool cold = (sh_code & 3) != 0;
    if (cold) {
        // every member's words from x with that member's own lane mapping, folded in the same order
        if (tid == 0) atomicOr(ws.status, (unsigned)(sh_code & 3));
        fold.init();
        for (int m = 0; m < g.Gs; ++m) {
            Mom a2, al2;
            a2.init();
            al2.init();
            for (int s = 0; s < K; s += 2) {
                float t[4], t1[4];
                bool in, in1;
                cold_load(m, s, t, in);
                cold_load(m, s + 1, t1, in1);
                if (in1) {
                    if (s >= KR) al2.template add8<RELU>(t, t1);
                    else a2.template add8<RELU>(t, t1);
                } else if (in) {
                    if (s >= KR) al2.template add4p<RELU>(t);
                    else a2.template add4p<RELU>(t);
                }
            }
            if constexpr (KL > 0) a2.template merge<true>(al2);
            StWords s2;
            reduce1(a2, s2);
            st_fold_add(fold, m, s2.w, NW1, true);
        }
        st_fold_finish(fold, NW1, true, sh_out, sh_mm);
        __syncthreads();
        // the tile is dead on this path (phase 2 recomputes too): say so, and its registers serve the loop above
#pragma unroll
        for (int j = 0; j < KR; ++j) { v[j][0] = 0.f; v[j][1] = 0.f; v[j][2] = 0.f; v[j][3] = 0.f; }
    }
    if constexpr (XR) {
        // the batch is sharded: lane w of wave 0 exchanges word w of the rank's fold with the other ranks (word 0: the pair; 1 ..
        // NW1 - 1: sums; ST_XW_COUNT: the rank's element count) - rank order, the same bits on every rank
        if (tid <= ST_XW_COUNT && (tid < NW1 || tid == ST_XW_COUNT)) {
            unsigned long long bits;
            if (tid == 0) bits = (unsigned long long)__float_as_uint(sh_mm[0]) | ((unsigned long long)__float_as_uint(sh_mm[1]) << 32);
            else bits = (unsigned long long)__double_as_longlong(tid == ST_XW_COUNT ? sa.count : sh_out[tid]);
            (void)xr_merge_word(xr, tid * g.C + c, member == 0, tid == 0, bits);
            if (tid == 0) {
                sh_mm[0] = __uint_as_float((unsigned)(bits & 0xffffffffull));
                sh_mm[1] = __uint_as_float((unsigned)(bits >> 32));
            } else if (tid == ST_XW_COUNT) {
                sh_count = __longlong_as_double((long long)bits);
            } else {
                sh_out[tid] = __longlong_as_double((long long)bits);
            }
        }
        __syncthreads();
    }
    // (re-read from LDS at every use under XR: a double held across phase 2 next to the tile is a spill candidate)
    auto count_of = [&]() -> double { if constexpr (XR) return sh_count; else return sa.count; };
    // mean / std / std_pos: every lane derives the same values (the formulas of k_combine); the folded words stay in LDS for
    // the final row - seven doubles held across phase 2 next to the tile were spilled
    float mean, sd, std_pos = 0.f;
    {
        const MomSum r{(double)sh_mm[0], (double)sh_mm[1], sh_out[1], sh_out[2], count_of(), RELU ? sh_out[3] : 0., RELU ? sh_out[4] : 0.};
        mean = mean_of(r);
        sd = std_of(r);
        if constexpr (RELU) {
            double rv = (r.rss - r.rs * (r.rs / r.cnt)) / (r.cnt - 1.);
            if (rv < 0.) rv = 0.;
            std_pos = (float)sqrt(rv);
        }
    }
    float vb = 0.f, kurt = 0.f;
    bool writer = member == 0;                  // who writes the channel's row: member 0, or - with phase 2 - its last arriver
    if (sa.need_dev) {
        // ---- phase 2 (k_absdev's arithmetic: fp32 difference, reciprocal of the std, fp64 sums)
        const float isd = sa.need_kurt ? 1.f / sd : 0.f;
        // per element k_absdev's arithmetic; round 6: the sums of a pair of steps in fp32, two elements per instruction (as add8),
        // one fp64 addition per eight elements and sum - 3.75 instead of 8 instructions per element
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 m2 = {mean, mean}, i2 = {isd, isd};
        auto dev8 = [&](const float (&t)[4], const float (&t1)[4], double& a, double& k4) {
            const f2 d0 = f2{t[0], t[1]} - m2, d1 = f2{t[2], t[3]} - m2, d2 = f2{t1[0], t1[1]} - m2, d3 = f2{t1[2], t1[3]} - m2;
            const f2 ab = f2{abs_add(d0.x, d0.y), abs_add(d1.x, d1.y)} + f2{abs_add(d2.x, d2.y), abs_add(d3.x, d3.y)};
            a += (double)hadd(ab);
            f2 z0 = d0 * i2, z1 = d1 * i2, z2 = d2 * i2, z3 = d3 * i2;
            z0 = z0 * z0; z1 = z1 * z1; z2 = z2 * z2; z3 = z3 * z3;
            const f2 q = ((z0 * z0 + z1 * z1) + z2 * z2) + z3 * z3;
            k4 += (double)hadd(q);
        };
        auto dev4p = [&](const float (&t)[4], double& a, double& k4) {
            const f2 d0 = f2{t[0], t[1]} - m2, d1 = f2{t[2], t[3]} - m2;
            a += (double)hadd(f2{abs_add(d0.x, d0.y), abs_add(d1.x, d1.y)});
            f2 z0 = d0 * i2, z1 = d1 * i2;
            z0 = z0 * z0; z1 = z1 * z1;
            const f2 q = z0 * z0 + z1 * z1;
            k4 += (double)hadd(q);
        };
        // Round 6: nobody WAITS for the second meeting any more.  A member publishes its two words and counts its arrival; whoever
        // arrives LAST - every other member's words are out by then - folds them (member order), exchanges them with the other
        // ranks (XR), writes the channel's row and re-arms the group's lines; everybody else leaves at once and its slot on the
        // CU goes to the next workgroup's loads.  (Round 5 had every member poll the second meeting too: a quarter of a
        // workgroup's life with its 128 KB of registers neither loading nor needed - the memory system idled through it.)
        bool have_fold = false;
        if (!cold) {
            // out of the registers.  (On the cold path the tile is NOT used again - this member's words come out of the loop
            // below like everybody's - so its registers are free there: the recompute code next to a live tile spilled it.)
            double da = 0., dk = 0.;
#pragma unroll
            for (int j = 0; j < KR; j += 2) {
                if (j + 1 < nvalid) dev8(v[j], v[j + 1], da, dk);
                else if (j < nvalid) dev4p(v[j], da, dk);
            }
            if constexpr (KL > 0) {
                double la = 0., lk = 0.;
#pragma unroll
                for (int l = 0; l < KL; l += 2) {
                    const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
                    const float4 q1 = *reinterpret_cast<const float4*>(sh_x + ((l + 1) * TPB + tid) * 4);
                    const float t[4] = {q.x, q.y, q.z, q.w}, t1[4] = {q1.x, q1.y, q1.z, q1.w};
                    if (KR + l + 1 < nvalid) dev8(t, t1, la, lk);
                    else if (KR + l < nvalid) dev4p(t, la, lk);
                }
                da = da + la;
                dk = dk + lk;
            }
            reduce2(da, dk, sw);
            encode(sw, ST_W2, false);
            st_publish(lines + (size_t)member * ST_LINE, sw.w, ST_W1, ST_W2);
        } else {
            // a member whose first meeting failed (or the test hook): every member's words from x with that member's own lane
            // mapping, folded in member order - the same bits whoever ends up writing the row
            fold.init();
            for (int m = 0; m < g.Gs; ++m) {
                double a2 = 0., k2 = 0., la2 = 0., lk2 = 0.;
                for (int s = 0; s < K; s += 2) {
                    float t[4], t1[4];
                    bool in, in1;
                    cold_load(m, s, t, in);
                    cold_load(m, s + 1, t1, in1);
                    if (in1) {
                        if (s >= KR) dev8(t, t1, la2, lk2);
                        else dev8(t, t1, a2, k2);
                    } else if (in) {
                        if (s >= KR) dev4p(t, la2, lk2);
                        else dev4p(t, a2, k2);
                    }
                }
                if constexpr (KL > 0) {
                    a2 = a2 + la2;
                    k2 = k2 + lk2;
                }
                StWords s2;
                reduce2(a2, k2, s2);
                st_fold_add(fold, m, s2.w, ST_W2, false);
                if (m == member) {                      // its own words go out like everybody's: the last arriver may be somebody else
                    encode(s2, ST_W2, false);
                    st_publish(lines + (size_t)member * ST_LINE, s2.w, ST_W1, ST_W2);
                }
            }
            __syncthreads();
            st_fold_finish(fold, ST_W2, false, sh_out2, sh_mm);
            have_fold = true;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the words have left the CU before the arrival is counted
        __syncthreads();
        if (tid == 0) sh_last = grp_arrive_last(grp_lines(ws.cnt, c, g.Gs, 0, 1), member, g.Gs) ? 1 : 0;
        __syncthreads();
        writer = sh_last != 0;
        if (writer) {
            if (!have_fold) {
                fold.init();
                if (st_fold_poll(fold, lines, g.Gs, ST_W1, ST_W2, false, tmo, &sh_code)) st_fold_finish(fold, ST_W2, false, sh_out2, sh_mm);
                __syncthreads();
            }
            if constexpr (XR) {
                if (tid < ST_W2) {
                    double s2 = sh_out2[tid];
                    (void)xr_merge_sum(xr, (ST_XW_COUNT + 1 + tid) * g.C + c, true, s2);     // this rank's one push per word
                    sh_out2[tid] = s2;
                }
                __syncthreads();
            }
            vb = (float)(sh_out2[0] / count_of());
            kurt = sa.need_kurt ? (float)(sh_out2[1] / count_of() - 3.) : 0.f;
        }
    }
    if (writer && tid == 0) {
        const size_t C = (size_t)g.C;
        sa.stats[(size_t)CNNQ_STAT_MIN * C + c] = sh_mm[0];
        sa.stats[(size_t)CNNQ_STAT_MAX * C + c] = sh_mm[1];
        sa.stats[(size_t)CNNQ_STAT_MEAN * C + c] = mean;
        sa.stats[(size_t)CNNQ_STAT_STD * C + c] = sd;
        sa.stats[(size_t)CNNQ_STAT_STD_POS * C + c] = std_pos;
        sa.stats[(size_t)CNNQ_STAT_B * C + c] = vb;
        sa.stats[(size_t)CNNQ_STAT_KURT * C + c] = kurt;
        if (sa.mom) {
            sa.mom[(size_t)CNNQ_MOM_MIN * C + c] = (double)sh_mm[0];
            sa.mom[(size_t)CNNQ_MOM_MAX * C + c] = (double)sh_mm[1];
            sa.mom[(size_t)CNNQ_MOM_SUM * C + c] = sh_out[1];
            sa.mom[(size_t)CNNQ_MOM_SUMSQ * C + c] = sh_out[2];
            sa.mom[(size_t)CNNQ_MOM_COUNT * C + c] = count_of();
            sa.mom[(size_t)CNNQ_MOM_SUM_RELU * C + c] = RELU ? sh_out[3] : 0.;
            sa.mom[(size_t)CNNQ_MOM_SUMSQ_RELU * C + c] = RELU ? sh_out[4] : 0.;
        }
    }
    // ---- leave the group.  With phase 2 the last ARRIVER above is also the last to need the lines: it re-arms them; without it
    //      the departures are counted and the last member out does
    __syncthreads();
    if (!sa.need_dev) {
        if (tid == 0) sh_last = grp_depart_last(grp_lines(ws.cnt, c, g.Gs, 0, 1), member, g.Gs) ? 1 : 0;
        __syncthreads();
    }
    if (sh_last)
        for (int m = tid; m < g.Gs * ST_LINE; m += TPB) __hip_atomic_store(lines + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---- row-piece tiles (the geometry of k_mmq_group / k_fused_group: a member is <= 256 float4 columns x <= K samples) -------
// The layers the flat tiles do not take: short channel rows (ResNet-50's 14x14 and 7x7 layers - a workgroup owns k whole
// channels, its lanes one channel each or, where H*W is not a multiple of 4, one channel per ELEMENT), and rows that are
// whole multiples of the workgroup.  A member publishes SG_NP words per owned channel: plane 0 the {min, max} pair, planes
// 1-4 the sums of phase 1, planes 5-6 the sums of phase 2; [member][plane][k] 8-byte words in the group's block of the slot
// region, every one of them the complement of its value (never zero: zero is "not arrived").  One poll per phase takes all
// planes at once: lane (plane, channel, j) folds the words of members j, j + L, ... in member order.
constexpr int SG_NP = 7;

template <int A, bool RELU>
struct SgAcc {
    float mn[A], mx[A];
    double s[A], ss[A], rs[RELU ? A : 1], rss[RELU ? A : 1];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            mn[a] = INFINITY; mx[a] = -INFINITY; s[a] = 0.; ss[a] = 0.;
            if constexpr (RELU) { rs[a] = 0.; rss[a] = 0.; }
        }
    }
    // one float4 of the lane's column: A == 4 - one value for each of four accumulator sets (Mom::add); (A == 1 - four values of
    // one channel as Mom::add4 - is no longer used by the kernel: its rows go in pairs, below)
    __device__ __forceinline__ void add(const float (&v)[4]) {
        if constexpr (A == 1) {
            mn[0] = fminf(fminf(mn[0], fminf(v[0], v[1])), fminf(v[2], v[3]));
            mx[0] = fmaxf(fmaxf(mx[0], fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            s[0] += (double)((v[0] + v[1]) + (v[2] + v[3]));
            ss[0] += (double)((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
            if constexpr (RELU) {
                const float r0 = fmaxf(v[0], 0.f), r1 = fmaxf(v[1], 0.f), r2 = fmaxf(v[2], 0.f), r3 = fmaxf(v[3], 0.f);
                rs[0] += (double)((r0 + r1) + (r2 + r3));
                rss[0] += (double)((r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mn[e] = pmin(mn[e], v[e]);
                mx[e] = pmax(mx[e], v[e]);
                const double d = (double)v[e];
                s[e] += d;
                ss[e] = fma(d, d, ss[e]);
                if constexpr (RELU) {
                    const double r = (double)fmaxf(v[e], 0.f);
                    rs[e] += r;
                    rss[e] = fma(r, r, rss[e]);
                }
            }
        }
    }
    // A == 1, round 6: a PAIR of float4 (two rows of the tile) as Mom::add8, a lone last row as Mom::add4p - two elements per
    // instruction, one fp64 addition per eight elements and sum
    __device__ __forceinline__ void add8(const float (&a)[4], const float (&b)[4]) {
        static_assert(A == 1, "one channel per lane");
        Mom m;
        m.mn = mn[0]; m.mx = mx[0]; m.s = s[0]; m.ss = ss[0]; m.rs = RELU ? rs[0] : 0.; m.rss = RELU ? rss[0] : 0.;
        m.template add8<RELU>(a, b);
        mn[0] = m.mn; mx[0] = m.mx; s[0] = m.s; ss[0] = m.ss;
        if constexpr (RELU) { rs[0] = m.rs; rss[0] = m.rss; }
    }
    __device__ __forceinline__ void add4p(const float (&a)[4]) {
        static_assert(A == 1, "one channel per lane");
        Mom m;
        m.mn = mn[0]; m.mx = mx[0]; m.s = s[0]; m.ss = ss[0]; m.rs = RELU ? rs[0] : 0.; m.rss = RELU ? rss[0] : 0.;
        m.template add4p<RELU>(a);
        mn[0] = m.mn; mx[0] = m.mx; s[0] = m.s; ss[0] = m.ss;
        if constexpr (RELU) { rs[0] = m.rs; rss[0] = m.rss; }
    }
    // rows r and r + 1 of a tile of nr rows (the pair is whole, or its first row is the tile's last, or it lies past the tile)
    __device__ __forceinline__ void add_rows(const float (&t0)[4], const float (&t1)[4], int r, int nr) {
        if constexpr (A == 1) {
            if (r + 1 < nr) add8(t0, t1);
            else if (r < nr) add4p(t0);
        } else {
            if (r < nr) add(t0);
            if (r + 1 < nr) add(t1);
        }
    }
    __device__ __forceinline__ void poison() {          // A == 1: v_min / v_max dropped a NaN; the sum of squares tells
        if constexpr (A == 1)
            if (ss[0] != ss[0]) { mn[0] = NAN; mx[0] = NAN; }
    }
};

// lane accumulators -> the tile's per-channel values: sh_mn / sh_mx [nch] and sh_q[plane][nch] (sum, sum of squares, relu sums)
template <int A, bool RELU>
__device__ __forceinline__ void sg_reduce1(const Geo& g, const Blk& b, bool ok, SgAcc<A, RELU>& ac, double* l_a, float* sh_mn,
                                           float* sh_mx, double* sh_q) {
    ac.poison();
    float* l_mn = reinterpret_cast<float*>(l_a);
    float* l_mx = l_mn + TPB * A;
    __syncthreads();                     // l_a and the tables may still be read from a previous call
    wg_channel_minmax<A>(g, b, ok, ac.mn, ac.mx, l_mn, l_mx, sh_mn, sh_mx);
    wg_channel_sums<A>(g, b, ok, ac.s, l_a, sh_q);
    wg_channel_sums<A>(g, b, ok, ac.ss, l_a, sh_q + MAXCH);
    if constexpr (RELU) {
        wg_channel_sums<A>(g, b, ok, ac.rs, l_a, sh_q + 2 * MAXCH);
        wg_channel_sums<A>(g, b, ok, ac.rss, l_a, sh_q + 3 * MAXCH);
    }
}

// The fold of np planes starting at plane p0 of every member's block ([member][SG_NP * kk] words at src, complemented).
// POLL: the words are slots (zero = not arrived; bounded wait, on expiry 1 is OR-ed into *sh_code and the function returns);
// otherwise the cold path's table.  Plane 0 is the pair plane.  Results: sh_mn / sh_mx [ch] and out[(plane - first sum
// plane) * MAXCH + ch].
template <int W, bool POLL>
__device__ __forceinline__ void sg_fold(const unsigned long long* src, int Gs, int kk, int nch, int p0, int np, long long tmo,
                                        double* out, float* sh_mn, float* sh_mx, int* sh_code) {
    const int tid = threadIdx.x;
    const int nvc = np * kk;
    int L = 1;
    {
        const int nv = nvc < TPB ? nvc : TPB;
        while (L < 64 && 2 * L * nv <= TPB) L <<= 1;
    }
    const int per = TPB / L, mstride = SG_NP * kk;
    const int sub = tid / L, j = tid - sub * L;
    long long t0 = 0;
    int spins = 0;
    for (int v0 = 0; v0 < nvc; v0 += per) {
        const int vc = v0 + sub;
        const int pl = vc / kk, ch = vc - pl * kk;      // plane (relative to p0), channel
        const bool active = vc < nvc && ch < nch;
        const bool is_pair = (p0 + pl) == 0;
        double acc = 0.;
        float amn = INFINITY, amx = -INFINITY;
        for (int w0 = 0; w0 * L < Gs; w0 += W) {
            const unsigned long long* p = src + (size_t)(j + L * w0) * mstride + (size_t)(p0 * kk + vc);
            const size_t step = (size_t)L * mstride;
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
                        expired = (now - t0 > tmo || spins > GRP_TIMEOUT_SPINS) ? 1 : 0;
                    }
                    if (__builtin_amdgcn_readfirstlane(expired)) {
                        if ((tid & 63) == 0) atomicOr(sh_code, 1);
                        return;
                    }
                    if (spins < 2) __builtin_amdgcn_s_sleep(8);
                    else if (spins < 6) __builtin_amdgcn_s_sleep(32);
                    else __builtin_amdgcn_s_sleep(64);
                }
            } else {
#pragma unroll
                for (int i = 0; i < W; ++i)
                    if ((mine >> i) & 1u) v[i] = __hip_atomic_load(p + i * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int i = 0; i < W; ++i)
                if ((mine >> i) & 1u) {                  // member order, whatever the arrival order was
                    if (is_pair) {
                        float a, b2;
                        unpack_pair(~v[i], a, b2);
                        amn = pmin(amn, a);
                        amx = pmax(amx, b2);
                    } else {
                        acc += sum_of_slot(v[i]);
                    }
                }
        }
        for (int m = L >> 1; m >= 1; m >>= 1) {
            acc += shfl_xor_d(acc, m);
            amn = pmin(amn, shfl_xor_f(amn, m));
            amx = pmax(amx, shfl_xor_f(amx, m));
        }
        if (active && j == 0) {
            if (is_pair) { sh_mn[ch] = amn; sh_mx[ch] = amx; }
            else out[(size_t)(pl - (p0 == 0 ? 1 : 0)) * MAXCH + ch] = acc;
        }
    }
}

template <int A, int KR, int KL, bool RELU, bool XR = false>
__global__ void __launch_bounds__(TPB, (KR + KL == 32 ? GRP_K32_WAVES : 1)) k_stats_group(const float* __restrict__ x, const Geo g, const int Gs,
                                                                                        const GWs ws, const St1Args sa, const unsigned flags,
                                                                                        const XRank xr = XRank{}) {
    constexpr int K = KR + KL;
    constexpr int NP1 = RELU ? 5 : 3;
    static_assert(MAXCH <= TPB, "one lane per channel");
    if constexpr (XR) xr_prologue(xr);                 // workgroup 0: the slots of the launch two back
    __shared__ double sh_cnt[XR ? MAXCH : 1];          // XR: the global batch's elements per channel
    __shared__ int sh_last;
    __shared__ __attribute__((aligned(16))) float sh_x[KL ? KL * TPB * 4 : 4];
    __shared__ double l_a[TPB * A];
    __shared__ double sh_q[4 * MAXCH];            // phase 1: sum, sum of squares, relu sums; phase 2: sum |x - mean|, sum z^4
    __shared__ float sh_mn[MAXCH], sh_mx[MAXCH], sh_mean[MAXCH], sh_isd[MAXCH];
    __shared__ int sh_code;
    const unsigned st0 = threadIdx.x == 0 ? __hip_atomic_load(ws.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const RBlk rb = rblk_of(g, Gs);
    const Blk& b = rb.b;
    const int tid = threadIdx.x;
    const int nch = b.c1 - b.c0;
    const bool ok = b.col0 + tid < b.col1;
    const unsigned colc = (unsigned)(ok ? b.col0 + tid : b.col0);      // idle lanes re-read the block's first column; results discarded
    const int nrows = b.n1 - b.n0;                                     // 1 .. K
    const size_t base = (size_t)b.n0 * (size_t)g.P + (size_t)colc * 4;
    const int kk = (g.mode == 1) ? 1 : g.k;
    const int NPK = SG_NP * kk;
    if (tid == 0) sh_code = ((flags & MMQ_FLAG_TEST_HOOK) ? 2 : 0) | ((st0 & 1u) ? 4 : 0);

    // ---- the tile: K 16-byte loads per lane back to back, the LAST KL of them straight into LDS (as asm behind the register
    //      loads: lds_dma16_behind - the pairs of phase 1 start as their own loads land)
    static_assert(KR % 2 == 0 && KL % 2 == 0, "rows in pairs");
    float v[KR][4];
#pragma unroll
    for (int j = 0; j < KR; ++j) {
        const int r = j < nrows ? j : nrows - 1;
        ldv_nt<4>(x + base + (size_t)r * (size_t)g.P, v[j]);
    }
    if constexpr (KL > 0) {
        // (tiles with LDS rows are one-channel-per-lane tiles of whole K-row splits or less: offsets fit 32 bits from the tile's base)
        const char* xt = reinterpret_cast<const char*>(x + (size_t)b.n0 * (size_t)g.P);
#pragma unroll
        for (int l = 0; l < KL; ++l) {
            const int r = KR + l < nrows ? KR + l : nrows - 1;
            lds_dma16_behind(xt, (unsigned)(((size_t)r * (size_t)g.P + (size_t)colc * 4) * 4), sh_x + (l * TPB + (tid & ~63)) * 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 1, rows in order and in pairs (SgAcc::add_rows)
    SgAcc<A, RELU> ac;
    ac.init();
#pragma unroll
    for (int j = 0; j < KR; j += 2) ac.add_rows(v[j], v[j + 1], j, nrows);
    if constexpr (KL > 0) {
        lds_dma_landed(ac.mn[0]);
#pragma unroll
        for (int l = 0; l < KL; l += 2) {
            const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
            const float4 q1 = *reinterpret_cast<const float4*>(sh_x + ((l + 1) * TPB + tid) * 4);
            const float t[4] = {q.x, q.y, q.z, q.w}, t1[4] = {q1.x, q1.y, q1.z, q1.w};
            ac.add_rows(t, t1, KR + l, nrows);
        }
    }
    sg_reduce1<A, RELU>(g, b, ok, ac, l_a, sh_mn, sh_mx, sh_q);

    // a member's tile from x again (the cold path), with that member's own lane mapping
    auto cold_phase1 = [&](const RBlk& mb) {
        const int mcol = mb.b.col0 + tid;
        const bool mok = mcol < mb.b.col1;
        const int mcolc = mok ? mcol : mb.b.col0;
        const int mrows = mb.b.n1 - mb.b.n0;
        SgAcc<A, RELU> a2;
        a2.init();
        for (int jj = 0; jj < K; jj += 2) {
            float t[4], t1[4];
            const int r = jj < mrows ? jj : mrows - 1, r1 = jj + 1 < mrows ? jj + 1 : mrows - 1;
            ldv<4>(x + ((size_t)(mb.b.n0 + r) * (size_t)g.P + (size_t)mcolc * 4), t);
            ldv<4>(x + ((size_t)(mb.b.n0 + r1) * (size_t)g.P + (size_t)mcolc * 4), t1);
            a2.add_rows(t, t1, jj, mrows);
        }
        sg_reduce1<A, RELU>(g, mb.b, mok, a2, l_a, sh_mn, sh_mx, sh_q);
    };
    // the tile's per-channel words of planes [p0, p0 + np) out of the LDS tables into a member's block at dst
    auto emit = [&](unsigned long long* dst, int p0, int np) {
        for (int vc = tid; vc < np * kk; vc += TPB) {
            const int pl = vc / kk, ch = vc - pl * kk;
            if (ch < nch) {
                const int p = p0 + pl;
                const unsigned long long wv = (p == 0) ? slot_of(sh_mn[ch], sh_mx[ch]) : slot_of_sum(sh_q[(size_t)(p - (p0 == 0 ? 1 : p0)) * MAXCH + ch]);
                __hip_atomic_store(dst + (size_t)p * kk + ch, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    unsigned long long* lines = ws.slots + (size_t)rb.group * ws.gstride;      // zero at rest
    unsigned long long* tab = ws.part + (size_t)rb.group * ws.gstride;         // the cold path's copy of the same block
    emit(lines + (size_t)rb.member * NPK, 0, NP1);
    __syncthreads();                                   // sh_code is set; the tables are free again
    const long long tmo = (sh_code & 4) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS;
    if (!(sh_code & 2)) sg_fold<4, true>(lines, Gs, kk, nch, 0, NP1, tmo, sh_q, sh_mn, sh_mx, &sh_code);
    __syncthreads();
    bool cold = (sh_code & 3) != 0;
    if (cold) {
        if (tid == 0) atomicOr(ws.status, (unsigned)(sh_code & 3));
        for (int m = 0; m < Gs; ++m) {
            const RBlk mb = rblk_at(g, rb.group, m);
            cold_phase1(mb);
            emit(tab + (size_t)m * NPK, 0, NP1);       // every workgroup that lands here writes the same values
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sg_fold<4, false>(tab, Gs, kk, nch, 0, NP1, 0, sh_q, sh_mn, sh_mx, &sh_code);
        __syncthreads();
    }
    if constexpr (XR) {
        // the batch is sharded (round 6): the owned channels' folded words - the pair, the sums, the rank's element count -
        // exchanged with the other ranks, rank order (cnnq_xrank.hip.h: slot = word * C + channel); member 0 pushes
        for (int i = tid; i < nch * (ST_XW_COUNT + 1); i += TPB) {
            const int w = i / nch, ch = i - w * nch;
            if (w < NP1 || w == ST_XW_COUNT) {
                unsigned long long bits;
                if (w == 0) bits = (unsigned long long)__float_as_uint(sh_mn[ch]) | ((unsigned long long)__float_as_uint(sh_mx[ch]) << 32);
                else bits = (unsigned long long)__double_as_longlong(w == ST_XW_COUNT ? sa.count : sh_q[(size_t)(w - 1) * MAXCH + ch]);
                (void)xr_merge_word(xr, w * g.C + b.c0 + ch, rb.member == 0, w == 0, bits);
                if (w == 0) {
                    sh_mn[ch] = __uint_as_float((unsigned)(bits & 0xffffffffull));
                    sh_mx[ch] = __uint_as_float((unsigned)(bits >> 32));
                } else if (w == ST_XW_COUNT) {
                    sh_cnt[ch] = __longlong_as_double((long long)bits);
                } else {
                    sh_q[(size_t)(w - 1) * MAXCH + ch] = __longlong_as_double((long long)bits);
                }
            }
        }
        __syncthreads();
    }
    auto count_of = [&](int ch) -> double { if constexpr (XR) return sh_cnt[ch]; else return sa.count; };
    // ---- mean / std / std_pos of the owned channels: identical in every member (the formulas of k_combine)
    if (tid < nch) {
        const int c = b.c0 + tid;
        const MomSum r{(double)sh_mn[tid], (double)sh_mx[tid], sh_q[tid], sh_q[MAXCH + tid], count_of(tid), RELU ? sh_q[2 * MAXCH + tid] : 0.,
                       RELU ? sh_q[3 * MAXCH + tid] : 0.};
        const float mean = mean_of(r), sd = std_of(r);
        sh_mean[tid] = mean;
        sh_isd[tid] = sa.need_kurt ? 1.f / sd : 0.f;
        if (rb.member == 0) {
            float std_pos = 0.f;
            if constexpr (RELU) {
                double rv = (r.rss - r.rs * (r.rs / r.cnt)) / (r.cnt - 1.);
                if (rv < 0.) rv = 0.;
                std_pos = (float)sqrt(rv);
            }
            const size_t C = (size_t)g.C;
            sa.stats[(size_t)CNNQ_STAT_MIN * C + c] = sh_mn[tid];
            sa.stats[(size_t)CNNQ_STAT_MAX * C + c] = sh_mx[tid];
            sa.stats[(size_t)CNNQ_STAT_MEAN * C + c] = mean;
            sa.stats[(size_t)CNNQ_STAT_STD * C + c] = sd;
            sa.stats[(size_t)CNNQ_STAT_STD_POS * C + c] = std_pos;
            if (!sa.need_dev) {
                sa.stats[(size_t)CNNQ_STAT_B * C + c] = 0.f;
                sa.stats[(size_t)CNNQ_STAT_KURT * C + c] = 0.f;
            }
            if (sa.mom) {
                sa.mom[(size_t)CNNQ_MOM_MIN * C + c] = r.mn;
                sa.mom[(size_t)CNNQ_MOM_MAX * C + c] = r.mx;
                sa.mom[(size_t)CNNQ_MOM_SUM * C + c] = r.s;
                sa.mom[(size_t)CNNQ_MOM_SUMSQ * C + c] = r.ss;
                sa.mom[(size_t)CNNQ_MOM_COUNT * C + c] = r.cnt;
                sa.mom[(size_t)CNNQ_MOM_SUM_RELU * C + c] = r.rs;
                sa.mom[(size_t)CNNQ_MOM_SUMSQ_RELU * C + c] = r.rss;
            }
        }
    }
    __syncthreads();
    if (sa.need_dev) {
        // ---- phase 2 (k_absdev's arithmetic: fp32 difference, reciprocal of the std, fp64 sums)
        auto dev_add = [&](const float (&t)[4], const float (&mean)[A], const float (&isd)[A], double (&da)[A], double (&dk)[A]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = A == 1 ? 0 : e;
                const float d = t[e] - mean[a];
                da[a] += (double)fabsf(d);
                const float z = d * isd[a];
                const float z2 = z * z;
                dk[a] += (double)(z2 * z2);
            }
        };
        // rows r and r + 1 of a tile of nr rows: A == 1 - the pair's sums in fp32, two elements per instruction (k_stats_flat's
        // dev8 / dev4p); A == 4 - row by row
        auto dev_rows = [&](const float (&t0)[4], const float (&t1)[4], int r, int nr, const float (&mean)[A], const float (&isd)[A],
                            double (&da)[A], double (&dk)[A]) {
            if constexpr (A == 1) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 m2 = {mean[0], mean[0]}, i2 = {isd[0], isd[0]};
                if (r + 1 < nr) {
                    const f2 d0 = f2{t0[0], t0[1]} - m2, d1 = f2{t0[2], t0[3]} - m2, d2 = f2{t1[0], t1[1]} - m2, d3 = f2{t1[2], t1[3]} - m2;
                    const f2 ab = f2{abs_add(d0.x, d0.y), abs_add(d1.x, d1.y)} + f2{abs_add(d2.x, d2.y), abs_add(d3.x, d3.y)};
                    da[0] += (double)hadd(ab);
                    f2 z0 = d0 * i2, z1 = d1 * i2, z2 = d2 * i2, z3 = d3 * i2;
                    z0 = z0 * z0; z1 = z1 * z1; z2 = z2 * z2; z3 = z3 * z3;
                    const f2 q = ((z0 * z0 + z1 * z1) + z2 * z2) + z3 * z3;
                    dk[0] += (double)hadd(q);
                } else if (r < nr) {
                    const f2 d0 = f2{t0[0], t0[1]} - m2, d1 = f2{t0[2], t0[3]} - m2;
                    da[0] += (double)hadd(f2{abs_add(d0.x, d0.y), abs_add(d1.x, d1.y)});
                    f2 z0 = d0 * i2, z1 = d1 * i2;
                    z0 = z0 * z0; z1 = z1 * z1;
                    const f2 q = z0 * z0 + z1 * z1;
                    dk[0] += (double)hadd(q);
                }
            } else {
                if (r < nr) dev_add(t0, mean, isd, da, dk);
                if (r + 1 < nr) dev_add(t1, mean, isd, da, dk);
            }
        };
        auto reduce2 = [&](const Blk& bb, bool bok, const double (&da)[A], const double (&dk)[A]) {
            wg_channel_sums<A>(g, bb, bok, da, l_a, sh_q);
            wg_channel_sums<A>(g, bb, bok, dk, l_a, sh_q + MAXCH);
        };
        // Round 6, as k_stats_flat: nobody waits for the second meeting - the LAST arriver folds it (member order), exchanges it
        // with the other ranks (XR), writes rows B / KURT and re-arms the group's block
        bool have_fold = false;
        if (!cold) {
            // out of the registers.  (On the cold path the tile is NOT used again: its registers are free there.)
            int tidq = threadIdx.x;
            asm volatile("" : "+v"(tidq));             // the lane's channel indices again, not carried across the meeting
            const unsigned colq = (unsigned)(b.col0 + tidq < b.col1 ? b.col0 + tidq : b.col0);
            float mean[A], isd[A];
            double da[A], dk[A];
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const int chl = (int)((colq * 4u + (unsigned)a) / (unsigned)g.HW) - b.c0;
                mean[a] = sh_mean[chl];
                isd[a] = sh_isd[chl];
                da[a] = 0.;
                dk[a] = 0.;
            }
#pragma unroll
            for (int j = 0; j < KR; j += 2) dev_rows(v[j], v[j + 1], j, nrows, mean, isd, da, dk);
            if constexpr (KL > 0) {
#pragma unroll
                for (int l = 0; l < KL; l += 2) {
                    const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
                    const float4 q1 = *reinterpret_cast<const float4*>(sh_x + ((l + 1) * TPB + tid) * 4);
                    const float t[4] = {q.x, q.y, q.z, q.w}, t1[4] = {q1.x, q1.y, q1.z, q1.w};
                    dev_rows(t, t1, KR + l, nrows, mean, isd, da, dk);
                }
            }
            reduce2(b, ok, da, dk);
            emit(lines + (size_t)rb.member * NPK, 5, 2);
        } else {
            for (int m = 0; m < Gs; ++m) {
                const RBlk mb = rblk_at(g, rb.group, m);
                const int mcol = mb.b.col0 + tid;
                const bool mok = mcol < mb.b.col1;
                const int mcolc = mok ? mcol : mb.b.col0;
                const int mrows = mb.b.n1 - mb.b.n0;
                float mean[A], isd[A];
                double da[A], dk[A];
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    const int chl = (int)(((unsigned)mcolc * 4u + (unsigned)a) / (unsigned)g.HW) - mb.b.c0;
                    mean[a] = sh_mean[chl];
                    isd[a] = sh_isd[chl];
                    da[a] = 0.;
                    dk[a] = 0.;
                }
                for (int jj = 0; jj < K; jj += 2) {
                    float t[4], t1[4];
                    const int r = jj < mrows ? jj : mrows - 1, r1 = jj + 1 < mrows ? jj + 1 : mrows - 1;
                    ldv<4>(x + ((size_t)(mb.b.n0 + r) * (size_t)g.P + (size_t)mcolc * 4), t);
                    ldv<4>(x + ((size_t)(mb.b.n0 + r1) * (size_t)g.P + (size_t)mcolc * 4), t1);
                    dev_rows(t, t1, jj, mrows, mean, isd, da, dk);
                }
                reduce2(mb.b, mok, da, dk);
                emit(tab + (size_t)m * NPK, 5, 2);
                if (m == rb.member) emit(lines + (size_t)rb.member * NPK, 5, 2);    // its own words go out like everybody's
                __syncthreads();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            sg_fold<4, false>(tab, Gs, kk, nch, 5, 2, 0, sh_q, sh_mn, sh_mx, &sh_code);
            __syncthreads();
            have_fold = true;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the words have left the CU before the arrival is counted
        __syncthreads();
        if (tid == 0) sh_last = grp_arrive_last(grp_lines(ws.cnt, rb.group, Gs, 0, 1), rb.member, Gs) ? 1 : 0;
        __syncthreads();
        if (sh_last) {
            if (!have_fold) {
                sg_fold<4, true>(lines, Gs, kk, nch, 5, 2, tmo, sh_q, sh_mn, sh_mx, &sh_code);
                __syncthreads();
            }
            if constexpr (XR) {
                for (int i = tid; i < nch * ST_W2; i += TPB) {
                    const int w = i / nch, ch = i - w * nch;
                    double s2 = sh_q[(size_t)w * MAXCH + ch];
                    (void)xr_merge_sum(xr, (ST_XW_COUNT + 1 + w) * g.C + b.c0 + ch, true, s2);      // this rank's one push per word
                    sh_q[(size_t)w * MAXCH + ch] = s2;
                }
                __syncthreads();
            }
            if (tid < nch) {
                const size_t C = (size_t)g.C;
                const int c = b.c0 + tid;
                sa.stats[(size_t)CNNQ_STAT_B * C + c] = (float)(sh_q[tid] / count_of(tid));
                sa.stats[(size_t)CNNQ_STAT_KURT * C + c] = sa.need_kurt ? (float)(sh_q[MAXCH + tid] / count_of(tid) - 3.) : 0.f;
            }
        }
    }
    // ---- leave the group.  With phase 2 the last ARRIVER above is also the last to need the block: it re-arms it; without it the
    //      departures are counted and the last member out does
    __syncthreads();
    if (!sa.need_dev) {
        if (tid == 0) sh_last = grp_depart_last(grp_lines(ws.cnt, rb.group, Gs, 0, 1), rb.member, Gs) ? 1 : 0;
        __syncthreads();
    }
    if (sh_last)
        for (int m = tid; m < Gs * NPK; m += TPB) __hip_atomic_store(lines + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- round 6: the chain's partial records made GLOBAL through the windows, no collective -----------------------------------------
// k_xr_moments behind k_moments: merges the G pass-A records of this shard per channel, exchanges the six words of the sums
// layout (cnnq_xrank.hip.h) with the other ranks - every rank pushes its own and adds the W ranks' words in rank order - and
// writes the table rows and the merged moment record of the GLOBAL batch (k_combine's formulas).  The first kernel of its launch
// number: its workgroups share the prologue (xr_prologue_all).
constexpr int XS_CPB = 2;      // channels per workgroup
__global__ void __launch_bounds__(TPB) k_xr_moments(const double* __restrict__ part, const int G, const int C, const int has_relu,
                                                    const XRank xr, double* __restrict__ mom, float* __restrict__ stats) {
    xr_prologue_all(xr);
    __shared__ unsigned long long sh_w[XS_CPB][8];
    // 2 channels x 8 words x 16 lanes: lane j merges records j, j + 16, ... (at most four of them), the sixteen lanes fold by a
    // fixed xor tree
    const int tid = threadIdx.x, j = tid & 15, w = (tid >> 4) & 7, cl = tid >> 7;
    const int c = (int)blockIdx.x * XS_CPB + cl;
    const bool live = c < C && (w <= 2 || w == ST_XW_COUNT || ((w == 3 || w == 4) && has_relu));
    double acc = 0., mn = INFINITY, mx = -INFINITY;
    if (live) {
        const int row = w == 1 ? CNNQ_MOM_SUM : w == 2 ? CNNQ_MOM_SUMSQ : w == 3 ? CNNQ_MOM_SUM_RELU : w == 4 ? CNNQ_MOM_SUMSQ_RELU : CNNQ_MOM_COUNT;
        for (int gi = j; gi < G; gi += 16) {
            const double* p = part + (size_t)gi * CNNQ_NMOM * C + c;
            if (w == 0) {
                mn = pmind(mn, p[(size_t)CNNQ_MOM_MIN * C]);
                mx = pmaxd(mx, p[(size_t)CNNQ_MOM_MAX * C]);
            } else {
                acc += p[(size_t)row * C];
            }
        }
    }
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) {
        acc += shfl_xor_d(acc, m);
        mn = pmind(mn, shfl_xor_d(mn, m));
        mx = pmaxd(mx, shfl_xor_d(mx, m));
    }
    if (j == 0) {
        unsigned long long bits = 0ull;
        if (live) {
            bits = w == 0 ? ((unsigned long long)__float_as_uint((float)mn) | ((unsigned long long)__float_as_uint((float)mx) << 32))
                          : (unsigned long long)__double_as_longlong(acc);
            (void)xr_merge_word(xr, w * C + c, true, w == 0, bits);
        }
        sh_w[cl][w] = bits;            // (skipped words: zero = +0.0)
    }
    __syncthreads();
    if (c < C && w == 0 && j == 0) {
        const unsigned long long pr = sh_w[cl][0];
        auto dw = [&](int i) { return __longlong_as_double((long long)sh_w[cl][i]); };
        const MomSum r{(double)__uint_as_float((unsigned)(pr & 0xffffffffull)), (double)__uint_as_float((unsigned)(pr >> 32)), dw(1), dw(2),
                       dw(ST_XW_COUNT), dw(3), dw(4)};
        if (mom) {
            mom[(size_t)CNNQ_MOM_MIN * C + c] = r.mn;
            mom[(size_t)CNNQ_MOM_MAX * C + c] = r.mx;
            mom[(size_t)CNNQ_MOM_SUM * C + c] = r.s;
            mom[(size_t)CNNQ_MOM_SUMSQ * C + c] = r.ss;
            mom[(size_t)CNNQ_MOM_COUNT * C + c] = r.cnt;
            mom[(size_t)CNNQ_MOM_SUM_RELU * C + c] = r.rs;
            mom[(size_t)CNNQ_MOM_SUMSQ_RELU * C + c] = r.rss;
        }
        stats[(size_t)CNNQ_STAT_MIN * C + c] = (float)r.mn;
        stats[(size_t)CNNQ_STAT_MAX * C + c] = (float)r.mx;
        stats[(size_t)CNNQ_STAT_MEAN * C + c] = mean_of(r);
        stats[(size_t)CNNQ_STAT_STD * C + c] = std_of(r);
        float std_pos = 0.f;
        if (has_relu) {
            double rv = (r.rss - r.rs * (r.rs / r.cnt)) / (r.cnt - 1.);
            if (rv < 0.) rv = 0.;
            std_pos = (float)sqrt(rv);
        }
        stats[(size_t)CNNQ_STAT_STD_POS * C + c] = std_pos;      // the table is written completely
        stats[(size_t)CNNQ_STAT_B * C + c] = 0.f;
        stats[(size_t)CNNQ_STAT_KURT * C + c] = 0.f;
    }
}

// k_xr_devsums behind k_absdev: the same for pass B's records - nw = 1: word 6 (sum |x - mean|: what k_fused_* exchange), nw = 2:
// words 6 and 7 (k_stats_flat's second meeting) - and rows B / KURT of the table from the global sums and the global count.  Never
// the first kernel of its launch number.  32 channels x 2 words x 4 lanes per workgroup.
__global__ void __launch_bounds__(TPB) k_xr_devsums(const double* __restrict__ part2, const int G, const int C, const int nw, const int want_kurt,
                                                    const double* __restrict__ count, const XRank xr, float* __restrict__ stats) {
    const int tid = threadIdx.x, j = tid & 3, w = (tid >> 2) & 1, cl = tid >> 3;
    const int c = (int)blockIdx.x * 32 + cl;
    const bool live = c < C && w < nw;
    double acc = 0.;
    if (live)
        for (int gi = j; gi < G; gi += 4) acc += part2[((size_t)gi * CNNQ_NDEV + (w ? CNNQ_DEV_Z4 : CNNQ_DEV_ABS)) * C + c];
    acc += shfl_xor_d(acc, 1);
    acc += shfl_xor_d(acc, 2);
    if (live && j == 0) {
        (void)xr_merge_sum(xr, (ST_XW_COUNT + 1 + w) * C + c, true, acc);
        const double cnt = count[c];
        if (w == 0) stats[(size_t)CNNQ_STAT_B * C + c] = (float)(acc / cnt);
        else stats[(size_t)CNNQ_STAT_KURT * C + c] = want_kurt ? (float)(acc / cnt - 3.) : 0.f;
    }
}

}  // namespace
