// cnnq_stats.hip.h - statistics passes: per-channel moments (pass A), mean absolute deviation / kurtosis (pass B) and their combine kernels.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"

namespace {

// Segmented reductions of the epilogues and merges: SEG lanes per channel (a power of two, SEG | 64) fold their
// share of the channel's entries serially (lane j: entries j, j + SEG, ...) and finish with a log2(SEG)-step xor
// tree.  A full wave per channel (SEG = 64) costs 6 x 7 double shuffles per channel whatever the entry count: with
// 20 channels of 49 entries per workgroup (14x14 and 7x7 rows) that tree was a third of the pass (measured: the
// 1024x14x14 layer 0.200 -> 0.155 ms without it); SEG = 8 folds 8 channels per wave in one 3-step tree.
// The segment width is a function of the entry count alone, so every kernel merging the same records agrees on
// the order of the additions.
__host__ __device__ constexpr int seg_of(int entries) { return entries <= 16 ? 1 : entries < 128 ? 8 : 64; }
// Lanes per channel when the G partial records of C channels are merged.  Narrow segments pay when a workgroup of
// pass B owns many channels (small H*W - those layers have C >= 512); with one or two channels per workgroup a full
// wave per channel loads the records in parallel and starts streaming sooner (measured: 8 lanes per channel on
// the 64..256-channel layers cost them 2-10 %).  A function of (G, C) alone, so that every kernel merging the same
// records - k_combine, k_combine_all, the prologue of k_absdev<RAW> - adds them in the same order.
__host__ __device__ constexpr int mseg_of(int G, int C) { return (G <= 64 && C >= 512) ? (G <= 16 ? 1 : 8) : 64; }
template <int V>
struct IntC { static constexpr int value = V; };
#define CNNQ_SEG_DISPATCH(seg, f)              \
    do {                                       \
        const int seg_ = (seg);                \
        if (seg_ == 64) f(IntC<64>{});         \
        else if (seg_ == 8) f(IntC<8>{});      \
        else f(IntC<1>{});                     \
    } while (0)

// ------------------------------------------------------------------------------------------
// Pass A: per-channel min / max / sum / sumsq / count (+ relu sums)
// ------------------------------------------------------------------------------------------
struct Mom {
    float mn, mx;
    double s, ss, rs, rss;
    __device__ __forceinline__ void init() {
        mn = INFINITY; mx = -INFINITY; s = 0.; ss = 0.; rs = 0.; rss = 0.;
    }
    // NaN: torch's min / max propagate it; the sums do so by themselves.  add4 (the hot form) keeps v_min / v_max
    // (they return the other operand) and k_moments poisons mn / mx afterwards when the sum of squares came out
    // NaN - that happens iff an element was NaN (inf * inf = inf, no cancellation); the merges propagate.
    template <bool RELU>
    __device__ __forceinline__ void add(float v) {
        mn = pmin(mn, v);
        mx = pmax(mx, v);
        const double d = (double)v;
        s += d;
        ss = fma(d, d, ss);
        if constexpr (RELU) {
            const double r = (double)fmaxf(v, 0.f);
            rs += r;
            rss = fma(r, r, rss);
        }
    }
    // four values of ONE channel (a float4 that does not straddle): the 4-sums are formed in fp32
    // (each rounding is unbiased and relative to a 4-term sum, far below the fp32 result precision
    // once thousands of them are accumulated in fp64) - 4 instead of 12 fp64-rate ops per float4
    template <bool RELU>
    __device__ __forceinline__ void add4(const float (&v)[4]) {
        mn = fminf(fminf(mn, fminf(v[0], v[1])), fminf(v[2], v[3]));
        mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        s += (double)((v[0] + v[1]) + (v[2] + v[3]));
        ss += (double)((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
        if constexpr (RELU) {
            const float r0 = fmaxf(v[0], 0.f), r1 = fmaxf(v[1], 0.f), r2 = fmaxf(v[2], 0.f), r3 = fmaxf(v[3], 0.f);
            rs += (double)((r0 + r1) + (r2 + r3));
            rss += (double)((r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3));
        }
    }
    // EIGHT values of one channel (two float4 of a register tile; the single-read kernels of cnnq_stats1.hip.h): the 8-sums in
    // fp32 TWO ELEMENTS PER INSTRUCTION (v_pk_add_f32 / v_pk_mul_f32: pairs (0,1) + (2,3) of a, then of b, the two halves added
    // last), min / max as v_min3 / v_max3 on the raw registers (fminf makes the compiler canonicalise every loaded value first:
    // one more instruction per element) - 6 instead of 9.5 instructions per element, which matters where a kernel has to hide
    // its arithmetic behind 4 bytes per element.  add4p: a alone - the bits add8 gives with b absent.  (Only for fully unrolled
    // tile loops: HIP treats inline asm as convergent, and a loop with a run-time trip count that contains it - the row loop of
    // k_moments - is no longer unrolled.  add4 therefore keeps fminf / fmaxf.)
    template <bool RELU>
    __device__ __forceinline__ void add8(const float (&a)[4], const float (&b)[4]) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        mn = min3_raw(min3_raw(mn, a[0], a[1]), a[2], a[3]);
        mn = min3_raw(min3_raw(mn, b[0], b[1]), b[2], b[3]);
        mx = max3_raw(max3_raw(mx, a[0], a[1]), a[2], a[3]);
        mx = max3_raw(max3_raw(mx, b[0], b[1]), b[2], b[3]);
        const f2 a0 = {a[0], a[1]}, a1 = {a[2], a[3]}, b0 = {b[0], b[1]}, b1 = {b[2], b[3]};
        const f2 ps = ((a0 + a1) + b0) + b1;
        s += (double)hadd(ps);
        const f2 pq = ((a0 * a0 + a1 * a1) + b0 * b0) + b1 * b1;
        ss += (double)hadd(pq);
        if constexpr (RELU) {
            const f2 r0 = {relu_raw(a[0]), relu_raw(a[1])}, r1 = {relu_raw(a[2]), relu_raw(a[3])};
            const f2 r2 = {relu_raw(b[0]), relu_raw(b[1])}, r3 = {relu_raw(b[2]), relu_raw(b[3])};
            const f2 pr = ((r0 + r1) + r2) + r3;
            rs += (double)hadd(pr);
            const f2 pp = ((r0 * r0 + r1 * r1) + r2 * r2) + r3 * r3;
            rss += (double)hadd(pp);
        }
    }
    template <bool RELU>
    __device__ __forceinline__ void add4p(const float (&a)[4]) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        mn = min3_raw(min3_raw(mn, a[0], a[1]), a[2], a[3]);
        mx = max3_raw(max3_raw(mx, a[0], a[1]), a[2], a[3]);
        const f2 a0 = {a[0], a[1]}, a1 = {a[2], a[3]};
        const f2 ps = a0 + a1;
        s += (double)hadd(ps);
        const f2 pq = a0 * a0 + a1 * a1;
        ss += (double)hadd(pq);
        if constexpr (RELU) {
            const f2 r0 = {relu_raw(a[0]), relu_raw(a[1])}, r1 = {relu_raw(a[2]), relu_raw(a[3])};
            const f2 pr = r0 + r1;
            rs += (double)hadd(pr);
            const f2 pp = r0 * r0 + r1 * r1;
            rss += (double)hadd(pp);
        }
    }
    template <bool RELU>
    __device__ __forceinline__ void merge(const Mom& o) {
        mn = pmin(mn, o.mn);
        mx = pmax(mx, o.mx);
        s += o.s;
        ss += o.ss;
        if constexpr (RELU) { rs += o.rs; rss += o.rss; }
    }
    template <bool RELU, int SEG>
    __device__ __forceinline__ void seg_reduce() {
#pragma unroll
        for (int m = SEG >> 1; m >= 1; m >>= 1) {   // constant masks: the compiler may use DPP instead of ds_bpermute
            Mom o;
            o.mn = shfl_xor_f(mn, m);
            o.mx = shfl_xor_f(mx, m);
            o.s = shfl_xor_d(s, m);
            o.ss = shfl_xor_d(ss, m);
            if constexpr (RELU) { o.rs = shfl_xor_d(rs, m); o.rss = shfl_xor_d(rss, m); }
            merge<RELU>(o);
        }
    }
    template <bool RELU>
    __device__ __forceinline__ void wave_reduce() {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            Mom o;
            o.mn = shfl_xor_f(mn, m);
            o.mx = shfl_xor_f(mx, m);
            o.s = shfl_xor_d(s, m);
            o.ss = shfl_xor_d(ss, m);
            if constexpr (RELU) { o.rs = shfl_xor_d(rs, m); o.rss = shfl_xor_d(rss, m); }
            merge<RELU>(o);
        }
    }
};

template <bool RELU>
__device__ __forceinline__ void write_mom(double* __restrict__ part, int grp, int C, int ch, const Mom& m,
                                          double count) {
    double* p = part + (size_t)grp * CNNQ_NMOM * C + ch;
    p[(size_t)CNNQ_MOM_MIN * C] = (double)m.mn;
    p[(size_t)CNNQ_MOM_MAX * C] = (double)m.mx;
    p[(size_t)CNNQ_MOM_SUM * C] = m.s;
    p[(size_t)CNNQ_MOM_SUMSQ * C] = m.ss;
    p[(size_t)CNNQ_MOM_COUNT * C] = count;
    p[(size_t)CNNQ_MOM_SUM_RELU * C] = RELU ? m.rs : 0.;
    p[(size_t)CNNQ_MOM_SUMSQ_RELU * C] = RELU ? m.rss : 0.;
}

template <int VEC, int A, int J, bool RELU, bool NTL>
__global__ void __launch_bounds__(TPB) k_moments(const float* __restrict__ x, const Geo g,
                                                 double* __restrict__ part) {
    constexpr int NE = TPB * J * A;  // LDS entries (one per column, or per element when straddling)
    __shared__ float l_mn[NE], l_mx[NE];
    __shared__ double l_s[NE], l_ss[NE];
    __shared__ double l_rs[RELU ? NE : 1], l_rss[RELU ? NE : 1];

    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    int col[J];
    bool ok[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;  // idle slots re-read the block's first column; results discarded
    }
    Mom acc[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int a = 0; a < A; ++a) acc[j][a].init();

    const float* row = x + (size_t)b.n0 * (size_t)g.P;
    constexpr int NU = (8 / J) < 2 ? 2 : 8 / J;   // 8 16-byte loads in flight per lane
#pragma unroll NU
    for (int n = b.n0; n < b.n1; ++n, row += g.P) {
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv_sel<VEC, NTL>(row + (size_t)col[j] * VEC, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if constexpr (VEC == 4 && A == 1) {
                acc[j][0].template add4<RELU>(v[j]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[j][A == 1 ? 0 : e].template add<RELU>(v[j][e]);
            }
        }
    }

#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int a = 0; a < A; ++a)
            if (acc[j][a].ss != acc[j][a].ss) { acc[j][a].mn = NAN; acc[j][a].mx = NAN; }
    const double rows = (double)(b.n1 - b.n0);
    if (g.mode == 1) {
        // one channel per workgroup: registers -> wave shuffle -> 4 LDS entries
        Mom t;
        t.init();
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (ok[j]) t.template merge<RELU>(acc[j][0]);
        t.template wave_reduce<RELU>();
        const int wv = tid >> 6;
        if ((tid & 63) == 0) {
            l_mn[wv] = t.mn; l_mx[wv] = t.mx; l_s[wv] = t.s; l_ss[wv] = t.ss;
            if constexpr (RELU) { l_rs[wv] = t.rs; l_rss[wv] = t.rss; }
        }
        __syncthreads();
        if (tid == 0) {
            Mom r;
            r.init();
            for (int i = 0; i < TPB / 64; ++i) {
                Mom o;
                o.mn = l_mn[i]; o.mx = l_mx[i]; o.s = l_s[i]; o.ss = l_ss[i];
                if constexpr (RELU) { o.rs = l_rs[i]; o.rss = l_rss[i]; }
                r.template merge<RELU>(o);
            }
            write_mom<RELU>(part, b.grp, g.C, b.c0, r, (double)(b.col1 - b.col0) * VEC * rows);
        }
        return;
    }
    // k whole channels per workgroup: per-column results to LDS, then one wave per channel
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (ok[j]) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const int e = (j * TPB + tid) * A + a;
                l_mn[e] = acc[j][a].mn; l_mx[e] = acc[j][a].mx; l_s[e] = acc[j][a].s; l_ss[e] = acc[j][a].ss;
                if constexpr (RELU) { l_rs[e] = acc[j][a].rs; l_rss[e] = acc[j][a].rss; }
            }
        }
    }
    __syncthreads();
    const int epc = g.HW * A / VEC;  // LDS entries per channel
    const double count = (double)g.HW * rows;
    const int nch = b.c1 - b.c0;
    auto fold = [&](auto SEGC) {
        constexpr int SEG = decltype(SEGC)::value;
        for (int base = 0; base < nch; base += TPB / SEG) {   // uniform: whole waves take part in the shuffles
            const int i = base + tid / SEG, j = tid & (SEG - 1);
            Mom r;
            r.init();
            if (i < nch)
                for (int e = i * epc + j; e < (i + 1) * epc; e += SEG) {
                    Mom o;
                    o.mn = l_mn[e]; o.mx = l_mx[e]; o.s = l_s[e]; o.ss = l_ss[e];
                    if constexpr (RELU) { o.rs = l_rs[e]; o.rss = l_rss[e]; }
                    r.template merge<RELU>(o);
                }
            r.template seg_reduce<RELU, SEG>();
            if (i < nch && j == 0) write_mom<RELU>(part, b.grp, g.C, b.c0 + i, r, count);
        }
    };
    CNNQ_SEG_DISPATCH(seg_of(epc), fold);
}

// SEG = mseg_of(G, C) lanes merge the G pass-A records of channel c (lane j: records j, j + SEG, ..., then the xor tree;
// inactive lanes only take part in the shuffles): the single definition of the merge order, shared by k_combine,
// k_combine_all and the prologue of k_absdev<RAW>, so the mean a fused pass B subtracts is bit for bit the mean the
// statistics table reports.
struct MomSum {
    double mn, mx, s, ss, cnt, rs, rss;
};
template <int SEG>
__device__ __forceinline__ MomSum merge_moments_seg(const double* __restrict__ part, int G, int C, int c, bool has_relu,
                                                    bool active) {
    const int j = threadIdx.x & (SEG - 1);
    MomSum r{INFINITY, -INFINITY, 0., 0., 0., 0., 0.};
    if (active)
        for (int gi = j; gi < G; gi += SEG) {
            const double* p = part + (size_t)gi * CNNQ_NMOM * C + c;
            r.mn = pmind(r.mn, p[(size_t)CNNQ_MOM_MIN * C]);
            r.mx = pmaxd(r.mx, p[(size_t)CNNQ_MOM_MAX * C]);
            r.s += p[(size_t)CNNQ_MOM_SUM * C];
            r.ss += p[(size_t)CNNQ_MOM_SUMSQ * C];
            r.cnt += p[(size_t)CNNQ_MOM_COUNT * C];
            if (has_relu) {
                r.rs += p[(size_t)CNNQ_MOM_SUM_RELU * C];
                r.rss += p[(size_t)CNNQ_MOM_SUMSQ_RELU * C];
            }
        }
#pragma unroll
    for (int m = SEG >> 1; m >= 1; m >>= 1) {
        r.mn = pmind(r.mn, shfl_xor_d(r.mn, m));
        r.mx = pmaxd(r.mx, shfl_xor_d(r.mx, m));
        r.s += shfl_xor_d(r.s, m);
        r.ss += shfl_xor_d(r.ss, m);
        r.cnt += shfl_xor_d(r.cnt, m);
        r.rs += shfl_xor_d(r.rs, m);
        r.rss += shfl_xor_d(r.rss, m);
    }
    return r;
}
__device__ __forceinline__ MomSum merge_moments(const double* __restrict__ part, int G, int C, int c, bool has_relu,
                                                bool active) {
    const int seg = mseg_of(G, C);
    if (seg == 64) return merge_moments_seg<64>(part, G, C, c, has_relu, active);
    if (seg == 8) return merge_moments_seg<8>(part, G, C, c, has_relu, active);
    return merge_moments_seg<1>(part, G, C, c, has_relu, active);
}
// The two sums pass B needs (mean; the standard deviation for kurtosis), merged in exactly the order of
// merge_moments - the min / max / count / relu rows are not even loaded (the count is N * H*W by construction).
template <int SEG, bool NEED_SS>
__device__ __forceinline__ void merge_sums_seg(const double* __restrict__ part, int G, int C, int c, bool active, double& s,
                                               double& ss) {
    const int j = threadIdx.x & (SEG - 1);
    s = 0.;
    ss = 0.;
    if (active)
        for (int gi = j; gi < G; gi += SEG) {
            const double* p = part + (size_t)gi * CNNQ_NMOM * C + c;
            s += p[(size_t)CNNQ_MOM_SUM * C];
            if constexpr (NEED_SS) ss += p[(size_t)CNNQ_MOM_SUMSQ * C];
        }
#pragma unroll
    for (int m = SEG >> 1; m >= 1; m >>= 1) {
        s += shfl_xor_d(s, m);
        if constexpr (NEED_SS) ss += shfl_xor_d(ss, m);
    }
}
template <bool NEED_SS>
__device__ __forceinline__ void merge_sums(const double* __restrict__ part, int G, int C, int c, bool active, double& s,
                                           double& ss) {
    const int seg = mseg_of(G, C);
    if (seg == 64) merge_sums_seg<64, NEED_SS>(part, G, C, c, active, s, ss);
    else if (seg == 8) merge_sums_seg<8, NEED_SS>(part, G, C, c, active, s, ss);
    else merge_sums_seg<1, NEED_SS>(part, G, C, c, active, s, ss);
}
// pass-B sums of channel c over the G records, same lane layout
template <int SEG>
__device__ __forceinline__ void merge_dev_seg(const double* __restrict__ part2, int G, int C, int c, bool active,
                                              double& sa, double& sk) {
    const int j = threadIdx.x & (SEG - 1);
    sa = 0.;
    sk = 0.;
    if (active)
        for (int gi = j; gi < G; gi += SEG) {
            const double* p = part2 + (size_t)gi * CNNQ_NDEV * C + c;
            sa += p[(size_t)CNNQ_DEV_ABS * C];
            sk += p[(size_t)CNNQ_DEV_Z4 * C];
        }
#pragma unroll
    for (int m = SEG >> 1; m >= 1; m >>= 1) { sa += shfl_xor_d(sa, m); sk += shfl_xor_d(sk, m); }
}
__device__ __forceinline__ void merge_dev(const double* __restrict__ part2, int G, int C, int c, bool active, double& sa,
                                          double& sk) {
    const int seg = mseg_of(G, C);
    if (seg == 64) merge_dev_seg<64>(part2, G, C, c, active, sa, sk);
    else if (seg == 8) merge_dev_seg<8>(part2, G, C, c, active, sa, sk);
    else merge_dev_seg<1>(part2, G, C, c, active, sa, sk);
}
// channels a 256-thread workgroup of the merge kernels handles, and their grid
__host__ __device__ constexpr int merge_cpw(int G, int C) { return TPB / mseg_of(G, C); }
__device__ __forceinline__ float mean_of(const MomSum& r) { return (float)(r.s / r.cnt); }
__device__ __forceinline__ float std_of(const MomSum& r) {
    const double mean = r.s / r.cnt;
    double var = (r.ss - r.s * mean) / (r.cnt - 1.);
    if (var < 0.) var = 0.;
    return (float)sqrt(var);
}

// merge G records per channel; mseg_of(G, C) lanes per channel (grid: ceil(C / merge_cpw(G, C)))
__global__ void __launch_bounds__(TPB) k_combine(const double* __restrict__ part, int G, int C, int has_relu,
                                                 double* __restrict__ mom, float* __restrict__ stats) {
    const int seg = mseg_of(G, C);
    const int c = blockIdx.x * (TPB / seg) + threadIdx.x / seg;
    const MomSum r = merge_moments(part, G, C, c, has_relu != 0, c < C);
    if (c >= C || (threadIdx.x & (seg - 1)) != 0) return;
    if (mom) {
        mom[(size_t)CNNQ_MOM_MIN * C + c] = r.mn;
        mom[(size_t)CNNQ_MOM_MAX * C + c] = r.mx;
        mom[(size_t)CNNQ_MOM_SUM * C + c] = r.s;
        mom[(size_t)CNNQ_MOM_SUMSQ * C + c] = r.ss;
        mom[(size_t)CNNQ_MOM_COUNT * C + c] = r.cnt;
        mom[(size_t)CNNQ_MOM_SUM_RELU * C + c] = r.rs;
        mom[(size_t)CNNQ_MOM_SUMSQ_RELU * C + c] = r.rss;
    }
    if (stats) {
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
        // the table is written completely: rows nobody computed are zero (B / KURT are filled in later by k_combine_dev)
        stats[(size_t)CNNQ_STAT_STD_POS * C + c] = std_pos;
        stats[(size_t)CNNQ_STAT_B * C + c] = 0.f;
        stats[(size_t)CNNQ_STAT_KURT * C + c] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// Pass B: sum |x - mean| and sum ((x - mean)/std)^4 per channel
// ------------------------------------------------------------------------------------------
// RAW: `stats` is NULL and `part` holds the G UNMERGED pass-A records: every workgroup merges the records of its own
// channels in its prologue (one wave per channel, the k_combine arithmetic) - one launch less per tensor.
template <int VEC, int A, int J, bool KURT, bool NTL, bool RAW = false>
__global__ void __launch_bounds__(TPB) k_absdev(const float* __restrict__ x, const Geo g,
                                                const float* __restrict__ stats, double* __restrict__ part2,
                                                const double* __restrict__ part = nullptr, int G = 0) {
    constexpr int NE = TPB * J * A;
    __shared__ double l_a[NE];
    __shared__ double l_k[KURT ? NE : 1];
    __shared__ float sh_mean[MAXCH], sh_std[MAXCH];

    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    if constexpr (RAW) {
        const int mseg = mseg_of(G, g.C);
        for (int base = 0; base < b.c1 - b.c0; base += TPB / mseg) {
            const int i = base + tid / mseg;
            const bool act = i < b.c1 - b.c0;
            MomSum r{0., 0., 0., 0., (double)g.N * (double)g.HW, 0., 0.};
            merge_sums<KURT>(part, G, g.C, b.c0 + (act ? i : 0), act, r.s, r.ss);
            if (act && (tid & (mseg - 1)) == 0) { sh_mean[i] = mean_of(r); sh_std[i] = KURT ? std_of(r) : 1.f; }
        }
    } else {
        for (int i = tid; i < b.c1 - b.c0; i += TPB) {
            sh_mean[i] = stats[(size_t)CNNQ_STAT_MEAN * g.C + b.c0 + i];
            sh_std[i] = stats[(size_t)CNNQ_STAT_STD * g.C + b.c0 + i];
        }
    }
    __syncthreads();
    int col[J];
    bool ok[J];
    float mean[J][A], sd[J][A];
    double sa[J][A], sk[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const unsigned e = (unsigned)col[j] * VEC + (A == 1 ? 0 : a);
            const int ch = (int)(e / (unsigned)g.HW) - b.c0;
            mean[j][a] = sh_mean[ch];
            sd[j][a] = KURT ? 1.f / sh_std[ch] : 0.f;   // reciprocal of the standard deviation
            sa[j][a] = 0.;
            sk[j][a] = 0.;
        }
    }
    const int nrows = b.n1 - b.n0;
    constexpr int NU = (8 / J) < 2 ? 2 : 8 / J;
#pragma unroll NU
    for (int r = 0; r < nrows; ++r) {
        // pass B follows pass A over the same tensor: walking it backwards (g.rev) re-reads what
        // pass A touched last from the Infinity Cache
        const float* row = x + (size_t)(g.rev ? b.n1 - 1 - r : b.n0 + r) * (size_t)g.P;
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv_sel<VEC, NTL>(row + (size_t)col[j] * VEC, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                const float d = v[j][e] - mean[j][a];
                sa[j][a] += (double)fabsf(d);
                if constexpr (KURT) {
                    // (x - mean) * (1/std): one rounding more than the reference's division (<= 1 ulp in
                    // z, 2.4e-7 relative in z^4) - far inside the sensitivity of kurtosis to the last bit
                    // of the fp32 mean (see tests), and it removes a 10-instruction divide per element
                    const float z = d * sd[j][a];
                    const float z2 = z * z;
                    sk[j][a] += (double)(z2 * z2);
                }
            }
    }
    auto emit = [&](int ch, double ta, double tk) {
        double* p = part2 + (size_t)b.grp * CNNQ_NDEV * g.C + ch;
        p[(size_t)CNNQ_DEV_ABS * g.C] = ta;
        p[(size_t)CNNQ_DEV_Z4 * g.C] = KURT ? tk : 0.;
    };
    const int wv = tid >> 6, lane = tid & 63;
    if (g.mode == 1) {
        double ta = 0., tk = 0.;
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (ok[j]) { ta += sa[j][0]; tk += sk[j][0]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ta += shfl_xor_d(ta, m); tk += shfl_xor_d(tk, m); }
        if (lane == 0) { l_a[wv] = ta; if constexpr (KURT) l_k[wv] = tk; }
        __syncthreads();
        if (tid == 0) {
            double ra = 0., rk = 0.;
            for (int i = 0; i < TPB / 64; ++i) { ra += l_a[i]; if constexpr (KURT) rk += l_k[i]; }
            emit(b.c0, ra, rk);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const int e = (j * TPB + tid) * A + a;
                l_a[e] = sa[j][a];
                if constexpr (KURT) l_k[e] = sk[j][a];
            }
        }
    __syncthreads();
    const int epc = g.HW * A / VEC;
    const int nch = b.c1 - b.c0;
    auto fold = [&](auto SEGC) {
        constexpr int SEG = decltype(SEGC)::value;
        for (int base = 0; base < nch; base += TPB / SEG) {
            const int i = base + tid / SEG, j = tid & (SEG - 1);
            double ra = 0., rk = 0.;
            if (i < nch)
                for (int e = i * epc + j; e < (i + 1) * epc; e += SEG) { ra += l_a[e]; if constexpr (KURT) rk += l_k[e]; }
#pragma unroll
            for (int m = SEG >> 1; m >= 1; m >>= 1) { ra += shfl_xor_d(ra, m); if constexpr (KURT) rk += shfl_xor_d(rk, m); }
            if (i < nch && j == 0) emit(b.c0 + i, ra, rk);
        }
    };
    CNNQ_SEG_DISPATCH(seg_of(epc), fold);
}

// final merge of BOTH passes (the fused form: k_moments -> k_absdev<RAW> -> this): rows MIN, MAX, MEAN, STD (STD_POS)
// from the pass-A records, B (KURT) from the pass-B records, and the merged moment record
__global__ void __launch_bounds__(TPB) k_combine_all(const double* __restrict__ part, const double* __restrict__ part2,
                                                     int G, int C, int has_relu, int want_kurt, double* __restrict__ mom,
                                                     float* __restrict__ stats) {
    const int seg = mseg_of(G, C);
    const int c = blockIdx.x * (TPB / seg) + threadIdx.x / seg;
    const MomSum r = merge_moments(part, G, C, c, has_relu != 0, c < C);
    double sa, sk;
    merge_dev(part2, G, C, c, c < C, sa, sk);
    if (c >= C || (threadIdx.x & (seg - 1)) != 0) return;
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
    stats[(size_t)CNNQ_STAT_STD_POS * C + c] = std_pos;   // every row is written: no memset in front of the chain
    stats[(size_t)CNNQ_STAT_B * C + c] = (float)(sa / r.cnt);
    stats[(size_t)CNNQ_STAT_KURT * C + c] = want_kurt ? (float)(sk / r.cnt - 3.) : 0.f;
}

__global__ void __launch_bounds__(TPB) k_combine_dev(const double* __restrict__ part2, int G, int C,
                                                     const double* __restrict__ mom, int want_kurt,
                                                     double* __restrict__ dev_out, float* __restrict__ stats) {
    const int seg = mseg_of(G, C);
    const int c = blockIdx.x * (TPB / seg) + threadIdx.x / seg;
    double sa, sk;
    merge_dev(part2, G, C, c, c < C, sa, sk);
    if (c >= C || (threadIdx.x & (seg - 1)) != 0) return;
    if (dev_out) {
        dev_out[(size_t)CNNQ_DEV_ABS * C + c] = sa;
        dev_out[(size_t)CNNQ_DEV_Z4 * C + c] = sk;
    }
    if (stats) {
        const double cnt = mom[(size_t)CNNQ_MOM_COUNT * C + c];
        stats[(size_t)CNNQ_STAT_B * C + c] = (float)(sa / cnt);
        if (want_kurt) stats[(size_t)CNNQ_STAT_KURT * C + c] = (float)(sk / cnt - 3.);
    }
}

}  // namespace
