// cnnq_stats_group.hip.h - ALL per-channel statistics of a tensor (smpc.py:45-79: min, max, mean, std, std of the
// positive part, b = mean |x - mean|, kurtosis) in ONE launch and ONE read of x: 4 instead of 8 bytes per element.
// Part of the single translation unit cnnq_kernels.hip.
//
// The two-pass chain (cnnq_stats.hip.h) reads x twice because pass B needs the channel mean of pass A.  Here a
// workgroup keeps its tile (K samples x <= 256 float4 columns, the tiling of k_mmq_group) in REGISTERS across both
// passes and meets the other workgroups that hold pieces of the same channels twice inside the launch, with the
// meeting protocol of cnnq_group.hip.h (write-through records in fine-grained memory, arrival / departure / flag
// counter lines, bounded waits):
//
//   pass A on the registers -> per-channel records {min, max, sum, sumsq, count, relu sums} of the tile
//     -> exchange 1: every member merges the group's records in member order: bit-identical mean / std everywhere
//   pass B on the SAME registers (sum |x - mean|, sum ((x - mean) / std)^4) -> exchange 2 -> member 0 writes the table.
//
// Groups of more than GRP_SUB members (one channel row wider than a workgroup, mode 1) fold their records in two
// levels: the last arriver of each sub-group merges its sub-group's records into one, members then read one record
// per sub-group (13 instead of 208 for the 112x112 layers).
// A member whose wait times out recomputes what it needs from x itself (all samples of its channels, the cold path)
// and raises bit 0 of the status word; it still publishes its tile's records, so nobody waits for it in vain.
// Accumulation is the chain's (fp32 4-sums into fp64); the partition differs, so sums agree with the chain's to
// fp64 rounding, min / max exactly.
#pragma once
#include "cnnq_group.hip.h"
#include "cnnq_stats.hip.h"

namespace {

constexpr int SG_REC = 8;   // doubles per pass-A record (64 bytes): MIN MAX SUM SUMSQ COUNT SUM_RELU SUMSQ_RELU -
constexpr int SG_DEV = 2;   // doubles per pass-B record: sum |d|, sum z^4
#ifndef SG_K32_WAVES
#define SG_K32_WAVES 2      // waves per SIMD the K = 32 tile is compiled for (3 = 168 VGPRs spills the fp64 accumulators)
#endif

// workspace views: the counter region of GWs (two line sets per group), then per group a block of pass-A records
// [Gs * kk members' records][nsub sub-group records] and one of pass-B records with the same shape
struct SGWs {
    unsigned* status;
    unsigned* cnt;
    double* rec_a;
    double* rec_b;
    int slots;   // records per group block (members * kk + nsub)
};

__device__ __forceinline__ void st_rec(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_rec(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ms_init(MomSum& r) { r = MomSum{INFINITY, -INFINITY, 0., 0., 0., 0., 0.}; }
__device__ __forceinline__ void ms_merge(MomSum& r, const MomSum& o) {
    r.mn = pmind(r.mn, o.mn); r.mx = pmaxd(r.mx, o.mx);
    r.s += o.s; r.ss += o.ss; r.cnt += o.cnt; r.rs += o.rs; r.rss += o.rss;
}
__device__ __forceinline__ MomSum ms_load(const double* p) {
    MomSum o;
    o.mn = ld_rec(p + 0); o.mx = ld_rec(p + 1); o.s = ld_rec(p + 2); o.ss = ld_rec(p + 3);
    o.cnt = ld_rec(p + 4); o.rs = ld_rec(p + 5); o.rss = ld_rec(p + 6);
    return o;
}
__device__ __forceinline__ void ms_store(double* p, const MomSum& o) {
    st_rec(p + 0, o.mn); st_rec(p + 1, o.mx); st_rec(p + 2, o.s); st_rec(p + 3, o.ss);
    st_rec(p + 4, o.cnt); st_rec(p + 5, o.rs); st_rec(p + 6, o.rss);
}
__device__ __forceinline__ void ms_wave_reduce(MomSum& r) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        MomSum o;
        o.mn = shfl_xor_d(r.mn, m); o.mx = shfl_xor_d(r.mx, m); o.s = shfl_xor_d(r.s, m); o.ss = shfl_xor_d(r.ss, m);
        o.cnt = shfl_xor_d(r.cnt, m); o.rs = shfl_xor_d(r.rs, m); o.rss = shfl_xor_d(r.rss, m);
        ms_merge(r, o);
    }
}

struct SgLds {
    float mn[TPB], mx[TPB];
    double s[TPB], ss[TPB], rs[TPB], rss[TPB];
};

// per-lane pass-A accumulators (one channel per lane: A = 1) -> per-channel records of the workgroup in rec[c1 - c0];
// `cnt` = elements behind every channel's record
template <bool RELU>
__device__ __forceinline__ void wg_channel_moments(const Geo& g, const Blk& b, bool ok, Mom acc, double cnt, SgLds& l,
                                                   MomSum* rec) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (acc.ss != acc.ss) { acc.mn = NAN; acc.mx = NAN; }   // an element was NaN: torch.min / max propagate it
    if (!ok) acc.init();
    if (g.mode == 1) {
        acc.template wave_reduce<RELU>();
        if (lane == 0) {
            l.mn[wv] = acc.mn; l.mx[wv] = acc.mx; l.s[wv] = acc.s; l.ss[wv] = acc.ss;
            if constexpr (RELU) { l.rs[wv] = acc.rs; l.rss[wv] = acc.rss; }
        }
        __syncthreads();
        if (tid == 0) {
            MomSum r;
            ms_init(r);
            for (int i = 0; i < TPB / 64; ++i)
                ms_merge(r, MomSum{(double)l.mn[i], (double)l.mx[i], l.s[i], l.ss[i], 0., RELU ? l.rs[i] : 0., RELU ? l.rss[i] : 0.});
            r.cnt = cnt;
            rec[0] = r;
        }
        __syncthreads();
        return;
    }
    l.mn[tid] = acc.mn; l.mx[tid] = acc.mx; l.s[tid] = acc.s; l.ss[tid] = acc.ss;
    if constexpr (RELU) { l.rs[tid] = acc.rs; l.rss[tid] = acc.rss; }
    __syncthreads();
    const int epc = g.HW / 4;   // LDS entries (float4 columns) per channel
    auto entry = [&](int e) {
        return MomSum{(double)l.mn[e], (double)l.mx[e], l.s[e], l.ss[e], 0., RELU ? l.rs[e] : 0., RELU ? l.rss[e] : 0.};
    };
    if (epc <= 16) {
        for (int ch = tid; ch < b.c1 - b.c0; ch += TPB) {
            MomSum r;
            ms_init(r);
            for (int e = ch * epc; e < (ch + 1) * epc; ++e) ms_merge(r, entry(e));
            r.cnt = cnt;
            rec[ch] = r;
        }
    } else {
        for (int ch = wv; ch < b.c1 - b.c0; ch += TPB / 64) {
            MomSum r;
            ms_init(r);
            for (int e = ch * epc + lane; e < (ch + 1) * epc; e += 64) ms_merge(r, entry(e));
            ms_wave_reduce(r);
            r.cnt = cnt;
            if (lane == 0) rec[ch] = r;
        }
    }
    __syncthreads();
}

// per-lane pass-B sums -> per-channel sums of the workgroup in sa[c1 - c0], sk[c1 - c0]
__device__ __forceinline__ void wg_channel_dev(const Geo& g, const Blk& b, bool ok, double ta, double tk, SgLds& l,
                                               double* sa, double* sk) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (!ok) { ta = 0.; tk = 0.; }
    if (g.mode == 1) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ta += shfl_xor_d(ta, m); tk += shfl_xor_d(tk, m); }
        if (lane == 0) { l.s[wv] = ta; l.ss[wv] = tk; }
        __syncthreads();
        if (tid == 0) {
            double ra = 0., rk = 0.;
            for (int i = 0; i < TPB / 64; ++i) { ra += l.s[i]; rk += l.ss[i]; }
            sa[0] = ra;
            sk[0] = rk;
        }
        __syncthreads();
        return;
    }
    l.s[tid] = ta;
    l.ss[tid] = tk;
    __syncthreads();
    const int epc = g.HW / 4;
    if (epc <= 16) {
        for (int ch = tid; ch < b.c1 - b.c0; ch += TPB) {
            double ra = 0., rk = 0.;
            for (int e = ch * epc; e < (ch + 1) * epc; ++e) { ra += l.s[e]; rk += l.ss[e]; }
            sa[ch] = ra;
            sk[ch] = rk;
        }
    } else {
        for (int ch = wv; ch < b.c1 - b.c0; ch += TPB / 64) {
            double ra = 0., rk = 0.;
            for (int e = ch * epc + lane; e < (ch + 1) * epc; e += 64) { ra += l.s[e]; rk += l.ss[e]; }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { ra += shfl_xor_d(ra, m); rk += shfl_xor_d(rk, m); }
            if (lane == 0) { sa[ch] = ra; sk[ch] = rk; }
        }
    }
    __syncthreads();
}

// |d| and z^4 of one float4 (the arithmetic of k_absdev)
template <bool KURT>
__device__ __forceinline__ void dev_acc4(const float (&v)[4], float mean, float inv_sd, double& sa, double& sk) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d = v[e] - mean;
        sa += (double)fabsf(d);
        if constexpr (KURT) {
            const float z = d * inv_sd;
            const float z2 = z * z;
            sk += (double)(z2 * z2);
        }
    }
}

// cold path: the workgroup's columns (mode 2) or its whole channel row (mode 1), ALL samples, straight from x
template <typename F>
__device__ __forceinline__ void for_all_samples(const float* __restrict__ x, const Geo& g, const Blk& b, bool& ok, F f) {
    const int tid = threadIdx.x;
    if (g.mode == 1) {
        const int cpc = g.HW / 4;
        ok = true;
        for (int n = 0; n < g.N; ++n)
            for (int col = tid; col < cpc; col += TPB) {
                float v[4];
                ldv<4>(x + (size_t)n * (size_t)g.P + ((size_t)b.c0 * cpc + col) * 4, v);
                f(v);
            }
    } else {
        const int col = b.col0 + tid;
        ok = col < b.col1;
        if (ok)
            for (int n = 0; n < g.N; ++n) {
                float v[4];
                ldv<4>(x + (size_t)n * (size_t)g.P + (size_t)col * 4, v);
                f(v);
            }
    }
}

template <int K, bool RELU, bool KURT>
__global__ void __launch_bounds__(TPB, (K == 32 ? SG_K32_WAVES : 1)) k_stats_group(
    const float* __restrict__ x, const Geo g, const int Gs, const SGWs ws, const int need_dev, double* __restrict__ mom,
    float* __restrict__ stats, const unsigned flags) {
    __shared__ SgLds l;
    __shared__ MomSum sh_rec[MAXCH];
    __shared__ float sh_mean[MAXCH], sh_std[MAXCH];
    __shared__ double sh_a[MAXCH], sh_k[MAXCH];
    __shared__ int sh_timed_out;
    const RBlk rb = rblk_of(g, Gs);
    const Blk& b = rb.b;
    const int tid = threadIdx.x;
    const int col = b.col0 + tid;
    const bool ok = col < b.col1;
    const int colc = ok ? col : b.col0;
    const int nrows = b.n1 - b.n0;
    const size_t base = (size_t)b.n0 * (size_t)g.P + (size_t)colc * 4;
    const int nch = b.c1 - b.c0;
    const int kk = (g.mode == 1) ? 1 : g.k;
    const int nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    const bool two_level = (g.mode == 1) && nsub > 1;   // sub-group records (mode 2 members read every record)
    const int mych = (g.mode == 1) ? 0 : (int)(((unsigned)colc * 4u) / (unsigned)g.HW) - b.c0;

    // ---- the tile: K 16-byte loads per lane, issued back to back (rows past the tile re-read its last row)
    float v[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int r = j < nrows ? j : nrows - 1;
        ldv_nt<4>(x + base + (size_t)r * (size_t)g.P, v[j]);
    }
    // ---- pass A on the registers
    Mom acc;
    acc.init();
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < nrows) acc.template add4<RELU>(v[j]);
    const double cnt = (g.mode == 1) ? (double)(b.col1 - b.col0) * 4. * (double)nrows : (double)g.HW * (double)nrows;
    wg_channel_moments<RELU>(g, b, ok, acc, cnt, l, sh_rec);

    // ---- exchange 1: publish the tile's records, meet, merge the group's records
    double* blk_a = ws.rec_a + (size_t)rb.group * ws.slots * SG_REC;
    for (int ch = tid; ch < nch; ch += TPB) ms_store(blk_a + ((size_t)rb.member * kk + ch) * SG_REC, sh_rec[ch]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* lines1 = grp_lines(ws.cnt, rb.group, Gs, 0, 2);
    if (tid == 0) {
        const int timed_out = grp_meet(lines1, rb.member, Gs, flags, [&] {
            if (!two_level) return;
            asm volatile("" ::: "memory");
            const int si = rb.member / GRP_SUB;
            MomSum r;
            ms_init(r);
            for (int m = si * GRP_SUB; m < min(Gs, (si + 1) * GRP_SUB); ++m) ms_merge(r, ms_load(blk_a + (size_t)m * SG_REC));
            ms_store(blk_a + (size_t)(Gs + si) * SG_REC, r);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        });
        if (timed_out) atomicOr(ws.status, 1u);
        sh_timed_out = timed_out;
    }
    __syncthreads();
    bool cold = sh_timed_out != 0;
    if (cold) {
        Mom t;
        t.init();
        bool cok;
        for_all_samples(x, g, b, cok, [&](const float(&q)[4]) { t.template add4<RELU>(q); });
        wg_channel_moments<RELU>(g, b, cok, t, (double)g.HW * (double)g.N, l, sh_rec);
    } else if (g.mode == 1) {
        // one channel: lanes of wave 0 take one record each (members, or sub-groups), shuffle tree
        const int n = two_level ? nsub : Gs;                 // <= 32
        const double* src = blk_a + (two_level ? (size_t)Gs * SG_REC : 0);
        if (tid < 64) {
            MomSum r;
            ms_init(r);
            if (tid < n) r = ms_load(src + (size_t)tid * SG_REC);
            ms_wave_reduce(r);
            if (tid == 0) sh_rec[0] = r;
        }
        __syncthreads();
    } else {
        for (int ch = tid; ch < nch; ch += TPB) {
            MomSum r;
            ms_init(r);
            for (int s = 0; s < Gs; ++s) ms_merge(r, ms_load(blk_a + ((size_t)s * kk + ch) * SG_REC));
            sh_rec[ch] = r;
        }
        __syncthreads();
    }
    if (tid == 0) grp_depart(lines1, rb.member, Gs);
    for (int ch = tid; ch < nch; ch += TPB) {
        sh_mean[ch] = mean_of(sh_rec[ch]);
        sh_std[ch] = std_of(sh_rec[ch]);
    }
    __syncthreads();

    // ---- pass B on the same registers, exchange 2
    unsigned* lines2 = grp_lines(ws.cnt, rb.group, Gs, 1, 2);
    if (need_dev) {
        const float mean = sh_mean[mych];
        const float inv_sd = KURT ? 1.f / sh_std[mych] : 0.f;
        double ta = 0., tk = 0.;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < nrows) dev_acc4<KURT>(v[j], mean, inv_sd, ta, tk);
        wg_channel_dev(g, b, ok, ta, tk, l, sh_a, sh_k);
        double* blk_b = ws.rec_b + (size_t)rb.group * ws.slots * SG_DEV;
        for (int ch = tid; ch < nch; ch += TPB) {
            st_rec(blk_b + ((size_t)rb.member * kk + ch) * SG_DEV + 0, sh_a[ch]);
            st_rec(blk_b + ((size_t)rb.member * kk + ch) * SG_DEV + 1, sh_k[ch]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int timed_out = grp_meet(lines2, rb.member, Gs, flags, [&] {
                if (!two_level) return;
                asm volatile("" ::: "memory");
                const int si = rb.member / GRP_SUB;
                double ra = 0., rk = 0.;
                for (int m = si * GRP_SUB; m < min(Gs, (si + 1) * GRP_SUB); ++m) {
                    ra += ld_rec(blk_b + (size_t)m * SG_DEV + 0);
                    rk += ld_rec(blk_b + (size_t)m * SG_DEV + 1);
                }
                st_rec(blk_b + (size_t)(Gs + si) * SG_DEV + 0, ra);
                st_rec(blk_b + (size_t)(Gs + si) * SG_DEV + 1, rk);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }, /*wait=*/rb.member == 0);
            if (timed_out) atomicOr(ws.status, 1u);
            sh_timed_out = timed_out;
        }
        __syncthreads();
        // only member 0 reports (and waits); the others are done once the group has their records
        if (rb.member == 0) {
            if (sh_timed_out) {
                double ca = 0., ck = 0.;
                bool cok;
                if (g.mode == 1) {
                    const float m1 = sh_mean[0], i1 = KURT ? 1.f / sh_std[0] : 0.f;
                    for_all_samples(x, g, b, cok, [&](const float(&q)[4]) { dev_acc4<KURT>(q, m1, i1, ca, ck); });
                } else {
                    for_all_samples(x, g, b, cok, [&](const float(&q)[4]) { dev_acc4<KURT>(q, mean, inv_sd, ca, ck); });
                }
                wg_channel_dev(g, b, cok, ca, ck, l, sh_a, sh_k);
            } else if (g.mode == 1) {
                const int n = two_level ? nsub : Gs;
                const double* src = blk_b + (two_level ? (size_t)Gs * SG_DEV : 0);
                if (tid < 64) {
                    double ra = 0., rk = 0.;
                    if (tid < n) { ra = ld_rec(src + (size_t)tid * SG_DEV); rk = ld_rec(src + (size_t)tid * SG_DEV + 1); }
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) { ra += shfl_xor_d(ra, m); rk += shfl_xor_d(rk, m); }
                    if (tid == 0) { sh_a[0] = ra; sh_k[0] = rk; }
                }
                __syncthreads();
            } else {
                for (int ch = tid; ch < nch; ch += TPB) {
                    double ra = 0., rk = 0.;
                    for (int s = 0; s < Gs; ++s) {
                        ra += ld_rec(blk_b + ((size_t)s * kk + ch) * SG_DEV + 0);
                        rk += ld_rec(blk_b + ((size_t)s * kk + ch) * SG_DEV + 1);
                    }
                    sh_a[ch] = ra;
                    sh_k[ch] = rk;
                }
                __syncthreads();
            }
        }
    }

    // ---- the table (member 0 of every group)
    if (rb.member == 0) {
        const int C = g.C;
        for (int ch = tid; ch < nch; ch += TPB) {
            const int c = b.c0 + ch;
            const MomSum r = sh_rec[ch];
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
            stats[(size_t)CNNQ_STAT_MEAN * C + c] = sh_mean[ch];
            stats[(size_t)CNNQ_STAT_STD * C + c] = sh_std[ch];
            if constexpr (RELU) {
                double rv = (r.rss - r.rs * (r.rs / r.cnt)) / (r.cnt - 1.);
                if (rv < 0.) rv = 0.;
                stats[(size_t)CNNQ_STAT_STD_POS * C + c] = (float)sqrt(rv);
            }
            if (need_dev) {
                stats[(size_t)CNNQ_STAT_B * C + c] = (float)(sh_a[ch] / r.cnt);
                if constexpr (KURT) stats[(size_t)CNNQ_STAT_KURT * C + c] = (float)(sh_k[ch] / r.cnt - 3.);
            }
        }
    }
    if (need_dev && tid == 0) grp_depart(lines2, rb.member, Gs);
}

}  // namespace
