// cnnq_kernels.hip - gfx950 (MI355X / CDNA4) kernels behind include/cnnq_hip.h.
//
// Design: the whole path is HBM-bound elementwise + reduction work, so
// nothing here is shaped for MFMA.  All streaming kernels share ONE decomposition of the
// NCHW tensor x[N][C][HW]:
//
//   * the C*HW "plane" of one sample is cut into column blocks; a workgroup owns one column
//     block and walks it down the batch (n = n0 .. n1), so every lane keeps the SAME channel(s)
//     for its whole life: per-channel scale/zero-point are fetched once (staged through LDS)
//     and live in registers, the hot loop is load(16 B) -> ALU -> store(16 B), fully coalesced,
//     with no index division and no transposed copy;
//   * column blocks are aligned to channel boundaries: either a slice of ONE channel (mode 1,
//     large H*W) or k WHOLE channels (mode 2, small H*W), so reductions finish inside the
//     workgroup (wave64 shuffles + LDS) and each (group, channel) partial is written by
//     exactly one workgroup - no atomics, deterministic results;
//   * three load shapes: VEC4 (H*W % 4 == 0), VEC4-straddle (H*W % 4 != 0 but C*H*W % 4 == 0,
//     e.g. 7x7: a float4 may span two channels, per-element bookkeeping) and VEC1 (anything,
//     incl. unaligned base pointers);
//   * two launch geometries over that decomposition (make_geo): the REDUCTION passes (statistics)
//     use <= 64 batch splits, i.e. ~4096 long-lived workgroups that amortise their in-workgroup
//     reduction; the table-driven ELEMENTWISE passes (Q/DQ and friends) use ~14 KB of x per
//     workgroup, dispatched in address order - on MI355X read+write streaming reaches 6.1-6.7 TB/s
//     that way against 5.4 TB/s with long-lived workgroups; statistics passes over tensors too big
//     for the Infinity Cache use non-temporal loads (read-only 7.1 vs 6.3 TB/s).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (division must stay an IEEE
// divide followed by a separately rounded add: bit-exactness with the reference's aten ops).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "cnnq_hip.h"

namespace {

constexpr int TPB = 256;       // 4 wave64 per workgroup
constexpr int MAXCH = 256;     // max channels a workgroup owns (keeps the per-channel LDS tables at 1 KB each)

struct Geo {
    int N, C, HW;
    int P;      // C*HW, elements per sample plane (< 2^31)
    int mode;   // 1: block = slice of one channel, 2: block = k whole channels
    int nb, w;  // mode 1: blocks per channel, columns (loads) per block
    int k;      // mode 2: channels per block
    int ncb;    // column blocks per plane (of the channel range)
    int S;      // batch splits
    int cbeg;   // first channel of the range this launch covers
    int Cn;     // channels in the range
    int rev;    // 1: walk blocks and samples in descending address order (re-read what the
                //    previous pass touched LAST first: Infinity-Cache friendly)
};

struct Variant {
    int vec, A, J;  // elements per load, accumulator sets per load, loads per thread per sample
};

struct Blk {
    int c0, c1;      // channels [c0, c1)
    int col0, col1;  // plane columns [col0, col1), in units of VEC elements
    int n0, n1;      // samples [n0, n1)
    int grp;         // partial group index
};

template <int VEC>
__device__ __forceinline__ Blk blk_of(const Geo& g) {
    Blk b;
    const int bid = g.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int cb = bid % g.ncb;
    const int s = bid / g.ncb;
    b.n0 = (int)(((int64_t)s * g.N) / g.S);
    b.n1 = (int)(((int64_t)(s + 1) * g.N) / g.S);
    if (g.mode == 1) {
        const int cpc = g.HW / VEC;
        const int cr = cb / g.nb;
        const int bb = cb - cr * g.nb;
        const int c = g.cbeg + cr;
        b.c0 = c;
        b.c1 = c + 1;
        b.col0 = c * cpc + bb * g.w;
        b.col1 = min(b.col0 + g.w, (c + 1) * cpc);
        b.grp = s * g.nb + bb;
    } else {
        b.c0 = g.cbeg + cb * g.k;
        b.c1 = min(g.cbeg + g.Cn, b.c0 + g.k);
        b.col0 = (int)(((int64_t)b.c0 * g.HW) / VEC);
        b.col1 = (int)(((int64_t)b.c1 * g.HW) / VEC);
        b.grp = s;
    }
    return b;
}

template <int VEC>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}

template <int VEC>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        *p = v[0];
    }
}

typedef float f4_t __attribute__((ext_vector_type(4)));

// streaming (non-temporal) forms.  Read-only streaming with `nt` loads runs at 7.1 TB/s on MI355X
// against 6.3 TB/s with plain loads (tools/ubench_read.py); they do not allocate in the Infinity
// Cache, so the statistics passes use them only for tensors too large for the next pass to find
// anything still cached (NT_BYTES).  The Q/DQ pass always uses them: x is read for the last time and
// y is never re-read by this path.
constexpr int64_t NT_BYTES = (int64_t)384 << 20;   // swept 0..1000 MB on the ResNet-50 set: flat optimum 250-400

template <int VEC>
__device__ __forceinline__ void ldv_nt(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const f4_t t = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = __builtin_nontemporal_load(p);
    }
}
template <int VEC>
__device__ __forceinline__ void stv_nt(float* __restrict__ p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        f4_t t;
        t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        __builtin_nontemporal_store(t, reinterpret_cast<f4_t*>(p));
    } else {
        __builtin_nontemporal_store(v[0], p);
    }
}
template <int VEC, bool NTL>
__device__ __forceinline__ void ldv_sel(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (NTL) ldv_nt<VEC>(p, v); else ldv<VEC>(p, v);
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }

// ------------------------------------------------------------------------------------------
// Pass A: per-channel min / max / sum / sumsq / count (+ relu sums)
// ------------------------------------------------------------------------------------------
struct Mom {
    float mn, mx;
    double s, ss, rs, rss;
    __device__ __forceinline__ void init() {
        mn = INFINITY; mx = -INFINITY; s = 0.; ss = 0.; rs = 0.; rss = 0.;
    }
    template <bool RELU>
    __device__ __forceinline__ void add(float v) {
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
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
    template <bool RELU>
    __device__ __forceinline__ void merge(const Mom& o) {
        mn = fminf(mn, o.mn);
        mx = fmaxf(mx, o.mx);
        s += o.s;
        ss += o.ss;
        if constexpr (RELU) { rs += o.rs; rss += o.rss; }
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
#pragma unroll 2
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
    const int wv = tid >> 6, lane = tid & 63;
    const double count = (double)g.HW * rows;
    if (epc <= 16) {
        // tiny rows: one lane per channel, serial over its few entries
        for (int ch = b.c0 + tid; ch < b.c1; ch += TPB) {
            const int lo = (ch - b.c0) * epc;
            Mom r;
            r.init();
            for (int e = lo; e < lo + epc; ++e) {
                Mom o;
                o.mn = l_mn[e]; o.mx = l_mx[e]; o.s = l_s[e]; o.ss = l_ss[e];
                if constexpr (RELU) { o.rs = l_rs[e]; o.rss = l_rss[e]; }
                r.template merge<RELU>(o);
            }
            write_mom<RELU>(part, b.grp, g.C, ch, r, count);
        }
        return;
    }
    for (int ch = b.c0 + wv; ch < b.c1; ch += TPB / 64) {
        const int lo = (ch - b.c0) * epc;
        Mom r;
        r.init();
        for (int e = lo + lane; e < lo + epc; e += 64) {
            Mom o;
            o.mn = l_mn[e]; o.mx = l_mx[e]; o.s = l_s[e]; o.ss = l_ss[e];
            if constexpr (RELU) { o.rs = l_rs[e]; o.rss = l_rss[e]; }
            r.template merge<RELU>(o);
        }
        r.template wave_reduce<RELU>();
        if (lane == 0) write_mom<RELU>(part, b.grp, g.C, ch, r, count);
    }
}

// merge G records per channel; one wave64 per channel, lanes stride over the groups
__global__ void __launch_bounds__(TPB) k_combine(const double* __restrict__ part, int G, int C, int has_relu,
                                                 double* __restrict__ mom, float* __restrict__ stats) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * (TPB / 64) + wv;
    if (c >= C) return;
    double mn = INFINITY, mx = -INFINITY, s = 0., ss = 0., cnt = 0., rs = 0., rss = 0.;
    for (int gi = lane; gi < G; gi += 64) {
        const double* p = part + (size_t)gi * CNNQ_NMOM * C + c;
        mn = fmin(mn, p[(size_t)CNNQ_MOM_MIN * C]);
        mx = fmax(mx, p[(size_t)CNNQ_MOM_MAX * C]);
        s += p[(size_t)CNNQ_MOM_SUM * C];
        ss += p[(size_t)CNNQ_MOM_SUMSQ * C];
        cnt += p[(size_t)CNNQ_MOM_COUNT * C];
        if (has_relu) {
            rs += p[(size_t)CNNQ_MOM_SUM_RELU * C];
            rss += p[(size_t)CNNQ_MOM_SUMSQ_RELU * C];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mn = fmin(mn, shfl_xor_d(mn, m));
        mx = fmax(mx, shfl_xor_d(mx, m));
        s += shfl_xor_d(s, m);
        ss += shfl_xor_d(ss, m);
        cnt += shfl_xor_d(cnt, m);
        rs += shfl_xor_d(rs, m);
        rss += shfl_xor_d(rss, m);
    }
    if (lane != 0) return;
    if (mom) {
        mom[(size_t)CNNQ_MOM_MIN * C + c] = mn;
        mom[(size_t)CNNQ_MOM_MAX * C + c] = mx;
        mom[(size_t)CNNQ_MOM_SUM * C + c] = s;
        mom[(size_t)CNNQ_MOM_SUMSQ * C + c] = ss;
        mom[(size_t)CNNQ_MOM_COUNT * C + c] = cnt;
        mom[(size_t)CNNQ_MOM_SUM_RELU * C + c] = rs;
        mom[(size_t)CNNQ_MOM_SUMSQ_RELU * C + c] = rss;
    }
    if (stats) {
        const double mean = s / cnt;
        double var = (ss - s * mean) / (cnt - 1.);
        if (var < 0.) var = 0.;
        stats[(size_t)CNNQ_STAT_MIN * C + c] = (float)mn;
        stats[(size_t)CNNQ_STAT_MAX * C + c] = (float)mx;
        stats[(size_t)CNNQ_STAT_MEAN * C + c] = (float)mean;
        stats[(size_t)CNNQ_STAT_STD * C + c] = (float)sqrt(var);
        if (has_relu) {
            double rv = (rss - rs * (rs / cnt)) / (cnt - 1.);
            if (rv < 0.) rv = 0.;
            stats[(size_t)CNNQ_STAT_STD_POS * C + c] = (float)sqrt(rv);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Pass B: sum |x - mean| and sum ((x - mean)/std)^4 per channel
// ------------------------------------------------------------------------------------------
template <int VEC, int A, int J, bool KURT, bool NTL>
__global__ void __launch_bounds__(TPB) k_absdev(const float* __restrict__ x, const Geo g,
                                                const float* __restrict__ stats, double* __restrict__ part2) {
    constexpr int NE = TPB * J * A;
    __shared__ double l_a[NE];
    __shared__ double l_k[KURT ? NE : 1];
    __shared__ float sh_mean[MAXCH], sh_std[MAXCH];

    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    for (int i = tid; i < b.c1 - b.c0; i += TPB) {
        sh_mean[i] = stats[(size_t)CNNQ_STAT_MEAN * g.C + b.c0 + i];
        sh_std[i] = stats[(size_t)CNNQ_STAT_STD * g.C + b.c0 + i];
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
#pragma unroll 2
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
    if (epc <= 16) {
        for (int ch = b.c0 + tid; ch < b.c1; ch += TPB) {
            const int lo = (ch - b.c0) * epc;
            double ra = 0., rk = 0.;
            for (int e = lo; e < lo + epc; ++e) { ra += l_a[e]; if constexpr (KURT) rk += l_k[e]; }
            emit(ch, ra, rk);
        }
        return;
    }
    for (int ch = b.c0 + wv; ch < b.c1; ch += TPB / 64) {
        const int lo = (ch - b.c0) * epc;
        double ra = 0., rk = 0.;
        for (int e = lo + lane; e < lo + epc; e += 64) { ra += l_a[e]; if constexpr (KURT) rk += l_k[e]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ra += shfl_xor_d(ra, m); rk += shfl_xor_d(rk, m); }
        if (lane == 0) emit(ch, ra, rk);
    }
}

__global__ void __launch_bounds__(TPB) k_combine_dev(const double* __restrict__ part2, int G, int C,
                                                     const double* __restrict__ mom, int want_kurt,
                                                     double* __restrict__ dev_out, float* __restrict__ stats) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * (TPB / 64) + wv;
    if (c >= C) return;
    double sa = 0., sk = 0.;
    for (int gi = lane; gi < G; gi += 64) {
        const double* p = part2 + (size_t)gi * CNNQ_NDEV * C + c;
        sa += p[(size_t)CNNQ_DEV_ABS * C];
        sk += p[(size_t)CNNQ_DEV_Z4 * C];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { sa += shfl_xor_d(sa, m); sk += shfl_xor_d(sk, m); }
    if (lane != 0) return;
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

// ------------------------------------------------------------------------------------------
// statistics -> scale / zero point / qmax (one workgroup, no host round trips)
// ------------------------------------------------------------------------------------------
constexpr int PTPB = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = 0.;
    for (int i = 0; i < PTPB / 64; ++i) r += sh[i];
    return r;
}

__constant__ float c_laplace[9] = {1.05f, 1.86f, 2.83f, 3.89f, 5.03f, 6.2f, 7.41f, 8.64f, 9.89f};
__constant__ float c_laplace_pos[9] = {1.86f, 2.83f, 3.89f, 5.02f, 6.2f, 7.41f, 8.64f, 9.89f, 11.16f};
__constant__ float c_gaus[9] = {0.f, 1.24f, 1.71f, 2.15f, 2.55f, 2.93f, 3.28f, 3.61f, 3.92f};
__constant__ float c_gaus_pos[9] = {0.f, 1.71f, 2.15f, 2.55f, 2.93f, 3.28f, 3.61f, 3.92f, 4.2f};

__global__ void __launch_bounds__(PTPB) k_params(const float* __restrict__ stats, int C, const cnnq_params_cfg cfg,
                                                 float* __restrict__ qp, float* __restrict__ diag,
                                                 float* __restrict__ bits_ws) {
    __shared__ double sh[PTPB / 64];
    const int tid = threadIdx.x;
    const float* vmin = stats + (size_t)CNNQ_STAT_MIN * C;
    const float* vmax = stats + (size_t)CNNQ_STAT_MAX * C;
    const float* vmean = stats + (size_t)CNNQ_STAT_MEAN * C;
    const float* vstd = stats + (size_t)CNNQ_STAT_STD * C;
    const float* vb = stats + (size_t)CNNQ_STAT_B * C;
    const bool ba = cfg.bit_alloc && cfg.num_bits <= 4;

    if (ba) {
        // fixed-target bit allocation, iq.py:381-407 (fp32 tensor math, double target)
        const float* prior = cfg.prior_is_b ? vb : vstd;
        const float goal = (float)cfg.target;
        double target = cfg.target;
        double delta = 1.;
        // p = prior^(2/3) and its sum do not change between iterations
        double psum_d = 0.;
        for (int c = tid; c < C; c += PTPB) psum_d += (double)powf(prior[c], (float)(2. / 3));
        const float psum = (float)block_sum(psum_d, sh);
        for (int it = 0; it < 10 && fabs(2. * delta) > 0.01; ++it) {
            const float B = (float)((double)C * pow(2., target));
            double bsum = 0.;
            for (int c = tid; c < C; c += PTPB) {
                const float p = powf(prior[c], (float)(2. / 3));
                const float bins = (B * p) / psum;
                float bits = cfg.round_mode ? rintf(log2f(bins)) : ceilf(log2f(bins));
                if (bits < 0.f) bits = 0.f;
                if (bits > 8.f) bits = 8.f;
                bits_ws[c] = bits;
                bsum += (double)bits;
            }
            const float mean_bits = (float)block_sum(bsum, sh) / (float)C;
            delta = (double)((goal - mean_bits) / 2.f);
            target += delta;
        }
        __syncthreads();
    }
    for (int c = tid; c < C; c += PTPB) {
        const float bits = ba ? bits_ws[c] : (float)cfg.num_bits;
        float alpha = 0.f, delta, offset;
        if (cfg.clip == 0) {
            offset = cfg.positive ? 0.f : vmin[c];
            delta = vmax[c] - offset;
        } else {
            if (cfg.clip == 1) {
                const int ib = (int)bits;  // NaN bits cannot occur: clamped comparisons leave 0..8
                alpha = vb[c] * (cfg.positive ? c_laplace_pos[ib] : c_laplace[ib]);
            } else if (cfg.clip == 2) {
                alpha = vstd[c] * (cfg.positive ? c_gaus_pos[cfg.num_bits] : c_gaus[cfg.num_bits]);
            } else {
                alpha = cfg.pstd * vstd[c];
            }
            float range;
            if (cfg.positive) {
                range = fmaxf(vmean[c], 0.f) + alpha;
                offset = 0.f;
            } else {
                range = 2.f * alpha;
                offset = fmaxf(vmin[c], vmean[c] - alpha);
            }
            const float mx = offset + range;                   // iq.py:351
            delta = cfg.direct_range ? range : mx - offset;    // iq.py:443 (per channel) / :357 (per tensor)
        }
        float qmax, scale;
        if (ba) {
            qmax = exp2f(bits) - 1.f;
            scale = (qmax > 0.f) ? delta / qmax : 0.f;
        } else {
            qmax = (float)((1u << cfg.num_bits) - 1u);
            scale = delta / qmax;
        }
        scale = (scale < 1e-8f) ? 1e-8f : scale;  // NaN stays NaN, as torch.max does
        const float zp = rintf(0.f - offset / scale);
        qp[(size_t)CNNQ_QP_SCALE * C + c] = scale;
        qp[(size_t)CNNQ_QP_ZP * C + c] = zp;
        qp[(size_t)CNNQ_QP_QMAX * C + c] = qmax;
        if (diag) {
            diag[(size_t)CNNQ_DIAG_BITS * C + c] = bits;
            diag[(size_t)CNNQ_DIAG_ALPHA * C + c] = alpha;
            diag[(size_t)CNNQ_DIAG_DELTA * C + c] = delta;
            diag[(size_t)CNNQ_DIAG_OFFSET * C + c] = offset;
        }
    }
}

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
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int a = 0; a < A; ++a) { mn[j][a] = INFINITY; mx[j][a] = -INFINITY; }
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
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    mn[j][A == 1 ? 0 : e] = fminf(mn[j][A == 1 ? 0 : e], v[j][e]);
                    mx[j][A == 1 ? 0 : e] = fmaxf(mx[j][A == 1 ? 0 : e], v[j][e]);
                }
            }
        }
    }
    float* pn = pmm + (size_t)(2 * b.grp) * g.C;
    float* px = pn + g.C;
    const int wv = tid >> 6, lane = tid & 63;
    if (g.mode == 1) {
        float tn = INFINITY, tx = -INFINITY;
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (ok[j]) { tn = fminf(tn, mn[j][0]); tx = fmaxf(tx, mx[j][0]); }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { tn = fminf(tn, shfl_xor_f(tn, m)); tx = fmaxf(tx, shfl_xor_f(tx, m)); }
        if (lane == 0) { l_mn[wv] = tn; l_mx[wv] = tx; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < TPB / 64; ++i) { tn = fminf(tn, l_mn[i]); tx = fmaxf(tx, l_mx[i]); }
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
            for (int e = lo; e < lo + epc; ++e) { tn = fminf(tn, l_mn[e]); tx = fmaxf(tx, l_mx[e]); }
            pn[ch] = tn;
            px[ch] = tx;
        }
        return;
    }
    for (int ch = b.c0 + wv; ch < b.c1; ch += TPB / 64) {
        const int lo = (ch - b.c0) * epc;
        float tn = INFINITY, tx = -INFINITY;
        for (int e = lo + lane; e < lo + epc; e += 64) { tn = fminf(tn, l_mn[e]); tx = fmaxf(tx, l_mx[e]); }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { tn = fminf(tn, shfl_xor_f(tn, m)); tx = fmaxf(tx, shfl_xor_f(tx, m)); }
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
        mn = fminf(mn, pmm[(size_t)(2 * gi) * C + c]);
        mx = fmaxf(mx, pmm[(size_t)(2 * gi + 1) * C + c]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, shfl_xor_f(mn, m)); mx = fmaxf(mx, shfl_xor_f(mx, m)); }
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
    const float qm = (float)((1u << num_bits) - 1u);
    float sc = delta / qm;
    sc = (sc < 1e-8f) ? 1e-8f : sc;
    qp[(size_t)CNNQ_QP_SCALE * C + c] = sc;
    qp[(size_t)CNNQ_QP_ZP * C + c] = rintf(0.f - offset / sc);
    qp[(size_t)CNNQ_QP_QMAX * C + c] = qm;
}

// The fused Q/DQ.  Launched with MANY short workgroups in address order (about 14 KB of x each,
// see make_geo `fine`): measured on MI355X, read+write streaming runs at 6.1-6.7 TB/s this way
// against 5.4 TB/s when a workgroup walks 30+ samples (tools/ubench_copy.py, tools/split_probe.py).
template <int VEC, int A, int J, bool CODES, bool HIST>
__global__ void __launch_bounds__(TPB) k_qdq(const float* __restrict__ x, float* __restrict__ y, const Geo g,
                                             const float* __restrict__ qp, uint8_t* __restrict__ codes,
                                             unsigned long long* __restrict__ hist) {
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
    if (!single) {
        for (int i = tid; i < b.c1 - b.c0; i += TPB) {
            sh_sc[i] = qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0 + i];
            sh_zp[i] = qp[(size_t)CNNQ_QP_ZP * g.C + b.c0 + i];
            sh_qm[i] = qp[(size_t)CNNQ_QP_QMAX * g.C + b.c0 + i];
        }
        __syncthreads();
    }
    const float u_sc = single ? qp[(size_t)CNNQ_QP_SCALE * g.C + b.c0] : 0.f;
    const float u_zp = single ? qp[(size_t)CNNQ_QP_ZP * g.C + b.c0] : 0.f;
    const float u_qm = single ? qp[(size_t)CNNQ_QP_QMAX * g.C + b.c0] : 0.f;
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

// ------------------------------------------------------------------------------------------
// packed int4 storage: the integer codes of a <= 4-bit quantization, two per
// byte (even element in the low nibble), as the STORED activation format - 4 B read + 0.5 B written
// per element instead of 4 + 4; k_unpack4_dq reproduces the dequantized floats of k_qdq bit for bit
// ------------------------------------------------------------------------------------------
template <int J>
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
                const unsigned pk = ((unsigned)cd[0] & 15u) | (((unsigned)cd[1] & 15u) << 4) |
                                    (((unsigned)cd[2] & 15u) << 8) | (((unsigned)cd[3] & 15u) << 12);
                *reinterpret_cast<uint16_t*>(packed + (off + (size_t)col[j] * 4) / 2) = (uint16_t)pk;
            }
        }
    }
}

template <int J>
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
            const unsigned pk = *reinterpret_cast<const uint16_t*>(packed + (off + (size_t)col[j] * 4) / 2);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((float)((pk >> (4 * e)) & 15u) - zp[j]) * sc[j];   // iq.py:591-592
            if (ok[j]) stv_nt<4>(y + off + (size_t)col[j] * 4, o);
        }
    }
}

// ------------------------------------------------------------------------------------------
// mid-tread quantization with per-channel bin allocation (config 5, iq.py:128-225)
// ------------------------------------------------------------------------------------------
struct MtCfg {
    double target;  // bits; bins per channel on average = 2^target
    int clip;       // 1: laplace-prior clipping around the mean (activations), 0: min/max range (weights)
    int sym;        // 0: non-negative range (force_positive / half_range)
};

constexpr int MT_NB = CNNQ_MT_HIST_BINS;  // integer-code bins, codes -MT_NB/2 .. MT_NB/2-1

__global__ void __launch_bounds__(PTPB) k_mt_params(const float* __restrict__ stats, int C, const MtCfg cfg,
                                                    const double* __restrict__ tabs, int ntab,
                                                    float* __restrict__ mt) {
    __shared__ double sh[PTPB / 64];
    const int tid = threadIdx.x;
    const float* vmin = stats + (size_t)CNNQ_STAT_MIN * C;
    const float* vmax = stats + (size_t)CNNQ_STAT_MAX * C;
    const float* vmean = stats + (size_t)CNNQ_STAT_MEAN * C;
    const float* vstd = stats + (size_t)CNNQ_STAT_STD * C;
    const float* vb = stats + (size_t)CNNQ_STAT_B * C;
    const double* otab = tabs;
    const double* atab = tabs + ntab;
    // eq. 10 (iq.py:128-135): omega = round(C * 2^target * sigma^(2/3) / sum sigma^(2/3))
    double psum_d = 0.;
    for (int c = tid; c < C; c += PTPB) psum_d += (double)powf(vstd[c], (float)(2. / 3));
    const float psum = (float)block_sum(psum_d, sh);
    const float B = (float)((double)C * pow(2., cfg.target));
    for (int c = tid; c < C; c += PTPB) {
        const float p = powf(vstd[c], (float)(2. / 3));
        const float omega = rintf((B * p) / psum);
        float rng, am = 0.f;
        const float mu = vmean[c];
        const float mu0 = fmaxf(mu, 0.f);
        if (cfg.clip) {
            // linear interpolation in the (omega, alpha) table, fp64 like numpy (iq.py:137-145)
            const double om = (double)(cfg.sym ? omega : omega * 2.f);
            int lo = 0, hi = ntab;  // searchsorted, side='left'
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (otab[mid] < om) lo = mid + 1; else hi = mid;
            }
            const int i = lo < ntab ? lo : ntab - 1;     // (the reference raises beyond the table)
            const int im = i == 0 ? ntab - 1 : i - 1;    // numpy's index -1 wraps
            const double inc = (atab[i] - atab[im]) / (otab[i] - otab[im]);
            am = (float)(atab[i] - inc * (otab[i] - om));
            rng = cfg.sym ? (2.f * am) * vb[c] : mu0 + am * vb[c];
        } else {
            rng = cfg.sym ? vmax[c] - vmin[c] : vmax[c];
        }
        const float delta = (omega > 0.f) ? rng / omega : 3.402823466e+38f;
        float cmin = -INFINITY, cmax = INFINITY;
        if (cfg.clip) {
            const float muq = (cfg.sym ? mu : mu0) / delta;
            cmax = muq + (cfg.sym ? omega / 2.f : omega);
            cmin = cfg.sym ? muq - omega / 2.f : 0.f;
        }
        mt[(size_t)CNNQ_MT_DELTA * C + c] = delta;
        mt[(size_t)CNNQ_MT_CMIN * C + c] = cmin;
        mt[(size_t)CNNQ_MT_CMAX * C + c] = cmax;
        mt[(size_t)CNNQ_MT_OMEGA * C + c] = omega;
        mt[(size_t)CNNQ_MT_ALPHA * C + c] = am;
    }
}

// hist layout (uint64): [0, MT_NB) integer codes -MT_NB/2.., [MT_NB] below range, [MT_NB+1] above
// range, then C counts of "clamped to a non-integer c_min[c]" and C of "... c_max[c]".
template <int VEC, int A, int J, bool CLIP, bool HIST, bool CODES>
__global__ void __launch_bounds__(TPB) k_mt_qdq(const float* __restrict__ x, float* __restrict__ y, const Geo g,
                                                const float* __restrict__ mt, float* __restrict__ codes,
                                                unsigned long long* __restrict__ hist) {
    // LDS histogram window: MT_W integer codes starting at the smallest clamp bound of this
    // workgroup's channels (codes are >= c_min), MT_REP replicas by lane to spread equal codes;
    // codes beyond the window (a channel with > MT_W bins) go to the global bins directly.
    // The first MT_HOT codes of the window (where the mass is) get 32 lane-replicas = conflict-free,
    // the tail 8.
    constexpr int MT_W = 512, MT_HOT = 64, MT_REP = 8;
    constexpr int MT_WORDS = MT_HOT * 32 + (MT_W - MT_HOT) * MT_REP;
    auto hidx = [](unsigned kk, int tid) -> unsigned {
        return kk < (unsigned)MT_HOT ? kk * 32u + (unsigned)(tid & 31)
                                     : (unsigned)(MT_HOT * 32) + (kk - MT_HOT) * MT_REP + (unsigned)(tid & (MT_REP - 1));
    };
    __shared__ float sh_d[MAXCH], sh_lo[MAXCH], sh_hi[MAXCH];
    __shared__ unsigned sh_hist[HIST ? MT_WORDS : 1];
    __shared__ unsigned sh_clo[HIST ? MAXCH : 1], sh_chi[HIST ? MAXCH : 1];
    __shared__ int sh_wstart;
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    const int nch = b.c1 - b.c0;
    if constexpr (HIST) {
        for (int i = tid; i < MT_WORDS; i += TPB) sh_hist[i] = 0u;
        for (int i = tid; i < nch; i += TPB) { sh_clo[i] = 0u; sh_chi[i] = 0u; }
        if (tid == 0) sh_wstart = CLIP ? 0x7fffffff : -MT_W / 2;
    }
    __syncthreads();
    for (int i = tid; i < nch; i += TPB) {
        sh_d[i] = mt[(size_t)CNNQ_MT_DELTA * g.C + b.c0 + i];
        const float lo_i = mt[(size_t)CNNQ_MT_CMIN * g.C + b.c0 + i];
        sh_lo[i] = lo_i;
        sh_hi[i] = mt[(size_t)CNNQ_MT_CMAX * g.C + b.c0 + i];
        if constexpr (HIST && CLIP) atomicMin(&sh_wstart, (int)floorf(fminf(fmaxf(lo_i, -1e9f), 1e9f)));
    }
    __syncthreads();
    const int wstart = HIST ? max(sh_wstart, -MT_NB / 2) : 0;
    int col[J], chl[J][A];
    bool ok[J];
    float d[J][A], lo[J][A], hi[J][A];
    unsigned nzero = 0;  // code 0 (the mode of the distribution) is counted in a register, see k_qdq
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const unsigned e = (unsigned)col[j] * VEC + (A == 1 ? 0 : a);
            const int ch = (int)(e / (unsigned)g.HW) - b.c0;
            chl[j][a] = ch;
            d[j][a] = sh_d[ch];
            lo[j][a] = sh_lo[ch];
            hi[j][a] = sh_hi[ch];
        }
    }
    const int nrows = b.n1 - b.n0;
    constexpr int NU = (J == 1) ? 4 : 2;
#pragma unroll NU
    for (int r = 0; r < nrows; ++r) {
        const size_t off = (size_t)(b.n0 + r) * (size_t)g.P;
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv_nt<VEC>(x + off + (size_t)col[j] * VEC, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float o[VEC], q[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                float t = rintf(v[j][e] / d[j][a]);          // iq.py:202-203
                if constexpr (CLIP) {
                    // torch.min(t, hi) = t < hi ? t : hi and torch.max(t, lo) = t > lo ? t : lo, NaN kept
                    // (the bound wins ties: max(-0, +0) is +0, iq.py:213-214)
                    t = (t < hi[j][a] || t != t) ? t : hi[j][a];
                    t = (t > lo[j][a] || t != t) ? t : lo[j][a];
                }
                q[e] = t;
                o[e] = t * d[j][a];                          // iq.py:224
            }
            if (ok[j]) {
                stv_nt<VEC>(y + off + (size_t)col[j] * VEC, o);
                if constexpr (CODES) stv<VEC>(codes + off + (size_t)col[j] * VEC, q);
                if constexpr (HIST) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int a = (A == 1 ? 0 : e);
                        const float t = q[e];
                        // fast path (branch-light): integer code inside the LDS window
                        const int k = (int)t;                       // saturating; NaN -> 0
                        const bool isint = ((float)k == t);
                        const unsigned kk = (unsigned)(k - wstart);
                        if (t == 0.f) {
                            ++nzero;
                        } else if (isint && kk < (unsigned)MT_W) {
                            atomicAdd(&sh_hist[hidx(kk, tid)], 1u);
                        } else if (t == rintf(t)) {                 // rare: integer code outside the window
                            if (t >= (float)(-MT_NB / 2) && t < (float)(MT_NB / 2)) atomicAdd(&hist[(int)t + MT_NB / 2], 1ull);
                            else atomicAdd(&hist[t < 0.f ? MT_NB : MT_NB + 1], 1ull);
                        } else if (CLIP && t == hi[j][a]) {         // rare: clamped to a non-integer bound
                            atomicAdd(&sh_chi[chl[j][a]], 1u);
                        } else {
                            atomicAdd(&sh_clo[chl[j][a]], 1u);      // non-integer c_min (or NaN)
                        }
                    }
                }
            }
        }
    }
    if constexpr (HIST) {
        if (nzero) {
            const int kk = -wstart;
            if (kk >= 0 && kk < MT_W) atomicAdd(&sh_hist[hidx((unsigned)kk, tid)], nzero);
            else atomicAdd(&hist[MT_NB / 2], (unsigned long long)nzero);
        }
        __syncthreads();
        for (int i = tid; i < MT_W; i += TPB) {
            unsigned tot = 0;
            const int nrep = i < MT_HOT ? 32 : MT_REP;
            for (int r = 0; r < nrep; ++r) tot += sh_hist[hidx((unsigned)i, r + tid)];
            const int k = wstart + i;
            if (tot && k < MT_NB / 2) atomicAdd(&hist[k + MT_NB / 2], (unsigned long long)tot);
        }
        for (int i = tid; i < nch; i += TPB) {
            if (sh_clo[i]) atomicAdd(&hist[MT_NB + 2 + b.c0 + i], (unsigned long long)sh_clo[i]);
            if (sh_chi[i]) atomicAdd(&hist[MT_NB + 2 + g.C + b.c0 + i], (unsigned long long)sh_chi[i]);
        }
    }
}

// entropy over integer bins + the per-channel non-integer clamp values (equal values merged,
// as torch.unique would, utils/entropy.py:10)
__global__ void __launch_bounds__(PTPB) k_mt_entropy(const unsigned long long* __restrict__ hist,
                                                     const float* __restrict__ mt, int C, double total,
                                                     float* __restrict__ out) {
    __shared__ double sh[PTPB / 64];
    const int tid = threadIdx.x;
    const float ftotal = (float)total;
    double e = 0.;
    for (int i = tid; i < MT_NB + 2; i += PTPB) {
        const unsigned long long c = hist[i];
        if (c) { const float pr = (float)c / ftotal; e += (double)(-pr * log2f(pr)); }
    }
    // non-integer clamp values: one histogram entry per (channel, bound); equal values are merged.
    // The pairwise scan runs out of LDS (2*C <= MT_ENT entries), from global memory beyond that.
    const unsigned long long* cl = hist + MT_NB + 2;
    constexpr int MT_ENT = 4096;
    __shared__ float sv[MT_ENT];
    __shared__ unsigned sc[MT_ENT];
    const int n2 = 2 * C;
    const bool in_lds = n2 <= MT_ENT;
    if (in_lds) {
        for (int i = tid; i < n2; i += PTPB) {
            sv[i] = mt[(size_t)(i < C ? CNNQ_MT_CMIN : CNNQ_MT_CMAX) * C + (i < C ? i : i - C)];
            sc[i] = (unsigned)cl[i];
        }
    }
    __syncthreads();
    for (int i = tid; i < n2; i += PTPB) {
        const unsigned long long ci = in_lds ? sc[i] : cl[i];
        if (!ci) continue;
        const float vi = in_lds ? sv[i] : mt[(size_t)(i < C ? CNNQ_MT_CMIN : CNNQ_MT_CMAX) * C + (i < C ? i : i - C)];
        bool dup = false;
        unsigned long long cnt = ci;
        for (int j = 0; j < n2; ++j) {
            const unsigned long long cj = in_lds ? sc[j] : cl[j];
            if (j == i || !cj) continue;
            const float vj = in_lds ? sv[j] : mt[(size_t)(j < C ? CNNQ_MT_CMIN : CNNQ_MT_CMAX) * C + (j < C ? j : j - C)];
            if (vj == vi) { if (j < i) { dup = true; break; } cnt += cj; }
        }
        if (dup) continue;
        const float pr = (float)cnt / ftotal;
        e += (double)(-pr * log2f(pr));
    }
    const double r = block_sum(e, sh);
    if (tid == 0) out[0] = (float)r;
}

// ------------------------------------------------------------------------------------------
// bias / variance correction (iqm.py:180-196 activations, iqm.py:374-391 weights)
// ------------------------------------------------------------------------------------------
// per-channel affine update y = ((y - a) * m + a) - s + t  (weights): with
//   a = mean(w_q), m = std(w)/(std(w_q)+1e-8) (variance correction, optional), s = a, t = mean(w)
// evaluated with the reference's operation order so that equal constants give equal bits.
__global__ void __launch_bounds__(TPB) k_weight_correct(float* __restrict__ wq, int C, int HW,
                                                        const float* __restrict__ st_w,
                                                        const float* __restrict__ st_q, int vcorr, int bcorr) {
    const int c = blockIdx.y;
    const float bias_q = st_q[(size_t)CNNQ_STAT_MEAN * C + c];
    const float bias_o = st_w[(size_t)CNNQ_STAT_MEAN * C + c];
    const float var_corr = st_w[(size_t)CNNQ_STAT_STD * C + c] / (st_q[(size_t)CNNQ_STAT_STD * C + c] + 1e-8f);
    float* row = wq + (size_t)c * HW;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
        float v = row[i];
        if (vcorr) v = (v - bias_q) * var_corr + bias_q;   // iqm.py:387
        if (bcorr) v = v - bias_q + bias_o;                 // iqm.py:391 (bias_q is the pre-correction mean)
        row[i] = v;
    }
}

// activation bias correction, pass 1: per channel sum(x'), sum(y), count(x' > 0) with x' = relu(x)
// when the layer feeds a ReLU (iqm.py:188-193) -> part3[G][3][C] (fp64)
template <int VEC, int A, int J>
__global__ void __launch_bounds__(TPB) k_bcorr_sums(const float* __restrict__ x, const float* __restrict__ y,
                                                    const Geo g, int relu_first, double* __restrict__ part3) {
    constexpr int NE = TPB * J * A;
    __shared__ double l_sx[NE], l_sy[NE], l_cn[NE];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    int col[J];
    bool ok[J];
    double sx[J][A], sy[J][A], cn[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) { sx[j][a] = 0.; sy[j][a] = 0.; cn[j][a] = 0.; }
    }
    size_t off = (size_t)b.n0 * (size_t)g.P;
    for (int n = b.n0; n < b.n1; ++n, off += g.P) {
        float vx[J][VEC], vy[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            ldv<VEC>(x + off + (size_t)col[j] * VEC, vx[j]);
            ldv<VEC>(y + off + (size_t)col[j] * VEC, vy[j]);
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                const float xv = relu_first ? fmaxf(vx[j][e], 0.f) : vx[j][e];
                sx[j][a] += (double)xv;
                sy[j][a] += (double)vy[j][e];
                cn[j][a] += (xv > 0.f) ? 1. : 0.;
            }
    }
    auto emit = [&](int ch, double a, double bq, double c) {
        double* p = part3 + (size_t)b.grp * 3 * g.C + ch;
        p[0] = a;
        p[(size_t)g.C] = bq;
        p[(size_t)2 * g.C] = c;
    };
    const int wv = tid >> 6, lane = tid & 63;
    if (g.mode == 1) {
        double ta = 0., tb = 0., tc = 0.;
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (ok[j]) { ta += sx[j][0]; tb += sy[j][0]; tc += cn[j][0]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ta += shfl_xor_d(ta, m); tb += shfl_xor_d(tb, m); tc += shfl_xor_d(tc, m); }
        if (lane == 0) { l_sx[wv] = ta; l_sy[wv] = tb; l_cn[wv] = tc; }
        __syncthreads();
        if (tid == 0) {
            double ra = 0., rb = 0., rc = 0.;
            for (int i = 0; i < TPB / 64; ++i) { ra += l_sx[i]; rb += l_sy[i]; rc += l_cn[i]; }
            emit(b.c0, ra, rb, rc);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (ok[j]) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const int e = (j * TPB + tid) * A + a;
                l_sx[e] = sx[j][a]; l_sy[e] = sy[j][a]; l_cn[e] = cn[j][a];
            }
        }
    __syncthreads();
    const int epc = g.HW * A / VEC;
    for (int ch = b.c0 + wv; ch < b.c1; ch += TPB / 64) {
        const int lo = (ch - b.c0) * epc;
        double ra = 0., rb = 0., rc = 0.;
        for (int e = lo + lane; e < lo + epc; e += 64) { ra += l_sx[e]; rb += l_sy[e]; rc += l_cn[e]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { ra += shfl_xor_d(ra, m); rb += shfl_xor_d(rb, m); rc += shfl_xor_d(rc, m); }
        if (lane == 0) emit(ch, ra, rb, rc);
    }
}

// merge G records -> q_bias[c] = (sum x' - sum y) / (count + 1e-8)   (iqm.py:192-194); sums[3][C] optional
__global__ void __launch_bounds__(TPB) k_bcorr_bias(const double* __restrict__ part3, int G, int C,
                                                    double* __restrict__ sums, float* __restrict__ bias) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= C) return;
    double a = 0., bq = 0., cn = 0.;
    for (int gi = 0; gi < G; ++gi) {
        const double* p = part3 + (size_t)gi * 3 * C + c;
        a += p[0]; bq += p[(size_t)C]; cn += p[(size_t)2 * C];
    }
    if (sums) { sums[c] = a; sums[(size_t)C + c] = bq; sums[(size_t)2 * C + c] = cn; }
    if (bias) {
        const float qb = (float)a - (float)bq;
        bias[c] = qb / ((float)cn + 1e-8f);
    }
}

// pass 2: y += (y > 0) * q_bias[c]   (iqm.py:196), in place
template <int VEC, int A, int J>
__global__ void __launch_bounds__(TPB) k_bcorr_apply(float* __restrict__ y, const Geo g,
                                                     const float* __restrict__ bias) {
    __shared__ float sh_b[MAXCH];
    const Blk b = blk_of<VEC>(g);
    const int tid = threadIdx.x;
    for (int i = tid; i < b.c1 - b.c0; i += TPB) sh_b[i] = bias[b.c0 + i];
    __syncthreads();
    int col[J];
    bool ok[J];
    float qb[J][A];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = b.col0 + j * TPB + tid;
        ok[j] = c < b.col1;
        col[j] = ok[j] ? c : b.col0;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const unsigned e = (unsigned)col[j] * VEC + (A == 1 ? 0 : a);
            qb[j][a] = sh_b[(int)(e / (unsigned)g.HW) - b.c0];
        }
    }
    size_t off = (size_t)b.n0 * (size_t)g.P;
#pragma unroll 2
    for (int n = b.n0; n < b.n1; ++n, off += g.P) {
        float v[J][VEC];
#pragma unroll
        for (int j = 0; j < J; ++j) ldv<VEC>(y + off + (size_t)col[j] * VEC, v[j]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int a = (A == 1 ? 0 : e);
                v[j][e] = v[j][e] + ((v[j][e] > 0.f) ? 1.f : 0.f) * qb[j][a];
            }
            if (ok[j]) stv<VEC>(y + off + (size_t)col[j] * VEC, v[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// per-tensor GEMMLOWP path (replaces kernels/gemmlowp.cu)
// ------------------------------------------------------------------------------------------
// ptp: [0] scale [1] shift [2] qmax [3] true-zero flag [4] passthrough flag [5] range [6] offset
__global__ void __launch_bounds__(64) k_pt_setup(int have_host, float h_range, float h_offset,
                                                 const float* __restrict__ stats, int64_t stride, int rows,
                                                 int rows_mode, int zero_min, int num_bits, int int_exp, int etz,
                                                 float* __restrict__ ptp) {
    const int lane = threadIdx.x;
    float range, offset;
    bool ptz;
    if (have_host) {
        range = h_range;
        offset = h_offset;
        ptz = etz != 0;
    } else {
        const float* vmin = stats + (size_t)CNNQ_STAT_MIN * stride;
        const float* vmax = stats + (size_t)CNNQ_STAT_MAX * stride;
        float mn, mx;
        if (rows_mode == 0) {  // per-sample then mean over the batch (iq.py:515-526)
            double smn = 0., smx = 0.;
            for (int r = lane; r < rows; r += 64) { smn += (double)vmin[r]; smx += (double)vmax[r]; }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { smn += shfl_xor_d(smn, m); smx += shfl_xor_d(smx, m); }
            mn = (float)(smn / (double)rows);
            mx = (float)(smx / (double)rows);
        } else {  // whole tensor
            mn = INFINITY; mx = -INFINITY;
            for (int r = lane; r < rows; r += 64) { mn = fminf(mn, vmin[r]); mx = fmaxf(mx, vmax[r]); }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, shfl_xor_f(mn, m)); mx = fmaxf(mx, shfl_xor_f(mx, m)); }
        }
        if (zero_min) mn = 0.f;
        range = mx - mn;   // iq.py:379
        offset = mn;
        ptz = etz && ((offset + range) > 0.f) && (offset < 0.f);  // iq.py:613
    }
    if (lane != 0) return;
    const float qmax = (float)((1ll << num_bits) - 1);
    float scale = range / qmax;
    if (int_exp) scale = powf(2.f, (float)(int)ceilf(log2f(scale)));
    const float zero_point = roundf(-offset / scale);
    ptp[0] = scale;
    ptp[1] = ptz ? zero_point : -offset;
    ptp[2] = qmax;
    ptp[3] = ptz ? 1.f : 0.f;
    ptp[4] = (range <= 0.f) ? 1.f : 0.f;
    ptp[5] = range;
    ptp[6] = offset;
    ptp[7] = 0.f;
}

__device__ __forceinline__ float ptq1(float v, float scale, float shift, float qmax, bool etz, float nz) {
    float t = etz ? (v / scale) + shift : (v + shift) / scale;
    t = t + nz;  // the reference always adds the noise tensor (zeros when not stochastic)
    t = fminf(t, qmax);
    t = fmaxf(t, 0.f);
    t = roundf(t);
    return etz ? (t - shift) * scale : t * scale - shift;
}

// one-shot grid in address order, non-temporal streaming (the structure that reaches the copy
// ceiling on MI355X, tools/ubench_copy.py): every lane handles exactly one VEC-wide item
template <int VEC, bool NOISE>
__global__ void __launch_bounds__(TPB) k_pt_qdq(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                const float* __restrict__ ptp, const float* __restrict__ noise) {
    const float scale = ptp[0], shift = ptp[1], qmax = ptp[2];
    const bool etz = ptp[3] != 0.f, pass = ptp[4] != 0.f;
    const int64_t nv = n / VEC;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < nv) {
        float v[VEC], z[VEC], o[VEC];
        ldv_nt<VEC>(x + i * VEC, v);
        if constexpr (NOISE) ldv_nt<VEC>(noise + i * VEC, z);
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = pass ? v[e] : ptq1(v[e], scale, shift, qmax, etz, NOISE ? z[e] : 0.f);
        stv_nt<VEC>(y + i * VEC, o);
    }
    if constexpr (VEC > 1) {  // tail (n % VEC elements), handled by the first lanes of the grid
        const int64_t t = nv * VEC + i;
        if (i < VEC && t < n) y[t] = pass ? x[t] : ptq1(x[t], scale, shift, qmax, etz, NOISE ? noise[t] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// host side: geometry and launches
// ------------------------------------------------------------------------------------------
int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

constexpr int MAXG = 64;   // upper bound on batch splits S (partial groups per channel = S * nb)

// Load shape for a tensor.  For the aligned float4 shape the loads per lane per sample (J) adapt
// to the geometry: 4 when one channel row is long (H*W/4 > 1024 -> a workgroup owns a slice of a
// channel), otherwise the largest of {4, 2, 1} that still yields >= 2048 workgroups, so that
// small-H*W layers (14x14, 28x28 with few channels) fill the 256 CUs.
int choose_variant(int64_t N, int64_t C, int64_t HW, bool aligned16, Variant* v) {
    if (aligned16 && HW % 4 == 0) {
        const int64_t cpc = HW / 4;
        int J = 4;
        if (cpc <= TPB * 4) {
            const int64_t smax = N < MAXG ? N : MAXG;
            for (J = 4; J > 1; J >>= 1) {
                const int64_t cap = TPB * J;
                const int64_t ncb = (cpc > cap) ? C * ((cpc + cap - 1) / cap) : (C + cap / cpc - 1) / (cap / cpc);
                if (ncb * smax >= 2048) break;
            }
        }
        *v = {4, 1, J};
        return 0;
    }
    if (aligned16 && (C * HW) % 4 == 0) {
        const int m = 4 / gcd_i((int)(HW % 4), 4);
        if ((int64_t)m * HW <= TPB * 4) { *v = {4, 4, 1}; return 0; }
    }
    *v = {1, 1, 4};
    return 0;
}

// Geometry of one launch over channels [cbeg, cbeg + Cn) of x[N][C][HW].
// max_groups > 0 bounds the batch splits S of the passes that emit one partial record per group
// and channel (they all use MAXG, so they share one group count G = S * nb); `fine` requests the
// short-workgroup geometry of the table-driven elementwise passes instead.
int make_geo(int64_t N, int64_t C, int64_t HW, const Variant& v, int64_t cbeg, int64_t Cn, int max_groups, int rev,
             int fine, Geo* g) {
    if (N <= 0 || C <= 0 || HW <= 0 || cbeg < 0 || Cn <= 0 || cbeg + Cn > C) return CNNQ_EINVAL;
    if (C * HW >= (int64_t)1 << 31 || N >= (int64_t)1 << 31) return CNNQ_ERANGE;
    g->N = (int)N; g->C = (int)C; g->HW = (int)HW; g->P = (int)(C * HW);
    g->cbeg = (int)cbeg; g->Cn = (int)Cn; g->rev = rev;
    g->nb = 1; g->w = 0; g->k = 1;
    const int cap = TPB * v.J;  // loads per block per sample
    if (v.A == 4) {             // straddle: k whole channels with k*HW % 4 == 0
        const int m = 4 / gcd_i((int)(HW % 4), 4);
        if (cbeg % m != 0) return CNNQ_EINVAL;  // the range must start on a 16-byte boundary
        int k = (int)((cap * 4) / HW);
        if (k > MAXCH) k = MAXCH;
        k -= k % m;
        g->mode = 2;
        g->k = k;
        g->ncb = (int)((Cn + k - 1) / k);
    } else {
        const int64_t cpc = HW / v.vec;
        if (cpc > cap) {
            g->mode = 1;
            int64_t nb = (cpc + cap - 1) / cap;
            const int64_t w = (cpc + nb - 1) / nb;
            nb = (cpc + w - 1) / w;
            if (Cn * nb >= (int64_t)1 << 31) return CNNQ_ERANGE;
            g->nb = (int)nb;
            g->w = (int)w;
            g->ncb = (int)(Cn * nb);
        } else {
            g->mode = 2;
            g->k = (int)(cap / cpc);
            if (g->k > MAXCH) g->k = MAXCH;
            g->ncb = (int)((Cn + g->k - 1) / g->k);
        }
    }
    // enough workgroups to fill 256 CUs (x 6-8 resident each) a few times over
    const int64_t target = 4096;
    int64_t S = (target + g->ncb - 1) / g->ncb;
    if (S > N) S = N;
    if (max_groups > 0 && S > max_groups) S = max_groups;
    if (fine) {
        // table-driven elementwise passes (no per-workgroup reduction or partial record): many short
        // workgroups dispatched in address order - about 14 KB of x per workgroup - stream read+write
        // markedly faster than long-lived ones (6.1-6.7 vs 5.4 TB/s measured)
        const int64_t cols = (g->mode == 1) ? g->w : (int64_t)g->k * HW * v.A / v.vec / v.A;
        const int64_t row_bytes = cols * v.vec * 4;
        int64_t rows = (14336 + row_bytes / 2) / (row_bytes > 0 ? row_bytes : 1);   // swept 8-32 KB
        if (rows < 1) rows = 1;
        S = (N + rows - 1) / rows;
    }
    if (S < 1) S = 1;
    g->S = (int)S;
    if ((int64_t)g->S * g->ncb >= (int64_t)1 << 31) return CNNQ_ERANGE;
    return 0;
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline int launch_status() { return (int)hipGetLastError(); }

// one plan (load shape + geometry) per tensor, shared by every pass over it
int plan(int64_t N, int64_t C, int64_t HW, bool aligned16, int rev, Variant* v, Geo* g, int fine = 0) {
    choose_variant(N, C, HW, aligned16, v);
    return make_geo(N, C, HW, *v, 0, C, MAXG, rev, fine, g);
}

// dispatch on the runtime load shape: invokes F<VEC, A, J>()
#define CNNQ_DISPATCH(v, F)                                          \
    do {                                                             \
        if ((v).vec == 4 && (v).A == 1) {                            \
            if ((v).J == 4) { F(4, 1, 4); }                          \
            else if ((v).J == 2) { F(4, 1, 2); }                     \
            else { F(4, 1, 1); }                                     \
        } else if ((v).vec == 4) { F(4, 4, 1); }                     \
        else { F(1, 1, 4); }                                         \
    } while (0)

int launch_qdq(const float* x, float* y, const Geo& g, const Variant& v, const float* qp, uint8_t* codes,
               unsigned long long* h, hipStream_t st) {
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
#define LAUNCH_QDQ(VEC, A, J)                                                                                       \
    do {                                                                                                            \
        if (codes && h) hipLaunchKernelGGL((k_qdq<VEC, A, J, true, true>), grid, block, 0, st, x, y, g, qp, codes, h);   \
        else if (codes) hipLaunchKernelGGL((k_qdq<VEC, A, J, true, false>), grid, block, 0, st, x, y, g, qp, codes, h);  \
        else if (h) hipLaunchKernelGGL((k_qdq<VEC, A, J, false, true>), grid, block, 0, st, x, y, g, qp, codes, h);      \
        else hipLaunchKernelGGL((k_qdq<VEC, A, J, false, false>), grid, block, 0, st, x, y, g, qp, codes, h);            \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_QDQ);
#undef LAUNCH_QDQ
    return launch_status();
}

// ------------------------------------------------------------------------------------------
// KLD calibration (kld_threshold.py:6-84 per sample, statistic_manager.py:80-82)
//   k_kld_hist    per-row 2001-bin histogram over [-th, th], th = max(|min|, |max|): numpy.histogram's
//                 uniform-bin rule (float64 edges k*step + first, last edge inclusive; the index estimate
//                 is corrected by one step against the edges, so any estimate within one bin of the
//                 truth gives the identical, canonical bin - the multiply by 2001/(last-first) here
//                 instead of numpy's divide-then-multiply cannot change a count)
//   k_kld_search  one workgroup per (row, candidate): P = kept bins with the outliers folded into the
//                 ends, Q = 15 merged groups spread over their non-empty bins (the last group's
//                 expansion stops one bin short, kld_threshold.py:62-65), both smoothed in float32
//                 exactly as the reference, KL(P || Q) accumulated in fp64 (the reference: float32)
//   k_kld_pick    numpy.argmin over the 994 divergences (first NaN wins, else first minimum)
// ------------------------------------------------------------------------------------------
constexpr int KB = CNNQ_KLD_BINS;
constexpr int KQ = CNNQ_KLD_QBINS;
constexpr int KC = CNNQ_KLD_NCAND;
constexpr int KREP = 8;          // LDS replicas of the row histogram (64 KB)
constexpr int KCHUNK = 65536;    // elements of one row per workgroup
static_assert(KC == KB / 2 + 1 - KQ / 2, "candidate count");

struct KldRange {
    double first, last, step, scale;
};
__device__ __forceinline__ KldRange kld_range(float mn, float mx) {
    KldRange r;
    const double th = fmax(fabs((double)mn), fabs((double)mx));
    r.first = -th;
    r.last = th;
    if (r.first == r.last) {   // numpy widens an empty range (all-zero sample) by 0.5 either side
        r.first -= 0.5;
        r.last += 0.5;
    }
    const double den = r.last - r.first;
    r.step = den / (double)KB;
    r.scale = (double)KB / den;
    return r;
}
__device__ __forceinline__ double kld_edge(const KldRange& r, int k) {
    return k == KB ? r.last : (double)k * r.step + r.first;
}

template <int VEC>
__global__ void __launch_bounds__(TPB) k_kld_hist(const float* __restrict__ x, int64_t len,
                                                  const float* __restrict__ rowmm, int rows,
                                                  unsigned* __restrict__ hist) {
    __shared__ unsigned sh[KB * KREP];
    const int row = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < KB * KREP; i += TPB) sh[i] = 0;
    const KldRange r = kld_range(rowmm[row], rowmm[rows + row]);
    __syncthreads();
    const int64_t beg = (int64_t)blockIdx.x * KCHUNK;
    const int64_t end = min(beg + (int64_t)KCHUNK, len);
    const float* __restrict__ xr = x + (int64_t)row * len;
    const int rep = tid & (KREP - 1);
    auto put = [&](float v) {
        const double a = (double)v;
        if (!(a >= r.first && a <= r.last)) return;   // NaN (numpy keeps only first <= a <= last)
        int idx = (int)((a - r.first) * r.scale);
        idx = min(max(idx, 0), KB - 1);
        if (a < kld_edge(r, idx)) --idx;
        if (idx != KB - 1 && a >= kld_edge(r, idx + 1)) ++idx;
        idx = min(max(idx, 0), KB - 1);
        atomicAdd(&sh[idx * KREP + rep], 1u);
    };
    if constexpr (VEC == 4) {
        for (int64_t i = beg + (int64_t)tid * 4; i < end; i += TPB * 4) {
            if (i + 4 <= end) {
                float v[4];
                ldv_nt<4>(xr + i, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) put(v[e]);
            } else {
                for (int64_t j = i; j < end; ++j) put(xr[j]);
            }
        }
    } else {
        for (int64_t i = beg + tid; i < end; i += TPB) put(xr[i]);
    }
    __syncthreads();
    unsigned* __restrict__ hr = hist + (size_t)row * KB;
    for (int b = tid; b < KB; b += TPB) {
        unsigned c = 0;
#pragma unroll
        for (int q = 0; q < KREP; ++q) c += sh[b * KREP + q];
        if (c) atomicAdd(&hr[b], c);
    }
}

__global__ void __launch_bounds__(TPB) k_kld_search(const unsigned* __restrict__ hist, double* __restrict__ div) {
    constexpr int PER = (KB + TPB - 1) / TPB;   // bins per thread in the scan
    static_assert(PER * TPB > KB, "the scan must also produce the total");
    __shared__ unsigned sh[KB];
    __shared__ unsigned long long cs[KB + 1];   // cs[k] = counts in bins [0, k)
    __shared__ unsigned cz[KB + 1];             // cz[k] = non-empty bins in [0, k)
    __shared__ unsigned long long wtot[TPB / 64];
    __shared__ unsigned wnz[TPB / 64];
    __shared__ float qlevel[KQ];
    __shared__ double red[2 * (TPB / 64)];
    const int row = blockIdx.y, cand = blockIdx.x, tid = threadIdx.x;
    const int wv = tid >> 6, lane = tid & 63;
    const unsigned* __restrict__ h = hist + (size_t)row * KB;
    {
        unsigned v[PER];
        unsigned long long s = 0;
        unsigned z = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int k = tid * PER + e;
            v[e] = k < KB ? h[k] : 0u;
            s += v[e];
            z += v[e] != 0u;
        }
        unsigned long long si = s;
        unsigned zi = z;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long a = __shfl_up(si, d, 64);
            const unsigned b = __shfl_up(zi, d, 64);
            if (lane >= d) {
                si += a;
                zi += b;
            }
        }
        if (lane == 63) {
            wtot[wv] = si;
            wnz[wv] = zi;
        }
        __syncthreads();
        unsigned long long run = si - s;
        unsigned zr = zi - z;
        for (int i = 0; i < wv; ++i) {
            run += wtot[i];
            zr += wnz[i];
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int k = tid * PER + e;
            if (k <= KB) {
                cs[k] = run;
                cz[k] = zr;
            }
            if (k < KB) sh[k] = v[e];
            run += v[e];
            zr += v[e] != 0u;
        }
    }
    __syncthreads();
    const int i = cand + KQ / 2;               // bins kept either side of the zero bin
    const int start = KB / 2 - i, stop = KB / 2 + i + 1;
    const int m = stop - start, w = m / KQ;
    if (tid < KQ) {
        const int a = start + tid * w;
        const int b = tid == KQ - 1 ? stop : a + w;
        const int bn = tid == KQ - 1 ? stop - 1 : b;          // the expansion never writes the last bin
        const unsigned long long mass = cs[b] - cs[a];
        const unsigned norm = cz[bn] - cz[a];
        qlevel[tid] = norm ? (float)((double)mass / (double)norm) : 0.f;
    }
    const unsigned long long left = cs[start], right = cs[KB] - cs[stop];
    const unsigned long long p_first = sh[start] + left, p_last = sh[stop - 1] + right;
    const int pz = (m - 2) - (int)(cz[stop - 1] - cz[start + 1]) + (p_first == 0) + (p_last == 0);
    const int qz = (m - 1) - (int)(cz[stop - 1] - cz[start]) + 1;
    const float eps = 0.0001f;
    const float negp = (float)(-(0.0001 * (double)pz / (double)(m - pz)));
    const float negq = (float)(-(0.0001 * (double)qz / (double)(m - qz)));
    __syncthreads();
    float ps[PER], qs[PER];
    double P = 0., Q = 0.;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int kk = tid + e * TPB;
        ps[e] = 0.f;
        qs[e] = 0.f;
        if (kk < m) {
            const unsigned c = sh[start + kk];
            const unsigned long long pc = kk == 0 ? p_first : (kk == m - 1 ? p_last : (unsigned long long)c);
            ps[e] = (float)(long long)pc + (pc == 0 ? eps : negp);
            const float q = (c == 0u || kk == m - 1) ? 0.f : qlevel[min(kk / w, KQ - 1)];
            qs[e] = q + (q == 0.f ? eps : negq);
            P += (double)ps[e];
            Q += (double)qs[e];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        P += shfl_xor_d(P, d);
        Q += shfl_xor_d(Q, d);
    }
    if (lane == 0) {
        red[wv] = P;
        red[TPB / 64 + wv] = Q;
    }
    __syncthreads();
    P = 0.;
    Q = 0.;
    for (int j = 0; j < TPB / 64; ++j) {
        P += red[j];
        Q += red[TPB / 64 + j];
    }
    double kl = 0.;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int kk = tid + e * TPB;
        if (kk < m) {
            const double pk = (double)ps[e] / P, qk = (double)qs[e] / Q;
            kl += pk * log(pk / qk);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kl += shfl_xor_d(kl, d);
    __syncthreads();
    if (lane == 0) red[wv] = kl;
    __syncthreads();
    if (tid == 0) {
        kl = 0.;
        for (int j = 0; j < TPB / 64; ++j) kl += red[j];
        // an all-empty Q (nothing kept besides, at most, the last bin) or an empty row: the reference's
        // entropy() returns nan there (kld_threshold.py:72-76)
        if (qz == m || pz == m) kl = __longlong_as_double(0x7ff8000000000000LL);
        div[(size_t)row * KC + cand] = kl;
    }
}

__global__ void __launch_bounds__(64) k_kld_pick(const double* __restrict__ div, const float* __restrict__ rowmm,
                                                 int rows, double* __restrict__ out) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const double* __restrict__ d = div + (size_t)row * KC;
    int nan_idx = KC, best_idx = KC;
    double best = __longlong_as_double(0x7ff0000000000000LL);
    for (int c = lane; c < KC; c += 64) {
        const double v = d[c];
        if (v != v) nan_idx = min(nan_idx, c);
        else if (v < best || (v == best && c < best_idx)) {
            best = v;
            best_idx = c;
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const double ob = shfl_xor_d(best, s);
        const int oi = __shfl_xor(best_idx, s, 64);
        const int on = __shfl_xor(nan_idx, s, 64);
        nan_idx = min(nan_idx, on);
        if (ob < best || (ob == best && oi < best_idx)) {
            best = ob;
            best_idx = oi;
        }
    }
    if (lane == 0) {
        const int k = nan_idx < KC ? nan_idx : min(best_idx, KC - 1);
        const KldRange r = kld_range(rowmm[row], rowmm[rows + row]);
        out[(size_t)row * 3 + 0] = kld_edge(r, KB / 2 + (k + KQ / 2) + 1);
        out[(size_t)row * 3 + 1] = d[k];
        out[(size_t)row * 3 + 2] = (double)k;
    }
}

}  // namespace

extern "C" {

const char* cnnq_version(void) { return "cnnq-hip 0.2 gfx950"; }

int cnnq_pc_groups(int64_t N, int64_t C, int64_t HW, int aligned16) {
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, aligned16 != 0, 0, &v, &g);
    if (rc) return rc;
    return g.S * g.nb;
}

int cnnq_plan_describe(int64_t N, int64_t C, int64_t HW, int aligned16, int fine, int32_t out[12]) {
    if (!out) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, aligned16 != 0, 0, &v, &g, fine);
    if (rc) return rc;
    const int32_t vals[12] = {v.vec, v.A, v.J, g.mode, g.nb, g.w, g.k, g.ncb, g.S, TPB, g.S * g.nb, 0};
    for (int i = 0; i < 12; ++i) out[i] = vals[i];
    return 0;
}

int cnnq_pc_moments(const float* x, int64_t N, int64_t C, int64_t HW, int want_relu, double* part, void* stream) {
    if (!x || !part) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x), 0, &v, &g);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
#define LAUNCH_MOM(VEC, A, J)                                                                                        \
    do {                                                                                                             \
        if (want_relu && ntl) hipLaunchKernelGGL((k_moments<VEC, A, J, true, true>), grid, block, 0, st, x, g, part);   \
        else if (want_relu) hipLaunchKernelGGL((k_moments<VEC, A, J, true, false>), grid, block, 0, st, x, g, part);    \
        else if (ntl) hipLaunchKernelGGL((k_moments<VEC, A, J, false, true>), grid, block, 0, st, x, g, part);          \
        else hipLaunchKernelGGL((k_moments<VEC, A, J, false, false>), grid, block, 0, st, x, g, part);                  \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_MOM);
#undef LAUNCH_MOM
    return launch_status();
}

int cnnq_pc_combine(const double* part, int G, int64_t C, int has_relu, double* mom, float* stats, void* stream) {
    if (!part || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || (!mom && !stats)) return CNNQ_EINVAL;
    const dim3 grid((unsigned)((C + TPB / 64 - 1) / (TPB / 64))), block(TPB);
    hipLaunchKernelGGL(k_combine, grid, block, 0, (hipStream_t)stream, part, G, (int)C, has_relu, mom, stats);
    return launch_status();
}

int cnnq_pc_absdev(const float* x, int64_t N, int64_t C, int64_t HW, const float* stats, int want_kurt,
                   double* part2, void* stream) {
    if (!x || !stats || !part2) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x), /*rev=*/1, &v, &g);   // descending: follows the ascending pass A
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
#define LAUNCH_DEV(VEC, A, J)                                                                                             \
    do {                                                                                                                  \
        if (want_kurt && ntl) hipLaunchKernelGGL((k_absdev<VEC, A, J, true, true>), grid, block, 0, st, x, g, stats, part2);  \
        else if (want_kurt) hipLaunchKernelGGL((k_absdev<VEC, A, J, true, false>), grid, block, 0, st, x, g, stats, part2);   \
        else if (ntl) hipLaunchKernelGGL((k_absdev<VEC, A, J, false, true>), grid, block, 0, st, x, g, stats, part2);         \
        else hipLaunchKernelGGL((k_absdev<VEC, A, J, false, false>), grid, block, 0, st, x, g, stats, part2);                 \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_DEV);
#undef LAUNCH_DEV
    return launch_status();
}

int cnnq_pc_combine_dev(const double* part2, int G, int64_t C, const double* mom, int want_kurt, double* dev_out,
                        float* stats, void* stream) {
    if (!part2 || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || (!dev_out && !stats) || (stats && !mom))
        return CNNQ_EINVAL;
    const dim3 grid((unsigned)((C + TPB / 64 - 1) / (TPB / 64))), block(TPB);
    hipLaunchKernelGGL(k_combine_dev, grid, block, 0, (hipStream_t)stream, part2, G, (int)C, mom, want_kurt, dev_out,
                       stats);
    return launch_status();
}

int cnnq_pc_params(const float* stats, int64_t C, const cnnq_params_cfg* cfg, float* qp, float* diag,
                   void* stream) {
    if (!stats || !cfg || !qp || C <= 0 || C >= ((int64_t)1 << 31)) return CNNQ_EINVAL;
    if (cfg->num_bits < 1 || cfg->num_bits > 8 || cfg->clip < 0 || cfg->clip > 3) return CNNQ_EINVAL;
    if (cfg->bit_alloc && cfg->num_bits <= 4 && !diag) return CNNQ_EINVAL;  // bit table lives in diag
    float* bits_ws = diag ? diag + (size_t)CNNQ_DIAG_BITS * C : nullptr;
    hipLaunchKernelGGL(k_params, dim3(1), dim3(PTPB), 0, (hipStream_t)stream, stats, (int)C, *cfg, qp, diag,
                       bits_ws);
    return launch_status();
}

int cnnq_pc_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* qp, uint8_t* codes,
                uint64_t* hist, int reverse, void* stream) {
    if (!x || !y || !qp) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    // the histogram variant zeroes and flushes an LDS table per workgroup: keep its workgroups long
    const int rc = plan(N, C, HW, al16(x) && al16(y) && (!codes || ((uintptr_t)codes & 3) == 0), reverse ? 1 : 0, &v,
                        &g, /*fine=*/hist ? 0 : 1);
    if (rc) return rc;
    return launch_qdq(x, y, g, v, qp, codes, reinterpret_cast<unsigned long long*>(hist), (hipStream_t)stream);
}

int cnnq_pc_quantize_pack4(const float* x, uint8_t* packed, int64_t N, int64_t C, int64_t HW, const float* qp,
                           void* stream) {
    if (!x || !packed || !qp) return CNNQ_EINVAL;
    if (HW % 4 != 0 || !al16(x) || ((uintptr_t)packed & 1)) return CNNQ_EINVAL;   // whole float4s per channel row
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, true, 0, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    if (v.J == 4) hipLaunchKernelGGL((k_q_pack4<4>), grid, block, 0, st, x, packed, g, qp);
    else if (v.J == 2) hipLaunchKernelGGL((k_q_pack4<2>), grid, block, 0, st, x, packed, g, qp);
    else hipLaunchKernelGGL((k_q_pack4<1>), grid, block, 0, st, x, packed, g, qp);
    return launch_status();
}

int cnnq_pc_dequantize_pack4(const uint8_t* packed, float* y, int64_t N, int64_t C, int64_t HW, const float* qp,
                             void* stream) {
    if (!packed || !y || !qp) return CNNQ_EINVAL;
    if (HW % 4 != 0 || !al16(y) || ((uintptr_t)packed & 1)) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, true, 0, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    if (v.J == 4) hipLaunchKernelGGL((k_unpack4_dq<4>), grid, block, 0, st, packed, y, g, qp);
    else if (v.J == 2) hipLaunchKernelGGL((k_unpack4_dq<2>), grid, block, 0, st, packed, y, g, qp);
    else hipLaunchKernelGGL((k_unpack4_dq<1>), grid, block, 0, st, packed, y, g, qp);
    return launch_status();
}

int cnnq_pc_minmax(const float* x, int64_t N, int64_t C, int64_t HW, float* pmm, void* stream) {
    if (!x || !pmm) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x), 0, &v, &g);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    const bool ntl = N * C * HW * 4 > NT_BYTES;
#define LAUNCH_MM(VEC, A, J)                                                                         \
    do {                                                                                             \
        if (ntl) hipLaunchKernelGGL((k_minmax<VEC, A, J, true>), grid, block, 0, st, x, g, pmm);       \
        else hipLaunchKernelGGL((k_minmax<VEC, A, J, false>), grid, block, 0, st, x, g, pmm);          \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_MM);
#undef LAUNCH_MM
    return launch_status();
}

int cnnq_pc_minmax_reduce(const float* pmm, int G, int64_t C, float* out, void* stream) {
    if (!pmm || !out || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31)) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_minmax_reduce, dim3((unsigned)((C + TPB / 64 - 1) / (TPB / 64))), dim3(TPB), 0,
                       (hipStream_t)stream, pmm, G, (int)C, out);
    return launch_status();
}

int cnnq_pc_minmax_params(const float* pmm, int G, int64_t C, int num_bits, int positive, float* qp, void* stream) {
    if (!pmm || !qp || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || num_bits < 1 || num_bits > 8)
        return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_minmax_params, dim3((unsigned)((C + TPB / 64 - 1) / (TPB / 64))), dim3(TPB), 0,
                       (hipStream_t)stream, pmm, G, (int)C, num_bits, positive ? 1 : 0, qp);
    return launch_status();
}

int cnnq_pc_minmax_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, int num_bits, int positive,
                       float* pmm, float* qp, uint8_t* codes, uint64_t* hist, void* stream) {
    if (!x || !y || !pmm || !qp || num_bits < 1 || num_bits > 8) return CNNQ_EINVAL;
    const int G = cnnq_pc_groups(N, C, HW, al16(x) ? 1 : 0);
    if (G <= 0) return G ? G : CNNQ_EINVAL;
    int rc = cnnq_pc_minmax(x, N, C, HW, pmm, stream);
    if (rc) return rc;
    rc = cnnq_pc_minmax_params(pmm, G, C, num_bits, positive, qp, stream);
    if (rc) return rc;
    // descending address order: what the statistics pass read last is re-read first
    return cnnq_pc_qdq(x, y, N, C, HW, qp, codes, hist, /*reverse=*/1, stream);
}

int cnnq_pc_weight_correct(float* wq, int64_t C, int64_t HW, const float* stats_w, const float* stats_q, int vcorr,
                           int bcorr, void* stream) {
    if (!wq || !stats_w || !stats_q || C <= 0 || HW <= 0 || C > 65535 * 1024 || HW >= ((int64_t)1 << 31))
        return CNNQ_EINVAL;
    if (C > 65535) return CNNQ_ERANGE;
    int64_t bx = (HW + TPB - 1) / TPB;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_weight_correct, dim3((unsigned)bx, (unsigned)C), dim3(TPB), 0, (hipStream_t)stream, wq, (int)C,
                       (int)HW, stats_w, stats_q, vcorr, bcorr);
    return launch_status();
}

int cnnq_pc_bcorr_sums(const float* x, const float* y, int64_t N, int64_t C, int64_t HW, int relu_first,
                       double* part3, void* stream) {
    if (!x || !y || !part3) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x) && al16(y), 0, &v, &g);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BS(VEC, A, J) hipLaunchKernelGGL((k_bcorr_sums<VEC, A, J>), grid, block, 0, st, x, y, g, relu_first, part3)
    CNNQ_DISPATCH(v, LAUNCH_BS);
#undef LAUNCH_BS
    return launch_status();
}

int cnnq_pc_bcorr_bias(const double* part3, int G, int64_t C, double* sums, float* bias, void* stream) {
    if (!part3 || G <= 0 || C <= 0 || C >= ((int64_t)1 << 31) || (!sums && !bias)) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_bcorr_bias, dim3((unsigned)((C + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, part3, G,
                       (int)C, sums, bias);
    return launch_status();
}

int cnnq_pc_bcorr_apply(float* y, int64_t N, int64_t C, int64_t HW, const float* bias, void* stream) {
    if (!y || !bias) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(y), 0, &v, &g, /*fine=*/1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BA(VEC, A, J) hipLaunchKernelGGL((k_bcorr_apply<VEC, A, J>), grid, block, 0, st, y, g, bias)
    CNNQ_DISPATCH(v, LAUNCH_BA);
#undef LAUNCH_BA
    return launch_status();
}

int cnnq_pc_midtread_params(const float* stats, int64_t C, double target, int clip, int sym, const double* tables,
                            int ntab, float* mt, void* stream) {
    if (!stats || !mt || !tables || ntab < 2 || C <= 0 || C >= ((int64_t)1 << 31)) return CNNQ_EINVAL;
    const MtCfg cfg{target, clip ? 1 : 0, sym ? 1 : 0};
    hipLaunchKernelGGL(k_mt_params, dim3(1), dim3(PTPB), 0, (hipStream_t)stream, stats, (int)C, cfg, tables, ntab, mt);
    return launch_status();
}

int cnnq_pc_midtread_qdq(const float* x, float* y, int64_t N, int64_t C, int64_t HW, const float* mt, int clip,
                         float* codes, uint64_t* hist, void* stream) {
    if (!x || !y || !mt) return CNNQ_EINVAL;
    Variant v;
    Geo g;
    const int rc = plan(N, C, HW, al16(x) && al16(y) && (!codes || al16(codes)), 0, &v, &g, /*fine=*/hist ? 0 : 1);
    if (rc) return rc;
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* h = reinterpret_cast<unsigned long long*>(hist);
#define LAUNCH_MT3(VEC, A, J, CL, HI)                                                                            \
    do {                                                                                                        \
        if (codes) hipLaunchKernelGGL((k_mt_qdq<VEC, A, J, CL, HI, true>), grid, block, 0, st, x, y, g, mt, codes, h); \
        else hipLaunchKernelGGL((k_mt_qdq<VEC, A, J, CL, HI, false>), grid, block, 0, st, x, y, g, mt, codes, h);      \
    } while (0)
#define LAUNCH_MT(VEC, A, J)                                      \
    do {                                                          \
        if (clip && h) LAUNCH_MT3(VEC, A, J, true, true);         \
        else if (clip) LAUNCH_MT3(VEC, A, J, true, false);        \
        else if (h) LAUNCH_MT3(VEC, A, J, false, true);           \
        else LAUNCH_MT3(VEC, A, J, false, false);                 \
    } while (0)
    CNNQ_DISPATCH(v, LAUNCH_MT);
#undef LAUNCH_MT
#undef LAUNCH_MT3
    return launch_status();
}

int cnnq_midtread_entropy(const uint64_t* hist, const float* mt, int64_t C, int64_t total, float* out, void* stream) {
    if (!hist || !mt || !out || C <= 0 || total <= 0) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_mt_entropy, dim3(1), dim3(PTPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(hist), mt, (int)C, (double)total, out);
    return launch_status();
}

int cnnq_entropy(const uint64_t* hist, int nbins, float* out, void* stream) {
    if (!hist || !out || nbins <= 0) return CNNQ_EINVAL;
    hipLaunchKernelGGL(k_entropy, dim3(1), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(hist), nbins, out);
    return launch_status();
}

int cnnq_pt_setup(const float* range_offset_host, const float* stats, int64_t stats_stride, int rows, int rows_mode,
                  int zero_min, int num_bits, int int_exp, int enforce_true_zero, float* ptp, void* stream) {
    if (!ptp || num_bits < 1 || num_bits > 31) return CNNQ_EINVAL;
    if (!range_offset_host && (!stats || rows <= 0 || stats_stride < rows)) return CNNQ_EINVAL;
    const float hr = range_offset_host ? range_offset_host[0] : 0.f;
    const float ho = range_offset_host ? range_offset_host[1] : 0.f;
    hipLaunchKernelGGL(k_pt_setup, dim3(1), dim3(64), 0, (hipStream_t)stream, range_offset_host ? 1 : 0, hr, ho,
                       stats, stats_stride, rows, rows_mode, zero_min, num_bits, int_exp, enforce_true_zero, ptp);
    return launch_status();
}

int cnnq_pt_qdq(const float* x, float* y, int64_t n, const float* ptp, const float* noise, void* stream) {
    if (!x || !y || !ptp || n <= 0) return CNNQ_EINVAL;
    const bool vec = al16(x) && al16(y) && (!noise || al16(noise));
    const int64_t work = vec ? (n + 3) / 4 : n;
    const int64_t blocks = (work + TPB - 1) / TPB;
    if (blocks >= ((int64_t)1 << 31)) return CNNQ_ERANGE;
    const dim3 grid((unsigned)blocks), block(TPB);
    hipStream_t st = (hipStream_t)stream;
    if (vec) {
        if (noise) hipLaunchKernelGGL((k_pt_qdq<4, true>), grid, block, 0, st, x, y, n, ptp, noise);
        else hipLaunchKernelGGL((k_pt_qdq<4, false>), grid, block, 0, st, x, y, n, ptp, noise);
    } else {
        if (noise) hipLaunchKernelGGL((k_pt_qdq<1, true>), grid, block, 0, st, x, y, n, ptp, noise);
        else hipLaunchKernelGGL((k_pt_qdq<1, false>), grid, block, 0, st, x, y, n, ptp, noise);
    }
    return launch_status();
}

int cnnq_kld_hist(const float* x, int64_t rows, int64_t len, const float* rowmm, uint32_t* hist, void* stream) {
    if (!x || !rowmm || !hist || rows <= 0 || len <= 0) return CNNQ_EINVAL;
    if (rows > 65535 || len >= ((int64_t)1 << 31)) return CNNQ_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(hist, 0, (size_t)rows * KB * sizeof(uint32_t), st) != hipSuccess) return launch_status();
    const dim3 grid((unsigned)((len + KCHUNK - 1) / KCHUNK), (unsigned)rows), block(TPB);
    if (al16(x) && len % 4 == 0)
        hipLaunchKernelGGL((k_kld_hist<4>), grid, block, 0, st, x, len, rowmm, (int)rows, hist);
    else
        hipLaunchKernelGGL((k_kld_hist<1>), grid, block, 0, st, x, len, rowmm, (int)rows, hist);
    return launch_status();
}

int cnnq_kld_search(const uint32_t* hist, int64_t rows, const float* rowmm, double* div, double* out, void* stream) {
    if (!hist || !rowmm || !div || !out || rows <= 0) return CNNQ_EINVAL;
    if (rows > 65535) return CNNQ_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_kld_search, dim3(KC, (unsigned)rows), dim3(TPB), 0, st, hist, div);
    hipLaunchKernelGGL(k_kld_pick, dim3((unsigned)rows), dim3(64), 0, st, div, rowmm, (int)rows, out);
    return launch_status();
}

}  // extern "C"
