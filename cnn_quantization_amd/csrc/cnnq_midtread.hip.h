// cnnq_midtread.hip.h - mid-tread quantization with per-channel bin allocation and its entropy.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_params.hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// mid-tread quantization with per-channel bin allocation (config 5, iq.py:128-225)
// ------------------------------------------------------------------------------------------
struct MtCfg {
    double target;  // bits; bins per channel on average = 2^target
    int clip;       // 1: laplace-prior clipping around the mean (activations), 0: min/max range (weights)
    int sym;        // 0: non-negative range (force_positive / half_range)
};

constexpr int MT_NB = CNNQ_MT_HIST_BINS;  // integer-code bins, codes -MT_NB/2 .. MT_NB/2-1
constexpr int MT_W = CNNQ_MT_HIST_WINDOW; // codes of the histogram window [wstart, wstart + MT_W): LDS bins, replica bins
constexpr int MT_GR = CNNQ_MT_HIST_REPLICAS;
constexpr int MT_REP = 8;                 // LDS copies of a window bin (by lane)
// last word of the histogram: non-zero when any count went to the global bins [0, MT_NB + 2) (a code outside the window)
__host__ __device__ constexpr size_t mt_flag_word(int C) { return (size_t)MT_NB + 2 + 2 * (size_t)C + (size_t)MT_GR * MT_W; }

// eq. 10 (iq.py:128-135) for one channel: omega = round(C * 2^target * sigma^(2/3) / sum sigma^(2/3)), and the clipping
// multiplier of that many bins by linear interpolation in the (omega, alpha) table, fp64 like numpy (iq.py:137-145)
__device__ __forceinline__ void mt_omega_alpha(const MtCfg& cfg, float B, float psum, float vstd, const double* __restrict__ otab,
                                               const double* __restrict__ atab, int ntab, float& omega, float& am) {
    const float p = powf(vstd, (float)(2. / 3));
    omega = rintf((B * p) / psum);
    am = 0.f;
    if (cfg.clip) {
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
    }
}

// step size and clamp bounds of one channel (iq.py:193-214) - the single definition of this arithmetic: k_mt_params runs it
// per table row, the single-launch kernels (cnnq_aciq.hip.h) in every workgroup that holds a piece of the channel
struct MtChan {
    float delta, cmin, cmax;
};
__device__ __forceinline__ MtChan mt_channel(const MtCfg& cfg, float omega, float am, float vmin, float vmax, float mu, float vb) {
    MtChan r;
    const float mu0 = fmaxf(mu, 0.f);
    float rng;
    if (cfg.clip) rng = cfg.sym ? (2.f * am) * vb : mu0 + am * vb;
    else rng = cfg.sym ? vmax - vmin : vmax;
    r.delta = (omega > 0.f) ? rng / omega : 3.402823466e+38f;
    r.cmin = -INFINITY;
    r.cmax = INFINITY;
    if (cfg.clip) {
        const float muq = (cfg.sym ? mu : mu0) / r.delta;
        r.cmax = muq + (cfg.sym ? omega / 2.f : omega);
        r.cmin = cfg.sym ? muq - omega / 2.f : 0.f;
    }
    return r;
}

// GUESS: before pass B nothing but the std is known, so the single-launch form cannot compute the smallest clamp bound the
// histogram window starts at - it takes the bound a Laplace channel (b = std / sqrt 2) would have.  Exact (0) for the
// non-negative range; for the symmetric range only the speed of the histogram depends on it (codes outside the window are
// counted in the global bins, and the flag word tells k_mt_entropy to read them).
template <bool GUESS>
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
    double psum_d = 0.;
    for (int c = tid; c < C; c += PTPB) psum_d += (double)powf(vstd[c], (float)(2. / 3));
    const float psum = (float)block_sum(psum_d, sh);
    const float B = (float)((double)C * pow(2., cfg.target));
    float wmin = 1e9f;
    for (int c = tid; c < C; c += PTPB) {
        float omega, am;
        mt_omega_alpha(cfg, B, psum, vstd[c], otab, atab, ntab, omega, am);
        const MtChan r = mt_channel(cfg, omega, am, vmin[c], vmax[c], vmean[c], GUESS ? vstd[c] * 0.70710678f : vb[c]);
        if constexpr (!GUESS) {
            mt[(size_t)CNNQ_MT_DELTA * C + c] = r.delta;
            mt[(size_t)CNNQ_MT_CMIN * C + c] = r.cmin;
            mt[(size_t)CNNQ_MT_CMAX * C + c] = r.cmax;
        }
        mt[(size_t)CNNQ_MT_OMEGA * C + c] = omega;
        mt[(size_t)CNNQ_MT_ALPHA * C + c] = am;
        if (cfg.clip) wmin = fminf(wmin, floorf(fminf(fmaxf(r.cmin, -1e9f), 1e9f)));
    }
    // first code of the histogram window: the smallest clamp bound of the tensor (codes are >= c_min); without
    // clipping the window is centred on zero
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wmin = fminf(wmin, shfl_xor_f(wmin, m));
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = (double)wmin;
    __syncthreads();
    float w0 = 1e9f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) w0 = fminf(w0, (float)sh[i]);
    const float wstart = cfg.clip ? fmaxf(w0, (float)(-MT_NB / 2)) : (float)(-MT_W / 2);
    for (int c = tid; c < C; c += PTPB) mt[(size_t)CNNQ_MT_WSTART * C + c] = wstart;
}

// hist layout (uint64): [0, MT_NB) integer codes -MT_NB/2.., [MT_NB] below range, [MT_NB+1] above range, C counts of
// "clamped to a non-integer c_min[c]" and C of "... c_max[c]", then MT_GR replicas of the MT_W window bins
// (code = wstart + i; k_mt_entropy adds them to the integer bins).
//
// Where the histogram's cost was (measured, VGG-16 b512): not in the update - with the LDS atomics compiled out, or
// the branchy update made branch-free, the kernel ran the same 1190 us per large layer against 950 us without
// histogram - but in the launch shape the per-workgroup table forced: a 22 KB LDS table (512 codes, 8-32 copies)
// zeroed and flushed per workgroup, and ~30 global atomics per flush on the SAME few addresses, were only affordable
// with 4096 long-lived workgroups, and long-lived workgroups stream x -> y at 5.4 instead of 6.1 TB/s (a persistent
// loop over short tiles was slower still: 19.5 vs 18.0 ms even without histogram).  Now the histogram variant runs on
// the same short tiles as the plain one: a 4 KB LDS table (MT_W codes x 8 copies by lane; zero is counted in a
// register) and a flush of one atomic per live bin into one of MT_GR replica tables (by workgroup id), which keeps
// the same-address chains short.  Codes outside the window go to the global bins directly (correct, slow, rare: the
// window starts at the tensor's smallest clamp bound and a channel has ~2^target bins).
// The kernel walks tiles id = blockIdx.x, blockIdx.x + gridDim.x, ... of `total` (launched with one workgroup per tile).
template <int VEC, int A, int J, bool CLIP, bool HIST, bool CODES>
__global__ void __launch_bounds__(TPB) k_mt_qdq(const float* __restrict__ x, float* __restrict__ y, const Geo g,
                                                const float* __restrict__ mt, float* __restrict__ codes,
                                                unsigned long long* __restrict__ hist, const int total) {
    constexpr int MT_WORDS = MT_W * MT_REP;
    auto hidx = [](unsigned kk, int tid) -> unsigned { return kk * MT_REP + (unsigned)(tid & (MT_REP - 1)); };
    __shared__ float sh_d[MAXCH], sh_lo[MAXCH], sh_hi[MAXCH];
    __shared__ unsigned sh_hist[HIST ? MT_WORDS : 1];
    __shared__ unsigned sh_clo[HIST ? MAXCH : 1], sh_chi[HIST ? MAXCH : 1];
    const int tid = threadIdx.x;
    if constexpr (HIST) {
        for (int i = tid; i < MT_WORDS; i += TPB) sh_hist[i] = 0u;
        for (int i = tid; i < MAXCH; i += TPB) { sh_clo[i] = 0u; sh_chi[i] = 0u; }
        __syncthreads();
    }
    const int wstart = HIST ? (int)mt[(size_t)CNNQ_MT_WSTART * g.C] : 0;
    unsigned nzero = 0;  // code 0 (the mode of the distribution) is counted in a register, see k_qdq
    for (int vb = (int)blockIdx.x; vb < total; vb += (int)gridDim.x) {
        const Blk b = blk_of_id<VEC>(g, vb, total);
        const int nch = b.c1 - b.c0;
        for (int i = tid; i < nch; i += TPB) {
            sh_d[i] = mt[(size_t)CNNQ_MT_DELTA * g.C + b.c0 + i];
            sh_lo[i] = mt[(size_t)CNNQ_MT_CMIN * g.C + b.c0 + i];
            sh_hi[i] = mt[(size_t)CNNQ_MT_CMAX * g.C + b.c0 + i];
        }
        __syncthreads();
        int col[J], chl[J][A];
        bool ok[J];
        float d[J][A], lo[J][A], hi[J][A];
        bool hi_ni[J][A], lo_ni[J][A];      // histogram: the clamp bound is not an integer code
        unsigned nhi[J][A], nlo[J][A];      // ... and how often this lane clamped to it in this tile
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
                if constexpr (HIST && CLIP) {
                    hi_ni[j][a] = hi[j][a] != rintf(hi[j][a]);   // a non-integer bound is a value of its own
                    lo_ni[j][a] = lo[j][a] != rintf(lo[j][a]);
                    nhi[j][a] = 0u;
                    nlo[j][a] = 0u;
                }
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
                            // Branch-free for everything common: zero and "clamped to a non-integer bound" are
                            // counted in registers (carry adds), an integer code inside the window is one LDS
                            // atomic under the lane mask.  The if / else-if chain this replaces executed all its
                            // arms on nearly every step (some lane always takes each): 2.2x the VALU and 3.3x the
                            // SALU instructions of the kernel without histogram.
                            const bool z = (t == 0.f);
                            const bool at_hi = CLIP && hi_ni[j][a] && t == hi[j][a];
                            const bool at_lo = CLIP && lo_ni[j][a] && t == lo[j][a] && !at_hi;   // (c_min == c_max, a constant channel: ONE value, counted once - round 6)
                            nzero += z ? 1u : 0u;
                            if constexpr (CLIP) { nhi[j][a] += at_hi ? 1u : 0u; nlo[j][a] += at_lo ? 1u : 0u; }
                            const int k = (int)t;                       // saturating; NaN -> 0
                            const unsigned kk = (unsigned)(k - wstart);
                            const bool fast = ((float)k == t) && kk < (unsigned)MT_W;
                            if (fast && !z) {
#ifdef MT_EXP_NOATOMIC
                                nzero += kk;
#else
                                atomicAdd(&sh_hist[hidx(kk, tid)], 1u);
#endif
                            } else if (!(z || at_hi || at_lo)) {        // rare
                                if (t == rintf(t)) {                    // integer code outside the window (or inf)
                                    if (t >= (float)(-MT_NB / 2) && t < (float)(MT_NB / 2)) atomicAdd(&hist[(int)t + MT_NB / 2], 1ull);
                                    else atomicAdd(&hist[t < 0.f ? MT_NB : MT_NB + 1], 1ull);
                                    atomicAdd(&hist[mt_flag_word(g.C)], 1ull);   // the global bins are in use
                                } else {
                                    atomicAdd(&sh_clo[chl[j][a]], 1u);  // NaN
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (HIST && CLIP) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    if (nhi[j][a]) atomicAdd(&sh_chi[chl[j][a]], nhi[j][a]);
                    if (nlo[j][a]) atomicAdd(&sh_clo[chl[j][a]], nlo[j][a]);
                }
        }
        __syncthreads();   // the tile's channel tables (and clamp counters) are complete / no longer read
        if constexpr (HIST) {
            // the clamp counters are indexed by the tile's channels: hand them over before the next tile
            for (int i = tid; i < nch; i += TPB) {
                if (sh_clo[i]) { atomicAdd(&hist[MT_NB + 2 + b.c0 + i], (unsigned long long)sh_clo[i]); sh_clo[i] = 0u; }
                if (sh_chi[i]) { atomicAdd(&hist[MT_NB + 2 + g.C + b.c0 + i], (unsigned long long)sh_chi[i]); sh_chi[i] = 0u; }
            }
        }
    }
    if constexpr (HIST) {
        if (nzero) {
            const int kk = -wstart;
            if (kk >= 0 && kk < MT_W) {
                atomicAdd(&sh_hist[hidx((unsigned)kk, tid)], nzero);
            } else {
                // code 0 lies outside the window (a symmetric range with more than 2 * MT_W bins): its count goes to the
                // global bins, and the flag word must say so - k_mt_entropy reads those bins only when it is raised
                atomicAdd(&hist[MT_NB / 2], (unsigned long long)nzero);
                atomicAdd(&hist[mt_flag_word(g.C)], 1ull);
            }
        }
        __syncthreads();
        unsigned long long* rep = hist + MT_NB + 2 + 2 * (size_t)g.C + (size_t)(blockIdx.x & (MT_GR - 1)) * MT_W;
        for (int i = tid; i < MT_W; i += TPB) {
            unsigned tot = 0;
#pragma unroll
            for (int r = 0; r < MT_REP; ++r) tot += sh_hist[hidx((unsigned)i, r + tid)];
            if (tot) {
                if (wstart + i < MT_NB / 2) {
                    atomicAdd(&rep[i], (unsigned long long)tot);
                } else {
                    atomicAdd(&hist[MT_NB + 1], (unsigned long long)tot);
                    atomicAdd(&hist[mt_flag_word(g.C)], 1ull);
                }
            }
        }
    }
}

// entropy over integer bins + the per-channel non-integer clamp values (equal values merged,
// as torch.unique would, utils/entropy.py:10)
// count_dev (may be null): the elements per CHANNEL of the tensor the codes were counted over, in device memory (row COUNT of a
// merged moment record: a batch-sharded run knows the global batch's size on the device, not on the host) - total = count * C
// up to MT_ENT_BATCH histograms in one launch (round 6: the tensors of a forward at its end), one workgroup each
constexpr int MT_ENT_BATCH = 16;
struct MtEntBatch {
    const unsigned long long* hist[MT_ENT_BATCH];
    const float* mt[MT_ENT_BATCH];
    double total[MT_ENT_BATCH];
    int C[MT_ENT_BATCH];
};
__device__ __forceinline__ void mt_entropy_one(const unsigned long long* __restrict__ hist, const float* __restrict__ mt, int C, double total,
                                               float* __restrict__ out, const double* __restrict__ count_dev);
__global__ void __launch_bounds__(PTPB) k_mt_entropy(const unsigned long long* __restrict__ hist,
                                                     const float* __restrict__ mt, int C, double total,
                                                     float* __restrict__ out, const double* __restrict__ count_dev = nullptr) {
    mt_entropy_one(hist, mt, C, total, out, count_dev);
}
__global__ void __launch_bounds__(PTPB) k_mt_entropy_batch(const MtEntBatch b, float* __restrict__ out) {
    const int i = blockIdx.x;
    mt_entropy_one(b.hist[i], b.mt[i], b.C[i], b.total[i], out + i, nullptr);
}
__device__ __forceinline__ void mt_entropy_one(const unsigned long long* __restrict__ hist, const float* __restrict__ mt, int C, double total,
                                               float* __restrict__ out, const double* __restrict__ count_dev) {
    __shared__ double sh[PTPB / 64];
    __shared__ unsigned long long lrep[MT_W];
    const int tid = threadIdx.x;
    const float ftotal = (float)(count_dev ? count_dev[0] * (double)C : total);
    double e = 0.;
    // the window bins live in MT_GR replica tables: fold them (bin i of the window is code wstart + i)
    const unsigned long long* rep = hist + MT_NB + 2 + 2 * (size_t)C;
    const int wbase = (int)mt[(size_t)CNNQ_MT_WSTART * C] + MT_NB / 2;
    {   // all PTPB threads: thread (g, i) adds every (PTPB / MT_W)-th replica of bin i, then MT_W threads add the partials
        __shared__ unsigned long long lpart[PTPB / MT_W][MT_W];
        const int i = tid % MT_W, g = tid / MT_W;
        unsigned long long t = 0;
        for (int r = g; r < MT_GR; r += PTPB / MT_W) t += rep[(size_t)r * MT_W + i];
        lpart[g][i] = t;
        __syncthreads();
        if (tid < MT_W) {
            unsigned long long a = 0;
            for (int q = 0; q < PTPB / MT_W; ++q) a += lpart[q][tid];
            lrep[tid] = a;
        }
        __syncthreads();
    }
    // 131074 bins through one workgroup: 16 independent loads in flight per thread (a dependent
    // load per iteration made this kernel 150 us of pure latency); each thread still adds its bins in
    // ascending order, so the sum is unchanged
    constexpr int UNR = 16;
    const bool global_bins = hist[mt_flag_word(C)] != 0ull;   // uniform
    if (!global_bins) {
        // every count sits in the window (the normal case): 128 bins instead of 131074
        for (int i = tid; i < MT_W; i += PTPB)
            if (lrep[i]) { const float pr = (float)lrep[i] / ftotal; e += (double)(-pr * log2f(pr)); }
    }
    for (int base = 0; global_bins && base < MT_NB + 2; base += PTPB * UNR) {
        unsigned long long c[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = base + u * PTPB + tid;
            c[u] = i < MT_NB + 2 ? hist[i] : 0ull;
            if ((unsigned)(i - wbase) < (unsigned)MT_W && i < MT_NB) c[u] += lrep[i - wbase];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (c[u]) { const float pr = (float)c[u] / ftotal; e += (double)(-pr * log2f(pr)); }
    }
    // non-integer clamp values: one histogram entry per (channel, bound); equal values are merged.
    // Entries with hits are compacted (in index order) into an LDS list, then every thread owning a
    // list entry walks the whole list with broadcast reads, branch-free: the entry represents its value
    // if no earlier list entry has the same value, and then carries the hits of all equal entries.
    // (2*C <= MT_ENT entries; beyond that a plain global-memory scan.)
    const unsigned long long* cl = hist + MT_NB + 2;   // 2*C clamp counters
    constexpr int MT_ENT = 4096;
    __shared__ float lv[MT_ENT];
    __shared__ unsigned long long lc[MT_ENT];
    __shared__ int wcnt[PTPB / 64];
    const int n2 = 2 * C;
    auto val = [&](int i) -> float {
        return mt[(size_t)(i < C ? CNNQ_MT_CMIN : CNNQ_MT_CMAX) * C + (i < C ? i : i - C)];
    };
    if (n2 <= MT_ENT) {
        const int wv = tid >> 6, lane = tid & 63;
        const int seg = ((n2 + PTPB - 1) / PTPB) * 64;        // entries per wave, whole 64-lane steps
        int mine = 0;
        for (int k = 0; k < seg; k += 64) {
            const int i = wv * seg + k + lane;
            mine += __popcll(__ballot(i < n2 && cl[i] != 0ull));
        }
        if (lane == 0) wcnt[wv] = mine;
        __syncthreads();
        int base = 0, nl = 0;
        for (int w = 0; w < PTPB / 64; ++w) {
            if (w < wv) base += wcnt[w];
            nl += wcnt[w];
        }
        for (int k = 0; k < seg; k += 64) {
            const int i = wv * seg + k + lane;
            const unsigned long long c = i < n2 ? cl[i] : 0ull;
            const unsigned long long mask = __ballot(c != 0ull);
            if (c) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                lv[pos] = val(i);
                lc[pos] = c;
            }
            base += __popcll(mask);
        }
        __syncthreads();
        for (int m0 = tid; m0 < nl; m0 += PTPB) {
            const float vi = lv[m0];
            unsigned long long cnt = 0;
            bool dup = false;
#pragma unroll 8
            for (int m = 0; m < nl; ++m) {
                const bool same = lv[m] == vi;
                cnt += same ? lc[m] : 0ull;
                dup |= same && m < m0;
            }
            if (!dup) {
                const float pr = (float)cnt / ftotal;
                e += (double)(-pr * log2f(pr));
            }
        }
    } else {
        for (int i = tid; i < n2; i += PTPB) {
            const unsigned long long ci = cl[i];
            if (!ci) continue;
            const float vi = val(i);
            bool dup = false;
            unsigned long long cnt = ci;
            for (int j = 0; j < n2; ++j) {
                const unsigned long long cj = cl[j];
                if (j == i || !cj) continue;
                if (val(j) == vi) { if (j < i) { dup = true; break; } cnt += cj; }
            }
            if (dup) continue;
            const float pr = (float)cnt / ftotal;
            e += (double)(-pr * log2f(pr));
        }
    }
    const double r = block_sum(e, sh);
    if (tid == 0) out[0] = (float)r;
}

}  // namespace
