// cnnq_kld.hip.h - KLD calibration: per-sample histogram and threshold search.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"

namespace {

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
