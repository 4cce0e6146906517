// cnnq_pertensor.hip.h - per-tensor GEMMLOWP path (the replacement of kernels/gemmlowp.cu).
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_group.hip.h"   // grp_arrive_last: the two-level arrival counters

namespace {

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
        } else {  // whole tensor; torch.min / torch.max propagate NaN (iq.py:527-528), so do the merges
            mn = INFINITY; mx = -INFINITY;
            for (int r = lane; r < rows; r += 64) { mn = pmin(mn, vmin[r]); mx = pmax(mx, vmax[r]); }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { mn = pmin(mn, shfl_xor_f(mn, m)); mx = pmax(mx, shfl_xor_f(mx, m)); }
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

// ------------------------------------------------------------------------------------------------------------------
// config 1 in ONE launch (round 3): dynamic per-sample min / max -> range / offset (iq.py:361-379) -> GEMMLOWP Q/DQ
// (kernels/gemmlowp.cu:8-45).  The chain is four launches (per-row min/max, merge, k_pt_setup, k_pt_qdq): 54 us for
// the [32,64,112,112] tensor of BASELINE config 1.  Here one grid of co-resident workgroups makes two sweeps:
//   sweep 1  a workgroup owns a contiguous run of 16 KB tiles (each inside one row); plain loads (the tensor - 103 MB -
//            stays in the 256 MB Infinity Cache for sweep 2); ONE write-once record per tile {key(max), key(min), nan};
//   meeting  the two-level arrival counters of the group kernels (grp_arrive_last: <= 16 read-modify-writes per
//            word).  The LAST workgroup to arrive folds the tile records into one record per row and advances the
//            epoch word; the others poll it (every workgroup read the epoch when it started: nobody can have advanced
//            it before all arrived, so "epoch changed" is "rows ready", nothing needs re-arming and a HIP graph may
//            replay the launch).  The wait is bounded (20 ms); on expiry a workgroup rebuilds the row records from x
//            itself - the same values - and raises bit 0 of the status word (policy of the group kernels: forward
//            progress needs the grid co-resident, correctness does not);
//   sweep 2  every workgroup derives the parameters from the R row records with the arithmetic (and the summation
//            order) of k_pt_setup, then revisits its tiles: non-temporal load (served by the cache), ptq1, store.
// Same bits as the chain: min / max are exact, the batch mean is the same sequence of fp64 additions.
// Two earlier forms, measured on the [32,64,112,112] tensor (chain: 54 us): atomic max of every tile into 32 row
// words of the fine-grained workspace - 157 us (~200 serialised read-modify-writes per word); tiles claimed from a
// ticket counter with a count of finished tiles as the meeting - 93 us (a thousand workgroups on one ticket word:
// read-modify-writes on fine-grained memory run at the fabric, not in an L2).  Control flow: every loop condition is
// a scalar - a `for (;;) ... break` form made the compiler wrap barriers in exec-masked loops, and a workgroup whose
// waves disagree on a barrier never leaves it.
constexpr int PTF_OFF = 256;             // byte offset of the region in the exchange workspace's header
constexpr int PTF_MAX_ROWS = 1024;
constexpr int PTF_TILE4 = TPB * 4;       // float4 per tile (16 KB)
constexpr int PTF_STEP = 4;              // tiles in flight per workgroup (16 loads per lane)

struct PtfWs {
    unsigned* status;
    unsigned* cnt;       // counter lines of the exchange workspace (zero between launches; grp_arrive_last re-arms them)
    unsigned* epoch;     // advanced by the folding workgroup once per launch
    unsigned* rows;      // [R][4]: key(max), key(min), nan, -
    uint4* recs;         // [tiles]: the same per tile (write-once; lives in the pair region of the workspace)
};

__device__ __forceinline__ unsigned f2key(float f) {          // order-preserving: a < b  <=>  key(a) < key(b)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ptp-equivalent parameters from the row records: wave 0 of the workgroup, the code of k_pt_setup
__device__ __forceinline__ void ptf_params(const PtfWs& w, int rows, int rows_mode, int zero_min, int num_bits, int int_exp,
                                           int etz, float* sh_p) {
    const int lane = threadIdx.x;
    if (lane >= 64) return;
    float mn, mx;
    if (rows_mode == 0) {
        double smn = 0., smx = 0.;
        for (int r = lane; r < rows; r += 64) {
            const unsigned kmx = __hip_atomic_load(w.rows + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned kmn = __hip_atomic_load(w.rows + 4 * r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool rn = __hip_atomic_load(w.rows + 4 * r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            smn += rn ? (double)NAN : (double)key2f(kmn);
            smx += rn ? (double)NAN : (double)key2f(kmx);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { smn += shfl_xor_d(smn, m); smx += shfl_xor_d(smx, m); }
        mn = (float)(smn / (double)rows);
        mx = (float)(smx / (double)rows);
    } else {
        mn = INFINITY; mx = -INFINITY;
        bool nan = false;
        for (int r = lane; r < rows; r += 64) {
            const unsigned kmx = __hip_atomic_load(w.rows + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned kmn = __hip_atomic_load(w.rows + 4 * r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nan |= __hip_atomic_load(w.rows + 4 * r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            mn = fminf(mn, key2f(kmn));
            mx = fmaxf(mx, key2f(kmx));
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, shfl_xor_f(mn, m)); mx = fmaxf(mx, shfl_xor_f(mx, m)); }
        if (__any(nan)) { mn = NAN; mx = NAN; }
    }
    if (zero_min) mn = 0.f;
    const float range = mx - mn;
    const float offset = mn;
    const bool ptz = etz && ((offset + range) > 0.f) && (offset < 0.f);
    if (lane != 0) return;
    const float qmax = (float)((1ll << num_bits) - 1);
    float scale = range / qmax;
    if (int_exp) scale = powf(2.f, (float)(int)ceilf(log2f(scale)));
    const float zero_point = roundf(-offset / scale);
    sh_p[0] = scale;
    sh_p[1] = ptz ? zero_point : -offset;
    sh_p[2] = qmax;
    sh_p[3] = ptz ? 1.f : 0.f;
    sh_p[4] = (range <= 0.f) ? 1.f : 0.f;
    sh_p[5] = range;
    sh_p[6] = offset;
    sh_p[7] = 0.f;
}

template <bool NTL>
__global__ void __launch_bounds__(TPB) k_pt_fused(const float* __restrict__ x, float* __restrict__ y, const int rows,
                                                  const unsigned L4, const unsigned tpr, const unsigned per, const PtfWs w,
                                                  const int rows_mode, const int zero_min, const int num_bits,
                                                  const int int_exp, const int etz, float* __restrict__ ptp_out) {
    __shared__ unsigned sh_t;
    __shared__ float l_mn[PTF_STEP][TPB / 64], l_mx[PTF_STEP][TPB / 64];
    __shared__ int l_nan[PTF_STEP][TPB / 64];
    __shared__ unsigned sh_row[PTF_MAX_ROWS][3];     // the folder's (and the cold path's) per-row table
    __shared__ float sh_p[8];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const unsigned ntiles = (unsigned)rows * tpr;
    const f4_t* x4 = reinterpret_cast<const f4_t*>(x);
    f4_t* y4 = reinterpret_cast<f4_t*>(y);
    // the epoch as this launch found it (uniform; read before this workgroup arrives, hence before anyone advances it)
    const unsigned epoch0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)__hip_atomic_load(w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned t0 = min((unsigned)blockIdx.x * per, ntiles), t1 = min(t0 + per, ntiles);     // this workgroup's tiles
    auto tile_base = [&](unsigned t, unsigned& left) -> size_t {
        const unsigned r = t / tpr, j = t - r * tpr;
        left = L4 - j * PTF_TILE4;                  // float4 of the row from the tile's start
        return (size_t)r * L4 + (size_t)j * PTF_TILE4;
    };

    // ---- sweep 1
    for (unsigned ts = t0; ts < t1; ts += PTF_STEP) {
        f4_t v[PTF_STEP][4];
#pragma unroll
        for (int k = 0; k < PTF_STEP; ++k) {
            unsigned left;
            const size_t base = tile_base(min(ts + k, t1 - 1u), left);       // past the run: repeat its last tile
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned o = (unsigned)q * TPB + (unsigned)tid;
                const unsigned oc = o < left ? o : 0u;     // past the row: re-read the tile's first element
                v[k][q] = NTL ? __builtin_nontemporal_load(x4 + base + oc) : x4[base + oc];
            }
        }
#pragma unroll
        for (int k = 0; k < PTF_STEP; ++k) {
            float mn = INFINITY, mx = -INFINITY;
            bool nan = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mn = fminf(fminf(mn, fminf(v[k][q].x, v[k][q].y)), fminf(v[k][q].z, v[k][q].w));
                mx = fmaxf(fmaxf(mx, fmaxf(v[k][q].x, v[k][q].y)), fmaxf(v[k][q].z, v[k][q].w));
                nan |= __builtin_isunordered(v[k][q].x, v[k][q].y) | __builtin_isunordered(v[k][q].z, v[k][q].w);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, shfl_xor_f(mn, m)); mx = fmaxf(mx, shfl_xor_f(mx, m)); }
            const bool wnan = __any(nan);
            if (lane == 0) { l_mn[k][wv] = mn; l_mx[k][wv] = mx; l_nan[k][wv] = wnan ? 1 : 0; }
        }
        __syncthreads();
        if (tid < PTF_STEP && ts + (unsigned)tid < t1) {     // wave 0: one record per tile of the step, write-through
            const float a = fminf(fminf(l_mn[tid][0], l_mn[tid][1]), fminf(l_mn[tid][2], l_mn[tid][3]));
            const float b = fmaxf(fmaxf(l_mx[tid][0], l_mx[tid][1]), fmaxf(l_mx[tid][2], l_mx[tid][3]));
            const unsigned nn = (unsigned)(l_nan[tid][0] | l_nan[tid][1] | l_nan[tid][2] | l_nan[tid][3]);
            unsigned* rec = reinterpret_cast<unsigned*>(w.recs + (ts + (unsigned)tid));
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(rec),
                               (unsigned long long)f2key(b) | ((unsigned long long)f2key(a) << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rec + 2, nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // ---- the meeting: arrive (the records have left the CU), the last arriver folds, the others wait for the epoch
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh_t = grp_arrive_last(w.cnt, (int)blockIdx.x, (int)gridDim.x) ? 1u : 0u;
    }
    __syncthreads();
    const int folder = __builtin_amdgcn_readfirstlane((int)sh_t);
    __syncthreads();
    int timed_out = 0;
    if (!folder) {
        if (tid == 0) {
            long long c0 = 0;
            int to = 0;
            for (int spins = 0;; ++spins) {
                if (__hip_atomic_load(w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch0) break;
                __builtin_amdgcn_s_sleep(16);
                if ((spins & 31) == 31) {
                    const long long now = wall_clock64();
                    if (c0 == 0) c0 = now;
                    if (now - c0 > GRP_TIMEOUT_TICKS) { to = 1; break; }
                }
            }
            if (to) atomicOr(w.status, 1u);
            sh_t = (unsigned)to;
        }
        __syncthreads();
        timed_out = __builtin_amdgcn_readfirstlane((int)sh_t);
        __syncthreads();
    }
    if (folder || timed_out) {
        for (int r = tid; r < rows; r += TPB) { sh_row[r][0] = 0u; sh_row[r][1] = 0xffffffffu; sh_row[r][2] = 0u; }
        __syncthreads();
        if (folder) {
            // every tile record is in: fold them per row (LDS atomics on the keys).  Eight 16-byte loads in flight per
            // lane: a read of the fine-grained workspace is a microsecond, one dependent read per record was 2/3 of
            // the kernel.  Plain loads: the records are write-once and the arrival counters order them before this.
            for (unsigned tb = (unsigned)tid; tb < ntiles; tb += TPB * 8) {
                uint4 rec[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned t = tb + (unsigned)k * TPB;
                    rec[k] = w.recs[t < ntiles ? t : tb];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned t = tb + (unsigned)k * TPB;
                    if (t < ntiles) {
                        const unsigned r = t / tpr;
                        atomicMax(&sh_row[r][0], rec[k].x);
                        atomicMin(&sh_row[r][1], rec[k].y);
                        if (rec[k].z) atomicOr(&sh_row[r][2], 1u);
                    }
                }
            }
        } else {
            // cold path: the rows' extrema from x itself (the values the folder writes)
            for (int r = 0; r < rows; ++r) {
                float mn = INFINITY, mx = -INFINITY;
                bool nan = false;
                for (unsigned o = (unsigned)tid; o < L4; o += TPB) {
                    const f4_t v = x4[(size_t)r * L4 + o];
                    mn = fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w));
                    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                    nan |= __builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w);
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, shfl_xor_f(mn, m)); mx = fmaxf(mx, shfl_xor_f(mx, m)); }
                const bool wnan = __any(nan);
                if (lane == 0) {
                    atomicMax(&sh_row[r][0], f2key(mx));
                    atomicMin(&sh_row[r][1], f2key(mn));
                    if (wnan) atomicOr(&sh_row[r][2], 1u);
                }
            }
        }
        __syncthreads();
        for (int r = tid; r < rows; r += TPB) {
            __hip_atomic_store(w.rows + 4 * r, sh_row[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w.rows + 4 * r + 1, sh_row[r][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w.rows + 4 * r + 2, sh_row[r][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: the rows have left the CU
        __syncthreads();
        if (folder && tid == 0) __hip_atomic_store(w.epoch, epoch0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ptf_params(w, rows, rows_mode, zero_min, num_bits, int_exp, etz, sh_p);
    __syncthreads();
    const float scale = sh_p[0], shift = sh_p[1], qmax = sh_p[2];
    const bool tz = sh_p[3] != 0.f, pass = sh_p[4] != 0.f;
    if (blockIdx.x == 0 && tid < 8 && ptp_out) ptp_out[tid] = sh_p[tid];

    // ---- sweep 2: the same tiles, last first (what sweep 1 touched last is the likeliest to be cached)
    for (unsigned done = 0; t0 + done < t1; done += PTF_STEP) {
        const unsigned hi = t1 - done;                       // tiles [hi - PTF_STEP, hi) of the run, clipped at t0
        f4_t v[PTF_STEP][4];
#pragma unroll
        for (int k = 0; k < PTF_STEP; ++k) {
            unsigned left;
            const unsigned t = hi >= t0 + 1u + (unsigned)k ? hi - 1u - (unsigned)k : t0;
            const size_t base = tile_base(t, left);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned o = (unsigned)q * TPB + (unsigned)tid;
                v[k][q] = __builtin_nontemporal_load(x4 + base + (o < left ? o : 0u));
            }
        }
#pragma unroll
        for (int k = 0; k < PTF_STEP; ++k) {
            unsigned left;
            const bool live = hi >= t0 + 1u + (unsigned)k;
            const unsigned t = live ? hi - 1u - (unsigned)k : t0;
            const size_t base = tile_base(t, left);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned o = (unsigned)q * TPB + (unsigned)tid;
                f4_t r4;
                r4.x = pass ? v[k][q].x : ptq1(v[k][q].x, scale, shift, qmax, tz, 0.f);
                r4.y = pass ? v[k][q].y : ptq1(v[k][q].y, scale, shift, qmax, tz, 0.f);
                r4.z = pass ? v[k][q].z : ptq1(v[k][q].z, scale, shift, qmax, tz, 0.f);
                r4.w = pass ? v[k][q].w : ptq1(v[k][q].w, scale, shift, qmax, tz, 0.f);
                if (live && o < left) __builtin_nontemporal_store(r4, y4 + base + o);
            }
        }
    }
}

}  // namespace
