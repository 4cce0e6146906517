// cnnq_common.hip.h - shared constants, the column-walk decomposition (Geo / Blk / blk_of), vector and non-temporal load/store helpers.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
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

// tile `id` of `total` (a launch with one workgroup per tile passes blockIdx.x / gridDim.x; persistent kernels walk
// id = blockIdx.x, blockIdx.x + gridDim.x, ...)
template <int VEC>
__device__ __forceinline__ Blk blk_of_id(const Geo& g, int id, int total) {
    Blk b;
    const int bid = g.rev ? total - 1 - id : id;
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
__device__ __forceinline__ Blk blk_of(const Geo& g) {
    return blk_of_id<VEC>(g, (int)blockIdx.x, (int)gridDim.x);
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
// Development knobs (kernel sweeps: tile heights, dispatch orders, ...) exist only in builds with -DCNNQ_DEV_KNOBS
// (tools/build_alt.sh knobs -DCNNQ_DEV_KNOBS): the shipped library reads them as their defaults, compile-time constants.
#ifdef CNNQ_DEV_KNOBS
inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
#else
constexpr int env_int(const char*, int dflt) { return dflt; }
#endif

constexpr int64_t NT_BYTES_DEFAULT = (int64_t)384 << 20;   // swept 0..1000 MB on the ResNet-50 set: flat optimum 250-400
// CNNQ_NT_BYTES - the ONE environment variable the shipped library reads (once per process; documented in
// include/cnnq_hip.h): the tensor size above which the read-only passes use non-temporal loads.  It is a property of the
// part's Infinity Cache, not of the path, hence tunable at deployment; 0 forces the non-temporal template instances on every
// tensor, which is also how the parity tests reach the instances that otherwise only the > 384 MB layers select.
inline int64_t nt_bytes() {
    static const int64_t v = [] {
        const char* e = getenv("CNNQ_NT_BYTES");
        return (e && *e) ? (int64_t)atoll(e) : NT_BYTES_DEFAULT;
    }();
    return v;
}
#define NT_BYTES nt_bytes()

template <int VEC>
__device__ __forceinline__ void ldv_nt(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const f4_t t = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = __builtin_nontemporal_load(p);
    }
}
// four floats from an address that is only 4-byte aligned (rows of H*W % 4 != 0 elements): still one global_load_dwordx4
typedef f4_t f4a4_t __attribute__((aligned(4)));
__device__ __forceinline__ void ldv4_nt_a4(const float* __restrict__ p, float (&v)[4]) {
    const f4_t t = __builtin_nontemporal_load(reinterpret_cast<const f4a4_t*>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void stv4_nt_a4(float* __restrict__ p, const float (&v)[4]) {
    f4_t t;
    t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
    __builtin_nontemporal_store(t, reinterpret_cast<f4a4_t*>(p));
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

// torch.min / torch.max propagate NaN (a NaN activation poisons its channel's range, iq.py:416,423); v_min/v_max
// return the other operand.  The hot loop keeps v_min/v_max plus one unordered-compare per two elements and
// poisons the lane's result afterwards; every merge above the lane uses these propagating forms.
__device__ __forceinline__ float pmin(float a, float b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ float pmax(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ double pmind(double a, double b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ double pmaxd(double a, double b) { return (a > b || a != a) ? a : b; }

// zero_point = round(qmin - offset / scale) with qmin = 0 (iq.py:570-572): +0 when the quotient is +-0.  The compiler
// otherwise folds rint(0 - q) into rint(-q), which is -0 for q = +0 (seen in k_minmax_params' ISA: `v_rndne_f32 -v`);
// the values agree, the bits of the parameter table would not.
__device__ __forceinline__ float zero_point_of(float offset, float scale) {
    float t = 0.f - offset / scale;
    asm volatile("" : "+v"(t));
    return rintf(t);
}

// qmax = 2.**num_bits - 1. (iq.py:559: a Python float, i.e. fp64, rounded to fp32 when it meets the tensor): exact up
// to 24 bits, 2^32 for 'int32' - any width the reference's __gemmlowpQuantize1__ accepts
__device__ __host__ __forceinline__ float qmax_of(int num_bits) { return (float)(exp2((double)num_bits) - 1.0); }

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }

// Instructions written out (round 6).  v_min3 / v_max3 / v_max on the RAW registers: fminf / fmaxf make the compiler canonicalise
// every loaded value first (a v_max v, v, v per element); for numbers the results are the same, a quiet NaN operand is dropped
// like fminf drops it, a signalling one comes out as a NaN that the next step drops - callers poison their extrema separately when
// a NaN was seen, as they did with fminf.  HIP treats inline asm as convergent: use these in fully unrolled tile loops only - a
// loop with a run-time trip count that contains one is no longer unrolled (k_moments' row loop keeps fminf).
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float relu_raw(float a) {      // fmaxf(a, 0.f): 0 for a NaN, as there
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
    return r;
}
// the two halves of a packed sum, one plain add (written out: left to itself the compiler packs the horizontal adds of two
// different sums into one v_pk_add_f32 behind three register moves)
template <class F2>
__device__ __forceinline__ float hadd(const F2 p) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(p.x), "v"(p.y));
    return r;
}
__device__ __forceinline__ float abs_add(float a, float b) {      // |a| + |b|, one instruction
    float r;
    asm("v_add_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One LDS-DMA load (16 bytes per lane: lane i of the wave lands at lds + 16 i) issued BEHIND the compiler's back.  With the
// builtin in flight the compiler waits for vmcnt(0) before the first use of ANY loaded register (it does not count across the
// two kinds of load), which serialises "the whole tile has landed" before the first addition.  Written as asm, issued AFTER the
// register loads of the tile: the loads return in order, so the compiler's own counts for the register steps - it believes
// fewer loads are in flight than there are - only ever wait longer than they must, never shorter; the reader of the LDS steps
// waits for vmcnt(0) explicitly.  base: wave-uniform; lds: wave-uniform.
// (m0 is a reserved register: the compiler sets it in front of each of its own uses; naming it in the clobber list documents the
//  write and draws -Winline-asm)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16_behind(const char* base, unsigned off, const float* lds) {
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds);
    const unsigned long long b = (unsigned long long)base;
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    const unsigned long long bs = (unsigned long long)blo | ((unsigned long long)bhi << 32);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(bs), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop
// ... and the wait of their reader: the LDS-DMA steps of this wave have landed (its lanes read only their own wave's slots).  `dep`:
// a value the register steps produced - the wait is volatile, but arithmetic on registers may be scheduled across a volatile asm,
// and a wait hoisted above the register steps would make all of them wait for the whole tile again
template <class T>
__device__ __forceinline__ void lds_dma_landed(T& dep) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dep) : : "memory");
}

}  // namespace
