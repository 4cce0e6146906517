// cnnq_plan.hip.h - host side: load-shape choice, launch geometry, template dispatch.
// Part of the single translation unit cnnq_kernels.hip (see its header for the design).
#pragma once
#include <stdio.h>
#include "cnnq_common.hip.h"
#include "cnnq_resident.hip.h"
#include "cnnq_group.hip.h"
#include "cnnq_aciq.hip.h"
#include "cnnq_stats1.hip.h"

namespace {

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
                static const int64_t minwgs = env_int("CNNQ_PLAN_MINWGS", 2048);   // development knob
                if (ncb * smax >= minwgs) break;
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
    // enough workgroups to fill 256 CUs (x 6-8 resident each) a few times over; half as many when a workgroup owns
    // many channels (small H*W): every workgroup reduces and writes one partial record per channel, and pass B reads
    // them all back (measured on the 14x14 / 7x7 layers of ResNet-50 b512: 4-12 % per layer)
    static const int64_t forced = env_int("CNNQ_PLAN_WGS", 0);   // development knob
    const int64_t target = forced ? forced : (g->mode == 2 && g->k >= 8) ? 2048 : 4096;
    int64_t S = (target + g->ncb - 1) / g->ncb;
    if (S > N) S = N;
    if (max_groups > 0 && S > max_groups) S = max_groups;
    if (fine) {
        // table-driven elementwise passes (no per-workgroup reduction or partial record): many short
        // workgroups dispatched in address order - about 14 KB of x per workgroup - stream read+write
        // markedly faster than long-lived ones (6.1-6.7 vs 5.4 TB/s measured)
        const int64_t cols = (g->mode == 1) ? g->w : (int64_t)g->k * HW * v.A / v.vec / v.A;
        const int64_t row_bytes = cols * v.vec * 4;
        const int64_t tile = fine > 1 ? (int64_t)fine : 14336;                      // fine > 1: bytes per tile
        int64_t rows = (tile + row_bytes / 2) / (row_bytes > 0 ? row_bytes : 1);    // swept 8-32 KB
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
// the register-resident single-launch form of config 2 (cnnq_resident.hip.h)
// ------------------------------------------------------------------------------------------
// below this many workgroups the whole-channel form leaves CUs idle: prefer the group form
inline int res_min_wgs() {
    static const int v = env_int("CNNQ_RES_MIN_WGS", 192);   // development knob
    return v;
}
#define RES_MIN_WGS res_min_wgs()

// How many members a group may have and still be co-resident: the K = 32 tiles run three workgroups per CU, and the bounds the
// kernels were tuned with - GRP_GS_MAX = 512 of the 768 slots of a 256-CU part, GRP_GS_BIG = 704 for the 160 KB tiles of a channel
// that is alone on the chip - scale with the CUs of the device the process runs on (a partitioned part, a smaller one), read
// once.  Without a device (the planner tests on a CPU box) the 256 CUs of an MI355X in SPX mode are assumed.  A group that is
// not co-resident after all is slow (bounded waits, then the exact recompute), never wrong.
inline int chip_cus() {
    static const int cus = [] {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        return c;
    }();
    return cus;
}
inline int64_t grp_gs_max() { const int64_t v = (int64_t)chip_cus() * 2; return v < GRP_GS_MAX ? v : GRP_GS_MAX; }
inline int64_t grp_gs_big() { const int64_t v = (int64_t)chip_cus() * 3 - chip_cus() / 4; return v < GRP_GS_BIG ? v : GRP_GS_BIG; }

struct WPlan {
    int A, T, K;   // parameter sets per float4, threads per workgroup, samples (16-byte loads) per lane
    WGeo g;
    int wgs;       // workgroups = ceil(C / k)
};

// A channel block (k channels, a multiple of the m channels that share float4s) of the WHOLE batch must fit
// RL x K samples of T / CL row lanes.  Smallest workgroup first; CNNQ_ENOTSUP when nothing fits.
int plan_whole(int64_t N, int64_t C, int64_t HW, bool aligned16, WPlan* p) {
    if (N <= 0 || C <= 0 || HW <= 0) return CNNQ_EINVAL;
    if (C * HW >= (int64_t)1 << 31 || N >= (int64_t)1 << 31) return CNNQ_ERANGE;
    if (!aligned16) return CNNQ_ENOTSUP;
    int m = 1;
    if (HW % 4 == 0) {
        p->A = 1;
    } else if ((C * HW) % 4 == 0) {
        m = 4 / gcd_i((int)(HW % 4), 4);
        p->A = 4;
    } else {
        return CNNQ_ENOTSUP;
    }
    const int64_t u = m * HW / 4;   // float4 columns of the smallest channel block
    if (u > 1024) return CNNQ_ENOTSUP;
    static const int forceT = env_int("CNNQ_RES_T", 0);
    static const int Ts[3] = {256, 512, 1024};
    // at least ~512 contiguous bytes per sample and workgroup when the layer has the channels for it - and ~1 KB when that
    // still leaves two workgroups per CU (round 5, the 64-sample shard: [64,1024,14,14] 25.6 -> 22.0 us with two channels per
    // workgroup, 512 workgroups; layers that would drop to 256 or fewer workgroups lose with it)
    static const int min_knob = env_int("CNNQ_RES_UNITS", 0);   // development knob: float4 per sample and workgroup
    for (int pass = 0; pass < 2; ++pass)
    for (int ti = 0; ti < 3; ++ti) {
        const int min_f4 = min_knob ? min_knob : (pass == 0 ? 64 : 32);
        if (min_knob && pass == 1) break;
        const int T = Ts[ti];
        if (forceT && T != forceT) continue;
        if (p->A == 4 && T == 512) continue;   // not instantiated
        if (u > T) continue;
        int64_t units = (min_f4 + u - 1) / u;
        if (units > T / u) units = T / u;
        if (units * m > C) units = (C + m - 1) / m;
        if (units * m > MAXCH) units = MAXCH / m;
        if (units < 1) continue;
        const int64_t CL = units * u;
        int64_t RL = T / CL;
        if (RL > N) RL = N;
        const int64_t need = (N + RL - 1) / RL;
        const int K = need <= 8 ? 8 : need <= 16 ? 16 : 32;
        if (need > 32 || (T == 1024 && K > 16) || (T == 512 && K < 16)) continue;
        if (!min_knob && pass == 0 && (C + units * m - 1) / (units * m) < 512) continue;   // the wider block only with >= 512 workgroups
        p->T = T;
        p->K = K;
        p->g.N = (int)N; p->g.C = (int)C; p->g.HW = (int)HW; p->g.P = (int)(C * HW);
        p->g.k = (int)(units * m);
        p->g.CL = (int)CL;
        p->g.RL = (int)RL;
        p->wgs = (int)((C + p->g.k - 1) / p->g.k);
        return 0;
    }
    return CNNQ_ENOTSUP;
}

// out = 0: y; 1: y + codes / histogram; 2: packed 4-bit codes instead of y (the XOut of cnnq_qdq.hip.h)
// CNNQ_IEEE_DIVIDE=1: the single-launch kernels take the hardware divide for every channel (A/B against qdq1_fast)
inline unsigned mmq_env_flags() {
    static const unsigned f = env_int("CNNQ_IEEE_DIVIDE", 0) ? MMQ_FLAG_IEEE_DIVIDE : 0u;
    return f;
}

int launch_whole(const float* x, float* y, const WPlan& p, int num_bits, int positive, float* qp, float* mm,
                 hipStream_t st, int out = 0, const XOut& xo = XOut{}, unsigned flags = 0u, const XRank* xrp = nullptr) {
    const dim3 grid((unsigned)p.wgs);
    flags |= mmq_env_flags();
    const bool xrank = xrp && xrp->world > 0;          // the batch is sharded: the cross-rank stage of cnnq_xrank.hip.h (y, or y + codes / histogram)
    if (xrank && out == 2) return CNNQ_ENOTSUP;
    const XRank xr = xrank ? *xrp : XRank{};
    const size_t hb = xhist_lds_bytes(out, xo.hist, 1 << (num_bits < 8 ? num_bits : 8));
#define LAUNCH_W(A, T, K)                                                                                                     \
    do {                                                                                                                      \
        if (xrank && out == 1) hipLaunchKernelGGL((k_mmq_whole<A, T, K, 1, true>), grid, dim3(T), hb, st, x, y, p.g, num_bits, positive, qp, mm, flags, xo, xr); \
        else if (xrank) hipLaunchKernelGGL((k_mmq_whole<A, T, K, 0, true>), grid, dim3(T), 0, st, x, y, p.g, num_bits, positive, qp, mm, flags, xo, xr); \
        else if (out == 0) hipLaunchKernelGGL((k_mmq_whole<A, T, K, 0>), grid, dim3(T), 0, st, x, y, p.g, num_bits, positive, qp, mm, flags, xo);      \
        else if (out == 1) hipLaunchKernelGGL((k_mmq_whole<A, T, K, 1>), grid, dim3(T), hb, st, x, y, p.g, num_bits, positive, qp, mm, flags, xo); \
        else hipLaunchKernelGGL((k_mmq_whole<A, T, K, 2>), grid, dim3(T), 0, st, x, y, p.g, num_bits, positive, qp, mm, flags, xo);               \
    } while (0)
    if (p.A == 1) {
        if (p.T == 256) { if (p.K == 8) LAUNCH_W(1, 256, 8); else if (p.K == 16) LAUNCH_W(1, 256, 16); else LAUNCH_W(1, 256, 32); }
        else if (p.T == 512) { if (p.K == 16) LAUNCH_W(1, 512, 16); else LAUNCH_W(1, 512, 32); }
        else { if (p.K == 8) LAUNCH_W(1, 1024, 8); else LAUNCH_W(1, 1024, 16); }
    } else {
        if (p.T == 256) { if (p.K == 8) LAUNCH_W(4, 256, 8); else if (p.K == 16) LAUNCH_W(4, 256, 16); else LAUNCH_W(4, 256, 32); }
        else { if (p.K == 8) LAUNCH_W(4, 1024, 8); else LAUNCH_W(4, 1024, 16); }
    }
#undef LAUNCH_W
    return launch_status();
}

// ------------------------------------------------------------------------------------------
// the group-exchange single-launch form of config 2 (cnnq_group.hip.h)
// ------------------------------------------------------------------------------------------
struct GPlan {
    Variant v;     // {4, A, 1}
    Geo g;         // column blocks of <= 256 float4 columns, S = batch splits of <= K samples
    int K;         // samples (16-byte loads) a lane holds
    int Gs;        // workgroups per group (the ones that exchange extrema)
    int ngroups;   // groups = arrival counters
    int gstride;   // 8-byte pairs per group block (a multiple of 16 = 128 bytes)
    size_t ws_bytes;
    int flat;      // 1: k_mmq_flat (a group = one channel, members = flat tiles of 256*(K+KL) float4), geometry in fg
    int KL;        // steps of a flat tile that live in LDS (0, or 8 with K = 32: development knob CNNQ_FLAT_KL)
    FGeo fg;
};

// ws layout, the SAME for every geometry (the counters must never alias another launch's pairs: they are only
// ever zero or mid-count; counters of different geometries may alias each other, every launch leaves them zero):
// status word, GRP_MAX_LINES counter lines, then the pair blocks
// header: [0] the status word; [PTF_OFF ..) the region of the fused per-tensor kernel (cnnq_pertensor.hip.h: four
// counter lines and one {max key, inverted min key} record per row), zero whenever no launch is in flight
constexpr size_t GRP_WS_HDR = 65536;
constexpr int GRP_MAX_LINES = 16384;   // counter lines (4 MB)
constexpr size_t GRP_WS_SLOTS = GRP_WS_HDR + (size_t)GRP_MAX_LINES * GRP_CNT_STRIDE * 4;   // the slot meeting's slots: ZERO AT REST
constexpr size_t GRP_WS_SLOT_BYTES = (size_t)16 * GRP_MAX_LINES * 8;   // 2 MB: groups * ceil16(members) <= 16 * GRP_MAX_LINES by the line bound
constexpr size_t GRP_WS_PAIRS = GRP_WS_SLOTS + GRP_WS_SLOT_BYTES;      // pair blocks / records: written before read, may hold anything

// flat tiles (k_mmq_flat) when a channel row is long enough that the row-piece tiling of k_mmq_group would idle lanes
// lds_rows: 0 - never; 1 - eight more tile rows in LDS (KL = 8) only for channels too populous for GRP_GS_MAX plain K = 32
// tiles (VGG-16 b512 [512,64,224,224]: 103 MB per channel -> 628 members of 160 KB, one channel on the chip at a time);
// 2 - for every K = 32 plan (development knob CNNQ_FLAT_KL=8).  allow_full: also rows that are whole multiples of the
// workgroup (the row pieces of k_mmq_group fill every lane there and are tried first).
int plan_flat(int64_t N, int64_t C, int64_t HW, GPlan* p, int lds_rows, bool allow_full) {
    static const int allow = env_int("CNNQ_GRP_FLAT", 1);       // development knob
    static const int forceK = env_int("CNNQ_GRP_K", 0);
    static const int target = env_int("CNNQ_GRP_WGS", 1024);
    if (!allow || HW % 4 != 0) return CNNQ_ENOTSUP;
    const int64_t cpc = HW / 4;
    static const int mincpc = env_int("CNNQ_FLAT_MINCPC", 128);   // development knob
    if (cpc < mincpc || C * HW >= (int64_t)1 << 31 || N * cpc >= (int64_t)1 << 31) return CNNQ_ENOTSUP;
    if (cpc % TPB == 0 && !allow_full) return CNNQ_ENOTSUP;     // the row pieces already fill every lane
    const int64_t total = N * cpc;
    int K = 8;
    if (forceK == 8 || forceK == 16 || forceK == 32) {
        K = forceK;
    } else {
        for (K = 32; K > 8; K >>= 1)
            if (C * ((total + TPB * K - 1) / (TPB * K)) >= target) break;
    }
    while (K < 32 && (total + TPB * K - 1) / (TPB * K) > grp_gs_max()) K <<= 1;
    // a channel in more than ~8 members pays for the meeting more than for the fewer workgroups of a higher tile (round 5, the
    // 64-sample shard: [64,64,56,56] K = 8 / 16 / 32 -> 25 / 13 / 7 members: 23.9 / 21.7 / 19.4 us)
    if (!forceK)
        while (K < 32 && (total + TPB * K - 1) / (TPB * K) > 8) K <<= 1;
    // round 4: eight more steps per tile in LDS (160 KB per workgroup; flat_lds_rows() below: a development knob).  Measured +1.5 %
    // on the packed single launch with the counter meeting, nothing with the slot meeting, -1 % on the b512 step
    int KL = (K == 32 && lds_rows == 2) ? 8 : 0;
    int64_t Gs = (total + TPB * (K + KL) - 1) / (TPB * (K + KL));
    int64_t gs_max = grp_gs_max();
    if (Gs > gs_max && K == 32 && lds_rows >= 1) {          // a big channel: 160 KB tiles, the channel alone on the chip
        KL = 8;
        Gs = (total + TPB * (K + KL) - 1) / (TPB * (K + KL));
        gs_max = grp_gs_big();
    }
    if (Gs > gs_max || Gs < 2) return CNNQ_ENOTSUP;
    const int64_t rows = (TPB * (K + KL)) / cpc + 3;             // samples a tile can touch, with slack
    if (rows * C * HW * 4 >= (int64_t)1 << 32) return CNNQ_ENOTSUP;
    const int64_t nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    if (C * (nsub > 1 ? nsub + 1 : 1) > GRP_MAX_LINES) return CNNQ_ENOTSUP;
    p->flat = 1;
    p->KL = KL;
    p->v = {4, 1, 1};
    p->K = K;
    p->Gs = (int)Gs;
    p->ngroups = (int)C;
    p->gstride = (int)(((Gs + 15) / 16) * 16);
    p->ws_bytes = GRP_WS_PAIRS + (size_t)C * p->gstride * 8;
    FGeo& f = p->fg;
    f.N = (int)N; f.C = (int)C; f.HW = (int)HW; f.P = (int)(C * HW);
    f.cpc = (unsigned)cpc; f.total = (unsigned)total; f.Gs = (int)Gs;
    f.q256 = (unsigned)(TPB / cpc); f.r16 = (unsigned)(TPB % cpc) * 16u; f.rs = (unsigned)(C * HW * 4);
    // Dispatch order: consecutive workgroups hold the same member tile of 4 adjacent channels, so that per sample the
    // chip reads runs of 4 channel rows together (A/B on one box, three rounds: the b512 step 9.14 -> 9.05 ms, k_mmq_flat
    // 0.640 -> 0.651 of peak; 2 and 4 alike, 8 and more no better).  CNNQ_GRP_CB=1: member fastest (development knob).
    static const int cb_knob = env_int("CNNQ_GRP_CB", 4);
    const int64_t cb_fit = ((int64_t)chip_cus() * 3 / 2) / Gs;   // a block's groups must be resident together: half of the slots at K = 32 (384 of 768)
    f.cb = (int)(cb_knob > 1 ? (cb_fit < cb_knob ? (cb_fit < 1 ? 1 : cb_fit) : cb_knob) : 1);
    // describe(): a Geo that tells the same story
    p->g = Geo{};
    p->g.N = (int)N; p->g.C = (int)C; p->g.HW = (int)HW; p->g.P = (int)(C * HW);
    p->g.mode = 3; p->g.S = (int)Gs; p->g.ncb = (int)C; p->g.Cn = (int)C; p->g.nb = 1; p->g.k = 1;
    return 0;
}

int plan_group_compute(int64_t N, int64_t C, int64_t HW, bool aligned16, GPlan* p, bool allow_flat, int lds_rows, int a4_kmax);

// the lds_rows argument of the config-2 kernels' plans: k_mmq_flat has KL = 8 instances for the plain and the packed output
// (<32, 0 / 2, false, 8>), none for the codes output and the cross-rank stage; CNNQ_FLAT_KL=8 (development knob): every
// K = 32 plan takes them
inline int flat_lds_rows(int out, bool xrank) {
    static const int kl_knob = env_int("CNNQ_FLAT_KL", 0);
    return (out == 1 || xrank) ? 0 : (kl_knob == 8 ? 2 : 1);
}

// plans are pure functions of their arguments (the development knobs are read once): the hot call asks for the same
// handful of geometries over and over, so each host thread remembers the last 64
// a4_kmax: the tallest tile of the straddling row pieces (A = 4): 32 for the extrema kernels; the kernels that carry SUMS across
// the meeting (k_fused_group, k_stats_group) hold per-element accumulators and parameters next to the tile and take 16 (plan_sums
// below) - their K = 32 instances spilled a tile row (round 6: no instance of the library uses scratch)
int plan_group(int64_t N, int64_t C, int64_t HW, bool aligned16, GPlan* p, bool allow_flat = true, int lds_rows = 0, int a4_kmax = 32) {
    struct Entry { int64_t N, C, HW; int key, rc; GPlan plan; };
    constexpr int SLOTS = 64;
    thread_local Entry cache[SLOTS];
    thread_local int used = 0, next = 0;
    const int key = (aligned16 ? 1 : 0) | (allow_flat ? 2 : 0) | (lds_rows << 2) | (a4_kmax << 4);
    for (int i = 0; i < used; ++i)
        if (cache[i].N == N && cache[i].C == C && cache[i].HW == HW && cache[i].key == key) {
            *p = cache[i].plan;
            return cache[i].rc;
        }
    const int rc = plan_group_compute(N, C, HW, aligned16, p, allow_flat, lds_rows, a4_kmax);
    Entry& e = cache[next];
    e.N = N; e.C = C; e.HW = HW; e.key = key; e.rc = rc; e.plan = *p;
    next = (next + 1) % SLOTS;
    if (used < SLOTS) ++used;
    return rc;
}

int plan_rows(int64_t N, int64_t C, int64_t HW, GPlan* p, int a4_kmax = 32);
// the plan of the single-launch kernels that exchange sums (configs 3 / 4 / 5)
inline int plan_sums(int64_t N, int64_t C, int64_t HW, bool aligned16, GPlan* p, int lds_rows) {
    return plan_group(N, C, HW, aligned16, p, true, lds_rows, 16);
}

int plan_group_compute(int64_t N, int64_t C, int64_t HW, bool aligned16, GPlan* p, bool allow_flat, int lds_rows, int a4_kmax) {
    if (N <= 0 || C <= 0 || HW <= 0) return CNNQ_EINVAL;
    if (!aligned16) return CNNQ_ENOTSUP;
    p->flat = 0;
    p->KL = 0;
    if (allow_flat && plan_flat(N, C, HW, p, lds_rows, false) == 0) return 0;
    const int rc = plan_rows(N, C, HW, p, a4_kmax);
    if (rc != CNNQ_ENOTSUP || !allow_flat || lds_rows == 0) return rc;
    // neither fits: a channel too populous for 512 plain tiles - flat tiles with eight more rows in LDS, if that is enough
    GPlan q;
    if (plan_flat(N, C, HW, &q, lds_rows, true) != 0 || q.KL != 8) return rc;
    *p = q;
    return 0;
}

// the row-piece tiling of k_mmq_group
int plan_rows(int64_t N, int64_t C, int64_t HW, GPlan* p, int a4_kmax) {
    p->flat = 0;
    p->KL = 0;
    if (HW % 4 == 0) {
        p->v = {4, 1, 1};
    } else if ((C * HW) % 4 == 0) {
        const int m = 4 / gcd_i((int)(HW % 4), 4);
        if ((int64_t)m * HW > TPB * 4) return CNNQ_ENOTSUP;
        p->v = {4, 4, 1};
    } else {
        return CNNQ_ENOTSUP;
    }
    const int rc = make_geo(N, C, HW, p->v, 0, C, 0, 0, 0, &p->g);
    if (rc) return rc;
    Geo& g = p->g;
    static const int forceK = env_int("CNNQ_GRP_K", 0);      // development knobs (kernel sweeps)
    static const int target = env_int("CNNQ_GRP_WGS", 1024);
    const int members_per_split = (g.mode == 1) ? g.nb : 1;
    int K = 4;
    if (forceK == 4 || forceK == 8 || forceK == 16 || forceK == 32) {
        K = forceK;
    } else {
        for (K = 32; K > 4; K >>= 1)   // the largest tile that still yields `target` workgroups
            if ((int64_t)g.ncb * ((N + K - 1) / K) >= target) break;
    }
    if (p->v.A == 4 && K > a4_kmax) K = a4_kmax;
    const int kcap = p->v.A == 4 ? a4_kmax : 32;
    while (K < kcap && ((N + K - 1) / K) * members_per_split > grp_gs_max()) K <<= 1;   // groups stay co-resident
    const int64_t S = (N + K - 1) / K;
    if (S * members_per_split > grp_gs_max()) return CNNQ_ENOTSUP;
    if (S * g.ncb >= (int64_t)1 << 31) return CNNQ_ERANGE;
    g.S = (int)S;
    p->K = K;
    p->Gs = (int)(S * members_per_split);
    p->ngroups = (g.mode == 1) ? g.Cn : g.ncb;
    {   // counter lines: one per group, plus one per sub-group of GRP_SUB members when a group has several
        const int64_t nsub = (p->Gs + GRP_SUB - 1) / GRP_SUB;
        if ((int64_t)p->ngroups * (nsub > 1 ? nsub + 1 : 1) > GRP_MAX_LINES) return CNNQ_ENOTSUP;
    }
    const int64_t pairs = (int64_t)p->Gs * ((g.mode == 1) ? 1 : g.k);
    p->gstride = (int)(((pairs + 15) / 16) * 16);
    p->ws_bytes = GRP_WS_PAIRS + (size_t)p->ngroups * p->gstride * 8;
    return 0;
}

int launch_group(const float* x, float* y, const GPlan& p, int num_bits, int positive, void* ws, float* qp, float* mm,
                 unsigned flags, hipStream_t st, int out = 0, const XOut& xo = XOut{}, const XRank* xrp = nullptr) {
    flags |= mmq_env_flags();
    const bool xrank = xrp && xrp->world > 0;
    if (xrank && out == 2) return CNNQ_ENOTSUP;
    const XRank xr = xrank ? *xrp : XRank{};
    const size_t hb = xhist_lds_bytes(out, xo.hist, 1 << (num_bits < 8 ? num_bits : 8));
    GWs w;
    w.status = reinterpret_cast<unsigned*>(ws);
    w.cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + GRP_WS_HDR);
    w.part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_PAIRS);
    w.slots = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_SLOTS);
    w.gstride = p.gstride;
    const dim3 grid((unsigned)((int64_t)p.g.S * p.g.ncb)), block(TPB);
    // the slot meeting (cnnq_group.hip.h): +11 % on the packed output at b512 (nothing hides its waits there), +-0.5 % on the
    // b512 / b64 steps with y to write; MMQ_FLAG_COUNTERS (tests, A/B): the counter meeting of round 2
    if (!(flags & MMQ_FLAG_COUNTERS) && (size_t)p.ngroups * p.gstride * 8 <= GRP_WS_SLOT_BYTES) flags |= MMQ_FLAG_SLOTS;
    if (p.flat) {
        const dim3 fgrid((unsigned)((int64_t)p.fg.C * p.fg.Gs));
        static const int pk_narrow = env_int("CNNQ_PK_NARROW", 0);          // development knob: the 2-byte stores of round 3
        if (out == 2 && (pk_narrow || ((uintptr_t)xo.packed & 15))) flags |= MMQ_FLAG_PK_NARROW;
        static const int pk_plain = env_int("CNNQ_PK_PLAIN", 0);
        if (out == 2 && pk_plain) flags |= MMQ_FLAG_PK_PLAIN;
        // With no y to write the launch is bound by what the resident workgroups hold for how long, not by the stores:
        // a channel's members dispatched in one burst (member fastest) wait 2-4 us for each other, blocks of 4 channels
        // ~9 us (a channel's members then start over four slot releases).  [512,256,56,56]: 464 -> 421 us (round 4).
        FGeo fg = p.fg;
        static const int cb_forced = env_int("CNNQ_GRP_CB", 0);
        if (out == 2 && !cb_forced) fg.cb = 1;
#define LAUNCH_F(K)                                                                                                                  \
    do {                                                                                                                             \
        if (xrank && out == 1) hipLaunchKernelGGL((k_mmq_flat<K, 1, true>), fgrid, block, hb, st, x, y, fg, num_bits, positive, w, qp, mm, flags, xo, xr); \
        else if (xrank) hipLaunchKernelGGL((k_mmq_flat<K, 0, true>), fgrid, block, 0, st, x, y, fg, num_bits, positive, w, qp, mm, flags, xo, xr); \
        else if (out == 0) hipLaunchKernelGGL((k_mmq_flat<K, 0>), fgrid, block, 0, st, x, y, fg, num_bits, positive, w, qp, mm, flags, xo);      \
        else if (out == 1) hipLaunchKernelGGL((k_mmq_flat<K, 1>), fgrid, block, hb, st, x, y, fg, num_bits, positive, w, qp, mm, flags, xo); \
        else hipLaunchKernelGGL((k_mmq_flat<K, 2>), fgrid, block, 0, st, x, y, fg, num_bits, positive, w, qp, mm, flags, xo);               \
    } while (0)
        if (p.K == 32 && p.KL == 8) {
            if (xrank || out == 1) return CNNQ_EINVAL;      // flat_lds_rows() never plans these
            if (out == 0) hipLaunchKernelGGL((k_mmq_flat<32, 0, false, 8>), fgrid, block, 0, st, x, y, fg, num_bits, positive, w, qp, mm, flags, xo);
            else hipLaunchKernelGGL((k_mmq_flat<32, 2, false, 8>), fgrid, block, 0, st, x, y, fg, num_bits, positive, w, qp, mm, flags, xo);
        } else if (p.K == 32) LAUNCH_F(32); else if (p.K == 16) LAUNCH_F(16); else LAUNCH_F(8);
#undef LAUNCH_F
        return launch_status();
    }
#define LAUNCH_G(A, K)                                                                                                                     \
    do {                                                                                                                                   \
        if (xrank && out == 1) hipLaunchKernelGGL((k_mmq_group<A, K, 1, true>), grid, block, hb, st, x, y, p.g, p.Gs, num_bits, positive, w, qp, mm, flags, xo, xr); \
        else if (xrank) hipLaunchKernelGGL((k_mmq_group<A, K, 0, true>), grid, block, 0, st, x, y, p.g, p.Gs, num_bits, positive, w, qp, mm, flags, xo, xr); \
        else if (out == 0) hipLaunchKernelGGL((k_mmq_group<A, K, 0>), grid, block, 0, st, x, y, p.g, p.Gs, num_bits, positive, w, qp, mm, flags, xo);      \
        else if (out == 1) hipLaunchKernelGGL((k_mmq_group<A, K, 1>), grid, block, hb, st, x, y, p.g, p.Gs, num_bits, positive, w, qp, mm, flags, xo); \
        else hipLaunchKernelGGL((k_mmq_group<A, K, 2>), grid, block, 0, st, x, y, p.g, p.Gs, num_bits, positive, w, qp, mm, flags, xo);               \
    } while (0)
    if (p.v.A == 4) {
        if (p.K == 32) LAUNCH_G(4, 32); else if (p.K == 16) LAUNCH_G(4, 16); else if (p.K == 8) LAUNCH_G(4, 8); else LAUNCH_G(4, 4);
    } else {
        if (p.K == 32) LAUNCH_G(1, 32); else if (p.K == 16) LAUNCH_G(1, 16); else if (p.K == 8) LAUNCH_G(1, 8); else LAUNCH_G(1, 4);
    }
#undef LAUNCH_G
    return launch_status();
}

// the single-launch ACIQ / mid-tread kernels (cnnq_aciq.hip.h; mode 0 / 1) on the plan and the workspace of launch_group;
// slot meeting only.  out: mode 0 - 1 with codes / histogram (xo); mode 1 - 1 with the code histogram (fa.hist)
// xrp (round 6): the batch is sharded - the instances with the cross-rank stage (y only in mode 0; y / y + histogram in mode 1)
int launch_fused(int mode, const float* x, float* y, const GPlan& p, const FusedArgs& fa, void* ws, unsigned flags, hipStream_t st,
                 int out, const XOut& xo, const XRank* xrp = nullptr) {
    const bool xrank = xrp && xrp->world > 0;
    if (xrank && mode == 0 && out == 1) return CNNQ_ENOTSUP;
    const XRank xr = xrank ? *xrp : XRank{};
    if ((size_t)p.ngroups * p.gstride * 8 > GRP_WS_SLOT_BYTES) return CNNQ_ENOTSUP;
    if (p.KL && !(p.flat && p.K == 32 && p.KL == 8)) return CNNQ_ENOTSUP;
    if (mode == 1 && !p.flat && p.v.A != 1) return CNNQ_ENOTSUP;      // no straddling mid-tread instance
    if (p.KL && mode == 0 && out == 1) return CNNQ_ENOTSUP;          // 32 KB of LDS rows + the 32 KB code table: two workgroups per CU, a big channel needs three
    flags |= mmq_env_flags();
    // mode 0 counts codes in the table of the config-2 kernels: 256 bins with bit allocation (a channel's width is its own), else 2^bits
    const bool ba = fa.cfg.bit_alloc && fa.cfg.num_bits <= 4;
    const size_t hb = mode == 0 ? xhist_lds_bytes(out, xo.hist, ba ? 256 : 1 << (fa.cfg.num_bits < 8 ? fa.cfg.num_bits : 8)) : 0;
    GWs w;
    w.status = reinterpret_cast<unsigned*>(ws);
    w.cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + GRP_WS_HDR);
    w.part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_PAIRS);
    w.slots = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_SLOTS);
    w.gstride = p.gstride;
    const dim3 block(TPB);
    if (p.flat) {
        const dim3 fgrid((unsigned)((int64_t)p.fg.C * p.fg.Gs));
#define LAUNCH_FF(K, KL)                                                                                                   \
    do {                                                                                                                   \
        if (xrank && mode == 0) hipLaunchKernelGGL((k_fused_flat<K, 0, KL, 0, true>), fgrid, block, 0, st, x, y, p.fg, w, fa, flags, xo, xr);       \
        else if (xrank && out == 1) hipLaunchKernelGGL((k_fused_flat<K, 1, KL, 1, true>), fgrid, block, 0, st, x, y, p.fg, w, fa, flags, xo, xr);   \
        else if (xrank) hipLaunchKernelGGL((k_fused_flat<K, 0, KL, 1, true>), fgrid, block, 0, st, x, y, p.fg, w, fa, flags, xo, xr);               \
        else if (mode == 0 && out == 1) hipLaunchKernelGGL((k_fused_flat<K, 1, 0, 0>), fgrid, block, hb, st, x, y, p.fg, w, fa, flags, xo);   \
        else if (mode == 0) hipLaunchKernelGGL((k_fused_flat<K, 0, KL, 0>), fgrid, block, 0, st, x, y, p.fg, w, fa, flags, xo);         \
        else if (out == 1) hipLaunchKernelGGL((k_fused_flat<K, 1, KL, 1>), fgrid, block, 0, st, x, y, p.fg, w, fa, flags, xo);          \
        else hipLaunchKernelGGL((k_fused_flat<K, 0, KL, 1>), fgrid, block, 0, st, x, y, p.fg, w, fa, flags, xo);                        \
    } while (0)
        if (p.K == 32 && p.KL == 8) LAUNCH_FF(32, 8);
        else if (p.K == 32) LAUNCH_FF(32, 0);
        else if (p.K == 16) LAUNCH_FF(16, 0);
        else LAUNCH_FF(8, 0);
#undef LAUNCH_FF
        return launch_status();
    }
    const dim3 grid((unsigned)((int64_t)p.g.S * p.g.ncb));
#define LAUNCH_FG(A, K, MODE)                                                                                                   \
    do {                                                                                                                        \
        if (xrank && out == 1) { if constexpr (MODE == 1) hipLaunchKernelGGL((k_fused_group<A, K, 1, MODE, true>), grid, block, 0, st, x, y, p.g, p.Gs, w, fa, flags, xo, xr); } \
        else if (xrank) hipLaunchKernelGGL((k_fused_group<A, K, 0, MODE, true>), grid, block, 0, st, x, y, p.g, p.Gs, w, fa, flags, xo, xr);   \
        else if (out == 1) hipLaunchKernelGGL((k_fused_group<A, K, 1, MODE>), grid, block, (MODE == 0 ? hb : 0), st, x, y, p.g, p.Gs, w, fa, flags, xo);   \
        else hipLaunchKernelGGL((k_fused_group<A, K, 0, MODE>), grid, block, 0, st, x, y, p.g, p.Gs, w, fa, flags, xo);            \
    } while (0)
    if (mode == 1) {
        if (p.K == 32) LAUNCH_FG(1, 32, 1); else if (p.K == 16) LAUNCH_FG(1, 16, 1); else if (p.K == 8) LAUNCH_FG(1, 8, 1); else LAUNCH_FG(1, 4, 1);
    } else if (p.v.A == 4) {
        if (p.K == 32) return CNNQ_ENOTSUP;      // plan_sums never plans it (no such instance: it spilled)
        if (p.K == 16) LAUNCH_FG(4, 16, 0); else if (p.K == 8) LAUNCH_FG(4, 8, 0); else LAUNCH_FG(4, 4, 0);
    } else {
        if (p.K == 32) LAUNCH_FG(1, 32, 0); else if (p.K == 16) LAUNCH_FG(1, 16, 0); else if (p.K == 8) LAUNCH_FG(1, 8, 0); else LAUNCH_FG(1, 4, 0);
    }
#undef LAUNCH_FG
    return launch_status();
}

// the single-read statistics kernel (cnnq_stats1.hip.h) on a flat plan and the workspace of launch_group
int launch_stats_flat(const float* x, const GPlan& p, const St1Args& sa, void* ws, unsigned flags, bool ntl, hipStream_t st,
                      const XRank* xrp = nullptr, bool dry = false) {
    const bool xrank = xrp && xrp->world > 0;          // the batch is sharded: both phases' folds exchanged inside the launch
    const XRank xr = xrank ? *xrp : XRank{};
    if (!p.flat || p.KL) return CNNQ_ENOTSUP;
    if ((size_t)p.ngroups * p.gstride * ST_LINE * 8 > GRP_WS_SLOT_BYTES) return CNNQ_ENOTSUP;
    if (dry) return 0;                                 // cnnq_pc_stats_route: the checks alone
    GWs w;
    w.status = reinterpret_cast<unsigned*>(ws);
    w.cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + GRP_WS_HDR);
    w.part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_PAIRS);
    w.slots = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_SLOTS);
    w.gstride = p.gstride;
    const dim3 grid((unsigned)((int64_t)p.fg.C * p.fg.Gs)), block(TPB);
    FGeo fg = p.fg;
    fg.cb = 1;          // member fastest: with nothing to write the launch is bound by how long a group's members wait for each other
#define LAUNCH_SF(KR, KL)                                                                                              \
    do {                                                                                                               \
        if (xrank) {                                                                                                   \
            if (sa.need_relu) hipLaunchKernelGGL((k_stats_flat<KR, KL, true, false, true>), grid, block, 0, st, x, fg, w, sa, flags, xr);   \
            else hipLaunchKernelGGL((k_stats_flat<KR, KL, false, false, true>), grid, block, 0, st, x, fg, w, sa, flags, xr);                \
        }                                                                                                              \
        else if (sa.need_relu && ntl) hipLaunchKernelGGL((k_stats_flat<KR, KL, true, true>), grid, block, 0, st, x, fg, w, sa, flags);   \
        else if (sa.need_relu) hipLaunchKernelGGL((k_stats_flat<KR, KL, true, false>), grid, block, 0, st, x, fg, w, sa, flags);    \
        else if (ntl) hipLaunchKernelGGL((k_stats_flat<KR, KL, false, true>), grid, block, 0, st, x, fg, w, sa, flags);             \
        else hipLaunchKernelGGL((k_stats_flat<KR, KL, false, false>), grid, block, 0, st, x, fg, w, sa, flags);                     \
    } while (0)
    if (p.K == 32) LAUNCH_SF(24, 8); else if (p.K == 16) LAUNCH_SF(16, 0); else LAUNCH_SF(8, 0);
#undef LAUNCH_SF
    return launch_status();
}

// the single-read statistics kernel on a row-piece plan (k_stats_group): SG_NP words per member and owned channel
// Where the row-piece single launch beats the three-launch chain (measured: tools/bench_stats_small.py at batch 64 and 512, round 6 -
// after its second meeting stopped being waited for): every one-channel-per-lane shape (14x14: 19.6 against 30.5 us at
// [64,256,14,14], 47.1 / 49.4 at [512,256,14,14], 134 / 156 at [512,1024,14,14]); straddling rows (7x7) only while the tensor is
// small ([64,512,7,7] 28.6 against 37.9 us; [64,2048,7,7] 60.7 / 35.7 and [512,512,7,7] 90 / 47 lose: many channels per workgroup
// reduce, publish and fold seven words each in lockstep).
inline bool stats_group_pays(const GPlan& p, int64_t N, int64_t C, int64_t HW) {
    return p.v.A == 1 || N * C * HW * 4 <= ((int64_t)8 << 20);
}

int launch_stats_group(const float* x, const GPlan& p, const St1Args& sa, void* ws, size_t ws_bytes, unsigned flags, hipStream_t st,
                       const XRank* xrp = nullptr, bool dry = false) {
    const bool xrank = xrp && xrp->world > 0;          // the batch is sharded: both phases' folds exchanged inside the launch
    const XRank xr = xrank ? *xrp : XRank{};
    if (p.flat) return CNNQ_ENOTSUP;
    const int kk = (p.g.mode == 1) ? 1 : p.g.k;
    const int64_t words = (((int64_t)p.Gs * SG_NP * kk + 15) / 16) * 16;
    if (kk > MAXCH || words >= (int64_t)1 << 30) return CNNQ_ENOTSUP;
    const size_t bytes = (size_t)p.ngroups * (size_t)words * 8;
    if (bytes > GRP_WS_SLOT_BYTES || GRP_WS_PAIRS + bytes > ws_bytes) return CNNQ_ENOTSUP;
    if (p.v.A == 4 && p.K == 32) return CNNQ_ENOTSUP;      // plan_sums never plans it (no such instance: it spilled)
    if (p.K == 32 && (int64_t)p.g.P * 4 * 32 >= ((int64_t)1 << 32)) return CNNQ_ENOTSUP;     // the LDS rows' 32-bit offsets from the tile's base
    if (dry) return 0;
    GWs w;
    w.status = reinterpret_cast<unsigned*>(ws);
    w.cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + GRP_WS_HDR);
    w.part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_PAIRS);
    w.slots = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_SLOTS);
    w.gstride = (int)words;
    const dim3 grid((unsigned)((int64_t)p.g.S * p.g.ncb)), block(TPB);
#define LAUNCH_SG(A, KR, KL)                                                                                             \
    do {                                                                                                                 \
        if (xrank && sa.need_relu) hipLaunchKernelGGL((k_stats_group<A, KR, KL, true, true>), grid, block, 0, st, x, p.g, p.Gs, w, sa, flags, xr);  \
        else if (xrank) hipLaunchKernelGGL((k_stats_group<A, KR, KL, false, true>), grid, block, 0, st, x, p.g, p.Gs, w, sa, flags, xr);            \
        else if (sa.need_relu) hipLaunchKernelGGL((k_stats_group<A, KR, KL, true>), grid, block, 0, st, x, p.g, p.Gs, w, sa, flags);  \
        else hipLaunchKernelGGL((k_stats_group<A, KR, KL, false>), grid, block, 0, st, x, p.g, p.Gs, w, sa, flags);              \
    } while (0)
    if (p.v.A == 4) {
        if (p.K == 32) return CNNQ_ENOTSUP;      // plan_sums never plans it (no such instance: it spilled)
        if (p.K == 16) LAUNCH_SG(4, 16, 0); else if (p.K == 8) LAUNCH_SG(4, 8, 0); else LAUNCH_SG(4, 4, 0);
    } else {
        if (p.K == 32) LAUNCH_SG(1, 24, 8); else if (p.K == 16) LAUNCH_SG(1, 16, 0); else if (p.K == 8) LAUNCH_SG(1, 8, 0); else LAUNCH_SG(1, 4, 0);
    }
#undef LAUNCH_SG
    return launch_status();
}

// rank-local extrema in one launch (k_minmax_group): the plan and workspace of launch_group
int launch_minmax_group(const float* x, const GPlan& p, void* ws, float* out, hipStream_t st) {
    GWs w;
    w.status = reinterpret_cast<unsigned*>(ws);
    w.cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + GRP_WS_HDR);
    w.part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_PAIRS);
    w.slots = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + GRP_WS_SLOTS);
    w.gstride = p.gstride;
    const dim3 grid((unsigned)((int64_t)p.g.S * p.g.ncb)), block(TPB);
#define LAUNCH_MG(A, K) hipLaunchKernelGGL((k_minmax_group<A, K>), grid, block, 0, st, x, p.g, p.Gs, w, out)
    if (p.v.A == 4) {
        if (p.K == 32) LAUNCH_MG(4, 32); else if (p.K == 16) LAUNCH_MG(4, 16); else if (p.K == 8) LAUNCH_MG(4, 8); else LAUNCH_MG(4, 4);
    } else {
        if (p.K == 32) LAUNCH_MG(1, 32); else if (p.K == 16) LAUNCH_MG(1, 16); else if (p.K == 8) LAUNCH_MG(1, 8); else LAUNCH_MG(1, 4);
    }
#undef LAUNCH_MG
    return launch_status();
}

// fused Q/DQ whose workgroups derive their parameters from the W gathered {min, max} records (k_qdq<GATH>)
int launch_qdq_gathered(const float* x, float* y, const Geo& g, const Variant& v, const float* gathered, const GathArgs& ga,
                        hipStream_t st) {
    const dim3 grid((unsigned)(g.S * g.ncb)), block(TPB);
    uint8_t* nocodes = nullptr;
    unsigned long long* nohist = nullptr;
#define LAUNCH_QG(VEC, A, J) \
    hipLaunchKernelGGL((k_qdq<VEC, A, J, false, false, true>), grid, block, 0, st, x, y, g, gathered, nocodes, nohist, ga)
    CNNQ_DISPATCH(v, LAUNCH_QG);
#undef LAUNCH_QG
    return launch_status();
}

}  // namespace
