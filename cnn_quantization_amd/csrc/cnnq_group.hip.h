// cnnq_group.hip.h - config 2 in ONE launch and ONE read of x for tensors whose channels do NOT fit one workgroup
// (cnnq_resident.hip.h covers those that do): 8 instead of 12 bytes per element.
// Part of the single translation unit cnnq_kernels.hip (see its header for the shared column-block decomposition).
//
// A workgroup's share of x stays in REGISTERS between the statistics and the Q/DQ.  It owns one column block
// (<= 256 float4 columns aligned to channel boundaries) and R <= K consecutive samples of it, issues its K
// 16-byte loads per lane back to back, reduces them to per-channel {min, max}, and then meets the OTHER workgroups
// that hold pieces of the same channels - the "group": the S batch splits, times the nb column slices when one
// channel row is wider than a workgroup.  The exchange follows the write-through publish / counter recipe of
// cdna_hip_programming.md Guideline 16 (R1), with what the measurements of this kernel added:
//
//   * partial {min, max} pairs are 8-byte agent-scope (sc1, write-through) stores, every storing wave drains its
//     vmcnt, then ONE lane bumps the group's arrival counter.  NO release fence: buffer_wbl2 would have to write
//     back the megabytes of y that this very kernel keeps dirtying in the XCD's L2 - with it the kernel ran 2-3x
//     slower than the two-pass chain;
//   * the workspace is fine-grained (uncached) device memory and every group's pairs live in their own
//     128-byte-aligned block, written only by that group and read only after its counter is complete: what a
//     member reads never depends on what an XCD's L2 (not coherent with the other XCDs') or a CU's L1 may still
//     hold, within a launch or from the previous one - so no agent-scope acquire is needed either (buffer_inv sc1
//     per workgroup cost 6 %; switch GRP_ACQUIRE);
//   * every counter has a 256-byte line of its own, in a region of the workspace that no geometry ever uses for
//     pairs (16 counters per line serialised 16 groups' traffic: 3x slower than the chain; counters at a
//     geometry-dependent offset got overwritten by another geometry's pairs: wrong scales for groups >= 64);
//   * a line holds three words - arrivals, departures, ready flag - and whoever performs the LAST departure zeroes
//     it: arrivals are never confused with early leavers (timeout, test flag), every launch leaves the workspace
//     zero under any interleaving, the caller zeroes it ONCE, the launch replays from a HIP graph.  Groups of more
//     than GRP_SUB members arrive in sub-groups whose last arrivers meet on a top counter; the very last raises one
//     flag per sub-group (208 members on one word cost the 112x112 layer 60 % of its time);
//   * lane 0 polls (relaxed sc1 load, s_sleep back-off 0.2 -> 1.7 us), every member then reads the group's block with
//     relaxed agent-scope loads (sc1: they bypass the L1 like the poll; round 3 - the block is uncached memory either
//     way, this only says so to the memory model); its own departure is counted after its stores are issued;
//   * forward progress does not depend on dispatch order: the wait is bounded twice (20 ms of the 100 MHz clock,
//     2^20 polls); a workgroup that gives up recomputes its channels' extrema from x itself (exact, hence the
//     same bits) and raises bit 0 of the status word (bit 1: the recompute was forced by the test hook, flags & 1).
//     While bit 0 is up (cnnq_group_ws_status_clear lowers it) the bound is 0.5 ms instead of 20: a device shared with
//     another process's workgroups loses 20 ms per expiry otherwise, and a meeting that works takes 10-40 us.
#pragma once
#include "cnnq_common.hip.h"
#include "cnnq_qdq.hip.h"
#include "cnnq_xrank.hip.h"
#include "cnnq_resident.hip.h"   // pmin / pmax / lane_acc

namespace {

constexpr long long GRP_TIMEOUT_TICKS = 2000000;   // 20 ms of the 100 MHz constant clock
constexpr long long GRP_TIMEOUT_SHORT = 50000;     // 0.5 ms: once a wait HAS expired on this workspace (status bit 0) the
                                                   // device is evidently shared - another process's workgroups hold the
                                                   // slots a group needs - and every further expiry would cost 20 ms; a
                                                   // meeting that works takes 10-40 us, so 0.5 ms changes nothing for it
constexpr int GRP_TIMEOUT_SPINS = 1 << 20;         // second bound on the same wait
constexpr int GRP_CNT_STRIDE = 64;                 // words between two groups' counters: one 256-byte line each - with
                                                   // 16 counters per line every arrival, departure and poll of 16
                                                   // groups serialised on one line (~130 ns per workgroup, measured)
constexpr int GRP_GS_MAX = 512;                    // members of a group (all co-resident: capacity >= 512 workgroups)
constexpr int GRP_GS_BIG = 704;                    // ... of a big-channel group (K = 32 + 8 LDS rows: the channel alone on the chip's 768 slots)
#ifndef GRP_ACQUIRE
#define GRP_ACQUIRE 0     // 1: agent-scope acquire (buffer_inv sc1) between the wait and the reads of the group's pairs.
                          // Not needed by construction - the pairs live in fine-grained (uncached) memory, a block is
                          // written only by its group and read only after the group is complete - and it costs 6 %
                          // (each buffer_inv drops the CU's whole L1 under three streaming workgroups); kept as a switch.
#endif
constexpr int GRP_SUB = 16;                        // members per arrival counter
#ifndef GRP_K32_WAVES
#define GRP_K32_WAVES 3   // waves per SIMD the K = 32 tile is compiled for (168 VGPRs)
#endif

#ifdef GRP_TRACE
// development build only (tools/trace_group.py): lane 0 of every workgroup stamps its phases with the 100 MHz clock
__device__ unsigned long long* g_grp_trace = nullptr;   // [workgroup][16]: 0-6 phases, 7 hardware id, 8 stores drained
#define GRP_STAMP(i)                                                                                     \
    do {                                                                                                 \
        if (g_grp_trace && threadIdx.x == 0) g_grp_trace[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define GRP_STAMP(i)
#endif

__device__ __forceinline__ unsigned long long pack_pair(float mn, float mx) {
    return (unsigned long long)__float_as_uint(mn) | ((unsigned long long)__float_as_uint(mx) << 32);
}
__device__ __forceinline__ void unpack_pair(unsigned long long p, float& mn, float& mx) {
    mn = __uint_as_float((unsigned)(p & 0xffffffffull));
    mx = __uint_as_float((unsigned)(p >> 32));
}

struct RBlk {
    Blk b;
    int group, member;
};

// blockIdx -> (group, member) -> tile.  Members of a group are consecutive workgroups.
__device__ __forceinline__ RBlk rblk_of(const Geo& g, int Gs) {
    RBlk r;
    const int bid = (int)blockIdx.x;
    r.group = bid / Gs;
    r.member = bid - r.group * Gs;
    int s;
    if (g.mode == 1) {
        const int cpc = g.HW / 4;
        s = r.member / g.nb;
        const int bb = r.member - s * g.nb;
        const int c = g.cbeg + r.group;
        r.b.c0 = c;
        r.b.c1 = c + 1;
        r.b.col0 = c * cpc + bb * g.w;
        r.b.col1 = min(r.b.col0 + g.w, (c + 1) * cpc);
    } else {
        s = r.member;
        r.b.c0 = g.cbeg + r.group * g.k;
        r.b.c1 = min(g.cbeg + g.Cn, r.b.c0 + g.k);
        r.b.col0 = (int)(((int64_t)r.b.c0 * g.HW) / 4);
        r.b.col1 = (int)(((int64_t)r.b.c1 * g.HW) / 4);
    }
    r.b.n0 = (int)(((int64_t)s * g.N) / g.S);
    r.b.n1 = (int)(((int64_t)(s + 1) * g.N) / g.S);
    r.b.grp = s;
    return r;
}

// per-lane accumulators -> per-channel extrema of the workgroup's tile in sh_mn / sh_mx [c1 - c0]
template <int A>
__device__ __forceinline__ void wg_channel_minmax(const Geo& g, const Blk& b, bool ok, const float (&mn)[A],
                                                  const float (&mx)[A], float* l_mn, float* l_mx, float* sh_mn,
                                                  float* sh_mx) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (g.mode == 1) {
        float tn = ok ? mn[0] : INFINITY, tx = ok ? mx[0] : -INFINITY;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { tn = pmin(tn, shfl_xor_f(tn, m)); tx = pmax(tx, shfl_xor_f(tx, m)); }
        if (lane == 0) { l_mn[wv] = tn; l_mx[wv] = tx; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < TPB / 64; ++i) { tn = pmin(tn, l_mn[i]); tx = pmax(tx, l_mx[i]); }
            sh_mn[0] = tn;
            sh_mx[0] = tx;
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
        l_mn[tid * A + a] = ok ? mn[a] : INFINITY;
        l_mx[tid * A + a] = ok ? mx[a] : -INFINITY;
    }
    __syncthreads();
    const int epc = g.HW * A / 4;   // LDS entries per channel
    if (epc <= 16) {
        for (int ch = tid; ch < b.c1 - b.c0; ch += TPB) {
            float tn = INFINITY, tx = -INFINITY;
            for (int e = ch * epc; e < (ch + 1) * epc; ++e) { tn = pmin(tn, l_mn[e]); tx = pmax(tx, l_mx[e]); }
            sh_mn[ch] = tn;
            sh_mx[ch] = tx;
        }
    } else {
        for (int ch = wv; ch < b.c1 - b.c0; ch += TPB / 64) {
            float tn = INFINITY, tx = -INFINITY;
            for (int e = ch * epc + lane; e < (ch + 1) * epc; e += 64) { tn = pmin(tn, l_mn[e]); tx = pmax(tx, l_mx[e]); }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { tn = pmin(tn, shfl_xor_f(tn, m)); tx = pmax(tx, shfl_xor_f(tx, m)); }
            if (lane == 0) { sh_mn[ch] = tn; sh_mx[ch] = tx; }
        }
    }
    __syncthreads();
}

// cold path (a wait timed out, or the test flag): the extrema of the group's channels straight from x, all samples
template <int A>
__device__ __forceinline__ void group_minmax_from_x(const float* __restrict__ x, const Geo& g, const Blk& b, float* l_mn,
                                                 float* l_mx, float* sh_mn, float* sh_mx) {
    const int tid = threadIdx.x;
    float mn[A], mx[A];
    bool nan = false;
#pragma unroll
    for (int a = 0; a < A; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
    bool ok = true;
    if (g.mode == 1) {
        const int cpc = g.HW / 4;
        for (int n = 0; n < g.N; ++n)
            for (int col = tid; col < cpc; col += TPB) {
                float v[4];
                ldv<4>(x + (size_t)n * (size_t)g.P + ((size_t)b.c0 * cpc + col) * 4, v);
                lane_acc<A>(v, mn, mx, nan);
            }
    } else {
        const int col = b.col0 + tid;
        ok = col < b.col1;
        if (ok)
            for (int n = 0; n < g.N; ++n) {
                float v[4];
                ldv<4>(x + (size_t)n * (size_t)g.P + (size_t)col * 4, v);
                lane_acc<A>(v, mn, mx, nan);
            }
    }
    if (A == 1 && nan) { mn[0] = NAN; mx[0] = NAN; }
    __syncthreads();   // l_mn / l_mx may still be read by a previous reduction
    wg_channel_minmax<A>(g, b, ok, mn, mx, l_mn, l_mx, sh_mn, sh_mx);
}

// one departure from a counter line (words: [0] arrivals, [1] departures, [2] ready flag); the last of `actors` zeroes it
__device__ __forceinline__ void grp_leave(unsigned* ln, unsigned actors) {
    if (__hip_atomic_fetch_add(ln + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == actors - 1u) {
        __hip_atomic_store(ln + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ln + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ln + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the meeting of a group's workgroups (lane 0 of each member) -------------------------------------------------
// Counter lines (256 bytes each; words: [0] arrivals A, [1] departures D, [2] ready flag F).  Up to GRP_SUB members
// share one line; larger groups arrive in sub-groups of GRP_SUB whose last arrivers meet on the group's top line, and
// the very last one raises every sub-group's flag: no word ever sees more than GRP_SUB + 1 increments or GRP_SUB
// pollers (208 members on ONE word cost the 112x112 layer 60 % of its time).  Arrivals and departures are counted
// apart, so a workgroup that gives up waiting (or the test flag) and leaves early cannot be mistaken for an arrival;
// whoever performs the LAST departure of a line - members and, with two levels, the flag raiser - zeroes it: every
// launch leaves the workspace zero under any interleaving.
// A group owns lines_per_group(Gs) consecutive lines per exchange (`nex` line sets per group for a kernel with
// several exchanges per launch, `ex` selects one).
inline int grp_lines_per_group_host(int Gs) {
    const int nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    return nsub > 1 ? nsub + 1 : 1;
}
__device__ __forceinline__ int grp_lines_per_group(int Gs) {
    const int nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    return nsub > 1 ? nsub + 1 : 1;
}
__device__ __forceinline__ unsigned* grp_lines(unsigned* cnt, int group, int Gs, int ex, int nex) {
    return cnt + ((size_t)group * nex + ex) * grp_lines_per_group(Gs) * GRP_CNT_STRIDE;
}

// arrive and wait; returns 1 when the wait expired, 2 when the test hook (flags & 1) skipped it, else 0.  `on_sub_last()` runs in the LAST
// arriver of a sub-group, before it reports to the top line (two-level groups only): the place to fold the
// sub-group's records.
template <typename F>
__device__ __forceinline__ int grp_meet(unsigned* top, int member, int Gs, unsigned flags, F on_sub_last,
                                        long long timeout_ticks = GRP_TIMEOUT_TICKS) {
    const int nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    const int si = member / GRP_SUB;
    const unsigned m_i = (unsigned)min(GRP_SUB, Gs - si * GRP_SUB);
    unsigned* line = (nsub > 1) ? top + (size_t)(1 + si) * GRP_CNT_STRIDE : top;
    const unsigned seen = __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    bool ready = (nsub == 1) && seen >= m_i;
    if (nsub > 1 && seen == m_i) {
        on_sub_last();
        // last arrival of this sub-group -> the group's top counter (exactly nsub arrivals per launch)
        if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nsub - 1u) {
            __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int j = 0; j < nsub; ++j)
                __hip_atomic_store(top + (size_t)(1 + j) * GRP_CNT_STRIDE + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the flags are out before this actor departs
            for (int j = 0; j < nsub; ++j)
                grp_leave(top + (size_t)(1 + j) * GRP_CNT_STRIDE, (unsigned)min(GRP_SUB, Gs - j * GRP_SUB) + 1u);
            ready = true;
        }
    }
    int timed_out = (flags & MMQ_FLAG_TEST_HOOK) ? 2 : 0;      // 2: the test hook, 1: a wait that really expired (status bits 1 / 0)
    if (!timed_out && !ready) {
        const unsigned* pw = (nsub > 1) ? line + 2 : line;     // two levels: the flag; one level: the arrivals
        const unsigned want = (nsub > 1) ? 1u : m_i;
        long long t0 = 0;
        for (int spins = 0;; ++spins) {
            // back-off in units of 64 clocks (swept 4/8/16 ... 32/64/127: +-1.5 %, the schedule hardly matters)
            if (spins < 2) __builtin_amdgcn_s_sleep(8);
            else if (spins < 6) __builtin_amdgcn_s_sleep(32);
            else __builtin_amdgcn_s_sleep(64);
            if (__hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
            // the clock is a scalar-memory round trip of its own (microseconds under a streaming load): consult it
            // every 32 polls (~50 us apart) - the bound stays 20 ms, the common case pays for the poll alone
            if ((spins & 31) == 31) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > timeout_ticks) { timed_out = 1; break; }
            }
            if (spins > GRP_TIMEOUT_SPINS) { timed_out = 1; break; }
        }
    }
#if GRP_ACQUIRE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    return timed_out;
}

// leave the group (the last departure of a counter line re-arms it for the next launch)
__device__ __forceinline__ void grp_depart(unsigned* top, int member, int Gs) {
    const int nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    const int si = member / GRP_SUB;
    const unsigned m_i = (unsigned)min(GRP_SUB, Gs - si * GRP_SUB);
    grp_leave((nsub > 1) ? top + (size_t)(1 + si) * GRP_CNT_STRIDE : top, (nsub > 1) ? m_i + 1u : m_i);
}

// Arrival WITHOUT waiting: true in exactly one member per launch - the last one to arrive, which by then may read what
// every other member published before its own arrival - and nobody waits, so there is nothing to time out.  The last
// arriver of a line re-arms it (all of that line's arrivals are in).
__device__ __forceinline__ bool grp_arrive_last(unsigned* top, int member, int Gs) {
    const int nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    const int si = member / GRP_SUB;
    const unsigned m_i = (unsigned)min(GRP_SUB, Gs - si * GRP_SUB);
    unsigned* line = (nsub > 1) ? top + (size_t)(1 + si) * GRP_CNT_STRIDE : top;
    if (__hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != m_i) return false;
    __hip_atomic_store(line, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nsub == 1) return true;
    if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)nsub - 1u) return false;
    __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Departure that tells who was last (the slot meeting below): true in exactly one member per launch, after every other
// member has departed; the departure words it went through are zero again.
__device__ __forceinline__ bool grp_depart_last(unsigned* top, int member, int Gs) {
    const int nsub = (Gs + GRP_SUB - 1) / GRP_SUB;
    const int si = member / GRP_SUB;
    const unsigned m_i = (unsigned)min(GRP_SUB, Gs - si * GRP_SUB);
    unsigned* line = (nsub > 1) ? top + (size_t)(1 + si) * GRP_CNT_STRIDE : top;
    if (__hip_atomic_fetch_add(line + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != m_i) return false;
    __hip_atomic_store(line + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nsub == 1) return true;
    if (__hip_atomic_fetch_add(top + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)nsub - 1u) return false;
    __hip_atomic_store(top + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// ---- the slot meeting (round 4; MMQ_FLAG_SLOTS) ---------------------------------------------------------------------
// The counter meeting above is a chain of four dependent memory round trips per member (pair stored and acknowledged ->
// arrival counted -> counter polled -> pairs read), 1-2 us each under a streaming load.  Here a member's {min, max} pair
// IS its arrival: slots are zero at rest, a member stores the complement of its pair (NaNs canonical, so the complement is
// never zero) with ONE 8-byte store and nobody waits for the acknowledgement; every member then polls the group's slots
// (wave 0 in k_mmq_flat, every wave its share in k_mmq_group; a few slots in flight per lane) until none is zero - one
// round trip after the last store landed.  Departures are
// counted (word [1] of the counter lines); whoever departs last zeroes the slots again: every launch leaves them zero.
__device__ __forceinline__ unsigned long long slot_of(float mn, float mx) {
    const bool nn = (mn != mn) || (mx != mx);
    return ~pack_pair(nn ? NAN : mn, nn ? NAN : mx);
}

// Wave 0 of the member does the meeting (no barrier inside the wait: the other waves sit at the caller's barrier): lane l
// watches members l, l + 64, ... (12 at most, in windows of 4).  Returns 0, or 1 (a wait expired) / 2 (the test hook),
// meaningful in thread 0; tn / tx: this lane's share of the fold (identities outside wave 0).
__device__ __forceinline__ int slots_meet(unsigned long long* slots, int member, int Gs, float cmn, float cmx, unsigned flags,
                                          long long timeout_ticks, float& tn, float& tx) {
    static_assert(GRP_GS_BIG <= 12 * 64, "a lane of wave 0 watches at most 12 members (three windows of 4)");
    const int tid = threadIdx.x;
    tn = INFINITY;
    tx = -INFINITY;
    if (tid >= 64) return 0;
    if (tid == 0) __hip_atomic_store(slots + member, slot_of(cmn, cmx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (flags & MMQ_FLAG_TEST_HOOK) return 2;
    long long t0 = 0;
    int spins = 0;
    // windows of 4 members per lane (the tile's registers are live across the wait: 8 values in flight per lane cost the
    // K = 16 / 8 kernels a wave of occupancy); one window unless the group has more than 256 members
    for (int w0 = 0; w0 * 64 < Gs; w0 += 4) {
        const unsigned long long* p = slots + tid + 64 * w0;
        unsigned pend = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) pend |= (tid + 64 * (w0 + i) < Gs) ? (1u << i) : 0u;
        for (;; ++spins) {
            unsigned long long v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((pend >> i) & 1u) ? __hip_atomic_load(p + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (((pend >> i) & 1u) && v[i]) {
                    float u, w;
                    unpack_pair(~v[i], u, w);
                    tn = pmin(tn, u);
                    tx = pmax(tx, w);
                    pend &= ~(1u << i);
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
    }
    return 0;
}

// The slot meeting of k_mmq_group: `kk` pairs per member, slots [member][kk].  Lane (ch, j) = (tid / L, tid % L) watches channel
// ch of members j, j + L, ...; L = the largest power of two <= min(64, 256 / kk), so a channel's watchers sit in one wave
// and fold with shuffles.  Every wave polls on its own (no barrier inside the wait); a wave whose wait expired ORs 1 into
// *sh_code.  Writes the group's extrema of channel ch < nch to sh_mn / sh_mx (kk > 1), or returns the lane's share in tn / tx
// (whole_wg - mode 1, kk == 1: one channel, all 256 lanes watch).  The caller's barrier comes after.
template <int W>
__device__ __forceinline__ void slots_meet_group(const unsigned long long* slots, int Gs, int kk, int nch, bool whole_wg, long long timeout_ticks,
                                                 float* sh_mn, float* sh_mx, int* sh_code, float& tn, float& tx) {
    const int tid = threadIdx.x;
    int L = 1;
    while (L < 64 && 2 * L * kk <= TPB) L <<= 1;
    if (whole_wg) L = TPB;      // mode 1: one channel per group, the caller folds the 256 shares (kk == 1)
    const int ch = tid / L, j = tid - ch * L;
    const bool active = ch < nch;
    tn = INFINITY;
    tx = -INFINITY;
    long long t0 = 0;
    int spins = 0;
    for (int w0 = 0; w0 * L < Gs; w0 += W) {        // windows of W members per lane (one window unless Gs > W L)
        const unsigned long long* p = slots + (unsigned)((j + L * w0) * kk + ch);
        const unsigned step = (unsigned)(L * kk);
        unsigned pend = 0u;
#pragma unroll
        for (int i = 0; i < W; ++i) pend |= (active && j + L * (w0 + i) < Gs) ? (1u << i) : 0u;
        for (;; ++spins) {
            unsigned long long v[W];
#pragma unroll
            for (int i = 0; i < W; ++i)
                v[i] = ((pend >> i) & 1u) ? __hip_atomic_load(p + i * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
            for (int i = 0; i < W; ++i)
                if (((pend >> i) & 1u) && v[i]) {
                    float u, w;
                    unpack_pair(~v[i], u, w);
                    tn = pmin(tn, u);
                    tx = pmax(tx, w);
                    pend &= ~(1u << i);
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
    }
    if (!whole_wg) {
        for (int m = L >> 1; m >= 1; m >>= 1) { tn = pmin(tn, shfl_xor_f(tn, m)); tx = pmax(tx, shfl_xor_f(tx, m)); }
        if (active && j == 0) { sh_mn[ch] = tn; sh_mx[ch] = tx; }
    }
}

// ws: [0] status word, [256 ..) 16384 arrival/departure counters, one per group and per 256-byte line (a fixed
// region, so that no geometry's pairs ever land on another geometry's counters), then one 128-byte-aligned block of 8-byte
// {min, max} pairs per group: [member] (mode 1) or [member][k] (mode 2)
struct GWs {
    unsigned* status;
    unsigned* cnt;
    unsigned long long* part;
    unsigned long long* slots;   // the slot meeting's region (zero at rest), blocks laid out like `part`
    int gstride;   // pairs per group block
};

#ifndef FLAT_ABL
#define FLAT_ABL 0        // development builds only (tools/runs/r4_ablate.sh; timing, WRONG results): bit 0 - the stores of y compiled out,
                          // bit 1 - the meeting compiled out (every workgroup uses its own tile's extrema),
                          // bit 2 - the loads of k_mmq_flat compiled out (synthetic values), bit 3 - the Q/DQ arithmetic of
                          // k_mmq_flat compiled out (y = x): with bit 1 the kernel is the bare copy of its own address stream
#endif
#if FLAT_ABL & 1
#define FLAT_ABL_NOSTORE(o) && (o)[0] == 3.0e38f      // never true; the arithmetic stays
#else
#define FLAT_ABL_NOSTORE(o)
#endif
template <int A, int K, int OUT = 0, bool XR = false>
__global__ void __launch_bounds__(TPB, (K == 32 ? GRP_K32_WAVES : 1)) k_mmq_group(
    const float* __restrict__ x, float* __restrict__ y, const Geo g, const int Gs, const int num_bits, const int positive,
    const GWs ws, float* __restrict__ qp, float* __restrict__ mm, const unsigned flags, const XOut xo = XOut{},
    const XRank xr = XRank{}) {
    if constexpr (XR) xr_prologue(xr);                 // workgroup 0: the slots of the launch two back, the sequence mirror
    __shared__ float l_mn[TPB * A], l_mx[TPB * A];
    extern __shared__ unsigned cnnq_dyn_lds[];     // OUT == 1 with a histogram: 2^min(num_bits, 8) bins x HREP replicas, sized by the launch (xhist_lds_bytes)
    unsigned* const sh_hist = cnnq_dyn_lds;
    if constexpr (OUT == 1) {
        if (xo.hist) xhist_zero(sh_hist, 1 << (num_bits < 8 ? num_bits : 8));      // ordered before the first count by the barriers of the exchange
    }
    __shared__ float sh_mn[MAXCH], sh_mx[MAXCH], sh_sc[MAXCH], sh_zp[MAXCH], sh_rs[MAXCH];
    __shared__ int sh_timed_out, sh_slow;
    if (threadIdx.x == 0) sh_slow = 0;      // the barriers of the reduction and of the exchange come before its writers
    // has a wait expired on this workspace before?  (asked now, needed after the tile has landed)
    const unsigned st0 = threadIdx.x == 0 ? __hip_atomic_load(ws.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    GRP_STAMP(0);
    const RBlk rb = rblk_of(g, Gs);
    const Blk& b = rb.b;
    const int tid = threadIdx.x;
    const int col = b.col0 + tid;
    const bool ok = col < b.col1;
    const int colc = ok ? col : b.col0;   // idle lanes re-read the block's first column; results discarded
    const int nrows = b.n1 - b.n0;        // 1 .. K
    const size_t base = (size_t)b.n0 * (size_t)g.P + (size_t)colc * 4;

    // ---- the tile: K 16-byte loads per lane, issued back to back (rows past the tile re-read its last row)
    float v[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int r = j < nrows ? j : nrows - 1;
        ldv_nt<4>(x + base + (size_t)r * (size_t)g.P, v[j]);
    }
    // nothing that consumes a loaded value may be scheduled in between the loads (as in k_mmq_flat: the scheduler otherwise
    // starts folding after ~10 loads and issues the rest one by one as earlier ones land)
    __builtin_amdgcn_sched_barrier(0);
    GRP_STAMP(1);
    float mn[A], mx[A];
    bool nan = false;
#pragma unroll
    for (int a = 0; a < A; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
#pragma unroll
    for (int j = 0; j < K; ++j) lane_acc<A>(v[j], mn, mx, nan);
    if (A == 1 && nan) { mn[0] = NAN; mx[0] = NAN; }
    wg_channel_minmax<A>(g, b, ok, mn, mx, l_mn, l_mx, sh_mn, sh_mx);
    const int nch = b.c1 - b.c0;
    GRP_STAMP(2);

    // ---- publish this workgroup's pairs (write-through), arrive, wait for the group
    unsigned long long* blk = ws.part + (size_t)rb.group * ws.gstride;
    const int kk = (g.mode == 1) ? 1 : g.k;
    const bool use_slots = (flags & MMQ_FLAG_SLOTS) != 0;
    unsigned long long* slots = ws.slots + (size_t)rb.group * ws.gstride;    // the slot meeting's blocks: zero at rest
    if (use_slots) {
        // a pair IS the arrival: stored once, nobody waits for the acknowledgement (sh_mn / sh_mx are read here and
        // rewritten by the meeting: the barrier in between)
        for (int ch = tid; ch < nch; ch += TPB)
            __hip_atomic_store(slots + (size_t)rb.member * kk + ch, slot_of(sh_mn[ch], sh_mx[ch]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // bit 1: the test hook; bit 2: a wait has expired on this workspace before (thread 0 asked): the short timeout
        if (tid == 0) sh_timed_out = ((flags & MMQ_FLAG_TEST_HOOK) ? 2 : 0) | ((st0 & 1u) ? 4 : 0);
        __syncthreads();
        GRP_STAMP(3);
    } else {
    for (int ch = tid; ch < nch; ch += TPB)
        __hip_atomic_store(blk + (size_t)rb.member * kk + ch, pack_pair(sh_mn[ch], sh_mx[ch]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its pairs have left the CU
    __syncthreads();
    GRP_STAMP(3);
    }
    float sn = INFINITY, sx = -INFINITY;
#if FLAT_ABL & 2
    if (tid == 0) sh_timed_out = 0;
    __syncthreads();
    if (true) {
    } else
#else
    if (use_slots) {
        const int c0 = sh_timed_out;
        if (!(c0 & 2)) slots_meet_group<(K >= 32 ? 4 : 2)>(slots, Gs, kk, nch, g.mode == 1, (c0 & 4) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS, sh_mn, sh_mx, &sh_timed_out, sn, sx);
        __syncthreads();
        if (tid == 0) {
            const int code = sh_timed_out & 3;
            if (code) atomicOr(ws.status, (unsigned)code);
            GRP_STAMP(4);
        }
    } else {
    if (tid == 0) {
        const int timed_out = grp_meet(grp_lines(ws.cnt, rb.group, Gs, 0, 1), rb.member, Gs, flags, [] {},
                                       (st0 & 1u) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS);
        if (timed_out) atomicOr(ws.status, (unsigned)timed_out);   // bit 0: a wait expired, bit 1: the test hook
        sh_timed_out = timed_out;
        GRP_STAMP(4);
    }
    __syncthreads();
    }
#endif
    if (sh_timed_out & 3) {
        group_minmax_from_x<A>(x, g, b, l_mn, l_mx, sh_mn, sh_mx);
    } else if (use_slots) {
        if (g.mode == 1) {
            const float one_n[1] = {sn}, one_x[1] = {sx};
            wg_channel_minmax<1>(g, b, true, one_n, one_x, l_mn, l_mx, sh_mn, sh_mx);   // the one-channel reduction
        }
    } else if (g.mode == 1) {
        float tn = INFINITY, tx = -INFINITY;
        for (int m = tid; m < Gs; m += TPB) {
            float a, c;
            unpack_pair(__hip_atomic_load(blk + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, c);
            tn = pmin(tn, a);
            tx = pmax(tx, c);
        }
        const float one_n[1] = {tn}, one_x[1] = {tx};
        wg_channel_minmax<1>(g, b, true, one_n, one_x, l_mn, l_mx, sh_mn, sh_mx);   // the one-channel reduction
    } else {
        for (int ch = tid; ch < nch; ch += TPB) {
            float tn = INFINITY, tx = -INFINITY;
#pragma unroll 8
            for (int s = 0; s < Gs; ++s) {
                float a, c;
                unpack_pair(__hip_atomic_load(blk + ((size_t)s * kk + ch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, c);
                tn = pmin(tn, a);
                tx = pmax(tx, c);
            }
            sh_mn[ch] = tn;
            sh_mx[ch] = tx;
        }
        __syncthreads();
    }

    // ---- scale / zero point of the owned channels (iq.py:559-572), identical in every member
    const float qm = qmax_of(num_bits);
    for (int ch = tid; ch < nch; ch += TPB) {
        float cmn = sh_mn[ch], cmx = sh_mx[ch];
        if constexpr (XR) (void)xr_merge(xr, b.c0 + ch, rb.member == 0, cmn, cmx);   // the batch is sharded: every rank's extrema
        const float offset = positive ? 0.f : cmn;
        const float delta = cmx - offset;
        float sc = delta / qm;
        sc = (sc < 1e-8f) ? 1e-8f : sc;
        const float zp = zero_point_of(offset, sc);
        sh_sc[ch] = sc;
        sh_zp[ch] = zp;
        sh_rs[ch] = 1.0f / sc;
        if (!qdq_fast_domain(cmn, cmx, sc) || (flags & MMQ_FLAG_IEEE_DIVIDE)) sh_slow = 1;   // any writer, same value
        if (rb.member == 0) {
            const int c = b.c0 + ch;
            qp[(size_t)CNNQ_QP_SCALE * g.C + c] = sc;
            qp[(size_t)CNNQ_QP_ZP * g.C + c] = zp;
            qp[(size_t)CNNQ_QP_QMAX * g.C + c] = qm;
            if (mm) { mm[c] = cmn; mm[g.C + c] = cmx; }
        }
    }
    __syncthreads();
    GRP_STAMP(5);
    float sc[A], zp[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const unsigned e = (unsigned)colc * 4u + (unsigned)a;
        const int ch = (int)(e / (unsigned)g.HW) - b.c0;
        sc[a] = sh_sc[ch];
        zp[a] = sh_zp[ch];
    }

    // ---- Q/DQ out of the registers
    unsigned nzp[A];
#pragma unroll
    for (int a = 0; a < A; ++a) nzp[a] = 0u;
    if (!__builtin_amdgcn_readfirstlane(sh_slow)) {
        // every channel of the block inside qdq_fast_domain: the exact quotient without the divide
        float rs[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const unsigned e = (unsigned)colc * 4u + (unsigned)a;
            rs[a] = sh_rs[(int)(e / (unsigned)g.HW) - b.c0];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < nrows) {
                float o[4], cd[4];
                if constexpr (A == 1) {
                    qdq4_fast(v[j], sc[0], rs[0], zp[0], qm, o, cd);          // two elements per instruction, the same bits (as k_mmq_flat)
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = qdq1_fast(v[j][e], sc[e], rs[e], zp[e], qm, cd[e]);
                }
                if (ok FLAT_ABL_NOSTORE(o))
                    xstore<OUT, A>(xo, reinterpret_cast<char*>(y), xo.codes, xo.packed, (base + (size_t)j * (size_t)g.P) * 4, o,
                                   cd, sh_hist, zp, nzp);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < nrows) {
                float o[4], cd[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = qdq1(v[j][e], sc[A == 1 ? 0 : e], zp[A == 1 ? 0 : e], qm, cd[e]);
                if (ok FLAT_ABL_NOSTORE(o))
                    xstore<OUT, A>(xo, reinterpret_cast<char*>(y), xo.codes, xo.packed, (base + (size_t)j * (size_t)g.P) * 4, o,
                                   cd, sh_hist, zp, nzp);
            }
        }
    }
    if constexpr (OUT == 1) {
        if (xo.hist) xhist_flush<A>(sh_hist, xo.hist, 1 << (num_bits < 8 ? num_bits : 8), zp, nzp);
    }
    // ---- leave the group (after the stores are issued: the round trip hides behind them); the last departure of a
    //      counter line re-arms it for the next launch
    GRP_STAMP(6);
#if !(FLAT_ABL & 2)
    if (use_slots) {
        if (tid == 0) sh_timed_out = grp_depart_last(grp_lines(ws.cnt, rb.group, Gs, 0, 1), rb.member, Gs) ? 1 : 0;
        __syncthreads();
        if (sh_timed_out)      // the last member out re-arms the group's slots: every other member has read them
            for (int m = tid; m < Gs * kk; m += TPB) __hip_atomic_store(slots + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (tid == 0) grp_depart(grp_lines(ws.cnt, rb.group, Gs, 0, 1), rb.member, Gs);
#endif
#ifdef GRP_TRACE
    if (g_grp_trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // what s_endpgm waits for anyway: the tile's stores acknowledged
        __syncthreads();
        if (tid == 0) {
            g_grp_trace[(size_t)blockIdx.x * 16 + 8] = wall_clock64();
            g_grp_trace[(size_t)blockIdx.x * 16 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                                       (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
#endif
}

// ---- flat tiles: every lane busy when a channel row is not a multiple of the workgroup --------------------------------
// k_mmq_group cuts a channel row of H*W/4 float4 into nb pieces of <= 256 lanes: 56x56 (784 float4) gives 4 pieces of
// 196, 28x28 one piece of 196 - a quarter of the lanes (registers, VALU slots, load slots) idles on 70 % of ResNet-50's
// bytes.  Here a group is ONE channel and a member's tile is 256*K CONSECUTIVE float4 of the channel's flattened
// [N][H*W/4] space: lane t's j-th load is element f0 + 256 j + t, which lives in sample (f / cpc) at column (f % cpc).
// The walk is incremental and branch-free (a conditional around a load makes the compiler wait for the previous load):
// byte offsets relative to the tile's first sample, 32 bits (the plan bounds a tile's rows * plane size below 4 GB).
struct FGeo {
    int N, C, HW, P;
    unsigned cpc;         // float4 per channel row
    unsigned total;       // N * cpc: float4 per channel (< 2^31)
    int Gs;               // members (tiles) per channel
    unsigned q256, r16;   // 256 / cpc and (256 % cpc) * 16: one step of 256 float4 in rows and in bytes
    unsigned rs;          // bytes between two samples of the tensor (P * 4)
    int cb;               // dispatch order: blocks of cb adjacent channels, channel fastest inside a block (1: member fastest)
};

struct FWalk {
    unsigned ro, co;      // row offset and column offset, bytes
    __device__ __forceinline__ void step(const FGeo& g) {
        co += g.r16;
        ro += g.q256 * g.rs;
        const bool wrap = co >= g.cpc * 16u;
        co -= wrap ? g.cpc * 16u : 0u;
        ro += wrap ? g.rs : 0u;
    }
};

// one channel's extrema over the workgroup: wave shuffles, then the four wave results through LDS
__device__ __forceinline__ void wg_minmax1(float tn, float tx, float* l_mn, float* l_mx, float& cmn, float& cmx) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { tn = pmin(tn, shfl_xor_f(tn, m)); tx = pmax(tx, shfl_xor_f(tx, m)); }
    __syncthreads();                     // l_mn / l_mx may still be read from a previous call
    if (lane == 0) { l_mn[wv] = tn; l_mx[wv] = tx; }
    __syncthreads();
    cmn = pmin(pmin(l_mn[0], l_mn[1]), pmin(l_mn[2], l_mn[3]));
    cmx = pmax(pmax(l_mx[0], l_mx[1]), pmax(l_mx[2], l_mx[3]));
}

// OUT = 2 of k_mmq_flat: the tile's packed words (one 16-bit word per float4, in the flat order of the tile: word
// 256 j + t is lane t's step j) back out of LDS as 2 * PKL CONSECUTIVE bytes of the stream per lane, consecutive lanes
// consecutive pieces: a wave's store instruction covers 128 * PKL contiguous bytes (inside a row), like a store of y.
// Rows are whole groups of PKL float4 and sit on 2 * PKL byte boundaries of the stream (H*W % (4 * PKL) == 0 and an
// aligned buffer: the caller checked).
template <int K, int PKL, bool NT>
__device__ __forceinline__ void flat_pk_flush(const FGeo& g, const uint16_t* sh_pk, uint8_t* pbb, unsigned f0, unsigned n_first,
                                              unsigned lim) {
    static_assert(PKL == 8 || PKL == 4, "16- or 8-byte stores");
    const unsigned i0 = (unsigned)threadIdx.x * PKL;           // first word of this lane's piece in pass 0
    const unsigned u2 = f0 + i0;
    const unsigned n2 = u2 / g.cpc;
    unsigned ro2 = (n2 - n_first) * g.rs, co2 = (u2 - n2 * g.cpc) * 16u;
    constexpr unsigned stepf = 256u * PKL;                     // float4 between two passes
    const unsigned sq = stepf / g.cpc, sr16 = (stepf % g.cpc) * 16u;
#pragma unroll
    for (int r = 0; r < K / PKL; ++r) {
        const uint16_t* src = sh_pk + (r * stepf + i0);
        if constexpr (PKL == 8) {
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            const u4v q = *reinterpret_cast<const u4v*>(src);
            if (ro2 < lim) {
                if (NT) __builtin_nontemporal_store(q, reinterpret_cast<u4v*>(pbb + (ro2 + co2) / 8));
                else *reinterpret_cast<u4v*>(pbb + (ro2 + co2) / 8) = q;
            }
        } else {
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            const u2v q = *reinterpret_cast<const u2v*>(src);
            if (ro2 < lim) {
                if (NT) __builtin_nontemporal_store(q, reinterpret_cast<u2v*>(pbb + (ro2 + co2) / 8));
                else *reinterpret_cast<u2v*>(pbb + (ro2 + co2) / 8) = q;
            }
        }
        co2 += sr16;
        ro2 += sq * g.rs;
        const bool wrap = co2 >= g.cpc * 16u;
        co2 -= wrap ? g.cpc * 16u : 0u;
        ro2 += wrap ? g.rs : 0u;
    }
}

// KL (round 4, opt-in: CNNQ_FLAT_KL=8): KL more steps of the tile live in LDS - filled by LDS-DMA (global_load_lds_dwordx4: no
// staging registers), lane-linear, read back by the lane that owns them - so a workgroup holds 256 * (K + KL) float4 at
// the same register budget: 160 instead of 128 KB (three workgroups keep 96 of the CU's 160 KB of LDS busy).
template <int K, int OUT = 0, bool XR = false, int KL = 0>
__global__ void __launch_bounds__(TPB, (K == 32 ? GRP_K32_WAVES : 1)) k_mmq_flat(
    const float* __restrict__ x, float* __restrict__ y, const FGeo g, const int num_bits, const int positive, const GWs ws,
    float* __restrict__ qp, float* __restrict__ mm, const unsigned flags, const XOut xo = XOut{}, const XRank xr = XRank{}) {
    static_assert(TPB == 256, "wg_minmax1 folds four waves");
    if constexpr (XR) xr_prologue(xr);                 // workgroup 0: the slots of the launch two back, the sequence mirror
    __shared__ float l_mn[TPB / 64], l_mx[TPB / 64];
    extern __shared__ unsigned cnnq_dyn_lds[];     // OUT == 1 with a histogram: 2^min(num_bits, 8) bins x HREP replicas, sized by the launch (xhist_lds_bytes)
    unsigned* const sh_hist = cnnq_dyn_lds;
    if constexpr (OUT == 1) {
        if (xo.hist) xhist_zero(sh_hist, 1 << (num_bits < 8 ? num_bits : 8));      // ordered before the first count by the barriers of the exchange
    }
    __shared__ int sh_timed_out;
    // OUT = 2: the tile's packed nibbles go through a strip of LDS per wave - written as one 16-bit word per float4
    // ([step][lane], what a lane produces), read back as 16 (or 8) CONSECUTIVE bytes of the stream per lane and stored
    // with one dwordx4 (dwordx2) per lane: 4 (8) store instructions per lane instead of 32 two-byte ones
    __shared__ __attribute__((aligned(16))) uint16_t sh_pk[OUT == 2 ? TPB * (K + KL) : 1];
    __shared__ __attribute__((aligned(16))) float sh_x[KL ? KL * TPB * 4 : 4];
    const unsigned st0 = threadIdx.x == 0 ? __hip_atomic_load(ws.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    GRP_STAMP(0);
    const int tid = threadIdx.x;
    int c, member;
    if (g.cb <= 1) {
        c = (int)blockIdx.x / g.Gs;
        member = (int)blockIdx.x - c * g.Gs;
    } else {
        // consecutive workgroups hold the SAME member tile of cb adjacent channels: per sample they read one run of
        // cb channel rows together
        const int per = g.cb * g.Gs, blk = (int)blockIdx.x / per, r = (int)blockIdx.x - blk * per;
        const int c0 = blk * g.cb, cbl = min(g.cb, g.C - c0);
        member = r / cbl;
        c = c0 + (r - member * cbl);
    }
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
    uint8_t* cbb = (OUT == 1 && xo.codes) ? xo.codes + base / 4 : nullptr;     // the tile's bases of the other outputs
    uint8_t* pbb = (OUT == 2) ? xo.packed + base / 8 : nullptr;
    // lanes per wide packed store: 8 (16 bytes) when rows are whole groups of 8 float4 and sit on 16-byte boundaries of
    // the stream, 4 (8 bytes) for rows of whole groups of 4 (28x28), else 1 (the 2-byte store per float4)
    const int pkl = (OUT != 2 || (flags & MMQ_FLAG_PK_NARROW)) ? 1 : (g.cpc % 8u == 0u) ? 8 : (g.cpc % 4u == 0u) ? 4 : 1;

    // ---- the tile: K 16-byte loads per lane, back to back; past the end of the channel a lane re-reads the tile
    //      base's row start (an element of the same channel: harmless for the extrema, never stored)
    float v[K][4];
    FWalk w = w0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const unsigned off = w.ro < lim ? w.ro + w.co : 0u;
#if FLAT_ABL & 4
        for (int e = 0; e < 4; ++e) v[j][e] = (float)(int)((off >> 4) + (unsigned)e) * 1e-4f - 3.f;      // no loads: synthetic values
        (void)xb;
#else
        ldv_nt<4>(reinterpret_cast<const float*>(xb + off), v[j]);
#endif
        w.step(g);
    }
    if constexpr (KL > 0) {
        // the tile's LAST KL steps go straight into LDS: each wave's 64 x 16 bytes land lane-linear at its own 1 KB slot.  Round 6:
        // as asm behind the register loads (lds_dma16_behind, cnnq_common.hip.h) - with the builtin, which had to be issued
        // FIRST, the compiler waited for vmcnt(0) before the first use of any register step
#pragma unroll
        for (int l = 0; l < KL; ++l) {
            const unsigned off = w.ro < lim ? w.ro + w.co : 0u;
            lds_dma16_behind(xb, off, sh_x + (l * TPB + (tid & ~63)) * 4);
            w.step(g);
        }
    }
    // nothing that consumes a loaded value may be scheduled in between the loads (the scheduler otherwise folds the
    // first rows into the extrema while it still has loads to issue, and waits for them first)
    __builtin_amdgcn_sched_barrier(0);
    GRP_STAMP(1);
    float mn[1] = {INFINITY}, mx[1] = {-INFINITY};
    bool nan = false;
#pragma unroll
    for (int j = 0; j < K; ++j) lane_acc<1>(v[j], mn, mx, nan);
    if constexpr (KL > 0) {
        lds_dma_landed(mn[0]);
#pragma unroll
        for (int l = 0; l < KL; ++l) {
            float t[4];
            const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
            t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
            lane_acc<1>(t, mn, mx, nan);
        }
    }
    if (nan) { mn[0] = NAN; mx[0] = NAN; }
    float cmn, cmx;
    wg_minmax1(mn[0], mx[0], l_mn, l_mx, cmn, cmx);
    GRP_STAMP(2);

    // ---- publish, arrive, wait (the protocol of k_mmq_group; one pair per member)
    unsigned long long* blk = ws.part + (size_t)c * ws.gstride;
#if FLAT_ABL & 2
    if (tid == 0) sh_timed_out = 0;
    const float own_mn = cmn, own_mx = cmx;
#else
    const bool use_slots = (flags & MMQ_FLAG_SLOTS) != 0;
    unsigned long long* slots = ws.slots + (size_t)c * ws.gstride;   // zero at rest
    float sn = INFINITY, sx = -INFINITY;
    if (use_slots) {
        const long long tmo = (st0 & 1u) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS;      // lane 0's copy is the one consulted
        GRP_STAMP(3);
        const int timed_out = slots_meet(slots, member, g.Gs, cmn, cmx, flags, tmo, sn, sx);
        if (tid == 0) {
            if (timed_out) atomicOr(ws.status, (unsigned)timed_out);
            sh_timed_out = timed_out;
        }
        GRP_STAMP(4);
    } else if (tid == 0) {
        __hip_atomic_store(blk + member, pack_pair(cmn, cmx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pair has left the CU
        GRP_STAMP(3);
        const int timed_out = grp_meet(grp_lines(ws.cnt, c, g.Gs, 0, 1), member, g.Gs, flags, [] {},
                                       (st0 & 1u) ? GRP_TIMEOUT_SHORT : GRP_TIMEOUT_TICKS);
        if (timed_out) atomicOr(ws.status, (unsigned)timed_out);
        sh_timed_out = timed_out;
        GRP_STAMP(4);
    }
#endif
    __syncthreads();
    float tn = INFINITY, tx = -INFINITY;
    if (sh_timed_out) {
        // cold path: the channel's extrema straight from x, all samples
        bool nn = false;
        for (int s = 0; s < g.N; ++s)
            for (unsigned col = (unsigned)tid; col < g.cpc; col += TPB) {
                float t[4], a[1] = {tn}, b[1] = {tx};
                ldv<4>(x + (size_t)s * (size_t)g.P + (size_t)c * (size_t)g.HW + (size_t)col * 4, t);
                lane_acc<1>(t, a, b, nn);
                tn = a[0];
                tx = b[0];
            }
        if (nn) { tn = NAN; tx = NAN; }
    } else {
#if FLAT_ABL & 2
        tn = own_mn;
        tx = own_mx;
#else
        if (use_slots) {
            tn = sn;
            tx = sx;
        } else {
            for (int m = tid; m < g.Gs; m += TPB) {
                float a, b;
                unpack_pair(__hip_atomic_load(blk + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, b);
                tn = pmin(tn, a);
                tx = pmax(tx, b);
            }
        }
#endif
    }
    wg_minmax1(tn, tx, l_mn, l_mx, cmn, cmx);
    if constexpr (XR) {
        // the batch is sharded: one thread exchanges the channel's extrema with the other ranks (cnnq_xrank.hip.h)
        __shared__ float sh_x[2];
        if (tid == 0) {
            float a = cmn, b = cmx;
            (void)xr_merge(xr, c, member == 0, a, b);
            sh_x[0] = a;
            sh_x[1] = b;
        }
        __syncthreads();
        cmn = sh_x[0];
        cmx = sh_x[1];
    }

    // ---- scale / zero point (iq.py:559-572): every lane derives the same values from the same extrema
    const float qm = qmax_of(num_bits);
    const float offset = positive ? 0.f : cmn;
    const float delta = cmx - offset;
    float sc = delta / qm;
    sc = (sc < 1e-8f) ? 1e-8f : sc;
    const float zp = zero_point_of(offset, sc);
    const bool fast = qdq_fast_domain(cmn, cmx, sc) && !(flags & MMQ_FLAG_IEEE_DIVIDE);
    if (member == 0 && tid == 0) {
        qp[(size_t)CNNQ_QP_SCALE * g.C + c] = sc;
        qp[(size_t)CNNQ_QP_ZP * g.C + c] = zp;
        qp[(size_t)CNNQ_QP_QMAX * g.C + c] = qm;
        if (mm) { mm[c] = cmn; mm[g.C + c] = cmx; }
    }
    GRP_STAMP(5);

    // ---- Q/DQ out of the registers.  The walk is repeated, and hidden from the optimiser: it would otherwise keep the
    //      K offsets of the load phase alive across the meeting (K more registers than the tile leaves: spills)
    w = w0;
    asm volatile("" : "+v"(w.ro), "+v"(w.co));
    const float zpa[1] = {zp};
    unsigned nzp[1] = {0u};
    if (__builtin_amdgcn_readfirstlane((int)fast)) {
        // the channel's values are inside qdq_fast_domain: the exact quotient without the divide, parameters in
        // scalar registers (one channel per workgroup)
        const float s_sc = uniform_f(sc), s_rs = uniform_f(1.0f / sc), s_zp = uniform_f(zp);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float o[4], cd[4];
#if FLAT_ABL & 8
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = v[j][e]; cd[e] = 0.f; }
#else
            qdq4_fast(v[j], s_sc, s_rs, s_zp, qm, o, cd);
#endif
            if constexpr (OUT == 2) {
                if (pkl > 1) sh_pk[j * 256 + tid] = pack4_fast(cd);
                else if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
            } else {
                if (w.ro < lim FLAT_ABL_NOSTORE(o)) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
            }
            w.step(g);
        }
        if constexpr (KL > 0) {
#pragma unroll
            for (int l = 0; l < KL; ++l) {
                const float4 q = *reinterpret_cast<const float4*>(sh_x + (l * TPB + tid) * 4);
                const float t[4] = {q.x, q.y, q.z, q.w};
                float o[4], cd[4];
                qdq4_fast(t, s_sc, s_rs, s_zp, qm, o, cd);
                if constexpr (OUT == 2) {
                    if (pkl > 1) sh_pk[(K + l) * 256 + tid] = pack4_fast(cd);
                    else if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                } else {
                    if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                }
                w.step(g);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float o[4], cd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = qdq1(v[j][e], sc, zp, qm, cd[e]);
            if constexpr (OUT == 2) {
                if (pkl > 1) sh_pk[j * 256 + tid] = pack4_of(cd);
                else if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
            } else {
                if (w.ro < lim FLAT_ABL_NOSTORE(o)) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
            }
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
                if constexpr (OUT == 2) {
                    if (pkl > 1) sh_pk[(K + l) * 256 + tid] = pack4_of(cd);
                    else if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                } else {
                    if (w.ro < lim) xstore<OUT, 1>(xo, yb, cbb, pbb, w.ro + w.co, o, cd, sh_hist, zpa, nzp);
                }
                w.step(g);
            }
        }
    }
    if constexpr (OUT == 1) {
        if (xo.hist) xhist_flush<1>(sh_hist, xo.hist, 1 << (num_bits < 8 ? num_bits : 8), zpa, nzp);
    }
    if constexpr (OUT == 2) {
        if (pkl > 1) {
            __syncthreads();
            if (flags & MMQ_FLAG_PK_PLAIN) {
                if (pkl == 8) flat_pk_flush<K + KL, 8, false>(g, sh_pk, pbb, f0, n_first, lim);
                else flat_pk_flush<K + KL, 4, false>(g, sh_pk, pbb, f0, n_first, lim);
            } else {
                if (pkl == 8) flat_pk_flush<K + KL, 8, true>(g, sh_pk, pbb, f0, n_first, lim);
                else flat_pk_flush<K + KL, 4, true>(g, sh_pk, pbb, f0, n_first, lim);
            }
        }
    }
    GRP_STAMP(6);
#if !(FLAT_ABL & 2)
    if (use_slots) {
        if (tid == 0) sh_timed_out = grp_depart_last(grp_lines(ws.cnt, c, g.Gs, 0, 1), member, g.Gs) ? 1 : 0;
        __syncthreads();
        if (sh_timed_out)      // the last member out re-arms the group's slots: every other member has read them
            for (int m = tid; m < g.Gs; m += TPB) __hip_atomic_store(slots + m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (tid == 0) grp_depart(grp_lines(ws.cnt, c, g.Gs, 0, 1), member, g.Gs);
#endif
#ifdef GRP_TRACE
    if (g_grp_trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // what s_endpgm waits for anyway: the tile's stores acknowledged
        __syncthreads();
        if (tid == 0) {
            g_grp_trace[(size_t)blockIdx.x * 16 + 8] = wall_clock64();
            g_grp_trace[(size_t)blockIdx.x * 16 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                                       (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
#endif
}

// The rank-local extrema out[2][C] of a batch shard in ONE launch (the multi-GPU form of config 2 exchanges them
// between its statistics pass and its Q/DQ pass): the tiling and the pair blocks of k_mmq_group, but nobody waits - the
// LAST workgroup of a channel group to arrive folds the group's pairs and writes the channels' record.  Replaces
// k_minmax + k_minmax_reduce (one launch boundary less in front of every exchange).  Plain loads: the host uses it for
// tensors the Q/DQ pass will still find in the Infinity Cache (larger ones keep the streaming k_minmax).
template <int A, int K>
__global__ void __launch_bounds__(TPB) k_minmax_group(const float* __restrict__ x, const Geo g, const int Gs, const GWs ws,
                                                      float* __restrict__ out) {
    __shared__ float l_mn[TPB * A], l_mx[TPB * A];
    __shared__ float sh_mn[MAXCH], sh_mx[MAXCH];
    __shared__ int sh_last;
    const RBlk rb = rblk_of(g, Gs);
    const Blk& b = rb.b;
    const int tid = threadIdx.x;
    const int col = b.col0 + tid;
    const bool ok = col < b.col1;
    const int colc = ok ? col : b.col0;
    const int nrows = b.n1 - b.n0;
    const size_t base = (size_t)b.n0 * (size_t)g.P + (size_t)colc * 4;
    float v[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int r = j < nrows ? j : nrows - 1;
        ldv<4>(x + base + (size_t)r * (size_t)g.P, v[j]);
    }
    __builtin_amdgcn_sched_barrier(0);      // the whole tile in flight before the first value is consumed
    float mn[A], mx[A];
    bool nan = false;
#pragma unroll
    for (int a = 0; a < A; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
#pragma unroll
    for (int j = 0; j < K; ++j) lane_acc<A>(v[j], mn, mx, nan);
    if (A == 1 && nan) { mn[0] = NAN; mx[0] = NAN; }
    wg_channel_minmax<A>(g, b, ok, mn, mx, l_mn, l_mx, sh_mn, sh_mx);
    const int nch = b.c1 - b.c0;
    unsigned long long* blk = ws.part + (size_t)rb.group * ws.gstride;
    const int kk = (g.mode == 1) ? 1 : g.k;
    for (int ch = tid; ch < nch; ch += TPB)
        __hip_atomic_store(blk + (size_t)rb.member * kk + ch, pack_pair(sh_mn[ch], sh_mx[ch]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its pairs have left the CU
    __syncthreads();
    if (tid == 0) sh_last = grp_arrive_last(grp_lines(ws.cnt, rb.group, Gs, 0, 1), rb.member, Gs) ? 1 : 0;
    __syncthreads();
    if (!sh_last) return;
    if (g.mode == 1) {
        float tn = INFINITY, tx = -INFINITY;
        for (int m = tid; m < Gs; m += TPB) {
            float a, c;
            unpack_pair(__hip_atomic_load(blk + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, c);
            tn = pmin(tn, a);
            tx = pmax(tx, c);
        }
        const float one_n[1] = {tn}, one_x[1] = {tx};
        wg_channel_minmax<1>(g, b, true, one_n, one_x, l_mn, l_mx, sh_mn, sh_mx);
        if (tid == 0) { out[b.c0] = sh_mn[0]; out[g.C + b.c0] = sh_mx[0]; }
    } else {
        for (int ch = tid; ch < nch; ch += TPB) {
            float tn = INFINITY, tx = -INFINITY;
#pragma unroll 8
            for (int s = 0; s < Gs; ++s) {
                float a, c;
                unpack_pair(__hip_atomic_load(blk + ((size_t)s * kk + ch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, c);
                tn = pmin(tn, a);
                tx = pmax(tx, c);
            }
            out[b.c0 + ch] = tn;
            out[g.C + b.c0 + ch] = tx;
        }
    }
}

}  // namespace
