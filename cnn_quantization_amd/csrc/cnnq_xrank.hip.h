// cnnq_xrank.hip.h - config 2 in ONE launch and ONE read of x when the batch is sharded over W GPUs (opt-in; the default
// multi-GPU form stays statistics pass -> RCCL all_gather -> Q/DQ pass, 12 bytes per element).  Part of the single
// translation unit cnnq_kernels.hip.
//
// The single-launch kernels (k_mmq_whole / k_mmq_group / k_mmq_flat) hold their tile of x in registers between the
// statistics and the Q/DQ.  With the batch sharded, a channel's extrema are the fold of the W ranks' extrema; the one
// thread that finishes a channel's LOCAL extrema
//   push   stores {min, max}, drains the stores, and then stores the launch's sequence number into the record
//          [parity][own rank][channel] of EVERY rank's window - fine-grained device memory exported with hipIpc and mapped
//          by all peers, the windows of cnnq_p2p.hip.h's kind (only the workgroup that is member 0 of the channel's
//          group pushes; every member knows the same local extrema);
//   wait   polls the W records [parity][0..W)[channel] of its OWN window until they carry the sequence number,
//          folds them in rank order (NaN-propagating min / max: exact, the same bits on every rank and as on one GPU
//          holding the whole batch), and goes on to scale / zero point and the Q/DQ out of the registers.
// Nothing is pushed after something is waited for, so no rank can wait for a record whose producer waits for it: the
// ranks need the co-residency the local exchange needs and nothing more.  Sequence numbers come from the host (one per
// launch, the same on every rank: the ranks issue the same launches in the same order on ONE stream each; round 4: or from a
// device word that the launch's last unit - or a one-thread kernel behind the launch - advances, which makes the launch
// capturable); two parities
// suffice because a rank can start launch s + 2 only after it has received every rank's records of launch s + 1, which
// a rank pushes only after its launch s has completed.  A wait gives up after `timeout` ticks of the 100 MHz clock:
// the channel's outputs are then NaN and bit 2 of the status word is raised (a peer that never launches would otherwise
// hang the device); once raised, later waits give up at once.  The host checks the word at its next synchronisation
// point and falls back to the collective.
#pragma once
#include "cnnq_common.hip.h"

namespace {

struct XRec {
    unsigned long long pair;   // {min, max} as two fp32
    unsigned seq;              // written last, after the pair has been acknowledged
    unsigned pad;
};

struct XRank {
    void* const* windows;      // [world] device pointers, the own window at [rank]; world == 0: no cross-rank stage
    int rank, world;
    unsigned seq;              // 1, 2, 3, ... (host-side numbering: seq_dev == nullptr)
    const unsigned* seq_dev;   // device-side numbering (round 4): the launch's number is *seq_dev + 1; k_xr_bump advances the
                               // word behind the launch, so a captured graph replays with fresh numbers
    unsigned* done;            // device-side numbering, in-kernel advance (round 4): seq_dev + 1, a count of finished units
                               // (workgroups / groups) that is zero between launches; whoever finishes last zeroes it and
                               // advances *seq_dev - no one-thread kernel behind the launch.  nullptr: k_xr_bump does it
    int cmax;                  // channels a window holds per (parity, rank)
    unsigned* status;          // |= XR_STATUS_PEER_TIMEOUT
    long long timeout;         // ticks of the 100 MHz clock
};

constexpr unsigned XR_STATUS_PEER_TIMEOUT = 4u;

__host__ __device__ inline size_t xr_window_bytes(int world, int cmax) { return (size_t)2 * world * cmax * sizeof(XRec); }

__device__ __forceinline__ unsigned long long xr_pack(float mn, float mx) {
    return (unsigned long long)__float_as_uint(mn) | ((unsigned long long)__float_as_uint(mx) << 32);
}

__device__ __forceinline__ XRec* xr_rec(void* win, const XRank& xr, int r, int c) {
    return reinterpret_cast<XRec*>(win) + ((size_t)((int)(xr.seq & 1u) * xr.world + r) * (size_t)xr.cmax + (size_t)c);
}

// One thread per channel: push the local extrema (if `push`), then wait for every rank's and fold them.  Returns false
// when a wait expired (mn / mx are NaN then and the status word is raised).
__device__ __forceinline__ bool xr_merge(const XRank& xr_in, int c, bool push, float& mn, float& mx) {
    XRank xr = xr_in;
    // the word is only written by k_xr_bump, between launches of one stream: every thread of a launch reads the same value
    if (xr.seq_dev) xr.seq = __hip_atomic_load(xr.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (push) {
        const unsigned long long pr = xr_pack(mn, mx);
        // The windows are uncached (fine-grained) memory: every access goes to the owner's memory, so ordering is a
        // matter of ISSUE order - no release / acquire fences, which at system scope write back and invalidate the
        // whole L2 (measured: the b512 forward 2.3 ms slower).  All pairs first, drained (the stores have been
        // acknowledged by their destinations), then the sequence numbers.
        for (int r = 0; r < xr.world; ++r)
            __hip_atomic_store(&xr_rec(xr.windows[r], xr, xr.rank, c)->pair, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int r = 0; r < xr.world; ++r)
            __hip_atomic_store(&xr_rec(xr.windows[r], xr, xr.rank, c)->seq, xr.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    void* own = xr.windows[xr.rank];
    float a = INFINITY, b = -INFINITY;
    // one expired wait poisons every later one (a peer that is gone would otherwise cost `timeout` per channel)
    bool ok = !(__hip_atomic_load(xr.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & XR_STATUS_PEER_TIMEOUT);
    const long long t0 = wall_clock64();
    for (int r = 0; r < xr.world && ok; ++r) {
        const XRec* s = xr_rec(own, xr, r, c);
        int polls = 0;
        while (__hip_atomic_load(&s->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != xr.seq) {
            if ((++polls & 31) == 0 && wall_clock64() - t0 > xr.timeout) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (ok) {
            // issued after the sequence number has been SEEN (the loop above consumed its value): the pair was in
            // the window before the number was
            const unsigned long long pr = __hip_atomic_load(&s->pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float p = __uint_as_float((unsigned)(pr & 0xffffffffull)), q = __uint_as_float((unsigned)(pr >> 32));
            a = pmin(a, p);
            b = pmax(b, q);
        }
    }
    if (!ok) {
        atomicOr(xr.status, XR_STATUS_PEER_TIMEOUT);
        a = NAN;
        b = NAN;
    }
    mn = a;
    mx = b;
    return ok;
}

// device-side sequence numbers: one thread, enqueued behind an exchanging launch whose kernels cannot advance the word
// themselves (the two-pass form, the counter meeting)
__global__ void k_xr_bump(unsigned* seq_dev) { *seq_dev += 1u; }

// in-kernel advance: called by ONE thread of every unit (a workgroup of k_mmq_whole, the last member out of a group of
// k_mmq_group / k_mmq_flat) after the unit's last xr_merge; `units` of them per launch.  Every reader of *seq_dev belongs to
// a unit that has not reported yet, so the word changes only after its last reader of this launch.
__device__ __forceinline__ void xr_unit_done(const XRank& xr, unsigned units) {
    if (!xr.done) return;
    if (__hip_atomic_fetch_add(xr.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == units - 1u) {
        __hip_atomic_store(xr.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(const_cast<unsigned*>(xr.seq_dev), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the exchange alone, for a rank whose shard has no single-launch kernel: mm[2][C] local extrema in, folded extrema out
__global__ void __launch_bounds__(TPB) k_xr_exchange(float* __restrict__ mm, const int C, const XRank xr) {
    const int c = (int)blockIdx.x * TPB + (int)threadIdx.x;
    if (c >= C) return;
    float mn = mm[c], mx = mm[C + c];
    (void)xr_merge(xr, c, true, mn, mx);
    mm[c] = mn;
    mm[C + c] = mx;
}

}  // namespace
