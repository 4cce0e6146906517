// cnnq_xrank.hip.h - config 2 in ONE launch and ONE read of x when the batch is sharded over W GPUs (opt-in; the default
// multi-GPU form is statistics pass -> RCCL all_gather -> Q/DQ pass, 12 bytes per element).  Part of the single
// translation unit cnnq_kernels.hip.
//
// The single-launch kernels (k_mmq_whole / k_mmq_group / k_mmq_flat) hold their tile of x in registers between the
// statistics and the Q/DQ.  With the batch sharded, a channel's extrema are the fold of the W ranks' extrema; the one
// thread that finishes a channel's LOCAL extrema
//   push   stores the COMPLEMENT of its {min, max} pair (NaNs canonical: never zero) into the slot
//          [parity][own rank][channel] of EVERY rank's window - fine-grained device memory exported with hipIpc and mapped
//          by all peers (cnnq_xrank_alloc / cnnq_xrank_open) - (only the workgroup that is member 0 of the channel's
//          group pushes; every member knows the same local extrema).  One 8-byte store per destination and nothing to
//          wait for: the pair IS the signal (round 4, as in the local slot meeting; round 3 sent pair -> drain -> sequence
//          number, one more remote round trip per launch);
//   wait   polls the W slots [parity][0..W)[channel] of its OWN window until none is zero, folds them in rank order
//          (NaN-propagating min / max: exact, the same bits on every rank and as on one GPU holding the whole batch), and
//          goes on to scale / zero point and the Q/DQ out of the registers.
// Nothing is pushed after something is waited for, so no rank can wait for a slot whose producer waits for it: the
// ranks need the co-residency the local exchange needs and nothing more.  Slots must be zero again before they are pushed
// to the next time.  A rank can push for launch s + 2 only after it completed launch s + 1, for which it needed this rank's
// push of launch s + 1, which this rank's launch s + 1 issued after its launch s had completed - so anything this rank's
// stream does up to the end of launch s is ahead of every push for launch s + 2.  Two ways to use that (four parities, the
// launch number modulo 4):
//   host numbering (round 5, the eager path): the launch number is a kernel argument; workgroup 0 of launch s zeroes the
//     slots launch s - 2 used (parity (s + 2) & 3; zero_c = that launch's channel count, from the host) - no reader of
//     this rank is left, no pusher can be there yet - and mirrors s into the device word.  ONE launch per tensor: the small
//     kernel of round 4 behind every launch was 4 us of dependent launch each, 0.2 ms of the 1.44 ms shard step;
//   device numbering (capturable): the launch's number is the device word + 1, and a one-workgroup kernel enqueued BEHIND
//     the launch (k_xr_finish: every reader of this rank is done by then) zeroes the launch's own slots and advances the
//     word.  A stream switches from the first to the second at its first captured launch and stays there (replays advance
//     the word, not the host's count); the first two launches after the switch still clean up behind the last two of the
//     host numbering (zero_c: redundant zeroing of slots nobody is using is harmless).  A wait gives up after `timeout` ticks of the 100 MHz clock: the channel's outputs are then NaN and bit 2 of
// the status word is raised (a peer that never launches would otherwise hang the device); once raised, later waits give up
// at once.  The host checks the word (periodically, without synchronising) and falls back to the collective.
#pragma once
#include "cnnq_common.hip.h"

namespace {

struct XRank {
    void* const* windows;      // [world] device pointers, the own window at [rank]; world == 0: no cross-rank stage
    int rank, world;
    unsigned seq;              // 1, 2, 3, ... (host-side numbering: seq_dev == nullptr)
    const unsigned* seq_dev;   // device-side numbering (round 4): the launch's number is *seq_dev + 1; k_xr_finish advances the
                               // word behind the launch, so a captured graph replays with fresh numbers
    unsigned* seq_mirror;      // host-side numbering: workgroup 0 stores seq here (the device word a later capture continues from)
    int zero_c;                // channels of the launch two back: workgroup 0 zeroes its slots (0: nothing to clean)
    int cmax;                  // channels a window holds per (parity, rank)
    unsigned* status;          // |= XR_STATUS_PEER_TIMEOUT
    long long timeout;         // ticks of the 100 MHz clock
};

constexpr unsigned XR_STATUS_PEER_TIMEOUT = 4u;

constexpr unsigned XR_PARITIES = 4u;
__host__ __device__ inline size_t xr_window_bytes(int world, int cmax) { return (size_t)XR_PARITIES * world * cmax * sizeof(unsigned long long); }

// the slot's content: the complement of {min, max} as two fp32 (NaNs canonical, so a published slot is never zero)
__device__ __forceinline__ unsigned long long xr_slot_of(float mn, float mx) {
    const bool nn = (mn != mn) || (mx != mx);
    const float a = nn ? NAN : mn, b = nn ? NAN : mx;
    return ~((unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32));
}

__device__ __forceinline__ unsigned long long* xr_slot(void* win, unsigned parity, int world, int cmax, int r, int c) {
    return reinterpret_cast<unsigned long long*>(win) + ((size_t)((int)parity * world + r) * (size_t)cmax + (size_t)c);
}

// One thread per channel: push the local extrema (if `push`), then wait for every rank's and fold them.  Returns false
// when a wait expired (mn / mx are NaN then and the status word is raised).
__device__ __forceinline__ bool xr_merge(const XRank& xr, int c, bool push, float& mn, float& mx) {
    // the word is only written by k_xr_finish, between launches of one stream: every thread of a launch reads the same value
    const unsigned seq = xr.seq_dev ? __hip_atomic_load(xr.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : xr.seq;
    const unsigned par = seq & (XR_PARITIES - 1u);
    if (push) {
        // The windows are uncached (fine-grained) memory: every access goes to the owner's memory - no release / acquire
        // fences, which at system scope write back and invalidate the whole L2 (measured: the b512 forward 2.3 ms slower)
        const unsigned long long v = xr_slot_of(mn, mx);
        for (int r = 0; r < xr.world; ++r)
            __hip_atomic_store(xr_slot(xr.windows[r], par, xr.world, xr.cmax, xr.rank, c), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    void* own = xr.windows[xr.rank];
    float a = INFINITY, b = -INFINITY;
    // one expired wait poisons every later one (a peer that is gone would otherwise cost `timeout` per channel)
    bool ok = !(__hip_atomic_load(xr.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & XR_STATUS_PEER_TIMEOUT);
    const long long t0 = wall_clock64();
    for (int r = 0; r < xr.world && ok; ++r) {
        const unsigned long long* s = xr_slot(own, par, xr.world, xr.cmax, r, c);
        unsigned long long v;
        int polls = 0;
        while ((v = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == 0ull) {
            if ((++polls & 31) == 0 && wall_clock64() - t0 > xr.timeout) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (ok) {
            v = ~v;
            const float p = __uint_as_float((unsigned)(v & 0xffffffffull)), q = __uint_as_float((unsigned)(v >> 32));
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

// Workgroup 0 of every exchanging launch, before anything else: the slots of the launch two back are zeroed (see the header of
// this file for why nobody can be using them), and with host numbering the device word follows the host's count.
__device__ __forceinline__ void xr_prologue(const XRank& xr) {
    if (blockIdx.x != 0) return;
    if (xr.zero_c > 0) {
        const unsigned seq = xr.seq_dev ? __hip_atomic_load(xr.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : xr.seq;
        const unsigned par = (seq + 2u) & (XR_PARITIES - 1u);
        void* own = xr.windows[xr.rank];
        for (int i = (int)threadIdx.x; i < xr.world * xr.zero_c; i += (int)blockDim.x) {
            const int r = i / xr.zero_c, c = i - r * xr.zero_c;
            __hip_atomic_store(xr_slot(own, par, xr.world, xr.cmax, r, c), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (xr.seq_mirror && threadIdx.x == 0) *xr.seq_mirror = xr.seq;      // nobody reads the word under host numbering
}

// device numbering: enqueued behind every exchanging launch (ONE workgroup): every reader of this rank is done, so the slots of
// the launch's parity are zeroed, and the device-side launch number advances
__global__ void __launch_bounds__(1024) k_xr_finish(void* const* windows, const int rank, const int world, const int cmax, const int C,
                                                    const unsigned seq, unsigned* seq_dev) {
    void* own = windows[rank];
    const unsigned par = (seq_dev ? *seq_dev + 1u : seq) & (XR_PARITIES - 1u);
    __syncthreads();                                   // everybody has read the word before thread 0 advances it
    for (int i = (int)threadIdx.x; i < world * C; i += (int)blockDim.x) {
        const int r = i / C, c = i - r * C;
        __hip_atomic_store(xr_slot(own, par, world, cmax, r, c), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (seq_dev && threadIdx.x == 0) *seq_dev += 1u;
}

// the exchange alone, for a rank whose shard has no single-launch kernel: mm[2][C] local extrema in, folded extrema out
__global__ void __launch_bounds__(TPB) k_xr_exchange(float* __restrict__ mm, const int C, const XRank xr) {
    xr_prologue(xr);
    const int c = (int)blockIdx.x * TPB + (int)threadIdx.x;
    if (c >= C) return;
    float mn = mm[c], mx = mm[C + c];
    (void)xr_merge(xr, c, true, mn, mx);
    mm[c] = mn;
    mm[C + c] = mx;
}

}  // namespace
