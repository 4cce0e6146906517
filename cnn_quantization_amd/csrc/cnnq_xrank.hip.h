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
//
// Round 6: the same windows carry SUMS (configs 3 / 4 / 5 sharded: k_fused_flat / k_fused_group / k_stats_flat with the cross-rank
// stage).  A slot is one 8-byte word whatever it holds; a launch addresses its slots as [word][channel] (word * C + channel: up
// to 8 C slots for the two phases of the statistics kernel), a sum travels as the complement of its fp64 bits (xr_merge_sum) and
// the W ranks' words are added in RANK order by every reader: the same bits on every rank.  And the clean-up no longer depends
// on what the host remembers: workgroup 0 of a launch records the number of slots the launch uses in the device array
// cdev[parity] and zeroes the slots cdev[(parity + 2) & 3] recorded two launches back - eager and captured launches, replays in
// any order and a switch from host to device numbering all leave the same trail.
// The sums layout (one launch NUMBER may span several kernels - the first one runs the prologue, the others carry no_prologue):
//   word 0 {min, max} pair   1 sum   2 sum of squares   3 / 4 the sums of relu(x)   5 the rank's element count
//   word 6 sum |x - mean|    7 sum ((x - mean) / std)^4
// Words 0-5 make the pass-A record global (k_stats_flat's first meeting on a flat-tile shard; k_xr_moments behind k_moments
// otherwise and for configs 3 / 5, whose pass A is a launch of its own), words 6-7 pass B's (k_fused_* take word 6 only).  No
// collective is left on these paths: a sharded config 3 is k_moments, k_xr_moments, k_bitalloc, k_fused - the launches of one GPU.
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
    int zero_c;                // slots of the launch two back: workgroup 0 zeroes them (0: nothing to clean); only without cdev
    unsigned* cdev;            // [4] slots in use per parity, kept by the launches themselves (round 6; null: zero_c from the host)
    int nslots;                // slots this launch uses per rank (C for the extrema; 8 C for the launches that carry sums)
    int slot0;                 // first slot of the fused kernels' sum (cnnq_aciq.hip.h: word 6 of the sums layout, 6 C)
    int no_prologue;           // this kernel is not the first of its launch number: an earlier kernel of the sequence cleaned up
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

// a partial sum in a slot: the complement of its bits, NaNs made canonical first (a stored word is never zero: zero
// means "not arrived"; the sum 0.0 of a dead channel becomes all ones)
__device__ __forceinline__ unsigned long long slot_of_sum(double s) {
    const unsigned long long b = (s != s) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(s);
    return ~b;
}
__device__ __forceinline__ double sum_of_slot(unsigned long long v) { return __longlong_as_double((long long)~v); }

__device__ __forceinline__ unsigned long long* xr_slot(void* win, unsigned parity, int world, int cmax, int r, int c) {
    return reinterpret_cast<unsigned long long*>(win) + ((size_t)((int)parity * world + r) * (size_t)cmax + (size_t)c);
}

// the launch's number: the host's, or the device word + 1 (written only between launches of one stream: every thread of a launch
// reads the same value)
__device__ __forceinline__ unsigned xr_seq(const XRank& xr) {
    return xr.seq_dev ? __hip_atomic_load(xr.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : xr.seq;
}

// One thread per slot: push this rank's word (if `push`) into slot s of every rank's window, wait for the W words of the own
// window and fold them in rank order - `bits` in and out: a {min, max} pair as two fp32 (is_pair: NaN-propagating min / max,
// exact) or the fp64 bits of a partial sum (added in rank order: the same total on every rank, bit for bit).  What travels is
// the complement, NaNs canonical first, so a published slot is never zero.  Returns false when a wait expired (the result is
// NaN then and the status word is raised).  Lanes of one wave may call it together with different slots and kinds (one loop).
__device__ __forceinline__ bool xr_merge_word(const XRank& xr, int s, bool push, bool is_pair, unsigned long long& bits) {
    const unsigned par = xr_seq(xr) & (XR_PARITIES - 1u);
    if (push) {
        // The windows are uncached (fine-grained) memory: every access goes to the owner's memory - no release / acquire
        // fences, which at system scope write back and invalidate the whole L2 (measured: the b512 forward 2.3 ms slower)
        unsigned long long v;
        if (is_pair) v = xr_slot_of(__uint_as_float((unsigned)(bits & 0xffffffffull)), __uint_as_float((unsigned)(bits >> 32)));
        else v = slot_of_sum(__longlong_as_double((long long)bits));
        for (int r = 0; r < xr.world; ++r)
            __hip_atomic_store(xr_slot(xr.windows[r], par, xr.world, xr.cmax, xr.rank, s), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    void* own = xr.windows[xr.rank];
    float a = INFINITY, b = -INFINITY;
    double acc = 0.;
    // one expired wait poisons every later one (a peer that is gone would otherwise cost `timeout` per channel)
    bool ok = !(__hip_atomic_load(xr.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & XR_STATUS_PEER_TIMEOUT);
    const long long t0 = wall_clock64();
    for (int r = 0; r < xr.world && ok; ++r) {
        const unsigned long long* p = xr_slot(own, par, xr.world, xr.cmax, r, s);
        unsigned long long v;
        int polls = 0;
        while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == 0ull) {
            if ((++polls & 31) == 0 && wall_clock64() - t0 > xr.timeout) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (ok) {
            v = ~v;
            if (is_pair) {
                a = pmin(a, __uint_as_float((unsigned)(v & 0xffffffffull)));
                b = pmax(b, __uint_as_float((unsigned)(v >> 32)));
            } else {
                acc += __longlong_as_double((long long)v);      // rank order, whatever the arrival order was
            }
        }
    }
    if (!ok) {
        atomicOr(xr.status, XR_STATUS_PEER_TIMEOUT);
        a = NAN;
        b = NAN;
        acc = NAN;
    }
    bits = is_pair ? ((unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32))
                   : (unsigned long long)__double_as_longlong(acc);
    return ok;
}

// the extrema of a channel (config 2): push the local pair, fold every rank's
__device__ __forceinline__ bool xr_merge(const XRank& xr, int c, bool push, float& mn, float& mx) {
    unsigned long long bits = (unsigned long long)__float_as_uint(mn) | ((unsigned long long)__float_as_uint(mx) << 32);
    const bool ok = xr_merge_word(xr, c, push, true, bits);
    mn = __uint_as_float((unsigned)(bits & 0xffffffffull));
    mx = __uint_as_float((unsigned)(bits >> 32));
    return ok;
}

// a partial sum (round 6: configs 3 / 4 / 5): push this rank's, add every rank's in rank order
__device__ __forceinline__ bool xr_merge_sum(const XRank& xr, int s, bool push, double& sum) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(sum);
    const bool ok = xr_merge_word(xr, s, push, false, bits);
    sum = __longlong_as_double((long long)bits);
    return ok;
}

// Workgroup 0 of every exchanging launch, before anything else: the slots of the launch two back are zeroed (see the header of
// this file for why nobody can be using them), and with host numbering the device word follows the host's count.  With `cdev`
// the launches keep the books themselves: cdev[parity] = slots in use; this launch zeroes what cdev[(parity + 2) & 3] says, clears
// that entry and records its own count - whatever the host captured, replayed or ran eagerly in between.
__device__ __forceinline__ void xr_prologue(const XRank& xr) {
    if (blockIdx.x != 0 || xr.no_prologue) return;
    const unsigned seq = xr_seq(xr);
    const unsigned par = (seq + 2u) & (XR_PARITIES - 1u);
    const int zc = xr.cdev ? (int)__hip_atomic_load(xr.cdev + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : xr.zero_c;
    if (zc > 0) {
        void* own = xr.windows[xr.rank];
        for (int i = (int)threadIdx.x; i < xr.world * zc; i += (int)blockDim.x) {
            const int r = i / zc, c = i - r * zc;
            __hip_atomic_store(xr_slot(own, par, xr.world, xr.cmax, r, c), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (xr.cdev) {
        __syncthreads();                               // every thread of workgroup 0 has read the count before it is cleared
        if (threadIdx.x == 0) {
            xr.cdev[par] = 0u;
            xr.cdev[seq & (XR_PARITIES - 1u)] = (unsigned)xr.nslots;
        }
    }
    if (xr.seq_mirror && threadIdx.x == 0) *xr.seq_mirror = xr.seq;      // nobody reads the word under host numbering
}

// The same for a kernel whose EVERY workgroup can lend a hand (k_xr_moments: the launches that carry sums use eight slots per
// channel, and 16 K uncached stores from one workgroup were 5-10 us of a 14 us kernel): each workgroup zeroes its slice of the
// slots of the launch two back.  All of them read the same count - nobody writes that entry during this launch (its next writer
// is workgroup 0 of the launch after next, recording its own slots) - so it is not cleared here.
__device__ __forceinline__ void xr_prologue_all(const XRank& xr) {
    if (xr.no_prologue) return;
    const unsigned seq = xr_seq(xr);
    const unsigned par = (seq + 2u) & (XR_PARITIES - 1u);
    const int zc = xr.cdev ? (int)__hip_atomic_load(xr.cdev + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : xr.zero_c;
    const int total = xr.world * zc;
    if (total > 0) {
        void* own = xr.windows[xr.rank];
        const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
        const int i0 = (int)blockIdx.x * per, i1 = min(total, i0 + per);
        for (int i = i0 + (int)threadIdx.x; i < i1; i += (int)blockDim.x) {
            const int r = i / zc, c = i - r * zc;
            __hip_atomic_store(xr_slot(own, par, xr.world, xr.cmax, r, c), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (xr.cdev) xr.cdev[seq & (XR_PARITIES - 1u)] = (unsigned)xr.nslots;
        if (xr.seq_mirror) *xr.seq_mirror = xr.seq;
    }
}

// device numbering: enqueued behind every exchanging launch (ONE workgroup): every reader of this rank is done, so the slots of
// the launch's parity are zeroed (nslots per rank), its cdev entry cleared, and the device-side launch number advances
__global__ void __launch_bounds__(1024) k_xr_finish(void* const* windows, const int rank, const int world, const int cmax, const int C,
                                                    const unsigned seq, unsigned* seq_dev, unsigned* cdev) {
    void* own = windows[rank];
    const unsigned par = (seq_dev ? *seq_dev + 1u : seq) & (XR_PARITIES - 1u);
    __syncthreads();                                   // everybody has read the word before thread 0 advances it
    for (int i = (int)threadIdx.x; i < world * C; i += (int)blockDim.x) {
        const int r = i / C, c = i - r * C;
        __hip_atomic_store(xr_slot(own, par, world, cmax, r, c), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) {
        if (cdev) cdev[par] = 0u;
        if (seq_dev) *seq_dev += 1u;
    }
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
