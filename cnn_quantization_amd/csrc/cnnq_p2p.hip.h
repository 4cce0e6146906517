// cnnq_p2p.hip.h - one-shot all-gather of small per-channel records over xGMI peer-to-peer stores (opt-in
// alternative to the RCCL all_gather of the statistics exchange; see distributed.P2PExchange).
// Part of the single translation unit cnnq_kernels.hip.
//
// Every rank owns a window (fine-grained, uncached device memory, exported with hipIpc and mapped by all
// peers):   float slots[2][W][slot_floats];  uint32 flags[2][W] (one 64-byte line each)
// Exchange number `seq` (1, 2, 3, ...; parity p = seq & 1):
//   post  rank r writes its record into slots[p][r] of EVERY rank's window (remote stores), fences at system
//         scope and then release-stores flags[p][r] = seq there;
//   wait  rank r acquire-spins on its OWN flags[p][0..W) until all carry seq, then copies slots[p][*] out
// - both in ONE launch of W workgroups (workgroup b posts to rank b, then waits for rank b).
// Two parities suffice: a rank posts seq + 2 only after it consumed seq + 1, which every rank posted only after
// consuming seq.  A spin gives up after P2P_TIMEOUT_TICKS of the constant 100 MHz clock and reports through
// `status` instead of hanging the device; once set, later exchanges return NaN records immediately.
#pragma once
#include "cnnq_common.hip.h"

namespace {

constexpr int P2P_FLAG_STRIDE = 16;                 // uint32 per flag line (64 B)
constexpr long long P2P_TIMEOUT_TICKS = 200000000;  // 2 s at 100 MHz

__host__ __device__ inline size_t p2p_window_bytes(int world, int slot_floats) {
    return (size_t)2 * world * slot_floats * sizeof(float) + (size_t)2 * world * P2P_FLAG_STRIDE * sizeof(uint32_t);
}
__device__ __forceinline__ float* p2p_slot(void* win, int world, int slot_floats, int parity, int r) {
    return reinterpret_cast<float*>(win) + ((size_t)parity * world + r) * slot_floats;
}
__device__ __forceinline__ uint32_t* p2p_flag(void* win, int world, int slot_floats, int parity, int r) {
    uint32_t* f = reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(win) + (size_t)2 * world * slot_floats);
    return f + ((size_t)parity * world + r) * P2P_FLAG_STRIDE;
}

// grid = world: workgroup b writes this rank's record into rank b's window (post), then waits for rank b's
// record in the own window and copies it out.  out[world][nfloat]; status[0] |= 1 on timeout.
__global__ void __launch_bounds__(TPB) k_p2p_all_gather(const float* __restrict__ rec, int nfloat,
                                                        void* const* __restrict__ windows, int rank, int world,
                                                        int slot_floats, uint32_t seq, float* __restrict__ out,
                                                        int* __restrict__ status) {
    __shared__ int ok;
    const int peer = blockIdx.x, parity = (int)(seq & 1u);
    void* remote = windows[peer];
    float* dst = p2p_slot(remote, world, slot_floats, parity, rank);
    for (int i = threadIdx.x; i < nfloat; i += TPB) __hip_atomic_store(dst + i, rec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    void* own = windows[rank];
    if (threadIdx.x == 0) {
        __hip_atomic_store(p2p_flag(remote, world, slot_floats, parity, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t* f = p2p_flag(own, world, slot_floats, parity, peer);
        const long long t0 = wall_clock64();
        int good = (*reinterpret_cast<volatile int*>(status) & 1) ? 0 : 1;   // one timeout poisons all later exchanges
        while (good && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > P2P_TIMEOUT_TICKS) { good = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        ok = good;
        if (!good) atomicOr(status, 1);
    }
    __syncthreads();
    const float* src = p2p_slot(own, world, slot_floats, parity, peer);
    float* o = out + (size_t)peer * nfloat;
    for (int i = threadIdx.x; i < nfloat; i += TPB)
        o[i] = ok ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __builtin_nanf("");
}

}  // namespace
