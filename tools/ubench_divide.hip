// tools/ubench_divide.hip - is a three-instruction divide exact enough for the Q/DQ?  (measurement aid, not product code)
// For a scale s with correctly rounded reciprocal r = 1 / s:  q0 = x * r;  e = fma(-q0, s, x);  q1 = fma(e, r, q0)
// (Markstein's quotient refinement); an infinite or zero q0 stays (the refinement would turn +-inf into NaN, -0 into +0).  The kernel runs ALL 2^32 bit patterns of x against `ns` scales and counts
//   [0] quotients q1 != x / s (bitwise, NaN == NaN),  [1] integer codes that differ:
//       rint(clamp(q + zp, 0, qmax)) with the IEEE quotient vs with q1  (iq.py:573-590),
//   [3] quotient mismatches whose IEEE quotient is a normal number (the rest sit in the underflow range),
//   [2] mid-tread outputs clamp(rint(q), lo, hi) * s that differ (bitwise, signed zeros included).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/ubench_divide.hip -o tools/ubench_divide.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float code_of(float q, float zp, float qmax) {
    q = q + zp;
    q = (q > qmax) ? qmax : q;
    q = (q < 0.f) ? 0.f : q;
    return rintf(q);
}

__global__ void __launch_bounds__(256) k_div(const float* __restrict__ scales, const float* __restrict__ zps, int ns,
                                             float qmax, unsigned long long* __restrict__ counts) {
    unsigned nq = 0, nc = 0, ny = 0, nn = 0;
    for (int half = 0; half < 2; ++half)                               // grid = 2^23 blocks x 2: every float
    for (int i = 0; i < ns; ++i) {
        const uint32_t xbits = ((uint32_t)half << 31) | (blockIdx.x * 256u + threadIdx.x);
        const float x = __uint_as_float(xbits);
        const float s = scales[i], zp = zps[i];
        const float r = 1.0f / s;
        const float qt = x / s;
        const float q0 = x * r;
        const float e = fmaf(-q0, s, x);
        float q1 = fmaf(e, r, q0);
        q1 = __builtin_amdgcn_classf(q0, 0x264) ? q0 : q1;          // +-inf would refine to NaN, -0 to +0: keep q0
        const bool same_q = (__float_as_uint(qt) == __float_as_uint(q1)) || (qt != qt && q1 != q1);
        nq += same_q ? 0u : 1u;
        nn += (!same_q && fabsf(qt) >= 1.1754944e-38f) ? 1u : 0u;     // ... of which with a NORMAL IEEE quotient
        const float ct = code_of(qt, zp, qmax), cf = code_of(q1, zp, qmax);
        const bool same_c = (__float_as_uint(ct) == __float_as_uint(cf)) || (ct != ct && cf != cf);
        nc += same_c ? 0u : 1u;
        // [2]: the mid-tread form (iq.py:202-224): t = clamp(rint(q), lo, hi) with float bounds, y = t * s
        const float lo = -zp - 0.37f, hi = qmax - zp + 0.61f;
        float tt = rintf(qt), tf = rintf(q1);
        tt = (tt < hi || tt != tt) ? tt : hi; tt = (tt > lo || tt != tt) ? tt : lo;
        tf = (tf < hi || tf != tf) ? tf : hi; tf = (tf > lo || tf != tf) ? tf : lo;
        const float yt = tt * s, yf = tf * s;
        const bool same_y = (__float_as_uint(yt) == __float_as_uint(yf)) || (yt != yt && yf != yf);
        ny += same_y ? 0u : 1u;
    }
    if (nq) atomicAdd(&counts[0], (unsigned long long)nq);
    if (nc) atomicAdd(&counts[1], (unsigned long long)nc);
    if (ny) atomicAdd(&counts[2], (unsigned long long)ny);
    if (nn) atomicAdd(&counts[3], (unsigned long long)nn);
}

extern "C" int udivide(const float* scales, const float* zps, int ns, float qmax, unsigned long long* counts) {
    hipLaunchKernelGGL(k_div, dim3(1u << 23), dim3(256), 0, 0, scales, zps, ns, qmax, counts);
    return (int)hipDeviceSynchronize();
}
