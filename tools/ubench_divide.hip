// tools/ubench_divide.hip - exhaustive device-side check of the divide-free exact quotient (measurement aid, not product
// code; the product form is qdq1_fast in csrc/cnnq_qdq.hip.h).  For a scale s with correctly rounded reciprocal r = 1 / s:
//   five operations:  q0 = x r; r0 = fma(-s, q0, x); q1 = fma(r0, r, q0); r1 = fma(-s, q1, x); q = fma(r1, r, q1)
//   three operations: q0 = x r; r0 = fma(-s, q0, x); q  = fma(r0, r, q0)
// The kernel runs ALL 2^32 bit patterns of x against `ns` scales and counts, for dividends inside the product's domain
// (|x| <= 2^70; NaN and larger values take the hardware divide in the product):
//   [0] / [1]  quotients of the five- / three-operation form that differ from x / s, for 2^-70 <= |x|,
//   [2] / [3]  (code, y) pairs that differ between qdq1 (divide, compare+select clamp) and the five- / three-operation
//              form with the v_med3 clamp, for every |x| <= 2^70 (zeros, denormals and tiny numbers included),
//   [4]        dividends examined.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/ubench_divide.hip -o tools/ubench_divide.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float qdq_ref(float x, float s, float zp, float qmax, float& code) {
    float q = x / s;
    q = q + zp;
    q = (q > qmax) ? qmax : q;
    q = (q < 0.f) ? 0.f : q;
    q = rintf(q);
    code = q;
    return (q - zp) * s;
}

__device__ __forceinline__ float finish(float q, float s, float zp, float qmax, float& code) {
    q = q + zp;
    q = __builtin_amdgcn_fmed3f(q, 0.f, qmax);
    q = rintf(q);
    code = q;
    return (q - zp) * s;
}

__global__ void __launch_bounds__(256) k_div(const float* __restrict__ scales, const float* __restrict__ zps, int ns,
                                             float qmax, unsigned long long* __restrict__ counts) {
    unsigned n5 = 0, n3 = 0, c5 = 0, c3 = 0, nx = 0;
    for (int half = 0; half < 2; ++half)                               // grid = 2^23 blocks x 2: every float
        for (int i = 0; i < ns; ++i) {
            const uint32_t xbits = ((uint32_t)half << 31) | (blockIdx.x * 256u + threadIdx.x);
            const float x = __uint_as_float(xbits);
            if (!(fabsf(x) <= 0x1p70f)) continue;
            const float s = scales[i], zp = zps[i];
            const float r = 1.0f / s;
            const float qt = x / s;
            float q = x * r;
            float e = fmaf(-s, q, x);
            q = fmaf(e, r, q);
            const float q3 = q;
            e = fmaf(-s, q, x);
            q = fmaf(e, r, q);
            ++nx;
            if (fabsf(x) >= 0x1p-70f) {
                n5 += (__float_as_uint(qt) != __float_as_uint(q)) ? 1u : 0u;
                n3 += (__float_as_uint(qt) != __float_as_uint(q3)) ? 1u : 0u;
            }
            float ct, cf, cg;
            const float yt = qdq_ref(x, s, zp, qmax, ct), yf = finish(q, s, zp, qmax, cf), yg = finish(q3, s, zp, qmax, cg);
            c5 += (__float_as_uint(ct) != __float_as_uint(cf) || __float_as_uint(yt) != __float_as_uint(yf)) ? 1u : 0u;
            c3 += (__float_as_uint(ct) != __float_as_uint(cg) || __float_as_uint(yt) != __float_as_uint(yg)) ? 1u : 0u;
        }
    if (n5) atomicAdd(&counts[0], (unsigned long long)n5);
    if (n3) atomicAdd(&counts[1], (unsigned long long)n3);
    if (c5) atomicAdd(&counts[2], (unsigned long long)c5);
    if (c3) atomicAdd(&counts[3], (unsigned long long)c3);
    atomicAdd(&counts[4], (unsigned long long)nx);
}

extern "C" int udivide(const float* scales, const float* zps, int ns, float qmax, unsigned long long* counts) {
    hipLaunchKernelGGL(k_div, dim3(1u << 23), dim3(256), 0, 0, scales, zps, ns, qmax, counts);
    return (int)hipDeviceSynchronize();
}
