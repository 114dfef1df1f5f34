// mireg_common.h -- device-side helpers shared by the gfx950 kernels.
// wave = 64 lanes everywhere (CDNA4); no 32-lane assumptions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MIREG_OK 0
#define MIREG_ERR_ARG (-1)      // bad shape / stride / null pointer
#define MIREG_ERR_LAUNCH (-2)   // hipLaunch / hipGetLastError failure
#define MIREG_ERR_UNSUPPORTED (-3)

#define MIREG_CHECK_ARG(cond) do { if (!(cond)) return MIREG_ERR_ARG; } while (0)
#define MIREG_LAUNCH_RET() do { return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH; } while (0)

namespace mireg {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum of N per-thread floats; result valid in thread 0.
// red must hold N * (blockDim.x / 64) floats of LDS.
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[wid * N + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] += red[w * N + i];
    }
  }
}

// q = i / d, rem = i % d for a non-negative flat index: ONE 32-bit division when both fit 32 bits (every launch of this library does), the
// 64-bit pair otherwise.  A 64-bit division is ~100 vector instructions; per-element index splits with two or three of them made the
// elementwise / resize / tail kernels ALU-bound.
__device__ __forceinline__ long fast_divmod(long i, long d, long& rem) {
  if ((((unsigned long)i | (unsigned long)d) >> 32) == 0) {
    const unsigned q = (unsigned)i / (unsigned)d;
    rem = (long)((unsigned)i - q * (unsigned)d);
    return (long)q;
  }
  const long q = i / d;
  rem = i - q * d;
  return q;
}

// Charbonnier penalty (x^2 + eps^2)^0.25 with eps = 1e-9 (reference loss.py:33-35).  The argument of every root here lies in
// [1e-18, 1e18]: normal floats, so the bare v_sqrt_f32 / v_rsq_f32 (1 ulp) replace sqrtf / rsqrtf, whose denormal scaling and
// refinement steps cost ~25 vector instructions each and made the smoothness kernels ALU-bound (47 -> ~8 instructions per term).
__device__ __forceinline__ float charb(float d) { return __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(d * d + 1e-18f)); }
// d/dd of the above: 0.5 * d * (d^2 + eps^2)^(-0.75)
__device__ __forceinline__ float charb_grad(float d) {
  const float t = d * d + 1e-18f;
  const float r = __builtin_amdgcn_rsqf(t);                 // t^-0.5
  return 0.5f * d * r * __builtin_amdgcn_sqrtf(r);          // t^-0.75
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

}  // namespace mireg
