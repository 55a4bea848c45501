// Shared device/host helpers for the adm_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ADM_OK 0
#define ADM_EINVAL (-22)
#define ADM_ELAUNCH (-5)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ADM_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) return ADM_ELAUNCH;                \
  } while (0)

static inline int adm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + __expf(-u)); }
// d/du silu(u) = s (1 + u (1 - s)),  s = sigmoid(u)
__device__ __forceinline__ float silu_grad_f(float u) {
  float s = 1.0f / (1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Counter-based RNG for dropout: one 32-bit draw per (seed, element index); stateless so the
// backward pass regenerates the mask instead of storing it.
__device__ __forceinline__ uint32_t adm_hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float dropout_keep_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  // keep iff uniform >= p
  float u = (float)(adm_hash32(seed, idx) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}
