// Shared device/host helpers for the adm_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ADM_OK 0
#define ADM_EINVAL (-22)
#define ADM_ELAUNCH (-5)
// `splits` of the atomic weight-gradient entry points: 0 = chosen by the launcher (the workspace is cleared first);
// -1 = chosen by the launcher, the caller GUARANTEES the workspace is all zero on entry (zero-at-rest: adm_unpack_wgrad_table
// clears what it reads), so no memset is enqueued
#define ADM_SPLITS_AUTO_PREZEROED (-1)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "../../include/adm_hip.h"
#ifdef __HIPCC__
// bound vectors (ADM_AMAX_SLOTS x ADM_AMAX_STRIDE floats: include/adm_hip.h).  Producer side: the wave's maximum goes to the slot of this
// wave; consumer side: every lane reads one slot, wave maximum (all lanes return it).
__device__ __forceinline__ void adm_amax_commit(float am, float* amax) {
  if (!amax) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
  if ((threadIdx.x & 63) == 0 && am > 0.f) {
    const unsigned wave = (unsigned)blockIdx.x * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6) + (unsigned)blockIdx.y * 7u;
    atomicMax(reinterpret_cast<unsigned*>(amax + (wave % ADM_AMAX_SLOTS) * ADM_AMAX_STRIDE), __float_as_uint(am));
  }
}
__device__ __forceinline__ float adm_amax_read(const float* amax) {
  float v = amax[((threadIdx.x & 63) % ADM_AMAX_SLOTS) * ADM_AMAX_STRIDE];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif

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

// Counter-based RNG for dropout: stateless, so the backward pass regenerates the mask instead of storing it.  One call
// decides a whole channel QUAD (element index / 4): two 32-bit murmur-style finalisers give four 16-bit uniforms, each
// compared with p * 65536 (p = 0.1 -> 6554 / 65536, relative bias 6e-5).  32-bit multiplies only: the 64-bit-multiply
// hash this replaces cost ~48 quarter-rate v_mul per quad and made the GroupNorm kernels with dropout VALU-bound.
__device__ __forceinline__ uint32_t adm_mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ f32x4 dropout_keep4(uint64_t seed, uint64_t quad_idx, float p, float inv_keep) {
  const uint32_t q = (uint32_t)quad_idx, qh = (uint32_t)(quad_idx >> 32);
  const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  const uint32_t a = adm_mix32(q * 0x9E3779B1u + (uint32_t)seed + qh * 0x632BE5ABu);
  const uint32_t b = adm_mix32((q + 0x7F4A7C15u) * 0xB5297A4Du + (uint32_t)(seed >> 32) + qh);
  f32x4 r;
  r[0] = (a & 0xFFFFu) >= thr ? inv_keep : 0.0f;
  r[1] = (a >> 16) >= thr ? inv_keep : 0.0f;
  r[2] = (b & 0xFFFFu) >= thr ? inv_keep : 0.0f;
  r[3] = (b >> 16) >= thr ? inv_keep : 0.0f;
  return r;
}
