// Diagnostic (not part of the product): sustained fp32-input MFMA rate of this device, and what each kind of
// co-issued instruction costs it.  Build+run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: MFMA only.  1: + 16 ds_read_b128 per 64 MFMA (operands from LDS).  2: mode 1 + one barrier per 64 MFMA.
// 3: mode 0 + 64 independent v_fma per 64 MFMA (VALU contention probe).
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
  for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) lds[i] = a0 + i * 1e-7f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i);
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  const float* Ab = lds + lr * 36 + lh * 4;
  const float* Bb = lds + 128 * 36 + lr * 36 + lh * 4;
  float v0 = a0, v1 = b0, v2 = a0 + 1, v3 = b0 + 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[2], b[2];
      if (MODE == 1 || MODE == 2) {
        a[0] = *reinterpret_cast<const f32x4*>(Ab + g * 8); a[1] = *reinterpret_cast<const f32x4*>(Ab + 32 * 36 + g * 8);
        b[0] = *reinterpret_cast<const f32x4*>(Bb + g * 8); b[1] = *reinterpret_cast<const f32x4*>(Bb + 32 * 36 + g * 8);
      } else {
        a[0] = a[1] = f32x4{a0, a0, a0, a0}; b[0] = b[1] = f32x4{b0, b0, b0, b0};
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i * 2 + j], 0, 0, 0);
            if (MODE == 3) { v0 = __builtin_fmaf(v0, v1, v2); v1 = __builtin_fmaf(v1, v2, v3); v2 = __builtin_fmaf(v2, v3, v0); v3 = __builtin_fmaf(v3, v0, v1); }
          }
    }
    if (MODE == 2) __syncthreads();
  }
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* out) {
  int iters = 400;
  dim3 grid(256 * 2), block(256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, block>>>(out, 10, 1.0001f, 0.9999f);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flops = (double)grid.x * 4 * iters * 64.0 * (2.0 * 32 * 32 * 2);
  printf("%-52s %.3f ms  %.1f TFLOP/s\n", name, best, flops / best / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<0>("MFMA only (2 WG/CU)", out);
  run<1>("+ 16 ds_read_b128 per 64 MFMA", out);
  run<2>("+ 16 ds_read_b128 + 1 barrier per 64 MFMA", out);
  run<3>("+ 256 v_fma per 64 MFMA (VALU contention)", out);
  return 0;
}
