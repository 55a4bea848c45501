// Diagnostic (not part of the product): sustained fp32-input MFMA rate of this device, registers only.
// Build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, float* out) {
  int iters = 2000;
  dim3 grid(256 * wgs_per_cu), block(256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<grid, block>>>(out, 10, 1.0001f, 0.9999f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<NACC><<<grid, block>>>(out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 /*waves*/ * iters * 16.0 * NACC * (2.0 * 32 * 32 * 2);
    printf("NACC=%d wgs/cu=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, ms, flops / ms / 1e9);
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<4>(1, out); run<4>(2, out); run<1>(2, out);
  return 0;
}
