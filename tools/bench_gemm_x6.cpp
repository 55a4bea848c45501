// Diagnostic micro-benchmark of the split-bf16 1x1 kernel (not part of the product); -DG6_HALF=1 drops three of the six MFMAs per
// product (wrong results: prices what a two-term format could gain on the matrix pipe alone).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops tools/bench_gemm_x6.cpp -Ladm_amd -ladm_hip -o tools/_bg && LD_LIBRARY_PATH=adm_amd tools/_bg
#include "../adm_amd/csrc/conv_gemm_x6.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
static void run(long M, int K, int N) {
  size_t nx = (size_t)M * K, nw = (size_t)N * K, ny = (size_t)M * N;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
  float *x, *w, *y; void* w6;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); hipMalloc(&w6, nw * 6);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  adm_split3_rows(w, w6, N, K, K, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int rc = 0;
  for (int i = 0; i < 3; ++i) rc |= adm_gemm_x6(x, w6, nullptr, nullptr, y, M, K, K, N, N, N, N, 0);
  hipDeviceSynchronize();
  if (rc) { printf("rc=%d\n", rc); return; }
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) adm_gemm_x6(x, w6, nullptr, nullptr, y, M, K, K, N, N, N, N, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("M=%ld K=%d N=%d: %.3f ms  %.1f TFLOP/s (f32-equivalent)\n", M, K, N, ms, 2.0 * M * K * N / ms / 1e9);
  hipFree(x); hipFree(w); hipFree(y); hipFree(w6);
}
int main() {
  run(131072, 192, 384); run(131072, 576, 384); run(131072, 384, 256);
  run(32768, 384, 1152); run(32768, 384, 384); run(32768, 768, 384); run(32768, 1152, 384);
  run(8192, 768, 384); run(8192, 384, 384);
  return 0;
}
