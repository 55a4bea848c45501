// Diagnostic micro-benchmark for the implicit-GEMM kernel (not part of the product).
// Build on the GPU box:  hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DADM_EXP_x] tools/bench_igemm.cpp -o /tmp/bi && /tmp/bi
#include "../adm_amd/csrc/conv_igemm.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
static void run(int B, int H, int Cin, int N, int ks, int tile) {
  size_t nx = (size_t)B * H * H * Cin, nw = (size_t)N * ks * ks * Cin, ny = (size_t)B * H * H * N;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.02f;
  float *x, *w, *y, *r;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); hipMalloc(&r, ny * 4);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  hipMemset(r, 0, ny * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) adm_conv_fwd(x, w, nullptr, r, y, B, H, H, Cin, Cin, N, N, N, N, ks, 0, tile, 0);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) adm_conv_fwd(x, w, nullptr, r, y, B, H, H, Cin, Cin, N, N, N, N, ks, 0, tile, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double fl = 2.0 * B * H * H * (double)N * ks * ks * Cin;
  std::vector<float> hy(4); hipMemcpy(hy.data(), y, 16, hipMemcpyDeviceToHost);
  printf("B=%d H=%d Cin=%d N=%d ks=%d tile=%d: %.3f ms  %.1f TFLOP/s  (y0=%g)\n", B, H, Cin, N, ks, tile, ms, fl / ms / 1e9, hy[0]);
  hipFree(x); hipFree(w); hipFree(y); hipFree(r);
}
int main() {
  run(128, 32, 384, 384, 3, 0);
  run(128, 16, 384, 384, 3, 1);
  run(128, 32, 192, 192, 3, 1);
  run(128, 16, 768, 384, 3, 1);
  run(128, 32, 384, 384, 3, 2);
  run(128, 8, 384, 384, 3, 2);
  run(128, 16, 384, 384, 1, 1);
  return 0;
}
