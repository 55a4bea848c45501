// Diagnostic micro-benchmark / ablation harness for the 2-D Winograd kernel (not part of the product).
// Build on the GPU box:  hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DW2_ABL=n] tools/bench_wino2d.cpp -Ladm_amd -ladm_hip -o /tmp/bw
//                         && LD_LIBRARY_PATH=adm_amd /tmp/bw          (W2_NOSPLIT=1: no split-K workspace)
#include "../adm_amd/csrc/conv_wino2d.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
static void run(int B, int H, int Cin, int N) {
  size_t nx = (size_t)B * H * H * Cin, nw = (size_t)16 * N * Cin, ny = (size_t)B * H * H * N;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.02f;
  float *x, *w, *y, *ws;
  long wsn = getenv("W2_NOSPLIT") ? 0 : (long)adm_wino2d_splitk(B, H, H, Cin, N) * ny;
  if (wsn < (long)ny * 2) wsn = 0;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); ws = nullptr; if (wsn) hipMalloc(&ws, wsn * 4);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) adm_conv_fwd_wino2d(x, w, nullptr, nullptr, y, ws, wsn, B, H, H, Cin, Cin, N, N, N, N, 0);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) adm_conv_fwd_wino2d(x, w, nullptr, nullptr, y, ws, wsn, B, H, H, Cin, Cin, N, N, N, N, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double fl = 2.0 * B * H * H * (double)N * 9 * Cin;
  printf("split=%d ABL=%d B=%d H=%d Cin=%d N=%d: %.3f ms  %.1f TFLOP/s algorithmic  %.1f executed\n", wsn ? (int)(wsn / ny) : 1, W2_ABL, B, H, Cin, N, ms, fl / ms / 1e9,
         fl * 4 / 9 / ms / 1e9);
  hipFree(x); hipFree(w); hipFree(y);
}
int main() {
  if (getenv("W2_WS")) adm_wino2d_variant(atoi(getenv("W2_WS")));
  printf("variant ws=%d\n", g_w2_ws);
  run(128, 32, 384, 384);
  run(128, 32, 192, 192);
  run(128, 16, 384, 384);
#ifndef W2_QUICK
  run(128, 16, 768, 384);
  run(128, 8, 384, 384);
  run(128, 8, 768, 384);
#endif
  return 0;
}
