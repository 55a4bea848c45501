// Diagnostic micro-benchmark for the split-bf16 2-D Winograd kernel (not part of the product).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops tools/bench_wino2d_x6.cpp -Ladm_amd -ladm_hip -o /tmp/bx && LD_LIBRARY_PATH=adm_amd /tmp/bx
#include "../adm_amd/csrc/conv_wino2d_x6.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
static void run(int B, int H, int Cin, int N, bool check) {
  size_t nx = (size_t)B * H * H * Cin, nw = (size_t)16 * N * Cin, ny = (size_t)B * H * H * N;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.02f;
  float *x, *w, *y, *y2, *ws = nullptr, *res = nullptr; void* w6;
  long wsn = (long)adm_wino2d_x6_splitk(B, H, H, Cin, N) * ny;
  if (wsn < (long)ny * 2 || getenv("X6_NOSPLIT")) wsn = 0;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); hipMalloc(&y2, ny * 4); hipMalloc(&w6, nw * 6);
  if (wsn) hipMalloc(&ws, wsn * 4);
#if X6_TL
  if (!ws) { hipMalloc(&ws, 1 << 20); hipMemset(ws, 0, 1 << 20); }
#endif
  if (getenv("X6_RES")) { hipMalloc(&res, ny * 4); hipMemset(res, 0, ny * 4); }
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  adm_split3_bf16(w, w6, N, Cin, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) adm_conv_fwd_wino2d_x6(x, w6, nullptr, res, y, ws, wsn, B, H, H, Cin, Cin, N, N, N, N, 0);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) adm_conv_fwd_wino2d_x6(x, w6, nullptr, res, y, ws, wsn, B, H, H, Cin, Cin, N, N, N, N, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double fl = 2.0 * B * H * H * (double)N * 9 * Cin;
  printf("x6 split=%d ABL=%d B=%d H=%d Cin=%d N=%d: %.3f ms  %.1f TFLOP/s algorithmic", wsn ? (int)(wsn / ny) : 1, X6_ABL, B, H, Cin, N, ms, fl / ms / 1e9);
  if (check) adm_conv_fwd_wino2d(x, w, nullptr, nullptr, y2, nullptr, 0, B, H, H, Cin, Cin, N, N, N, N, 0);
  {   // fp16 format: 64-cout workgroups, then the wide (128-cout) form
    void* wh; float* am; int* flag;
    hipMalloc(&wh, nw * 4); hipMalloc(&am, ADM_AMAX_FLOATS * 4); hipMemset(am, 0, ADM_AMAX_FLOATS * 4); hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    const float one = 1.0f; hipMemcpy(am, &one, 4, hipMemcpyHostToDevice);
    adm_split2_f16(w, wh, N, Cin, 2048.f, flag, 0);
    for (int wide : {0, 1, 3}) {
      adm_wino2d_h3_wide(wide);
      for (int i = 0; i < 3; ++i) adm_conv_fwd_wino2d_h3(x, wh, nullptr, res, y, ws, wsn, B, H, H, Cin, Cin, N, N, N, N, am, 2048.f, 0, 0);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) adm_conv_fwd_wino2d_h3(x, wh, nullptr, res, y, ws, wsn, B, H, H, Cin, Cin, N, N, N, N, am, 2048.f, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float m2; hipEventElapsedTime(&m2, e0, e1); m2 /= reps;
      printf("  | h3%s %.3f ms %.1f TF", wide == 1 ? " 128" : wide == 3 ? " 96" : " 64", m2, fl / m2 / 1e9);
      if (check) {
        std::vector<float> a(ny), b(ny);
        hipMemcpy(a.data(), y, ny * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y2, ny * 4, hipMemcpyDeviceToHost);
        double mx = 0, sc = 0;
        for (size_t i = 0; i < ny; ++i) { mx = fmax(mx, fabs((double)a[i] - b[i])); sc = fmax(sc, fabs((double)b[i])); }
        printf(" (max|diff to f32 kernel| %.2e of %.2e)", mx, sc);
      }
    }
    adm_wino2d_h3_wide(-1);
    hipFree(wh); hipFree(am); hipFree(flag);
  }
  printf("\n");
#if X6_TL
  if (B == 128 && H == 32 && Cin == 384) {
    static unsigned long long h[2048]; hipMemcpy(h, ws, sizeof(h), hipMemcpyDeviceToHost);
    printf("consumer wave 0 of workgroup 0, stages 8..31: [wait at barrier | DMA issue + fragment reads + MFMA issue | tail (transform) -> next stage start]\n");
    for (int t = 8; t < 32; ++t) printf("  t=%2d barrier %5llu  work %5llu  to-next %5llu\n", t, h[t * 4 + 1] - h[t * 4], h[t * 4 + 2] - h[t * 4 + 1], h[(t + 1) * 4] - h[t * 4 + 2]);
    printf("producer wave 4: [transform + split + LDS stores (lgkmcnt 0) | row-load issue | wait at barrier]\n");
    for (int t = 8; t < 32; ++t) { const unsigned long long* q = h + 1024 + t * 4; printf("  t=%2d store %5llu  issue %5llu  barrier %5llu\n", t, q[1] - q[0], q[2] - q[1], q[3] - q[2]); }
  }
#endif
  hipFree(x); hipFree(w); hipFree(y); hipFree(y2); hipFree(w6);
}
int main() {
#if X6_ABL
  run(128, 32, 384, 384, false);
  return 0;
#endif
  if (getenv("X6_ONLY")) { run(128, 32, 384, 384, false); return 0; }
  run(2, 8, 32, 64, true);
  run(3, 8, 64, 160, true);
  run(128, 32, 384, 384, true);
  run(128, 32, 192, 192, true);
  run(128, 32, 576, 192, false);
  run(128, 32, 384, 192, false);
  run(128, 32, 192, 576, false);
  run(128, 16, 384, 384, false);
  run(128, 16, 768, 384, false);
  run(128, 8, 384, 384, true);
  run(128, 8, 768, 384, false);
  run(128, 4, 384, 384, true);
  run(128, 4, 768, 384, false);
  return 0;
}
