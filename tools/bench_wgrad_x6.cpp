// Diagnostic micro-benchmark / check of the split-bf16 2-D Winograd weight gradient against the f32-MFMA kernel (not part of the product).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops tools/bench_wgrad_x6.cpp -Ladm_amd -ladm_hip -o /tmp/bwx && LD_LIBRARY_PATH=adm_amd /tmp/bwx
#include "../adm_amd/csrc/conv_wgrad_x6.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
static void run(int B, int H, int Cin, int Cout) {
  size_t nx = (size_t)B * H * H * Cin, ny = (size_t)B * H * H * Cout, nw = (size_t)Cout * 12 * Cin;
  std::vector<float> hx(nx), hy(ny);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hy) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
  float *x, *dy, *w0, *w1, *b0, *b1, *am;
  hipMalloc(&x, nx * 4); hipMalloc(&dy, ny * 4); hipMalloc(&w0, nw * 4); hipMalloc(&w1, nw * 4); hipMalloc(&b0, Cout * 4); hipMalloc(&b1, Cout * 4); hipMalloc(&am, 2 * ADM_AMAX_FLOATS * 4); hipMemset(am, 0, 2 * ADM_AMAX_FLOATS * 4);
  { const float hx1 = 1.0f, hy1 = 0.05f; hipMemcpy(am, &hx1, 4, hipMemcpyHostToDevice); hipMemcpy(am + ADM_AMAX_FLOATS, &hy1, 4, hipMemcpyHostToDevice); }        // bound vectors of |x| and |dy|
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dy, hy.data(), ny * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[4];
  for (int which = 0; which < 4; ++which) {
    float* w = which ? w1 : w0; float* b = which ? b1 : b0;
    auto call = [&]() {
      hipMemsetAsync(b, 0, Cout * 4, 0);
      adm_wgrad_h3_blocks(which == 3 ? 2 : 1);
      return which >= 2 ? adm_conv_wgrad_x6_h3(x, dy, w, b, B, H, H, Cin, Cin, Cout, Cout, 0, 0, 0, am, am + ADM_AMAX_FLOATS, 0)
           : which ? adm_conv_wgrad_x6(x, dy, w, b, B, H, H, Cin, Cin, Cout, Cout, 0, 0)
                   : adm_conv_wgrad_wino2d(x, dy, w, b, B, H, H, Cin, Cin, Cout, Cout, 0, 0);
    };
    int rc = 0;
    for (int i = 0; i < 2; ++i) rc |= call();
    hipDeviceSynchronize();
    if (rc) { printf("rc=%d\n", rc); return; }
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) call();
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[which], e0, e1); ms[which] /= reps;
  }
  std::vector<float> a(nw), c(nw), ba(Cout), bc(Cout);
  hipMemcpy(a.data(), w0, nw * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), w1, nw * 4, hipMemcpyDeviceToHost);
  hipMemcpy(ba.data(), b0, Cout * 4, hipMemcpyDeviceToHost); hipMemcpy(bc.data(), b1, Cout * 4, hipMemcpyDeviceToHost);
  double mx = 0, sc = 0, bm = 0, bs = 0;
  for (size_t i = 0; i < nw; ++i) { mx = fmax(mx, fabs((double)a[i] - c[i])); sc = fmax(sc, fabs((double)a[i])); }
  for (int i = 0; i < Cout; ++i) { bm = fmax(bm, fabs((double)ba[i] - bc[i])); bs = fmax(bs, fabs((double)ba[i])); }
  double fl = 2.0 * B * H * H * (double)Cout * 9 * Cin;
  printf("B=%d H=%d Cin=%d Cout=%d: f32 %.3f ms (%.1f TF alg)  x6 %.3f ms (%.1f TF alg)  h3 %.3f ms (%.1f TF alg, splits %d)  h3 128-cout %.3f ms (%.1f TF)   max|dw(h3 128) - dw(f32)| %.2e of %.2e   max|db diff| %.2e of %.2e\n",
         B, H, Cin, Cout, ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, ms[2], fl / ms[2] / 1e9, adm_conv_wgrad_x6_plan(B, H, H, Cin, Cout), ms[3], fl / ms[3] / 1e9, mx, sc, bm, bs);
  hipFree(x); hipFree(dy); hipFree(w0); hipFree(w1); hipFree(b0); hipFree(b1);
}
int main() {
#if XW_ABL
  printf("ABL=%d ", XW_ABL); run(128, 32, 192, 192); return 0;
#endif
  run(2, 8, 32, 64);
  run(8, 16, 96, 64);
  run(4, 16, 96, 160);
  run(128, 32, 192, 192);
  run(128, 32, 384, 192);
  run(128, 32, 576, 192);
  run(128, 16, 384, 384);
  run(128, 16, 768, 384);
  run(128, 8, 384, 384);
  return 0;
}
