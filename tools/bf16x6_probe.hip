// Probe (not part of the product): accuracy and rate of f32 GEMM emulated on the bf16 MFMA by a 3-term bf16 split
// (a = a0 + a1 + a2 exactly; six products a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 in f32 accumulators), against the f32 MFMA.
// hipcc -O3 --offload-arch=gfx950 tools/bf16x6_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float a, unsigned short& h, unsigned short& m, unsigned short& l) {
  unsigned u = __float_as_uint(a);
  unsigned hu = u & 0xFFFF0000u;
  float r = a - __uint_as_float(hu);
  unsigned mu = __float_as_uint(r) & 0xFFFF0000u;
  float r2 = r - __uint_as_float(mu);
  h = hu >> 16; m = mu >> 16; l = __float_as_uint(r2) >> 16;
}

// one wave: C[32][32] = A[32][K] * B[K][32]; A row-major, B given as Bt[32][K]
template <int MODE>   // 0: f32 MFMA, 6: six products one accumulator, 7: six products, corrections in a second accumulator, 3: three products
__global__ void gemm_probe(const float* A, const float* Bt, float* C, int K, int reps) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc = {0}, acc2 = {0};
  for (int rep = 0; rep < reps; ++rep) {
    if (MODE == 0) {
      for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], Bt[r * K + k + h], acc, 0, 0, 0);
    } else {
      for (int k = 0; k < K; k += 16) {
        s16x8 a[3], b[3];
        for (int j = 0; j < 8; ++j) {
          unsigned short x0, x1, x2;
          split3(A[r * K + k + 8 * h + j], x0, x1, x2); a[0][j] = x0; a[1][j] = x1; a[2][j] = x2;
          split3(Bt[r * K + k + 8 * h + j], x0, x1, x2); b[0][j] = x0; b[1][j] = x1; b[2][j] = x2;
        }
#define MM(i, j, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), c, 0, 0, 0)
        if (MODE == 3) { MM(0, 1, acc); MM(1, 0, acc); MM(0, 0, acc); }
        if (MODE == 6) { MM(0, 2, acc); MM(2, 0, acc); MM(1, 1, acc); MM(0, 1, acc); MM(1, 0, acc); MM(0, 0, acc); }
        if (MODE == 7) { MM(0, 2, acc2); MM(2, 0, acc2); MM(1, 1, acc2); MM(0, 1, acc2); MM(1, 0, acc2); MM(0, 0, acc); }
      }
    }
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i] + acc2[i];
}

int main() {
  for (int K : {576, 3456, 6912}) {
    std::vector<float> A(32 * K), Bt(32 * K);
    for (int dist = 0; dist < 2; ++dist) {
      srand(1);
      for (auto& v : A) v = dist ? (rand() / (float)RAND_MAX) : (rand() / (float)RAND_MAX) * 2 - 1;     // dist 1: all-positive (no cancellation)
      for (auto& v : Bt) v = dist ? (rand() / (float)RAND_MAX) * 0.02f : ((rand() / (float)RAND_MAX) * 2 - 1) * 0.02f;
      std::vector<double> ref(1024), mag(1024);
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0, m = 0;
        for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * Bt[j * K + k]; m += fabs((double)A[i * K + k] * Bt[j * K + k]); }
        ref[i * 32 + j] = s; mag[i * 32 + j] = m;
      }
      float *dA, *dB, *dC;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dC, 4096);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
      auto report = [&](const char* name) {
        std::vector<float> C(1024);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        double mx = 0, rms = 0;
        for (int i = 0; i < 1024; ++i) { double e = fabs(C[i] - ref[i]) / mag[i]; mx = fmax(mx, e); rms += e * e; }
        printf("K=%d dist=%d %-22s max|err|/sum|ab| = %.3e   rms = %.3e\n", K, dist, name, mx, sqrt(rms / 1024));
      };
      hipLaunchKernelGGL(gemm_probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1); hipDeviceSynchronize(); report("f32 MFMA");
      hipLaunchKernelGGL(gemm_probe<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1); hipDeviceSynchronize(); report("bf16 x6 (1 acc)");
      hipLaunchKernelGGL(gemm_probe<7>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1); hipDeviceSynchronize(); report("bf16 x6 (2 acc)");
      hipLaunchKernelGGL(gemm_probe<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1); hipDeviceSynchronize(); report("bf16 x3 (trunc)");
      hipFree(dA); hipFree(dB); hipFree(dC);
    }
  }
  return 0;
}
