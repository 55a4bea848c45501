// Probe (not part of the product): do VALU work and bf16 MFMA work of DIFFERENT waves of one SIMD overlap?
// One 768-thread workgroup per CU: waves 0-3 run MFMA chains, waves 4-11 run packed-f32 / integer VALU chains (no memory, no LDS).
// mode bit 0: MFMA waves work, bit 1: VALU waves work.  Prints the time of each alone and of both.
// hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/_ovl && tools/_ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int VKIND>
__global__ __launch_bounds__(768) void probe(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    s16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (short)(0x3f80 + threadIdx.x + j); bv[j] = (short)(0x3f00 + j); }
    const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    f32x16 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {           // 24 MFMAs per iteration
#pragma unroll
      for (int k = 0; k < 24 / CHAINS; ++k)
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    if (s == 12345.f) out[threadIdx.x] = s;
  } else {
    if (!(mode & 2)) return;
    if (VKIND == 0) {                              // packed f32 adds, 8 independent chains: 112 instructions per iteration
      f32x2 v[8];
      for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x, (float)i};
      const f32x2 d = {1.0f, 0.5f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 14; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(d));
      }
      float s = 0.f;
      for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
      if (s == 12345.f) out[threadIdx.x] = s;
    } else {                                       // 32-bit integer ands / perms, 8 chains: 112 instructions per iteration
      unsigned v[8];
      for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[i]));
            asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(0x07060302u));
          }
      }
      unsigned s = 0;
      for (int i = 0; i < 8; ++i) s += v[i];
      if (s == 12345u) out[threadIdx.x] = (float)s;
    }
  }
}

template <int CHAINS, int VKIND>
static void run(const char* name) {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  float ms[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((probe<CHAINS, VKIND>), dim3(256), dim3(768), 0, 0, out, 10, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<CHAINS, VKIND>), dim3(256), dim3(768), 0, 0, out, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[mode], e0, e1);
  }
  // cycles per iteration at 2.4 GHz
  auto cyc = [&](float m) { return m * 1e-3 * 2.4e9 / iters; };
  printf("%-34s mfma alone %.0f cyc/iter (24 MFMA: %.1f each)   valu alone %.0f (2 waves x 112: %.2f each)   both %.0f   sum %.0f\n", name,
         cyc(ms[1]), cyc(ms[1]) / 24, cyc(ms[2]), cyc(ms[2]) / 224, cyc(ms[3]), cyc(ms[1]) + cyc(ms[2]));
  hipFree(out);
}

int main() {
  run<1, 0>("1 chain,  pk_add_f32");
  run<4, 0>("4 chains, pk_add_f32");
  run<1, 1>("1 chain,  and/perm");
  run<4, 1>("4 chains, and/perm");
  return 0;
}
