// Probe (not part of the product): semantics of ds_read_b64_tr_b16 on an [rows][64 cols] 16-bit image with a row stride of 96 elements.
// hipcc -O3 --offload-arch=gfx950 tools/tr_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int STRIDE = 96;     // elements (192 bytes)
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[16 * STRIDE];
  for (int i = threadIdx.x; i < 16 * STRIDE; i += 64) lds[i] = (short)((i / STRIDE) * 256 + (i % STRIDE));   // value = row * 256 + col
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // group g: block rows (tiles) 8 * (g >> 1) + q, columns 16 * (g & 1) + 4 p .. + 3
  const int row = 8 * (g >> 1) + q, col = 16 * (g & 1) + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * STRIDE + col));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] >> 8, h[l * 4 + e] & 255);
    printf("\n");
  }
  return 0;
}
