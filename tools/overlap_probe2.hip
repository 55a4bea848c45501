// Probe (not part of the product): WHICH work overlaps with the bf16 MFMA on one SIMD of gfx950, and how.
// Round 2's probe (overlap_probe.hip) found that v_pk_add_f32 of other waves does not overlap with v_mfma_f32_32x32x16_bf16 at
// all and that and/perm hide ~40 %.  This one separates the causes: instruction kind (packed f32, plain f32, integer, moves),
// LDS fragment reads (other wave / same wave), and same-wave software interleaving (VALU or ds_read issued between MFMAs).
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_probe2.hip -o tools/_ovl2 && tools/_ovl2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- other-wave work: waves 4..4+NV-1 run KIND while waves 0-3 run 24 MFMAs per iteration (4 chains)
template <int KIND>
__device__ __forceinline__ void valu_iter(f32x2 (&v)[8], unsigned (&u)[8], const unsigned short* lds, u32x4 (&f)[4]) {
  const f32x2 d = {1.0f, 0.5f};
  if (KIND == 0) {          // 112 v_pk_add_f32
#pragma unroll
    for (int k = 0; k < 14; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(d));
  } else if (KIND == 1) {   // 112 v_add_f32
#pragma unroll
    for (int k = 0; k < 14; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i][0]) : "v"(d[0]));
  } else if (KIND == 2) {   // 112 v_and_b32 (literal)
#pragma unroll
    for (int k = 0; k < 14; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
  } else if (KIND == 3) {   // 112 v_perm_b32
#pragma unroll
    for (int k = 0; k < 14; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(0x07060302u));
  } else if (KIND == 4) {   // 112 v_pk_fma_f32
#pragma unroll
    for (int k = 0; k < 14; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(d));
  } else if (KIND == 5) {   // 24 ds_read_b128 (conflict-free: lane * 16 bytes, consecutive KBs)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const volatile u32x4*>(lds + (k * 4 + i) * 512 + (threadIdx.x & 63) * 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] += f[i][0] ^ f[i][3];
    }
  } else if (KIND == 6) {   // 12 ds_write_b64
#pragma unroll
    for (int k = 0; k < 12; ++k)
      *reinterpret_cast<volatile f32x2*>(const_cast<unsigned short*>(lds) + 24576 + k * 256 + (threadIdx.x & 255) * 4) = v[k & 7];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int KIND, int NV, int PRIO = 0, int CHAINS = 4>
__global__ __launch_bounds__(64 * (4 + NV)) void probe_other(float* out, int iters, int mode, const float* gsrc = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32768];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  if (wave < 4) {
    if (!(mode & 1)) return;
    s16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (short)(0x3f80 + threadIdx.x + j); bv[j] = (short)(0x3f00 + j); }
    const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    f32x16 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
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
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    f32x2 v[8]; unsigned u[8]; u32x4 f[4];
    for (int i = 0; i < 8; ++i) { v[i] = f32x2{(float)threadIdx.x, (float)i}; u[i] = threadIdx.x * 2654435761u + i; }
    if (KIND == 7) {          // 8 buffer_load_dwordx4 per iteration (L1/L2-resident 64 KB window), waited for once per iteration
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, 1 << 20, 0x00020000);
      for (int it = 0; it < iters; ++it) {
        u32x4 g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          g[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((threadIdx.x * 16 + i * 8192 + (it & 7) * 65536) & 0xFFFF0), 0, 0));
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] += g[i][0] ^ g[i][3];
      }
    } else
    for (int it = 0; it < iters; ++it) valu_iter<KIND>(v, u, lds, f);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1] + (float)u[i];
    if (s == 12345.f) out[threadIdx.x] = s;
  }
}

// ---- same-wave interleave: 4 waves per workgroup, each: 24 MFMAs per iteration with NI ops of KIND after every MFMA
template <int KIND, int NI>
__global__ __launch_bounds__(256) void probe_same(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32768];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  s16x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = (short)(0x3f80 + threadIdx.x + j); bv[j] = (short)(0x3f00 + j); }
  bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  f32x2 v[8]; unsigned u[8]; u32x4 f[4] = {};
  for (int i = 0; i < 8; ++i) { v[i] = f32x2{(float)threadIdx.x, (float)i}; u[i] = threadIdx.x * 2654435761u + i; }
  const f32x2 d = {1.0f, 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int r = (k * 4 + i + j) & 7;
          if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(d));
          else if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r][0]) : "v"(d[0]));
          else if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[r]));
          else if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(0x07060302u));
          else if (KIND == 5) {     // one ds_read_b128 per slot, waited for four MFMAs later (here: never explicitly; the loop-carried use is at the end)
            asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(f[(k * 4 + i + j) & 3]) : "v"((unsigned)(((k * 4 + i) * 1024 + (threadIdx.x & 63) * 16) & 65535)));
          }
        }
      }
    if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1] + (float)u[i];
  for (int i = 0; i < 4; ++i) s += (float)(f[i][0] ^ f[i][3]);
  if (s == 12345.f) out[threadIdx.x] = s;
}

static const int ITERS = 2000;
static float* g_out;
template <typename F>
static float timeit(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(10); hipDeviceSynchronize();
  hipEventRecord(e0); launch(ITERS); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f * 2.4e9f / ITERS;      // "cycles" per iteration at a nominal 2.4 GHz
}

static float* g_src;
template <int KIND, int NV, int PRIO = 0, int CHAINS = 4>
static void other(const char* name) {
  float t[4];
  for (int mode = 1; mode <= 3; ++mode)
    t[mode] = timeit([&](int it) { hipLaunchKernelGGL((probe_other<KIND, NV, PRIO, CHAINS>), dim3(256), dim3(64 * (4 + NV)), 0, 0, g_out, it, mode, g_src); });
  printf("prio=%d chains=%d ", PRIO, CHAINS);
  printf("other-wave %-30s x%d waves: mfma %.0f  other %.0f  both %.0f  (sum %.0f, max %.0f) -> hidden %.0f%%\n", name, NV, t[1], t[2], t[3],
         t[1] + t[2], t[1] > t[2] ? t[1] : t[2], 100.f * (t[1] + t[2] - t[3]) / (t[1] < t[2] ? t[1] : t[2]));
}
template <int KIND, int NI>
static void same(const char* name) {
  const float t = timeit([&](int it) { hipLaunchKernelGGL((probe_same<KIND, NI>), dim3(256), dim3(256), 0, 0, g_out, it); });
  printf("same-wave  %-30s %d per MFMA: %.0f cycles per 24 MFMAs (+%d ops)\n", name, NI, t, 24 * NI);
}

int main() {
  hipMalloc(&g_out, 1 << 16);
  hipMalloc(&g_src, 1 << 21); hipMemset(g_src, 0, 1 << 21);
  other<5, 4, 1>("ds_read_b128 (24)"); other<6, 4, 1>("ds_write_b64 (12)"); other<7, 4, 0>("buffer_load_b128 (8)"); other<7, 4, 1>("buffer_load_b128 (8)");
  other<5, 4, 0, 1>("ds_read_b128 (24)"); other<6, 4, 0, 1>("ds_write_b64 (12)"); other<5, 4, 1, 1>("ds_read_b128 (24)"); other<6, 4, 1, 1>("ds_write_b64 (12)");
  other<7, 4, 1, 1>("buffer_load_b128 (8)"); other<0, 4, 1, 1>("v_pk_add_f32 (112)"); other<1, 4, 1, 1>("v_add_f32 (112)");
  same<1, 0>("(MFMA only)");
  other<0, 4>("v_pk_add_f32 (112)"); other<1, 4>("v_add_f32 (112)"); other<2, 4>("v_and_b32 (112)"); other<3, 4>("v_perm_b32 (112)");
  other<4, 4>("v_pk_fma_f32 (112)"); other<5, 4>("ds_read_b128 (24)"); other<6, 4>("ds_write_b64 (12)");
  other<1, 8>("v_add_f32 (112)"); other<5, 8>("ds_read_b128 (24)");
  same<0, 1>("v_pk_add_f32"); same<0, 2>("v_pk_add_f32"); same<0, 4>("v_pk_add_f32");
  same<1, 1>("v_add_f32"); same<1, 2>("v_add_f32"); same<1, 4>("v_add_f32"); same<1, 6>("v_add_f32");
  same<2, 2>("v_and_b32"); same<2, 4>("v_and_b32"); same<3, 2>("v_perm_b32"); same<3, 4>("v_perm_b32");
  same<5, 1>("ds_read_b128"); same<5, 2>("ds_read_b128");
  return 0;
}
