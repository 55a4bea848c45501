// 3x3 stride-1 convolution (forward and data gradient): 2-D Winograd F(2x2, 3x3) with the f32 products carried on the bf16 MFMA.
//
// gfx950 has no reduced-precision fast path for f32 matrix operands: v_mfma_f32_32x32x2_f32 runs at the f32 vector rate, 1/16 of
// the bf16 MFMA.  An f32 value splits EXACTLY into three bf16 terms, a = a0 + a1 + a2 (8 + 8 + 8 mantissa bits, same exponent
// range, by truncation: a0 = top 16 bits of a, a1 = top 16 bits of a - a0, a2 = a - a0 - a1), and bf16 x bf16 products are exact in
// the f32 accumulator, so
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0) + O(2^-24 |a b|)
// -- six v_mfma_f32_32x32x16_bf16 (6 x 32 cycles for K = 16) replace eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles): 2.67x less
// matrix-pipe time at f32 accuracy (tools/bf16x6_probe.hip: the six products in a short accumulator chain started from C = 0
// and added to the running sum with one f32 add measure BELOW the f32 MFMA's own rounding error against fp64, on zero-mean and
// on all-positive data; six truncating matrix adds into one long-running accumulator would be 3x worse on the latter).
//
// The algorithm is conv_wino2d.hip's (same transforms, four ex accumulator tiles per wave, output rows folded in registers);
// the loop order and the division of labour are not:
//   * the matrix pipe is no longer the bound, the data path is.  The f32 kernel's stages are as long as a global load takes
//     (every barrier of a __syncthreads also drains vmcnt), which the 2048-cycle f32 MFMA phase used to cover; a 768-cycle bf16
//     phase does not.  So the roles are split (512 threads, one workgroup per CU):
//       - waves 4-7 PRODUCE the A operand: thread = (tile, 16-byte channel quad); the loads of a stage (2 rows x 4 pixels, raw
//         buffer loads, range check = zero padding) are issued FOUR stages ahead into one of four register sets -- these waves hold
//         no accumulators, so they have the registers for it -- then y-combined and B^T-transformed in f32 exactly as before, split
//         into three bf16 terms (and / sub / and / sub per element, v_perm packing) and written to LDS (twelve ds_write_b64);
//       - waves 0-3 CONSUME: 32 tiles x 32 couts each, six MFMAs per ex plane and stage.  Stages are ordered K block (4 chunks) >
//         ey > chunk: inside a (block, ey) group the accumulators run on (<= 24 matrix adds each, the first from C = 0), at its
//         end the x / y output transforms are applied and the result is added to the output rows held in registers with f32 adds
//         -- the long-running sums see rounded f32 adds, not truncating matrix adds: measured against fp64 the error is at or
//         below the f32 MFMA kernel's (tests/test_hip_ops.py::test_conv_x6_error_vs_fp64).  The four passes over a block also
//         re-read the input rows 4 stages apart (L2 hits) instead of a whole Cin pass apart.
//         These waves also issue the weight DMA (their only global accesses: the explicit vmcnt wait before a barrier
//         never touches the producers' prefetch);
//       - barriers are s_barrier with LDS-scoped fences (lgkmcnt only): the producers' loads stay in flight across them.
//   * LDS: A[2][4 ex][3 terms][64 tiles][16 ch] bf16 (2 x 24 KB) + B[4][4 ex][3 terms][64 couts][16 ch] (4 x 24 KB; weights are
//     split once per optimiser step at pack time into the K-chunk-tiled layout of adm_split3_bf16 -- one contiguous KB per DMA
//     instruction -- and arrive by LDS-DMA three stages ahead).  Rows are 32
//     bytes with the two 16-byte halves of rows 8-15 (mod 16) swapped, so that the 16 lanes of a fragment-read group hit all 64 banks
//     (un-swizzled: 40 % of the LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT).
// Replaces F.conv2d of Conv2d.forward and its autograd data gradient (/root/reference/unet/uncond_unet.py:98-110).
#include "common.h"
#include "../../include/adm_hip.h"
#include <type_traits>

#ifndef X6_TL
#define X6_TL 0      // diagnostic build: per-stage timeline of one consumer and one producer wave of workgroup 0 into p.ws (tools/bench_wino2d_x6.cpp)
#endif
#ifndef X6_SHARE
#define X6_SHARE 0   // 1: patch columns shared between neighbouring tiles by cross-lane moves (measured slower: see the producer)
#endif
#ifndef X6_ABL
#define X6_ABL 0     // diagnostic builds (tools/bench_wino2d_x6.cpp): 1 no A global loads, 2 no A transform / split / LDS stores, 4 no B DMA, 8 no MFMAs, 16 no LDS fragment reads
#endif

namespace {

struct X6P {
  const float* x; const unsigned short* w; const float* bias; const float* res; float* y;
  int Mt, N, H, W, Hh, Wh, Cin, ldx, ldy, ldr, wrows, tilesN, xbytes, wbytes, plane;   // plane = wrows * Cin (elements of one [term] image)
  int splitk, chunks_per_split; float* ws;
  int ybytes, rbytes;                  // > 0: y / res fit 32-bit byte offsets (branch-free buffer epilogue)
  int up;                              // 1: x is [B][H/2][W/2][ldx] and the conv runs on its nearest x2 up-sampling (Conv2d(up=True))
  const float* amax_x; float wscale;   // fp16 format only: bound vector (adm_hip.h) of |x|; the weights' (power-of-two) scale
};

typedef __attribute__((address_space(3))) void x6_lds_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int X6P_T = 64, X6N = 64, X6K = 16;      // tiles x couts x K step
[[maybe_unused]] constexpr int X6_A_STAGE = 4 * 3 * X6P_T * X6K;    // bf16 elements per A sub-stage image (24 KB; the bf16 format)
[[maybe_unused]] constexpr int X6_B_STAGE = 4 * 3 * X6N * X6K;
// Number formats of the split (template parameter FMT of the kernel):
//   0  a = a0 + a1 + a2, three bf16 terms by truncation, exact; a b from SIX products (small ones first, the three below 2^-24 dropped).
//      Needs nothing but the operands.
//   1  s a = h0 + h1, two fp16 terms by round-to-nearest, |s a - h0 - h1| <= 2^-24 |s a| while h1 is a normal fp16 number and
//      <= 2^-25 absolutely below that; a b from THREE products (h1 h1' <= 2^-24 |a b| dropped).  s is a power of two chosen from an
//      upper bound of max |a| so that the Winograd input transform (sums of four values) stays inside the fp16 range: the caller
//      passes that bound as a device scalar (the GroupNorm kernel that produced the activation wrote it).  Against fp64 it is as
//      accurate as format 0 on normal, all-positive, heavy-tailed and single-outlier data (tools/fp16x3_accuracy.py; on the GPU:
//      tests/test_hip_ops.py::test_conv_h3_error_vs_fp64) and halves the MFMAs, the fragment reads, the split stores and the weight
//      DMA of a stage: 1.25 -> 1.02 ms on 128 x 32 x 32 x 384 -> 384 (tools/exp_wino2d_h3.hip).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// NB = 32-cout blocks per workgroup.  2: 64 couts, four consumer waves (512 threads).  4 (the "wide" form, fp16 format only): 128
// couts, EIGHT consumer waves, two per SIMD, next to the same four producer waves (768 threads, <= 168 registers): one A image
// (loads from L2, transform, split, LDS stores -- the part of a stage that does not shrink with the format) now feeds twice the
// MFMAs, and the launch pulls half the activation bytes out of the L2 (every cout tile re-reads the whole input: at Cin = 384 the
// 64-cout form reads 74 KB per pixel, ~9.5 TB/s over the launch, more than half of what the L2s deliver).
template <int FMT, int NB = 2> struct X6Fmt {
  static constexpr int TERMS = FMT ? 2 : 3;
  static constexpr int NT = 32 * NB;                 // couts per workgroup
  static constexpr int CONS = 2 * NB;                // consumer waves: (2 tile halves) x (NB cout blocks)
  static constexpr int THREADS = (CONS + 4) * 64;
  static constexpr int A_STAGE = 4 * TERMS * X6P_T * X6K;
  static constexpr int B_STAGE = 4 * TERMS * NT * X6K;
  static constexpr int DMA_PER_WAVE = 2 * TERMS;     // 4 ex x TERMS images x NB 32-row blocks / CONS consumer waves
  static constexpr int RB = NB > 2 ? 3 : 4;         // weight ring depth (LDS: 2 x 16 + 3 x 32 KB wide; 2 x 16 + 4 x 16 KB fp16, 2 x 24 + 4 x 24 KB bf16)
  static constexpr int PD = NB > 2 ? 3 : 4;         // producer register sets = stages of loads in flight
};
// power-of-two scale that puts 4 * amax below the fp16 range (65504): s * amax <= 16000
__device__ __host__ inline float h3_scale(float amax) {
  if (!(amax > 0.f) || !(amax < 3e38f)) return 1.f;
  int e;
  frexpf(16000.f / amax, &e);                         // 16000 / amax = m 2^e, m in [0.5, 1)
  return ldexpf(1.f, e - 1);
}
constexpr int X6_RA = 2;                           // A ring depth (the weights' is X6Fmt::RB: RB - 1 stages ahead, a DMA queues behind the producers' loads)

// LDS-only workgroup barrier: waits for this wave's LDS traffic (lgkmcnt), NOT for its global loads
__device__ __forceinline__ void x6_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// (this file is compiled WITHOUT packed-f32 instruction selection -- csrc/Makefile, NOPK -- so vector adds below are plain v_add_f32 /
//  v_sub_f32; the explicit helpers keep the producer's arithmetic plain even if that flag is dropped)
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) { return a - b; }
// PRODUCER arithmetic: plain (one-lane-one-value) f32 adds.  tools/overlap_probe2.hip: next to v_mfma_f32_32x32x16_bf16 of another
// wave on the same SIMD, v_add_f32 / v_and_b32 / v_perm_b32 are 91-96 % hidden, the PACKED forms (v_pk_add_f32, v_pk_fma_f32) not at
// all (0-3 %: they share the matrix pipe's data path) -- the packed form halves the instruction count and doubles the cost.  The
// compiler packs every f32x2-shaped add it sees, hence the inline assembly.
__device__ __forceinline__ float p_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float p_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 p_add4(f32x4 a, f32x4 b) { return f32x4{p_add(a[0], b[0]), p_add(a[1], b[1]), p_add(a[2], b[2]), p_add(a[3], b[3])}; }
__device__ __forceinline__ f32x4 p_sub4(f32x4 a, f32x4 b) { return f32x4{p_sub(a[0], b[0]), p_sub(a[1], b[1]), p_sub(a[2], b[2]), p_sub(a[3], b[3])}; }

__device__ __forceinline__ f32x16 sub16(f32x16 a, f32x16 b) { return a - b; }

// v = v0 + v1 + v2 exactly, each term a bf16 (packed top halves: two dwords per term for the four channels)
__device__ __forceinline__ void split3_pack(const f32x4 v, u32x2& t0, u32x2& t1, u32x2& t2) {
  f32x4 h, mh;
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = __uint_as_float(__float_as_uint(v[i]) & 0xFFFF0000u);
  const f32x4 r = p_sub4(v, h);
#pragma unroll
  for (int i = 0; i < 4; ++i) mh[i] = __uint_as_float(__float_as_uint(r[i]) & 0xFFFF0000u);
  const f32x4 r2 = p_sub4(r, mh);
  t0 = u32x2{__builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07060302u)};
  t1 = u32x2{__builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(r[3]), __float_as_uint(r[2]), 0x07060302u)};
  t2 = u32x2{__builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u)};
}

// v * s = h0 + h1 (two fp16 terms, round to nearest), four channels -> two dwords per term
__device__ __forceinline__ void split2_pack(const f32x4 v, float s, u32x2& t0, u32x2& t1) {
  _Float16 h0[4], h1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float vs = v[i] * s;
    h0[i] = (_Float16)vs;
    h1[i] = (_Float16)(vs - (float)h0[i]);
  }
  t0 = u32x2{__builtin_bit_cast(unsigned, f16x2{h0[0], h0[1]}), __builtin_bit_cast(unsigned, f16x2{h0[2], h0[3]})};
  t1 = u32x2{__builtin_bit_cast(unsigned, f16x2{h1[0], h1[1]}), __builtin_bit_cast(unsigned, f16x2{h1[2], h1[3]})};
}
__device__ __forceinline__ void h3_store(const f32x4 (&e)[4], unsigned short* la, float s) {
  if (X6_ABL & 2) return;
  const f32x4 v[4] = {p_sub4(e[0], e[2]), p_add4(e[1], e[2]), p_sub4(e[2], e[1]), p_sub4(e[1], e[3])};
#pragma unroll
  for (int ex = 0; ex < 4; ++ex) {
    u32x2 t0, t1;
    split2_pack(v[ex], s, t0, t1);
    *reinterpret_cast<u32x2*>(la + (ex * 2 + 0) * X6P_T * X6K) = t0;
    *reinterpret_cast<u32x2*>(la + (ex * 2 + 1) * X6P_T * X6K) = t1;
  }
}

// producer: y-combined rows e[4] (one per pixel of the patch row) -> B^T along x -> three-term split -> the [ex][term] images
__device__ __forceinline__ void x6_store(const f32x4 (&e)[4], unsigned short* la) {
  if (X6_ABL & 2) return;
  const f32x4 v[4] = {p_sub4(e[0], e[2]), p_add4(e[1], e[2]), p_sub4(e[2], e[1]), p_sub4(e[1], e[3])};
#pragma unroll
  for (int ex = 0; ex < 4; ++ex) {
    u32x2 t0, t1, t2;
    split3_pack(v[ex], t0, t1, t2);
    *reinterpret_cast<u32x2*>(la + (ex * 3 + 0) * X6P_T * X6K) = t0;
    *reinterpret_cast<u32x2*>(la + (ex * 3 + 1) * X6P_T * X6K) = t1;
    *reinterpret_cast<u32x2*>(la + (ex * 3 + 2) * X6P_T * X6K) = t2;
  }
}

// Stage order shared by both roles: K blocks of X6_KB chunks outermost, the four ey passes inside a block, chunks innermost
// (a stage = one (ey, chunk) pair).  Inside a (block, ey) group the MFMA accumulators run on (<= 24 matrix adds each); at its end
// they are transformed and added to the output rows with f32 adds.
#ifndef X6_KB_N
#define X6_KB_N 4
#endif
constexpr int X6_KB = X6_KB_N;
// With the fused nearest x2 up-sampling the patch rows r1 and r2 are the SAME source row, so the pass ey = 2 (r2 - r1) is
// identically zero and is skipped (three passes per block).
struct X6Seq {
  int k0, len, ey, cc, skip2;   // block start (chunk), block length, pass, chunk inside the block, skip pass 2
  __device__ __forceinline__ void init(int chunks, int up) { k0 = 0; len = min(X6_KB, chunks); ey = 0; cc = 0; skip2 = up; }
  __device__ __forceinline__ int chunk() const { return k0 + cc; }
  __device__ __forceinline__ bool done() const { return len <= 0; }
  __device__ __forceinline__ void next(int chunks) {
    if (++cc == len) {
      cc = 0;
      ++ey;
      if (skip2 && ey == 2) ++ey;
      if (ey == 4) { ey = 0; k0 += len; len = min(X6_KB, chunks - k0); }
    }
  }
};

template <int FMT, int NB>
__global__ __launch_bounds__((2 * NB + 4) * 64) void wino2d_x6_kernel(X6P p) {
  using F = X6Fmt<FMT, NB>;
  constexpr int TERMS = F::TERMS;
  float sa = 1.f, inv_scale = 1.f;                     // fp16 format: operand scale and the factor that undoes both scales in the epilogue
  if (FMT) { sa = h3_scale(adm_amax_read(p.amax_x)); inv_scale = 1.f / (sa * p.wscale); }
  extern __shared__ __attribute__((aligned(16))) unsigned short smem6[];
  unsigned short* As = smem6;                          // [X6_RA][4 ex][3 terms][X6P_T][X6K]
  unsigned short* Bs = smem6 + X6_RA * F::A_STAGE;     // [RB][4 ex][3 terms][X6N][X6K]
  const int tid = threadIdx.x, lane = tid & 63, hw_wid = tid >> 6;
  const bool producer = hw_wid >= F::CONS;
  const int wid = producer ? hw_wid - F::CONS : hw_wid;      // role-local wave index
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap, m-fastest inside an n-tile (see conv_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int mt0 = tm * X6P_T, n0 = tn * F::NT;
  constexpr unsigned OOB = 0x80000000u;

  const int c_begin = (p.splitk > 1) ? (int)blockIdx.y * p.chunks_per_split : 0;
  const int chunks = (p.splitk > 1) ? min(p.chunks_per_split, (p.Cin >> 4) - c_begin) : (p.Cin >> 4);     // 16-channel chunks (even)
  const int S = (p.up ? 3 : 4) * chunks;              // stages (an even number: Cin is a multiple of 32)

  if (producer) {
    // ================================================================ producer waves: A operand
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
    const int ptid = wid * 64 + lane;
    const int pl = ptid >> 2, aq = ptid & 3;          // tile, channel quad
    unsigned a_base = 0, colmask = 0, rowmask = 0;
    // Patch columns shared between neighbouring tiles: tile xp reads input columns 2xp - 1 .. 2xp + 2, so its column 0 is column 2 of
    // tile xp - 1 and its column 3 is column 1 of tile xp + 1 -- tiles that sit FOUR LANES away in this wave (thread = (tile, quad)).
    // With X6_SHARE a lane whose neighbour is in its wave and its tile row takes the y-combined value from it with one cross-lane
    // move per dword instead of asking for the same bytes again: half the activation requests of a stage.  Correct (the parity suite
    // passes with it) and SLOWER on every shape (tools/bench_wino2d_x6.cpp, 128 x 32 x 32 x 384 -> 384: 0.905 -> 1.106 ms with 64
    // couts, 0.789 -> 0.865 with 128): the repeated columns were L1 hits, and the eight ds_bpermute per thread and stage are LDS
    // operations, which do not overlap with the MFMAs of the SIMD's other waves.  Off by default; not for the fused up-sampling.
    bool take_l = false, take_r = false;
    {
      const int t = mt0 + pl;
      if (t < p.Mt) {
        const int xp = t % p.Wh;
        const int u = t / p.Wh;
        const int ty = u % p.Hh, b = u / p.Hh;
        a_base = p.up ? (unsigned)((((long)b * p.Hh + ty) * p.Wh + xp) * p.ldx + aq * 4) * 4u        // source pixel (b, ty, xp)
                      : (unsigned)((((long)b * p.H + 2 * ty) * p.W + 2 * xp) * p.ldx + aq * 4) * 4u;
        colmask = (xp > 0 ? 1u : 0u) | 6u | (2 * xp + 2 < p.W ? 8u : 0u);
        rowmask = (ty > 0 ? 1u : 0u) | 6u | (2 * ty + 2 < p.H ? 8u : 0u);
        if (X6_SHARE && !p.up) {
          take_l = xp > 0 && (pl & 15) != 0;                // (tile pl - 1 = lane - 4 of this wave, same tile row)
          take_r = xp + 1 < p.Wh && (pl & 15) != 15 && t + 1 < p.Mt;
          if (take_l) colmask &= ~1u;                       // no request for what the neighbour loads
          if (take_r) colmask &= ~8u;
        }
      }
    }
    // rows are 32 bytes; a 16-lane group of a fragment read covers 16 rows at one 16-byte half, i.e. only half of the banks, unless
    // the halves of rows 8-15 (mod 16) are swapped: physical half = logical half ^ ((row >> 3) & 1), for A and B alike
    unsigned short* la = As + pl * X6K + ((((aq >> 1) ^ (pl >> 3)) & 1) << 3) + (aq & 1) * 4;
    X6Seq ld; ld.init(chunks, p.up);
    unsigned a_voff[2][4];
    int voff_ey = -1;
    auto set_rows = [&]() {     // pass ey combines input rows (iA, iB): 0: +r0 -r2   1: +r1 +r2   2: -r1 +r2   3: +r1 -r3; past the end: nothing
      const int ey = ld.ey;
      const int iA = (ey == 0) ? 0 : 1, iB = (ey == 3) ? 3 : 2;
      const bool live = !ld.done();
      const bool vA = ((rowmask >> iA) & 1u) && live, vB = ((rowmask >> iB) & 1u) && live;
      // up-sampled row 2ty - 1 + i reads source row ty + (i + 1) / 2 - 1 = ty - 1, ty, ty, ty + 1 (columns likewise)
      const int offA = (p.up ? ((iA + 1) >> 1) - 1 : iA - 1) * (p.up ? p.Wh : p.W) * p.ldx * 4;
      const int offB = (p.up ? ((iB + 1) >> 1) - 1 : iB - 1) * (p.up ? p.Wh : p.W) * p.ldx * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cv = (colmask >> j) & 1u;
        const int cj = (p.up ? ((j + 1) >> 1) - 1 : j - 1) * p.ldx * 4;
        a_voff[0][j] = (vA && cv) ? a_base + (unsigned)(offA + cj) : OOB;
        a_voff[1][j] = (vB && cv) ? a_base + (unsigned)(offB + cj) : OOB;
      }
      voff_ey = live ? ey : 4;
    };
    constexpr int D = F::PD;                          // stages in flight
    f32x4 dA[D][4], dB[D][4];
    int set_ey[D];
    auto issue = [&](int d) {                         // next stage of the sequence -> register set d
      if (voff_ey != (ld.done() ? 4 : ld.ey)) set_rows();
      const int soff = (c_begin + ld.chunk()) << 6;   // 16 floats = 64 bytes per chunk
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (X6_ABL & 1) { dA[d][j] = f32x4{1.f, 2.f, 3.f, (float)soff}; dB[d][j] = f32x4{0.5f, 0.25f, (float)j, 1.f}; continue; }
        dA[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[0][j], soff, 0));
        dB[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[1][j], soff, 0));
      }
      set_ey[d] = ld.ey;
      if (!ld.done()) ld.next(chunks);
    };
    auto store = [&](int d, int slot) {
      f32x4 e[4];
      if (set_ey[d] == 1) {                           // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = p_add4(dA[d][j], dB[d][j]);
      } else if (set_ey[d] == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = p_sub4(dB[d][j], dA[d][j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = p_sub4(dA[d][j], dB[d][j]);
      }
      if (X6_SHARE && !p.up) {                          // (uniform) every lane takes part in the moves
        f32x4 el, er;
#pragma unroll
        for (int i = 0; i < 4; ++i) { el[i] = __shfl_up(e[2][i], 4, 64); er[i] = __shfl_down(e[1][i], 4, 64); }
        if (take_l) e[0] = el;
        if (take_r) e[3] = er;
      }
      if (FMT) h3_store(e, la + slot * F::A_STAGE, sa);
      else x6_store(e, la + slot * F::A_STAGE);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d);
    __builtin_amdgcn_sched_barrier(0);
    // barrier t separates "A(t) written" from compute(t); A(t) lives in slot t & 1; set t % D is refilled with stage t + D
    constexpr int TRIP = (D & 1) ? 2 * D : D;         // stages per trip: static register sets AND static slots
    for (int t = 0; t < S; t += TRIP) {
#pragma unroll
      for (int k = 0; k < TRIP; ++k) {
        const int d = k % D;
        if (t + k < S) {                              // (uniform; S is even, a multiple of 4 without the up-sampling)
          const bool tl = X6_TL && p.ws && blockIdx.x == 0 && tid == F::CONS * 64 && t + k < 48;
          unsigned long long* TL = reinterpret_cast<unsigned long long*>(p.ws) + 1024 + (t + k) * 4;
          if (tl) TL[0] = __builtin_readcyclecounter();
          store(d, k & 1);
          __builtin_amdgcn_sched_barrier(0);
          if (tl) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TL[1] = __builtin_readcyclecounter(); }
          issue(d);                                   // stages past the end read nothing (all offsets out of range)
          __builtin_amdgcn_sched_barrier(0);
          if (tl) TL[2] = __builtin_readcyclecounter();
          x6_barrier();
          if (tl) TL[3] = __builtin_readcyclecounter();
        }
      }
    }
    return;
  }

  // ================================================================== consumer waves: weight DMA, MFMA, output transform
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.wbytes, 0x00020000);
  const int wm = wid / NB, wn = wid % NB;
  const int lr = lane & 31, lh = lane >> 5;
  if (p.splitk > 1) {
    p.y = p.ws + (long)blockIdx.y * ((long)p.Mt * 4) * p.N;
    p.ldy = p.N; p.bias = nullptr; p.res = nullptr;
    const long sb = (long)p.Mt * 4 * p.N * 4;
    p.ybytes = sb < (1L << 31) ? (int)sb : 0;
  }
  // B loader (LDS-DMA): 24 one-KB instructions per stage = (ex, term) image pt x 32-row half; wave w issues q = 6w .. 6w+5.
  // Lane l of an instruction covers row (q & 1) * 32 + (l >> 1), 16-byte half (l & 1) of the 32-byte row.
  constexpr int QW = F::DMA_PER_WAVE;                 // 6 (bf16) / 4 (fp16) one-KB instructions per wave and stage
  unsigned b_voff[QW];
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int q = wid * QW + i, pt = q / NB;          // image (ex, term) and 32-row block q % NB of it
    const int row = (q % NB) * 32 + (lane >> 1);
    const int n = n0 + row;
    const int half = (lane ^ (row >> 3)) & 1;           // logical half stored at physical half (lane & 1)
    b_voff[i] = (n < p.wrows) ? (unsigned)((((long)pt * p.wrows + n) * 16 + half * 8) * 2) : OOB;
  }
  X6Seq lb; lb.init(chunks, p.up);
  int ld_slot = 0;
  auto issue_b = [&]() {                              // weights of the next stage of the sequence -> next ring slot
    const int kb = ((lb.ey * (p.Cin >> 4) + c_begin + lb.chunk()) * (4 * TERMS) * p.wrows) << 5;   // (ey, chunk) block of twelve [ex][term] images of wrows x 32 bytes
    unsigned short* dst = Bs + ld_slot * F::B_STAGE + (wid * QW) * 512;           // 512 elements = one KB per instruction
    if (!(X6_ABL & 4)) {
#pragma unroll
      for (int i = 0; i < QW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (x6_lds_void*)(dst + i * 512), 16, (int)b_voff[i], kb, 0, 0);
    }
    lb.next(chunks);
    if (++ld_slot == F::RB) ld_slot = 0;
  };

  f32x16 acc[4];
  f32x16 Y[2][2];                                     // [output row][output column of the pair]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[a][b][r] = 0.f;
  const int a_foff = (wm * 32 + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;   // fragment: row = tile / cout, 8 bf16 = 16 bytes at k = 8 (lane >> 5)
  const int b_foff = (wn * 32 + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  X6Seq cs; cs.init(chunks, p.up);
  int slot_b = 0;
#pragma unroll
  for (int i = 0; i < F::RB - 1; ++i) issue_b();
  for (int t = 0; t < S; ++t) {
    const bool tl = X6_TL && p.ws && blockIdx.x == 0 && tid == 0 && t < 48;
    unsigned long long* TL = reinterpret_cast<unsigned long long*>(p.ws) + t * 4;
    if (tl) TL[0] = __builtin_readcyclecounter();
    // B(t) was issued three stages ago; B(t+1) and B(t+2) (six instructions each) may still be in flight.  Plain s_barrier +
    // explicit counters: a release fence would drain the weight prefetch (vmcnt(0)).
    if (F::RB == 4 && t + 2 < S) { if (FMT) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    else if (t + 1 < S) { if (FMT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tl) TL[1] = __builtin_readcyclecounter();
    if (t + F::RB - 1 < S) issue_b();
    const unsigned short* Ab = As + (t & 1) * F::A_STAGE + a_foff;
    const unsigned short* Bb = Bs + slot_b * F::B_STAGE + b_foff;
    if (++slot_b == F::RB) slot_b = 0;
    const bool first = cs.cc == 0;                    // first stage of a (block, ey) group: C = 0, no accumulator clearing
    // Fragment reads are issued ONE ex GROUP AHEAD of the MFMAs that use them (two register sets): left to itself the compiler
    // reads a group's six fragments right before its six MFMAs, so every group starts with an exposed LDS round trip (~130 cycles,
    // four times per stage) -- a read returns while the matrix pipe works only if it was issued before the chain it follows
    // (tools/overlap_probe2.hip: ds_read_b128 interleaved with MFMAs of the same wave costs ~6 cycles each, not a latency).
    u32x4 fa[NB > 2 ? 1 : 2][TERMS], fb[NB > 2 ? 1 : 2][TERMS];            // 8 bf16 / fp16 values per fragment
    auto frag = [&](int xi, int set) {
#pragma unroll
      for (int k = 0; k < TERMS; ++k) {
        if (X6_ABL & 16) {
          fa[set][k] = u32x4{0x3f803f80u, (unsigned)t, 0x3f803f80u, (unsigned)xi};
          fb[set][k] = u32x4{0x3f003f00u, (unsigned)k, 0x3f003f00u, (unsigned)lane};
          continue;
        }
        fa[set][k] = *reinterpret_cast<const u32x4*>(Ab + (xi * TERMS + k) * X6P_T * X6K);
        fb[set][k] = *reinterpret_cast<const u32x4*>(Bb + (xi * TERMS + k) * F::NT * X6K);
      }
    };
    auto products = [&](auto first_tag) {
      if (NB > 2) {
        // two consumer waves per SIMD; ONE fragment set: a second set (the 64-cout form's read-ahead) and a zero vector for the first
        // stage of a group do not fit next to 64 + 64 accumulator registers in 168 -- written with a read-ahead of the next group
        // (whole, or only of the fragments that are still live) the compiler spills into the loop: 0.80 -> 1.27 ms
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
          frag(xi, 0);
          if (X6_ABL & 8) continue;
          const f16x8 a0 = __builtin_bit_cast(f16x8, fa[0][0]), a1 = __builtin_bit_cast(f16x8, fa[0][1]);
          const f16x8 b0 = __builtin_bit_cast(f16x8, fb[0][0]), b1 = __builtin_bit_cast(f16x8, fb[0][1]);
          f32x16 c;
          if (decltype(first_tag)::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a0), "v"(b1));
          else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[xi], 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
        }
        return;
      }
      frag(0, 0);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const int cur = xi & 1;
        if (xi < 3) frag(xi + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);            // (keeps the next group's reads above this group's chain)
        if (X6_ABL & 8) continue;
        const f32x16 c0 = decltype(first_tag)::value ? zero : acc[xi];
        if (FMT) {                                  // three fp16 products, small ones first
          const f16x8 a0 = __builtin_bit_cast(f16x8, fa[cur][0]), a1 = __builtin_bit_cast(f16x8, fa[cur][1]);
          const f16x8 b0 = __builtin_bit_cast(f16x8, fb[cur][0]), b1 = __builtin_bit_cast(f16x8, fb[cur][1]);
          f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c0, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
        } else {                                    // six bf16 products, small ones first
          bf16x8 a[3], b[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) { a[k] = __builtin_bit_cast(bf16x8, fa[cur][k % TERMS]); b[k] = __builtin_bit_cast(bf16x8, fb[cur][k % TERMS]); }
          f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c0, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (first) products(std::true_type{});            // (wave-uniform)
    else products(std::false_type{});
    if (tl) TL[2] = __builtin_readcyclecounter();
    const int ey = cs.ey;
    const bool last = cs.cc + 1 == cs.len;
    cs.next(chunks);
    if (last) {
      // end of a (block, ey) group: A^T along x (z0 = m0 + m1 + m2, z1 = m1 - m2 - m3), then A^T along y into the output rows
      // (row 0 += Z(ey = 0, 1, 2); row 1 += Z(1) - Z(2) - Z(3)) with f32 adds
      const f32x16 z0 = acc[0] + acc[1] + acc[2], z1 = sub16(sub16(acc[1], acc[2]), acc[3]);
      if (ey <= 2) { Y[0][0] += z0; Y[0][1] += z1; }
      if (ey == 1) { Y[1][0] += z0; Y[1][1] += z1; }
      if (ey >= 2) { Y[1][0] = sub16(Y[1][0], z0); Y[1][1] = sub16(Y[1][1], z1); }
    }
  }

  // ---- epilogue.  C/D layout col = lane&31 (cout), row = (r&3) + 8 (r>>2) + 4 (lane>>5) (tile)
  const int n = n0 + wn * 32 + lr;
  const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
  const int tb = mt0 + wm * 32 + 4 * lh;
  if (p.ybytes > 0) {
    // branch-free: residual loads and output stores through buffer descriptors, masked lanes at an out-of-range offset (a missing
    // residual = an empty descriptor: reads return 0).  With `if`s per element the compiler serialises the 64 residual loads of a
    // thread, each waiting for the previous one -- tens of microseconds per workgroup.
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.rbytes : 0, 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tb + (r & 3) + 8 * (r >> 2);
      const bool ok = t < p.Mt && n < p.N;
      const int xp = t % p.Wh;
      const int u = t / p.Wh;                      // = b * Hh + ty
      const unsigned px0 = ((unsigned)u * 2u) * (unsigned)p.W + 2u * (unsigned)xp;     // pixel (b, 2ty, 2xp): (b*H + 2ty) * W + 2xp
      unsigned oy[2][2], orr[2][2];
      float rv[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned px = px0 + (unsigned)a * (unsigned)p.W + (unsigned)c;
          oy[a][c] = ok ? (px * (unsigned)p.ldy + (unsigned)n) * 4u : OOB;
          orr[a][c] = ok ? (px * (unsigned)p.ldr + (unsigned)n) * 4u : OOB;
          rv[a][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (int)orr[a][c], 0, 0));
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (FMT ? Y[a][c][r] * inv_scale : Y[a][c][r]) + bv + rv[a][c]), rs_y, (int)oy[a][c], 0, 0);
    }
    return;
  }
  if (n >= p.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = tb + (r & 3) + 8 * (r >> 2);
    if (t >= p.Mt) continue;
    const int xp = t % p.Wh;
    const int u = t / p.Wh;                        // = b * Hh + ty
    const long px0 = ((long)u * 2) * p.W + 2 * xp; // pixel (b, 2ty, 2xp) in units of pixels: (b*H + 2ty) * W + 2xp
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const long px = px0 + (long)a * p.W;
      float y0 = (FMT ? Y[a][0][r] * inv_scale : Y[a][0][r]) + bv, y1 = (FMT ? Y[a][1][r] * inv_scale : Y[a][1][r]) + bv;
      if (p.res) { y0 += p.res[px * p.ldr + n]; y1 += p.res[(px + 1) * p.ldr + n]; }
      p.y[px * p.ldy + n] = y0;
      p.y[(px + 1) * p.ldy + n] = y1;
    }
  }
}

// Operand layout of the kernel: Wq6[ey][chunk][ex][term][n][16] -- the twelve 16-channel images a stage needs for its 64 rows are
// 64 x 32 contiguous bytes each, so one LDS-DMA instruction reads one whole KB (plane-major rows would be 32-byte pieces of 32
// different 128-byte lines per instruction).
__global__ void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols) {
  const long per = (long)rows * cols, total = per * 16;
  const int chunks = cols >> 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int img = (int)(i / per);
    const long rc = i - img * per;
    const int n = (int)(rc / cols), c = (int)(rc - (long)n * cols);
    const int ey = img >> 2, ex = img & 3;
    const float a = src[i];
    const unsigned u = __float_as_uint(a);
    const float r = a - __uint_as_float(u & 0xFFFF0000u);
    const unsigned m = __float_as_uint(r);
    const float r2 = r - __uint_as_float(m & 0xFFFF0000u);
    unsigned short* d = dst + ((((long)(ey * chunks + (c >> 4)) * 12 + ex * 3) * rows + n) << 4) + (c & 15);
    const long term = (long)rows << 4;
    d[0] = (unsigned short)(u >> 16);
    d[term] = (unsigned short)(m >> 16);
    d[2 * term] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
}

// fp16 format: dst[ey][cols/16][ex][term(2)][rows][16] <- the two-term round-to-nearest split of scale * src (the sixteen Winograd
// planes); *overflow is raised when a scaled value leaves the fp16 range (the caller then falls back to the bf16 format)
__global__ void split2_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols, float scale,
                              int* __restrict__ overflow) {
  const long per = (long)rows * cols, total = per * 16;
  const int chunks = cols >> 4;
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int img = (int)(i / per);
    const long rc = i - img * per;
    const int n = (int)(rc / cols), c = (int)(rc - (long)n * cols);
    const int ey = img >> 2, ex = img & 3;
    const float a = src[i] * scale;
    bad |= !(fabsf(a) < 65000.f);
    const _Float16 h0 = (_Float16)a, h1 = (_Float16)(a - (float)h0);
    unsigned short* d = dst + ((((long)(ey * chunks + (c >> 4)) * 8 + ex * 2) * rows + n) << 4) + (c & 15);
    d[0] = __builtin_bit_cast(unsigned short, h0);
    d[(long)rows << 4] = __builtin_bit_cast(unsigned short, h1);
  }
  if (bad && overflow) *overflow = 1;
}

}  // namespace

int adm_splitk_reduce(const float* ws, const float* bias, const float* res, float* y, long M, int N, int ldy, int ldr, int splitk,
                      hipStream_t stream);       // conv_igemm.hip

// dst (bf16 bit patterns, 48 * rows * cols of them, layout [ey][cols/16][ex][term][rows][16]) <- the exact three-term split
// a = a0 + a1 + a2 of the sixteen Winograd planes src[ey * 4 + ex][rows][cols] (f32); cols % 16 == 0
extern "C" int adm_split3_bf16(const float* src, void* dst, int rows, int cols, hipStream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 15)) return ADM_EINVAL;
  const long total = (long)rows * cols * 16;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, stream, src, static_cast<unsigned short*>(dst), rows, cols);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// fp16 format of the weight operand (see X6Fmt<1>): dst = 32 * rows * cols fp16 values, layout [ey][cols/16][ex][term(2)][rows][16]
extern "C" int adm_split2_f16(const float* src, void* dst, int rows, int cols, float scale, int* overflow, hipStream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 15) || !(scale > 0.f)) return ADM_EINVAL;
  const long total = (long)rows * cols * 16;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(split2_kernel, dim3(grid), dim3(256), 0, stream, src, static_cast<unsigned short*>(dst), rows, cols, scale, overflow);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// Same contract as adm_conv_fwd_wino2d, with wq6 = adm_split3_bf16 of the adm_pack_weight_wino2d operand (16 planes of
// wrows x Cin).
static int g_h3_wide = -1;      // -1: chosen per launch, 0: never, 1: whenever the launch qualifies (fp16 format, no split-K)
extern "C" int adm_wino2d_h3_wide(int v) { const int old = g_h3_wide; g_h3_wide = v; return old; }

static int wino2d_x6_launch(const float* x, const void* wq6, const float* bias, const float* res, float* y, float* ws, long ws_floats,
                            int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr, int up, hipStream_t stream,
                            const float* amax_x = nullptr, float wscale = 0.f) {
  const bool h3 = amax_x != nullptr;                   // fp16 format: wq6 is the adm_split2_f16 image (scale wscale), amax_x >= max |x| on the device
  if (!x || !wq6 || !y || B <= 0 || H < 2 || W < 2 || (W & 1) || (H & 1)) return ADM_EINVAL;
  if ((Cin & 31) || (ldx & 3) || N <= 0 || wrows < N) return ADM_EINVAL;      // an even number of 16-channel chunks
  if (((uintptr_t)x | (uintptr_t)wq6) & 15) return ADM_EINVAL;
  X6P p;
  p.x = x; p.w = static_cast<const unsigned short*>(wq6); p.bias = bias; p.res = res; p.y = y;
  const long Mt = (long)B * (H / 2) * (W / 2);
  const long xb = (long)B * H * W * ldx * 4 / (up ? 4 : 1), wb = (h3 ? 32L : 48L) * wrows * Cin * 2;
  if (h3 && !(wscale > 0.f)) return ADM_EINVAL;
  if (Mt >= (1L << 30) || xb >= (1L << 31) || wb >= (1L << 31)) return ADM_EINVAL;
  p.Mt = (int)Mt; p.N = N; p.H = H; p.W = W; p.Hh = H / 2; p.Wh = W / 2; p.Cin = Cin; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr;
  p.wrows = wrows; p.xbytes = (int)xb; p.wbytes = (int)wb; p.plane = wrows * Cin; p.up = up ? 1 : 0;
  p.amax_x = amax_x; p.wscale = wscale;
  p.tilesN = adm_cdiv(N, X6N);
  const long mtiles = adm_cdiv(Mt, X6P_T);
  p.splitk = 1; p.chunks_per_split = 0; p.ws = X6_TL ? ws : nullptr;
  const long yb = (long)B * H * W * ldy * 4, rb = res ? (long)B * H * W * ldr * 4 : 0;
  p.ybytes = (yb < (1L << 31) && rb < (1L << 31)) ? (int)yb : 0;
  p.rbytes = (int)rb;
  const int sk = (ws && !(N & 3) && !(ldy & 3) && (!res || !(ldr & 3))) ? adm_wino2d_x6_splitk(B, H, W, Cin, N) : 1;
  if (sk > 1 && ws_floats >= (long)sk * Mt * 4 * N) {
    const int chunks = Cin >> 4;
    p.chunks_per_split = ((chunks + sk - 1) / sk + 1) & ~1;        // the producer double-buffers chunk pairs
    p.splitk = (chunks + p.chunks_per_split - 1) / p.chunks_per_split;
    p.ws = ws;
  }
  constexpr int smem0 = (X6_RA * X6Fmt<0>::A_STAGE + X6Fmt<0>::RB * X6Fmt<0>::B_STAGE) * (int)sizeof(unsigned short);
  constexpr int smem1 = (X6_RA * X6Fmt<1>::A_STAGE + X6Fmt<1>::RB * X6Fmt<1>::B_STAGE) * (int)sizeof(unsigned short);
  constexpr int smemw = (X6_RA * X6Fmt<1, 4>::A_STAGE + X6Fmt<1, 4>::RB * X6Fmt<1, 4>::B_STAGE) * (int)sizeof(unsigned short);
  constexpr int smem3 = (X6_RA * X6Fmt<1, 3>::A_STAGE + X6Fmt<1, 3>::RB * X6Fmt<1, 3>::B_STAGE) * (int)sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_x6_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem0) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_x6_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem1) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_x6_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, smemw) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_x6_kernel<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem3) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  // 64, 96 or 128 couts per workgroup (four, six, eight consumer waves).  What a CU fetches per stage bounds all three forms: 32 KB of
  // activations + 16 / 24 / 32 KB of weights, 24 B/clk at the measured stage times against the ~30 B/clk a CU gets out of its L2; one
  // workgroup per CU, so a launch takes (rounds of 256 workgroups) x (bytes per stage) -- the 96 form priced a little above its bytes
  // (its six consumer waves load two SIMDs more than the others).  This model orders every shape of tools/bench_wino2d_x6.cpp
  // correctly (B = 128, ms for 64 / 96 / 128 couts: 32x32 384->384 0.91 / 0.90 / 0.81, 192->192 0.27 / 0.26 / 0.29, 576->192
  // 0.72 / 0.60 / 0.65, 192->576 0.75 / 0.63 / 0.59; 16x16 384->384 0.26 / 0.20 / 0.22; 8x8 384->384 0.083 / 0.100 / 0.108).
  int form = 2;
  if (h3 && p.splitk == 1 && g_h3_wide != 0 && N > X6N) {
    if (g_h3_wide == 1) form = 4;
    else if (g_h3_wide == 3) form = 3;
    else {
      const long cost[3] = {48, 60, 64};
      long best = -1;
      for (int nb = 2; nb <= 4; ++nb) {
        const long t = (mtiles * adm_cdiv(N, 32 * nb) + 255) / 256 * cost[nb - 2];
        if (best < 0 || t < best) { best = t; form = nb; }
      }
    }
  }
  const bool three = form == 3, wide = form == 4;
  if (three) {
    p.tilesN = adm_cdiv(N, X6Fmt<1, 3>::NT);
    hipLaunchKernelGGL((wino2d_x6_kernel<1, 3>), dim3((unsigned)(mtiles * p.tilesN), 1), dim3(X6Fmt<1, 3>::THREADS), smem3, stream, p);
  } else if (wide) {
    p.tilesN = adm_cdiv(N, X6Fmt<1, 4>::NT);
    hipLaunchKernelGGL((wino2d_x6_kernel<1, 4>), dim3((unsigned)(mtiles * p.tilesN), 1), dim3(X6Fmt<1, 4>::THREADS), smemw, stream, p);
  } else if (h3) {
    hipLaunchKernelGGL((wino2d_x6_kernel<1, 2>), dim3((unsigned)(mtiles * p.tilesN), p.splitk), dim3(512), smem1, stream, p);
  } else {
    hipLaunchKernelGGL((wino2d_x6_kernel<0, 2>), dim3((unsigned)(mtiles * p.tilesN), p.splitk), dim3(512), smem0, stream, p);
  }
  ADM_CHECK_LAUNCH();
  if (p.splitk > 1) return adm_splitk_reduce(ws, bias, res, y, Mt * 4, N, ldy, ldr, p.splitk, stream);
  return ADM_OK;
}

// fp16 format (three fp16 products per f32 product): wqh = adm_split2_f16(planes, scale = wscale), amax_x = device scalar >= max |x|
// (an upper bound; too large only costs precision on the smallest values); up != 0: the fused nearest x2 form
extern "C" int adm_conv_fwd_wino2d_h3(const float* x, const void* wqh, const float* bias, const float* res, float* y, float* ws,
                                      long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                                      const float* amax_x, float wscale, int up, hipStream_t stream) {
  if (!amax_x) return ADM_EINVAL;
  return wino2d_x6_launch(x, wqh, bias, res, y, ws, ws_floats, B, H, W, Cin, ldx, N, wrows, ldy, ldr, up, stream, amax_x, wscale);
}

extern "C" int adm_conv_fwd_wino2d_x6(const float* x, const void* wq6, const float* bias, const float* res, float* y, float* ws,
                                      long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                                      hipStream_t stream) {
  return wino2d_x6_launch(x, wq6, bias, res, y, ws, ws_floats, B, H, W, Cin, ldx, N, wrows, ldy, ldr, 0, stream);
}

// Conv2d(up=True): the same convolution on the nearest x2 up-sampling of x[B][H/2][W/2][ldx]; H x W is the OUTPUT grid (even)
extern "C" int adm_conv_fwd_wino2d_x6_up(const float* x, const void* wq6, const float* bias, const float* res, float* y, float* ws,
                                         long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                                         hipStream_t stream) {
  return wino2d_x6_launch(x, wq6, bias, res, y, ws, ws_floats, B, H, W, Cin, ldx, N, wrows, ldy, ldr, 1, stream);
}

// Split count over the input channels for launches with fewer workgroups than CUs (one workgroup per CU here)
extern "C" int adm_wino2d_x6_splitk(int B, int H, int W, int Cin, int N) {
  const long wgs = (long)adm_cdiv((long)B * (H / 2) * (W / 2), X6P_T) * adm_cdiv(N, X6N);
  const int chunks = Cin >> 4;
  if (wgs >= 192 || chunks < 8) return 1;
  // one workgroup per CU: the launch takes (rounds of 256 workgroups) x (stages of a split + ~12 stages' worth of prologue and
  // epilogue).  Round 2 took min(768 / wgs, chunks / 4, 6) splits: 48 workgroups x 6 = 288 = a second round for 32 of them.
  int best = 1;
  long best_t = (wgs + 255) / 256 * (4L * chunks + 12);
  for (int s = 2; s <= 6 && s <= chunks / 4; ++s) {
    const int cps = ((chunks + s - 1) / s + 1) & ~1;            // (the producer double-buffers chunk pairs)
    const int splits = (chunks + cps - 1) / cps;
    const long t = (wgs * splits + 255) / 256 * (4L * cps + 12);
    if (t < best_t) { best_t = t; best = s; }
  }
  return best;
}
