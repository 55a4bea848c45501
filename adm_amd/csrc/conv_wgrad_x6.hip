// Weight gradient of the 3x3 stride-1 convs: transposed 2-D Winograd F(3x3, 2x2) (conv_wgrad_wino.hip, MODE 2) with the f32 products
// carried on the bf16 MFMA through the exact three-term bf16 split of BOTH operands (see conv_wino2d_x6.hip for the arithmetic:
// a = a0 + a1 + a2 exactly, six bf16 products per f32 product, f32 accumulation, error at the f32 MFMA's level).
//
//   per 2x2 tile of dY and the 4x4 patch of X around it:  a = A e A^T,  b = B^T d B,  m[ey][ex] = sum_tiles a[ey][ex] (x) b[ey][ex],
//   dW = G^T m G.   As in the f32 kernel the pass `ey` belongs to the workgroup: it reduces over TILES the y-combined rows
//   (r0, r0 + r1, r0 - r1, -r1)[ey] of dY and (r0 - r2, r1 + r2, r2 - r1, r1 - r3)[ey] of X through the x transforms (A e, B^T d), runs
//   the four ex products, applies G^T along x in the epilogue and writes wx[co][ey][kx][ci]; adm_unpack_wgrad_wino2d (or the
//   end-of-backward adm_unpack_wgrad_table) applies G^T along y.  The grid is one-dimensional, (split, ey, tile) in split-major
//   order behind an XCD-aware remap.
//
// The reduction index (tiles) is the K of the MFMA for both operands, and both arrive tile-major from HBM.  The f32 kernel
// transposes them with 32 ds_write_b32 per thread and stage; here they stay tile-major in LDS ([tile][64 channels] bf16 rows of 192
// bytes: 128 of data + 64 of padding, which makes the reads below conflict-free) and the consumer reads them TRANSPOSED with
// ds_read_b64_tr_b16 (a 16-lane group gets 4 tiles x 16 channels column-major: lane i receives channel i of four consecutive
// tiles, exactly the k-run a 32x32x16 operand lane needs; tools/tr_probe.hip).
//
// Wave-specialised like conv_wino2d_x6.hip (768 threads = twelve waves, one workgroup per CU):
//   * waves 4-7 PRODUCE the dY-side operand, waves 8-11 the X-side operand: thread = (tile of the stage, 16-byte channel quad), a wave
//     covers 4 tiles x 64 channels = four whole 256-byte runs per load instruction; loads of a stage (<= 4 of dY / 8 of X) are issued
//     three stages ahead into one of three register sets; y combination, x transform (f32), three-term split, 12 ds_write_b64 per stage;
//   * waves 0-3 CONSUME: wave w owns the ex = w plane of the whole 64 x 64 tile (2 x 2 blocks of 32 x 32: every fragment feeds two
//     block products); per stage 24 transposed fragment reads and 24 MFMAs: six products per block from C = 0, then one rounded f32
//     add into the running totals (the long sums see rounded f32 adds, not truncating matrix adds); the four planes meet in LDS once,
//     after the last stage, for G^T along x;
//   * the producers run at s_setprio 3: VALU work and the bf16 MFMA of different waves do not overlap on a SIMD
//     (tools/overlap_probe.hip), a stage costs their sum, and the producers are the longer dependency chain;
//   * one s_barrier per stage with LDS-only counters: the producers' loads stay in flight across it.
// Replaces the autograd weight gradient of Conv2d.forward (/root/reference/unet/uncond_unet.py:98-110).
#include <algorithm>
#include "common.h"
#include "../../include/adm_hip.h"

#ifndef XW_D
#define XW_D (CB == 2 ? 2 : 3)      // producer register sets = stages of loads in flight (sixteen waves: 128 registers, two sets)
#endif
#ifndef XW_ABL
#define XW_ABL 0      // diagnostic builds (tools/bench_wgrad_x6.cpp): 1 no global loads, 2 no transform / split / LDS stores, 8 no MFMAs, 16 no fragment reads
#endif

namespace {

struct WxP {
  const float* x; const float* dy; float* dwp; float* dbias;
  int Pp, H, W, lw, Cin, ldx, Cout, lddy, tilesN, chunk, atomic, xbytes, dybytes;     // Pp = 2x2 tiles in all; chunk = tiles per split
  long split_stride, bias_stride;      // > 0: deterministic mode, partials of split z at dwp + z * split_stride (plain stores)
  int up;                              // 1: x is [B][H/2][W/2][ldx], the conv ran on its nearest x2 up-sampling (Conv2d(up=True))
  int tiles;                           // (cout tiles) x (cin tiles): the grid is tiles x passes x splits, one-dimensional
  const float* amax_x; const float* amax_dy;     // FMT 1 only: device upper bounds of |x| and |dy| (the per-tensor scales come from them)
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int XT = 64, XK = 16;                    // 64 x 64 channel tile, 16 tiles (2x2 pixels each) per stage
constexpr int XROW = 96;                           // bf16 elements per LDS row: 64 channels + 32 of padding (192 bytes)
// FMT 0: three bf16 terms by truncation, six products.  FMT 1: s v = h0 + h1, two fp16 terms (round to nearest), s a power of two
// from an upper bound of the tensor's |v| (conv_wino2d_x6.hip: s max|v| <= 16000, the transforms' sums of four stay finite); three
// products (h0 h0' + h0 h1' + h1 h0'), the scales undone once in the epilogue.  Same accuracy (tools/fp16x3_accuracy.py).
template <int FMT> struct XFmt {
  static constexpr int TERMS = FMT ? 2 : 3;
  static constexpr int IMG = 4 * TERMS * XK * XROW;     // one operand image of a stage: [4 ex][TERMS][16 tiles][XROW] = 36 / 24 KB
};
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ inline float xh3_scale(float amax) {       // = h3_scale of conv_wino2d_x6.hip
  if (!(amax > 0.f) || !(amax < 3e38f)) return 1.f;
  int e;
  frexpf(16000.f / amax, &e);
  return ldexpf(1.f, e - 1);
}

// All the arithmetic next to the MFMAs is written with PLAIN (one value per lane) f32 instructions: tools/overlap_probe2.hip
// measures v_add_f32 / v_and_b32 / v_perm_b32 of another wave 91-96 % hidden behind v_mfma_f32_32x32x16_bf16 on the same SIMD, and
// the packed forms (v_pk_add_f32, v_pk_fma_f32) not at all -- they share the matrix pipe's data path, so "half the instructions"
// costs the whole instruction.  The compiler packs every pair of f32 adds it sees, hence the inline assembly.
__device__ __forceinline__ float x_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float x_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 add4x(f32x4 a, f32x4 b) { return f32x4{x_add(a[0], b[0]), x_add(a[1], b[1]), x_add(a[2], b[2]), x_add(a[3], b[3])}; }
__device__ __forceinline__ f32x4 sub4x(f32x4 a, f32x4 b) { return f32x4{x_sub(a[0], b[0]), x_sub(a[1], b[1]), x_sub(a[2], b[2]), x_sub(a[3], b[3])}; }
// v = v0 + v1 + v2 exactly, each term a bf16 (packed top halves: two dwords per term for the four channels)
__device__ __forceinline__ void split3x(const f32x4 v, u32x2& t0, u32x2& t1, u32x2& t2) {
  f32x4 h, mh;
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = __uint_as_float(__float_as_uint(v[i]) & 0xFFFF0000u);
  const f32x4 r = sub4x(v, h);
#pragma unroll
  for (int i = 0; i < 4; ++i) mh[i] = __uint_as_float(__float_as_uint(r[i]) & 0xFFFF0000u);
  const f32x4 r2 = sub4x(r, mh);
  t0 = u32x2{__builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07060302u)};
  t1 = u32x2{__builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(r[3]), __float_as_uint(r[2]), 0x07060302u)};
  t2 = u32x2{__builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u)};
}
// a + sgn b, sgn = +-1 uniform (exact; one fma per value -- a uniform "add or subtract" written as a conditional costs a branch per use)
__device__ __forceinline__ f32x4 fma4s(f32x4 a, f32x4 b, f32x2 sgn) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(b[i]), "v"(sgn[0]), "v"(a[i]));
  return r;
}
__device__ __forceinline__ void split2x(const f32x4 v, float s, u32x2& t0, u32x2& t1) {
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
template <int FMT>
__device__ __forceinline__ void store_planes(const f32x4 (&v)[4], unsigned short* l, float s) {
  if (FMT) {
#pragma unroll
    for (int ex = 0; ex < 4; ++ex) {
      u32x2 t0, t1;
      split2x(v[ex], s, t0, t1);
      *reinterpret_cast<u32x2*>(l + (ex * 2 + 0) * XK * XROW) = t0;
      *reinterpret_cast<u32x2*>(l + (ex * 2 + 1) * XK * XROW) = t1;
    }
    return;
  }
#pragma unroll
  for (int ex = 0; ex < 4; ++ex) {
    u32x2 t0, t1, t2;
    split3x(v[ex], t0, t1, t2);
    *reinterpret_cast<u32x2*>(l + (ex * 3 + 0) * XK * XROW) = t0;
    *reinterpret_cast<u32x2*>(l + (ex * 3 + 1) * XK * XROW) = t1;
    *reinterpret_cast<u32x2*>(l + (ex * 3 + 2) * XK * XROW) = t2;
  }
}
__device__ __forceinline__ void lds_barrier() {    // waits for this wave's LDS traffic only
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// MODE 0: the 3x3 Winograd form above.  MODE 1: the weight gradient of a 1x1 conv, dW[co][ci] = sum_pixels dY[p][co] X[p][ci]: the same
// images, reads and MFMAs with the four `ex` planes of a stage holding four consecutive 16-pixel chunks (64 pixels per stage, no
// transforms, gridDim.y = 1); the epilogue adds the four accumulators.
// XBF: x holds bf16 (the bf16-storage activations of the opt-in bf16 mode): 8-byte loads, widened in the producer
// CB = 64-cout blocks per workgroup.  2 (fp16 format only): 128 couts x 64 cins -- the dY side is TWO 64-cout images, produced by the
// same four waves (second register set: the one the X side uses for its second row), consumed by eight waves (plane x half): sixteen
// waves, 128 registers.  The X rows a workgroup fetches and transforms then feed twice the MFMAs: like the forward kernel this one
// runs at the rate a CU gets bytes out of its L2 (48 KB per stage and 64-cout block before, 21 B/clk at the measured stage time).
template <int MODE, bool XBF, int FMT, int CB>
__global__ __launch_bounds__((4 * CB + 8) * 64) void wgrad_x6_kernel(WxP p) {
  constexpr int X_IMG = XFmt<FMT>::IMG, TERMS = XFmt<FMT>::TERMS;
  constexpr int NCONS = 4 * CB;
  float s_x = 1.f, s_dy = 1.f;
  if (FMT) { s_x = xh3_scale(adm_amax_read(p.amax_x)); s_dy = xh3_scale(adm_amax_read(p.amax_dy)); }
  extern __shared__ __attribute__((aligned(16))) unsigned short smx[];
  unsigned short* As = smx;                        // [CB][2][X_IMG]  dY side: rows = tiles, columns = couts (one image per 64-cout block)
  unsigned short* Bs = smx + CB * 2 * X_IMG;       // [2][X_IMG]  X side:  rows = tiles, columns = cins
  const int tid = threadIdx.x, lane = tid & 63, hw_wid = tid >> 6;
  // twelve waves, three per SIMD: 0-3 consume, 4-7 produce the dY-side operand, 8-11 the X-side operand.  With both operands on
  // ONE producer group the kernel was producer-bound (ablation: producers alone 0.373 of 0.448 ms, consumers alone 0.183), and a
  // producer wave is latency-bound, not VALU-throughput-bound (its ~200 instructions per stage use 22 % of the SIMD's issue slots)
  const bool producer = hw_wid >= NCONS;
  const int wid = producer ? (hw_wid - NCONS) & 3 : hw_wid;      // consumers: plane = wid & 3, 64-cout block = wid >> 2
  // XCD-aware bijective remap: the hardware deals workgroup i to XCD i % 8; the logical order is split-major (z, ey, tile), so the
  // workgroups of one XCD walk ONE pixel range (and its neighbours) together -- all of a range's (cout tile, cin tile, ey)
  // workgroups read the same x and dy rows.  Dealt round-robin, every XCD streamed every range through its own 4 MB L2
  // (47 % L2 misses, 0.6 GB per launch from the Infinity Cache for 0.2 GB of operands) and the loads, not the SIMDs, set the pace
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int NEY = MODE == 1 ? 1 : 4;
  const int bz = bid / (p.tiles * NEY), brem = bid - bz * (p.tiles * NEY);
  const int by = brem / p.tiles, bx = brem - by * p.tiles;
  const int tn = bx % p.tilesN, tm = bx / p.tilesN;
  const int co0 = tm * XT * CB, ci0 = tn * XT;
  const int ey = by;
  const int pbeg = bz * p.chunk;
  const int pend = min(p.Pp, pbeg + p.chunk);
  if (pbeg >= pend) return;                        // (whole workgroup)
  constexpr int STEP = MODE == 1 ? 4 * XK : XK;    // reduction items (tiles / pixels) per stage
  int KT = (pend - pbeg + STEP - 1) / STEP;        // stages
  // fused nearest x2: patch rows r1 and r2 are the same source row, so the X side of the pass ey = 2 (r2 - r1) is identically zero;
  // these workgroups only write their zeros (plain-store modes) -- no stages, no barriers, for either role
  if (MODE == 0 && p.up && ey == 2) KT = 0;
  constexpr unsigned OOB = 0x80000000u;

  // VALU work and the bf16 MFMA of different waves of one SIMD do not overlap (tools/overlap_probe.hip: the two times add), so a SIMD's
  // stage is its MFMA time PLUS its producers' VALU time; producers first (measured: 0.401 -> 0.379 ms; consumers first: no change)
  if (producer) __builtin_amdgcn_s_setprio(3);
  if (producer) {
    // ================================================================ one operand per producer group (side is compile-time)
    auto produce = [&](auto side_c) {
    constexpr bool x_side = decltype(side_c)::value;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
    const int tl = wid * 4 + (lane >> 4), quad = lane & 15;       // tile of the stage, channel quad
    const unsigned a_col = (co0 + quad * 4 < p.Cout) ? (unsigned)(co0 + quad * 4) * 4u : OOB;
    const unsigned a_col1 = (CB == 2 && co0 + XT + quad * 4 < p.Cout) ? (unsigned)(co0 + XT + quad * 4) * 4u : OOB;     // second 64-cout block
    constexpr unsigned XB = XBF ? 2u : 4u;          // bytes per x element
    const unsigned b_col = (ci0 + quad * 4 < p.Cin) ? (unsigned)(ci0 + quad * 4) * XB : OOB;
    const int Wh = p.W >> 1, Hh = p.H >> 1, lwh = p.lw - 1;
    const unsigned ystep = (unsigned)p.lddy * 4u, yrow = (unsigned)p.W * ystep;
    const unsigned xstep = (unsigned)p.ldx * XB;
    const int iA = (ey == 0) ? 0 : 1, iB = (ey == 3) ? 3 : 2;       // X rows of this pass: (r0 - r2, r1 + r2, r2 - r1, r1 - r3)[ey]
    // y combination of this pass as a + sgn b: dY rows (r0, r0 + r1, r0 - r1, -r1)[ey], X rows (r0 - r2, r1 + r2, -(r2 - r1), r1 - r3)[ey]
    const float sg = x_side ? (ey == 1 ? 1.f : -1.f) : (ey <= 1 ? 1.f : -1.f);
    const f32x2 sgn = {sg, sg};
    constexpr int D = XW_D;                        // stages in flight
    f32x4 u0[D][4], u1[D][4];                      // dY side: u0 = (r0 px0, r0 px1, r1 px0, r1 px1); X side: u0 / u1 = rows iA / iB
    // four channels of x at a byte offset: f32 = one 16-byte load; bf16 = one 8-byte load kept RAW in two lanes (widening at the load
    // would put a wait right behind it) and widened by wide4() when the set is consumed
    auto ldx4 = [&](unsigned off) -> f32x4 {
      if (!XBF) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
      const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, (int)off, 0, 0));
      return f32x4{__uint_as_float(r[0]), __uint_as_float(r[1]), 0.f, 0.f};
    };
    auto wide4 = [&](f32x4 v) -> f32x4 {
      if (!XBF) return v;
      const unsigned r0 = __float_as_uint(v[0]), r1 = __float_as_uint(v[1]);
      return f32x4{__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xFFFF0000u), __uint_as_float(r1 << 16), __uint_as_float(r1 & 0xFFFF0000u)};
    };
    auto issue = [&](int d, int s) {               // loads of stage s -> register set d (stages past the end read nothing)
      if (MODE == 1) {                             // chunk xi of the stage: pixel pbeg + 64 s + 16 xi + tl, one quad of dY or of X
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
          const int px = pbeg + s * STEP + xi * XK + tl;
          const bool pv = px < pend && s < KT;
          if (!x_side) {
            u0[d][xi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)((pv && a_col != OOB) ? (unsigned)px * ystep + a_col : OOB), 0, 0));
            if (CB == 2) u1[d][xi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)((pv && a_col1 != OOB) ? (unsigned)px * ystep + a_col1 : OOB), 0, 0));
          } else u0[d][xi] = ldx4((pv && b_col != OOB) ? (unsigned)px * xstep + b_col : OOB);
        }
        return;
      }
      if (XW_ABL & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { u0[d][j] = f32x4{1.f, 2.f, (float)s, (float)j}; u1[d][j] = f32x4{0.5f, 0.25f, (float)j, 1.f}; }
        return;
      }
      const int pr = pbeg + s * XK + tl;
      const bool pv = pr < pend && s < KT;
      const int xp = pr & (Wh - 1), ty = (pr >> lwh) & (Hh - 1);
      const unsigned pix = ((unsigned)(pr >> lwh) << (p.lw + 1)) + 2u * (unsigned)xp;      // top-left output pixel of the tile
      // dY rows 2ty (r0) and 2ty + 1 (r1): a_y = (r0, r0 + r1, r0 - r1, -r1)[ey]; a row the pass does not use is read out of range
      if (!x_side) {
        const unsigned ya = (pv && a_col != OOB) ? pix * ystep + a_col : OOB;
        const unsigned y0 = (ey != 3) ? ya : OOB, y1 = (ey != 0 && ya != OOB) ? ya + yrow : OOB;
        u0[d][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)y0, 0, 0));
        u0[d][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(y0 != OOB ? y0 + ystep : OOB), 0, 0));
        u0[d][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)y1, 0, 0));
        u0[d][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(y1 != OOB ? y1 + ystep : OOB), 0, 0));
        if (CB == 2) {
          const unsigned yb = (pv && a_col1 != OOB) ? pix * ystep + a_col1 : OOB;
          const unsigned z0 = (ey != 3) ? yb : OOB, z1 = (ey != 0 && yb != OOB) ? yb + yrow : OOB;
          u1[d][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)z0, 0, 0));
          u1[d][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(z0 != OOB ? z0 + ystep : OOB), 0, 0));
          u1[d][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)z1, 0, 0));
          u1[d][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(z1 != OOB ? z1 + ystep : OOB), 0, 0));
        }
        return;
      }
      const bool vA = pv && b_col != OOB && (iA != 0 || ty > 0), vB = pv && b_col != OOB && (iB != 3 || ty < Hh - 1);
      // up-sampled: the tile's source pixel is pixel `pr` of the half-resolution image; up-sampled row 2ty - 1 + i reads source row
      // ty + (i + 1) / 2 - 1 (columns likewise)
      const unsigned xa = p.up ? ((unsigned)pr + (unsigned)((((iA + 1) >> 1) - 1) * Wh)) * xstep + b_col
                               : pix * xstep + b_col + (unsigned)((iA - 1) * p.W - 1) * xstep;      // row iA, column 2xp - 1 (may wrap: masked)
      const unsigned xb = p.up ? ((unsigned)pr + (unsigned)((((iB + 1) >> 1) - 1) * Wh)) * xstep + b_col
                               : pix * xstep + b_col + (unsigned)((iB - 1) * p.W - 1) * xstep;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cv = (j != 0 || xp > 0) && (j != 3 || xp < Wh - 1);
        const unsigned cj = p.up ? (unsigned)(((j + 1) >> 1) - 1) * xstep : (unsigned)j * xstep;
        u0[d][j] = ldx4((vA && cv) ? xa + cj : OOB);
        u1[d][j] = ldx4((vB && cv) ? xb + cj : OOB);
      }
    };
    // the bias gradient = sum of every dY pixel = the ex = 1 component (e0 + e1 along x) of the ey = 1 workgroups (r0 + r1 along y)
    const bool bias_wg = p.dbias != nullptr && tn == 0 && ey == (MODE == 1 ? 0 : 1);     // (workgroup-uniform)
    const bool do_bias = bias_wg && !x_side;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, bsum1 = {0.f, 0.f, 0.f, 0.f};
    unsigned short* la = As + tl * XROW + quad * 4;
    unsigned short* lb = Bs + tl * XROW + quad * 4;
    auto store = [&](int d, int slot) {            // y combination, x transforms, split, into slot
      if (XW_ABL & 2) return;
      if (MODE == 1) {
        if (do_bias) bsum = add4x(bsum, add4x(add4x(u0[d][0], u0[d][1]), add4x(u0[d][2], u0[d][3])));
        if (do_bias && CB == 2) bsum1 = add4x(bsum1, add4x(add4x(u1[d][0], u1[d][1]), add4x(u1[d][2], u1[d][3])));
        if (!x_side && CB == 2) store_planes<FMT>(u1[d], la + (2 + slot) * X_IMG, s_dy);
        if (x_side && XBF) {
          const f32x4 w[4] = {wide4(u0[d][0]), wide4(u0[d][1]), wide4(u0[d][2]), wide4(u0[d][3])};
          store_planes<FMT>(w, lb + slot * X_IMG, s_x);
          return;
        }
        store_planes<FMT>(u0[d], (x_side ? lb : la) + slot * X_IMG, x_side ? s_x : s_dy);
        return;
      }
      if (!x_side) {
        const f32x4 e[2] = {fma4s(u0[d][0], u0[d][2], sgn), fma4s(u0[d][1], u0[d][3], sgn)};     // r0 +- r1 (an unused row was read as zeros)
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a[4] = {e[0], add4x(e[0], e[1]), sub4x(e[0], e[1]), sub4x(zero, e[1])};
        if (do_bias) bsum = add4x(bsum, a[1]);
        store_planes<FMT>(a, la + slot * X_IMG, s_dy);
        if (CB == 2) {
          const f32x4 f[2] = {fma4s(u1[d][0], u1[d][2], sgn), fma4s(u1[d][1], u1[d][3], sgn)};
          const f32x4 c[4] = {f[0], add4x(f[0], f[1]), sub4x(f[0], f[1]), sub4x(zero, f[1])};
          if (do_bias) bsum1 = add4x(bsum1, c[1]);
          store_planes<FMT>(c, la + (2 + slot) * X_IMG, s_dy);
        }
        return;
      }
      f32x4 dd[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dd[j] = fma4s(wide4(u0[d][j]), wide4(u1[d][j]), sgn);     // rows iA +- iB; the pass ey = 2 wants iB - iA: the epilogue negates
      const f32x4 b[4] = {sub4x(dd[0], dd[2]), add4x(dd[1], dd[2]), sub4x(dd[2], dd[1]), sub4x(dd[1], dd[3])};
      store_planes<FMT>(b, lb + slot * X_IMG, s_x);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    __builtin_amdgcn_sched_barrier(0);
    // barrier s separates "images(s) written" from compute(s); images(s) live in slot s & 1; set s % D is refilled with stage s + D
    for (int s0 = 0; s0 < KT; s0 += 2 * D) {       // 2 D = 6 stages per trip: static register sets and slots
#pragma unroll
      for (int k = 0; k < 2 * D; ++k) {
        if (s0 + k < KT) {                         // (uniform)
          store(k % D, k & 1);
          __builtin_amdgcn_sched_barrier(0);
          issue(k % D, s0 + k + D);
          __builtin_amdgcn_sched_barrier(0);
          lds_barrier();
        }
      }
    }
    if (bias_wg) {     // lanes l, l + 16, l + 32, l + 48 of a wave hold four tiles of the same channel quad; four dY-side waves
#pragma unroll
      for (int j = 0; j < 4; ++j) { bsum[j] += __shfl_xor(bsum[j], 16, 64); bsum[j] += __shfl_xor(bsum[j], 32, 64); }
      if (CB == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { bsum1[j] += __shfl_xor(bsum1[j], 16, 64); bsum1[j] += __shfl_xor(bsum1[j], 32, 64); }
      }
      float* red = reinterpret_cast<float*>(smx);  // every image has been consumed: the consumers are past the last barrier ...
      lds_barrier();                               // ... once they arrive here (all waves make the same two extra barriers)
      if (do_bias && lane < 16) *reinterpret_cast<f32x4*>(red + (wid * 16 + lane) * 4) = bsum;
      if (do_bias && CB == 2 && lane < 16) *reinterpret_cast<f32x4*>(red + ((4 + wid) * 16 + lane) * 4) = bsum1;
      lds_barrier();
#pragma unroll
      for (int hb = 0; hb < CB; ++hb) {
        if (do_bias && wid == 0 && lane < 16 && (hb ? a_col1 : a_col) != OOB) {
          f32x4 v = *reinterpret_cast<const f32x4*>(red + (hb * 64 + lane) * 4);
#pragma unroll
          for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + ((hb * 4 + w) * 16 + lane) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (p.split_stride > 0) p.dbias[(long)bz * p.bias_stride + co0 + hb * XT + quad * 4 + j] = v[j];
            else atomicAdd(&p.dbias[co0 + hb * XT + quad * 4 + j], v[j]);
          }
        }
      }
    }
    };
    if (hw_wid >= NCONS + 4) produce(std::true_type{});
    else produce(std::false_type{});
    lds_barrier();                                 // the consumers' plane exchange (two barriers for all twelve waves)
    lds_barrier();
    return;
  }

  // ================================================================== consumer waves
  // wave w owns the ex = w plane (MODE 1: pixel chunk w) of the whole 64 x 64 tile: 2 x 2 blocks of 32 x 32, so every fragment it
  // reads feeds two block products (12 KB of LDS reads per stage and wave; 32 x 32 x 4 planes per wave read 24 KB)
  // transposed fragment read: 16-lane group g, lane i = 4 q + p of it: row (tile) 8 (g >> 1) + q [+ 4 for the second read],
  // columns 16 (g & 1) + 4 p .. + 3 of a 32-channel block; the lane receives channel 16 (g & 1) + i of those four tiles
  const int g = lane >> 4, gi = lane & 15;
  const int cpl = wid & 3, chalf = wid >> 2;         // plane (MODE 1: pixel chunk) and 64-cout block of this wave
  const int foff = (8 * (g >> 1) + (gi >> 2)) * XROW + 16 * (g & 1) + 4 * (gi & 3) + cpl * TERMS * XK * XROW;
  f32x16 tot[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[i >> 1][i & 1][r] = 0.f;
  auto frag = [&](const unsigned short* base) -> bf16x8 {
    struct { s16x4 lo, hi; } v;
    v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * XROW));
    return __builtin_bit_cast(bf16x8, v);
  };
  for (int s = 0; s < KT; ++s) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned short* Ab = As + (chalf * 2 + (s & 1)) * X_IMG + foff;
    const unsigned short* Bb = Bs + (s & 1) * X_IMG + foff;
    bf16x8 a[2][TERMS], b[2][TERMS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < TERMS; ++k) {
        if (XW_ABL & 16) {
          struct { s16x4 lo, hi; } fk = {{(short)0x3f80, (short)s, (short)0x3f80, (short)h}, {(short)0x3f00, (short)k, (short)0x3f00, (short)lane}};
          a[h][k] = __builtin_bit_cast(bf16x8, fk); b[h][k] = a[h][k];
          continue;
        }
        a[h][k] = frag(Ab + k * XK * XROW + h * 32);
        b[h][k] = frag(Bb + k * XK * XROW + h * 32);
      }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if (XW_ABL & 8) { tot[mb][nb][0] += (float)(__builtin_bit_cast(s16x4, __builtin_bit_cast(u32x2, *(u32x2*)&a[mb][0]))[0] ^ __builtin_bit_cast(s16x4, *(u32x2*)&b[nb][TERMS - 1])[1]); continue; }
        if (FMT) {                                 // three products from C = 0 (small ones first), then the one rounded add
          const f16x8 a0 = __builtin_bit_cast(f16x8, a[mb][0]), a1 = __builtin_bit_cast(f16x8, a[mb][1]);
          const f16x8 b0 = __builtin_bit_cast(f16x8, b[nb][0]), b1 = __builtin_bit_cast(f16x8, b[nb][1]);
          f32x16 c;
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a0), "v"(b1));
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
          tot[mb][nb] += c;
          continue;
        }
        // six products of this stage from C = 0 (small ones first), then ONE rounded f32 add into the running total
        f32x16 c;                                  // C = the inline constant 0 (the builtin with a zero vector first copies sixteen zeros)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a[mb][0]), "v"(b[nb][TERMS - 1]));
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][TERMS - 1], b[nb][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][1], b[nb][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[nb][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][1], b[nb][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[nb][0], c, 0, 0, 0);
        tot[mb][nb] += c;          // (this file is compiled WITHOUT packed-f32 instruction selection: sixteen plain v_add_f32, which run in
      }                            //  the shadow of the next block's MFMAs, with the MFMA -> VALU wait states the compiler inserts)
  }
  if (p.dbias != nullptr && tn == 0 && ey == (MODE == 1 ? 0 : 1)) {  // the producers' bias reduction uses two more barriers
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  }

  // ---- epilogue: the four planes meet in LDS ([plane][cout 64][cin 64] f32 = 64 KB; all twelve waves make these two barriers), then
  // G^T along x.  C/D layout: col = lane & 31 (cin), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (cout)
  float* ex_t = reinterpret_cast<float*>(smx);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the images (and the bias scratch)
  {
    const int lr = lane & 31, lh = lane >> 5;
    const float osgn = ((MODE == 0 && ey == 2) ? -1.f : 1.f) / (s_x * s_dy);   // the X side of the pass ey = 2 was produced negated; FMT 1: powers of two out
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ex_t[((chalf * 4 + cpl) * XT + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * XT + nb * 32 + lr] = osgn * tot[mb][nb][r];
  }
  lds_barrier();
  const int ci = ci0 + lane;                       // wave w: couts 16 w .. 16 w + 15 of the tile, lane = cin (256-byte rows)
  if (ci >= p.Cin) return;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int cr = cpl * 16 + r, co = co0 + chalf * XT + cr;
    if (co >= p.Cout) break;
    const float* eh = ex_t + chalf * 4 * XT * XT;
    const float t0 = eh[(0 * XT + cr) * XT + lane], t1 = eh[(1 * XT + cr) * XT + lane];
    const float t2 = eh[(2 * XT + cr) * XT + lane], t3 = eh[(3 * XT + cr) * XT + lane];
    if (MODE == 1) {
      const float w = (t0 + t1) + (t2 + t3);
      float* dst1 = p.dwp + (long)bz * p.split_stride + (long)co * p.Cin + ci;
      if (p.atomic) atomicAdd(dst1, w); else dst1[0] = w;
      continue;
    }
    const float h = 0.5f * (t1 + t2);
    const float w0 = t0 + h, w1 = 0.5f * (t1 - t2), w2 = h + t3;
    float* dst = p.dwp + (long)bz * p.split_stride + ((long)co * 12 + ey * 3) * p.Cin + ci;
    if (p.atomic) {
      atomicAdd(dst, w0); atomicAdd(dst + p.Cin, w1); atomicAdd(dst + 2 * p.Cin, w2);
    } else {
      dst[0] = w0; dst[p.Cin] = w1; dst[2 * p.Cin] = w2;
    }
  }
}

// amax_x / amax_dy both non-null: the fp16 format (FMT 1) with scales from these device bounds (f32 activations only)
int wgrad_x6_impl(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin, int ldx, int Cout,
                  int lddy, int splits, bool det, bool plan_only, int up, hipStream_t stream, int xbf = 0,
                  const float* amax_x = nullptr, const float* amax_dy = nullptr);

int g_wgrad_cb = -1;      // -1: chosen per launch, 1 / 2: 64-cout blocks per workgroup of the fp16-format kernel (2 where it qualifies)
// 128 couts per workgroup: fp16 format, not the deterministic workspace mode (its split plan is made without the format), > 64 couts
inline int wgrad_cb(int fmt, bool det, int Cout) { return (fmt && !det && Cout > XT && g_wgrad_cb != 1) ? 2 : 1; }

template <int MODE>
int wgrad_x6_launch(const WxP& p, dim3 grid, int xbf, int fmt, int cb, hipStream_t stream) {
  static bool attr_set = false;
  constexpr int smem0 = 4 * XFmt<0>::IMG * (int)sizeof(unsigned short), smem1 = 4 * XFmt<1>::IMG * (int)sizeof(unsigned short);
  constexpr int smem2 = 6 * XFmt<1>::IMG * (int)sizeof(unsigned short);
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x6_kernel<MODE, false, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem0) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x6_kernel<MODE, true, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem0) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x6_kernel<MODE, false, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem1) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x6_kernel<MODE, false, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem2) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  if (fmt && cb == 2) hipLaunchKernelGGL((wgrad_x6_kernel<MODE, false, 1, 2>), grid, dim3(1024), smem2, stream, p);
  else if (fmt) hipLaunchKernelGGL((wgrad_x6_kernel<MODE, false, 1, 1>), grid, dim3(768), smem1, stream, p);
  else if (xbf) hipLaunchKernelGGL((wgrad_x6_kernel<MODE, true, 0, 1>), grid, dim3(768), smem0, stream, p);
  else hipLaunchKernelGGL((wgrad_x6_kernel<MODE, false, 0, 1>), grid, dim3(768), smem0, stream, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
int wgrad_x6_1x1_impl(const float* x, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout, int lddy, int splits,
                      bool det, bool plan_only, hipStream_t stream, int xbf = 0, const float* amax_x = nullptr,
                      const float* amax_dy = nullptr);

int wgrad_x6_impl(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin, int ldx, int Cout,
                  int lddy, int splits, bool det, bool plan_only, int up, hipStream_t stream, int xbf, const float* amax_x,
                  const float* amax_dy) {
  if (!plan_only && (!x || !dy || !dwp)) return ADM_EINVAL;
  const int fmt = amax_x != nullptr && amax_dy != nullptr;
  if ((amax_x != nullptr) != (amax_dy != nullptr) || (fmt && xbf)) return ADM_EINVAL;
  if (B <= 0 || H < 2 || W < 2) return ADM_EINVAL;
  if ((Cin & 31) || (Cout & 31) || (ldx & 3) || (lddy & 3)) return ADM_EINVAL;
  if (!plan_only && (((uintptr_t)x | (uintptr_t)dy) & 15)) return ADM_EINVAL;
  auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  const int lw = ilog2(W), lh = ilog2(H);
  if (lw < 1 || lh < 1) return ADM_EINVAL;                    // power-of-two H, W (>= 2) only
  WxP p;
  p.x = x; p.dy = dy; p.dwp = dwp; p.dbias = dbias; p.amax_x = amax_x; p.amax_dy = amax_dy;
  const long P = (long)B * H * W;
  const long xb = (up ? P / 4 : P) * ldx * (xbf ? 2 : 4), db = P * lddy * 4;
  if (xb >= (1L << 31) - (1L << 22) || db >= (1L << 31) - (1L << 22)) return ADM_EINVAL;   // 32-bit offsets
  p.up = up ? 1 : 0;
  p.Pp = (int)(P / 4); p.H = H; p.W = W; p.lw = lw; p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.lddy = lddy;
  p.xbytes = (int)xb; p.dybytes = (int)db;
  p.tilesN = adm_cdiv(Cin, XT);
  const int cb = wgrad_cb(fmt, det, Cout);
  const long tiles = (long)adm_cdiv(Cout, XT * cb) * p.tilesN * 4;
  const bool prezeroed = splits == ADM_SPLITS_AUTO_PREZEROED;      // zero-at-rest workspace: no memset
  if (splits <= 0) {
    // one workgroup per CU: 256 slots.  The split count whose workgroup total fills whole rounds best, with a mild preference for
    // fewer splits; >= 96 tiles (6 stages) per split
    const long slots = 256;
    const int maxs = (int)std::min<long>((p.Pp + 95) / 96, 64);
    double best = -1.0;
    splits = 1;
    for (int sN = 1; sN <= maxs; ++sN) {
      const long wgs = tiles * sN;
      const long rounds = (wgs + slots - 1) / slots;
      const double fill = (double)wgs / (double)(rounds * slots) - 0.002 * sN;
      if (fill > best + 1e-9) { best = fill; splits = sN; }
    }
  }
  int chunk = ((p.Pp + splits - 1) / splits + XK - 1) / XK * XK;
  splits = (p.Pp + chunk - 1) / chunk;
  if (plan_only) return splits;
  p.chunk = chunk;
  p.split_stride = det ? (long)Cout * 12 * Cin : 0;
  p.bias_stride = det ? Cout : 0;
  p.atomic = splits > 1 && !det;
  if (p.atomic && !prezeroed && hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)Cout * 12 * Cin, stream) != hipSuccess) return ADM_ELAUNCH;
  p.tiles = adm_cdiv(Cout, XT * cb) * p.tilesN;
  return wgrad_x6_launch<0>(p, dim3(p.tiles * 4 * splits), xbf, fmt, cb, stream);
}

int wgrad_x6_1x1_impl(const float* x, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout, int lddy, int splits,
                      bool det, bool plan_only, hipStream_t stream, int xbf, const float* amax_x, const float* amax_dy) {
  if (!plan_only && (!x || !dy || !dwp)) return ADM_EINVAL;
  const int fmt = amax_x != nullptr && amax_dy != nullptr;
  if ((amax_x != nullptr) != (amax_dy != nullptr) || (fmt && xbf)) return ADM_EINVAL;
  if (P <= 0 || (Cin & 31) || (Cout & 31) || (ldx & 3) || (lddy & 3)) return ADM_EINVAL;
  if (!plan_only && (((uintptr_t)x | (uintptr_t)dy) & 15)) return ADM_EINVAL;
  const long xb = P * ldx * (xbf ? 2 : 4), db = P * lddy * 4;
  if (P >= (1L << 30) || xb >= (1L << 31) || db >= (1L << 31)) return ADM_EINVAL;
  WxP p;
  p.x = x; p.dy = dy; p.dwp = dwp; p.dbias = dbias; p.amax_x = amax_x; p.amax_dy = amax_dy;
  p.Pp = (int)P; p.H = 0; p.W = 0; p.lw = 0; p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.lddy = lddy; p.up = 0;
  p.xbytes = (int)xb; p.dybytes = (int)db;
  p.tilesN = adm_cdiv(Cin, XT);
  // (64 pixels per stage, no transforms: the 128-cout form measured SLOWER here -- 7.9 vs 6.4 ms per step over the UNet's 69 launches --
  //  so it is taken only when asked for: adm_wgrad_h3_blocks(2))
  const int cb = g_wgrad_cb == 2 ? wgrad_cb(fmt, det, Cout) : 1;
  const long tiles = (long)adm_cdiv(Cout, XT * cb) * p.tilesN;
  constexpr int STEP = 4 * XK;
  const bool prezeroed = splits == ADM_SPLITS_AUTO_PREZEROED;
  if (splits <= 0) {        // whole rounds of 256 workgroups (one per CU); >= 6 stages (384 pixels) per split
    const long slots = 256;
    const int maxs = (int)std::min<long>((P + 6 * STEP - 1) / (6 * STEP), 128);
    double best = -1.0;
    splits = 1;
    for (int sN = 1; sN <= maxs; ++sN) {
      const long wgs = tiles * sN;
      const long rounds = (wgs + slots - 1) / slots;
      const double fill = (double)wgs / (double)(rounds * slots) - 0.001 * sN;
      if (fill > best + 1e-9) { best = fill; splits = sN; }
    }
  }
  int chunk = (int)(((P + splits - 1) / splits + STEP - 1) / STEP * STEP);
  splits = (int)((P + chunk - 1) / chunk);
  if (plan_only) return splits;
  p.chunk = chunk;
  p.split_stride = det ? (long)Cout * Cin : 0;
  p.bias_stride = det ? Cout : 0;
  p.atomic = splits > 1 && !det;
  if (p.atomic && !prezeroed && hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)Cout * Cin, stream) != hipSuccess) return ADM_ELAUNCH;
  p.tiles = adm_cdiv(Cout, XT * cb) * p.tilesN;
  return wgrad_x6_launch<1>(p, dim3(p.tiles * splits), xbf, fmt, cb, stream);
}

}  // namespace

// Same contract as adm_conv_wgrad_wino2d: dwp2[Cout][4 ey][3 kx][Cin] x-folded planes (adm_unpack_wgrad_wino2d applies G^T along y),
// dbias += column sums of dy; splits = 0 picks the split count; H and W powers of two >= 2.
extern "C" int adm_conv_wgrad_x6(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx,
                                 int Cout, int lddy, int splits, hipStream_t stream) {
  return wgrad_x6_impl(x, dy, dwp2, dbias, B, H, W, Cin, ldx, Cout, lddy, splits, false, false, 0, stream);
}
// Weight gradient of Conv2d(up=True): x is the conv's HALF-resolution input [B][H/2][W/2][ldx]; H x W is dy's grid
extern "C" int adm_conv_wgrad_x6_up(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx,
                                    int Cout, int lddy, int splits, hipStream_t stream) {
  return wgrad_x6_impl(x, dy, dwp2, dbias, B, H, W, Cin, ldx, Cout, lddy, splits, false, false, 1, stream);
}
// Deterministic workspace mode: split z writes its partial planes to ws[z][Cout][12][Cin] and its bias partial to bws[z][Cout]
// (plain stores); splits must be adm_conv_wgrad_x6_plan(...)
extern "C" int adm_conv_wgrad_x6_ws(const float* x, const float* dy, float* ws, float* bws, int B, int H, int W, int Cin, int ldx,
                                    int Cout, int lddy, int splits, int up, hipStream_t stream) {
  return wgrad_x6_impl(x, dy, ws, bws, B, H, W, Cin, ldx, Cout, lddy, splits, true, false, up, stream);
}
extern "C" int adm_conv_wgrad_x6_plan(int B, int H, int W, int Cin, int Cout) {
  return wgrad_x6_impl(nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Cin, Cout, Cout, 0, false, true, 0, nullptr);
}

// Weight gradient of a 1x1 conv / pixel-wise linear map on the same kernel (MODE 1): dwp[Cout][Cin] (+)= sum over P pixels of
// dy[p][co] x[p][ci], dbias += column sums of dy.  splits = 0: chosen by the launcher.  _ws: deterministic mode (split z stores its
// partial at ws[z][Cout][Cin] / bws[z][Cout]); splits must be adm_gemm_wgrad_x6_plan(...).
extern "C" int adm_gemm_wgrad_x6(const float* x, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout, int lddy,
                                 int splits, hipStream_t stream) {
  return wgrad_x6_1x1_impl(x, dy, dwp, dbias, P, Cin, ldx, Cout, lddy, splits, false, false, stream);
}
extern "C" int adm_gemm_wgrad_x6_ws(const float* x, const float* dy, float* ws, float* bws, long P, int Cin, int ldx, int Cout, int lddy,
                                    int splits, hipStream_t stream) {
  return wgrad_x6_1x1_impl(x, dy, ws, bws, P, Cin, ldx, Cout, lddy, splits, true, false, stream);
}
extern "C" int adm_gemm_wgrad_x6_plan(long P, int Cin, int Cout) {
  return wgrad_x6_1x1_impl(nullptr, nullptr, nullptr, nullptr, P, Cin, Cin, Cout, Cout, 0, false, true, nullptr);
}

// The same kernels with x stored as bf16 (the opt-in bf16 mode keeps its GroupNorm outputs in bf16: adm_gn_fwd_bf16out): the products
// are then exact in x and three-term exact in dy.  x16 = [B][H][W][ldx] bf16 ([B][H/2][W/2][ldx] with up); everything else as in
// adm_conv_wgrad_x6 / adm_gemm_wgrad_x6.
extern "C" int adm_conv_wgrad_x6_bf16a(const void* x16, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx,
                                       int Cout, int lddy, int splits, int up, hipStream_t stream) {
  return wgrad_x6_impl(static_cast<const float*>(x16), dy, dwp2, dbias, B, H, W, Cin, ldx, Cout, lddy, splits, false, false, up ? 1 : 0,
                       stream, 1);
}
extern "C" int adm_gemm_wgrad_x6_bf16a(const void* x16, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout,
                                       int lddy, int splits, hipStream_t stream) {
  return wgrad_x6_1x1_impl(static_cast<const float*>(x16), dy, dwp, dbias, P, Cin, ldx, Cout, lddy, splits, false, false, stream, 1);
}

// The same kernels on the fp16 format (FMT 1 above): amax_x / amax_dy = device floats, upper bounds of |x| and |dy| (the producers of
// the two tensors wrote them: adm_gn_fwd_amax, adm_gn_bwd_add_amax, adm_add3).  A bound that is too small overflows the fp16 terms
// (inf / nan in dW: loud, not silent).  det != 0: the deterministic workspace mode of adm_conv_wgrad_x6_ws / adm_gemm_wgrad_x6_ws
// (dwp = ws, dbias = bws, splits from the _plan call); else splits as for adm_conv_wgrad_x6.
extern "C" int adm_conv_wgrad_x6_h3(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx,
                                    int Cout, int lddy, int splits, int up, int det, const float* amax_x, const float* amax_dy,
                                    hipStream_t stream) {
  if (!amax_x || !amax_dy) return ADM_EINVAL;
  return wgrad_x6_impl(x, dy, dwp2, dbias, B, H, W, Cin, ldx, Cout, lddy, splits, det != 0, false, up ? 1 : 0, stream, 0, amax_x, amax_dy);
}
extern "C" int adm_gemm_wgrad_x6_h3(const float* x, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout,
                                    int lddy, int splits, int det, const float* amax_x, const float* amax_dy, hipStream_t stream) {
  if (!amax_x || !amax_dy) return ADM_EINVAL;
  return wgrad_x6_1x1_impl(x, dy, dwp, dbias, P, Cin, ldx, Cout, lddy, splits, det != 0, false, stream, 0, amax_x, amax_dy);
}

// form of the fp16-format weight-gradient kernels: -1 (default) 128 couts per workgroup where the launch qualifies, 1 always 64, 2 as -1;
// returns the old value
extern "C" int adm_wgrad_h3_blocks(int v) { const int old = g_wgrad_cb; g_wgrad_cb = v; return old; }
