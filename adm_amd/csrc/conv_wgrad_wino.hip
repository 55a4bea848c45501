// Weight gradient of the 3x3 stride-1 convs through the TRANSPOSED 1-D Winograd form F(3,2) along x (fp32 MFMA).
//
//   dW[co][ky][kx][ci] = sum_{b,y,x} dY[b,y,x][co] * X[b, y+ky-1, x+kx-1][ci]
// Per output-pixel pair (x0 = 2xp, x0+1) and filter row ky, with e = dY[x0], dY[x0+1] and d_i = X[.., x0-1+i] (i = 0..3), the
// three taps are W_kx = e0 d_kx + e1 d_{kx+1}: 6 multiplies.  Transposing F(2,3) (conv_wino.hip) gives 4:
//     a = A e   = (e0, e0+e1, e0-e1, -e1)            b = B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)          (only +-1)
//     m_xi = sum_pairs a_xi * b_xi                     (four GEMMs, reduction over pixel pairs)
//     W_0 = m0 + (m1+m2)/2     W_1 = (m1-m2)/2     W_2 = (m1+m2)/2 + m3                                 (epilogue)
// i.e. 1.5x fewer MFMA flops than conv_wgrad.hip; constants +-1 and 1/2 only (fp32 error at the direct kernel's level).
//
// Structure = conv_wgrad.hip's: both operands arrive pixel-major and are TRANSPOSED on their way into LDS (channel rows,
// pair columns) after the 3 + 4 float4 add/sub of the transforms; grid = (Cout/64 x Cin/64 tiles, 3 filter rows, splits
// of the pair range), fp32 atomics when split.  One workgroup = 64 couts x 64 cins x 4 xi (4 accumulator tiles per wave),
// K-step 16 pairs, 32 KB LDS (both operands register-staged and single-buffered with two barriers per stage -> three
// workgroups per CU instead of two: +7 %; rows of 16 floats, slots XOR-swizzled as in conv_wino.hip; the four quad groups of a wave
// write DIFFERENT channel residues in each store round so that the transposing ds_write_b32 stays conflict-free).
// The bias gradient rides along: sum(e0 + e1) is the xi = 1 component of A e.
// Power-of-two H and W only (every DDM shape); anything else stays on conv_wgrad.hip.
//
// MODE 2 = the same nested along y: the transposed 2-D form F(3x3, 2x2).  Per 2x2 tile of dY and the 4x4 patch of X around it,
//     a = A e A^T (4x4 from 2x2)     b = B^T d B (4x4)     m[ey][ex] = sum_tiles a[ey][ex] * b[ey][ex]     dW = G^T m G
// i.e. 16 products per (tile, co, ci) instead of the 1-D form's 24 (the direct kernel: 36): 1.5x less MFMA work again.  The y
// index takes the place of the filter row in the grid (gridDim.y = 4 instead of 3): workgroup `ey` reduces over TILES the
// y-combined rows  a_y = (e_r0, e_r0 + e_r1, e_r0 - e_r1, -e_r1)[ey]  of dY and  (r0 - r2, r1 + r2, r2 - r1, r1 - r3)[ey]  of X
// through the unchanged x pipeline (A e / B^T d, four ex GEMMs, G^T along x in the epilogue) and writes wx[co][ey][kx][ci];
// the y half of G^T, dW[ky] = sum_ey Gt[ky][ey] wx[ey], is applied by the unpack launch that follows anyway
// (adm_unpack_wgrad_wino2d).  Same accumulators, LDS and occupancy as the 1-D form; 11 instead of 6 pixel loads per stage for
// 2/3 of the stages.
#include <algorithm>
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

struct WwP {
  const float* x; const float* dy; float* dwp; float* dbias;
  int Pp, H, W, lw, lh, Cin, ldx, Cout, lddy, tilesN, chunk, atomic, xbytes, dybytes;   // Pp = pixel pairs (MODE 2: 2x2 tiles), chunk likewise
  int taps;                                                                              // rows per cout of dwp: 9 (1-D) or 12 = 4 ey x 3 kx (MODE 2)
  long split_stride, bias_stride;      // > 0: deterministic mode, partials of split z at dwp + z * split_stride (plain stores)
};

constexpr int WT = 64, WKK = 16;       // 64 x 64 channel tile, 16 pairs per stage

template <int MODE>      // 0: 1-D F(3,2); 1: the same with the conv's fused nearest x2 (x is half resolution); 2: 2-D F(3x3, 2x2)
__global__ __launch_bounds__(256) void wgrad_wino_kernel(WwP p) {
  constexpr bool UP = MODE == 1, TWOD = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                               // [4][WT][WKK]   a_xi, rows = cout  (single-buffered: see the loop)
  float* Bs = smem + 4 * WT * WKK;                // [4][WT][WKK]   b_xi, rows = cin
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int tn = blockIdx.x % p.tilesN, tm = blockIdx.x / p.tilesN;
  const int co0 = tm * WT, ci0 = tn * WT;
  const int ky = blockIdx.y;
  const int pbeg = blockIdx.z * p.chunk;
  const int pend = min(p.Pp, pbeg + p.chunk);
  if (pbeg >= pend) return;
  const int KT = (pend - pbeg + WKK - 1) / WKK;

  // loader: thread = (pair lk of the stage, channel quad): 16 pairs x 16 quads (64 channels) per operand
  const int lk = lane & 15, lq = lane >> 4;
  const int quad = wid * 4 + lq;                  // 0..15
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dybytes, 0x00020000);
  // X is addressed relative to pixel (2p) + (ky-1) W - 1, the position of d0: fold that shift into the descriptor base
  const int shift = ((ky - 1) * p.W - 1) * p.ldx;
  const __amdgpu_buffer_rsrc_t rs_xt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + shift, 0, max(0, p.xbytes - shift * 4), 0x00020000);
  const unsigned a_col = (co0 + quad * 4 < p.Cout) ? (unsigned)(co0 + quad * 4) * 4u : OOB;
  const unsigned b_col = (ci0 + quad * 4 < p.Cin) ? (unsigned)(ci0 + quad * 4) * 4u : OOB;
  const unsigned a_voff = (a_col != OOB) ? (unsigned)(2 * lk * p.lddy) * 4u + a_col : OOB;     // e0; e1 is one pixel further
  const unsigned b_voff = (b_col != OOB) ? (unsigned)(2 * lk * p.ldx) * 4u + b_col : OOB;      // d0; d_i is i pixels further
  const int Wh = p.W >> 1;

  f32x4 e[2], d[4];
  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  auto load_stage = [&](int s) {
    const int pb = pbeg + s * WKK;
    const int pr = pb + lk;                                         // this thread's pair (MODE 2: its 2x2 tile)
    const bool pv = pr < pend;
    if (TWOD) {
      // tile (b, ty, xp): top-left output pixel (b, 2 ty, 2 xp) = ((pr >> lwh) << (lw + 1)) + 2 xp    (H, W powers of two)
      const int ey = ky;                                            // blockIdx.y = y index of the transform
      const int lwh = p.lw - 1;
      const int xp = pr & (Wh - 1), ty = (pr >> lwh) & ((p.H >> 1) - 1);
      const unsigned pix = ((unsigned)(pr >> lwh) << (p.lw + 1)) + 2u * (unsigned)xp;
      // dY rows 2ty (r0) and 2ty + 1 (r1): a_y = (r0, r0 + r1, r0 - r1, -r1)[ey]
      const unsigned ya = (pv && a_col != OOB) ? pix * (unsigned)p.lddy * 4u + a_col : OOB;
      const unsigned ystep = (unsigned)p.lddy * 4u, yrow = (unsigned)p.W * ystep;
      // rows that a pass does not use are 'loaded' out of range (zeros, no memory traffic): ey = 0 uses r0 only, ey = 3 r1 only
      const unsigned y0 = (ey != 3) ? ya : OOB, y1 = (ey != 0 && ya != OOB) ? ya + yrow : OOB;
      f32x4 t0[2], t1[2];
      t0[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)y0, 0, 0));
      t0[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(y0 != OOB ? y0 + ystep : OOB), 0, 0));
      t1[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)y1, 0, 0));
      t1[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(y1 != OOB ? y1 + ystep : OOB), 0, 0));
      // X rows 2ty - 1 + i: (r0 - r2, r1 + r2, r2 - r1, r1 - r3)[ey], columns 2xp - 1 .. 2xp + 2
      const int iA = (ey == 0) ? 0 : 1, iB = (ey == 3) ? 3 : 2;
      const bool vA = pv && b_col != OOB && (iA != 0 || ty > 0), vB = pv && b_col != OOB && (iB != 3 || ty < (p.H >> 1) - 1);
      const unsigned xstep = (unsigned)p.ldx * 4u;
      const unsigned xa = pix * xstep + b_col + (unsigned)((iA - 1) * p.W - 1) * xstep;      // row iA, column 2xp - 1 (may wrap: masked)
      const unsigned xb = pix * xstep + b_col + (unsigned)((iB - 1) * p.W - 1) * xstep;
      f32x4 u0[4], u1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cv = (j != 0 || xp > 0) && (j != 3 || xp < Wh - 1);
        u0[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x0, (int)((vA && cv) ? xa + j * xstep : OOB), 0, 0));
        u1[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x0, (int)((vB && cv) ? xb + j * xstep : OOB), 0, 0));
      }
      if (ey <= 1) { e[0] = t0[0] + t1[0]; e[1] = t0[1] + t1[1]; }      // (r0, r0 + r1, r0 - r1, -r1)[ey] with the unused row = 0
      else { e[0] = t0[0] - t1[0]; e[1] = t0[1] - t1[1]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = (ey == 1) ? u0[j] + u1[j] : (ey == 2) ? u1[j] - u0[j] : u0[j] - u1[j];
      return;
    }
    const int xp = pr & (Wh - 1), y = (pr >> (p.lw - 1)) & (p.H - 1);
    const bool rv = pv && (unsigned)(y + ky - 1) < (unsigned)p.H;
    const int a_soff = 2 * pb * p.lddy * 4, b_soff = 2 * pb * p.ldx * 4;
    const unsigned av = pv ? a_voff : OOB;
    e[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)av, a_soff, 0));
    e[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(av + (pv ? p.lddy * 4 : 0)), a_soff, 0));
    const unsigned step = (unsigned)p.ldx * 4u;
    if (UP) {
      // fused nearest x2 (Conv2d(up=True)): X is [B][H/2][W/2]; up-sampled row y+ky-1 reads input row (y+ky-1)>>1 and the
      // columns 2xp-1..2xp+2 read xp-1, xp, xp, xp+1
      const int bimg = pr >> (p.lw - 1 + p.lh);
      const int irow = (y + ky - 1) >> 1;
      const unsigned base = (rv && b_col != OOB)
          ? (unsigned)((((long)bimg * (p.H >> 1) + irow) * Wh + xp) * p.ldx) * 4u + b_col : OOB;
      const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
      d[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((base != OOB && xp > 0) ? base - step : OOB), 0, 0));
      d[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)base, 0, 0));
      d[2] = d[1];
      d[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((base != OOB && xp < Wh - 1) ? base + step : OOB), 0, 0));
      return;
    }
    const unsigned bv = rv ? b_voff : OOB;
    d[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)((rv && xp > 0) ? bv : OOB), b_soff, 0));
    d[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)(rv ? bv + step : OOB), b_soff, 0));
    d[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)(rv ? bv + 2 * step : OOB), b_soff, 0));
    d[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)((rv && xp < Wh - 1) ? bv + 3 * step : OOB), b_soff, 0));
  };
  // the bias gradient = sum of every dY pixel = the ex = 1 component (e0 + e1 along x) of the filter row 0 workgroups; in MODE 2
  // of the ey = 1 workgroups, whose y combination is r0 + r1
  const bool do_bias = p.dbias != nullptr && tn == 0 && ky == (TWOD ? 1 : 0);
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  // transposing store: element (channel c = quad*4 + jj, pair lk) of plane xi -> row c, swizzled column
  //   float offset = c*16 + (((lk >> 2) ^ ((c >> 2) & 3)) << 2) + (lk & 3),  (c >> 2) & 3 = quad & 3
  const int colw = ((((lk >> 2) ^ (quad & 3)) << 2) + (lk & 3));
  auto store_stage = [&](int buf) {
    f32x4 a[4], b[4];
    a[0] = e[0]; a[1] = e[0] + e[1]; a[2] = e[0] - e[1]; a[3] = -e[1];
    b[0] = d[0] - d[2]; b[1] = d[1] + d[2]; b[2] = d[2] - d[1]; b[3] = d[1] - d[3];
    if (do_bias) bsum += a[1];
    float* la = As + quad * 4 * WKK + colw;
    float* lb = Bs + quad * 4 * WKK + colw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jj = (j + lq) & 3;                // rotate the channel residue across the wave's four quad groups
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        la[xi * WT * WKK + jj * WKK] = a[xi][jj];
        lb[xi * WT * WKK + jj * WKK] = b[xi][jj];
      }
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
  int foff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) foff[g] = lr * WKK + (((2 * g + lh) ^ ((lr >> 2) & 3)) << 2);

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < KT; ++s) {
    if (s + 1 < KT) load_stage(s + 1);
    const float* Ab = As + wm * 32 * WKK;
    const float* Bb = Bs + wn * 32 * WKK;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        a[xi] = *reinterpret_cast<const f32x4*>(Ab + xi * WT * WKK + foff[g]);
        b[xi] = *reinterpret_cast<const f32x4*>(Bb + xi * WT * WKK + foff[g]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[xi][k], b[xi][k], acc[xi], 0, 0, 0);
    }
    // both operands are register-staged, so LDS is single-buffered (32 KB per workgroup -> three workgroups per CU, the
    // register limit, instead of two): one barrier after this stage's reads, one after the next stage's stores
    __syncthreads();
    if (s + 1 < KT) store_stage(0);
    __syncthreads();
  }

  if (do_bias) {       // lanes lk = 0..15 of a quad group hold different pairs of the same channel quad
    f32x4 v = bsum;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], o, 64);
    if (lk == 0 && a_col != OOB) {
      if (p.split_stride > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p.dbias[(long)blockIdx.z * p.bias_stride + co0 + quad * 4 + j] = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&p.dbias[co0 + quad * 4 + j], v[j]);
      }
    }
  }
  // ---- epilogue: G^T m.  rows = cout, cols = cin
  const int ci = ci0 + wn * 32 + lr;
  if (ci >= p.Cin) return;
  const int cb = co0 + wm * 32 + 4 * lh;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = cb + (r & 3) + 8 * (r >> 2);
    if (co >= p.Cout) continue;
    const float h = 0.5f * (acc[1][r] + acc[2][r]);
    const float w0 = acc[0][r] + h, w1 = 0.5f * (acc[1][r] - acc[2][r]), w2 = h + acc[3][r];
    float* dst = p.dwp + (long)blockIdx.z * p.split_stride + ((long)co * p.taps + ky * 3) * p.Cin + ci;
    if (p.atomic) {
      atomicAdd(dst, w0); atomicAdd(dst + p.Cin, w1); atomicAdd(dst + 2 * p.Cin, w2);
    } else {
      dst[0] = w0; dst[p.Cin] = w1; dst[2 * p.Cin] = w2;
    }
  }
}

}  // namespace

namespace {
int wgrad_wino_impl(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin, int ldx, int Cout,
                    int lddy, int splits, int mode, bool det, bool plan_only, hipStream_t stream) {
  const int up = mode == 1, twod = mode == 2;
  if (mode < 0 || mode > 2 || (twod && H < 2)) return ADM_EINVAL;
  if (!plan_only && (!x || !dy || !dwp)) return ADM_EINVAL;
  if (B <= 0 || H <= 0 || W < 2) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1))) return ADM_EINVAL;
  if ((Cin & 31) || (Cout & 31) || (ldx & 3) || (lddy & 3)) return ADM_EINVAL;
  if (!plan_only && (((uintptr_t)x | (uintptr_t)dy) & 15)) return ADM_EINVAL;
  auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  const int lw = ilog2(W), lh = ilog2(H);
  if (lw < 1 || lh < 0) return ADM_EINVAL;                    // power-of-two H, W (W >= 2) only
  WwP p;
  p.x = x; p.dy = dy; p.dwp = dwp; p.dbias = dbias;
  const long P = (long)B * H * W;
  const long xb = (up ? P / 4 : P) * ldx * 4, db = P * lddy * 4;
  if (xb >= (1L << 31) - (1L << 22) || db >= (1L << 31) - (1L << 22)) return ADM_EINVAL;   // 32-bit offsets, with room for the tap shift
  p.Pp = (int)(twod ? P / 4 : P / 2); p.taps = twod ? 12 : 9; p.H = H; p.W = W; p.lw = lw; p.lh = lh; p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.lddy = lddy;
  p.xbytes = (int)xb; p.dybytes = (int)db;
  p.tilesN = adm_cdiv(Cin, WT);
  const int gy = twod ? 4 : 3;
  const long tiles = (long)adm_cdiv(Cout, WT) * p.tilesN * gy;
  const bool prezeroed = splits == ADM_SPLITS_AUTO_PREZEROED;      // zero-at-rest workspace: no memset
  if (splits <= 0) {
    // 768 resident slots (3 workgroups per CU).  Pick the split count whose workgroup total fills whole rounds best
    // (tiles * s close below a multiple of 768), with a mild preference for fewer splits (atomics, shorter K loops);
    // >= 64 pairs per split.  E.g. 384 x 384: 108 tiles -> s = 4 fills 84 % of one round, s = 14 fills 98 % of three.
    const long slots = 768;                      // 3 workgroups per CU (32 KB of LDS, 151 registers)
    const int maxs = (int)std::min<long>((p.Pp + 63) / 64, 32);
    double best = -1.0;
    splits = 1;
    for (int sN = 1; sN <= maxs; ++sN) {
      const long wgs = tiles * sN;
      const long rounds = (wgs + slots - 1) / slots;
      const double fill = (double)wgs / (double)(rounds * slots) - 0.002 * sN;
      if (fill > best + 1e-9) { best = fill; splits = sN; }
    }
  }
  int chunk = ((p.Pp + splits - 1) / splits + WKK - 1) / WKK * WKK;
  splits = (p.Pp + chunk - 1) / chunk;
  if (plan_only) return splits;
  p.chunk = chunk;
  p.split_stride = det ? (long)Cout * p.taps * Cin : 0;
  p.bias_stride = det ? Cout : 0;
  p.atomic = splits > 1 && !det;
  if (p.atomic && !prezeroed && hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)Cout * p.taps * Cin, stream) != hipSuccess) return ADM_ELAUNCH;
  constexpr int smem = 4 * (WT + WT) * WKK * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wino_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wino_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wino_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  dim3 grid(adm_cdiv(Cout, WT) * p.tilesN, gy, splits);
  if (twod) hipLaunchKernelGGL(wgrad_wino_kernel<2>, grid, dim3(256), smem, stream, p);
  else if (up) hipLaunchKernelGGL(wgrad_wino_kernel<1>, grid, dim3(256), smem, stream, p);
  else hipLaunchKernelGGL(wgrad_wino_kernel<0>, grid, dim3(256), smem, stream, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
}  // namespace

extern "C" int adm_conv_wgrad_wino(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin,
                                   int ldx, int Cout, int lddy, int splits, hipStream_t stream) {
  return wgrad_wino_impl(x, dy, dwp, dbias, B, H, W, Cin, ldx, Cout, lddy, splits, 0, false, false, stream);
}

// Weight gradient of Conv2d(up=True): x is the conv's HALF-resolution input [B][H/2][W/2][ldx]; H x W is dy's grid.
extern "C" int adm_conv_wgrad_wino_up(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin,
                                      int ldx, int Cout, int lddy, int splits, hipStream_t stream) {
  return wgrad_wino_impl(x, dy, dwp, dbias, B, H, W, Cin, ldx, Cout, lddy, splits, 1, false, false, stream);
}

// Used by adm_conv_wgrad_plan / adm_conv_wgrad_ws (conv_wgrad.hip): the split count the launcher picks, and the deterministic
// workspace mode (split z writes its partial tile to ws[z][Cout][9][Cin] and its bias partial to bws[z][Cout]).
int adm_wgrad_wino_plan(int B, int H, int W, int Cin, int Cout, int mode) {
  return wgrad_wino_impl(nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Cin, Cout, Cout, 0, mode, false, true, nullptr);
}
int adm_wgrad_wino_ws(const float* x, const float* dy, float* ws, float* bws, int B, int H, int W, int Cin, int ldx, int Cout,
                      int lddy, int splits, int mode, hipStream_t stream) {
  return wgrad_wino_impl(x, dy, ws, bws, B, H, W, Cin, ldx, Cout, lddy, splits, mode, true, false, stream);
}

// 2-D F(3x3, 2x2) weight gradient (MODE 2 above): dwp2[Cout][4 ey][3 kx][Cin] holds the x-folded planes; adm_unpack_wgrad_wino2d
// applies G^T along y.  H and W powers of two (H >= 2); dbias as adm_conv_wgrad_wino.
extern "C" int adm_conv_wgrad_wino2d(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin,
                                     int ldx, int Cout, int lddy, int splits, hipStream_t stream) {
  return wgrad_wino_impl(x, dy, dwp2, dbias, B, H, W, Cin, ldx, Cout, lddy, splits, 2, false, false, stream);
}
