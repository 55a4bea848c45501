// Weight gradient of the 3x3 / 1x1 / Linear layers on the fp32-input MFMA (gfx950).
//
//   dWp[co][tap][ci] = sum_p dY[p][co] * X[pix(p) + tap][ci]
//
// GEMM view: M' = Cout, N' = Cin, K' = pixels.  Both operands arrive pixel-major (NHWC), i.e. with the
// REDUCTION index outermost, while the MFMA wants each lane to supply consecutive k of one row.  The tiles
// are therefore TRANSPOSED on their way into LDS ([channel][32 pixels + 4 pad]) so that the compute loop is
// the same VALU-free loop as the forward kernel (conv_igemm.hip): per 32-pixel step only conflict-free
// ds_read_b128 fragment reads with immediate offsets and 64 MFMAs per wave -- on gfx950 the fp32 MFMA shares
// the SIMD's fp32 lanes, so any vector-ALU instruction left in the loop is MFMA time lost.
//
//   * global -> registers: raw buffer loads, lane = (pixel = lane & 15, channel quad = lane >> 4): 16 pixels
//     x 64 B per wave instruction; the descriptor's range check returns zeros for out-of-image taps,
//     channel-edge tiles and pixels past the end (no selects);
//   * registers -> LDS: four ds_write_b32 per float4, to rows c..c+3 at column `pixel`; with the 36-float row
//     stride a 32-lane group (2 quads x 16 pixels) covers all 32 banks exactly once;
//   * the reduction over pixels is split across gridDim.z; partial tiles are combined with fp32 global atomics
//     shaped as 128-byte row segments, or plain stores when splits == 1.
// Autograd counterpart of F.conv2d's weight gradient in the reference
// (/root/reference/unet/uncond_unet.py:98-110 under loss.backward()).
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

struct WgradP {
  const float* x; const float* dy; float* dwp;
  int P, H, W, Hin, Win, Cin, ldx, Cout, lddy, ks, up, tilesN, chunk, atomic, lw, lh, xbytes, dybytes;
  int stride, pad_lo;      // generic path: tap (ky, kx) of output (oy, ox) reads input (oy*stride + ky - pad_lo, ox*stride + kx - pad_lo)
  float* dbias;      // optional: dbias[co] += sum_p dY[p][co] (the conv's bias gradient), by the tap-0 / ci-tile-0 workgroups
  // deterministic mode (split_stride > 0): split z stores its partial tile at dwp + z * split_stride and its bias partial at
  // dbias + z * bias_stride with plain stores; adm_unpack_wgrad_splits sums the splits in a fixed order
  long split_stride, bias_stride;
};

constexpr int WLDS = 36;   // floats per LDS row: 32 pixels + 4 pad

template <int TM, int TN, bool FAST>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradP p) {
  constexpr int WM = 2, WN = 2;
  constexpr int MT = TM / (WM * 32), NT = TN / (WN * 32);
  constexpr int AI = TM / 32, BI = TN / 32;            // float4 per thread per stage (32 px * T/4 quads / 256)
  __shared__ __attribute__((aligned(16))) float As[2][TM][WLDS];
  __shared__ __attribute__((aligned(16))) float Bs[2][TN][WLDS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int tn = blockIdx.x % p.tilesN, tm = blockIdx.x / p.tilesN;
  const int co0 = tm * TM, ci0 = tn * TN;
  const int tap = blockIdx.y;
  const int pad = p.ks >> 1;
  // offsets relative to the output pixel scaled by the stride: (ky - pad_lo, kx - pad_lo); = tap - ks/2 for the 'same' convs
  const int dy_ = tap / p.ks - p.pad_lo, dx_ = tap % p.ks - p.pad_lo;
  (void)pad;
  const int pbeg = blockIdx.z * p.chunk;
  const int pend = min(p.P, pbeg + p.chunk);
  if (pbeg >= pend) return;
  const int KT = (pend - pbeg + 31) >> 5;

  // wave instruction q = wid*I + i covers pixels (q&1)*16 + (lane&15) and channel quads (q>>1)*4 + (lane>>4)
  const int lk = lane & 15, lq = lane >> 4;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const int shift = (dy_ * p.W + dx_) * p.ldx;                       // floats; may be negative (FAST path only)
  const __amdgpu_buffer_rsrc_t rs_xt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + shift, 0, max(0, p.xbytes - shift * 4), 0x00020000);
  // (a tap shift larger than the whole tensor -- 1 x 2 images -- must not wrap num_records around: 0 records = every
  //  access out of range = zeros, which is what such a tap contributes)

  int a_pix[AI], a_row[AI], b_pix[BI], b_row[BI];     // stage-local pixel, tile-local first channel of the quad
  unsigned a_voff[AI], b_voff[BI], b_colb[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int q = wid * AI + i;
    a_pix[i] = (q & 1) * 16 + lk;
    a_row[i] = ((q >> 1) * 4 + lq) * 4;
    a_voff[i] = (co0 + a_row[i] < p.Cout) ? (unsigned)(a_pix[i] * p.lddy + co0 + a_row[i]) * 4u : OOB;
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int q = wid * BI + i;
    b_pix[i] = (q & 1) * 16 + lk;
    b_row[i] = ((q >> 1) * 4 + lq) * 4;
    b_colb[i] = (ci0 + b_row[i] < p.Cin) ? (unsigned)(ci0 + b_row[i]) * 4u : OOB;
    const bool xv = (unsigned)((b_pix[i] & (p.W - 1)) + dx_) < (unsigned)p.W;      // FAST: column test is per-thread constant
    b_voff[i] = (xv && b_colb[i] != OOB) ? (unsigned)(b_pix[i] * p.ldx) * 4u + b_colb[i] : OOB;
  }

  f32x4 ra[AI], rb[BI];
  // bias gradient = column sums of dY: the workgroups of the first Cin tile and first tap already stream every dY
  // element of their Cout rows through registers, so they keep running sums (16 v_add per 32-pixel stage, on 1 in
  // tilesN * taps workgroups) instead of a separate pass over dY
  const bool do_bias = p.dbias != nullptr && tn == 0 && tap == 0;
  f32x4 bsum[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_stage = [&](int s) {
    const int pb = pbeg + (s << 5);
    const bool full = pb + 32 <= pend;          // wave-uniform: only a ragged last stage needs per-row tests
    const int a_soff = pb * p.lddy * 4;
    if (full) {
#pragma unroll
      for (int i = 0; i < AI; ++i)
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)a_voff[i], a_soff, 0));
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        unsigned vo = (pb + a_pix[i] < pend) ? a_voff[i] : OOB;
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)vo, a_soff, 0));
      }
    }
    if constexpr (FAST) {
      // power-of-two image, W <= 32, no up-sampling: the byte offset inside the stage is a per-thread constant,
      // the stage's pixel base rides in the scalar offset and the tap shift is folded into the descriptor's
      // base -> per load only the image-row test (nothing at all for the centre-row taps).
      const int b_soff = pb * p.ldx * 4;
      const int U = pb >> p.lw;
      if (dy_ == 0 && full) {
#pragma unroll
        for (int i = 0; i < BI; ++i)
          rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)b_voff[i], b_soff, 0));
      } else {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
          const int iy = ((U + (b_pix[i] >> p.lw)) & (p.H - 1)) + dy_;
          const bool v = (unsigned)iy < (unsigned)p.H && (full || pb + b_pix[i] < pend);
          rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)(v ? b_voff[i] : OOB), b_soff, 0));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int pp = pb + b_pix[i];
        int ox = pp % p.W;
        int t = pp / p.W;
        int oy = t % p.H;
        int b = t / p.H;
        int iy = oy * p.stride + dy_, ix = ox * p.stride + dx_;
        // bounds in the grid the taps walk on: the up-sampled grid for `up`, else the (possibly larger, strided) input image
        const int HB = p.up ? p.H : p.Hin, WB = p.up ? p.W : p.Win;
        bool v = pp < pend && (unsigned)iy < (unsigned)HB && (unsigned)ix < (unsigned)WB && b_colb[i] != OOB;
        if (p.up) { iy >>= 1; ix >>= 1; }
        unsigned voff = v ? (unsigned)(((b * p.Hin + iy) * p.Win + ix) * p.ldx) * 4u + b_colb[i] : OOB;
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff, 0, 0));
      }
    }
  };
  auto store_stage = [&](int buf) {        // transpose: channel rows, pixel columns
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < AI; ++i) bsum[i] += ra[i];
    }
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) As[buf][a_row[i] + j][a_pix[i]] = ra[i][j];
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[buf][b_row[i] + j][b_pix[i]] = rb[i][j];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < KT; ++s) {
    const int buf = s & 1;
    if (s + 1 < KT) load_stage(s + 1);
    const float* Ab = &As[buf][wm * MT * 32 + lr][lh * 4];
    const float* Bb = &Bs[buf][wn * NT * 32 + lr][lh * 4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {          // lanes 0-31 take pixels 8g..8g+3, lanes 32-63 pixels 8g+4..8g+7
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * WLDS + g * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * WLDS + g * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < KT) store_stage(buf ^ 1);
    __syncthreads();
  }

  if (do_bias) {
    // instructions i and i^1 of a wave cover the two 16-pixel halves of the same channel quad; lanes lk = 0..15 of a
    // quad group hold different pixels
#pragma unroll
    for (int i = 0; i < AI; i += 2) {
      f32x4 v = bsum[i] + bsum[i + 1];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], o, 64);
      if (lk == 0 && co0 + a_row[i] < p.Cout) {
        if (p.split_stride > 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) p.dbias[(long)blockIdx.z * p.bias_stride + co0 + a_row[i] + j] = v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicAdd(&p.dbias[co0 + a_row[i] + j], v[j]);
        }
      }
    }
  }
  const int taps = p.ks * p.ks;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ci = ci0 + (wn * NT + j) * 32 + lr;
    if (ci >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int cb = co0 + (wm * MT + i) * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cb + (r & 3) + 8 * (r >> 2);
        if (co < p.Cout) {
          float* dst = p.dwp + (long)blockIdx.z * p.split_stride + ((long)co * taps + tap) * p.Cin + ci;
          if (p.atomic) atomicAdd(dst, acc[i][j][r]);
          else *dst = acc[i][j][r];
        }
      }
    }
  }
}

template <int TM, int TN>
int launch_wgrad(WgradP p, int splits, hipStream_t st) {
  p.tilesN = adm_cdiv(p.Cin, TN);
  dim3 grid(adm_cdiv(p.Cout, TM) * p.tilesN, p.ks * p.ks, splits);
  const bool fast = p.lw >= 0 && p.W <= 32 && !p.up && p.stride == 1;       // see load_stage
  if (fast) hipLaunchKernelGGL((wgrad_f32_kernel<TM, TN, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((wgrad_f32_kernel<TM, TN, false>), grid, dim3(256), 0, st, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

}  // namespace

namespace {
// plan_only: return the split count the launcher would pick.  split_stride > 0: deterministic workspace mode.
int wgrad_impl(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin, int ldx, int Cout,
               int lddy, int ks, int up, int splits, long split_stride, long bias_stride, bool plan_only, hipStream_t stream,
               int stride = 1, int pad_lo = -1, int Hin = 0, int Win = 0) {
  if (!plan_only && (!x || !dy || !dwp)) return ADM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0) return ADM_EINVAL;
  if ((Cin & 31) || (Cout & 31) || (ldx & 3) || (lddy & 3) || ks < 1 || ks > 7) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1) || ks > 3 || stride != 1)) return ADM_EINVAL;
  if (pad_lo < 0) pad_lo = ks >> 1;
  if (stride < 1 || stride > 4) return ADM_EINVAL;
  if (!plan_only && (((uintptr_t)x | (uintptr_t)dy) & 15)) return ADM_EINVAL;
  WgradP p;
  p.x = x; p.dy = dy; p.dwp = dwp; p.dbias = dbias;
  p.P = B * H * W; p.H = H; p.W = W; p.Hin = up ? H / 2 : (Hin > 0 ? Hin : H); p.Win = up ? W / 2 : (Win > 0 ? Win : W);
  p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.lddy = lddy; p.ks = ks; p.up = up; p.tilesN = 0;
  p.split_stride = split_stride; p.bias_stride = bias_stride; p.stride = stride; p.pad_lo = pad_lo;
  if (stride == 1 && (p.Hin != (up ? H / 2 : H) || p.Win != (up ? W / 2 : W))) return ADM_EINVAL;
  const long xb = (long)B * p.Hin * p.Win * ldx * 4, db = (long)p.P * lddy * 4;
  if (xb >= (1L << 31) || db >= (1L << 31)) return ADM_EINVAL;      // 32-bit buffer offsets
  p.xbytes = (int)xb; p.dybytes = (int)db;
  auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  p.lw = ilog2(W); p.lh = ilog2(H);
  if (p.lw < 0 || p.lh < 0) p.lw = p.lh = -1;
  // Cout = 192 / 576 (the 32x32-resolution blocks): a 192 x 64 tile (3 MFMA rows per wave, 73.7 KB of LDS like the
  // 128 x 128 one) instead of 64-row tiles, which halve the operand reuse
  const int TM = (Cout % 128 == 0) ? 128 : (Cout % 192 == 0) ? 192 : 64;
  const int TN = (TM == 192) ? 64 : (Cin % 128 == 0) ? 128 : 64;
  const bool prezeroed = splits == ADM_SPLITS_AUTO_PREZEROED;      // zero-at-rest workspace: no memset
  if (splits <= 0) {
    // Fill the resident slots (256 CUs x 2 workgroups, 4 for the 64x64 tile) in WHOLE rounds: tiles * splits must
    // not exceed a multiple of the slot count by a few workgroups (a 513th workgroup costs a full extra round).
    const long tiles = (long)adm_cdiv(Cout, TM) * adm_cdiv(Cin, TN) * ks * ks;
    const long slots = 256L * ((TM == 64 && TN == 64) ? 4 : 2);     // resident workgroups (LDS-limited)
    const int maxs = (p.P + 127) / 128;                     // keep >= 128 pixels (4 K-steps) per split
    if (tiles >= slots) {
      splits = 1;
    } else {
      splits = (int)(slots / tiles);
      if (splits > maxs) splits = maxs;
      if (splits < 1) splits = 1;
    }
  }
  int chunk = ((p.P + splits - 1) / splits + 31) & ~31;
  splits = (p.P + chunk - 1) / chunk;
  if (plan_only) return splits;
  p.chunk = chunk;
  p.atomic = splits > 1 && split_stride == 0;
  if (p.atomic && !prezeroed &&
      hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)Cout * ks * ks * Cin, stream) != hipSuccess)
    return ADM_ELAUNCH;
  if (TM == 192) return launch_wgrad<192, 64>(p, splits, stream);
  if (TM == 128 && TN == 128) return launch_wgrad<128, 128>(p, splits, stream);
  if (TM == 128 && TN == 64) return launch_wgrad<128, 64>(p, splits, stream);
  if (TM == 64 && TN == 128) return launch_wgrad<64, 128>(p, splits, stream);
  return launch_wgrad<64, 64>(p, splits, stream);
}
}  // namespace

extern "C" int adm_conv_wgrad(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx,
                              int Cout, int lddy, int ks, int up, int splits, hipStream_t stream) {
  return adm_conv_wgrad_bias(x, dy, dwp, nullptr, B, H, W, Cin, ldx, Cout, lddy, ks, up, splits, stream);
}

extern "C" int adm_conv_wgrad_bias(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin,
                                   int ldx, int Cout, int lddy, int ks, int up, int splits, hipStream_t stream) {
  return wgrad_impl(x, dy, dwp, dbias, B, H, W, Cin, ldx, Cout, lddy, ks, up, splits, 0, 0, false, stream);
}

int adm_wgrad_wino_plan(int B, int H, int W, int Cin, int Cout, int mode);      // conv_wgrad_wino.hip; mode 0 / 1 (up) / 2 (2-D)
int adm_wgrad_wino_ws(const float* x, const float* dy, float* ws, float* bws, int B, int H, int W, int Cin, int ldx, int Cout,
                      int lddy, int splits, int mode, hipStream_t stream);

extern "C" int adm_conv_wgrad_plan(int B, int H, int W, int Cin, int Cout, int ks, int up, int wino) {
  if (wino) return adm_wgrad_wino_plan(B, H, W, Cin, Cout, wino == 2 ? 2 : (up ? 1 : 0));
  return wgrad_impl(nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Cin, Cout, Cout, ks, up, 0, 0, 0, true, nullptr);
}

extern "C" int adm_conv_wgrad_ws(const float* x, const float* dy, float* ws, float* bws, int B, int H, int W, int Cin, int ldx,
                                 int Cout, int lddy, int ks, int up, int splits, int wino, hipStream_t stream) {
  if (splits < 1 || !ws) return ADM_EINVAL;
  if (wino) {
    if (ks != 3) return ADM_EINVAL;
    return adm_wgrad_wino_ws(x, dy, ws, bws, B, H, W, Cin, ldx, Cout, lddy, splits, wino == 2 ? 2 : (up ? 1 : 0), stream);
  }
  return wgrad_impl(x, dy, ws, bws, B, H, W, Cin, ldx, Cout, lddy, ks, up, splits, (long)Cout * ks * ks * Cin, Cout, false,
                    stream);
}

// Weight gradient of a strided conv with explicit top/left padding (the conditional UNet's Downsample = Conv2d(C, C', 4, 2, 1),
// /root/reference/unet/cond_unet_sd.py:341-342): dy is [B][Hout][Wout][Cout], x is [B][Hin][Win][Cin]; tap (ky, kx) of output
// (oy, ox) reads x(oy*stride + ky - pad_lo, ox*stride + kx - pad_lo), zero outside.  dwp[Cout][ks*ks][Cin] as adm_conv_wgrad.
extern "C" int adm_conv_wgrad_strided(const float* x, const float* dy, float* dwp, float* dbias, int B, int Hin, int Win, int Hout,
                                      int Wout, int Cin, int ldx, int Cout, int lddy, int ks, int stride, int pad_lo,
                                      hipStream_t stream) {
  if (Hin <= 0 || Win <= 0 || pad_lo < 0 || pad_lo >= ks) return ADM_EINVAL;
  if ((long)(Hout - 1) * stride - pad_lo >= Hin || (long)(Wout - 1) * stride - pad_lo >= Win) return ADM_EINVAL;
  return wgrad_impl(x, dy, dwp, dbias, B, Hout, Wout, Cin, ldx, Cout, lddy, ks, 0, 0, 0, 0, false, stream, stride, pad_lo, Hin, Win);
}
