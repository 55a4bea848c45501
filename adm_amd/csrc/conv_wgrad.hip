// Weight gradient of the 3x3 / 1x1 / Linear layers on the fp32-input MFMA (gfx950).
//
//   dWp[co][tap][ci] += sum_{p in split} dY[p][co] * X[pix(p) + tap][ci]
//
// GEMM view: M' = Cout (rows of dY^T), N' = Cin, K' = pixels.  Both operands arrive pixel-major
// (NHWC), i.e. "k-major" with the m / n index contiguous, so tiles are staged into LDS exactly as
// they lie in memory ([32 pixels][TM] and [32 pixels][TN]) and MFMA fragments are read with
// conflict-free ds_read_b32 (lane = consecutive channel; the two wave halves read pixel 2s and 2s+1).
// The reduction over pixels is split across gridDim.z; partial tiles are combined with fp32 global
// atomics shaped as 128-byte row segments (the full-rate shape on MI355X), or plain stores when
// splits == 1.  Autograd counterpart of F.conv2d's weight gradient in the reference
// (/root/reference/unet/uncond_unet.py:98-110 under loss.backward()).
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

struct WgradP {
  const float* x; const float* dy; float* dwp;
  int P, H, W, Hin, Win, Cin, ldx, Cout, lddy, ks, up, tilesN, chunk, atomic, lw, lh, xbytes, dybytes;
};

template <int TM, int TN, bool FAST>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradP p) {
  constexpr int WM = 2, WN = 2;
  constexpr int MT = TM / (WM * 32), NT = TN / (WN * 32);
  constexpr int AQ = TM / 4, BQ = TN / 4;              // float4 per tile row
  constexpr int AR = 256 / AQ, BR = 256 / BQ;          // rows covered per pass
  constexpr int AI = 32 / AR, BI = 32 / BR;            // passes
  __shared__ __attribute__((aligned(16))) float As[2][32][TM];
  __shared__ __attribute__((aligned(16))) float Bs[2][32][TN];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int tn = blockIdx.x % p.tilesN, tm = blockIdx.x / p.tilesN;
  const int co0 = tm * TM, ci0 = tn * TN;
  const int tap = blockIdx.y;
  const int pad = p.ks >> 1;
  const int dy_ = (p.ks == 3) ? tap / 3 - pad : 0, dx_ = (p.ks == 3) ? tap % 3 - pad : 0;
  const int pbeg = blockIdx.z * p.chunk;
  const int pend = min(p.P, pbeg + p.chunk);
  if (pbeg >= pend) return;
  const int KT = (pend - pbeg + 31) >> 5;

  const int a_c = tid % AQ, a_r = tid / AQ;
  const int b_c = tid % BQ, b_r = tid / BQ;
  // Raw buffer loads (see conv_igemm.hip): 32-bit offsets, and the descriptor's range check supplies the
  // zeros for out-of-image taps, channel-edge tiles and pixels past the end -- no selects, little VALU.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dybytes, 0x00020000);
  unsigned a_voff[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i)
    a_voff[i] = (co0 + a_c * 4 < p.Cout) ? (unsigned)((a_r + i * AR) * p.lddy + co0 + a_c * 4) * 4u : OOB;
  const unsigned b_colb = (ci0 + b_c * 4 < p.Cin) ? (unsigned)(ci0 + b_c * 4) * 4u : OOB;
  // fast path state (see load_stage)
  const int shift = (dy_ * p.W + dx_) * p.ldx;                       // floats; may be negative
  const __amdgpu_buffer_rsrc_t rs_xt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + shift, 0, p.xbytes - shift * 4, 0x00020000);
  unsigned b_voff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int r = b_r + i * BR;
    const bool xv = (unsigned)((r & (p.W - 1)) + dx_) < (unsigned)p.W;
    b_voff[i] = (xv && b_colb != OOB) ? (unsigned)(r * p.ldx) * 4u + b_colb : OOB;
  }

  f32x4 ra[AI], rb[BI];
  auto load_stage = [&](int s) {
    const int pb = pbeg + (s << 5);
    const int a_soff = pb * p.lddy * 4;       // rows past P fall outside the descriptor -> zeros
    if (pb + 32 <= pend) {          // wave-uniform: only a ragged last stage needs the per-row test
#pragma unroll
      for (int i = 0; i < AI; ++i)
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)a_voff[i], a_soff, 0));
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        unsigned vo = (pb + a_r + i * AR < pend) ? a_voff[i] : OOB;
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)vo, a_soff, 0));
      }
    }
    if constexpr (FAST) {
      // power-of-two image, W <= 32, no up-sampling: the column test and the byte offset inside the stage are
      // per-thread constants, the stage's pixel base rides in the scalar offset and the tap shift is folded into
      // the descriptor's base -> per row only the image-row test (and nothing at all for the centre-row taps).
      const int b_soff = pb * p.ldx * 4;
      const int U = pb >> p.lw;
      const bool full = pb + 32 <= pend;
      if (dy_ == 0 && full) {
#pragma unroll
        for (int i = 0; i < BI; ++i)
          rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)b_voff[i], b_soff, 0));
      } else {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
          const int r = b_r + i * BR;
          const int iy = ((U + (r >> p.lw)) & (p.H - 1)) + dy_;
          const bool v = (unsigned)iy < (unsigned)p.H && (full || pb + r < pend);
          rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_xt, (int)(v ? b_voff[i] : OOB), b_soff, 0));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        int pp = pb + b_r + i * BR;
        int ox, oy, b;
        if (p.lw >= 0) {            // power-of-two image: shifts instead of integer division
          ox = pp & (p.W - 1);
          int t = pp >> p.lw;
          oy = t & (p.H - 1);
          b = t >> p.lh;
        } else {
          ox = pp % p.W;
          int t = pp / p.W;
          oy = t % p.H;
          b = t / p.H;
        }
        int iy = oy + dy_, ix = ox + dx_;
        bool v = pp < pend && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        if (p.up) { iy >>= 1; ix >>= 1; }
        unsigned voff = v ? (unsigned)(((b * p.Hin + iy) * p.Win + ix) * p.ldx) * 4u + b_colb : OOB;
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff, 0, 0));
      }
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[buf][a_r + i * AR][a_c * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][b_r + i * BR][b_c * 4]) = rb[i];
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
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = As[buf][2 * k + lh][(wm * MT + i) * 32 + lr];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = Bs[buf][2 * k + lh][(wn * NT + j) * 32 + lr];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < KT) store_stage(buf ^ 1);
    __syncthreads();
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
          float* dst = p.dwp + ((long)co * taps + tap) * p.Cin + ci;
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
  const bool fast = p.lw >= 0 && p.W <= 32 && !p.up;       // see load_stage
  if (fast) hipLaunchKernelGGL((wgrad_f32_kernel<TM, TN, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((wgrad_f32_kernel<TM, TN, false>), grid, dim3(256), 0, st, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

}  // namespace

extern "C" int adm_conv_wgrad(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx,
                              int Cout, int lddy, int ks, int up, int splits, hipStream_t stream) {
  if (!x || !dy || !dwp || B <= 0 || H <= 0 || W <= 0) return ADM_EINVAL;
  if ((Cin & 31) || (Cout & 31) || (ldx & 3) || (lddy & 3) || (ks != 1 && ks != 3)) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1))) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)dy) & 15) return ADM_EINVAL;
  WgradP p;
  p.x = x; p.dy = dy; p.dwp = dwp;
  p.P = B * H * W; p.H = H; p.W = W; p.Hin = up ? H / 2 : H; p.Win = up ? W / 2 : W;
  p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.lddy = lddy; p.ks = ks; p.up = up; p.tilesN = 0;
  const long xb = (long)B * p.Hin * p.Win * ldx * 4, db = (long)p.P * lddy * 4;
  if (xb >= (1L << 31) || db >= (1L << 31)) return ADM_EINVAL;      // 32-bit buffer offsets
  p.xbytes = (int)xb; p.dybytes = (int)db;
  auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  p.lw = ilog2(W); p.lh = ilog2(H);
  if (p.lw < 0 || p.lh < 0) p.lw = p.lh = -1;
  const int TM = (Cout % 128 == 0) ? 128 : 64;
  const int TN = (Cin % 128 == 0) ? 128 : 64;
  if (splits <= 0) {   // aim for >= ~512 workgroups, keep >= 256 pixels per split
    long tiles = (long)adm_cdiv(Cout, TM) * adm_cdiv(Cin, TN) * ks * ks;
    splits = (int)((512 + tiles / 2) / tiles);      // ~one full round of 2 workgroups per CU
    int maxs = (p.P + 511) / 512;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  int chunk = ((p.P + splits - 1) / splits + 31) & ~31;
  splits = (p.P + chunk - 1) / chunk;
  p.chunk = chunk;
  p.atomic = splits > 1;
  if (p.atomic &&
      hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)Cout * ks * ks * Cin, stream) != hipSuccess)
    return ADM_ELAUNCH;
  if (TM == 128 && TN == 128) return launch_wgrad<128, 128>(p, splits, stream);
  if (TM == 128 && TN == 64) return launch_wgrad<128, 64>(p, splits, stream);
  if (TM == 64 && TN == 128) return launch_wgrad<64, 128>(p, splits, stream);
  return launch_wgrad<64, 64>(p, splits, stream);
}
