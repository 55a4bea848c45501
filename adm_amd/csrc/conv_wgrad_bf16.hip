// bf16-MFMA variant of the weight-gradient kernel (see conv_wgrad.hip for the formulation and conv_igemm_bf16.hip
// for the precision contract: fp32 tensors in HBM, operands rounded to bf16 into LDS, fp32 accumulation, fp32
// gradient out).  K-step = 64 pixels per barrier; tiles are transposed on the way into LDS as
// [channel][64 pixels + 8 pad] bf16: each thread loads the SAME channel quad of two adjacent pixels and writes four
// packed (pixel, pixel+1) pairs with ds_write_b32 -- a 32-lane group (2 quads x 16 pixel pairs) covers the 32 banks
// exactly once -- so the compute loop reads 8 consecutive pixels of a channel row with one ds_read_b128.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct WgradBP {
  const float* x; const float* dy; float* dwp;
  int P, H, W, Hin, Win, Cin, ldx, Cout, lddy, ks, up, tilesN, chunk, atomic, lw, lh, xbytes, dybytes, xbf;
};

constexpr int PK = 64;            // pixels per stage
constexpr int LROW = PK + 8;      // bf16 per LDS row (144 bytes)

// XBF: x is ALREADY bf16 in HBM (the bf16-storage mode: written by adm_gn_fwd_bf16out): 8-byte loads of a channel quad, widened to
// the f32 values the f32 path would have rounded to -- bit-identical results, half the x traffic.
template <int TM, int TN, bool FAST, bool XBF>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgradBP p) {
  constexpr int WM = 2, WN = 2;
  constexpr unsigned XEB = XBF ? 2u : 4u;          // bytes per x element
  auto load_x = [](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff) -> f32x4 {
    if constexpr (XBF) {
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t h = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, soff, 0));
      return f32x4{__uint_as_float(h[0] << 16), __uint_as_float(h[0] & 0xFFFF0000u), __uint_as_float(h[1] << 16),
                   __uint_as_float(h[1] & 0xFFFF0000u)};
    } else {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
    }
  };
  constexpr int MT = TM / (WM * 32), NT = TN / (WN * 32);
  constexpr int AU = TM / 32, BU = TN / 32;        // (32 pixel x 16 channel) units per wave per stage
  __shared__ __attribute__((aligned(16))) __bf16 As[2][TM][LROW];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TN][LROW];

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
  const int KT = (pend - pbeg + PK - 1) / PK;

  // unit u = wid*U + i: pixel half (u & 1) -> pixels 32 (u&1) + 2 (lane & 15) + {0,1}; channel quad (u >> 1)*4 + (lane >> 4)
  const int lp = lane & 15, lq = lane >> 4;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const int shift = (dy_ * p.W + dx_) * p.ldx;       // elements
  const __amdgpu_buffer_rsrc_t rs_xt = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(p.x)) + (long)shift * (int)XEB, 0, max(0, p.xbytes - shift * (int)XEB), 0x00020000);
  // (a tap shift larger than the whole tensor -- 1 x 2 images -- must not wrap num_records around: 0 records = every
  //  access out of range = zeros, which is what such a tap contributes)

  int a_pix[AU], a_row[AU], b_pix[BU], b_row[BU];        // first pixel of the pair (stage-local), first channel of the quad
  unsigned a_voff[AU], b_voff[BU][2], b_colb[BU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = wid * AU + i;
    a_pix[i] = (u & 1) * 32 + 2 * lp;
    a_row[i] = ((u >> 1) * 4 + lq) * 4;
    a_voff[i] = (co0 + a_row[i] < p.Cout) ? (unsigned)(a_pix[i] * p.lddy + co0 + a_row[i]) * 4u : OOB;
  }
#pragma unroll
  for (int i = 0; i < BU; ++i) {
    const int u = wid * BU + i;
    b_pix[i] = (u & 1) * 32 + 2 * lp;
    b_row[i] = ((u >> 1) * 4 + lq) * 4;
    b_colb[i] = (ci0 + b_row[i] < p.Cin) ? (unsigned)(ci0 + b_row[i]) * XEB : OOB;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int px = b_pix[i] + e;
      const bool xv = (unsigned)((px & (p.W - 1)) + dx_) < (unsigned)p.W;
      b_voff[i][e] = (xv && b_colb[i] != OOB) ? (unsigned)(px * p.ldx) * XEB + b_colb[i] : OOB;
    }
  }

  f32x4 ra[AU][2], rb[BU][2];
  auto load_stage = [&](int s) {
    const int pb = pbeg + s * PK;
    const bool full = pb + PK <= pend;
    const int a_soff = pb * p.lddy * 4;
#pragma unroll
    for (int i = 0; i < AU; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        unsigned vo = a_voff[i] == OOB ? OOB : a_voff[i] + (unsigned)(e * p.lddy * 4);
        if (!full && pb + a_pix[i] + e >= pend) vo = OOB;
        ra[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)vo, a_soff, 0));
      }
    if constexpr (FAST) {
      const int b_soff = pb * p.ldx * (int)XEB;
      const int U = pb >> p.lw;
#pragma unroll
      for (int i = 0; i < BU; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int px = b_pix[i] + e;
          bool v = full || pb + px < pend;
          if (dy_ != 0) {
            const int iy = ((U + (px >> p.lw)) & (p.H - 1)) + dy_;
            v = v && (unsigned)iy < (unsigned)p.H;
          }
          rb[i][e] = load_x(rs_xt, v ? b_voff[i][e] : OOB, b_soff);
        }
    } else {
#pragma unroll
      for (int i = 0; i < BU; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int pp = pb + b_pix[i] + e;
          int ox = pp % p.W;
          int t = pp / p.W;
          int oy = t % p.H;
          int b = t / p.H;
          int iy = oy + dy_, ix = ox + dx_;
          bool v = pp < pend && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && b_colb[i] != OOB;
          if (p.up) { iy >>= 1; ix >>= 1; }
          unsigned voff = v ? (unsigned)(((b * p.Hin + iy) * p.Win + ix) * p.ldx) * XEB + b_colb[i] : OOB;
          rb[i][e] = load_x(rs_x, voff, 0);
        }
    }
  };
  auto store_stage = [&](int buf) {        // transpose + round: channel rows, (pixel, pixel+1) pairs as one dword
#pragma unroll
    for (int i = 0; i < AU; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x2 v = {(__bf16)ra[i][0][j], (__bf16)ra[i][1][j]};
        *reinterpret_cast<bf16x2*>(&As[buf][a_row[i] + j][a_pix[i]]) = v;
      }
#pragma unroll
    for (int i = 0; i < BU; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x2 v = {(__bf16)rb[i][0][j], (__bf16)rb[i][1][j]};
        *reinterpret_cast<bf16x2*>(&Bs[buf][b_row[i] + j][b_pix[i]]) = v;
      }
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
    const __bf16* Ab = &As[buf][wm * MT * 32 + lr][lh * 8];
    const __bf16* Bb = &Bs[buf][wn * NT * 32 + lr][lh * 8];
#pragma unroll
    for (int ks = 0; ks < PK / 16; ++ks) {
      bf16x8 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LROW + ks * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * LROW + ks * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
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
int launch_wb(WgradBP p, int splits, hipStream_t st) {
  p.tilesN = adm_cdiv(p.Cin, TN);
  dim3 grid(adm_cdiv(p.Cout, TM) * p.tilesN, p.ks * p.ks, splits);
  const bool fast = p.lw >= 0 && p.W <= 32 && !p.up;
  if (p.xbf) {
    if (fast) hipLaunchKernelGGL((wgrad_bf16_kernel<TM, TN, true, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<TM, TN, false, true>), grid, dim3(256), 0, st, p);
  } else {
    if (fast) hipLaunchKernelGGL((wgrad_bf16_kernel<TM, TN, true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<TM, TN, false, false>), grid, dim3(256), 0, st, p);
  }
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

}  // namespace

static int conv_wgrad_bf16_impl(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx, int Cout, int lddy,
                                int ks, int up, int splits, int xbf, hipStream_t stream);

extern "C" int adm_conv_wgrad_bf16(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx,
                                   int Cout, int lddy, int ks, int up, int splits, hipStream_t stream) {
  return conv_wgrad_bf16_impl(x, dy, dwp, B, H, W, Cin, ldx, Cout, lddy, ks, up, splits, 0, stream);
}

// the same with the saved activation ALREADY stored as bf16: x16[B][Hin][Win][ldx] (bf16 elements)
extern "C" int adm_conv_wgrad_bf16a(const void* x16, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx, int Cout,
                                    int lddy, int ks, int up, int splits, hipStream_t stream) {
  return conv_wgrad_bf16_impl(static_cast<const float*>(x16), dy, dwp, B, H, W, Cin, ldx, Cout, lddy, ks, up, splits, 1, stream);
}

static int conv_wgrad_bf16_impl(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx, int Cout, int lddy,
                                int ks, int up, int splits, int xbf, hipStream_t stream) {
  if (!x || !dy || !dwp || B <= 0 || H <= 0 || W <= 0) return ADM_EINVAL;
  if ((Cin & 31) || (Cout & 31) || (ldx & 3) || (lddy & 3) || (ks != 1 && ks != 3)) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1))) return ADM_EINVAL;
  if ((((uintptr_t)x) & (xbf ? 7 : 15)) || (((uintptr_t)dy) & 15)) return ADM_EINVAL;
  WgradBP p;
  p.x = x; p.dy = dy; p.dwp = dwp; p.xbf = xbf;
  p.P = B * H * W; p.H = H; p.W = W; p.Hin = up ? H / 2 : H; p.Win = up ? W / 2 : W;
  p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.lddy = lddy; p.ks = ks; p.up = up; p.tilesN = 0;
  const long xb = (long)B * p.Hin * p.Win * ldx * (xbf ? 2 : 4), db = (long)p.P * lddy * 4;
  if (xb >= (1L << 31) || db >= (1L << 31)) return ADM_EINVAL;
  p.xbytes = (int)xb; p.dybytes = (int)db;
  auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  p.lw = ilog2(W); p.lh = ilog2(H);
  if (p.lw < 0 || p.lh < 0) p.lw = p.lh = -1;
  const int TM = (Cout % 128 == 0) ? 128 : 64;
  const int TN = (Cin % 128 == 0) ? 128 : 64;
  const bool prezeroed = splits == ADM_SPLITS_AUTO_PREZEROED;      // zero-at-rest workspace: no memset
  if (splits <= 0) {
    const long tiles = (long)adm_cdiv(Cout, TM) * adm_cdiv(Cin, TN) * ks * ks;
    const long slots = 256L * 2;
    const int maxs = (p.P + 1023) / 1024;
    if (tiles >= slots) splits = 1;
    else {
      splits = (int)(slots / tiles);
      if (splits > maxs) splits = maxs;
      if (splits < 1) splits = 1;
    }
  }
  int chunk = ((p.P + splits - 1) / splits + PK - 1) / PK * PK;
  splits = (p.P + chunk - 1) / chunk;
  p.chunk = chunk;
  p.atomic = splits > 1;
  if (p.atomic && !prezeroed &&
      hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)Cout * ks * ks * Cin, stream) != hipSuccess)
    return ADM_ELAUNCH;
  if (TM == 128 && TN == 128) return launch_wb<128, 128>(p, splits, stream);
  if (TM == 128 && TN == 64) return launch_wb<128, 64>(p, splits, stream);
  if (TM == 64 && TN == 128) return launch_wb<64, 128>(p, splits, stream);
  return launch_wb<64, 64>(p, splits, stream);
}
