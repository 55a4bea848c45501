// bf16-MFMA variant of the implicit-GEMM convolution (BASELINE.json configs[2]: "bf16").
//
// Same contract, layouts and addressing as conv_igemm.hip (NHWC fp32 activations in HBM, buffer loads, range check
// as zero padding, fp32 bias/residual epilogue, fp32 output); only the contraction changes: operands are rounded to
// bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into LDS and multiplied on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 16x the per-clock rate of the fp32 MFMA, and, unlike it, the
// bf16 MFMA overlaps with the vector ALU, so the conversions are free.  Weights are pre-packed in bf16
// (adm_pack_weight_bf16).  This is a reduced-precision mode: it is never the default, parity tests use a bf16
// tolerance, and bench.py labels results obtained with it as dtype "bf16".
//
// Tiling: K-step 64 (one barrier per 16 MFMAs per wave), LDS rows of 64 bf16 + 8 pad = 144 B: a ds_read_b128
// fragment read (8 consecutive k of one row) by a 16-lane group touches 16 distinct 16-byte slots.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

struct IgemmBP {
  const float* x; const unsigned short* w; const float* bias; const float* res; float* y;
  int M, N, H, W, Hin, Win, Cin, ldx, K, ldy, ldr, ks, up, wrows, tilesN, xbytes, wbytes, abf;
};

constexpr int BK = 64;
constexpr int LROW = BK + 8;     // bf16 elements per LDS row (144 bytes)

__device__ __forceinline__ bf16x4 cvt4(f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
  return r;
}

// ABF: the A operand is ALREADY bf16 in HBM (x16[B][H][W][ldx] bf16, e.g. written by adm_gn_fwd_bf16out): 16-byte loads of 8
// channels go straight to LDS -- half the bytes, no conversion.  Values equal what the f32 path would round to: bit-identical results.
template <int BM, int BN, int WM, int WN, bool UP, bool ABF>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(IgemmBP p) {
  constexpr int MT = BM / (WM * 32), NT = BN / (WN * 32);
  constexpr int AI = ABF ? BM / 32 : BM / 16;      // A: f32 float4 (4 k) per thread per stage: BM rows x 16 quads / 256; bf16: x 8 octets
  constexpr int ARS = ABF ? 32 : 16;               // row step between a thread's A items
  constexpr int AEB = ABF ? 2 : 4;                 // bytes per A element
  constexpr int BI = BN / 32;          // B: 8 bf16 (16 bytes) per thread per stage: BN rows x 8 octets / 256
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_b[];
  __bf16* As = smem_b;                       // [2][BM][LROW]
  __bf16* Bs = smem_b + 2 * BM * LROW;       // [2][BN][LROW]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int lr = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap, m-fastest inside an n-tile (see conv_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int m0 = tm * BM, n0 = tn * BN;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.wbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  // A loader: thread owns fp32 quad a_c4 (of 16) of rows a_r0 + 16 i  (bf16 A: octet a_c4 (of 8) of rows a_r0 + 32 i)
  const int a_c4 = ABF ? (tid & 7) : (tid & 15), a_r0 = ABF ? (tid >> 3) : (tid >> 4);
  unsigned a_pix[AI], a_mask[AI], a_voff[AI];
  const int pad = p.ks >> 1;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + a_r0 + ARS * i;
    bool ok = m < p.M;
    int mm = ok ? m : 0;
    int ox = mm % p.W;
    int t = mm / p.W;
    int oy = t % p.H;
    int b = t / p.H;
    unsigned mask = 0;
    if (ok) {
      if (p.ks == 3) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          int iy = oy + tp / 3 - 1, ix = ox + tp % 3 - 1;
          if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mask |= 1u << tp;
        }
      } else {
        mask = 1u;
      }
    }
    if (UP) mask |= ((unsigned)(oy & 1) << 16) | ((unsigned)(ox & 1) << 17);
    a_mask[i] = mask;
    int py = UP ? (oy >> 1) : oy, px = UP ? (ox >> 1) : ox;
    a_pix[i] = (unsigned)(((b * p.Hin + py) * p.Win + px) * p.ldx + a_c4 * (ABF ? 8 : 4)) * (unsigned)AEB;
    a_voff[i] = OOB;
  }
  // B loader: thread owns bf16 octet b_c8 (of 8) of rows b_r0 + 32 i
  const int b_c8 = tid & 7, b_r0 = tid >> 3;
  unsigned b_voff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = n0 + b_r0 + 32 * i;
    b_voff[i] = (n < p.wrows) ? (unsigned)(n * p.K + b_c8 * 8) * 2u : OOB;
  }
  const int cchunks = p.Cin / BK;
  const int KT = p.ks * p.ks * cchunks;

  f32x4 ra[AI];
  f32x4 rb[BI];                     // 8 bf16 each, kept as raw 16 bytes
  int ld_tap = 0, ld_cc = 0;
  auto load_stage = [&]() {
    const int tap = ld_tap;
    if (ld_cc == 0) {
      int dy = 0, dx = 0;
      if (p.ks == 3) { dy = tap / 3 - pad; dx = tap - (tap / 3) * 3 - pad; }
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool v = (a_mask[i] >> tap) & 1u;
        int off = (dy * p.Win + dx) * p.ldx;
        if (UP) {
          int py = (int)((a_mask[i] >> 16) & 1u), px = (int)((a_mask[i] >> 17) & 1u);
          int qy = ((py + dy + 2) >> 1) - 1, qx = ((px + dx + 2) >> 1) - 1;
          off = (qy * p.Win + qx) * p.ldx;
        }
        a_voff[i] = v ? a_pix[i] + (unsigned)(off * AEB) : OOB;
      }
    }
    const int c0 = ld_cc * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[i], c0 * AEB, 0));
    const int kb = (tap * p.Cin + c0) * 2;
#pragma unroll
    for (int i = 0; i < BI; ++i)
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)b_voff[i], kb, 0));
    if (++ld_cc == cchunks) { ld_cc = 0; ++ld_tap; }
  };
  auto store_stage = [&](int buf) {
    __bf16* Ab = As + buf * BM * LROW;
    __bf16* Bb = Bs + buf * BN * LROW;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      if (ABF) *reinterpret_cast<f32x4*>(&Ab[(a_r0 + ARS * i) * LROW + a_c4 * 8]) = ra[i];       // 8 bf16, as loaded
      else *reinterpret_cast<bf16x4*>(&Ab[(a_r0 + ARS * i) * LROW + a_c4 * 4]) = cvt4(ra[i]);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i)
      *reinterpret_cast<f32x4*>(&Bb[(b_r0 + 32 * i) * LROW + b_c8 * 8]) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_stage();
  store_stage(0);
  __syncthreads();

  for (int s = 0; s < KT; ++s) {
    const int buf = s & 1;
    if (s + 1 < KT) load_stage();
    const __bf16* Ab = As + buf * BM * LROW + (wm * MT * 32 + lr) * LROW + lh * 8;
    const __bf16* Bb = Bs + buf * BN * LROW + (wn * NT * 32 + lr) * LROW + lh * 8;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {      // lane (row r, half h) supplies k = 16 ks + 8 h + 0..7
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

  const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if (full) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + (wn * NT + j) * 32 + lr;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const long mb = m0 + (wm * MT + i) * 32 + 4 * lh;
        float rv[16];
        if (p.res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = p.res[(mb + (r & 3) + 8 * (r >> 2)) * p.ldr + n];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) p.y[(mb + (r & 3) + 8 * (r >> 2)) * p.ldy + n] = acc[i][j][r] + bv + rv[r];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 32 + lr;
    if (n >= p.N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int mb = m0 + (wm * MT + i) * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < p.M) {
          float v = acc[i][j][r] + bv;
          if (p.res) v += p.res[(long)m * p.ldr + n];
          p.y[(long)m * p.ldy + n] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, bool UP, bool ABF>
int launch_b_up(IgemmBP p, hipStream_t st) {
  static bool attr_set = false;
  constexpr int smem = 2 * (BM + BN) * LROW * 2;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_bf16_kernel<BM, BN, WM, WN, UP, ABF>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  p.tilesN = adm_cdiv(p.N, BN);
  long grid = (long)adm_cdiv(p.M, BM) * p.tilesN;
  hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, UP, ABF>), dim3((unsigned)grid), dim3(256), smem, st, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
template <int BM, int BN, int WM, int WN>
int launch_b(IgemmBP p, hipStream_t st) {
  if (p.abf) return p.up ? launch_b_up<BM, BN, WM, WN, true, true>(p, st) : launch_b_up<BM, BN, WM, WN, false, true>(p, st);
  return p.up ? launch_b_up<BM, BN, WM, WN, true, false>(p, st) : launch_b_up<BM, BN, WM, WN, false, false>(p, st);
}

__global__ void pack_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    __bf16 v = (__bf16)src[i];
    dst[i] = __builtin_bit_cast(unsigned short, v);
  }
}

}  // namespace

extern "C" int adm_f32_to_bf16(const float* src, unsigned short* dst, long n, hipStream_t stream) {
  if (!src || !dst || n <= 0) return ADM_EINVAL;
  long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)b), dim3(256), 0, stream, src, dst, n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

static int conv_fwd_bf16_impl(const float* x, const unsigned short* wp, const float* bias, const float* res, float* y, int B, int H, int W,
                              int Cin, int ldx, int N, int wrows, int ldy, int ldr, int ks, int up, int tile, int abf,
                              hipStream_t stream);

extern "C" int adm_conv_fwd_bf16(const float* x, const unsigned short* wp, const float* bias, const float* res,
                                 float* y, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                                 int ks, int up, int tile, hipStream_t stream) {
  return conv_fwd_bf16_impl(x, wp, bias, res, y, B, H, W, Cin, ldx, N, wrows, ldy, ldr, ks, up, tile, 0, stream);
}

// the same with the activation ALREADY stored as bf16: x16[B][Hin][Win][ldx] (bf16 elements; ldx % 8 == 0)
extern "C" int adm_conv_fwd_bf16a(const void* x16, const unsigned short* wp, const float* bias, const float* res, float* y, int B,
                                  int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr, int ks, int up, int tile,
                                  hipStream_t stream) {
  if (ldx & 7) return ADM_EINVAL;
  return conv_fwd_bf16_impl(static_cast<const float*>(x16), wp, bias, res, y, B, H, W, Cin, ldx, N, wrows, ldy, ldr, ks, up, tile, 1,
                            stream);
}

static int conv_fwd_bf16_impl(const float* x, const unsigned short* wp, const float* bias, const float* res, float* y, int B, int H, int W,
                              int Cin, int ldx, int N, int wrows, int ldy, int ldr, int ks, int up, int tile, int abf,
                              hipStream_t stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0) return ADM_EINVAL;
  if ((Cin % BK) || (ldx & 3) || (ks != 1 && ks != 3) || N <= 0 || wrows < N) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1))) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)wp) & 15) return ADM_EINVAL;
  IgemmBP p;
  p.x = x; p.w = wp; p.bias = bias; p.res = res; p.y = y;
  p.M = B * H * W; p.N = N; p.H = H; p.W = W;
  p.Hin = up ? H / 2 : H; p.Win = up ? W / 2 : W;
  p.Cin = Cin; p.ldx = ldx; p.K = ks * ks * Cin; p.ldy = ldy; p.ldr = ldr; p.ks = ks; p.up = up; p.wrows = wrows;
  p.tilesN = 0; p.abf = abf;
  const long xb = (long)B * p.Hin * p.Win * ldx * (abf ? 2 : 4), wb = (long)wrows * p.K * 2;
  if (xb >= (1L << 31) || wb >= (1L << 31)) return ADM_EINVAL;
  p.xbytes = (int)xb; p.wbytes = (int)wb;
  if (tile < 0) {
    struct Cand { int id, bm, bn, per_cu; double eff; };
    const Cand cands[3] = {{0, 128, 128, 2, 1.00}, {1, 128, 96, 2, 0.95}, {2, 64, 64, 4, 0.80}};
    double best = 1e300;
    for (const Cand& c : cands) {
      long tiles = (long)adm_cdiv(p.M, c.bm) * adm_cdiv(N, c.bn);
      long slots = 256L * c.per_cu;
      long rounds = (tiles + slots - 1) / slots;
      double t = (double)rounds * c.per_cu * c.bm * c.bn / c.eff;
      if (t < best) { best = t; tile = c.id; }
    }
  }
  switch (tile) {
    case 0: return launch_b<128, 128, 2, 2>(p, stream);
    case 1: return launch_b<128, 96, 4, 1>(p, stream);
    case 2: return launch_b<64, 64, 2, 2>(p, stream);
    default: return ADM_EINVAL;
  }
}
