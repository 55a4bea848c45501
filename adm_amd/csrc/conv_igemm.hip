// Implicit-GEMM convolution (3x3 pad 1 / 1x1 / Linear) on the fp32-input MFMA of gfx950.
//
// Replaces, on the hot path, the reference's F.conv2d / conv_transpose2d / `x @ W.t()` calls
// (/root/reference/unet/uncond_unet.py:62-66, 98-113) for NHWC fp32 activations.
//
//   Y[m][n] = sum_{tap, c} X[pix(m) + tap][c] * Wp[n][tap*Cin + c]  (+ bias[n]) (+ R[m][n])
//
//   M = B*H*W output pixels, N = output channels, K = taps*Cin.
//   A operand  : gathered on the fly from the NHWC input (zero padding; optional nearest x2
//                up-sampling folded into the gather: `up` blocks never materialise the 4x tensor)
//   B operand  : pre-packed weights Wp[N][K] (k contiguous), see pack_weights.hip
//   math       : v_mfma_f32_32x32x2_f32 -- exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD,
//                which is the fp32 peak of the chip (157 TFLOP/s); no reduced precision anywhere.
//
// Tiling: 256 threads = 4 waves; workgroup tile BM x BN, K-step 32, double-buffered LDS, one barrier per K-step.
// Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`): the tiles go global -> LDS with no VGPR round trip, no
// ds_write and no vector-ALU work (which matters doubly here: the fp32 MFMA shares the SIMD's fp32 lanes).  One
// wave instruction writes 1 KiB = 8 rows x 32 floats LINEARLY, so rows cannot be padded; instead the 16-byte
// slots of a row are XOR-swizzled with (row >> 1) & 7 -- applied to the per-lane SOURCE address on the way in and
// to the ds_read_b128 address on the way out -- which spreads the 16 rows read by a lane group over the 16 slots
// of a 256-byte bank-row pair (conflict-free).  The buffer descriptor's range check writes ZEROS for out-of-image
// taps / edge rows (offset 0x80000000), which is the convolution's zero padding.
// Each lane reads float4 = 4 consecutive k of its row; lanes 0-31 take k 0-3 of an 8-group and lanes 32-63 take
// k 4-7, so MFMA j of the group consumes k = j (low half) and k = 4 + j (high half) -- a fixed permutation of the
// k order applied to A and B alike.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

struct IgemmP {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int M, N, H, W, Hin, Win, Cin, ldx, K, ldy, ldr, ks, up, wrows, tilesN, xbytes, wbytes;
  int splitk, kt_per_split; float* ws;     // split-K: partial tiles go to ws[z][M][N], summed by splitk_reduce_kernel
  int stride, cshift;                      // input centre of output (oy, ox) = (oy*stride + cshift, ox*stride + cshift)
};

constexpr int LDSS = 32;   // floats per LDS row (unpadded: LDS-DMA writes linearly; slots are XOR-swizzled)
typedef __attribute__((address_space(3))) void lds_void;

template <int BM, int BN, int WM, int WN, bool UP>
__global__ __launch_bounds__(256) void igemm_f32_kernel(IgemmP p) {
  constexpr int MT = BM / (WM * 32), NT = BN / (WN * 32);
  constexpr int AI = BM / 32, BI = BN / 32;         // LDS-DMA instructions per wave per stage (8 rows each)
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM][LDSS]
  float* Bs = smem + 2 * BM * LDSS;       // [2][BN][LDSS]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int lr = lane & 31, lh = lane >> 5;
  // Workgroup -> tile map, XCD-aware: consecutive block ids are dealt round-robin over the 8 XCDs, each with its own
  // L2.  Give every XCD a CONTIGUOUS chunk of the tile list (bijective remap) and order the list m-fastest inside an
  // n-tile, so the workgroups resident on one XCD at a time share the same weight slice (BN x K floats, fits the 4 MB
  // L2) and neighbouring pixel tiles (shared halo rows).  Placement only affects speed / traffic, never results.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- loader state.  DMA instruction i of wave w fills rows (w*I + i)*8 .. +7 of the tile: lane -> row +(lane>>3),
  // physical 16-byte slot lane&7, and it must FETCH logical slot (lane&7) ^ ((row>>1)&7) of that row.
  // Per row a 32-bit byte offset that already includes the current tap; recomputed only when the tap changes (every
  // Cin/32 K-steps) and set to 0x80000000 (-> zeros from the range check) for rows / taps outside the image; the
  // channel-chunk offset rides in the instruction's scalar offset: no vector ALU work per K-step at all.
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.wbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  unsigned a_pix[AI], a_mask[AI], a_voff[AI];
  const int pad = p.ks >> 1;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (wid * AI + i) * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    bool ok = m < p.M;
    int mm = ok ? m : 0;
    int ox = mm % p.W;
    int t = mm / p.W;
    int oy = t % p.H;
    int b = t / p.H;
    unsigned mask = 0;
    // centre (tap (1,1) of a 3x3) in the coordinates the bounds are tested in: the up-sampled grid for UP, else the
    // input image; a strided conv (the KL autoencoder's Downsample) moves it to oy*stride + (1 - pad_lo)
    const int cy = UP ? oy : oy * p.stride + p.cshift, cx = UP ? ox : ox * p.stride + p.cshift;
    const int HB = UP ? p.H : p.Hin, WB = UP ? p.W : p.Win;
    if (p.ks > 3) {
      // generic filter (7x7 stem, 4x4 stride-2 down-conv of the conditional UNet): keep the centre coordinates, biased by
      // 64 so that a negative centre (pad_lo > ks/2 never happens, but cshift may be 0 with ky - pad_lo < 0) stays unsigned;
      // validity is evaluated per tap in issue_stage.  Rows past M get a centre no tap can bring into the image.
      mask = ok ? (((unsigned)(cy + 64) << 16) | (unsigned)(cx + 64)) : 0xFFFFFFFFu;
    } else if (ok) {
      if (p.ks == 3) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          int iy = cy + tp / 3 - 1, ix = cx + tp % 3 - 1;
          if ((unsigned)iy < (unsigned)HB && (unsigned)ix < (unsigned)WB) mask |= 1u << tp;
        }
      } else {
        mask = 1u;
      }
    }
    // for the fused nearest-x2 the tap offset depends on the parity of (oy, ox): keep it in the high bits
    if (UP) mask |= ((unsigned)(oy & 1) << 16) | ((unsigned)(ox & 1) << 17);      // (UP implies ks <= 3: checked on the host)
    a_mask[i] = mask;
    int py = UP ? (oy >> 1) : cy, px = UP ? (ox >> 1) : cx;
    a_pix[i] = (unsigned)(((b * p.Hin + py) * p.Win + px) * p.ldx + ls * 4) * 4u;     // byte offset of the centre pixel
    a_voff[i] = OOB;
  }
  const int cchunks = p.Cin >> 5;
  const int KTall = p.ks * p.ks * cchunks;
  const int s_begin = (p.splitk > 1) ? (int)blockIdx.y * p.kt_per_split : 0;
  const int KT = (p.splitk > 1) ? min(p.kt_per_split, KTall - s_begin) : KTall;     // stages of THIS workgroup
  if (p.splitk > 1) {          // raw partial tile, no bias / residual: those are applied by the reduction
    p.y = p.ws + (long)blockIdx.y * p.M * p.N;
    p.ldy = p.N; p.bias = nullptr; p.res = nullptr;
  }
  unsigned b_voff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = (wid * BI + i) * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((row >> 1) & 7);
    int n = n0 + row;
    b_voff[i] = (n < p.wrows) ? (unsigned)(n * p.K + ls * 4) * 4u : OOB;
  }

  int ld_tap = s_begin / cchunks, ld_cc = s_begin - ld_tap * cchunks;   // (tap, channel chunk) of the NEXT stage to load
  bool a_fresh = true;             // a split-K range may start in the middle of a tap
  auto issue_stage = [&](int buf) {
    const int tap = ld_tap;
    if (ld_cc == 0 || a_fresh) {   // new tap: rebuild the per-row offsets (wave-uniform branch, every Cin/32 steps)
      a_fresh = false;
      int dy = 0, dx = 0;
      if (p.ks >= 3) { dy = tap / p.ks - pad; dx = tap - (tap / p.ks) * p.ks - pad; }
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        bool v;
        if (p.ks > 3) {
          const int iy = (int)(a_mask[i] >> 16) - 64 + dy, ix = (int)(a_mask[i] & 0xFFFFu) - 64 + dx;
          v = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        } else {
          v = (a_mask[i] >> tap) & 1u;
        }
        int off = (dy * p.Win + dx) * p.ldx;
        if (UP) {   // input pixel of output (oy+dy, ox+dx) is ((oy+dy)>>1, (ox+dx)>>1): floor((parity + d) / 2)
          int py = (int)((a_mask[i] >> 16) & 1u), px = (int)((a_mask[i] >> 17) & 1u);
          int qy = ((py + dy + 2) >> 1) - 1, qx = ((px + dx + 2) >> 1) - 1;
          off = (qy * p.Win + qx) * p.ldx;
        }
        a_voff[i] = v ? a_pix[i] + (unsigned)(off * 4) : OOB;
      }
    }
    const int c0b = ld_cc << 7;                                  // 32 floats = 128 bytes per chunk
    const int kb = (tap * p.Cin) * 4 + c0b;
    float* la = As + buf * BM * LDSS + wid * AI * 8 * LDSS;
    float* lb = Bs + buf * BN * LDSS + wid * BI * 8 * LDSS;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(la + i * 8 * LDSS), 16, (int)a_voff[i], c0b, 0, 0);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(lb + i * 8 * LDSS), 16, (int)b_voff[i], kb, 0, 0);
    if (++ld_cc == cchunks) { ld_cc = 0; ++ld_tap; }
  };
  // fragment offsets (floats) inside a 32-row block for g = 0..3: row lr, physical slot (2g + lh) ^ ((lr >> 1) & 7)
  int foff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) foff[g] = lr * LDSS + (((2 * g + lh) ^ ((lr >> 1) & 7)) << 2);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue_stage(0);
  __syncthreads();                 // (the compiler drains the DMA with s_waitcnt vmcnt(0) ahead of the barrier)

  for (int s = 0; s < KT; ++s) {
    const int buf = s & 1;
    if (s + 1 < KT) issue_stage(buf ^ 1);        // buffer buf^1 was last read in step s-1; every wave passed its barrier
    const float* Ab = As + buf * BM * LDSS + wm * MT * 32 * LDSS;
    const float* Bb = Bs + buf * BN * LDSS + wn * NT * 32 * LDSS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDSS + foff[g]);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDSS + foff[g]);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8 (r>>2) + 4 (lane>>5) ----
  // Fast path for interior tiles: no per-element predicates, so the residual loads of a 32x32 block are
  // issued together and waited for once (the predicated form serialises 64 dependent load->store pairs).
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

template <int BM, int BN, int WM, int WN, bool UP>
int launch_igemm_up(IgemmP p, hipStream_t st) {
  static bool attr_set = false;
  constexpr int smem = 2 * (BM + BN) * LDSS * (int)sizeof(float);
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_f32_kernel<BM, BN, WM, WN, UP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  p.tilesN = adm_cdiv(p.N, BN);
  long grid = (long)adm_cdiv(p.M, BM) * p.tilesN;
  hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, UP>), dim3((unsigned)grid, p.splitk > 1 ? p.splitk : 1), dim3(256), smem, st, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(IgemmP p, hipStream_t st) {
  return p.up ? launch_igemm_up<BM, BN, WM, WN, true>(p, st) : launch_igemm_up<BM, BN, WM, WN, false>(p, st);
}

// y[m][n] = sum_z ws[z][m][n] + bias[n] + res[m][n]   (fixed summation order: deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                     const float* __restrict__ res, float* __restrict__ y, long M, int N, int ldy,
                                     int ldr, int splitk) {
  const int N4 = N >> 2;
  const long total = M * N4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / N4;
    const int n = (int)(i - m * N4) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(ws + m * N + n);
    for (int z = 1; z < splitk; ++z) a += *reinterpret_cast<const f32x4*>(ws + ((long)z * M + m) * N + n);
    if (bias) a += *reinterpret_cast<const f32x4*>(bias + n);
    if (res) a += *reinterpret_cast<const f32x4*>(res + m * ldr + n);
    *reinterpret_cast<f32x4*>(y + m * ldy + n) = a;
  }
}

int splitk_plan(long M, int N, int K) {
  // Only the small-M layers (4x4 resolution, the embedding Linears): too few 64x64 tiles to fill 256 CUs.
  const long tiles = ((M + 63) / 64) * ((N + 63) / 64);
  const int KT = K / 32;
  if (tiles > 256 || KT < 16 || (N & 3)) return 1;
  long s = 1024 / tiles;
  const long cap = KT >= 512 ? 32 : 8;        // (K >= 16384: the one GEMM over all blocks' scale/shift gradients, ops.affine_group)
  if (s > cap) s = cap;
  if (s > KT / 8) s = KT / 8;
  return s < 2 ? 1 : (int)s;
}

}  // namespace

extern "C" int adm_conv_splitk(int M, int N, int K) { return splitk_plan(M, N, K); }

extern "C" int adm_conv_fwd_ws(const float* x, const float* wp, const float* bias, const float* res, float* y,
                               float* ws, long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows,
                               int ldy, int ldr, int ks, int up, hipStream_t stream);

namespace {
int conv_fwd_impl(const float* x, const float* wp, const float* bias, const float* res, float* y, int B, int H, int W,
                  int Hin, int Win, int Cin, int ldx, int N, int wrows, int ldy, int ldr, int ks, int up, int stride,
                  int pad_lo, int tile, hipStream_t stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || Hin <= 0 || Win <= 0) return ADM_EINVAL;
  if ((Cin & 31) || (ldx & 3) || ks < 1 || ks > 7 || N <= 0 || wrows < N) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1) || ks > 3)) return ADM_EINVAL;
  if (ks > 3 && (Hin >= 32000 || Win >= 32000)) return ADM_EINVAL;      // packed 16-bit centre coordinates
  if (((uintptr_t)x | (uintptr_t)wp) & 15) return ADM_EINVAL;
  if ((long)B * H * W >= (1L << 31)) return ADM_EINVAL;
  IgemmP p;
  p.x = x; p.w = wp; p.bias = bias; p.res = res; p.y = y;
  p.M = B * H * W; p.N = N; p.H = H; p.W = W;
  p.Hin = Hin; p.Win = Win;
  p.Cin = Cin; p.ldx = ldx; p.K = ks * ks * Cin; p.ldy = ldy; p.ldr = ldr; p.ks = ks; p.up = up; p.wrows = wrows;
  p.tilesN = 0; p.splitk = 1; p.kt_per_split = 0; p.ws = nullptr;
  p.stride = stride; p.cshift = (ks >> 1) - pad_lo;
  const long xb = (long)B * p.Hin * p.Win * ldx * 4, wb = (long)wrows * p.K * 4;
  if (xb >= (1L << 31) || wb >= (1L << 31)) return ADM_EINVAL;     // 32-bit buffer offsets; 0x80000000 must stay out of range
  p.xbytes = (int)xb; p.wbytes = (int)wb;
  if (tile < 0) {
    // Cost model: time ~ rounds * (work of one tile) / (per-tile efficiency), rounds = ceil(tiles / resident slots).
    // Resident workgroups per CU follow from the LDS footprint (2 for the 128-row tiles, 4 for 64x64).
    struct Cand { int id, bm, bn, per_cu; double eff; };
    const Cand cands[4] = {{0, 128, 128, 2, 1.00}, {1, 128, 96, 2, 0.97}, {2, 64, 64, 4, 0.90}, {3, 128, 32, 4, 0.70}};
    double best = 1e300;
    for (const Cand& c : cands) {
      long tiles = (long)adm_cdiv(p.M, c.bm) * adm_cdiv(N, c.bn);
      long slots = 256L * c.per_cu;
      long rounds = (tiles + slots - 1) / slots;
      // measured: a workgroup alone on a CU takes as long as two sharing it (it cannot hide its own
      // barrier stalls), so time goes by whole rounds of resident slots; a round keeps a CU busy for the
      // work of all its resident workgroups
      double t = (double)rounds * c.per_cu * c.bm * c.bn / c.eff;
      if (t < best) { best = t; tile = c.id; }
    }
  }
  switch (tile) {
    case 0: return launch_igemm<128, 128, 2, 2>(p, stream);
    case 1: return launch_igemm<128, 96, 4, 1>(p, stream);
    case 2: return launch_igemm<64, 64, 2, 2>(p, stream);
    case 3: return launch_igemm<128, 32, 4, 1>(p, stream);
    default: return ADM_EINVAL;
  }
}
}  // namespace

extern "C" int adm_conv_fwd(const float* x, const float* wp, const float* bias, const float* res, float* y,
                            int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                            int ks, int up, int tile, hipStream_t stream) {
  return conv_fwd_impl(x, wp, bias, res, y, B, H, W, up ? H / 2 : H, up ? W / 2 : W, Cin, ldx, N, wrows, ldy, ldr, ks,
                       up, 1, ks >> 1, tile, stream);
}

// Strided conv with explicit (possibly asymmetric) zero padding: tap (ky, kx) of output (oy, ox) reads input
// (oy*stride + ky - pad_lo, ox*stride + kx - pad_lo); whatever falls outside [0, Hin) x [0, Win) is zero, which covers
// any bottom/right padding.  The KL autoencoder's Downsample (pad (0,1,0,1) + 3x3 stride 2:
// /root/reference/ddm/encoder_decoder.py:78-96) is stride = 2, pad_lo = 0, Hout = Hin / 2.
extern "C" int adm_conv_fwd_strided(const float* x, const float* wp, const float* bias, const float* res, float* y,
                                    int B, int Hin, int Win, int Hout, int Wout, int Cin, int ldx, int N, int wrows,
                                    int ldy, int ldr, int ks, int stride, int pad_lo, hipStream_t stream) {
  if (stride < 1 || stride > 4 || pad_lo < 0 || pad_lo > (ks >> 1) || Hout <= 0 || Wout <= 0) return ADM_EINVAL;
  if (ks == 2) return ADM_EINVAL;
  if ((long)(Hout - 1) * stride - pad_lo >= Hin || (long)(Wout - 1) * stride - pad_lo >= Win) return ADM_EINVAL;
  return conv_fwd_impl(x, wp, bias, res, y, B, Hout, Wout, Hin, Win, Cin, ldx, N, wrows, ldy, ldr, ks, 0, stride,
                       pad_lo, -1, stream);
}

// Same as adm_conv_fwd with a caller-provided workspace: when adm_conv_splitk(M, N, K) > 1 the K range is split
// over gridDim.y, partial tiles go to ws[splitk][M][N] and a second launch sums them (+ bias, + residual) in a
// fixed order, so the result stays deterministic.  ws_floats >= splitk * M * N.
extern "C" int adm_conv_fwd_ws(const float* x, const float* wp, const float* bias, const float* res, float* y,
                               float* ws, long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows,
                               int ldy, int ldr, int ks, int up, hipStream_t stream) {
  const long M = (long)B * H * W;
  const int K = ks * ks * Cin;
  const int sk = (ws && (Cin & 31) == 0) ? splitk_plan(M, N, K) : 1;
  if (sk <= 1 || ws_floats < (long)sk * M * N)
    return adm_conv_fwd(x, wp, bias, res, y, B, H, W, Cin, ldx, N, wrows, ldy, ldr, ks, up, -1, stream);
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0) return ADM_EINVAL;
  if ((ldx & 3) || (ks != 1 && ks != 3) || N <= 0 || wrows < N || (ldy & 3) || (ldr & 3)) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1))) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)ws | (uintptr_t)y) & 15) return ADM_EINVAL;
  IgemmP p;
  p.x = x; p.w = wp; p.bias = bias; p.res = res; p.y = y;
  p.M = (int)M; p.N = N; p.H = H; p.W = W;
  p.Hin = up ? H / 2 : H; p.Win = up ? W / 2 : W;
  p.Cin = Cin; p.ldx = ldx; p.K = K; p.ldy = ldy; p.ldr = ldr; p.ks = ks; p.up = up; p.wrows = wrows;
  p.tilesN = 0; p.stride = 1; p.cshift = 0;
  const long xb = (long)B * p.Hin * p.Win * ldx * 4, wb = (long)wrows * p.K * 4;
  if (xb >= (1L << 31) || wb >= (1L << 31)) return ADM_EINVAL;
  p.xbytes = (int)xb; p.wbytes = (int)wb;
  const int KT = K / 32;
  p.splitk = sk; p.kt_per_split = (KT + sk - 1) / sk; p.ws = ws;
  p.splitk = (KT + p.kt_per_split - 1) / p.kt_per_split;
  int rc = launch_igemm<64, 64, 2, 2>(p, stream);
  if (rc != ADM_OK) return rc;
  const long total = M * (N / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ws, bias, res, y, M, N, ldy, ldr,
                     p.splitk);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// y[m][n] = sum_z ws[z][m][n] + bias[n] + res[m][n] in a fixed order: shared by the split-K paths of conv_wino2d.hip
int adm_splitk_reduce(const float* ws, const float* bias, const float* res, float* y, long M, int N, int ldy, int ldr, int splitk,
                      hipStream_t stream) {
  const long total = M * (N / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ws, bias, res, y, M, N, ldy, ldr, splitk);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
