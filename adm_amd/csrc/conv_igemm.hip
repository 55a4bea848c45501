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
// Tiling: 256 threads = 4 waves; workgroup tile BM x BN, K-step 32; A and B tiles staged through
// LDS as [row][32+4] floats (the +4 pad makes the ds_read_b128 fragment reads conflict-free:
// 16-lane groups hit 16 distinct 4-bank slots), register-staged double buffering with one barrier
// per K-step.  Each lane reads float4 = 4 consecutive k of its row; lanes 0-31 take k 0-3 of an
// 8-group and lanes 32-63 take k 4-7, so MFMA j of the group consumes k = j (low half) and
// k = 4 + j (high half) -- a fixed permutation of the k order applied to A and B alike.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

struct IgemmP {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int M, N, H, W, Hin, Win, Cin, ldx, K, ldy, ldr, ks, up, wrows, tilesN;
};

constexpr int LDSS = 36;   // floats per LDS row (32 + 4 pad)

template <int BM, int BN, int WM, int WN, bool UP>
__global__ __launch_bounds__(256) void igemm_f32_kernel(IgemmP p) {
  constexpr int MT = BM / (WM * 32), NT = BN / (WN * 32);
  constexpr int AI = BM / 32, BI = BN / 32;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM][LDSS]
  float* Bs = smem + 2 * BM * LDSS;       // [2][BN][LDSS]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int tn = blockIdx.x % p.tilesN, tm = blockIdx.x / p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- loader state: each thread owns float4 column c4 of rows r0 + 32 i ----
  // Per row: a base pointer at tap (0,0) and a 9-bit mask of the taps that fall inside the image, both
  // computed once; per K-step only `mask >> tap`, one add and the load remain.  Out-of-image taps read a
  // valid dummy address (the row's centre pixel) and are zeroed with a select: no divergent branches.
  const int c4 = tid & 7, r0 = tid >> 3;
  const float* a_base[AI];
  unsigned a_mask[AI];
  const int pad = p.ks >> 1;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + r0 + 32 * i;
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
    a_mask[i] = mask;
    // pointer to input pixel (oy, ox) [or (oy/2, ox/2) for the fused nearest-x2], channel quad c4
    int py = UP ? (oy >> 1) : oy, px = UP ? (ox >> 1) : ox;
    a_base[i] = p.x + ((long)(b * p.Hin + py) * p.Win + px) * p.ldx + c4 * 4;
    // for `up`, the tap offset depends on the parity of (oy, ox); keep those in the mask's high bits
    if (UP) a_mask[i] |= ((unsigned)(oy & 1) << 16) | ((unsigned)(ox & 1) << 17);
  }
  const int cchunks = p.Cin >> 5;
  const int KT = p.ks * p.ks * cchunks;
  const float* b_base[BI];
  bool b_ok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = n0 + r0 + 32 * i;
    b_ok[i] = n < p.wrows;
    b_base[i] = p.w + (long)(b_ok[i] ? n : 0) * p.K + c4 * 4;
  }

  f32x4 ra[AI], rb[BI];
  int ld_tap = 0, ld_cc = 0;       // (tap, channel chunk) of the NEXT stage to load: no division per stage
  int st_tap = 0;                  // tap of the stage held in ra/rb (for the zero-padding select at store time)
  // Loads are issued raw (always from a valid address); the out-of-image select is applied only when the
  // registers are written to LDS, AFTER the MFMAs of the current stage, so the loads stay in flight.
  auto load_stage = [&]() {
    const int tap = ld_tap, c0 = ld_cc << 5;
    int dy = 0, dx = 0;
    if (p.ks == 3) { dy = tap / 3 - pad; dx = tap - (tap / 3) * 3 - pad; }
    const long tap_off = ((long)dy * p.Win + dx) * p.ldx + c0;      // wave-uniform (non-up case)
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool v = (a_mask[i] >> tap) & 1u;
      long off = tap_off;
      if (UP) {     // input pixel of output (oy+dy, ox+dx) is ((oy+dy)>>1, (ox+dx)>>1)
        int py = (int)((a_mask[i] >> 16) & 1u), px = (int)((a_mask[i] >> 17) & 1u);
        int qy = ((py + dy + 2) >> 1) - 1, qx = ((px + dx + 2) >> 1) - 1;     // floor((parity + d) / 2)
        off = ((long)qy * p.Win + qx) * p.ldx + c0;
      }
      ra[i] = *reinterpret_cast<const f32x4*>(a_base[i] + (v ? off : (long)c0));
    }
    const int koff = tap * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_base[i] + koff);
    st_tap = tap;
    if (++ld_cc == cchunks) { ld_cc = 0; ++ld_tap; }
  };
  auto store_stage = [&](int buf) {
    float* Ab = As + buf * BM * LDSS;
    float* Bb = Bs + buf * BN * LDSS;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < AI; ++i)
      *reinterpret_cast<f32x4*>(&Ab[(r0 + 32 * i) * LDSS + c4 * 4]) = ((a_mask[i] >> st_tap) & 1u) ? ra[i] : zero;
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bb[(r0 + 32 * i) * LDSS + c4 * 4]) = b_ok[i] ? rb[i] : zero;
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
    const float* Ab = As + buf * BM * LDSS + (wm * MT * 32 + lr) * LDSS + lh * 4;
    const float* Bb = Bs + buf * BN * LDSS + (wn * NT * 32 + lr) * LDSS + lh * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDSS + g * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDSS + g * 8);
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
  hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, UP>), dim3((unsigned)grid), dim3(256), smem, st, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(IgemmP p, hipStream_t st) {
  return p.up ? launch_igemm_up<BM, BN, WM, WN, true>(p, st) : launch_igemm_up<BM, BN, WM, WN, false>(p, st);
}

}  // namespace

extern "C" int adm_conv_fwd(const float* x, const float* wp, const float* bias, const float* res, float* y,
                            int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                            int ks, int up, int tile, hipStream_t stream) {
  if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0) return ADM_EINVAL;
  if ((Cin & 31) || (ldx & 3) || (ks != 1 && ks != 3) || N <= 0 || wrows < N) return ADM_EINVAL;
  if (up && ((H & 1) || (W & 1))) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)wp) & 15) return ADM_EINVAL;
  IgemmP p;
  p.x = x; p.w = wp; p.bias = bias; p.res = res; p.y = y;
  p.M = B * H * W; p.N = N; p.H = H; p.W = W;
  p.Hin = up ? H / 2 : H; p.Win = up ? W / 2 : W;
  p.Cin = Cin; p.ldx = ldx; p.K = ks * ks * Cin; p.ldy = ldy; p.ldr = ldr; p.ks = ks; p.up = up; p.wrows = wrows;
  p.tilesN = 0;
  if (tile < 0) {
    // Cost model: time ~ rounds * (work of one tile) / (per-tile efficiency), rounds = ceil(tiles / resident slots).
    // Resident workgroups per CU follow from the LDS footprint (2 for the 128-row tiles, 4 for 64x64).
    struct Cand { int id, bm, bn, per_cu; double eff; };
    const Cand cands[4] = {{0, 128, 128, 2, 1.00}, {1, 128, 96, 2, 0.97}, {2, 64, 64, 4, 0.80}, {3, 128, 32, 4, 0.70}};
    double best = 1e300;
    for (const Cand& c : cands) {
      long tiles = (long)adm_cdiv(p.M, c.bm) * adm_cdiv(N, c.bn);
      long slots = 256L * c.per_cu;
      long rounds = (tiles + slots - 1) / slots;
      // measured: a workgroup alone on a CU takes as long as two sharing it (it cannot hide its own
      // barrier stalls), so time goes by whole rounds of resident slots
      double t = (double)rounds * c.bm * c.bn / c.eff;
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
