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

template <int BM, int BN, int WM, int WN>
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
  const int c4 = tid & 7, r0 = tid >> 3;
  int a_b[AI], a_oy[AI], a_ox[AI];
  bool a_ok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + r0 + 32 * i;
    a_ok[i] = m < p.M;
    int mm = a_ok[i] ? m : 0;
    a_ox[i] = mm % p.W;
    int t = mm / p.W;
    a_oy[i] = t % p.H;
    a_b[i] = t / p.H;
  }
  const int cchunks = p.Cin >> 5;
  const int KT = p.ks * p.ks * cchunks;
  const int pad = p.ks >> 1;

  f32x4 ra[AI], rb[BI];
  auto load_stage = [&](int s) {
    int tap = s / cchunks, c0 = (s - tap * cchunks) << 5;
    int dy = (p.ks == 3) ? tap / 3 - pad : 0, dx = (p.ks == 3) ? tap % 3 - pad : 0;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
      bool v = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      if (p.up) { iy >>= 1; ix >>= 1; }
      const float* ptr = p.x + ((long)(a_b[i] * p.Hin + iy) * p.Win + ix) * p.ldx + c0 + c4 * 4;
      ra[i] = v ? *reinterpret_cast<const f32x4*>(ptr) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = n0 + r0 + 32 * i;
      const float* ptr = p.w + (long)n * p.K + tap * p.Cin + c0 + c4 * 4;
      rb[i] = (n < p.wrows) ? *reinterpret_cast<const f32x4*>(ptr) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_stage = [&](int buf) {
    float* Ab = As + buf * BM * LDSS;
    float* Bb = Bs + buf * BN * LDSS;
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&Ab[(r0 + 32 * i) * LDSS + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bb[(r0 + 32 * i) * LDSS + c4 * 4]) = rb[i];
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

template <int BM, int BN, int WM, int WN>
int launch_igemm(IgemmP p, hipStream_t st) {
  static bool attr_set = false;
  constexpr int smem = 2 * (BM + BN) * LDSS * (int)sizeof(float);
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_f32_kernel<BM, BN, WM, WN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  p.tilesN = adm_cdiv(p.N, BN);
  long grid = (long)adm_cdiv(p.M, BM) * p.tilesN;
  hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN>), dim3((unsigned)grid), dim3(256), smem, st, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
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
  if (tile < 0) {  // heuristic: biggest tile that still gives >= ~1.5 waves of workgroups over 256 CUs
    long m128 = adm_cdiv(p.M, 128), m64 = adm_cdiv(p.M, 64);
    if (N % 128 == 0 && m128 * (N / 128) >= 384) tile = 0;
    else if (N % 96 == 0 && m128 * (N / 96) >= 384) tile = 1;
    else if (N <= 32) tile = 3;
    else if (N % 128 == 0 && m128 * (N / 128) >= 200) tile = 0;
    else if (N % 96 == 0 && m128 * (N / 96) >= 200) tile = 1;
    else tile = 2;
    (void)m64;
  }
  switch (tile) {
    case 0: return launch_igemm<128, 128, 2, 2>(p, stream);
    case 1: return launch_igemm<128, 96, 4, 1>(p, stream);
    case 2: return launch_igemm<64, 64, 2, 2>(p, stream);
    case 3: return launch_igemm<128, 32, 4, 1>(p, stream);
    default: return ADM_EINVAL;
  }
}
