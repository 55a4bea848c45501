// 3x3 stride-1 convolution (forward and data gradient) with a 1-D Winograd F(2,3) transform along x, on the fp32 MFMA.
//
// The direct implicit GEMM (conv_igemm.hip) spends 9 multiply-adds per (pixel, cin, cout).  F(2,3) computes two
// horizontally adjacent outputs from four inputs with 4 multiplies instead of 6:
//     d0..d3 = x[.., 2xp-1 .. 2xp+2, c]                       (per filter row ky)
//     v0 = d0 - d2   v1 = d1 + d2   v2 = d2 - d1   v3 = d1 - d3          (B^T d; only +-1)
//     u0 = g0        u1 = (g0+g1+g2)/2   u2 = (g0-g1+g2)/2   u3 = g2      (G g; done at weight-pack time)
//     m_xi = sum_{ky, c} v_xi * u_xi                                       (four GEMMs over K' = 3 Cin)
//     y[2xp] = m0 + m1 + m2          y[2xp+1] = m1 - m2 - m3               (A^T m; epilogue)
// i.e. 6 multiply-adds per (pixel, cin, cout): 1.5x less MFMA work on the layers that hold ~95% of the FLOPs.  The
// transform constants are +-1 (inputs, outputs) and 1/2 (weights), so the fp32 error stays at the level of the direct
// kernel's own rounding (tests: rtol 1e-3 against the oracle like every other kernel, observed ~1e-6).
//
// GEMM view per xi: M' = B*H*W/2 pixel PAIRS, N = Cout, K' = (ky, cin).  One workgroup (4 waves, 2x2) owns 64 pairs x
// 64 couts for ALL four xi (4 accumulator tiles per wave = 64 AGPRs, like the direct kernel); K-step 16.  The weight
// operand is double-buffered (LDS-DMA overlaps the MFMAs), the register-staged pixel operand is SINGLE-buffered with a
// second barrier per stage: (4 x 64 x 16 + 2 x 4 x 64 x 16) floats = 48 KB of LDS -> three workgroups per CU (the
// register limit) instead of two, which is worth +3 % (the extra wave per SIMD fills the barrier and transform gaps).
//   A: thread = (pair, 16-byte channel quad): four raw buffer loads d0..d3 (range check = zero padding), four float4
//      add/sub, four ds_write_b128 (rows of 64 B, slots XOR-swizzled with (row >> 2) & 3: conflict-free fragment reads)
//   B: transformed weights Wq[xi][n][ky][cin] straight to LDS by LDS-DMA (same swizzle on the source address)
// The transform costs 16 vector-ALU instructions per thread per stage against 32 MFMAs per wave.
// Used for every 3x3 stride-1 conv / data gradient with an even width and M >= 8192 (smaller maps keep the direct
// kernel with its deterministic split-K); the fused nearest-x2, strided and 1x1 cases stay on conv_igemm.hip.
// Replaces F.conv2d of Conv2d.forward and its autograd data gradient (/root/reference/unet/uncond_unet.py:98-110).
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

struct WinoP {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int Mp, N, H, W, Wh, Cin, ldx, ldy, ldr, wrows, tilesN, xbytes, wbytes, plane;   // plane = floats per xi plane of w
};

typedef __attribute__((address_space(3))) void wino_lds_void;
constexpr int WP = 64, WN_ = 64, WK = 16;         // pairs x couts x K-step

template <bool UP>
__global__ __launch_bounds__(256) void igemm_wino_kernel(WinoP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                               // [4][WP][WK]  (single buffer: see the loop)
  float* Bs = smem + 4 * WP * WK;                 // [2][4][WN_][WK]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap, m-fastest inside an n-tile (see conv_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int mp0 = tm * WP, n0 = tn * WN_;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.wbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // ---- A loader: this thread owns pair `pl` and channel quad `aq` of every stage
  const int pl = tid >> 2, aq = tid & 3;
  unsigned a_base = 0;            // byte offset of pixel (b, y, 2xp), channel quad aq
  unsigned colmask = 0;           // bit i: d_i's column 2xp-1+i is inside the image
  unsigned rowmask = 0;           // bit ky: row y+ky-1 is inside the image
  int ypar = 0;                   // UP: parity of y (selects the input row of each filter row)
  {
    const int pr = mp0 + pl;
    if (pr < p.Mp) {
      const int xp = pr % p.Wh;
      const int t = pr / p.Wh;
      const int y = t % p.H;
      const int x0 = 2 * xp;
      if (UP) {   // fused nearest x2: (y, x0) are coordinates of the up-sampled grid; the input is [B][H/2][W/2]
        const int b = t / p.H;
        a_base = (unsigned)((((long)b * (p.H >> 1) + (y >> 1)) * p.Wh + xp) * p.ldx + aq * 4) * 4u;
        ypar = y & 1;
      } else {
        a_base = (unsigned)(((long)t * p.W + x0) * p.ldx + aq * 4) * 4u;     // t = b*H + y
      }
      colmask = (x0 > 0 ? 1u : 0u) | 6u | (x0 + 2 < p.W ? 8u : 0u);
      rowmask = (y > 0 ? 1u : 0u) | 2u | (y + 1 < p.H ? 4u : 0u);
    }
  }
  const int a_slot = (aq ^ ((pl >> 2) & 3)) << 2;       // swizzled float offset inside the 16-float row
  unsigned a_voff[4] = {OOB, OOB, OOB, OOB};

  // ---- B loader (LDS-DMA): wave w streams plane xi = w; instruction i covers rows i*16 + (lane >> 2), slot lane & 3
  unsigned b_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + (lane >> 2);
    const int ls = (lane & 3) ^ ((row >> 2) & 3);
    const int n = n0 + row;
    b_voff[i] = (n < p.wrows) ? (unsigned)(wid * p.plane + n * 3 * p.Cin + ls * 4) * 4u : OOB;
  }

  const int chunks = p.Cin >> 4;                  // 16-channel chunks per filter row
  const int KT = 3 * chunks;
  int ld_ky = 0, ld_cc = 0;
  f32x4 d[4];
  auto issue_stage = [&](int buf) {               // global -> registers (A), global -> LDS (B) for the NEXT stage
    if (ld_cc == 0) {
      const bool rv = (rowmask >> ld_ky) & 1u;
      if (UP) {
        // input row of up-sampled row y+ky-1 is (y>>1) + floor((ypar + ky - 1) / 2); columns 2xp-1..2xp+2 map to
        // xp-1, xp, xp, xp+1 (d1 == d2: the xi = 2 component B^T d vanishes identically)
        const int dr = ((ypar + ld_ky + 1) >> 1) - 1;
        const int rowoff = dr * p.Wh * p.ldx * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int dc = (i == 0) ? -1 : (i == 3) ? 1 : 0;
          a_voff[i] = (rv && ((colmask >> i) & 1u)) ? a_base + (unsigned)(rowoff + dc * p.ldx * 4) : OOB;
        }
      } else {
        const int rowoff = (ld_ky - 1) * p.W * p.ldx * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a_voff[i] = (rv && ((colmask >> i) & 1u)) ? a_base + (unsigned)(rowoff + (i - 1) * p.ldx * 4) : OOB;
      }
    }
    const int soff = ld_cc << 6;                  // 16 floats = 64 bytes per chunk
#pragma unroll
    for (int i = 0; i < 4; ++i)
      d[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[i], soff, 0));
    const int kb = (ld_ky * p.Cin) * 4 + soff;
    float* lb = Bs + (buf * 4 + wid) * WN_ * WK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (wino_lds_void*)(lb + i * 16 * WK), 16, (int)b_voff[i], kb, 0, 0);
    if (++ld_cc == chunks) { ld_cc = 0; ++ld_ky; }
  };
  auto store_stage = [&](int buf) {               // B^T d, into the four xi planes
    float* la = As + pl * WK + a_slot;
    *reinterpret_cast<f32x4*>(la + 0 * WP * WK) = d[0] - d[2];
    *reinterpret_cast<f32x4*>(la + 1 * WP * WK) = d[1] + d[2];
    *reinterpret_cast<f32x4*>(la + 2 * WP * WK) = d[2] - d[1];
    *reinterpret_cast<f32x4*>(la + 3 * WP * WK) = d[1] - d[3];
  };

  f32x16 acc[4];
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

  // fragment offsets (floats) for the two 8-k groups of a stage: row lr of the wave's 32, physical slot (2g + lh) ^ swz
  int foff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) foff[g] = lr * WK + (((2 * g + lh) ^ ((lr >> 2) & 3)) << 2);

  issue_stage(0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < KT; ++s) {
    const int buf = s & 1;
    if (s + 1 < KT) issue_stage(buf ^ 1);
    const float* Ab = As + wm * 32 * WK;
    const float* Bb = Bs + buf * 4 * WN_ * WK + wn * 32 * WK;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        a[xi] = *reinterpret_cast<const f32x4*>(Ab + xi * WP * WK + foff[g]);
        b[xi] = *reinterpret_cast<const f32x4*>(Bb + xi * WN_ * WK + foff[g]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[xi][k], b[xi][k], acc[xi], 0, 0, 0);
    }
    // A is single-buffered (48 KB of LDS per workgroup -> THREE workgroups per CU instead of two): one barrier after the
    // reads of this stage, one after the next stage's stores (which also covers the B operand's DMA)
    __syncthreads();
    if (s + 1 < KT) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: A^T m.  C/D layout col = lane&31 (cout), row = (r&3) + 8 (r>>2) + 4 (lane>>5) (pair)
  const int n = n0 + wn * 32 + lr;
  if (n >= p.N) return;
  const float bv = p.bias ? p.bias[n] : 0.f;
  const int prb = mp0 + wm * 32 + 4 * lh;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pr = prb + (r & 3) + 8 * (r >> 2);
    if (pr >= p.Mp) continue;
    const float m1 = acc[1][r], m2 = acc[2][r];
    float y0 = acc[0][r] + m1 + m2 + bv;
    float y1 = m1 - m2 - acc[3][r] + bv;
    const long px = 2L * pr;
    if (p.res) { y0 += p.res[px * p.ldr + n]; y1 += p.res[(px + 1) * p.ldr + n]; }
    p.y[px * p.ldy + n] = y0;
    p.y[(px + 1) * p.ldy + n] = y1;
  }
}

// G g for both operand layouts, from the reference's OIHW parameter:
//   wf[xi][co][ky][ci]            (forward B operand)
//   wb[xi][ci][ky'][co]           (data-gradient B operand: taps flipped in y and x, channels transposed)
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int Co, int Ci,
                                 int Co_pad, int Ci_pad) {
  const long total = (long)Co_pad * Ci_pad * 3;
  const long planef = (long)Co_pad * 3 * Ci_pad, planeb = (long)Ci_pad * 3 * Co_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci_pad);
    const long t = i / Ci_pad;
    const int ky = (int)(t % 3), co = (int)(t / 3);
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (co < Co && ci < Ci) {
      const float* g = w + (((long)co * Ci + ci) * 3 + ky) * 3;
      g0 = g[0]; g1 = g[1]; g2 = g[2];
    }
    if (wf) {
      const long o = ((long)co * 3 + ky) * Ci_pad + ci;
      wf[o] = g0;
      wf[o + planef] = (g0 + g1 + g2) * 0.5f;
      wf[o + 2 * planef] = (g0 - g1 + g2) * 0.5f;
      wf[o + 3 * planef] = g2;
    }
    if (wb) {   // g'(ky', kx') = w(2-ky', 2-kx'): row ky' = 2-ky, taps reversed
      const long o = ((long)ci * 3 + (2 - ky)) * Co_pad + co;
      wb[o] = g2;
      wb[o + planeb] = (g2 + g1 + g0) * 0.5f;
      wb[o + 2 * planeb] = (g2 - g1 + g0) * 0.5f;
      wb[o + 3 * planeb] = g0;
    }
  }
}

}  // namespace

extern "C" int adm_pack_weight_wino(const float* w, float* wf, float* wb, int Co, int Ci, int Co_pad, int Ci_pad,
                                    hipStream_t stream) {
  if (!w || (!wf && !wb) || Co <= 0 || Ci <= 0 || Co_pad < Co || Ci_pad < Ci) return ADM_EINVAL;
  const long total = (long)Co_pad * Ci_pad * 3;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_wino_kernel, dim3(grid), dim3(256), 0, stream, w, wf, wb, Co, Ci, Co_pad, Ci_pad);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

namespace {
int conv_wino_impl(const float* x, const float* wq, const float* bias, const float* res, float* y, int B, int H, int W,
                   int Cin, int ldx, int N, int wrows, int ldy, int ldr, int up, hipStream_t stream) {
  if (!x || !wq || !y || B <= 0 || H <= 0 || W < 2 || (W & 1)) return ADM_EINVAL;
  if (up && (H & 1)) return ADM_EINVAL;
  if ((Cin & 15) || (ldx & 3) || N <= 0 || wrows < N) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)wq) & 15) return ADM_EINVAL;
  WinoP p;
  p.x = x; p.w = wq; p.bias = bias; p.res = res; p.y = y;
  const long Mp = (long)B * H * (W / 2);
  const long xb = (up ? (long)B * (H / 2) * (W / 2) : (long)B * H * W) * ldx * 4, wb = 4L * wrows * 3 * Cin * 4;
  if (Mp >= (1L << 30) || xb >= (1L << 31) || wb >= (1L << 31)) return ADM_EINVAL;
  p.Mp = (int)Mp; p.N = N; p.H = H; p.W = W; p.Wh = W / 2; p.Cin = Cin; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr;
  p.wrows = wrows; p.xbytes = (int)xb; p.wbytes = (int)wb; p.plane = wrows * 3 * Cin;
  p.tilesN = adm_cdiv(N, WN_);
  constexpr int smem = (4 * WP + 2 * 4 * WN_) * WK * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  const long grid = (long)adm_cdiv(Mp, WP) * p.tilesN;
  if (up) hipLaunchKernelGGL(igemm_wino_kernel<true>, dim3((unsigned)grid), dim3(256), smem, stream, p);
  else hipLaunchKernelGGL(igemm_wino_kernel<false>, dim3((unsigned)grid), dim3(256), smem, stream, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
}  // namespace

extern "C" int adm_conv_fwd_wino(const float* x, const float* wq, const float* bias, const float* res, float* y, int B,
                                 int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr, hipStream_t stream) {
  return conv_wino_impl(x, wq, bias, res, y, B, H, W, Cin, ldx, N, wrows, ldy, ldr, 0, stream);
}

// Same with the reference's `up` resampling fused: x is [B][H/2][W/2][ldx], nearest-x2 up-sampled on the fly to the H x W
// grid the conv runs on (Conv2d.forward with up=True, uncond_unet.py:98-104; H, W = OUTPUT size, both even).
extern "C" int adm_conv_fwd_wino_up(const float* x, const float* wq, const float* bias, const float* res, float* y, int B,
                                    int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr, hipStream_t stream) {
  return conv_wino_impl(x, wq, bias, res, y, B, H, W, Cin, ldx, N, wrows, ldy, ldr, 1, stream);
}
