// 3x3 stride-1 convolution (forward and data gradient): 2-D Winograd F(2x2, 3x3) with the f32 products carried on the bf16 MFMA.
//
// gfx950 has no reduced-precision fast path for f32 matrix operands: v_mfma_f32_32x32x2_f32 runs at the f32 vector rate, 1/16 of
// the bf16 MFMA.  An f32 value splits EXACTLY into three bf16 terms, a = a0 + a1 + a2 (8 + 8 + 8 mantissa bits, same exponent
// range, by truncation: a0 = top 16 bits of a, a1 = top 16 bits of a - a0, a2 = a - a0 - a1), and bf16 x bf16 products are exact in
// the f32 accumulator, so
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0) + O(2^-24 |a b|)
// -- six v_mfma_f32_32x32x16_bf16 (6 x 32 cycles for K = 16) replace eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles): 2.67x less
// matrix-pipe time at f32 accuracy.  The five small products are summed in their own accumulator chain (started from C = 0) and
// added to the running sum once per K step with one f32 add: the accumulator sees ONE matrix add per 16 channels (not six), which
// measures BELOW the f32 MFMA's own rounding error against an fp64 reference on both zero-mean and all-positive data
// (tools/bf16x6_probe.hip; tests/test_hip_ops.py::test_conv_x6_error_vs_fp64).
//
// Everything else is conv_wino2d.hip's algorithm (same transforms, same pass structure: ey a loop in time, four ex accumulator
// tiles per wave, output rows folded in registers).  What changes is the balance -- the matrix pipe is no longer the bound, the data
// path is -- and with it the shape:
//   * workgroup = 512 threads = 8 waves (4 x 2), 128 tiles x 64 couts, ONE per CU: the split operands take 6 bytes per element
//     (A 48 KB + B 24 KB per stage, double buffered = 144 KB of the 160 KB LDS);
//   * A: thread = (tile, 16-byte channel quad): 2 rows x 4 pixels raw buffer loads (range check = zero padding), y combination and
//     B^T along x in f32 exactly as before, then the three-term split (and / sub / and / sub per element, v_perm packing) and
//     twelve ds_write_b64 into the [ex plane][term] images.  Rows are 32 bytes (16 bf16 channels): both the 8-byte writes and the
//     16-byte fragment reads of a wave cover contiguous LDS, conflict-free without a swizzle;
//   * B: weights are split once per optimiser step at pack time (adm_split3_bf16) into Wq6[ey][ex][term][n][cin] and go straight
//     to LDS by LDS-DMA (24 one-KB wave instructions per stage, three per wave).
// Replaces F.conv2d of Conv2d.forward and its autograd data gradient (/root/reference/unet/uncond_unet.py:98-110).
#include "common.h"
#include "../../include/adm_hip.h"

#ifndef X6_ACC2
#define X6_ACC2 1     // 1: small products in their own chain (see above); 0: all six into the running accumulator
#endif

namespace {

struct X6P {
  const float* x; const unsigned short* w; const float* bias; const float* res; float* y;
  int Mt, N, H, W, Hh, Wh, Cin, ldx, ldy, ldr, wrows, tilesN, xbytes, wbytes, plane;   // plane = wrows * Cin (elements of one [term] image)
  int splitk, chunks_per_split; float* ws;
};

typedef __attribute__((address_space(3))) void x6_lds_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int X6P_T = 128, X6N = 64, X6K = 16;     // tiles x couts x K step
constexpr int X6_A_STAGE = 4 * 3 * X6P_T * X6K;    // bf16 elements per A stage
constexpr int X6_B_STAGE = 4 * 3 * X6N * X6K;

// v = v0 + v1 + v2 exactly, each term a bf16 (returned as the packed top halves of four lanes' worth: two dwords per term)
__device__ __forceinline__ void split3_pack(const f32x4 v, u32x2& t0, u32x2& t1, u32x2& t2) {
  unsigned u[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i] = __float_as_uint(v[i]);
    const float r = v[i] - __uint_as_float(u[i] & 0xFFFF0000u);
    m[i] = __float_as_uint(r);
    const float r2 = r - __uint_as_float(m[i] & 0xFFFF0000u);
    l[i] = __float_as_uint(r2);
  }
  t0 = u32x2{__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u)};
  t1 = u32x2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
  t2 = u32x2{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
}

__global__ __launch_bounds__(512) void wino2d_x6_kernel(X6P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem6[];
  unsigned short* As = smem6;                      // [2][4 ex][3 terms][X6P_T][X6K]
  unsigned short* Bs = smem6 + 2 * X6_A_STAGE;     // [2][4 ex][3 terms][X6N][X6K]
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
  const int mt0 = tm * X6P_T, n0 = tn * X6N;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.wbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // ---- A loader: this thread owns tile `pl` and channel quad `aq` of every stage
  const int pl = tid >> 2, aq = tid & 3;
  unsigned a_base = 0;            // byte offset of pixel (b, 2ty, 2xp), channel quad aq
  unsigned colmask = 0;           // bit j: column 2xp - 1 + j is inside the image
  unsigned rowmask = 0;           // bit i: row 2ty - 1 + i is inside the image
  {
    const int t = mt0 + pl;
    if (t < p.Mt) {
      const int xp = t % p.Wh;
      const int u = t / p.Wh;
      const int ty = u % p.Hh, b = u / p.Hh;
      a_base = (unsigned)((((long)b * p.H + 2 * ty) * p.W + 2 * xp) * p.ldx + aq * 4) * 4u;
      colmask = (xp > 0 ? 1u : 0u) | 6u | (2 * xp + 2 < p.W ? 8u : 0u);
      rowmask = (ty > 0 ? 1u : 0u) | 6u | (2 * ty + 2 < p.H ? 8u : 0u);
    }
  }
  unsigned a_voff[2][4];          // [row A / row B of the current pass][pixel j]
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) a_voff[r][j] = OOB;

  // ---- B loader (LDS-DMA): 24 one-KB instructions per stage = (ex, term) image pt x 32-row half; wave w issues q = 3w .. 3w+2.
  // Lane l of an instruction covers row (q & 1) * 32 + (l >> 1), 16-byte half (l & 1) of the 32-byte row.
  unsigned b_voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = wid * 3 + i, pt = q >> 1;
    const int row = (q & 1) * 32 + (lane >> 1);
    const int n = n0 + row;
    b_voff[i] = (n < p.wrows) ? (unsigned)(((long)pt * p.plane + (long)n * p.Cin + (lane & 1) * 8) * 2) : OOB;
  }

  const int c_begin = (p.splitk > 1) ? (int)blockIdx.y * p.chunks_per_split : 0;
  const int chunks = (p.splitk > 1) ? min(p.chunks_per_split, (p.Cin >> 4) - c_begin) : (p.Cin >> 4);     // 16-channel chunks
  const int KT = 4 * chunks;                      // four passes (ey) over this workgroup's K range
  if (p.splitk > 1) {
    p.y = p.ws + (long)blockIdx.y * ((long)p.Mt * 4) * p.N;
    p.ldy = p.N; p.bias = nullptr; p.res = nullptr;
  }
  int ld_ey = 0, ld_cc = 0;
  f32x4 dA[4], dB[4];
  int st_ey = 0;                                  // pass of the stage being LOADED (consumed by store_stage: selects the signs)
  auto issue_stage = [&](int buf) {               // global -> registers (A), global -> LDS (B) for the NEXT stage
    if (ld_cc == 0) {
      // pass ey combines input rows (iA, iB) of the 4-row patch: 0: +r0 -r2   1: +r1 +r2   2: -r1 +r2   3: +r1 -r3
      const int iA = (ld_ey == 0) ? 0 : 1, iB = (ld_ey == 3) ? 3 : 2;
      const bool vA = (rowmask >> iA) & 1u, vB = (rowmask >> iB) & 1u;
      const int offA = (iA - 1) * p.W * p.ldx * 4, offB = (iB - 1) * p.W * p.ldx * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cv = (colmask >> j) & 1u;
        a_voff[0][j] = (vA && cv) ? a_base + (unsigned)(offA + (j - 1) * p.ldx * 4) : OOB;
        a_voff[1][j] = (vB && cv) ? a_base + (unsigned)(offB + (j - 1) * p.ldx * 4) : OOB;
      }
    }
    const int cidx = c_begin + ld_cc;
    const int soff = cidx << 6;                   // 16 floats = 64 bytes per chunk
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dA[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[0][j], soff, 0));
      dB[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[1][j], soff, 0));
    }
    st_ey = ld_ey;
    const int kb = (ld_ey * 12 * p.plane) * 2 + (cidx << 5);       // (ey) block of twelve [ex][term] images; 16 bf16 = 32 bytes per chunk
    unsigned short* lb = Bs + buf * X6_B_STAGE + (wid * 3) * 512;  // 512 elements = one KB per instruction
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (x6_lds_void*)(lb + i * 512), 16, (int)b_voff[i], kb, 0, 0);
    if (++ld_cc == chunks) { ld_cc = 0; ++ld_ey; }
  };
  auto store_stage = [&](int buf) {               // y combination, B^T along x (f32), three-term split, into the [ex][term] images
    f32x4 e[4];
    if (st_ey == 1) {                             // wave-uniform: one add / sub per element instead of a multiply-add pair
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = dA[j] + dB[j];
    } else if (st_ey == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = dB[j] - dA[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = dA[j] - dB[j];
    }
    unsigned short* la = As + buf * X6_A_STAGE + pl * X6K + aq * 4;
    const f32x4 v[4] = {e[0] - e[2], e[1] + e[2], e[2] - e[1], e[1] - e[3]};
#pragma unroll
    for (int ex = 0; ex < 4; ++ex) {
      u32x2 t0, t1, t2;
      split3_pack(v[ex], t0, t1, t2);
      *reinterpret_cast<u32x2*>(la + (ex * 3 + 0) * X6P_T * X6K) = t0;
      *reinterpret_cast<u32x2*>(la + (ex * 3 + 1) * X6P_T * X6K) = t1;
      *reinterpret_cast<u32x2*>(la + (ex * 3 + 2) * X6P_T * X6K) = t2;
    }
  };

  f32x16 acc[4];
  f32x16 Y[2][2];                                 // [output row][output column of the pair]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[a][b][r] = 0.f;
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

  const int a_foff = (wm * 32 + lr) * X6K + lh * 8;       // fragment: row = tile, 8 bf16 = 16 bytes at k = 8 (lane >> 5)
  const int b_foff = (wn * 32 + lr) * X6K + lh * 8;

  issue_stage(0);
  store_stage(0);
  __syncthreads();
  int cc = 0, ey = 0;                             // (chunk, pass) of the stage being COMPUTED
  for (int s = 0; s < KT; ++s) {
    const int buf = s & 1;
    if (s + 1 < KT) issue_stage(buf ^ 1);
    const unsigned short* Ab = As + buf * X6_A_STAGE + a_foff;
    const unsigned short* Bb = Bs + buf * X6_B_STAGE + b_foff;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      bf16x8 a[3], b[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab + (xi * 3 + t) * X6P_T * X6K));
        b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bb + (xi * 3 + t) * X6N * X6K));
      }
#if X6_ACC2
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], zero, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[xi], 0, 0, 0);
      acc[xi] += c;
#else
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[xi], 0, 0, 0);
#endif
    }
    if (s + 1 < KT) store_stage(buf ^ 1);         // the other buffer was last read in stage s-1: every wave passed its barrier
    if (++cc == chunks) {
      // end of pass ey: A^T along x, then fold into the output rows (A^T along y: row 0 = Z0 + Z1 + Z2, row 1 = Z1 - Z2 - Z3)
      cc = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m1 = acc[1][r], m2 = acc[2][r];
        const float z0 = acc[0][r] + m1 + m2, z1 = m1 - m2 - acc[3][r];
        if (ey <= 2) { Y[0][0][r] += z0; Y[0][1][r] += z1; }
        if (ey == 1) { Y[1][0][r] += z0; Y[1][1][r] += z1; }
        if (ey >= 2) { Y[1][0][r] -= z0; Y[1][1][r] -= z1; }
      }
#pragma unroll
      for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
      ++ey;
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D layout col = lane&31 (cout), row = (r&3) + 8 (r>>2) + 4 (lane>>5) (tile)
  const int n = n0 + wn * 32 + lr;
  if (n >= p.N) return;
  const float bv = p.bias ? p.bias[n] : 0.f;
  const int tb = mt0 + wm * 32 + 4 * lh;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = tb + (r & 3) + 8 * (r >> 2);
    if (t >= p.Mt) continue;
    const int xp = t % p.Wh;
    const int u = t / p.Wh;                        // = b * Hh + ty
    const long px0 = ((long)u * 2) * p.W + 2 * xp; // pixel (b, 2ty, 2xp) in units of pixels: (b*H + 2ty) * W + 2xp
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const long px = px0 + (long)a * p.W;
      float y0 = Y[a][0][r] + bv, y1 = Y[a][1][r] + bv;
      if (p.res) { y0 += p.res[px * p.ldr + n]; y1 += p.res[(px + 1) * p.ldr + n]; }
      p.y[px * p.ldy + n] = y0;
      p.y[(px + 1) * p.ldy + n] = y1;
    }
  }
}

// dst[img][term][i] = bf16 term `term` of src[img][i]   (img = Winograd plane; i over rows x cin)
__global__ void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long per_img, int imgs) {
  const long total = per_img * imgs;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long img = i / per_img, k = i - img * per_img;
    const float a = src[i];
    const unsigned u = __float_as_uint(a);
    const float r = a - __uint_as_float(u & 0xFFFF0000u);
    const unsigned m = __float_as_uint(r);
    const float r2 = r - __uint_as_float(m & 0xFFFF0000u);
    unsigned short* d = dst + img * 3 * per_img + k;
    d[0] = (unsigned short)(u >> 16);
    d[per_img] = (unsigned short)(m >> 16);
    d[2 * per_img] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
}

}  // namespace

int adm_splitk_reduce(const float* ws, const float* bias, const float* res, float* y, long M, int N, int ldy, int ldr, int splitk,
                      hipStream_t stream);       // conv_igemm.hip

// dst[imgs][3][per_img] (bf16 bit patterns) <- the exact three-term split of src[imgs][per_img] (f32)
extern "C" int adm_split3_bf16(const float* src, void* dst, long per_img, int imgs, hipStream_t stream) {
  if (!src || !dst || per_img <= 0 || imgs <= 0) return ADM_EINVAL;
  const long total = per_img * imgs;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, stream, src, static_cast<unsigned short*>(dst), per_img, imgs);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// Same contract as adm_conv_fwd_wino2d, with wq6 = adm_split3_bf16 of the adm_pack_weight_wino2d operand (16 planes of
// wrows x Cin): [16][3][wrows][Cin] bf16.
extern "C" int adm_conv_fwd_wino2d_x6(const float* x, const void* wq6, const float* bias, const float* res, float* y, float* ws,
                                      long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                                      hipStream_t stream) {
  if (!x || !wq6 || !y || B <= 0 || H < 2 || W < 2 || (W & 1) || (H & 1)) return ADM_EINVAL;
  if ((Cin & 15) || (ldx & 3) || N <= 0 || wrows < N) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)wq6) & 15) return ADM_EINVAL;
  X6P p;
  p.x = x; p.w = static_cast<const unsigned short*>(wq6); p.bias = bias; p.res = res; p.y = y;
  const long Mt = (long)B * (H / 2) * (W / 2);
  const long xb = (long)B * H * W * ldx * 4, wb = 48L * wrows * Cin * 2;
  if (Mt >= (1L << 30) || xb >= (1L << 31) || wb >= (1L << 31)) return ADM_EINVAL;
  p.Mt = (int)Mt; p.N = N; p.H = H; p.W = W; p.Hh = H / 2; p.Wh = W / 2; p.Cin = Cin; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr;
  p.wrows = wrows; p.xbytes = (int)xb; p.wbytes = (int)wb; p.plane = wrows * Cin;
  p.tilesN = adm_cdiv(N, X6N);
  p.splitk = 1; p.chunks_per_split = 0; p.ws = nullptr;
  const int sk = (ws && !(N & 3) && !(ldy & 3) && (!res || !(ldr & 3))) ? adm_wino2d_x6_splitk(B, H, W, Cin, N) : 1;
  if (sk > 1 && ws_floats >= (long)sk * Mt * 4 * N) {
    const int chunks = Cin >> 4;
    p.chunks_per_split = (chunks + sk - 1) / sk;
    p.splitk = (chunks + p.chunks_per_split - 1) / p.chunks_per_split;
    p.ws = ws;
  }
  constexpr int smem = 2 * (X6_A_STAGE + X6_B_STAGE) * (int)sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_x6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
        hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  const long grid = (long)adm_cdiv(Mt, X6P_T) * p.tilesN;
  hipLaunchKernelGGL(wino2d_x6_kernel, dim3((unsigned)grid, p.splitk), dim3(512), smem, stream, p);
  ADM_CHECK_LAUNCH();
  if (p.splitk > 1) return adm_splitk_reduce(ws, bias, res, y, Mt * 4, N, ldy, ldr, p.splitk, stream);
  return ADM_OK;
}

// Split count over the input channels for launches with fewer workgroups than CUs (one workgroup per CU here)
extern "C" int adm_wino2d_x6_splitk(int B, int H, int W, int Cin, int N) {
  const long wgs = (long)adm_cdiv((long)B * (H / 2) * (W / 2), X6P_T) * adm_cdiv(N, X6N);
  const int chunks = Cin >> 4;
  if (wgs >= 192 || chunks < 8) return 1;
  int s = (int)(256 / wgs);
  if (s > chunks / 4) s = chunks / 4;
  if (s > 4) s = 4;
  return s < 2 ? 1 : s;
}
