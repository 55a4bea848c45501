// Self-attention forward of UNetBlock (/root/reference/unet/uncond_unet.py:205-208) with the f32 products of both matrix
// products carried on the 16-bit MFMA by the two-term fp16 split of conv_wino2d_x6.hip (FMT 1): s a = h0 + h1 (round to nearest, s a
// power of two from a bound of max |qkv|), a b = (h0 h0' + h0 h1' + h1 h0') / (s s'): three v_mfma_f32_32x32x16_f16 (32 cycles
// for K = 16) in place of eight v_mfma_f32_32x32x2f32 (64 cycles for K = 2) -- 5.3x less matrix-pipe time at f32 accuracy.
//
// Same structure as attention.hip (one workgroup per (batch, head), one wave per 32 queries, S^T = K Q^T with the KEYS on the
// accumulator rows and the query on the lane, softmax in registers, P^T feeding the second product without leaving registers):
//   * K lives in LDS as two fp16 images [term][key][64 (+8)]: a lane reads the 8 consecutive d of its key row with one ds_read_b128
//     = the A operand of a 32x32x16 MFMA; the query fragment (B operand: 8 consecutive d per lane and chunk) is split once into
//     registers;
//   * O^T = V^T P^T needs, per lane, 8 CONSECUTIVE KEYS of one d row: V is stored transposed, [term][d][key'], with the keys of
//     every group of 16 permuted into the order in which the accumulator layout of S^T hands them out (lane half 0 holds rows
//     0-3, 8-11 of a 16-row group, half 1 rows 4-7, 12-15) -- P^T then is the B operand as it stands, no cross-lane move;
//   * P in [0, 1] is split with the fixed scale 2^14.
// L in {32, 64, 128, 256} (the UNet's attention levels) or a multiple of 256 (the latent configs' 32x32 level: 256-key chunks, online
// softmax); anything else stays on attention.hip.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

typedef _Float16 ah_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ah_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned ah_u32x4 __attribute__((ext_vector_type(4)));
constexpr int AH_KROW = 72;        // halves per K row (64 + 8: 144 bytes, 16-byte aligned, conflict-free ds_read_b128)

__device__ inline float ah_scale(float amax) {       // 16000 / max < s <= 32000 / max (= h3_scale of conv_wino2d_x6.hip)
  if (!(amax > 0.f) || !(amax < 3e38f)) return 1.f;
  int e;
  frexpf(16000.f / amax, &e);
  return ldexpf(1.f, e - 1);
}
__device__ __forceinline__ int ah_acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
// position of key k (0..15 of its group) in the permuted order: half = bit 2, then (k >> 3, k & 3)
__device__ __forceinline__ int ah_pos16(int k) { return (((k >> 2) & 1) << 3) | ((k >> 3) << 2) | (k & 3); }
// 8 values -> two fp16 terms, 4 dwords each
__device__ __forceinline__ void ah_split8(const float (&v)[8], float s, ah_u32x4& t0, ah_u32x4& t1) {
  unsigned a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = v[2 * i] * s, x1 = v[2 * i + 1] * s;
    const _Float16 h00 = (_Float16)x0, h01 = (_Float16)x1;
    const _Float16 h10 = (_Float16)(x0 - (float)h00), h11 = (_Float16)(x1 - (float)h01);
    a[i] = __builtin_bit_cast(unsigned, ah_f16x2{h00, h01});
    b[i] = __builtin_bit_cast(unsigned, ah_f16x2{h10, h11});
  }
  t0 = ah_u32x4{a[0], a[1], a[2], a[3]};
  t1 = ah_u32x4{b[0], b[1], b[2], b[3]};
}

// MULTI: L is a multiple of 256; gridDim.y = the 256-query chunk this workgroup owns, the keys pass through LDS 256 at a time with a
// running max / sum per query (online softmax), as attention.hip does for its long sequences
template <int NKT, bool MULTI>
__global__ __launch_bounds__(64 * NKT) void attn_fwd_h3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                float* __restrict__ lse, int L, int heads,
                                                                const float* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smh[];
  constexpr int CH = NKT * 32;
  constexpr int VROW = CH + 8;                       // halves per V^T row
  unsigned short* Kh = smh;                          // [2][CH][AH_KROW]
  unsigned short* Vt = smh + 2 * CH * AH_KROW;       // [2][64][VROW]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long rs = (long)heads * 192;
  const float* base = qkv + (long)b * L * rs + h * 192;
  const float sc = ah_scale(adm_amax_read(amax));
  // ---- the query fragment of this lane: 4 chunks of 16 d, 8 consecutive d per lane half; pre-scaled by 1/8 (exact)
  const int q = (MULTI ? (int)blockIdx.y * CH : 0) + wid * 32 + lr;
  ah_u32x4 q0[4], q1[4];
  {
    const float* qrow = base + (long)q * rs;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(qrow + 16 * c + 8 * lh);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(qrow + 16 * c + 8 * lh + 4);
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      ah_split8(v, sc * 0.125f, q0[c], q1[c]);
    }
  }
  const float inv_s = 1.f / (sc * sc);
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  const int nchunks = MULTI ? L / CH : 1;
#pragma unroll 1
  for (int kc = 0; kc < nchunks; ++kc) {
  const float* kbase = base + (long)kc * CH * rs;
  if (MULTI) __syncthreads();                        // the previous chunk's images are fully consumed
  // ---- K -> two fp16 images, rows as they are
  for (int i = tid; i < CH * 16; i += 64 * NKT) {
    const int key = i >> 4, c4 = i & 15;
    const f32x4 v = *reinterpret_cast<const f32x4*>(kbase + (long)key * rs + 64 + c4 * 4);
    unsigned t0[2], t1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float x0 = v[2 * j] * sc, x1 = v[2 * j + 1] * sc;
      const _Float16 h00 = (_Float16)x0, h01 = (_Float16)x1;
      t0[j] = __builtin_bit_cast(unsigned, ah_f16x2{h00, h01});
      t1[j] = __builtin_bit_cast(unsigned, ah_f16x2{(_Float16)(x0 - (float)h00), (_Float16)(x1 - (float)h01)});
    }
    *reinterpret_cast<uint2*>(Kh + key * AH_KROW + c4 * 4) = make_uint2(t0[0], t0[1]);
    *reinterpret_cast<uint2*>(Kh + (CH + key) * AH_KROW + c4 * 4) = make_uint2(t1[0], t1[1]);
  }
  // ---- V -> transposed, keys permuted within groups of 16; a thread takes two keys that are neighbours in the permuted order
  for (int i = tid; i < (CH / 2) * 16; i += 64 * NKT) {
    const int kp = i >> 4, c4 = i & 15;              // key pair, d quad
    const int grp = kp >> 3, p0 = (kp & 7) * 2;      // positions p0, p0 + 1 of group grp
    // inverse of ah_pos16 for position p: half = p >> 3, i = p & 7 -> key = (i & 3) + 8 (i >> 2) + 4 half
    const int k0 = grp * 16 + ((p0 & 7) & 3) + 8 * ((p0 & 7) >> 2) + 4 * (p0 >> 3);
    const f32x4 va = *reinterpret_cast<const f32x4*>(kbase + (long)k0 * rs + 128 + c4 * 4);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(kbase + (long)(k0 + 1) * rs + 128 + c4 * 4);      // position p0 + 1 = key k0 + 1 (p0 even)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = va[j] * sc, x1 = vb[j] * sc;
      const _Float16 h00 = (_Float16)x0, h01 = (_Float16)x1;
      const int d = c4 * 4 + j;
      *reinterpret_cast<unsigned*>(Vt + d * VROW + grp * 16 + p0) = __builtin_bit_cast(unsigned, ah_f16x2{h00, h01});
      *reinterpret_cast<unsigned*>(Vt + (64 + d) * VROW + grp * 16 + p0) =
          __builtin_bit_cast(unsigned, ah_f16x2{(_Float16)(x0 - (float)h00), (_Float16)(x1 - (float)h01)});
    }
  }
  __syncthreads();
  // ---- S^T tiles: keys on the accumulator rows, this lane's query on the column
  f32x16 s[NKT];
  float mc = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned short* krow = Kh + (kt * 32 + lr) * AH_KROW + 8 * lh;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const ah_f16x8 a0 = __builtin_bit_cast(ah_f16x8, *reinterpret_cast<const ah_u32x4*>(krow + 16 * c));
      const ah_f16x8 a1 = __builtin_bit_cast(ah_f16x8, *reinterpret_cast<const ah_u32x4*>(krow + CH * AH_KROW + 16 * c));
      const ah_f16x8 b0 = __builtin_bit_cast(ah_f16x8, q0[c]), b1 = __builtin_bit_cast(ah_f16x8, q1[c]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[kt][r] = acc[r] * inv_s;
      mc = fmaxf(mc, s[kt][r]);
    }
  }
  mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
  const float mn = fmaxf(m, mc);
  if constexpr (MULTI) {
    const float rsc = (m == -INFINITY) ? 0.f : __expf(m - mn);       // rescale of what has been accumulated so far
    l *= rsc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= rsc; o[1][r] *= rsc; }
  }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(s[kt][r] - mn);
      s[kt][r] = e;
      l += e;
    }
  m = mn;
  // ---- O^T = V^T P^T: d on the accumulator rows, the query on the lane
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const float pv[8] = {s[kt][8 * sl], s[kt][8 * sl + 1], s[kt][8 * sl + 2], s[kt][8 * sl + 3],
                           s[kt][8 * sl + 4], s[kt][8 * sl + 5], s[kt][8 * sl + 6], s[kt][8 * sl + 7]};
      ah_u32x4 p0, p1;
      ah_split8(pv, 16384.f, p0, p1);
      const ah_f16x8 b0 = __builtin_bit_cast(ah_f16x8, p0), b1 = __builtin_bit_cast(ah_f16x8, p1);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const unsigned short* vrow = Vt + (db * 32 + lr) * VROW + kt * 32 + 16 * sl + 8 * lh;
        const ah_f16x8 a0 = __builtin_bit_cast(ah_f16x8, *reinterpret_cast<const ah_u32x4*>(vrow));
        const ah_f16x8 a1 = __builtin_bit_cast(ah_f16x8, *reinterpret_cast<const ah_u32x4*>(vrow + 64 * VROW));
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, o[db], 0, 0, 0);
      }
    }
  }
  const float mc = m;
  l += __shfl_xor(l, 32, 64);                          // the two wave halves hold disjoint keys of the same query
  // out^T accumulators (d on rows, the query on the lane) -> 16-byte pieces of the query's row
  const float oscale = 1.f / (l * sc * 16384.f);
  float* orow = out + ((long)b * L + q) * heads * 64 + h * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 v = {o[db][4 * q4] * oscale, o[db][4 * q4 + 1] * oscale, o[db][4 * q4 + 2] * oscale, o[db][4 * q4 + 3] * oscale};
      *reinterpret_cast<f32x4*>(orow + db * 32 + 8 * q4 + 4 * lh) = v;
    }
  if (lh == 0 && lse) lse[((long)b * heads + h) * L + q] = mc + __logf(l);
}

template <int NKT, bool MULTI>
int launch_fwd_h3(const float* qkv, float* out, float* lse, const float* amax, int B, int L, int heads, hipStream_t st) {
  constexpr int CH = NKT * 32;
  constexpr int smem = (2 * CH * AH_KROW + 2 * 64 * (CH + 8)) * (int)sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_h3_kernel<NKT, MULTI>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_fwd_h3_kernel<NKT, MULTI>), dim3(B * heads, MULTI ? L / CH : 1), dim3(64 * NKT), smem, st, qkv, out, lse, L, heads, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward on the same format.  Both kernels keep attention.hip's orientations (queries on lanes for dQ, keys on lanes for dK / dV: no
// score tile is transposed or summed with atomics); what changes is where the operands of the K = 16 MFMAs come from:
//   * products whose reduction index is d (S, dP): one side is a ROW image [term][row][64] in LDS, the other a register fragment;
//   * products whose reduction index is the accumulator-row index of a score tile (dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS): the
//     score-side operand is the tile as it stands, the other side a TRANSPOSED image [term][d][row'] with the rows of every group of
//     16 in the accumulator layout's order (ah_pos16), as V^T in the forward.
// LDS holds 128 rows of the other side at a time (three / four images: 108 / 144 KB); dS gets a static scale from the bound
// |dS| <= |dP| + |delta| <= 128 max|dO| max|qkv| (loose by orders of magnitude: costs nothing but the smallest values' last bits).
__device__ __forceinline__ f32x16 ah_mfma3(const ah_u32x4 a0, const ah_u32x4 a1, const ah_u32x4 b0, const ah_u32x4 b1, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ah_f16x8, a0), __builtin_bit_cast(ah_f16x8, b1), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ah_f16x8, a1), __builtin_bit_cast(ah_f16x8, b0), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ah_f16x8, a0), __builtin_bit_cast(ah_f16x8, b0), c, 0, 0, 0);
}
// dst[term][row][AH_KROW] <- two-term split of scale * src[row][0..63] (rows at a stride of `rs` floats), CH rows
__device__ __forceinline__ void ah_fill_rows(unsigned short* dst, int CH, const float* src, long rs, float scale, int tid, int nthreads) {
  for (int i = tid; i < CH * 16; i += nthreads) {
    const int row = i >> 4, c4 = i & 15;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)row * rs + c4 * 4);
    unsigned t0[2], t1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float x0 = v[2 * j] * scale, x1 = v[2 * j + 1] * scale;
      const _Float16 h00 = (_Float16)x0, h01 = (_Float16)x1;
      t0[j] = __builtin_bit_cast(unsigned, ah_f16x2{h00, h01});
      t1[j] = __builtin_bit_cast(unsigned, ah_f16x2{(_Float16)(x0 - (float)h00), (_Float16)(x1 - (float)h01)});
    }
    *reinterpret_cast<uint2*>(dst + row * AH_KROW + c4 * 4) = make_uint2(t0[0], t0[1]);
    *reinterpret_cast<uint2*>(dst + (CH + row) * AH_KROW + c4 * 4) = make_uint2(t1[0], t1[1]);
  }
}
// both images of one tensor from ONE pass over its rows (the dkv kernel needs Q and dO in both layouts, the dq kernel K)
__device__ __forceinline__ void ah_fill_both(unsigned short* rimg, unsigned short* timg, int CH, const float* src, long rs, float scale,
                                             int tid, int nthreads) {
  const int VROW = CH + 8;
  for (int i = tid; i < (CH / 2) * 16; i += nthreads) {
    const int kp = i >> 4, c4 = i & 15;
    const int grp = kp >> 3, p0 = (kp & 7) * 2;
    const int r0 = grp * 16 + ((p0 & 7) & 3) + 8 * ((p0 & 7) >> 2) + 4 * (p0 >> 3);
    const f32x4 va = *reinterpret_cast<const f32x4*>(src + (long)r0 * rs + c4 * 4);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(src + (long)(r0 + 1) * rs + c4 * 4);
    _Float16 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = va[j] * scale, x1 = vb[j] * scale;
      a0[j] = (_Float16)x0; a1[j] = (_Float16)(x0 - (float)a0[j]);
      b0[j] = (_Float16)x1; b1[j] = (_Float16)(x1 - (float)b0[j]);
      const int d = c4 * 4 + j;
      *reinterpret_cast<unsigned*>(timg + d * VROW + grp * 16 + p0) = __builtin_bit_cast(unsigned, ah_f16x2{a0[j], b0[j]});
      *reinterpret_cast<unsigned*>(timg + (64 + d) * VROW + grp * 16 + p0) = __builtin_bit_cast(unsigned, ah_f16x2{a1[j], b1[j]});
    }
    *reinterpret_cast<uint2*>(rimg + r0 * AH_KROW + c4 * 4) =
        make_uint2(__builtin_bit_cast(unsigned, ah_f16x2{a0[0], a0[1]}), __builtin_bit_cast(unsigned, ah_f16x2{a0[2], a0[3]}));
    *reinterpret_cast<uint2*>(rimg + (CH + r0) * AH_KROW + c4 * 4) =
        make_uint2(__builtin_bit_cast(unsigned, ah_f16x2{a1[0], a1[1]}), __builtin_bit_cast(unsigned, ah_f16x2{a1[2], a1[3]}));
    *reinterpret_cast<uint2*>(rimg + (r0 + 1) * AH_KROW + c4 * 4) =
        make_uint2(__builtin_bit_cast(unsigned, ah_f16x2{b0[0], b0[1]}), __builtin_bit_cast(unsigned, ah_f16x2{b0[2], b0[3]}));
    *reinterpret_cast<uint2*>(rimg + (CH + r0 + 1) * AH_KROW + c4 * 4) =
        make_uint2(__builtin_bit_cast(unsigned, ah_f16x2{b1[0], b1[1]}), __builtin_bit_cast(unsigned, ah_f16x2{b1[2], b1[3]}));
  }
}
// a lane's fragment of its own row: 4 chunks of 16 d, 8 consecutive d per lane half
__device__ __forceinline__ void ah_row_frag(const float* row, int lh, float scale, ah_u32x4 (&t0)[4], ah_u32x4 (&t1)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + 16 * c + 8 * lh);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + 16 * c + 8 * lh + 4);
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    ah_split8(v, scale, t0[c], t1[c]);
  }
}
// 32 x 32 tile: rows of a row image (A operand, reduction over d) times the lane's fragment
__device__ __forceinline__ f32x16 ah_rows_times_frag(const unsigned short* img, int CH, int row0, const ah_u32x4 (&f0)[4], const ah_u32x4 (&f1)[4],
                                                     int lr, int lh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const unsigned short* rp = img + (row0 + lr) * AH_KROW + 8 * lh;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    acc = ah_mfma3(*reinterpret_cast<const ah_u32x4*>(rp + 16 * c), *reinterpret_cast<const ah_u32x4*>(rp + CH * AH_KROW + 16 * c), f0[c], f1[c], acc);
  return acc;
}
// o^T[d][lane] += sum over the tile's 32 rows of timg[d][row] * t[row] (t in the accumulator layout, scaled by `scale` for the split)
__device__ __forceinline__ void ah_accum_T(f32x16 (&o)[2], const unsigned short* timg, int CH, int row0, const f32x16& t, float scale, int lr, int lh) {
  const int VROW = CH + 8;
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const float tv[8] = {t[8 * sl], t[8 * sl + 1], t[8 * sl + 2], t[8 * sl + 3], t[8 * sl + 4], t[8 * sl + 5], t[8 * sl + 6], t[8 * sl + 7]};
    ah_u32x4 b0, b1;
    ah_split8(tv, scale, b0, b1);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const unsigned short* vp = timg + (db * 32 + lr) * VROW + row0 + 16 * sl + 8 * lh;
      o[db] = ah_mfma3(*reinterpret_cast<const ah_u32x4*>(vp), *reinterpret_cast<const ah_u32x4*>(vp + 64 * VROW), b0, b1, o[db]);
    }
  }
}
__device__ __forceinline__ void ah_store_T(const f32x16 (&o)[2], float* rowptr, int lh, float scale, float& am) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 v = {o[db][4 * q4] * scale, o[db][4 * q4 + 1] * scale, o[db][4 * q4 + 2] * scale, o[db][4 * q4 + 3] * scale};
      am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      *reinterpret_cast<f32x4*>(rowptr + db * 32 + 8 * q4 + 4 * lh) = v;
    }
}

// backward, part 1: queries on lanes.  dQ^T = K^T dS^T / 8, delta[q] = sum_d dO O
template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_h3_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                   const float* __restrict__ dout, const float* __restrict__ lse,
                                                                   float* __restrict__ dqkv, float* __restrict__ delta, int L, int heads,
                                                                   const float* __restrict__ amax_qkv, const float* __restrict__ amax_g,
                                                                   float* __restrict__ amax_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smh[];
  constexpr int CH = NW >= 4 ? 128 : 32 * NW;        // keys in LDS at a time
  unsigned short* Kh = smh;                          // [2][CH][AH_KROW]
  unsigned short* Vh = Kh + 2 * CH * AH_KROW;
  unsigned short* Kt = Vh + 2 * CH * AH_KROW;        // [2][64][CH + 8]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long rs = (long)heads * 192, ro = (long)heads * 64;
  const float* base = qkv + (long)b * L * rs + h * 192;
  const float aq = adm_amax_read(amax_qkv), ag = adm_amax_read(amax_g);
  const float sc = ah_scale(aq), sg = ah_scale(ag), sd = ah_scale(128.f * aq * ag);
  const int q = (int)blockIdx.y * 32 * NW + wid * 32 + lr;      // (gridDim.y = L / 256 for the long sequences)
  ah_u32x4 q0[4], q1[4], g0[4], g1[4];
  ah_row_frag(base + (long)q * rs, lh, sc * 0.125f, q0, q1);
  const float* grow = dout + ((long)b * L + q) * ro + h * 64;
  const float* orow = out + ((long)b * L + q) * ro + h * 64;
  ah_row_frag(grow, lh, sg, g0, g1);
  float dl = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(grow + 16 * c + 8 * lh + 4 * hh);
      const f32x4 ov = *reinterpret_cast<const f32x4*>(orow + 16 * c + 8 * lh + 4 * hh);
      dl += gv[0] * ov[0] + gv[1] * ov[1] + gv[2] * ov[2] + gv[3] * ov[3];
    }
  dl += __shfl_xor(dl, 32, 64);
  const float ls = lse[((long)b * heads + h) * L + q];
  if (lh == 0) delta[((long)b * heads + h) * L + q] = dl;
  const float inv_s = 1.f / (sc * sc), inv_g = 1.f / (sc * sg);
  f32x16 dq[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
#pragma unroll 1
  for (int k0 = 0; k0 < L; k0 += CH) {
    __syncthreads();
    ah_fill_both(Kh, Kt, CH, base + (long)k0 * rs + 64, rs, sc, tid, 64 * NW);
    ah_fill_rows(Vh, CH, base + (long)k0 * rs + 128, rs, sc, tid, 64 * NW);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < CH / 32; ++kt) {
      f32x16 s = ah_rows_times_frag(Kh, CH, kt * 32, q0, q1, lr, lh);
      const f32x16 dp = ah_rows_times_frag(Vh, CH, kt * 32, g0, g1, lr, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __expf(s[r] * inv_s - ls) * (dp[r] * inv_g - dl);      // dS^T[key][q]
      ah_accum_T(dq, Kt, CH, kt * 32, s, sd, lr, lh);
    }
  }
  float am = 0.f;
  ah_store_T(dq, dqkv + ((long)b * L + q) * rs + h * 192, lh, 0.125f / (sc * sd), am);
  adm_amax_commit(am, amax_out);
}

// backward, part 2: keys on lanes.  dV^T = dO^T P, dK^T = Q^T dS / 8
template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_h3_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    float* __restrict__ dqkv, int L, int heads,
                                                                    const float* __restrict__ amax_qkv, const float* __restrict__ amax_g,
                                                                    float* __restrict__ amax_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smh[];
  constexpr int CH = NW >= 4 ? 128 : 32 * NW;        // queries in LDS at a time
  unsigned short* Qh = smh;                          // [2][CH][AH_KROW]
  unsigned short* Gh = Qh + 2 * CH * AH_KROW;        // dO rows
  unsigned short* Qt = Gh + 2 * CH * AH_KROW;        // [2][64][CH + 8]
  unsigned short* Gt = Qt + 2 * 64 * (CH + 8);
  float* Ls = reinterpret_cast<float*>(Gt + 2 * 64 * (CH + 8));      // lse[CH] | delta[CH]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long rs = (long)heads * 192, ro = (long)heads * 64;
  const float* base = qkv + (long)b * L * rs + h * 192;
  const float aq = adm_amax_read(amax_qkv), ag = adm_amax_read(amax_g);
  const float sc = ah_scale(aq), sg = ah_scale(ag), sd = ah_scale(128.f * aq * ag);
  const int key = (int)blockIdx.y * 32 * NW + wid * 32 + lr;
  ah_u32x4 k0f[4], k1f[4], v0f[4], v1f[4];
  ah_row_frag(base + (long)key * rs + 64, lh, sc * 0.125f, k0f, k1f);
  ah_row_frag(base + (long)key * rs + 128, lh, sc, v0f, v1f);
  const float inv_s = 1.f / (sc * sc), inv_g = 1.f / (sg * sc);
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
#pragma unroll 1
  for (int q0 = 0; q0 < L; q0 += CH) {
    __syncthreads();
    const float* gsrc = dout + ((long)b * L + q0) * ro + h * 64;
    ah_fill_both(Qh, Qt, CH, base + (long)q0 * rs, rs, sc, tid, 64 * NW);
    ah_fill_both(Gh, Gt, CH, gsrc, ro, sg, tid, 64 * NW);
    for (int i = tid; i < CH; i += 64 * NW) {
      Ls[i] = lse[((long)b * heads + h) * L + q0 + i];
      Ls[CH + i] = delta[((long)b * heads + h) * L + q0 + i];
    }
    __syncthreads();
#pragma unroll 1
    for (int qt = 0; qt < CH / 32; ++qt) {
      f32x16 s = ah_rows_times_frag(Qh, CH, qt * 32, k0f, k1f, lr, lh);       // S[q][key] (already / 8)
      const f32x16 dp = ah_rows_times_frag(Gh, CH, qt * 32, v0f, v1f, lr, lh);   // dP[q][key]
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qt * 32 + ah_acc_row(r, lh);
        const float p = __expf(s[r] * inv_s - Ls[qq]);
        s[r] = p;
        ds[r] = p * (dp[r] * inv_g - Ls[CH + qq]);
      }
      ah_accum_T(dv, Gt, CH, qt * 32, s, 16384.f, lr, lh);
      ah_accum_T(dk, Qt, CH, qt * 32, ds, sd, lr, lh);
    }
  }
  float am = 0.f;
  float* orow = dqkv + ((long)b * L + key) * rs + h * 192;
  ah_store_T(dk, orow + 64, lh, 0.125f / (sc * sd), am);
  ah_store_T(dv, orow + 128, lh, 1.f / (sg * 16384.f), am);
  adm_amax_commit(am, amax_out);
}

template <int NW>
int launch_bwd_h3(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta, const float* amax_qkv,
                  const float* amax_g, float* amax_out, int B, int L, int heads, hipStream_t st) {
  constexpr int CH = NW >= 4 ? 128 : 32 * NW;
  constexpr int smem_dq = (4 * CH * AH_KROW + 2 * 64 * (CH + 8)) * (int)sizeof(unsigned short);
  constexpr int smem_dkv = (4 * CH * AH_KROW + 4 * 64 * (CH + 8)) * (int)sizeof(unsigned short) + 2 * CH * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_h3_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_dq) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_h3_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_dkv) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  const dim3 grid(B * heads, L / (32 * NW));
  hipLaunchKernelGGL((attn_bwd_dq_h3_kernel<NW>), grid, dim3(64 * NW), smem_dq, st, qkv, out, dout, lse, dqkv, delta, L, heads,
                     amax_qkv, amax_g, amax_out);
  hipLaunchKernelGGL((attn_bwd_dkv_h3_kernel<NW>), grid, dim3(64 * NW), smem_dkv, st, qkv, dout, lse, delta, dqkv, L, heads,
                     amax_qkv, amax_g, amax_out);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

}  // namespace

// adm_attn_bwd on the fp16 split format: amax_qkv / amax_dout = bound vectors of |qkv| and |dout|; amax_dqkv (may be NULL) = bound vector
// raised to max |dqkv|.  L as for adm_attn_fwd_h3; delta [B*heads][L] scratch as in adm_attn_bwd.
extern "C" int adm_attn_bwd_h3(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta,
                               const float* amax_qkv, const float* amax_dout, float* amax_dqkv, int B, int L, int heads, hipStream_t stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || !delta || !amax_qkv || !amax_dout || B <= 0 || heads <= 0) return ADM_EINVAL;
  if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout) & 15) return ADM_EINVAL;
  switch (L) {
    case 32: return launch_bwd_h3<1>(qkv, out, dout, lse, dqkv, delta, amax_qkv, amax_dout, amax_dqkv, B, L, heads, stream);
    case 64: return launch_bwd_h3<2>(qkv, out, dout, lse, dqkv, delta, amax_qkv, amax_dout, amax_dqkv, B, L, heads, stream);
    case 128: return launch_bwd_h3<4>(qkv, out, dout, lse, dqkv, delta, amax_qkv, amax_dout, amax_dqkv, B, L, heads, stream);
    case 256: return launch_bwd_h3<8>(qkv, out, dout, lse, dqkv, delta, amax_qkv, amax_dout, amax_dqkv, B, L, heads, stream);
    default:
      if (L > 256 && L % 256 == 0 && L <= 16384)
        return launch_bwd_h3<8>(qkv, out, dout, lse, dqkv, delta, amax_qkv, amax_dout, amax_dqkv, B, L, heads, stream);
      return ADM_EINVAL;
  }
}

// adm_attn_fwd on the fp16 split format: amax = bound vector (include/adm_hip.h) of |qkv| (the qkv conv's epilogue wrote it).  L in
// {32, 64, 128, 256} only (ADM_EINVAL otherwise: the caller stays on adm_attn_fwd); same outputs to f32 rounding.
extern "C" int adm_attn_fwd_h3(const float* qkv, float* out, float* lse, const float* amax, int B, int L, int heads, hipStream_t stream) {
  if (!qkv || !out || !amax || B <= 0 || heads <= 0 || ((uintptr_t)qkv & 15)) return ADM_EINVAL;
  switch (L) {
    case 32: return launch_fwd_h3<1, false>(qkv, out, lse, amax, B, L, heads, stream);
    case 64: return launch_fwd_h3<2, false>(qkv, out, lse, amax, B, L, heads, stream);
    case 128: return launch_fwd_h3<4, false>(qkv, out, lse, amax, B, L, heads, stream);
    case 256: return launch_fwd_h3<8, false>(qkv, out, lse, amax, B, L, heads, stream);
    default:
      if (L > 256 && L % 256 == 0 && L <= 16384) return launch_fwd_h3<8, true>(qkv, out, lse, amax, B, L, heads, stream);
      return ADM_EINVAL;
  }
}
