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
// L <= 256, L % 32 == 0 (the UNet's 8x8 / 16x16 attention levels); anything else stays on attention.hip.
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

template <int NKT>
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
  // ---- K -> two fp16 images, rows as they are
  for (int i = tid; i < CH * 16; i += 64 * NKT) {
    const int key = i >> 4, c4 = i & 15;
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)key * rs + 64 + c4 * 4);
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
    const f32x4 va = *reinterpret_cast<const f32x4*>(base + (long)k0 * rs + 128 + c4 * 4);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(base + (long)(k0 + 1) * rs + 128 + c4 * 4);      // position p0 + 1 = key k0 + 1 (p0 even)
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
  // ---- the query fragment of this lane: 4 chunks of 16 d, 8 consecutive d per lane half; pre-scaled by 1/8 (exact)
  const int q = wid * 32 + lr;
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
  __syncthreads();
  // ---- S^T tiles: keys on the accumulator rows, this lane's query on the column
  const float inv_s = 1.f / (sc * sc);
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
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(s[kt][r] - mc);
      s[kt][r] = e;
      l += e;
    }
  // ---- O^T = V^T P^T: d on the accumulator rows, the query on the lane
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
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

template <int NKT>
int launch_fwd_h3(const float* qkv, float* out, float* lse, const float* amax, int B, int L, int heads, hipStream_t st) {
  constexpr int CH = NKT * 32;
  constexpr int smem = (2 * CH * AH_KROW + 2 * 64 * (CH + 8)) * (int)sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_h3_kernel<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_fwd_h3_kernel<NKT>), dim3(B * heads), dim3(64 * NKT), smem, st, qkv, out, lse, L, heads, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

}  // namespace

// adm_attn_fwd on the fp16 split format: amax = bound vector (include/adm_hip.h) of |qkv| (the qkv conv's epilogue wrote it).  L in
// {32, 64, 128, 256} only (ADM_EINVAL otherwise: the caller stays on adm_attn_fwd); same outputs to f32 rounding.
extern "C" int adm_attn_fwd_h3(const float* qkv, float* out, float* lse, const float* amax, int B, int L, int heads, hipStream_t stream) {
  if (!qkv || !out || !amax || B <= 0 || heads <= 0 || ((uintptr_t)qkv & 15)) return ADM_EINVAL;
  switch (L) {
    case 32: return launch_fwd_h3<1>(qkv, out, lse, amax, B, L, heads, stream);
    case 64: return launch_fwd_h3<2>(qkv, out, lse, amax, B, L, heads, stream);
    case 128: return launch_fwd_h3<4>(qkv, out, lse, amax, B, L, heads, stream);
    case 256: return launch_fwd_h3<8>(qkv, out, lse, amax, B, L, heads, stream);
    default: return ADM_EINVAL;
  }
}
