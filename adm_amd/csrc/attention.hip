// Self-attention core of UNetBlock (/root/reference/unet/uncond_unet.py:205-208):
//   w = softmax_k(q^T k / sqrt(64)),  a = w v          per (batch, head), head dim 64, L = H*W (<= 256 in one
//   chunk; longer sequences, e.g. L = 1024 of the 64x64-latent configs, in 256-row chunks with an online softmax)
// on the fp32-input MFMA (exact fp32, so the rtol 1e-3 / atol 1e-4 parity bar holds without any
// reduced-precision caveat).
//
// Layout: qkv [B][L][heads*192] with each head's 192 channels packed as (q[64] | k[64] | v[64])
// -- the pack kernel permutes the reference's (c, {q,k,v}) interleave once, at weight-pack time.
//
// Structure ("score tile with the reduction index on the accumulator rows"):
//   * one workgroup per (b, head); K and V of that head live in LDS ([L][64+4] floats each; the +4
//     pad makes the 16-byte row reads conflict-free), one wave per 32 queries;
//   * S^T = K Q^T is computed with KEYS on the accumulator rows and the QUERY on the lane, so a
//     lane owns one query: the softmax max / sum run over registers plus ONE cross-half shuffle
//     (no LDS, no 32-step butterfly), and P^T is already laid out as the B operand of the next
//     product O^T = V^T P^T (sum over keys = accumulator-row index), so P never leaves registers;
//   * Q is pre-scaled by 1/8 (a power of two: bit-identical to the reference's k/sqrt(64)).
// The backward kernels reuse the same structure in both orientations (queries on lanes for dQ,
// keys on lanes for dK/dV) so that no score tile is ever transposed or summed with atomics.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

constexpr int DH = 64;          // head dim
constexpr int KS = DH + 4;      // LDS row stride (floats)

// acc[r] holds row (r&3) + 8 (r>>2) + 4 (lane>>5) of a 32x32 tile, column lane&31
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// rows[32 x 64] (LDS, stride KS) times per-lane fragment (4 consecutive d per 8-group):
// out[row][lane] = sum_d rows[row][d] * frag(lane)[d]
__device__ __forceinline__ f32x16 rows_times_frag(const float* rows_lds, const f32x4 (&frag)[8], int lr, int lh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* base = rows_lds + lr * KS + lh * 4;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    f32x4 a = *reinterpret_cast<const f32x4*>(base + g * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], frag[g][k], acc, 0, 0, 0);
  }
  return acc;
}

// out^T[dblk*32 + row][lane] += sum_{r} cols_lds[tile_row0 + acc_row(r)][dblk*32 + lane] * p[r]
// (A operand = transposed LDS rows read with lane = d: conflict-free ds_read_b32)
__device__ __forceinline__ void accum_T(f32x16 (&o)[2], const float* lds_rows, int tile_row0, const f32x16& p, int lr,
                                        int lh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* row = lds_rows + (tile_row0 + acc_row(r, lh)) * KS + lr;
    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[0], p[r], o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32], p[r], o[1], 0, 0, 0);
  }
}

__device__ __forceinline__ void load_rows_to_lds(float* lds, const float* g, long row_stride, int L, int tid,
                                                 int nthreads) {
  for (int i = tid; i < L * 16; i += nthreads) {
    int row = i >> 4, c4 = i & 15;
    *reinterpret_cast<f32x4*>(lds + row * KS + c4 * 4) = *reinterpret_cast<const f32x4*>(g + row * row_stride + c4 * 4);
  }
}

// per-lane fragment of row `row` (or zeros): 8 groups x float4 at d = 8g + 4 lh
__device__ __forceinline__ void load_frag(f32x4 (&f)[8], const float* rowptr, bool valid, int lh, float scale) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    f32x4 v = valid ? *reinterpret_cast<const f32x4*>(rowptr + g * 8 + lh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f[g] = v * scale;
  }
}

// store out^T accumulators (d on rows, owner row on lane) as 16-byte pieces of row `rowptr`
__device__ __forceinline__ void store_T(const f32x16 (&o)[2], float* rowptr, bool valid, int lh, float scale) {
  if (!valid) return;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      f32x4 v = {o[db][4 * q4] * scale, o[db][4 * q4 + 1] * scale, o[db][4 * q4 + 2] * scale, o[db][4 * q4 + 3] * scale};
      *reinterpret_cast<f32x4*>(rowptr + db * 32 + 8 * q4 + 4 * lh) = v;
    }
}

// max |o * scale| of the accumulators a lane stores with store_T (0 for lanes that store nothing)
__device__ __forceinline__ float amax_T(const f32x16 (&o)[2], bool valid, float scale) {
  float am = 0.f;
  if (valid) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(o[db][r] * scale));
  }
  return am;
}

// Sequences longer than one chunk (CH = NKT*32 keys, at most 256) are processed flash-style: gridDim.y = number of
// 256-row chunks the workgroup OWNS one of (queries for fwd / dq, keys for dkv) and it loops over all chunks of the
// other side, re-filling LDS each time; the forward keeps a running max / sum per query (online softmax).
template <int NKT, bool MULTI>
__global__ __launch_bounds__(64 * (NKT > 8 ? 8 : NKT)) void attn_fwd_kernel(const float* __restrict__ qkv,
                                                                           float* __restrict__ out,
                                                                           float* __restrict__ lse, int L, int heads) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CH = NKT * 32;
  float* Ks = smem;
  float* Vs = smem + CH * KS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long rs = (long)heads * 192;
  const float* base = qkv + (long)b * L * rs + h * 192;
  const int nchunks = MULTI ? (L + CH - 1) / CH : 1;     // single-chunk instantiation: no rescale code, no spills
  const int q = blockIdx.y * CH + wid * 32 + lr;
  const bool qok = q < L;
  f32x4 qf[8];
  load_frag(qf, base + (long)q * rs, qok, lh, 0.125f);

  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
#pragma unroll 1
  for (int kc = 0; kc < nchunks; ++kc) {
    const int k0 = kc * CH, kn = min(CH, L - k0);
    __syncthreads();                                   // previous chunk fully consumed
    if (kn < CH) {                                     // zero-fill so masked keys read finite values
      for (int i = tid; i < CH * KS * 2; i += blockDim.x) smem[i] = 0.f;
      __syncthreads();
    }
    load_rows_to_lds(Ks, base + (long)k0 * rs + 64, rs, kn, tid, blockDim.x);
    load_rows_to_lds(Vs, base + (long)k0 * rs + 128, rs, kn, tid, blockDim.x);
    __syncthreads();
    f32x16 s[NKT];
    float mc = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = rows_times_frag(Ks + kt * 32 * KS, qf, lr, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kt * 32 + acc_row(r, lh) >= kn) s[kt][r] = -INFINITY;
        mc = fmaxf(mc, s[kt][r]);
      }
    }
    mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
    const float mn = fmaxf(m, mc);
    if constexpr (MULTI) {
      const float sc = (m == -INFINITY) ? 0.f : __expf(m - mn);    // rescale of what has been accumulated so far
      l *= sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= sc; o[1][r] *= sc; }
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = __expf(s[kt][r] - mn);
        s[kt][r] = e;
        l += e;
      }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) accum_T(o, Vs, kt * 32, s[kt], lr, lh);
    m = mn;
  }
  l += __shfl_xor(l, 32, 64);                          // the two wave halves hold disjoint keys of the same query
  store_T(o, out + ((long)b * L + q) * heads * 64 + h * 64, qok, lh, 1.f / l);
  if (qok && lh == 0 && lse) lse[((long)b * heads + h) * L + q] = m + __logf(l);
}

// ---------------------------------------------------------------------------------------------
// backward, part 1: queries on lanes.  dQ^T = K^T dS^T / 8, and delta[q] = sum_d dO O
// ---------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(64 * (NKT > 8 ? 8 : NKT)) void attn_bwd_dq_kernel(
    const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ dqkv, float* __restrict__ delta, int L, int heads, float* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CH = NKT * 32;
  float* Ks = smem;
  float* Vs = smem + CH * KS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long rs = (long)heads * 192, ro = (long)heads * 64;
  const float* base = qkv + (long)b * L * rs + h * 192;
  const int nchunks = (L + CH - 1) / CH;
  const int q = blockIdx.y * CH + wid * 32 + lr;
  const bool qok = q < L;
  f32x4 qf[8], gf[8];
  load_frag(qf, base + (long)q * rs, qok, lh, 0.125f);
  const float* orow = out + ((long)b * L + q) * ro + h * 64;
  const float* grow = dout + ((long)b * L + q) * ro + h * 64;
  load_frag(gf, grow, qok, lh, 1.f);
  float dl = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    f32x4 ov = qok ? *reinterpret_cast<const f32x4*>(orow + g * 8 + lh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    dl += ov[0] * gf[g][0] + ov[1] * gf[g][1] + ov[2] * gf[g][2] + ov[3] * gf[g][3];
  }
  dl += __shfl_xor(dl, 32, 64);
  const float ls = qok ? lse[((long)b * heads + h) * L + q] : 0.f;
  if (qok && lh == 0) delta[((long)b * heads + h) * L + q] = dl;

  f32x16 dq[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
#pragma unroll 1
  for (int kc = 0; kc < nchunks; ++kc) {
    const int k0 = kc * CH, kn = min(CH, L - k0);
    __syncthreads();
    if (kn < CH) {
      for (int i = tid; i < CH * KS * 2; i += blockDim.x) smem[i] = 0.f;
      __syncthreads();
    }
    load_rows_to_lds(Ks, base + (long)k0 * rs + 64, rs, kn, tid, blockDim.x);
    load_rows_to_lds(Vs, base + (long)k0 * rs + 128, rs, kn, tid, blockDim.x);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 s = rows_times_frag(Ks + kt * 32 * KS, qf, lr, lh);
      f32x16 dp = rows_times_frag(Vs + kt * 32 * KS, gf, lr, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = (kt * 32 + acc_row(r, lh) < kn) ? __expf(s[r] - ls) : 0.f;
        s[r] = p * (dp[r] - dl);                    // dS^T[key][q]
      }
      accum_T(dq, Ks, kt * 32, s, lr, lh);
    }
  }
  store_T(dq, dqkv + ((long)b * L + q) * rs + h * 192, qok, lh, 0.125f);
  if (amax) adm_amax_commit(amax_T(dq, qok, 0.125f), amax);      // bound vector of |dqkv| (the qkv conv's gradients may run on the fp16 format)
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: keys on lanes.  dV^T = dO^T P,  dK^T = Q^T dS / 8
// ---------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(64 * (NKT > 8 ? 8 : NKT)) void attn_bwd_dkv_kernel(
    const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, float* __restrict__ dqkv, int L, int heads, float* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CH = NKT * 32;
  float* Qs = smem;                          // [CH][KS]
  float* Gs = smem + CH * KS;                // dO rows
  float* Ls = Gs + CH * KS;                  // lse[CH] | delta[CH]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long rs = (long)heads * 192, ro = (long)heads * 64;
  const float* base = qkv + (long)b * L * rs + h * 192;
  const int nchunks = (L + CH - 1) / CH;
  const int key = blockIdx.y * CH + wid * 32 + lr;
  const bool kok = key < L;
  f32x4 kf[8], vf[8];
  load_frag(kf, base + (long)key * rs + 64, kok, lh, 0.125f);
  load_frag(vf, base + (long)key * rs + 128, kok, lh, 1.f);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
#pragma unroll 1
  for (int qc = 0; qc < nchunks; ++qc) {
    const int q0 = qc * CH, qn = min(CH, L - q0);
    __syncthreads();
    if (qn < CH) {
      for (int i = tid; i < CH * KS * 2; i += blockDim.x) smem[i] = 0.f;
      __syncthreads();
    }
    load_rows_to_lds(Qs, base + (long)q0 * rs, rs, qn, tid, blockDim.x);
    load_rows_to_lds(Gs, dout + ((long)b * L + q0) * ro + h * 64, ro, qn, tid, blockDim.x);
    for (int i = tid; i < CH; i += blockDim.x) {
      Ls[i] = i < qn ? lse[((long)b * heads + h) * L + q0 + i] : 0.f;
      Ls[CH + i] = i < qn ? delta[((long)b * heads + h) * L + q0 + i] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int qt = 0; qt < NKT; ++qt) {
      f32x16 s = rows_times_frag(Qs + qt * 32 * KS, kf, lr, lh);      // S[q][key] (already / 8)
      f32x16 dp = rows_times_frag(Gs + qt * 32 * KS, vf, lr, lh);     // dP[q][key]
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int qq = qt * 32 + acc_row(r, lh);
        float p = (qq < qn) ? __expf(s[r] - Ls[qq]) : 0.f;
        s[r] = p;
        ds[r] = p * (dp[r] - Ls[CH + qq]);
      }
      accum_T(dv, Gs, qt * 32, s, lr, lh);
      accum_T(dk, Qs, qt * 32, ds, lr, lh);
    }
  }
  float* orow = dqkv + ((long)b * L + key) * rs + h * 192;
  store_T(dk, orow + 64, kok, lh, 0.125f);
  store_T(dv, orow + 128, kok, lh, 1.f);
  if (amax) adm_amax_commit(fmaxf(amax_T(dk, kok, 0.125f), amax_T(dv, kok, 1.f)), amax);
}

template <int NKT>
int launch_attn(int which, const float* qkv, const float* out, const float* dout, float* o_out, float* lse,
                float* dqkv, float* delta, int B, int L, int heads, hipStream_t st, float* amax) {
  constexpr int NW = NKT > 8 ? 8 : NKT;
  const int smem_kv = NKT * 32 * KS * 2 * (int)sizeof(float);
  const int smem_dkv = smem_kv + NKT * 32 * 2 * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<NKT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_kv);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<NKT, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem_kv);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<NKT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem_kv);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<NKT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem_dkv);
    if (e != hipSuccess) return ADM_ELAUNCH;
    attr_set = true;
  }
  dim3 grid(B * heads, (L + NKT * 32 - 1) / (NKT * 32)), block(64 * NW);
  if (which == 0) {
    if (grid.y > 1) hipLaunchKernelGGL((attn_fwd_kernel<NKT, true>), grid, block, smem_kv, st, qkv, o_out, lse, L, heads);
    else hipLaunchKernelGGL((attn_fwd_kernel<NKT, false>), grid, block, smem_kv, st, qkv, o_out, lse, L, heads);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NKT>), grid, block, smem_kv, st, qkv, out, dout, lse, dqkv, delta, L, heads, amax);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<NKT>), grid, block, smem_dkv, st, qkv, dout, lse, delta, dqkv, L, heads, amax);
  }
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

int dispatch_attn(int which, const float* qkv, const float* out, const float* dout, float* o_out, float* lse,
                  float* dqkv, float* delta, int B, int L, int heads, hipStream_t st, float* amax = nullptr) {
  if (!qkv || B <= 0 || heads <= 0 || L <= 0 || L > 65536) return ADM_EINVAL;
  if (L > 32 && (L % 32) != 0) return ADM_EINVAL;
  if ((uintptr_t)qkv & 15) return ADM_EINVAL;
  const int nkt = L > 256 ? 8 : (L + 31) / 32;        // longer sequences: 256-row chunks, flash-style
  switch (nkt) {
    case 1: return launch_attn<1>(which, qkv, out, dout, o_out, lse, dqkv, delta, B, L, heads, st, amax);
    case 2: return launch_attn<2>(which, qkv, out, dout, o_out, lse, dqkv, delta, B, L, heads, st, amax);
    case 4: return launch_attn<4>(which, qkv, out, dout, o_out, lse, dqkv, delta, B, L, heads, st, amax);
    case 8: return launch_attn<8>(which, qkv, out, dout, o_out, lse, dqkv, delta, B, L, heads, st, amax);
    default: return ADM_EINVAL;
  }
}

}  // namespace

extern "C" int adm_attn_fwd(const float* qkv, float* out, float* lse, int B, int L, int heads, hipStream_t stream) {
  if (!out) return ADM_EINVAL;
  return dispatch_attn(0, qkv, nullptr, nullptr, out, lse, nullptr, nullptr, B, L, heads, stream);
}

extern "C" int adm_attn_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                            float* delta, int B, int L, int heads, hipStream_t stream) {
  if (!out || !dout || !lse || !dqkv || !delta) return ADM_EINVAL;
  return dispatch_attn(1, qkv, out, dout, nullptr, const_cast<float*>(lse), dqkv, delta, B, L, heads, stream);
}

// adm_attn_bwd that also raises the bound vector amax (include/adm_hip.h; zeroed by the caller) to max |dqkv|
extern "C" int adm_attn_bwd_amax(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                 float* delta, float* amax, int B, int L, int heads, hipStream_t stream) {
  if (!out || !dout || !lse || !dqkv || !delta) return ADM_EINVAL;
  return dispatch_attn(1, qkv, out, dout, nullptr, const_cast<float*>(lse), dqkv, delta, B, L, heads, stream, amax);
}
