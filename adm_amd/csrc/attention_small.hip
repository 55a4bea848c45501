// Attention variants of the conditional super-resolution denoiser, forward and backward:
//
//   * generic multi-head softmax attention with SEPARATE query / key lengths and head dims 16 / 32 / 64: the windowed
//     cross-attention of RelationNet (/root/reference/unet/cond_unet_sd.py:221-231: 8 heads of C/8 channels, 16 pooled
//     condition windows attending to up to 1024 pooled feature windows, NO 1/sqrt(d) factor) and the bottleneck's
//     Attention (:532-554: 4 heads x 32, q scaled by 32^-0.5, L = 256);
//   * LinearAttention (:502-530: 4 heads x 32; q softmax over the head dim, k softmax over the PIXELS, context = k v^T / N).
//
// These are small (<= 1 GFLOP per step at the DIV2K recipe's sizes) next to the convolutions, so they run on the vector
// ALU with LDS-staged tiles; every reduction has a fixed order (no atomics).  The unconditional UNet's d = 64
// self-attention stays on the MFMA kernels of attention.hip.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

// ================================================================================================ generic MHA
// q[b][i][h*D + d] with row stride ldq (likewise k, v, o, dO); one thread per query (fwd, dq) or per key (dk, dv).
template <int D>
__global__ __launch_bounds__(64) void mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                     const float* __restrict__ v, float* __restrict__ o, float* __restrict__ lse,
                                                     int Lq, int Lk, int H, int ldq, int ldk, int ldv, int ldo, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[64][D], Vs[64][D];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int i = blockIdx.y * 64 + threadIdx.x;
  const bool act = i < Lq;
  float qv[D], acc[D];
  const float* qp = q + ((long)b * Lq + (act ? i : 0)) * ldq + h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) { qv[d] = qp[d] * scale; acc[d] = 0.f; }
  float m = -3.0e38f, l = 0.f;
  for (int j0 = 0; j0 < Lk; j0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * (D / 4); e += 64) {
      const int j = e / (D / 4), c4 = e - j * (D / 4);
      f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
      if (j0 + j < Lk) {
        kv = *reinterpret_cast<const f32x4*>(k + ((long)b * Lk + j0 + j) * ldk + h * D + c4 * 4);
        vv = *reinterpret_cast<const f32x4*>(v + ((long)b * Lk + j0 + j) * ldv + h * D + c4 * 4);
      }
      *reinterpret_cast<f32x4*>(&Ks[j][c4 * 4]) = kv;
      *reinterpret_cast<f32x4*>(&Vs[j][c4 * 4]) = vv;
    }
    __syncthreads();
    const int jn = min(64, Lk - j0);
    for (int j = 0; j < jn; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += qv[d] * Ks[j][d];
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = acc[d] * corr + p * Vs[j][d];
      m = mn;
    }
  }
  if (act) {
    const float inv = 1.f / l;
    float* op = o + ((long)b * Lq + i) * ldo + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) op[d] = acc[d] * inv;
    if (lse) lse[(long)bh * Lq + i] = m + __logf(l);
  }
}

// dq_i = scale * sum_j p_ij (dO_i . v_j - delta_i) k_j ;  also writes delta_i = dO_i . O_i
template <int D>
__global__ __launch_bounds__(64) void mha_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ o,
                                                        const float* __restrict__ dO, const float* __restrict__ lse,
                                                        float* __restrict__ dq, float* __restrict__ delta, int Lq, int Lk, int H,
                                                        int ldq, int ldk, int ldv, int ldo, int lddq, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[64][D], Vs[64][D];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int i = blockIdx.y * 64 + threadIdx.x;
  const bool act = i < Lq;
  const long row = (long)b * Lq + (act ? i : 0);
  float qv[D], dov[D], acc[D];
  float dl = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    qv[d] = q[row * ldq + h * D + d] * scale;
    dov[d] = dO[row * ldo + h * D + d];
    dl += dov[d] * o[row * ldo + h * D + d];
    acc[d] = 0.f;
  }
  const float ls = act ? lse[(long)bh * Lq + i] : 0.f;
  for (int j0 = 0; j0 < Lk; j0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * (D / 4); e += 64) {
      const int j = e / (D / 4), c4 = e - j * (D / 4);
      f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
      if (j0 + j < Lk) {
        kv = *reinterpret_cast<const f32x4*>(k + ((long)b * Lk + j0 + j) * ldk + h * D + c4 * 4);
        vv = *reinterpret_cast<const f32x4*>(v + ((long)b * Lk + j0 + j) * ldv + h * D + c4 * 4);
      }
      *reinterpret_cast<f32x4*>(&Ks[j][c4 * 4]) = kv;
      *reinterpret_cast<f32x4*>(&Vs[j][c4 * 4]) = vv;
    }
    __syncthreads();
    const int jn = min(64, Lk - j0);
    for (int j = 0; j < jn; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += qv[d] * Ks[j][d]; dp += dov[d] * Vs[j][d]; }
      const float ds = __expf(s - ls) * (dp - dl);
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] += ds * Ks[j][d];
    }
  }
  if (act) {
    float* dp_ = dq + row * lddq + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) dp_[d] = acc[d] * scale;
    delta[(long)bh * Lq + i] = dl;
  }
}

// dv_j = sum_i p_ij dO_i ;  dk_j = scale * sum_i p_ij (dO_i . v_j - delta_i) q_i
template <int D>
__global__ __launch_bounds__(64) void mha_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ dO,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         float* __restrict__ dk, float* __restrict__ dv, int Lq, int Lk, int H,
                                                         int ldq, int ldk, int ldv, int ldo, int lddk, int lddv, float scale) {
  __shared__ __attribute__((aligned(16))) float Qs[64][D], Os[64][D];
  __shared__ float Ls[64], Ds[64];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int j = blockIdx.y * 64 + threadIdx.x;
  const bool act = j < Lk;
  const long row = (long)b * Lk + (act ? j : 0);
  float kv[D], vv[D], ak[D], av[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    kv[d] = k[row * ldk + h * D + d];
    vv[d] = v[row * ldv + h * D + d];
    ak[d] = 0.f; av[d] = 0.f;
  }
  for (int i0 = 0; i0 < Lq; i0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * (D / 4); e += 64) {
      const int i = e / (D / 4), c4 = e - i * (D / 4);
      f32x4 qv = {0, 0, 0, 0}, ov = {0, 0, 0, 0};
      if (i0 + i < Lq) {
        qv = *reinterpret_cast<const f32x4*>(q + ((long)b * Lq + i0 + i) * ldq + h * D + c4 * 4) * scale;
        ov = *reinterpret_cast<const f32x4*>(dO + ((long)b * Lq + i0 + i) * ldo + h * D + c4 * 4);
      }
      *reinterpret_cast<f32x4*>(&Qs[i][c4 * 4]) = qv;
      *reinterpret_cast<f32x4*>(&Os[i][c4 * 4]) = ov;
    }
    if (i0 + (int)threadIdx.x < Lq) {
      Ls[threadIdx.x] = lse[(long)bh * Lq + i0 + threadIdx.x];
      Ds[threadIdx.x] = delta[(long)bh * Lq + i0 + threadIdx.x];
    }
    __syncthreads();
    const int in = min(64, Lq - i0);
    for (int i = 0; i < in; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += Qs[i][d] * kv[d]; dp += Os[i][d] * vv[d]; }
      const float p = __expf(s - Ls[i]);
      const float ds = p * (dp - Ds[i]);
#pragma unroll
      for (int d = 0; d < D; ++d) { av[d] += p * Os[i][d]; ak[d] += ds * Qs[i][d]; }       // Qs already carries `scale`
    }
  }
  if (act) {
#pragma unroll
    for (int d = 0; d < D; ++d) { dk[row * lddk + h * D + d] = ak[d]; dv[row * lddv + h * D + d] = av[d]; }
  }
}

// ================================================================================================ linear attention
constexpr int LA_H = 4, LA_D = 32, LA_C = LA_H * LA_D;       // heads, head dim, hidden (cond_unet_sd.py:503-507)

// stage 1 of the key softmax over pixels: per (image, chunk) online (max, sum exp) of every k channel
__global__ __launch_bounds__(256) void la_kstat_part_kernel(const float* __restrict__ qkv, float* __restrict__ part, int N,
                                                            int rows_per_chunk) {
  __shared__ float sm[8][LA_C][2];
  const int b = blockIdx.x, ch = blockIdx.y, cq = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int n0 = ch * rows_per_chunk, n1 = min(N, n0 + rows_per_chunk);
  f32x4 m = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, l = {0, 0, 0, 0};
  for (int n = n0 + ry; n < n1; n += 8) {
    const f32x4 kv = *reinterpret_cast<const f32x4*>(qkv + ((long)b * N + n) * 3 * LA_C + LA_C + cq * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float mn = fmaxf(m[t], kv[t]);
      l[t] = l[t] * __expf(m[t] - mn) + __expf(kv[t] - mn);
      m[t] = mn;
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) { sm[ry][cq * 4 + t][0] = m[t]; sm[ry][cq * 4 + t][1] = l[t]; }
  __syncthreads();
  if (threadIdx.x < LA_C) {
    const int c = threadIdx.x;
    float mm = sm[0][c][0], ll = sm[0][c][1];
    for (int r = 1; r < 8; ++r) {
      const float mn = fmaxf(mm, sm[r][c][0]);
      ll = ll * __expf(mm - mn) + sm[r][c][1] * __expf(sm[r][c][0] - mn);
      mm = mn;
    }
    float* o = part + (((long)b * gridDim.y + ch) * LA_C + c) * 2;
    o[0] = mm; o[1] = ll;
  }
}

// stage 2: kst[b][c] = (max, 1 / sum exp)
__global__ void la_kstat_final_kernel(const float* __restrict__ part, float* __restrict__ kst, int chunks) {
  const int b = blockIdx.x, c = threadIdx.x;
  if (c >= LA_C) return;
  float mm = -3.0e38f, ll = 0.f;
  for (int ch = 0; ch < chunks; ++ch) {
    const float* p = part + (((long)b * chunks + ch) * LA_C + c) * 2;
    if (p[1] <= 0.f) continue;
    const float mn = fmaxf(mm, p[0]);
    ll = ll * __expf(mm - mn) + p[1] * __expf(p[0] - mn);
    mm = mn;
  }
  kst[((long)b * LA_C + c) * 2] = mm;
  kst[((long)b * LA_C + c) * 2 + 1] = 1.f / ll;
}

// Partial outer-product sums over a chunk of pixels, per (image, head):
//   MODE 0 (forward):  P[d][e] = sum_n ks[n][d] v[n][e]                    ks = exp(k - max) / sumexp   (softmax over pixels)
//   MODE 1 (backward): P[d][e] = sum_n qs[n][d] dOut[n][e]                 qs = softmax_d(q) * scale
// thread (d = tid / 8, e-quad = tid % 8); 64-pixel tiles staged in LDS.
template <int MODE>
__global__ __launch_bounds__(256) void la_outer_part_kernel(const float* __restrict__ qkv, const float* __restrict__ kst,
                                                            const float* __restrict__ dout, float* __restrict__ part, int N,
                                                            int rows_per_chunk, float scale) {
  __shared__ __attribute__((aligned(16))) float As[64][LA_D + 1], Bs[64][LA_D];
  const int bh = blockIdx.x, b = bh / LA_H, h = bh - b * LA_H, ch = blockIdx.y;
  const int n0 = ch * rows_per_chunk, n1 = min(N, n0 + rows_per_chunk);
  const int d = threadIdx.x >> 3, eq = threadIdx.x & 7;
  f32x4 acc = {0, 0, 0, 0};
  for (int t0 = n0; t0 < n1; t0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int r = e >> 3, c4 = e & 7;
      f32x4 a = {0, 0, 0, 0}, bb = {0, 0, 0, 0};
      if (t0 + r < n1) {
        const float* row = qkv + ((long)b * N + t0 + r) * 3 * LA_C;
        if (MODE == 0) {
          a = *reinterpret_cast<const f32x4*>(row + LA_C + h * LA_D + c4 * 4);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float* ks = kst + ((long)b * LA_C + h * LA_D + c4 * 4 + t) * 2;
            a[t] = __expf(a[t] - ks[0]) * ks[1];
          }
          bb = *reinterpret_cast<const f32x4*>(row + 2 * LA_C + h * LA_D + c4 * 4);
        } else {
          a = *reinterpret_cast<const f32x4*>(row + h * LA_D + c4 * 4);
          bb = *reinterpret_cast<const f32x4*>(dout + ((long)b * N + t0 + r) * LA_C + h * LA_D + c4 * 4);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) As[r][c4 * 4 + t] = a[t];
      *reinterpret_cast<f32x4*>(&Bs[r][c4 * 4]) = bb;
    }
    __syncthreads();
    if (MODE == 1) {          // softmax over the head dim, row by row (64 rows, one thread each)
      if (threadIdx.x < 64) {
        const int r = threadIdx.x;
        float mx = As[r][0];
        for (int t = 1; t < LA_D; ++t) mx = fmaxf(mx, As[r][t]);
        float sum = 0.f;
        for (int t = 0; t < LA_D; ++t) { const float e_ = __expf(As[r][t] - mx); As[r][t] = e_; sum += e_; }
        const float inv = (t0 + r < n1) ? scale / sum : 0.f;
        for (int t = 0; t < LA_D; ++t) As[r][t] *= inv;
      }
      __syncthreads();
    }
    const int rn = min(64, n1 - t0);
    for (int r = 0; r < rn; ++r) acc += *reinterpret_cast<const f32x4*>(&Bs[r][eq * 4]) * As[r][d];
  }
  *reinterpret_cast<f32x4*>(part + ((((long)bh * gridDim.y + ch) * LA_D + d) * LA_D) + eq * 4) = acc;
}

// out[bh][d][e] = mul * sum_chunks part ;  optional S[bh][d] = sum_e out[d][e] * other[bh][d][e]
__global__ __launch_bounds__(1024) void la_outer_final_kernel(const float* __restrict__ part, float* __restrict__ out, int chunks,
                                                              float mul, const float* __restrict__ other, float* __restrict__ S) {
  __shared__ float prod[LA_D][LA_D + 1];
  const int bh = blockIdx.x, d = threadIdx.x >> 5, e = threadIdx.x & 31;
  float a = 0.f;
  for (int ch = 0; ch < chunks; ++ch) a += part[(((long)bh * chunks + ch) * LA_D + d) * LA_D + e];
  a *= mul;
  out[((long)bh * LA_D + d) * LA_D + e] = a;
  if (S) {
    prod[d][e] = a * other[((long)bh * LA_D + d) * LA_D + e];
    __syncthreads();
    if (e == 0) {
      float s = 0.f;
      for (int t = 0; t < LA_D; ++t) s += prod[d][t];
      S[(long)bh * LA_D + d] = s;
    }
  }
}

// out[n][h*32 + e] = sum_d ctx[h][d][e] qs[n][d],  qs = softmax_d(q) * scale.  thread = (pixel, head); 64 pixels per block
__global__ __launch_bounds__(256) void la_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                     float* __restrict__ out, int N, float scale) {
  __shared__ __attribute__((aligned(16))) float cs[LA_H][LA_D][LA_D];
  const int b = blockIdx.x, h = threadIdx.x & 3, n = blockIdx.y * 64 + (threadIdx.x >> 2);
  for (int e = threadIdx.x; e < LA_H * LA_D * LA_D; e += 256) (&cs[0][0][0])[e] = ctx[(long)b * LA_H * LA_D * LA_D + e];
  __syncthreads();
  if (n >= N) return;
  const float* qp = qkv + ((long)b * N + n) * 3 * LA_C + h * LA_D;
  float qs[LA_D];
  float mx = -3.0e38f;
#pragma unroll
  for (int d = 0; d < LA_D; ++d) { qs[d] = qp[d]; mx = fmaxf(mx, qs[d]); }
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < LA_D; ++d) { qs[d] = __expf(qs[d] - mx); sum += qs[d]; }
  const float inv = scale / sum;
  float o[LA_D];
#pragma unroll
  for (int e = 0; e < LA_D; ++e) o[e] = 0.f;
#pragma unroll 4
  for (int d = 0; d < LA_D; ++d) {
    const float w = qs[d] * inv;
#pragma unroll
    for (int e = 0; e < LA_D; ++e) o[e] += cs[h][d][e] * w;
  }
  float* op = out + ((long)b * N + n) * LA_C + h * LA_D;
#pragma unroll
  for (int e = 0; e < LA_D; e += 4) *reinterpret_cast<f32x4*>(op + e) = f32x4{o[e], o[e + 1], o[e + 2], o[e + 3]};
}

// per (pixel, head): dq, dk, dv from dOut, ctx, dctx, the key statistics and S[d] = sum_e dctx[d][e] ctx[d][e]
__global__ __launch_bounds__(256) void la_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                     const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                     const float* __restrict__ kst, const float* __restrict__ S,
                                                     float* __restrict__ dqkv, int N, float scale) {
  __shared__ __attribute__((aligned(16))) float cs[LA_H][LA_D][LA_D], ds[LA_H][LA_D][LA_D];
  __shared__ float ks_m[LA_C], ks_r[LA_C], S_s[LA_C];
  const int b = blockIdx.x, h = threadIdx.x & 3, n = blockIdx.y * 64 + (threadIdx.x >> 2);
  for (int e = threadIdx.x; e < LA_H * LA_D * LA_D; e += 256) {
    (&cs[0][0][0])[e] = ctx[(long)b * LA_H * LA_D * LA_D + e];
    (&ds[0][0][0])[e] = dctx[(long)b * LA_H * LA_D * LA_D + e];
  }
  if (threadIdx.x < LA_C) {
    ks_m[threadIdx.x] = kst[((long)b * LA_C + threadIdx.x) * 2];
    ks_r[threadIdx.x] = kst[((long)b * LA_C + threadIdx.x) * 2 + 1];
    S_s[threadIdx.x] = S[(long)b * LA_C + threadIdx.x];
  }
  __syncthreads();
  if (n >= N) return;
  const float* row = qkv + ((long)b * N + n) * 3 * LA_C + h * LA_D;
  float* drow = dqkv + ((long)b * N + n) * 3 * LA_C + h * LA_D;
  const float invN = 1.f / (float)N;
  float go[LA_D];
#pragma unroll
  for (int e = 0; e < LA_D; ++e) go[e] = dout[((long)b * N + n) * LA_C + h * LA_D + e];
  {   // ---- dq: qsm = softmax_d(q); g_d = scale * sum_e ctx[d][e] dOut[e]; dq = qsm (g - sum_d g qsm)
    float qs[LA_D];
    float mx = -3.0e38f;
#pragma unroll
    for (int d = 0; d < LA_D; ++d) { qs[d] = row[d]; mx = fmaxf(mx, qs[d]); }
    float sum = 0.f;
#pragma unroll
    for (int d = 0; d < LA_D; ++d) { qs[d] = __expf(qs[d] - mx); sum += qs[d]; }
    const float inv = 1.f / sum;
    float dot = 0.f;
    float g[LA_D];
#pragma unroll 4
    for (int d = 0; d < LA_D; ++d) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < LA_D; ++e) a += cs[h][d][e] * go[e];
      qs[d] *= inv;
      g[d] = a * scale;
      dot += g[d] * qs[d];
    }
#pragma unroll
    for (int d = 0; d < LA_D; ++d) drow[d] = qs[d] * (g[d] - dot);
  }
  {   // ---- dk, dv: ks = exp(k - max) / sumexp; dv[e] = sum_d ks[d] dctx[d][e] / N; dk[d] = ks[d] (sum_e dctx[d][e] v[e] / N - S[d])
    float vv[LA_D], dv[LA_D];
#pragma unroll
    for (int e = 0; e < LA_D; ++e) { vv[e] = row[2 * LA_C + e]; dv[e] = 0.f; }
#pragma unroll 4
    for (int d = 0; d < LA_D; ++d) {
      const float ks = __expf(row[LA_C + d] - ks_m[h * LA_D + d]) * ks_r[h * LA_D + d];
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < LA_D; ++e) { a += ds[h][d][e] * vv[e]; dv[e] += ks * ds[h][d][e]; }
      drow[LA_C + d] = ks * (a * invN - S_s[h * LA_D + d]);
    }
#pragma unroll
    for (int e = 0; e < LA_D; ++e) drow[2 * LA_C + e] = dv[e] * invN;
  }
}

inline int la_chunks(int N) { int c = (N + 1023) / 1024; return c < 1 ? 1 : (c > 64 ? 64 : c); }

}  // namespace

// ================================================================================================ C ABI
namespace {
template <int D>
int mha_fwd_launch(const float* q, const float* k, const float* v, float* o, float* lse, int B, int Lq, int Lk, int H, int ldq,
                   int ldk, int ldv, int ldo, float scale, hipStream_t st) {
  hipLaunchKernelGGL((mha_fwd_kernel<D>), dim3(B * H, (Lq + 63) / 64), dim3(64), 0, st, q, k, v, o, lse, Lq, Lk, H, ldq, ldk, ldv,
                     ldo, scale);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
template <int D>
int mha_bwd_launch(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse, float* dq,
                   float* dk, float* dv, float* delta, int B, int Lq, int Lk, int H, int ldq, int ldk, int ldv, int ldo, int lddq,
                   int lddk, int lddv, float scale, hipStream_t st) {
  hipLaunchKernelGGL((mha_bwd_dq_kernel<D>), dim3(B * H, (Lq + 63) / 64), dim3(64), 0, st, q, k, v, o, dO, lse, dq, delta, Lq, Lk, H,
                     ldq, ldk, ldv, ldo, lddq, scale);
  hipLaunchKernelGGL((mha_bwd_dkv_kernel<D>), dim3(B * H, (Lk + 63) / 64), dim3(64), 0, st, q, k, v, dO, lse, delta, dk, dv, Lq, Lk,
                     H, ldq, ldk, ldv, ldo, lddk, lddv, scale);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
inline bool mha_ok(int B, int Lq, int Lk, int H, int D, int a, int b, int c, int d) {
  return B > 0 && Lq > 0 && Lk > 0 && H > 0 && (D == 4 || D == 8 || D == 16 || D == 32 || D == 64) && !((a | b | c | d) & 3) && a >= H * D && b >= H * D &&
         c >= H * D && d >= H * D;
}
}  // namespace

// softmax(scale * q k^T) v per (image, head): q[B][Lq][ldq], k[B][Lk][ldk], v[B][Lk][ldv] -> o[B][Lq][ldo]; head h uses
// columns [h*D, (h+1)*D) of each row.  lse[B*H][Lq] (may be NULL when no backward follows).  D in {4, 8, 16, 32, 64} (4 / 8 occur only in reduced-width test models).
extern "C" int adm_mha_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int Lq, int Lk, int H,
                           int D, int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t stream) {
  if (!q || !k || !v || !o || !mha_ok(B, Lq, Lk, H, D, ldq, ldk, ldv, ldo)) return ADM_EINVAL;
  if (((uintptr_t)k | (uintptr_t)v) & 15) return ADM_EINVAL;
  if (D == 4) return mha_fwd_launch<4>(q, k, v, o, lse, B, Lq, Lk, H, ldq, ldk, ldv, ldo, scale, stream);
  if (D == 8) return mha_fwd_launch<8>(q, k, v, o, lse, B, Lq, Lk, H, ldq, ldk, ldv, ldo, scale, stream);
  if (D == 16) return mha_fwd_launch<16>(q, k, v, o, lse, B, Lq, Lk, H, ldq, ldk, ldv, ldo, scale, stream);
  if (D == 32) return mha_fwd_launch<32>(q, k, v, o, lse, B, Lq, Lk, H, ldq, ldk, ldv, ldo, scale, stream);
  return mha_fwd_launch<64>(q, k, v, o, lse, B, Lq, Lk, H, ldq, ldk, ldv, ldo, scale, stream);
}

// delta = workspace [B*H][Lq]
extern "C" int adm_mha_bwd(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse,
                           float* dq, float* dk, float* dv, float* delta, int B, int Lq, int Lk, int H, int D, int ldq, int ldk,
                           int ldv, int ldo, int lddq, int lddk, int lddv, float scale, hipStream_t stream) {
  if (!q || !k || !v || !o || !dO || !lse || !dq || !dk || !dv || !delta || !mha_ok(B, Lq, Lk, H, D, ldq, ldk, ldv, ldo))
    return ADM_EINVAL;
  if (((lddq | lddk | lddv) & 3) || lddq < H * D || lddk < H * D || lddv < H * D) return ADM_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dO) & 15) return ADM_EINVAL;
  if (D == 4) return mha_bwd_launch<4>(q, k, v, o, dO, lse, dq, dk, dv, delta, B, Lq, Lk, H, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, stream);
  if (D == 8) return mha_bwd_launch<8>(q, k, v, o, dO, lse, dq, dk, dv, delta, B, Lq, Lk, H, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, stream);
  if (D == 16) return mha_bwd_launch<16>(q, k, v, o, dO, lse, dq, dk, dv, delta, B, Lq, Lk, H, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, stream);
  if (D == 32) return mha_bwd_launch<32>(q, k, v, o, dO, lse, dq, dk, dv, delta, B, Lq, Lk, H, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, stream);
  return mha_bwd_launch<64>(q, k, v, o, dO, lse, dq, dk, dv, delta, B, Lq, Lk, H, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, stream);
}

// workspace floats of the linear attention calls for N pixels per image
extern "C" long adm_linattn_ws_floats(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const long ch = la_chunks(N);
  return (long)B * ch * 128 * 2 + (long)B * 4 * ch * 32 * 32;
}

// LinearAttention core (cond_unet_sd.py:516-529): qkv [B][N][384] (q | k | v, 4 heads x 32 each) -> out [B][N][128].
// Saved for the backward: ctx [B][4][32][32] and kst [B][128][2] (key softmax statistics over the N pixels).
extern "C" int adm_linattn_fwd(const float* qkv, float* out, float* ctx, float* kst, float* ws, int B, int N, hipStream_t stream) {
  if (!qkv || !out || !ctx || !kst || !ws || B <= 0 || N <= 0 || ((uintptr_t)qkv & 15)) return ADM_EINVAL;
  const int ch = la_chunks(N), rows = (N + ch - 1) / ch;
  float* kpart = ws;
  float* cpart = ws + (long)B * ch * 128 * 2;
  const float scale = 0.17677669529663687f;      // 32^-0.5
  hipLaunchKernelGGL(la_kstat_part_kernel, dim3(B, ch), dim3(256), 0, stream, qkv, kpart, N, rows);
  hipLaunchKernelGGL(la_kstat_final_kernel, dim3(B), dim3(128), 0, stream, kpart, kst, ch);
  hipLaunchKernelGGL((la_outer_part_kernel<0>), dim3(B * 4, ch), dim3(256), 0, stream, qkv, kst, nullptr, cpart, N, rows, scale);
  hipLaunchKernelGGL(la_outer_final_kernel, dim3(B * 4), dim3(1024), 0, stream, cpart, ctx, ch, 1.f / (float)N, nullptr, nullptr);
  hipLaunchKernelGGL(la_out_kernel, dim3(B, (N + 63) / 64), dim3(256), 0, stream, qkv, ctx, out, N, scale);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// dctx [B][4][32][32] and S [B][128] are workspaces (outputs of the first stage, inputs of the second)
extern "C" int adm_linattn_bwd(const float* qkv, const float* dout, const float* ctx, const float* kst, float* dqkv, float* dctx,
                               float* S, float* ws, int B, int N, hipStream_t stream) {
  if (!qkv || !dout || !ctx || !kst || !dqkv || !dctx || !S || !ws || B <= 0 || N <= 0) return ADM_EINVAL;
  if (((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv) & 15) return ADM_EINVAL;
  const int ch = la_chunks(N), rows = (N + ch - 1) / ch;
  float* cpart = ws + (long)B * ch * 128 * 2;
  const float scale = 0.17677669529663687f;
  hipLaunchKernelGGL((la_outer_part_kernel<1>), dim3(B * 4, ch), dim3(256), 0, stream, qkv, kst, dout, cpart, N, rows, scale);
  hipLaunchKernelGGL(la_outer_final_kernel, dim3(B * 4), dim3(1024), 0, stream, cpart, dctx, ch, 1.f, ctx, S);
  hipLaunchKernelGGL(la_bwd_kernel, dim3(B, (N + 63) / 64), dim3(256), 0, stream, qkv, dout, ctx, dctx, kst, S, dqkv, N, scale);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
