// GroupNorm (+ adaptive scale/shift) + SiLU (+ dropout), NHWC fp32, forward and backward.
//
// Replaces F.group_norm / silu / addcmul / F.dropout of UNetBlock.forward
// (/root/reference/unet/uncond_unet.py:128, 191, 196, 200).  These kernels are HBM-bound: every
// tensor is touched with 16-byte coalesced accesses, each thread keeps ONE channel quad (so all
// per-channel coefficients live in registers) and walks rows with a fixed stride.
//
// Thread map (all kernels): C4 = C/4 float4 columns; block = C4 * R threads with R = 256 / C4 rows
// in flight (>=1); thread (cq = tid % C4, ry = tid / C4) visits rows hw0 + ry, + R, ...
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

// y element store: f32x4, or (out_bf16) four bf16 (round to nearest even) -- the bf16-storage mode of BASELINE configs[2]
__device__ __forceinline__ void gn_store4(float* y, long quad_idx, f32x4 u, int out_bf16) {
  if (out_bf16) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t h;
    h[0] = (__bf16)u[0]; h[1] = (__bf16)u[1]; h[2] = (__bf16)u[2]; h[3] = (__bf16)u[3];
    reinterpret_cast<bf16x4_t*>(y)[quad_idx] = h;
  } else {
    reinterpret_cast<f32x4*>(y)[quad_idx] = u;
  }
}

// max |y| of a GroupNorm output for the fp16-format conv that consumes it (conv_wino2d_x6.hip, X6Fmt<1>): per-thread maximum ->
// wave maximum -> one atomicMax per wave on the float's bit pattern (non-negative floats order like unsigned integers)
__device__ __forceinline__ float gn_amax4(float am, f32x4 u) {
  return fmaxf(fmaxf(am, fmaxf(fabsf(u[0]), fabsf(u[1]))), fmaxf(fabsf(u[2]), fabsf(u[3])));
}
__device__ __forceinline__ void gn_amax_commit(float am, float* amax) { adm_amax_commit(am, amax); }      // (a bound vector: common.h)

__host__ __device__ inline int gn_rows_par(int C) { int r = 256 / (C / 4); return r < 1 ? 1 : r; }

// ---------------------------------------------------------------- forward: moments
// Numerically robust moments ("Welford / two-pass"): every thread accumulates SHIFTED sums
// s1 = sum(x - K), s2 = sum((x - K)^2) with K = the first value it sees per channel, so s2 - s1^2/n has no catastrophic
// cancellation for data with |mean| >> std (the plain E[x^2] - mean^2 form loses log2(mean^2/var) bits in fp32); the
// per-thread (n, mean, M2) triples are merged exactly in fp64, in a fixed order (deterministic).
// Exact merge of sub-populations (n_i, mean_i, M2_i):  N = sum n_i,  mean = sum n_i mean_i / N,
// M2 = sum [M2_i + n_i (mean_i - mean)^2]  -- two passes over the (few) partials, one division.
__global__ void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ ws, int HW, int C, int G,
                                  int rows_per_split) {
  extern __shared__ float sm[];          // [R][C][3] : K, s1, s2
  const int C4 = C >> 2, R = blockDim.x / C4;
  const int cq = threadIdx.x % C4, ry = threadIdx.x / C4;
  const int b = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int hw0 = s * rows_per_split, hw1 = min(HW, hw0 + rows_per_split);
  const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long)b * HW * C);
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, K = {0, 0, 0, 0};
  if (hw0 + ry < hw1) K = xb[(long)(hw0 + ry) * C4 + cq];
#pragma unroll 4
  for (int hw = hw0 + ry; hw < hw1; hw += R) {
    f32x4 v = xb[(long)hw * C4 + cq] - K;
    s1 += v;
    s2 += v * v;
  }
  float* p1 = sm + (ry * C + cq * 4) * 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) { p1[3 * k] = K[k]; p1[3 * k + 1] = s1[k]; p1[3 * k + 2] = s2[k]; }
  __syncthreads();
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    double N = 0.0, sum = 0.0;
    for (int r = 0; r < R; ++r) {
      const int first = hw0 + r;
      if (first >= hw1) break;
      const double n = (double)((hw1 - first + R - 1) / R);      // rows thread-row r visited
      double a = 0.0;
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) a += (double)sm[(r * C + c) * 3] * n + (double)sm[(r * C + c) * 3 + 1];
      sum += a; N += n * cpg;
    }
    const double mean = N > 0.0 ? sum / N : 0.0;
    double m2 = 0.0;
    for (int r = 0; r < R; ++r) {
      const int first = hw0 + r;
      if (first >= hw1) break;
      const double n = (double)((hw1 - first + R - 1) / R), inv_n = 1.0 / n;
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        const float* q = sm + (r * C + c) * 3;
        const double a = (double)q[1], mt = (double)q[0] + a * inv_n, d = mt - mean;
        m2 += ((double)q[2] - a * a * inv_n) + n * d * d;
      }
    }
    double* o = ws + (((long)b * S + s) * G + g) * 2;
    o[0] = mean; o[1] = m2;
  }
}

__global__ void gn_finalize_kernel(const double* __restrict__ ws, float* __restrict__ stats, int BG, int G, int S,
                                   int HW, int rows_per_split, int cpg, float eps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // b*G + g
  if (i >= BG) return;
  int b = i / G, g = i - b * G;
  double N = 0.0, sum = 0.0;
  for (int s = 0; s < S; ++s) {
    const int hw0 = s * rows_per_split, hw1 = min(HW, hw0 + rows_per_split);
    const double n = (double)max(hw1 - hw0, 0) * cpg;
    sum += n * ws[(((long)b * S + s) * G + g) * 2]; N += n;
  }
  const double mean = sum / N;
  double m2 = 0.0;
  for (int s = 0; s < S; ++s) {
    const int hw0 = s * rows_per_split, hw1 = min(HW, hw0 + rows_per_split);
    const double n = (double)max(hw1 - hw0, 0) * cpg;
    const double* o = ws + (((long)b * S + s) * G + g) * 2;
    const double d = o[0] - mean;
    if (n > 0.0) m2 += o[1] + n * d * d;
  }
  double var = m2 / N;
  if (var < 0.0) var = 0.0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---------------------------------------------------------------- forward: apply
__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ ss, long ss_bstride, float* __restrict__ y, int HW, int C,
                                int G, int rows_per_split, int silu, float drop_p, uint64_t seed, int out_bf16, float* __restrict__ amax) {
  const int C4 = C >> 2, R = blockDim.x / C4;
  const int cq = threadIdx.x % C4, ry = threadIdx.x / C4;
  const int b = blockIdx.x;
  const int hw0 = blockIdx.y * rows_per_split, hw1 = min(HW, hw0 + rows_per_split);
  float am = 0.f;
  const int cpg = C / G;
  f32x4 ca, cb;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = cq * 4 + k, g = c / cpg;
    float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    float sc1 = 1.f, sh = 0.f;
    if (ss) { sc1 = 1.f + ss[b * ss_bstride + c]; sh = ss[b * ss_bstride + C + c]; }
    float ga = gamma[c] * rstd;
    ca[k] = ga * sc1;
    cb[k] = (beta[c] - mean * ga) * sc1 + sh;
  }
  const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long)b * HW * C);
  const long ybase = (long)b * HW * C4;             // in channel quads
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll 4
  for (int hw = hw0 + ry; hw < hw1; hw += R) {
    f32x4 v = xb[(long)hw * C4 + cq];
    f32x4 u = v * ca + cb;
    if (silu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = silu_f(u[k]);
    }
    if (drop_p > 0.f) u *= dropout_keep4(seed, ((uint64_t)b * HW + hw) * C4 + cq, drop_p, inv_keep);
    am = gn_amax4(am, u);
    gn_store4(y, ybase + (long)hw * C4 + cq, u, out_bf16);
  }
  gn_amax_commit(am, amax);
}

// ---------------------------------------------------------------- backward pass 1: per (b, split, c) partials
__global__ void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ ss, long ss_bstride,
                                      float* __restrict__ part, int HW, int C, int G, int rows_per_split, int silu,
                                      float drop_p, uint64_t seed) {
  extern __shared__ float sm[];          // [R][C][2]
  const int C4 = C >> 2, R = blockDim.x / C4;
  const int cq = threadIdx.x % C4, ry = threadIdx.x / C4;
  const int b = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int hw0 = s * rows_per_split, hw1 = min(HW, hw0 + rows_per_split);
  const int cpg = C / G;
  f32x4 ca, cb, cm, cr;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = cq * 4 + k, g = c / cpg;
    float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    float sc1 = 1.f, sh = 0.f;
    if (ss) { sc1 = 1.f + ss[b * ss_bstride + c]; sh = ss[b * ss_bstride + C + c]; }
    float ga = gamma[c] * rstd;
    ca[k] = ga * sc1;
    cb[k] = (beta[c] - mean * ga) * sc1 + sh;
    cm[k] = mean; cr[k] = rstd;
  }
  const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long)b * HW * C);
  const f32x4* gb = reinterpret_cast<const f32x4*>(dy + (long)b * HW * C);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  f32x4 r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0};
#pragma unroll 4
  for (int hw = hw0 + ry; hw < hw1; hw += R) {
    f32x4 v = xb[(long)hw * C4 + cq];
    f32x4 d = gb[(long)hw * C4 + cq];
    f32x4 u = v * ca + cb;
    if (drop_p > 0.f) d *= dropout_keep4(seed, ((uint64_t)b * HW + hw) * C4 + cq, drop_p, inv_keep);
    if (silu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] *= silu_grad_f(u[k]);
    }
    r1 += d;
    r2 += d * ((v - cm) * cr);
  }
  float* p1 = sm + (ry * C + cq * 4) * 2;
#pragma unroll
  for (int k = 0; k < 4; ++k) { p1[2 * k] = r1[k]; p1[2 * k + 1] = r2[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += sm[r * C * 2 + i];
    part[(((long)b * S + s) * C) * 2 + i] = a;
  }
}

// pass 2: per b: totals over splits, d scale/shift, group means (m1, m2)
__global__ void gn_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, const float* __restrict__ ss, long ss_bstride,
                                     float* __restrict__ tot, float* __restrict__ gm, float* __restrict__ dss, int S,
                                     int HW, int C, int G) {
  extern __shared__ float sm[];          // [C][2] : gamma'(R1), gamma'(R2)
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float R1 = 0.f, R2 = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* q = part + (((long)b * S + s) * C + c) * 2;
      R1 += q[0]; R2 += q[1];
    }
    tot[((long)b * C + c) * 2] = R1;
    tot[((long)b * C + c) * 2 + 1] = R2;
    float sc1 = ss ? 1.f + ss[b * ss_bstride + c] : 1.f;
    float gp = gamma[c] * sc1;
    sm[2 * c] = gp * R1;
    sm[2 * c + 1] = gp * R2;
    if (dss) {   // dss rows have the stride of the ss rows (a slice of one [B][sum of 2C] buffer for all blocks: ops.affine_group)
      const long ds = ss_bstride ? ss_bstride : 2L * C;
      dss[(long)b * ds + c] = gamma[c] * R2 + beta[c] * R1;       // d/d scale: sum du * z,  z = xhat*gamma + beta
      dss[(long)b * ds + C + c] = R1;                              // d/d shift
    }
  }
  __syncthreads();
  const int cpg = C / G;
  const float inv_n = 1.f / ((float)HW * (float)cpg);
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += sm[2 * c]; q += sm[2 * c + 1]; }
    gm[((long)b * G + g) * 2] = a * inv_n;
    gm[((long)b * G + g) * 2 + 1] = q * inv_n;
  }
}

// pass 3: dgamma[c] += sum_b (1+s) R2, dbeta[c] += sum_b (1+s) R1.  Block = 32 channels x 8 batch lanes;
// fixed summation order (deterministic).
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ tot, const float* __restrict__ ss,
                                                           long ss_bstride, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int B, int C) {
  __shared__ float sa[8][32], sq[8][32];
  const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a = 0.f, q = 0.f;
  if (c < C)
    for (int b = bl; b < B; b += 8) {
      float sc1 = ss ? 1.f + ss[b * ss_bstride + c] : 1.f;
      a += sc1 * tot[((long)b * C + c) * 2];
      q += sc1 * tot[((long)b * C + c) * 2 + 1];
    }
  sa[bl][cl] = a; sq[bl][cl] = q;
  __syncthreads();
  if (bl == 0 && c < C) {
#pragma unroll
    for (int k = 1; k < 8; ++k) { a += sa[k][cl]; q += sq[k][cl]; }
    dbeta[c] += a;
    dgamma[c] += q;
  }
}

// gn_bwd_param_kernel for ALL GroupNorm layers of a backward pass in one launch.  Row = 8 int64: {tot, ss (0: none), ss_bstride,
// dgamma, dbeta, B, C, block_begin}; a row owns ceil(C / 32) blocks, block_begin = exclusive prefix sum.  Same summation order.
__global__ __launch_bounds__(256) void gn_bwd_param_table_kernel(const long* __restrict__ table, int rows) {
  int lo = 0, hi = rows - 1;                        // last row whose block_begin <= blockIdx.x (uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * 8 + 7] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long* r = table + (long)lo * 8;
  const float* tot = reinterpret_cast<const float*>(r[0]);
  const float* ss = reinterpret_cast<const float*>(r[1]);
  const long ss_bstride = r[2];
  float* dgamma = reinterpret_cast<float*>(r[3]);
  float* dbeta = reinterpret_cast<float*>(r[4]);
  const int B = (int)r[5], C = (int)r[6];
  __shared__ float sa[8][32], sq[8][32];
  const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int c = ((int)((long)blockIdx.x - r[7])) * 32 + cl;
  float a = 0.f, q = 0.f;
  if (c < C)
    for (int b = bl; b < B; b += 8) {
      float sc1 = ss ? 1.f + ss[b * ss_bstride + c] : 1.f;
      a += sc1 * tot[((long)b * C + c) * 2];
      q += sc1 * tot[((long)b * C + c) * 2 + 1];
    }
  sa[bl][cl] = a; sq[bl][cl] = q;
  __syncthreads();
  if (bl == 0 && c < C) {
#pragma unroll
    for (int k = 1; k < 8; ++k) { a += sa[k][cl]; q += sq[k][cl]; }
    dbeta[c] += a;
    dgamma[c] += q;
  }
}

// pass 4: dx = rstd * (gamma' du - m1 - xhat m2)
__global__ void gn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ ss, long ss_bstride,
                                 const float* __restrict__ gm, const float* __restrict__ addend,
                                 float* __restrict__ dx, int HW, int C, int G, int rows_per_split, int silu, float drop_p,
                                 uint64_t seed, float* __restrict__ amax) {
  const int C4 = C >> 2, R = blockDim.x / C4;
  const int cq = threadIdx.x % C4, ry = threadIdx.x / C4;
  const int b = blockIdx.x;
  const int hw0 = blockIdx.y * rows_per_split, hw1 = min(HW, hw0 + rows_per_split);
  const int cpg = C / G;
  f32x4 ca, cb, cm, cr, cg, c1, c2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = cq * 4 + k, g = c / cpg;
    float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    float sc1 = 1.f, sh = 0.f;
    if (ss) { sc1 = 1.f + ss[b * ss_bstride + c]; sh = ss[b * ss_bstride + C + c]; }
    float ga = gamma[c] * rstd;
    ca[k] = ga * sc1;
    cb[k] = (beta[c] - mean * ga) * sc1 + sh;
    cm[k] = mean; cr[k] = rstd;
    cg[k] = ga * sc1;                               // rstd * gamma'
    c1[k] = rstd * gm[((long)b * G + g) * 2];       // rstd * m1
    c2[k] = rstd * gm[((long)b * G + g) * 2 + 1];   // rstd * m2
  }
  const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long)b * HW * C);
  const f32x4* gb = reinterpret_cast<const f32x4*>(dy + (long)b * HW * C);
  f32x4* ob = reinterpret_cast<f32x4*>(dx + (long)b * HW * C);
  const f32x4* ab = addend ? reinterpret_cast<const f32x4*>(addend + (long)b * HW * C) : nullptr;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float am = 0.f;
#pragma unroll 4
  for (int hw = hw0 + ry; hw < hw1; hw += R) {
    f32x4 v = xb[(long)hw * C4 + cq];
    f32x4 d = gb[(long)hw * C4 + cq];
    f32x4 extra = {0, 0, 0, 0};
    if (ab) extra = ab[(long)hw * C4 + cq];
    f32x4 u = v * ca + cb;
    if (drop_p > 0.f) d *= dropout_keep4(seed, ((uint64_t)b * HW + hw) * C4 + cq, drop_p, inv_keep);
    if (silu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] *= silu_grad_f(u[k]);
    }
    const f32x4 o = cg * d - c1 - ((v - cm) * cr) * c2 + extra;
    am = gn_amax4(am, o);
    ob[(long)hw * C4 + cq] = o;
  }
  gn_amax_commit(am, amax);               // max |dx|: the data-gradient conv that consumes dx may run on the fp16 format
}

// ---------------------------------------------------------------- small feature maps: one launch per direction
// For HW <= 256 the three (forward) / four (backward) launches above are latency-bound (0.5-1.8 TB/s measured at the
// 4x4 and 8x8 levels).  Here one workgroup owns (image, chunk of WHOLE groups) and keeps its slab in REGISTERS between
// the reduction and the apply pass: x (and dy) are read from HBM exactly once, and there is a single launch.
// Thread map as above restricted to the chunk: Cc4 = Cc/4 quads, R = blockDim / Cc4 rows in flight, each thread owns
// rows ry, ry + R, ... (<= MAXR of them).  Reductions keep the fixed summation order of the multi-launch path
// (deterministic), with the same fp64 group combine.
template <int MAXR, int THREADS>
__global__ __launch_bounds__(THREADS, (MAXR <= 8 ? 4 : 3)) void gn_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ ss,
                                                           long ss_bstride, float* __restrict__ y,
                                                           float* __restrict__ stats, int HW, int C, int G, int Cc,
                                                           float eps, int silu, float drop_p, uint64_t seed, int out_bf16,
                                                           float* __restrict__ amax) {
  extern __shared__ float sm[];                    // [R][Cc][2] partials | [Gc][2] mean, rstd
  const int Cc4 = Cc >> 2, R = blockDim.x / Cc4, C4 = C >> 2;
  const int cq = threadIdx.x % Cc4, ry = threadIdx.x / Cc4;
  const int b = blockIdx.y, c0 = blockIdx.x * Cc;        // slabs of one image are adjacent in launch order: they share cache lines
  const int cpg = C / G, Gc = Cc / cpg, g0 = c0 / cpg;
  const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long)b * HW * C + c0);
  f32x4 v[MAXR];
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int hw = ry + i * R;
    const f32x4 t = xb[(long)min(hw, HW - 1) * C4 + cq];          // unconditional (clamped) load: no branch per row
    v[i] = hw < HW ? t : f32x4{0, 0, 0, 0};
  }
  // the slab is in registers: a true two-pass per thread (own mean first, then squared deviations), merged in fp64
  int nrow = 0;
#pragma unroll
  for (int i = 0; i < MAXR; ++i)
    if (ry + i * R < HW) { s1 += v[i]; ++nrow; }
  const float inv_rows = nrow ? 1.f / (float)nrow : 0.f;
  const f32x4 tm = s1 * inv_rows;
#pragma unroll
  for (int i = 0; i < MAXR; ++i)
    if (ry + i * R < HW) { const f32x4 d = v[i] - tm; s2 += d * d; }
  float* p1 = sm + (ry * Cc + cq * 4) * 2;
#pragma unroll
  for (int k = 0; k < 4; ++k) { p1[2 * k] = tm[k]; p1[2 * k + 1] = s2[k]; }
  __syncthreads();
  // exact merge of the per-thread (n, mean, M2) triples in fp64, fixed order: per channel over the R row lanes (one thread per
  // channel), then per group over its channels -- sum_r n_r mean_rc first, the squared deviations against the GROUP mean second
  float* gs = sm + R * Cc * 2;                     // [Gc][2] mean, rstd
  double* chd = reinterpret_cast<double*>(gs + ((Gc * 2 + 1) & ~1));      // [Cc] per-channel fp64 partial
  const int Rl = min(R, HW);
  for (int cl = threadIdx.x; cl < Cc; cl += blockDim.x) {
    double a = 0.0;
    for (int r = 0; r < Rl; ++r) a += (double)sm[(r * Cc + cl) * 2] * (double)((HW - r + R - 1) / R);
    chd[cl] = a;
  }
  __syncthreads();
  if ((int)threadIdx.x < Gc) {
    const int g = threadIdx.x;
    double sum = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) sum += chd[c];
    gs[2 * g] = (float)(sum / ((double)HW * cpg));
  }
  __syncthreads();
  for (int cl = threadIdx.x; cl < Cc; cl += blockDim.x) {
    const double mean = (double)gs[2 * (cl / cpg)];       // the group mean rounded to f32, as it is applied
    double m2 = 0.0;
    for (int r = 0; r < Rl; ++r) {
      const double d = (double)sm[(r * Cc + cl) * 2] - mean;
      m2 += (double)sm[(r * Cc + cl) * 2 + 1] + (double)((HW - r + R - 1) / R) * d * d;
    }
    chd[cl] = m2;
  }
  __syncthreads();
  if ((int)threadIdx.x < Gc) {
    const int g = threadIdx.x;
    double m2 = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) m2 += chd[c];
    double var = m2 / ((double)HW * cpg);
    if (var < 0.0) var = 0.0;
    const float m = gs[2 * g], rs = (float)(1.0 / sqrt(var + (double)eps));
    gs[2 * g + 1] = rs;
    stats[((long)b * G + g0 + g) * 2] = m;
    stats[((long)b * G + g0 + g) * 2 + 1] = rs;
  }
  __syncthreads();
  f32x4 ca, cb;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cl = cq * 4 + k, c = c0 + cl, g = cl / cpg;
    const float mean = gs[2 * g], rstd = gs[2 * g + 1];
    float sc1 = 1.f, sh = 0.f;
    if (ss) { sc1 = 1.f + ss[b * ss_bstride + c]; sh = ss[b * ss_bstride + C + c]; }
    const float ga = gamma[c] * rstd;
    ca[k] = ga * sc1;
    cb[k] = (beta[c] - mean * ga) * sc1 + sh;
  }
  const long ybase = (long)b * HW * C4 + (c0 >> 2);  // in channel quads
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int hw = ry + i * R;
    if (hw >= HW) continue;
    f32x4 u = v[i] * ca + cb;
    if (silu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = silu_f(u[k]);
    }
    if (drop_p > 0.f) u *= dropout_keep4(seed, ((uint64_t)b * HW + hw) * C4 + (c0 >> 2) + cq, drop_p, inv_keep);
    am = gn_amax4(am, u);
    gn_store4(y, ybase + (long)hw * C4 + cq, u, out_bf16);
    __builtin_amdgcn_sched_barrier(0);            // one row at a time: interleaving the rows' hashes / exponentials costs registers
  }
  gn_amax_commit(am, amax);
}

// (second launch bound = waves per SIMD: left alone the compiler hoists the dropout hashes and SiLU derivatives of all rows and
// takes 186-256 registers, i.e. one or two waves per SIMD for a bandwidth-bound kernel)
template <int MAXR, int THREADS>
__global__ __launch_bounds__(THREADS, (MAXR <= 8 ? 4 : 2)) void gn_fused_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ ss,
                                                           long ss_bstride, const float* __restrict__ addend,
                                                           float* __restrict__ dx, float* __restrict__ tot,
                                                           float* __restrict__ dss, int HW, int C, int G, int Cc, int silu,
                                                           float drop_p, uint64_t seed, float* __restrict__ amax) {
  extern __shared__ float sm[];                    // [R][Cc][2] partials | [Cc][2] gamma' R1, gamma' R2 | [Gc][2] m1, m2
  const int Cc4 = Cc >> 2, R = blockDim.x / Cc4, C4 = C >> 2;
  const int cq = threadIdx.x % Cc4, ry = threadIdx.x / Cc4;
  const int b = blockIdx.y, c0 = blockIdx.x * Cc;        // slabs of one image are adjacent in launch order: they share cache lines
  const int cpg = C / G, Gc = Cc / cpg, g0 = c0 / cpg;
  f32x4 ca, cb, cm, cr;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + cq * 4 + k, g = c / cpg;
    const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    float sc1 = 1.f, sh = 0.f;
    if (ss) { sc1 = 1.f + ss[b * ss_bstride + c]; sh = ss[b * ss_bstride + C + c]; }
    const float ga = gamma[c] * rstd;
    ca[k] = ga * sc1;
    cb[k] = (beta[c] - mean * ga) * sc1 + sh;
    cm[k] = mean; cr[k] = rstd;
  }
  const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long)b * HW * C + c0);
  const f32x4* gb = reinterpret_cast<const f32x4*>(dy + (long)b * HW * C + c0);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  f32x4 xh[MAXR], d[MAXR];                         // xhat and du = dy * mask * act'(u), kept for the second pass
  f32x4 r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int hw = ry + i * R;
    const long off = (long)min(hw, HW - 1) * C4 + cq;              // unconditional (clamped) loads: no branch per row
    const f32x4 tx = xb[off], td = gb[off];
    xh[i] = hw < HW ? tx : f32x4{0, 0, 0, 0};
    d[i] = hw < HW ? td : f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int hw = ry + i * R;
    if (hw >= HW) continue;
    const f32x4 v = xh[i];
    const f32x4 u = v * ca + cb;
    f32x4 dd = d[i];
    if (drop_p > 0.f) dd *= dropout_keep4(seed, ((uint64_t)b * HW + hw) * C4 + (c0 >> 2) + cq, drop_p, inv_keep);
    if (silu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) dd[k] *= silu_grad_f(u[k]);
    }
    xh[i] = (v - cm) * cr;
    d[i] = dd;
    r1 += dd;
    r2 += dd * xh[i];
    __builtin_amdgcn_sched_barrier(0);            // one row at a time (see the forward kernel)
  }
  float* p1 = sm + (ry * Cc + cq * 4) * 2;
#pragma unroll
  for (int k = 0; k < 4; ++k) { p1[2 * k] = r1[k]; p1[2 * k + 1] = r2[k]; }
  __syncthreads();
  float* ct = sm + R * Cc * 2;                     // [Cc][2]
  float* gmv = ct + Cc * 2;                        // [Gc][2]
  for (int cl = threadIdx.x; cl < Cc; cl += blockDim.x) {
    float R1 = 0.f, R2 = 0.f;
    for (int r = 0; r < R; ++r) { R1 += sm[(r * Cc + cl) * 2]; R2 += sm[(r * Cc + cl) * 2 + 1]; }
    const int c = c0 + cl;
    tot[((long)b * C + c) * 2] = R1;
    tot[((long)b * C + c) * 2 + 1] = R2;
    const float sc1 = ss ? 1.f + ss[b * ss_bstride + c] : 1.f;
    const float gp = gamma[c] * sc1;
    ct[2 * cl] = gp * R1;
    ct[2 * cl + 1] = gp * R2;
    if (dss) {
      const long ds = ss_bstride ? ss_bstride : 2L * C;
      dss[(long)b * ds + c] = gamma[c] * R2 + beta[c] * R1;
      dss[(long)b * ds + C + c] = R1;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < Gc) {
    const int g = threadIdx.x;
    const float inv_n = 1.f / ((float)HW * (float)cpg);
    float a = 0.f, q = 0.f;
    for (int cl = g * cpg; cl < (g + 1) * cpg; ++cl) { a += ct[2 * cl]; q += ct[2 * cl + 1]; }
    gmv[2 * g] = a * inv_n; gmv[2 * g + 1] = q * inv_n;
  }
  __syncthreads();
  f32x4 c1, c2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = (cq * 4 + k) / cpg;
    c1[k] = cr[k] * gmv[2 * g];
    c2[k] = cr[k] * gmv[2 * g + 1];
  }
  (void)g0;
  f32x4* ob = reinterpret_cast<f32x4*>(dx + (long)b * HW * C + c0);
  const f32x4* ab = addend ? reinterpret_cast<const f32x4*>(addend + (long)b * HW * C + c0) : nullptr;
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int hw = ry + i * R;
    if (hw >= HW) continue;
    f32x4 o = ca * d[i] - c1 - xh[i] * c2;
    if (ab) o += ab[(long)hw * C4 + cq];
    am = gn_amax4(am, o);
    ob[(long)hw * C4 + cq] = o;
  }
  gn_amax_commit(am, amax);
}

// Plan of the one-launch path: slab width Cc (whole groups, whole channel quads, <= 128 channels), workgroup size and rows per
// thread (<= 14).
struct GnPlan { int Cc, threads, rows; };
inline GnPlan gn_fused_plan(int HW, int C, int G, bool backward) {
  const int cpg = C / G;
  GnPlan best{0, 0, 0};
  // Above 16x16 the streaming multi-pass kernels win: measured at 32x32 (tools/bench_gn.py, bs=128) the register-resident
  // forward is 8-60 % SLOWER (1024-thread workgroups, one per CU: loads, reduction and stores of a CU no longer overlap; and the
  // second read of a 100 MB map is served by the 256 MB Infinity Cache anyway), the backward equal within 1 %.
  if (HW > 256) return best;
  const int threads = 256;
  (void)backward;
  for (int k = 1; k <= G; ++k) {
    if (G % k) continue;
    const int Cc = cpg * k;
    if (Cc & 3) continue;                    // whole float4 quads (a quad may straddle two groups: the kernels map channels, not quads)
    if (Cc > 128) break;
    const int R = threads / (Cc / 4);
    if (R < 1) continue;
    const int rows = (HW + R - 1) / R;
    if (rows <= 14) best = GnPlan{Cc, (Cc / 4) * R, rows};      // widest slab that fits: longest contiguous row segments
  }
  return best;
}

inline bool gn_shape_ok(int B, int HW, int C, int G) {
  return B > 0 && HW > 0 && C > 0 && G > 0 && (C % 4) == 0 && (C % G) == 0 && C / 4 <= 1024;
}
inline int gn_threads(int C) { return (C / 4) * gn_rows_par(C); }

}  // namespace

int g_gn_fused = 1;       // 0: always the multi-pass kernels (diagnostic; tools only)
extern "C" int adm_gn_fused(int on) { const int old = g_gn_fused; if (on == 0 || on == 1) g_gn_fused = on; return old; }

extern "C" int adm_gn_splits(int HW, int C) {
  (void)C;
  // one workgroup per (image, split): 64-row splits up to 1024 pixels (the CIFAR maps: <= 16 splits, unchanged), then 256-row
  // splits so that the 128x128 latents / 512x512 autoencoder maps of the SR recipe still launch thousands of workgroups
  int s = HW / 64;
  if (s < 1) s = 1;
  if (s > 16) s = HW / 256;
  if (s > 1024) s = 1024;
  return s;
}

extern "C" int adm_gn_stats(const float* x, float* stats, double* ws, int B, int HW, int C, int G, float eps,
                            hipStream_t stream) {
  if (!x || !stats || !ws || !gn_shape_ok(B, HW, C, G)) return ADM_EINVAL;
  int S = adm_gn_splits(HW, C), rows = adm_cdiv(HW, S), R = gn_rows_par(C);
  size_t smem = (size_t)R * C * 3 * sizeof(float);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(B, S), dim3(gn_threads(C)), smem, stream, x, ws, HW, C, G, rows);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(adm_cdiv(B * G, 256)), dim3(256), 0, stream, ws, stats, B * G, G, S, HW,
                     rows, C / G, eps);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

static int gn_apply_impl(const float* x, const float* stats, const float* gamma, const float* beta, const float* ss, long ss_bstride,
                         float* y, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed, int out_bf16,
                         hipStream_t stream, float* amax = nullptr) {
  if (!x || !stats || !gamma || !beta || !y || !gn_shape_ok(B, HW, C, G) || drop_p < 0.f || drop_p >= 1.f)
    return ADM_EINVAL;
  int S = adm_gn_splits(HW, C), rows = adm_cdiv(HW, S);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(B, S), dim3(gn_threads(C)), 0, stream, x, stats, gamma, beta, ss,
                     ss_bstride, y, HW, C, G, rows, silu, drop_p, seed, out_bf16, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_gn_apply(const float* x, const float* stats, const float* gamma, const float* beta,
                            const float* ss, long ss_bstride, float* y, int B, int HW, int C, int G, int silu,
                            float drop_p, uint64_t seed, hipStream_t stream) {
  return gn_apply_impl(x, stats, gamma, beta, ss, ss_bstride, y, B, HW, C, G, silu, drop_p, seed, 0, stream);
}

static int gn_fwd_impl(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
                       long ss_bstride, float* y, int B, int HW, int C, int G, float eps, int silu, float drop_p, uint64_t seed,
                       int out_bf16, hipStream_t stream, float* amax = nullptr);

// adm_gn_fwd that also raises *amax (a device float the caller zeroed) to max |y|: the scale basis of the fp16-format conv that
// consumes y (adm_conv_fwd_wino2d_h3)
extern "C" int adm_gn_fwd_amax(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
                               long ss_bstride, float* y, float* amax, int B, int HW, int C, int G, float eps, int silu, float drop_p,
                               uint64_t seed, hipStream_t stream) {
  return gn_fwd_impl(x, stats, ws, gamma, beta, ss, ss_bstride, y, B, HW, C, G, eps, silu, drop_p, seed, 0, stream, amax);
}

extern "C" int adm_gn_fwd(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
                          long ss_bstride, float* y, int B, int HW, int C, int G, float eps, int silu, float drop_p,
                          uint64_t seed, hipStream_t stream) {
  return gn_fwd_impl(x, stats, ws, gamma, beta, ss, ss_bstride, y, B, HW, C, G, eps, silu, drop_p, seed, 0, stream);
}

// the same with y stored as bf16 (round to nearest even): y16[B][HW][C] -- the bf16-storage mode (BASELINE configs[2]) feeds the
// following conv's A operand without a conversion pass; statistics, affine map and activation stay f32
extern "C" int adm_gn_fwd_bf16out(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
                                  long ss_bstride, void* y16, int B, int HW, int C, int G, float eps, int silu, float drop_p,
                                  uint64_t seed, hipStream_t stream) {
  return gn_fwd_impl(x, stats, ws, gamma, beta, ss, ss_bstride, static_cast<float*>(y16), B, HW, C, G, eps, silu, drop_p, seed, 1,
                     stream);
}

static int gn_fwd_impl(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
                       long ss_bstride, float* y, int B, int HW, int C, int G, float eps, int silu, float drop_p, uint64_t seed,
                       int out_bf16, hipStream_t stream, float* amax) {
  if (!x || !stats || !ws || !gamma || !beta || !y || !gn_shape_ok(B, HW, C, G) || drop_p < 0.f || drop_p >= 1.f)
    return ADM_EINVAL;
  const GnPlan pl = g_gn_fused ? gn_fused_plan(HW, C, G, false) : GnPlan{0, 0, 0};
  if (pl.Cc == 0) {
    int rc = adm_gn_stats(x, stats, ws, B, HW, C, G, eps, stream);
    if (rc != ADM_OK) return rc;
    return gn_apply_impl(x, stats, gamma, beta, ss, ss_bstride, y, B, HW, C, G, silu, drop_p, seed, out_bf16, stream, amax);
  }
  const int Cc = pl.Cc, R = pl.threads / (Cc / 4), Gc = Cc / (C / G);
  const size_t smem = ((size_t)R * Cc * 2 + (size_t)((Gc * 2 + 1) & ~1)) * sizeof(float) + (size_t)Cc * sizeof(double);
  const dim3 grid(C / Cc, B), block(pl.threads);
#define GN_FWD(MAXR, THREADS)                                                                                                 \
  hipLaunchKernelGGL((gn_fused_fwd_kernel<MAXR, THREADS>), grid, block, smem, stream, x, gamma, beta, ss, ss_bstride, y, stats, \
                     HW, C, G, Cc, eps, silu, drop_p, seed, out_bf16, amax)
  if (pl.rows <= 2) GN_FWD(2, 256);
  else if (pl.rows <= 8) GN_FWD(8, 256);
  else GN_FWD(14, 256);
#undef GN_FWD
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// red layout: part [B][S][C][2] | tot [B][C][2] | gm [B][G][2]   (floats)
extern "C" int adm_gn_bwd_add(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                              const float* ss, long ss_bstride, const float* addend, float* dx, float* dss, float* dgamma,
                              float* dbeta, float* red, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed,
                              hipStream_t stream);
extern "C" int adm_gn_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                          const float* ss, long ss_bstride, float* dx, float* dss, float* dgamma, float* dbeta,
                          float* red, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed,
                          hipStream_t stream) {
  return adm_gn_bwd_add(x, dy, stats, gamma, beta, ss, ss_bstride, nullptr, dx, dss, dgamma, dbeta, red, B, HW, C, G, silu,
                        drop_p, seed, stream);
}

static int gn_bwd_add_impl(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                           const float* ss, long ss_bstride, const float* addend, float* dx, float* dss, float* dgamma,
                           float* dbeta, float* red, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed,
                           hipStream_t stream, float* amax) {

  if (!x || !dy || !stats || !gamma || !beta || !dx || !red || !gn_shape_ok(B, HW, C, G)) return ADM_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return ADM_EINVAL;
  int S = adm_gn_splits(HW, C), rows = adm_cdiv(HW, S), R = gn_rows_par(C);
  float* part = red;
  float* tot = part + (long)B * S * C * 2;
  float* gm = tot + (long)B * C * 2;
  const GnPlan pl = g_gn_fused ? gn_fused_plan(HW, C, G, true) : GnPlan{0, 0, 0};
  if (pl.Cc) {       // one launch (+ the parameter-gradient reduction over the batch)
    const int Cc = pl.Cc, Rf = pl.threads / (Cc / 4), Gc = Cc / (C / G);
    const size_t smf = ((size_t)Rf * Cc * 2 + (size_t)Cc * 2 + (size_t)Gc * 2) * sizeof(float);
    const dim3 grid(C / Cc, B), block(pl.threads);
#define GN_BWD(MAXR, THREADS)                                                                                                   \
  hipLaunchKernelGGL((gn_fused_bwd_kernel<MAXR, THREADS>), grid, block, smf, stream, x, dy, stats, gamma, beta, ss, ss_bstride, \
                     addend, dx, tot, dss, HW, C, G, Cc, silu, drop_p, seed, amax)
    if (pl.rows <= 2) GN_BWD(2, 256);
    else if (pl.rows <= 8) GN_BWD(8, 256);
    else GN_BWD(14, 256);
#undef GN_BWD
    if (dgamma)
      hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(adm_cdiv(C, 32)), dim3(256), 0, stream, tot, ss, ss_bstride, dgamma,
                         dbeta, B, C);
    ADM_CHECK_LAUNCH();
    return ADM_OK;
  }
  size_t smem = (size_t)R * C * 2 * sizeof(float);
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(B, S), dim3(gn_threads(C)), smem, stream, x, dy, stats, gamma, beta,
                     ss, ss_bstride, part, HW, C, G, rows, silu, drop_p, seed);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(B), dim3(256), (size_t)C * 2 * sizeof(float), stream, part, gamma,
                     beta, ss, ss_bstride, tot, gm, dss, S, HW, C, G);
  if (dgamma)
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(adm_cdiv(C, 32)), dim3(256), 0, stream, tot, ss, ss_bstride, dgamma,
                       dbeta, B, C);
  hipLaunchKernelGGL(gn_bwd_dx_kernel, dim3(B, S), dim3(gn_threads(C)), 0, stream, x, dy, stats, gamma, beta, ss,
                     ss_bstride, gm, addend, dx, HW, C, G, rows, silu, drop_p, seed, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_gn_bwd_add(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                              const float* ss, long ss_bstride, const float* addend, float* dx, float* dss, float* dgamma,
                              float* dbeta, float* red, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed,
                              hipStream_t stream) {
  return gn_bwd_add_impl(x, dy, stats, gamma, beta, ss, ss_bstride, addend, dx, dss, dgamma, dbeta, red, B, HW, C, G, silu, drop_p, seed,
                         stream, nullptr);
}
// ... that also raises the device float *amax (zeroed by the caller) to max |dx|
extern "C" int adm_gn_bwd_add_amax(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                                   const float* ss, long ss_bstride, const float* addend, float* dx, float* dss, float* dgamma,
                                   float* dbeta, float* red, float* amax, int B, int HW, int C, int G, int silu, float drop_p,
                                   uint64_t seed, hipStream_t stream) {
  return gn_bwd_add_impl(x, dy, stats, gamma, beta, ss, ss_bstride, addend, dx, dss, dgamma, dbeta, red, B, HW, C, G, silu, drop_p, seed,
                         stream, amax);
}

// The batch reduction of d(gamma) / d(beta) for every GroupNorm layer of a backward pass in ONE launch (adm_gn_bwd / adm_gn_bwd_add
// called with dgamma = dbeta = NULL leave the per-image sums at red + adm_gn_bwd_tot_offset(...) floats; they must stay alive until
// this runs).  table = device array of `rows` rows of 8 int64: {tot, ss (0: none), ss_bstride, dgamma, dbeta, B, C, block_begin};
// a row owns ceil(C / 32) blocks; total_blocks = their sum.  dgamma / dbeta are accumulated (+=).
extern "C" int adm_gn_bwd_param_table(const long* table, int rows, long total_blocks, hipStream_t stream) {
  if (!table || rows <= 0 || total_blocks <= 0 || total_blocks >= (1L << 31)) return ADM_EINVAL;
  hipLaunchKernelGGL(gn_bwd_param_table_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, table, rows);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
