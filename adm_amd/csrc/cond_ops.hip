// Bandwidth-bound operators of the conditional super-resolution denoiser (BASELINE
// configs[4]): everything /root/reference/unet/cond_unet_sd.py needs beyond the unconditional UNet's kernels, NHWC fp32,
// forward and backward.
//
//   weight standardisation            WeightStandardizedConv2d            cond_unet_sd.py:344-357
//   channel LayerNorm (gain only)     LayerNorm / PreNorm                 :359-378
//   BatchNorm2d (batch / running)     RelationNet.input_conv{1,2}[1]      :247-254
//   bilinear resize                   F.interpolate(..., 'bilinear')      :196, 235, 824
//   ReLU / GELU (+ dropout)           Mlp, time_mlp                       :132-150, 696-701
//   Gaussian Fourier features         GaussianFourierProjection           :396-405
//   SpatialAtt for any map size       SpatialAtt                          :112-130 (the 16x16 bottleneck: HW = 256)
//   col2im of a transposed conv       data gradient of Downsample = Conv2d(C, C', 4, 2, 1)   :341-342
//
// All reductions are two-stage with a fixed combination order (no atomics): results are bitwise reproducible.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

inline int co_grid(long n, int per_thread = 1, int cap = 4096) {
  long b = (n + 256L * per_thread - 1) / (256L * per_thread);
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {      // red: >= 4 floats of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------- weight standardisation
// one workgroup per output channel (row of K = Cin * kh * kw values)
__global__ __launch_bounds__(256) void ws_fwd_kernel(const float* __restrict__ w, float* __restrict__ wn,
                                                     float* __restrict__ stats, int K, float eps) {
  __shared__ float red[4];
  const float* r = w + (long)blockIdx.x * K;
  float s = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) s += r[i];
  const float mean = block_sum_256(s, red) / (float)K;
  float q = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) { const float d = r[i] - mean; q += d * d; }
  const float var = block_sum_256(q, red) / (float)K;
  const float rstd = rsqrtf(var + eps);
  for (int i = threadIdx.x; i < K; i += 256) wn[(long)blockIdx.x * K + i] = (r[i] - mean) * rstd;
  if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}

// dw = rstd (dwn - mean(dwn) - wn mean(dwn wn))
__global__ __launch_bounds__(256) void ws_bwd_kernel(const float* __restrict__ w, const float* __restrict__ stats,
                                                     const float* __restrict__ dwn, float* __restrict__ dw, int K,
                                                     int accumulate) {
  __shared__ float red[4];
  const long o = (long)blockIdx.x * K;
  const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) {
    const float g = dwn[o + i];
    a += g;
    b += g * (w[o + i] - mean) * rstd;
  }
  const float m1 = block_sum_256(a, red) / (float)K;
  const float m2 = block_sum_256(b, red) / (float)K;
  for (int i = threadIdx.x; i < K; i += 256) {
    const float v = rstd * (dwn[o + i] - m1 - (w[o + i] - mean) * rstd * m2);
    dw[o + i] = accumulate ? dw[o + i] + v : v;
  }
}

// ---------------------------------------------------------------- channel LayerNorm (per pixel), gain g, no bias
// one wave per pixel row; lane l owns channels 4l + 256 j.  C % 4 == 0, C <= 1024.
template <bool BWD>
__global__ __launch_bounds__(256) void lnc_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                  const float* __restrict__ g, float* __restrict__ out,
                                                  double* __restrict__ dg_part, long M, int C, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int C4 = C >> 2;
  f32x4 gv[4], dgacc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c4 = lane + 64 * j;
    gv[j] = (c4 < C4) ? reinterpret_cast<const f32x4*>(g)[c4] : f32x4{0, 0, 0, 0};
    dgacc[j] = f32x4{0, 0, 0, 0};
  }
  const float invC = 1.f / (float)C;
  for (long m = (long)blockIdx.x * 4 + wid; m < M; m += (long)gridDim.x * 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + m * C);
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c4 = lane + 64 * j;
      v[j] = (c4 < C4) ? xr[c4] : f32x4{0, 0, 0, 0};
      s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < C4) {
        const f32x4 d = v[j] - mean;
        q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invC + eps);
    f32x4* orow = reinterpret_cast<f32x4*>(out + m * C);
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < C4) orow[c4] = (v[j] - mean) * rstd * gv[j];
      }
    } else {
      const f32x4* dr = reinterpret_cast<const f32x4*>(dy + m * C);
      f32x4 xh[4], gd[4];
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c4 = lane + 64 * j;
        xh[j] = f32x4{0, 0, 0, 0}; gd[j] = f32x4{0, 0, 0, 0};
        if (c4 < C4) {
          const f32x4 d = dr[c4];
          xh[j] = (v[j] - mean) * rstd;
          gd[j] = d * gv[j];
          dgacc[j] += d * xh[j];
          a += gd[j][0] + gd[j][1] + gd[j][2] + gd[j][3];
          b += gd[j][0] * xh[j][0] + gd[j][1] * xh[j][1] + gd[j][2] * xh[j][2] + gd[j][3] * xh[j][3];
        }
      }
      const float m1 = wave_sum(a) * invC, m2 = wave_sum(b) * invC;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < C4) orow[c4] = (gd[j] - m1 - xh[j] * m2) * rstd;
      }
    }
  }
  if (BWD) {      // per-workgroup partial of dg: waves combined through LDS in wave order
    __shared__ float sm[4][1024];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c4 = lane + 64 * j;
      if (c4 < C4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sm[wid][c4 * 4 + k] = dgacc[j][k];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      dg_part[(long)blockIdx.x * C + c] = (double)sm[0][c] + (double)sm[1][c] + (double)sm[2][c] + (double)sm[3][c];
  }
}

// out[c] (+)= sum_blocks part[block][c]  (fixed order)
__global__ void colpart_final_kernel(const double* __restrict__ part, float* __restrict__ out, int nblocks, int C, int stride,
                                     int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0;
  for (int b = 0; b < nblocks; ++b) a += part[(long)b * stride + c];
  out[c] = accumulate ? out[c] + (float)a : (float)a;
}

// ---------------------------------------------------------------- per-channel column moments (BatchNorm)
// MODE 0: (sum x, sum x^2)   MODE 1: (sum dy, sum dy * xhat) with xhat = (x - mean[c]) * rstd[c]
// thread map: C4 columns x R rows in flight; fp64 accumulators (the data is HBM-bound: the fp64 adds are free)
template <int MODE>
__global__ __launch_bounds__(256) void colmom_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ mr, double* __restrict__ part, long M, int C,
                                                     long rows_per_block) {
  extern __shared__ double smd[];      // [R][C][2]
  const int C4 = C >> 2, R = blockDim.x / C4;
  const int cq = threadIdx.x % C4, ry = threadIdx.x / C4;
  const long m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  f32x4 mean = {0, 0, 0, 0}, rstd = {0, 0, 0, 0};
  if (MODE == 1 && ry < R) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { mean[k] = mr[2 * (cq * 4 + k)]; rstd[k] = mr[2 * (cq * 4 + k) + 1]; }
  }
  if (ry < R) {
    for (long m = m0 + ry; m < m1; m += R) {
      const f32x4 v = reinterpret_cast<const f32x4*>(x + m * C)[cq];
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += (double)v[k]; b[k] += (double)v[k] * (double)v[k]; }
      } else {
        const f32x4 d = reinterpret_cast<const f32x4*>(dy + m * C)[cq];
        const f32x4 xh = (v - mean) * rstd;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += (double)d[k]; b[k] += (double)(d[k] * xh[k]); }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { smd[(ry * C + cq * 4 + k) * 2] = a[k]; smd[(ry * C + cq * 4 + k) * 2 + 1] = b[k]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    double t = 0.0;
    for (int r = 0; r < R; ++r) t += smd[(long)r * C * 2 + i];
    part[(long)blockIdx.x * 2 * C + i] = t;
  }
}

// BatchNorm statistics from the partials: mr[c] = (mean, rstd); running statistics updated in place (momentum, unbiased var)
__global__ void bn_finalize_kernel(const double* __restrict__ part, float* __restrict__ mr, float* __restrict__ run_mean,
                                   float* __restrict__ run_var, int nblocks, int C, double M, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, q = 0.0;
  for (int b = 0; b < nblocks; ++b) { a += part[((long)b * C + c) * 2]; q += part[((long)b * C + c) * 2 + 1]; }
  const double mean = a / M;
  double var = q / M - mean * mean;
  if (var < 0.0) var = 0.0;
  mr[2 * c] = (float)mean;
  mr[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * (M / (M > 1.0 ? M - 1.0 : 1.0)));
  }
}

// eval mode: mr from the running statistics
__global__ void bn_running_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var, float* __restrict__ mr,
                                  int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mr[2 * c] = run_mean[c];
  mr[2 * c + 1] = rsqrtf(run_var[c] + eps);
}

// y = (x - mean) rstd gamma + beta
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mr, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, long M, int C) {
  const int C4 = C >> 2;
  const long total = M * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i], o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cq * 4 + k;
      o[k] = (v[k] - mr[2 * c]) * mr[2 * c + 1] * gamma[c] + beta[c];
    }
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
}

// training: dx = gamma rstd (dy - S1/M - xhat S2/M); eval (frozen statistics): dx = gamma rstd dy.  sums[c] = (S1, S2) floats
__global__ void bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mr,
                                 const float* __restrict__ gamma, const float* __restrict__ sums, float* __restrict__ dx, long M,
                                 int C, int training) {
  const int C4 = C >> 2;
  const long total = M * C4;
  const float invM = 1.f / (float)M;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i], d = reinterpret_cast<const f32x4*>(dy)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cq * 4 + k;
      const float rstd = mr[2 * c + 1], xh = (v[k] - mr[2 * c]) * rstd;
      o[k] = training ? gamma[c] * rstd * (d[k] - sums[2 * c] * invM - xh * sums[2 * c + 1] * invM) : gamma[c] * rstd * d[k];
    }
    reinterpret_cast<f32x4*>(dx)[i] = o;
  }
}

// sums[c] = (S1, S2) as floats from the fp64 partials; dgamma (+)= S2, dbeta (+)= S1
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, float* __restrict__ sums, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int nblocks, int C, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, q = 0.0;
  for (int b = 0; b < nblocks; ++b) { a += part[((long)b * C + c) * 2]; q += part[((long)b * C + c) * 2 + 1]; }
  sums[2 * c] = (float)a; sums[2 * c + 1] = (float)q;
  if (dgamma) { dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q; dbeta[c] = accumulate ? dbeta[c] + (float)a : (float)a; }
}

// ---------------------------------------------------------------- bilinear resize (NHWC)
__device__ __forceinline__ void bil_src(int d, int in, int out, int align, int& i0, int& i1, float& lam) {
  float s;
  if (align) s = out > 1 ? (float)d * ((float)(in - 1) / (float)(out - 1)) : 0.f;
  else { s = ((float)d + 0.5f) * ((float)in / (float)out) - 0.5f; if (s < 0.f) s = 0.f; }
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  lam = s - (float)i0;
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo,
                                    int C, int ldy, int coff, int align) {
  const int C4 = C >> 2;
  const long total = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    int y0, y1, x0, x1; float ly, lx;
    bil_src(oy, Hi, Ho, align, y0, y1, ly);
    bil_src(ox, Wi, Wo, align, x0, x1, lx);
    const f32x4* xb = reinterpret_cast<const f32x4*>(x + (long)b * Hi * Wi * C);
    const f32x4 v00 = xb[((long)y0 * Wi + x0) * C4 + cq], v01 = xb[((long)y0 * Wi + x1) * C4 + cq];
    const f32x4 v10 = xb[((long)y1 * Wi + x0) * C4 + cq], v11 = xb[((long)y1 * Wi + x1) * C4 + cq];
    const f32x4 top = v00 * (1.f - lx) + v01 * lx, bot = v10 * (1.f - lx) + v11 * lx;
    const f32x4 r = top * (1.f - ly) + bot * ly;
    float* dst = y + (((long)b * Ho + oy) * Wo + ox) * ldy + coff + cq * 4;
    if (coff & 3) { dst[0] = r[0]; dst[1] = r[1]; dst[2] = r[2]; dst[3] = r[3]; }      // e.g. behind the 3 latent channels of the stem input
    else *reinterpret_cast<f32x4*>(dst) = r;
  }
}

// gather form of the adjoint: one workgroup per SOURCE pixel; threads = C4 channel quads x R destination rows in flight.
// A destination index d touches source s iff i0(d) == s or i1(d) == s, i.e. src(d) in (s - 1, s + 1): scan that range.
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Hi, int Wi,
                                                           int Ho, int Wo, int C, int lddy, int coff, int align) {
  extern __shared__ float smb[];       // [R][C]
  const int C4 = C >> 2, R = blockDim.x / C4;
  const int cq = threadIdx.x % C4, ry = threadIdx.x / C4;
  const int sx = blockIdx.x % Wi, sy = (blockIdx.x / Wi) % Hi, b = blockIdx.x / (Wi * Hi);
  auto range = [&](int s, int in, int out, int& lo, int& hi) {
    const float r = align ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f) : (float)in / (float)out;
    if (r <= 0.f) { lo = 0; hi = out - 1; return; }
    const float off = align ? 0.f : 0.5f;      // src(d) = (d + off) r - off
    lo = (int)floorf(((float)(s - 1) + off) / r - off) - 1;
    hi = (int)ceilf(((float)(s + 1) + off) / r - off) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
  };
  int ylo, yhi, xlo, xhi;
  range(sy, Hi, Ho, ylo, yhi);
  range(sx, Wi, Wo, xlo, xhi);
  f32x4 acc = {0, 0, 0, 0};
  if (ry < R) {
    for (int oy = ylo + ry; oy <= yhi; oy += R) {
      int y0, y1; float ly;
      bil_src(oy, Hi, Ho, align, y0, y1, ly);
      const float wy = (y0 == sy ? 1.f - ly : 0.f) + (y1 == sy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1; float lx;
        bil_src(ox, Wi, Wo, align, x0, x1, lx);
        const float wx = (x0 == sx ? 1.f - lx : 0.f) + (x1 == sx ? lx : 0.f);
        if (wx == 0.f) continue;
        acc += *reinterpret_cast<const f32x4*>(dy + (((long)b * Ho + oy) * Wo + ox) * lddy + coff + cq * 4) * (wy * wx);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) smb[ry * C + cq * 4 + k] = acc[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t = 0.f;
    for (int r = 0; r < R; ++r) t += smb[r * C + c];
    dx[(((long)b * Hi + sy) * Wi + sx) * C + c] = t;
  }
}

// ---------------------------------------------------------------- ReLU / GELU (+ dropout)
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float u) {
  return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.39894228040143268f * __expf(-0.5f * u * u);
}

// act: 0 = identity, 1 = ReLU, 2 = GELU (exact erf form, nn.GELU()).  bwd = 0: y = drop(act(x)); bwd = 1: dx = dy mask act'(x)
__global__ void act_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, long n4, int act,
                           float drop_p, uint64_t seed, int bwd) {
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o;
    if (!bwd) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = act == 1 ? fmaxf(v[k], 0.f) : act == 2 ? gelu_f(v[k]) : v[k];
    } else {
      const f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = d[k] * (act == 1 ? (v[k] > 0.f ? 1.f : 0.f) : act == 2 ? gelu_grad_f(v[k]) : 1.f);
    }
    if (drop_p > 0.f) o *= dropout_keep4(seed, (uint64_t)i, drop_p, inv_keep);
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// out[b][0..D) = sin(2 pi x_b W), out[b][D..2D) = cos(...)
__global__ void fourier_kernel(const float* __restrict__ x, const float* __restrict__ W, float* __restrict__ out, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  const float p = x[b] * W[d] * 2.f * 3.14159265358979323846f;
  out[(long)b * 2 * D + d] = sinf(p);
  out[(long)b * 2 * D + D + d] = cosf(p);
}

// ---------------------------------------------------------------- SpatialAtt for any map size (HW <= 2560: six HW-float arrays in < 64 KB of LDS)
// att = channel 0 of [B][HW][ldatt]; q_i = qw a_i + qb, k_j = kw a_j + kb; av_i = sum_j softmax_j(q_i k_j) a_j;
// gate g_i = softsign(av_i); y = g_i h + xres.  The HW x HW score matrix is rank one: nothing is stored, every pass
// recomputes exp(q_i k_j - m_i) from the row maximum m_i (attained at the largest or smallest k: no search needed).
__global__ __launch_bounds__(256) void spatt_fwd_kernel(const float* __restrict__ att, int ldatt, const float* __restrict__ qk,
                                                        const float* __restrict__ h, const float* __restrict__ xres,
                                                        float* __restrict__ y, float* __restrict__ gate, int HW, int C) {
  extern __shared__ float sa[];        // a[HW] | g[HW]
  float* a_s = sa; float* g_s = sa + HW;
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float qw = qk[0], qb = qk[1], kw = qk[2], kb = qk[3];
  float kmx = -3.4e38f, kmn = 3.4e38f;
  for (int i = tid; i < HW; i += 256) {
    const float a = att[((long)b * HW + i) * ldatt];
    a_s[i] = a;
    const float k = kw * a + kb;
    kmx = fmaxf(kmx, k); kmn = fminf(kmn, k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { kmx = fmaxf(kmx, __shfl_xor(kmx, o, 64)); kmn = fminf(kmn, __shfl_xor(kmn, o, 64)); }
  if ((tid & 63) == 0) { red[tid >> 6] = kmx; red[4 + (tid >> 6)] = kmn; }
  __syncthreads();
  kmx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  kmn = fminf(fminf(red[4], red[5]), fminf(red[6], red[7]));
  for (int i = tid; i < HW; i += 256) {
    const float q = qw * a_s[i] + qb;
    const float m = q >= 0.f ? q * kmx : q * kmn;
    float l = 0.f, s = 0.f;
    for (int j = 0; j < HW; ++j) {
      const float e = __expf(q * (kw * a_s[j] + kb) - m);
      l += e; s += e * a_s[j];
    }
    const float av = s / l;
    g_s[i] = av / (1.f + fabsf(av));
    if (gate) gate[(long)b * HW * 2 + 2 * i] = av;          // saved for backward: av_i and the row's log-sum-exp
    if (gate) gate[(long)b * HW * 2 + 2 * i + 1] = m + __logf(l);
  }
  __syncthreads();
  const int C4 = C >> 2;
  const f32x4* hb = reinterpret_cast<const f32x4*>(h + (long)b * HW * C);
  const f32x4* xb = reinterpret_cast<const f32x4*>(xres + (long)b * HW * C);
  f32x4* yb = reinterpret_cast<f32x4*>(y + (long)b * HW * C);
  for (long i = tid; i < (long)HW * C4; i += 256) yb[i] = hb[i] * g_s[i / C4] + xb[i];
}

// gate = [B][HW][2] = (av_i, lse_i) from the forward.  Outputs: dh = g dy; datt (channel 0, others zero); dqk_part[b][4]
__global__ __launch_bounds__(256) void spatt_bwd_kernel(const float* __restrict__ att, int ldatt, const float* __restrict__ qk,
                                                        const float* __restrict__ h, const float* __restrict__ dy,
                                                        const float* __restrict__ gate, float* __restrict__ dh,
                                                        float* __restrict__ datt, float* __restrict__ dqk_part, int HW, int C) {
  extern __shared__ float sa[];        // a[HW] | da[HW] (= dL/dav_i) | av[HW] | lse[HW] | dq[HW] | dk[HW]
  float* a_s = sa; float* da_s = sa + HW; float* av_s = sa + 2 * HW; float* ls_s = sa + 3 * HW;
  float* dq_s = sa + 4 * HW; float* dk_s = sa + 5 * HW;
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float qw = qk[0], qb = qk[1], kw = qk[2], kb = qk[3];
  for (int i = tid; i < HW; i += 256) {
    a_s[i] = att[((long)b * HW + i) * ldatt];
    av_s[i] = gate[(long)b * HW * 2 + 2 * i];
    ls_s[i] = gate[(long)b * HW * 2 + 2 * i + 1];
  }
  __syncthreads();
  // dg_i = sum_c dy h (one wave per pixel), dh = g dy
  const int C4 = C >> 2;
  for (int i = wid; i < HW; i += 4) {
    const float av = av_s[i], g = av / (1.f + fabsf(av));
    const f32x4* hb = reinterpret_cast<const f32x4*>(h + ((long)b * HW + i) * C);
    const f32x4* db = reinterpret_cast<const f32x4*>(dy + ((long)b * HW + i) * C);
    f32x4* ob = reinterpret_cast<f32x4*>(dh + ((long)b * HW + i) * C);
    float acc = 0.f;
    for (int c = lane; c < C4; c += 64) {
      const f32x4 d = db[c], hv = hb[c];
      ob[c] = d * g;
      acc += d[0] * hv[0] + d[1] * hv[1] + d[2] * hv[2] + d[3] * hv[3];
    }
    acc = wave_sum(acc);
    if (lane == 0) { const float d1 = 1.f + fabsf(av); da_s[i] = acc / (d1 * d1); }
  }
  __syncthreads();
  // row pass (query i): dq_i = sum_j ds_ij k_j with ds_ij = p_ij da_i (a_j - av_i)
  for (int i = tid; i < HW; i += 256) {
    const float q = qw * a_s[i] + qb, ls = ls_s[i], av = av_s[i], da = da_s[i];
    float dq = 0.f;
    for (int j = 0; j < HW; ++j) {
      const float k = kw * a_s[j] + kb;
      dq += __expf(q * k - ls) * (a_s[j] - av) * k;
    }
    dq_s[i] = dq * da;
  }
  // column pass (key j): dk_j = sum_i ds_ij q_i ; direct datt_j = sum_i p_ij da_i
  for (int j = tid; j < HW; j += 256) {
    const float k = kw * a_s[j] + kb, aj = a_s[j];
    float dk = 0.f, dat = 0.f;
    for (int i = 0; i < HW; ++i) {
      const float q = qw * a_s[i] + qb;
      const float p = __expf(q * k - ls_s[i]) * da_s[i];
      dk += p * (aj - av_s[i]) * q;
      dat += p;
    }
    dk_s[j] = dk;
    float* o = datt + ((long)b * HW + j) * ldatt;
    o[0] = dat;                       // completed below, after dq_s of ALL rows is visible
    for (int c = 1; c < ldatt; ++c) o[c] = 0.f;
  }
  __syncthreads();
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int i = tid; i < HW; i += 256) {
    float* o = datt + ((long)b * HW + i) * ldatt;
    o[0] += qw * dq_s[i] + kw * dk_s[i];
    s0 += dq_s[i] * a_s[i]; s1 += dq_s[i];
    s2 += dk_s[i] * a_s[i]; s3 += dk_s[i];
  }
  s0 = block_sum_256(s0, red); s1 = block_sum_256(s1, red); s2 = block_sum_256(s2, red); s3 = block_sum_256(s3, red);
  if (tid == 0) { float* o = dqk_part + 4 * (long)b; o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; }
}

__global__ void dqk_sum_kernel(const float* __restrict__ part, float* __restrict__ dqk, int B) {
  const int k = threadIdx.x;
  if (k >= 4) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += part[4 * (long)b + k];
  dqk[k] += s;
}

// ---------------------------------------------------------------- col2im of a transposed convolution
// dx[b][iy][ix][c] = sum over taps (ky, kx) with (iy + pad_lo - ky) = stride oy, (ix + pad_lo - kx) = stride ox, (oy, ox) in the
// output grid:  col[(b, oy, ox)][ky * ks + kx][c]        (col = dy x W^T, adm_pack_weight_tconv)
__global__ void col2im_kernel(const float* __restrict__ col, float* __restrict__ dx, int B, int Hin, int Win, int Ho, int Wo, int C,
                              int ks, int stride, int pad_lo) {
  const int C4 = C >> 2;
  const long total = (long)B * Hin * Win * C4;
  const int taps = ks * ks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long t = i / C4;
    const int ix = (int)(t % Win); t /= Win;
    const int iy = (int)(t % Hin);
    const int b = (int)(t / Hin);
    f32x4 acc = {0, 0, 0, 0};
    for (int ky = 0; ky < ks; ++ky) {
      const int ny = iy + pad_lo - ky;
      if (ny < 0 || ny % stride) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < ks; ++kx) {
        const int nx = ix + pad_lo - kx;
        if (nx < 0 || nx % stride) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        acc += reinterpret_cast<const f32x4*>(col + ((((long)b * Ho + oy) * Wo + ox) * taps + ky * ks + kx) * C)[cq];
      }
    }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int adm_ws_fwd(const float* w, float* wn, float* stats, int O, int K, float eps, hipStream_t stream) {
  if (!w || !wn || !stats || O <= 0 || K <= 0) return ADM_EINVAL;
  hipLaunchKernelGGL(ws_fwd_kernel, dim3(O), dim3(256), 0, stream, w, wn, stats, K, eps);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_ws_bwd(const float* w, const float* stats, const float* dwn, float* dw, int O, int K, int accumulate,
                          hipStream_t stream) {
  if (!w || !stats || !dwn || !dw || O <= 0 || K <= 0) return ADM_EINVAL;
  hipLaunchKernelGGL(ws_bwd_kernel, dim3(O), dim3(256), 0, stream, w, stats, dwn, dw, K, accumulate);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_lnc_blocks(long M) { return M > 0 ? co_grid(M, 1, 2048) * 1 : 0; }

extern "C" int adm_lnc_fwd(const float* x, const float* g, float* y, long M, int C, float eps, hipStream_t stream) {
  if (!x || !g || !y || M <= 0 || C <= 0 || (C & 3) || C > 1024) return ADM_EINVAL;
  hipLaunchKernelGGL((lnc_kernel<false>), dim3(co_grid(M, 1, 8192)), dim3(256), 0, stream, x, nullptr, g, y, nullptr, M, C, eps);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// part = workspace of adm_lnc_blocks(M) * C doubles; dg (+)= sum over pixels of dy * xhat
extern "C" int adm_lnc_bwd(const float* x, const float* dy, const float* g, float* dx, float* dg, double* part, long M, int C,
                           float eps, int accumulate, hipStream_t stream) {
  if (!x || !dy || !g || !dx || !dg || !part || M <= 0 || C <= 0 || (C & 3) || C > 1024) return ADM_EINVAL;
  const int blocks = adm_lnc_blocks(M);
  hipLaunchKernelGGL((lnc_kernel<true>), dim3(blocks), dim3(256), 0, stream, x, dy, g, dx, part, M, C, eps);
  hipLaunchKernelGGL(colpart_final_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, part, dg, blocks, C, C, accumulate);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

namespace {
inline bool bn_ok(long M, int C) { return M > 0 && C > 0 && (C & 3) == 0 && C <= 1024; }
inline int bn_threads(int C) { int r = 256 / (C / 4); if (r < 1) r = 1; return (C / 4) * r; }
inline int bn_blocks_for(long M) { long b = (M + 511) / 512; return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }
}  // namespace

extern "C" int adm_bn_blocks(long M) { return M > 0 ? bn_blocks_for(M) : 0; }

// training != 0: batch statistics (mr = (mean, rstd) per channel is an OUTPUT, the running statistics are updated in place);
// training == 0: mr is derived from the running statistics.  part = adm_bn_blocks(M) * 2 C doubles.
extern "C" int adm_bn_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float* mr,
                          float* y, double* part, long M, int C, float eps, float momentum, int training, hipStream_t stream) {
  if (!x || !gamma || !beta || !run_mean || !run_var || !mr || !y || !bn_ok(M, C)) return ADM_EINVAL;
  if (training) {
    if (!part) return ADM_EINVAL;
    const int blocks = bn_blocks_for(M);
    const long rows = (M + blocks - 1) / blocks;
    const int th = bn_threads(C), R = th / (C / 4);
    hipLaunchKernelGGL((colmom_kernel<0>), dim3(blocks), dim3(th), (size_t)R * C * 2 * sizeof(double), stream, x, nullptr, nullptr,
                       part, M, C, rows);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, part, mr, run_mean, run_var, blocks, C,
                       (double)M, eps, momentum);
  } else {
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, run_mean, run_var, mr, C, eps);
  }
  hipLaunchKernelGGL(bn_apply_kernel, dim3(co_grid(M * (C / 4))), dim3(256), 0, stream, x, mr, gamma, beta, y, M, C);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// sums = workspace of 2 C floats; dgamma / dbeta may both be NULL
extern "C" int adm_bn_bwd(const float* x, const float* dy, const float* mr, const float* gamma, float* dx, float* dgamma,
                          float* dbeta, double* part, float* sums, long M, int C, int training, int accumulate,
                          hipStream_t stream) {
  if (!x || !dy || !mr || !gamma || !dx || !part || !sums || !bn_ok(M, C)) return ADM_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return ADM_EINVAL;
  const int blocks = bn_blocks_for(M);
  const long rows = (M + blocks - 1) / blocks;
  const int th = bn_threads(C), R = th / (C / 4);
  hipLaunchKernelGGL((colmom_kernel<1>), dim3(blocks), dim3(th), (size_t)R * C * 2 * sizeof(double), stream, x, dy, mr, part, M, C,
                     rows);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, part, sums, dgamma, dbeta, blocks, C,
                     accumulate);
  hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(co_grid(M * (C / 4))), dim3(256), 0, stream, x, dy, mr, gamma, sums, dx, M, C,
                     training);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// y[b][oy][ox][coff .. coff + C) of rows of ldy floats = bilinear(x[b][Hi][Wi][C]); align_corners as F.interpolate
extern "C" int adm_bilinear_fwd(const float* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int ldy, int coff,
                                int align_corners, hipStream_t stream) {
  if (!x || !y || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 3) || coff < 0 || coff + C > ldy)
    return ADM_EINVAL;
  if ((coff & 3) == 0 && (ldy & 3)) return ADM_EINVAL;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(co_grid((long)B * Ho * Wo * (C / 4))), dim3(256), 0, stream, x, y, B, Hi, Wi, Ho, Wo,
                     C, ldy, coff, align_corners);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_bilinear_bwd(const float* dy, float* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int lddy, int coff,
                                int align_corners, hipStream_t stream) {
  if (!dy || !dx || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 3) || C > 1024 || (lddy & 3) ||
      (coff & 3) || coff + C > lddy || (long)B * Hi * Wi >= (1L << 31))
    return ADM_EINVAL;
  const int th = bn_threads(C), R = th / (C / 4);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((unsigned)((long)B * Hi * Wi)), dim3(th), (size_t)R * C * sizeof(float), stream, dy,
                     dx, Hi, Wi, Ho, Wo, C, lddy, coff, align_corners);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// act: 0 identity, 1 ReLU, 2 GELU; optional dropout (same stateless hash as the GroupNorm kernels).  n % 4 == 0.
extern "C" int adm_act_fwd(const float* x, float* y, long n, int act, float drop_p, uint64_t seed, hipStream_t stream) {
  if (!x || !y || n <= 0 || (n & 3) || act < 0 || act > 2 || drop_p < 0.f || drop_p >= 1.f) return ADM_EINVAL;
  hipLaunchKernelGGL(act_kernel, dim3(co_grid(n / 4)), dim3(256), 0, stream, x, nullptr, y, n / 4, act, drop_p, seed, 0);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_act_bwd(const float* x, const float* dy, float* dx, long n, int act, float drop_p, uint64_t seed,
                           hipStream_t stream) {
  if (!x || !dy || !dx || n <= 0 || (n & 3) || act < 0 || act > 2 || drop_p < 0.f || drop_p >= 1.f) return ADM_EINVAL;
  hipLaunchKernelGGL(act_kernel, dim3(co_grid(n / 4)), dim3(256), 0, stream, x, dy, dx, n / 4, act, drop_p, seed, 1);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_fourier_features(const float* x, const float* W, float* out, int B, int D, hipStream_t stream) {
  if (!x || !W || !out || B <= 0 || D <= 0) return ADM_EINVAL;
  hipLaunchKernelGGL(fourier_kernel, dim3((B * D + 255) / 256), dim3(256), 0, stream, x, W, out, B, D);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// gate = [B][HW][2] workspace written by the forward (av_i, log-sum-exp_i) and read by the backward.  HW <= 2560.
extern "C" int adm_spatial_att_big_fwd(const float* att, int ldatt, const float* qk, const float* h, const float* xres, float* y,
                                       float* gate, int B, int HW, int C, hipStream_t stream) {
  if (!att || !qk || !h || !xres || !y || B <= 0 || HW <= 0 || HW > 2560 || (C & 3)) return ADM_EINVAL;
  hipLaunchKernelGGL(spatt_fwd_kernel, dim3(B), dim3(256), (size_t)2 * HW * sizeof(float), stream, att, ldatt, qk, h, xres, y, gate,
                     HW, C);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_spatial_att_big_bwd(const float* att, int ldatt, const float* qk, const float* h, const float* dy,
                                       const float* gate, float* dh, float* datt, float* dqk, float* dqk_part, int B, int HW, int C,
                                       hipStream_t stream) {
  if (!att || !qk || !h || !dy || !gate || !dh || !datt || !dqk || !dqk_part || B <= 0 || HW <= 0 || HW > 2560 || (C & 3))
    return ADM_EINVAL;
  hipLaunchKernelGGL(spatt_bwd_kernel, dim3(B), dim3(256), (size_t)6 * HW * sizeof(float), stream, att, ldatt, qk, h, dy, gate, dh,
                     datt, dqk_part, HW, C);
  hipLaunchKernelGGL(dqk_sum_kernel, dim3(1), dim3(64), 0, stream, dqk_part, dqk, B);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_col2im(const float* col, float* dx, int B, int Hin, int Win, int Ho, int Wo, int C, int ks, int stride,
                          int pad_lo, hipStream_t stream) {
  if (!col || !dx || B <= 0 || Hin <= 0 || Win <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 3) || ks < 1 || ks > 7 || stride < 1 ||
      pad_lo < 0)
    return ADM_EINVAL;
  hipLaunchKernelGGL(col2im_kernel, dim3(co_grid((long)B * Hin * Win * (C / 4))), dim3(256), 0, stream, col, dx, B, Hin, Win, Ho, Wo,
                     C, ks, stride, pad_lo);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
