// HBM-bound elementwise / small-reduction kernels of the DDM hot path: resampling, layout +
// preconditioning, time embedding, SpatialAtt gate, analytic-schedule updates, loss, optimiser.
// All are simple grid-stride kernels with 16-byte accesses where the layout allows.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

inline int ew_grid(long n, int per_thread = 1) {
  long b = (n / per_thread + 255) / 256;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  return (int)b;
}

// ---------------------------------------------------------------- resample
__global__ void down2x_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int B, int Ho, int Wo, int C4,
                              float scale, int acc) {
  long total = (long)B * Ho * Wo * C4;
  const int Wi = Wo * 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % C4;
    long t = i / C4;
    int ox = t % Wo; t /= Wo;
    int oy = t % Ho;
    int b = t / Ho;
    long base = (((long)b * Ho * 2 + oy * 2) * Wi + ox * 2) * C4 + c;
    f32x4 v = (x[base] + x[base + C4] + x[base + (long)Wi * C4] + x[base + (long)Wi * C4 + C4]) * scale;
    y[i] = acc ? y[i] + v : v;
  }
}
__global__ void up2x_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int B, int Hi, int Wi, int C4,
                            float scale, int acc) {
  const int Ho = Hi * 2, Wo = Wi * 2;
  long total = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % C4;
    long t = i / C4;
    int ox = t % Wo; t /= Wo;
    int oy = t % Ho;
    int b = t / Ho;
    f32x4 v = x[(((long)b * Hi + (oy >> 1)) * Wi + (ox >> 1)) * C4 + c] * scale;
    y[i] = acc ? y[i] + v : v;
  }
}

// ---------------------------------------------------------------- layout / preconditioning
template <typename T>
__global__ void nchw_to_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ mul, long mul_bstride,
                                    float* __restrict__ y, int B, int C, int HW, int Cpad, float* __restrict__ amax) {
  long total = (long)B * HW;
  float am = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int p = i % HW;
    int b = i / HW;
    float m = mul ? mul[b * mul_bstride] : 1.f;
    float* o = y + i * Cpad;
    for (int c = 0; c < Cpad; ++c) {
      const float v = c < C ? m * (float)x[((long)b * C + c) * HW + p] : 0.f;
      am = fmaxf(am, fabsf(v));
      o[c] = v;
    }
  }
  adm_amax_commit(am, amax);      // (amax may be null) bound vector of the UNet's input: the stem conv then runs on the fp16 format
}
template <typename T>
__global__ void precond_out_kernel(const T* __restrict__ x, const float* __restrict__ f, int ldf,
                                   const float* __restrict__ a, const float* __restrict__ s, long cbs,
                                   float* __restrict__ out, int B, int C, int HW) {
  long total = (long)B * C * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int p = i % HW;
    long t = i / HW;
    int c = t % C;
    int b = t / C;
    float v = s[b * cbs] * f[((long)b * HW + p) * ldf + c];
    if (x) v += a[b * cbs] * (float)x[i];        // (x == nullptr: the plain scaled NHWC -> NCHW transpose, adjoint of nchw_to_nhwc)
    out[i] = v;
  }
}
__global__ void precond_out_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ s, long cbs,
                                       float* __restrict__ df, int ldf, int B, int C, int HW, float* __restrict__ amax) {
  long total = (long)B * HW;
  float am = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int p = i % HW;
    int b = i / HW;
    float sv = s[b * cbs];
    float* o = df + i * ldf;
    for (int c = 0; c < ldf; ++c) {
      const float v = c < C ? sv * dout[((long)b * C + c) * HW + p] : 0.f;
      am = fmaxf(am, fabsf(v));
      o[c] = v;
    }
  }
  adm_amax_commit(am, amax);      // (amax may be null) bound vector of the output conv's dy
}
template <typename T>
__global__ void axpby_b_kernel(const T* __restrict__ x, const float* __restrict__ y, const float* __restrict__ a,
                               const float* __restrict__ s, long cbs, float* __restrict__ out, int B, long n) {
  long total = (long)B * n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = i / n;
    float v = s[b * cbs] * y[i];
    if (x) v += a[b * cbs] * (float)x[i];
    out[i] = v;
  }
}

__global__ void pos_embedding_kernel(const float* __restrict__ t, float* __restrict__ emb, int B, int C) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int half = C / 2;
  if (i >= B * half) return;
  int b = i / half, k = i - b * half;
  float f = powf(1.0f / 10000.0f, (float)k / (float)half);
  float ang = t[b] * f;
  emb[(long)b * C + k] = cosf(ang);
  emb[(long)b * C + half + k] = sinf(ang);
}

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = silu_f(x[i]);
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * silu_grad_f(x[i]);
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}
// y = a + b + c (c may be null), 16 bytes per lane: the gradient sum of a tensor with several consumers (ops.fanout)
__global__ void add3_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, const f32x4* __restrict__ c, f32x4* __restrict__ y,
                            long n4, float* __restrict__ amax) {
  float am = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = a[i] + b[i];
    if (c) v += c[i];
    am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    y[i] = v;
  }
  adm_amax_commit(am, amax);                     // max |y| (a bound vector) for a fp16-format conv that consumes the sum (conv_wino2d_x6.hip)
}
// y[m] = (a[m] | scale_b * b[m]) in one launch (dir 0: torch.cat of two NHWC tensors along the channels), or its adjoint (dir 1:
// a[m] = y[m][:Ca], b[m] = scale_b * y[m][Ca:]); amax (dir 0, may be null): bound vector of what was written
__global__ void concat2_kernel(float* __restrict__ a, int Ca4, float* __restrict__ b, int Cb4, float* __restrict__ y, long M, float scale_b,
                               int dir, float* __restrict__ amax) {
  const int C4 = Ca4 + Cb4;
  const long total = M * C4;
  float am = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const long m = i / C4;
    f32x4* yp = reinterpret_cast<f32x4*>(y) + i;
    const bool first = c < Ca4;
    f32x4* sp = first ? reinterpret_cast<f32x4*>(a) + m * Ca4 + c : reinterpret_cast<f32x4*>(b) + m * Cb4 + (c - Ca4);
    if (dir == 0) {
      f32x4 v = *sp;
      if (!first) v *= scale_b;
      *yp = v;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    } else {
      f32x4 v = *yp;
      if (!first) v *= scale_b;
      *sp = v;
    }
  }
  if (dir == 0) adm_amax_commit(am, amax);
}
__global__ void copy_channels_kernel(const float* __restrict__ src, int lds_, int src_off, float* __restrict__ dst,
                                     int ldd, int dst_off, long M, int C4, float scale, int acc, float* __restrict__ amax) {
  long total = M * C4;
  float am = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % C4;
    long m = i / C4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + m * lds_ + src_off + c * 4) * scale;
    f32x4* d = reinterpret_cast<f32x4*>(dst + m * ldd + dst_off + c * 4);
    if (acc) v += *d;
    *d = v;
    am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  adm_amax_commit(am, amax);                     // (amax may be null) bound vector of what was written
}

// ---------------------------------------------------------------- SpatialAtt gate
// one block per batch element; HW <= 64
__global__ __launch_bounds__(256) void spatial_att_fwd_kernel(const float* __restrict__ att, int ldatt,
                                                              const float* __restrict__ qk,
                                                              const float* __restrict__ h,
                                                              const float* __restrict__ xres, float* __restrict__ y,
                                                              int HW, int C) {
  __shared__ float a_s[64], g_s[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < HW) a_s[tid] = att[((long)b * HW + tid) * ldatt];
  __syncthreads();
  if (tid < HW) {
    const float qw = qk[0], qb = qk[1], kw = qk[2], kb = qk[3];
    float q = qw * a_s[tid] + qb;
    float m = -INFINITY;
    for (int j = 0; j < HW; ++j) m = fmaxf(m, q * (kw * a_s[j] + kb));
    float l = 0.f, acc = 0.f;
    for (int j = 0; j < HW; ++j) {
      float e = __expf(q * (kw * a_s[j] + kb) - m);
      l += e; acc += e * a_s[j];
    }
    float a = acc / l;
    g_s[tid] = a / (1.f + fabsf(a));
  }
  __syncthreads();
  const int C4 = C >> 2;
  const f32x4* hb = reinterpret_cast<const f32x4*>(h + (long)b * HW * C);
  const f32x4* xb = reinterpret_cast<const f32x4*>(xres + (long)b * HW * C);
  f32x4* yb = reinterpret_cast<f32x4*>(y + (long)b * HW * C);
  for (int i = tid; i < HW * C4; i += blockDim.x) yb[i] = hb[i] * g_s[i / C4] + xb[i];
}

__global__ __launch_bounds__(256) void spatial_att_bwd_kernel(const float* __restrict__ att, int ldatt,
                                                              const float* __restrict__ qk,
                                                              const float* __restrict__ h,
                                                              const float* __restrict__ dy, float* __restrict__ dh,
                                                              float* __restrict__ datt, float* __restrict__ dqk,
                                                              float* __restrict__ dqk_part, int HW, int C) {
  __shared__ float a_s[64], g_s[64], av_s[64], dg_s[64], da_s[64], dq_s[64], dk_s[64], p_s[64][65];
  __shared__ float part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float qw = qk[0], qb = qk[1], kw = qk[2], kb = qk[3];
  if (tid < HW) a_s[tid] = att[((long)b * HW + tid) * ldatt];
  __syncthreads();
  if (tid < HW) {
    float q = qw * a_s[tid] + qb;
    float m = -INFINITY;
    for (int j = 0; j < HW; ++j) m = fmaxf(m, q * (kw * a_s[j] + kb));
    float l = 0.f, acc = 0.f;
    for (int j = 0; j < HW; ++j) {
      float e = __expf(q * (kw * a_s[j] + kb) - m);
      p_s[tid][j] = e; l += e; acc += e * a_s[j];
    }
    float inv = 1.f / l;
    for (int j = 0; j < HW; ++j) p_s[tid][j] *= inv;
    float a = acc * inv;
    av_s[tid] = a;
    g_s[tid] = a / (1.f + fabsf(a));
  }
  __syncthreads();
  // dh = g * dy ; dg_i = sum_c dy h   (4 threads per pixel row when HW <= 64)
  const int C4 = C >> 2;
  const f32x4* hb = reinterpret_cast<const f32x4*>(h + (long)b * HW * C);
  const f32x4* gb = reinterpret_cast<const f32x4*>(dy + (long)b * HW * C);
  f32x4* ob = reinterpret_cast<f32x4*>(dh + (long)b * HW * C);
  const int tpr = blockDim.x / 64;                 // threads per row
  {
    int row = tid / tpr, sub = tid % tpr;
    float acc = 0.f;
    if (row < HW)
      for (int c = sub; c < C4; c += tpr) {
        f32x4 d = gb[row * C4 + c], hv = hb[row * C4 + c];
        ob[row * C4 + c] = d * g_s[row];
        acc += d[0] * hv[0] + d[1] * hv[1] + d[2] * hv[2] + d[3] * hv[3];
      }
    part[tid] = acc;
  }
  __syncthreads();
  if (tid < HW) {
    float dg = 0.f;
    for (int s = 0; s < tpr; ++s) dg += part[tid * tpr + s];
    float d1 = 1.f + fabsf(av_s[tid]);
    dg_s[tid] = dg;
    da_s[tid] = dg / (d1 * d1);
  }
  __syncthreads();
  if (tid < HW) {      // as query i = tid: dq_i = sum_j ds_ij k_j
    float dq = 0.f;
    for (int j = 0; j < HW; ++j) dq += p_s[tid][j] * da_s[tid] * (a_s[j] - av_s[tid]) * (kw * a_s[j] + kb);
    dq_s[tid] = dq;
  }
  if (tid >= 64 && tid < 64 + HW) {   // as key j: dk_j = sum_i ds_ij q_i ; datt_j (direct) = sum_i p_ij da_i
    int j = tid - 64;
    float dk = 0.f, dat = 0.f;
    for (int i = 0; i < HW; ++i) {
      float ds = p_s[i][j] * da_s[i] * (a_s[j] - av_s[i]);
      dk += ds * (qw * a_s[i] + qb);
      dat += p_s[i][j] * da_s[i];
    }
    dk_s[j] = dk;
    part[j] = dat;
  }
  __syncthreads();
  if (tid < HW) {
    float dat = part[tid] + qw * dq_s[tid] + kw * dk_s[tid];
    float* o = datt + ((long)b * HW + tid) * ldatt;
    o[0] = dat;
    for (int c = 1; c < ldatt; ++c) o[c] = 0.f;
  }
  if (tid == 0) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = 0; i < HW; ++i) {
      s0 += dq_s[i] * a_s[i]; s1 += dq_s[i];
      s2 += dk_s[i] * a_s[i]; s3 += dk_s[i];
    }
    if (dqk_part) {          // deterministic: per-image partials, summed in image order by dqk_reduce_kernel
      float* o = dqk_part + 4 * (long)b;
      o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    } else {
      atomicAdd(&dqk[0], s0); atomicAdd(&dqk[1], s1); atomicAdd(&dqk[2], s2); atomicAdd(&dqk[3], s3);
    }
  }
}

__global__ void dqk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dqk, int B) {
  const int k = threadIdx.x;
  if (k >= 4) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += part[4 * (long)b + k];
  dqk[k] += s;
}

// ---------------------------------------------------------------- analytic schedule
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                const float* __restrict__ t, float* __restrict__ xt, int B, long n, int schedule) {
  long total = (long)B * n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float tt = t[i / n];
    float g = schedule == 0 ? sqrtf(tt) : tt;
    float x = x0[i];
    float c = -1.f * x;
    xt[i] = x + c * tt + g * noise[i];
  }
}

__global__ __launch_bounds__(256) void ddm_loss_kernel(const float* __restrict__ cp, const float* __restrict__ np_,
                                                       const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const float* __restrict__ w, float* __restrict__ per_sample,
                                                       float* __restrict__ dc, float* __restrict__ dn, float gscale,
                                                       long n) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float w1 = w[2 * b], w2 = w[2 * b + 1];
  float acc = 0.f;
  for (long i = blockIdx.y * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.y * blockDim.x) {
    long k = (long)b * n + i;
    float e1 = cp[k] + x0[k];          // C_pred - C, C = -x0
    float e2 = np_[k] - noise[k];
    acc += w1 * e1 * e1 + w2 * e2 * e2;
    if (dc) { dc[k] = gscale * 2.f * w1 * e1; dn[k] = gscale * 2.f * w2 * e2; }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&per_sample[b], red[0] + red[1] + red[2] + red[3]);
}

// Latent variant (ddm_const_2.py:527-596): the weighted SSE above plus  w3 * sum |x_rec - x0|,
// x_rec = x_t - C_pred t - t noise_pred  (const_2: g(t) = t), evaluated in the reference's operation order.
// w = [B][3] = (w1, w2, w3);  per_sample[b] gets the SSE part, per_l1[b] the UN-weighted L1 sum.
__global__ __launch_bounds__(256) void ddm_loss_latent_kernel(const float* __restrict__ cp, const float* __restrict__ np_,
                                                              const float* __restrict__ x0, const float* __restrict__ noise,
                                                              const float* __restrict__ xt, const float* __restrict__ t,
                                                              const float* __restrict__ w, float* __restrict__ per_sample,
                                                              float* __restrict__ per_l1, float* __restrict__ dc,
                                                              float* __restrict__ dn, float gscale, long n, int schedule,
                                                              int use_l1) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  const float w1 = w[3 * b], w2 = w[3 * b + 1], w3 = w[3 * b + 2], tb = t[b];
  const float gt = schedule == 0 ? sqrtf(tb) : tb;       // 'const': sqrt(t) (ddm_const.py:290-293); 'const_2': t
  float acc = 0.f, l1 = 0.f;
  for (long i = blockIdx.y * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.y * blockDim.x) {
    long k = (long)b * n + i;
    float c = cp[k], e = np_[k];
    float e1 = c + x0[k];
    float e2 = e - noise[k];
    float d = ((xt[k] - c * tb) - gt * e) - x0[k];
    // use_l1 (ddm_const_2.py:556-559): loss_simple = [w1 (SSE_C + L1_C) + w2 (SSE_eps + L1_eps)] / 2
    acc += use_l1 ? 0.5f * (w1 * (e1 * e1 + fabsf(e1)) + w2 * (e2 * e2 + fabsf(e2))) : w1 * e1 * e1 + w2 * e2 * e2;
    l1 += fabsf(d);
    if (dc) {
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      float g1 = 2.f * w1 * e1, g2 = 2.f * w2 * e2;
      if (use_l1) {
        g1 = 0.5f * (g1 + w1 * (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f)));
        g2 = 0.5f * (g2 + w2 * (e2 > 0.f ? 1.f : (e2 < 0.f ? -1.f : 0.f)));
      }
      dc[k] = gscale * (g1 - w3 * tb * sg);
      dn[k] = gscale * (g2 - w3 * gt * sg);
    }
  }
  acc = wave_sum(acc); l1 = wave_sum(l1);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; red[4 + (threadIdx.x >> 6)] = l1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&per_sample[b], red[0] + red[1] + red[2] + red[3]);
    atomicAdd(&per_l1[b], red[4] + red[5] + red[6] + red[7]);
  }
}

__global__ void sampler_step_kernel(double* __restrict__ x, const float* __restrict__ cp, const float* __restrict__ np_,
                                    double t_cur, double t_next, double g_cur, double g_next, int clip_x0,
                                    double scale_input, int last, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double c = (double)cp[i], e = (double)np_[i];
    double x0 = x[i] - c * t_cur - e * g_cur;
    if (clip_x0) x0 = fmin(fmax(x0, -scale_input), scale_input);
    double xn = x0 + c * t_next + e * g_next;
    if (last) {
      xn = fmin(fmax(xn, -scale_input), scale_input);
      if (scale_input != 1.0) xn = xn / scale_input;
      xn = (xn + 1.0) * 0.5;
    }
    x[i] = xn;
  }
}

// Stochastic reverse step (ddm_const.py:296-303, 410-414 / ddm_const_2.py:185-197, 324-328): per image b
//   x0 = x - C t - g(t) eps_model ; [clamp] ; C' = -x0 ;
//   const  : mean = x + C'(t-s) - C' t - s/sqrt(t) eps_model,     sigma = sqrt(s (t-s) / t)
//   const_2: mean = x - C' s - (2 s t - s^2)/t eps_model,         sigma = sqrt(2 s t - s^2) (t-s)/t
//   x <- mean + sigma * z        (z = injected N(0,1) draw, fp64 state)
__global__ void sampler_step_stochastic_kernel(double* __restrict__ x, const float* __restrict__ cp,
                                               const float* __restrict__ np_, const double* __restrict__ z,
                                               const double* __restrict__ t, const double* __restrict__ s_,
                                               int schedule, int clip_x0, double scale_input, int last, long n,
                                               long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / n;
    const double tt = t[b], s = s_[b];
    const double c = (double)cp[i], e = (double)np_[i], xv = x[i];
    const double g = schedule == 0 ? sqrt(tt) : tt;
    double x0 = xv - c * tt - g * e;
    if (clip_x0) x0 = fmin(fmax(x0, -scale_input), scale_input);
    const double c2 = -x0;
    double mean, sigma;
    if (schedule == 0) {
      mean = xv + c2 * (tt - s) - c2 * tt - s / sqrt(tt) * e;
      sigma = sqrt(s * (tt - s) / tt);
    } else {
      mean = xv - c2 * s - (2.0 * s * tt - s * s) / tt * e;
      sigma = sqrt(2.0 * s * tt - s * s) * (tt - s) / tt;
    }
    double xn = mean + sigma * z[i];
    if (last) {
      xn = fmin(fmax(xn, -scale_input), scale_input);
      if (scale_input != 1.0) xn = xn / scale_input;
      xn = (xn + 1.0) * 0.5;
    }
    x[i] = xn;
  }
}

// ---------------------------------------------------------------- optimiser
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, double* __restrict__ out,
                                                    double* __restrict__ part, long n) {
  __shared__ double red[4];
  float acc = 0.f;
  long n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = g4[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
  double d = wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = red[0] + red[1] + red[2] + red[3];
    if (part) part[blockIdx.x] = t; else atomicAdd(out, t);
  }
}

// second stage of the deterministic norm: one workgroup sums the per-block partials in a fixed order
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ part, double* __restrict__ out, int nblocks) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) a += part[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] += red[0];
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, float* __restrict__ ema, const double* __restrict__ sumsq, long n,
                             float lr, float b1, float b2, float eps, float wd, float max_norm, float bc1, float bc2s,
                             float ema_w, float grad_scale) {
  float clip = 1.f;
  if (sumsq && max_norm > 0.f) {
    float nrm = (float)sqrt(*sumsq) * grad_scale;
    clip = fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  const float gs = clip * grad_scale;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gv = g[i] * gs;
    float pv = p[i] * (1.f - lr * wd);
    float mv = b1 * m[i] + (1.f - b1) * gv;
    float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv; v[i] = vv;
    float denom = sqrtf(vv) / bc2s + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[i] = pv;
    if (ema) ema[i] += ema_w * (pv - ema[i]);
  }
}

}  // namespace

namespace {
// In-place row softmax of scale * s: one workgroup per row, the row cached in registers (cols <= 256 * 4 * 8).
// Used by the KL autoencoder's single-head mid-block attention (/root/reference/ddm/encoder_decoder.py:196-204),
// whose 4096 x 4096 score matrix per image is produced / consumed by the implicit-GEMM kernel.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int cols, long ld, float scale) {
  __shared__ float red[4];
  float* row = s + (long)blockIdx.x * ld;
  const int tid = threadIdx.x, nv = cols >> 2;
  f32x4 v[8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + i * 256;
    if (q < nv) {
      v[i] = *reinterpret_cast<const f32x4*>(row + 4 * q) * scale;
      mx = fmaxf(mx, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + i * 256;
    if (q < nv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][j] = __expf(v[i][j] - mx); sum += v[i][j]; }
    }
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + i * 256;
    if (q < nv) *reinterpret_cast<f32x4*>(row + 4 * q) = v[i] * inv;
  }
}

// Rows longer than the register budget (the 512x512 autoencoder of the SR recipe: 128 x 128 = 16384 tokens): three passes
// over the row (max, sum of exponentials, write); a 64 KiB row stays in the L2 between them.
__global__ __launch_bounds__(256) void softmax_rows_long_kernel(float* __restrict__ s, int cols, long ld, float scale) {
  __shared__ float red[4];
  float* row = s + (long)blockIdx.x * ld;
  const int tid = threadIdx.x, nv = cols >> 2;
  float mx = -INFINITY;
  for (int q = tid; q < nv; q += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q) * scale;
    mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int q = tid; q < nv; q += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q) * scale;
    sum += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int q = tid; q < nv; q += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q) * scale;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __expf(v[j] - mx) * inv;
    *reinterpret_cast<f32x4*>(row + 4 * q) = v;
  }
}

// z = mean + exp(0.5 * clamp(logvar, -30, 20)) * eps  over NHWC moments [M][ldm] = (mean[0:C] | logvar[C:2C])
// (DiagonalGaussianDistribution.sample, /root/reference/ddm/encoder_decoder.py:855-867); eps NULL -> the mode.
__global__ void posterior_sample_kernel(const float* __restrict__ mom, int ldm, const float* __restrict__ eps,
                                        float* __restrict__ z, int ldz, long M, int C, float zscale) {
  const long total = M * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C;
    const int c = (int)(i - m * C);
    const float mean = mom[m * ldm + c];
    float r = mean;
    if (eps) {
      const float lv = fminf(fmaxf(mom[m * ldm + C + c], -30.0f), 20.0f);
      r = mean + expf(0.5f * lv) * eps[i];
    }
    z[m * ldz + c] = r * zscale;
  }
}
}  // namespace

extern "C" int adm_softmax_rows(float* s, long rows, int cols, long ld, float scale, hipStream_t stream) {
  if (!s || rows <= 0 || rows >= (1L << 31) || cols <= 0 || (cols & 3) || cols > (1 << 20) || ld < cols || (ld & 3)) return ADM_EINVAL;
  if ((uintptr_t)s & 15) return ADM_EINVAL;
  if (cols > 8192) hipLaunchKernelGGL(softmax_rows_long_kernel, dim3((unsigned)rows), dim3(256), 0, stream, s, cols, ld, scale);
  else hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, s, cols, ld, scale);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_posterior_sample(const float* moments, int ldm, const float* eps, float* z, int ldz, long M, int C,
                                    float zscale, hipStream_t stream) {
  if (!moments || !z || M <= 0 || C <= 0 || ldm < 2 * C || ldz < C) return ADM_EINVAL;
  hipLaunchKernelGGL(posterior_sample_kernel, dim3(ew_grid(M * C)), dim3(256), 0, stream, moments, ldm, eps, z, ldz, M, C, zscale);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_version(void) { return 1; }

extern "C" int adm_resample2x(const float* x, float* y, int B, int H, int W, int C, int mode, float scale, int acc,
                              hipStream_t stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || (C & 3)) return ADM_EINVAL;
  if (mode == 0) {
    if ((H & 1) || (W & 1)) return ADM_EINVAL;
    long total = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(down2x_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, (const f32x4*)x, (f32x4*)y, B, H / 2,
                       W / 2, C / 4, scale, acc);
  } else if (mode == 1) {
    long total = (long)B * H * 2 * W * 2 * (C / 4);
    hipLaunchKernelGGL(up2x_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, (const f32x4*)x, (f32x4*)y, B, H, W,
                       C / 4, scale, acc);
  } else return ADM_EINVAL;
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_nchw_to_nhwc(const void* x, int x_is_f64, const float* mul, long mul_bstride, float* y, int B, int C,
                                int HW, int Cpad, hipStream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || Cpad < C) return ADM_EINVAL;
  int grid = ew_grid((long)B * HW);
  if (x_is_f64)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)x, mul, mul_bstride, y, B, C, HW, Cpad, static_cast<float*>(nullptr));
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, mul, mul_bstride, y, B, C, HW, Cpad, static_cast<float*>(nullptr));
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
// ... that also raises the bound vector amax (include/adm_hip.h) to max |y|
extern "C" int adm_nchw_to_nhwc_amax(const void* x, int x_is_f64, const float* mul, long mul_bstride, float* y, float* amax, int B, int C,
                                     int HW, int Cpad, hipStream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || Cpad < C) return ADM_EINVAL;
  int grid = ew_grid((long)B * HW);
  if (x_is_f64)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)x, mul, mul_bstride, y, B, C, HW, Cpad, amax);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, mul, mul_bstride, y, B, C, HW, Cpad, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_precond_out(const void* x, int x_is_f64, const float* f, int ldf, const float* a, const float* s,
                               long coef_bstride, float* out, int B, int C, int HW, hipStream_t stream) {
  if (!f || !s || !out || (x && !a) || B <= 0 || C <= 0 || HW <= 0 || ldf < C) return ADM_EINVAL;
  int grid = ew_grid((long)B * C * HW);
  if (x_is_f64)
    hipLaunchKernelGGL(precond_out_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)x, f, ldf, a, s, coef_bstride, out, B, C, HW);
  else
    hipLaunchKernelGGL(precond_out_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, f, ldf, a, s, coef_bstride, out, B, C, HW);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_precond_out_bwd(const float* dout, const float* s, long coef_bstride, float* df, int ldf, int B,
                                   int C, int HW, hipStream_t stream) {
  if (!dout || !s || !df || B <= 0 || C <= 0 || HW <= 0 || ldf < C) return ADM_EINVAL;
  hipLaunchKernelGGL(precond_out_bwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, stream, dout, s, coef_bstride, df, ldf, B, C, HW,
                     static_cast<float*>(nullptr));
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
// ... that also raises the bound vector amax to max |df|
extern "C" int adm_precond_out_bwd_amax(const float* dout, const float* s, long coef_bstride, float* df, int ldf, float* amax, int B,
                                        int C, int HW, hipStream_t stream) {
  if (!dout || !s || !df || B <= 0 || C <= 0 || HW <= 0 || ldf < C) return ADM_EINVAL;
  hipLaunchKernelGGL(precond_out_bwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, stream, dout, s, coef_bstride, df, ldf, B, C, HW, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_axpby_b(const void* x, int x_is_f64, const float* y, const float* a, const float* s,
                           long coef_bstride, float* out, int B, long n, hipStream_t stream) {
  if (!y || !s || !out || B <= 0 || n <= 0 || (x && !a)) return ADM_EINVAL;
  int grid = ew_grid((long)B * n);
  if (x_is_f64)
    hipLaunchKernelGGL(axpby_b_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)x, y, a, s, coef_bstride, out, B, n);
  else
    hipLaunchKernelGGL(axpby_b_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, y, a, s, coef_bstride, out, B, n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_pos_embedding(const float* t, float* emb, int B, int C, hipStream_t stream) {
  if (!t || !emb || B <= 0 || C <= 0 || (C & 1)) return ADM_EINVAL;
  hipLaunchKernelGGL(pos_embedding_kernel, dim3(adm_cdiv((long)B * C / 2, 256)), dim3(256), 0, stream, t, emb, B, C);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_silu_fwd(const float* x, float* y, long n, hipStream_t stream) {
  if (!x || !y || n <= 0) return ADM_EINVAL;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, y, n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
extern "C" int adm_silu_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t stream) {
  if (!x || !dy || !dx || n <= 0) return ADM_EINVAL;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, dy, dx, n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
extern "C" int adm_add(const float* a, const float* b, float* y, long n, hipStream_t stream) {
  if (!a || !b || !y || n <= 0) return ADM_EINVAL;
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, a, b, y, n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
extern "C" int adm_add3(const float* a, const float* b, const float* c, float* y, float* amax, long n, hipStream_t stream) {
  if (!a || !b || !y || n <= 0 || (n & 3) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)y) & 15)) return ADM_EINVAL;
  hipLaunchKernelGGL(add3_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, reinterpret_cast<const f32x4*>(a),
                     reinterpret_cast<const f32x4*>(b), reinterpret_cast<const f32x4*>(c), reinterpret_cast<f32x4*>(y), n / 4, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
extern "C" int adm_copy_channels(const float* src, int lds, int src_off, float* dst, int ldd, int dst_off, long M, int C,
                                 float scale, int acc, hipStream_t stream) {
  if (!src || !dst || M <= 0 || C <= 0 || (C & 3) || (lds & 3) || (ldd & 3) || (src_off & 3) || (dst_off & 3))
    return ADM_EINVAL;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, stream, src, lds, src_off, dst, ldd,
                     dst_off, M, C / 4, scale, acc, static_cast<float*>(nullptr));
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
// torch.cat((a, scale_b * b), channels) of two NHWC tensors in ONE launch: y[M][Ca + Cb]; amax (may be NULL): bound vector of y
extern "C" int adm_concat2(const float* a, int Ca, const float* b, int Cb, float* y, long M, float scale_b, float* amax, hipStream_t stream) {
  if (!a || !b || !y || M <= 0 || Ca <= 0 || Cb <= 0 || (Ca & 3) || (Cb & 3)) return ADM_EINVAL;
  hipLaunchKernelGGL(concat2_kernel, dim3(ew_grid(M * ((Ca + Cb) / 4))), dim3(256), 0, stream, const_cast<float*>(a), Ca / 4,
                     const_cast<float*>(b), Cb / 4, y, M, scale_b, 0, amax);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
// its adjoint in one launch: da[M][Ca] = dy[:, :Ca], db[M][Cb] = scale_b * dy[:, Ca:]
extern "C" int adm_split2(const float* dy, float* da, int Ca, float* db, int Cb, long M, float scale_b, hipStream_t stream) {
  if (!dy || !da || !db || M <= 0 || Ca <= 0 || Cb <= 0 || (Ca & 3) || (Cb & 3)) return ADM_EINVAL;
  hipLaunchKernelGGL(concat2_kernel, dim3(ew_grid(M * ((Ca + Cb) / 4))), dim3(256), 0, stream, da, Ca / 4, db, Cb / 4,
                     const_cast<float*>(dy), M, scale_b, 1, static_cast<float*>(nullptr));
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}


extern "C" int adm_spatial_att_fwd(const float* att, int ldatt, const float* qk, const float* h, const float* xres,
                                   float* y, int B, int HW, int C, hipStream_t stream) {
  if (!att || !qk || !h || !xres || !y || B <= 0 || HW <= 0 || HW > 64 || (C & 3)) return ADM_EINVAL;
  hipLaunchKernelGGL(spatial_att_fwd_kernel, dim3(B), dim3(256), 0, stream, att, ldatt, qk, h, xres, y, HW, C);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
extern "C" int adm_spatial_att_bwd(const float* att, int ldatt, const float* qk, const float* h, const float* dy,
                                   float* dh, float* datt, float* dqk, float* dqk_part, int B, int HW, int C,
                                   hipStream_t stream) {
  if (!att || !qk || !h || !dy || !dh || !datt || !dqk || B <= 0 || HW <= 0 || HW > 64 || (C & 3)) return ADM_EINVAL;
  hipLaunchKernelGGL(spatial_att_bwd_kernel, dim3(B), dim3(256), 0, stream, att, ldatt, qk, h, dy, dh, datt, dqk, dqk_part, HW,
                     C);
  if (dqk_part) hipLaunchKernelGGL(dqk_reduce_kernel, dim3(1), dim3(64), 0, stream, dqk_part, dqk, B);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_q_sample(const float* x0, const float* noise, const float* t, float* xt, int B, long n, int schedule,
                            hipStream_t stream) {
  if (!x0 || !noise || !t || !xt || B <= 0 || n <= 0 || (schedule != 0 && schedule != 1)) return ADM_EINVAL;
  hipLaunchKernelGGL(q_sample_kernel, dim3(ew_grid((long)B * n)), dim3(256), 0, stream, x0, noise, t, xt, B, n, schedule);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_ddm_loss(const float* c_pred, const float* n_pred, const float* x0, const float* noise,
                            const float* w, float* per_sample, float* d_c, float* d_n, float gscale, int B, long n,
                            hipStream_t stream) {
  if (!c_pred || !n_pred || !x0 || !noise || !w || !per_sample || B <= 0 || n <= 0) return ADM_EINVAL;
  if ((d_c == nullptr) != (d_n == nullptr)) return ADM_EINVAL;
  if (hipMemsetAsync(per_sample, 0, sizeof(float) * B, stream) != hipSuccess) return ADM_ELAUNCH;
  int chunks = (int)((n + 1023) / 1024);
  if (chunks > 64) chunks = 64;
  if (n <= 65536) chunks = 1;          // one workgroup per sample: the reported per-sample sums are bitwise reproducible
  hipLaunchKernelGGL(ddm_loss_kernel, dim3(B, chunks), dim3(256), 0, stream, c_pred, n_pred, x0, noise, w, per_sample,
                     d_c, d_n, gscale, n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_ddm_loss_latent(const float* c_pred, const float* n_pred, const float* x0, const float* noise,
                                   const float* xt, const float* t, const float* w, float* per_sample, float* per_l1,
                                   float* d_c, float* d_n, float gscale, int B, long n, int schedule, int use_l1,
                                   hipStream_t stream) {
  if (!c_pred || !n_pred || !x0 || !noise || !xt || !t || !w || !per_sample || !per_l1 || B <= 0 || n <= 0) return ADM_EINVAL;
  if (schedule != 0 && schedule != 1) return ADM_EINVAL;
  if ((d_c == nullptr) != (d_n == nullptr)) return ADM_EINVAL;
  if (hipMemsetAsync(per_sample, 0, sizeof(float) * B, stream) != hipSuccess) return ADM_ELAUNCH;
  if (hipMemsetAsync(per_l1, 0, sizeof(float) * B, stream) != hipSuccess) return ADM_ELAUNCH;
  int chunks = (int)((n + 1023) / 1024);
  if (chunks > 64) chunks = 64;
  if (n <= 65536) chunks = 1;
  hipLaunchKernelGGL(ddm_loss_latent_kernel, dim3(B, chunks), dim3(256), 0, stream, c_pred, n_pred, x0, noise, xt, t, w,
                     per_sample, per_l1, d_c, d_n, gscale, n, schedule, use_l1);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_sampler_step(double* x, const float* c_pred, const float* n_pred, double t_cur, double t_next,
                                int schedule, int clip_x0, double scale_input, int last, long n, hipStream_t stream) {
  if (!x || !c_pred || !n_pred || n <= 0 || (schedule != 0 && schedule != 1)) return ADM_EINVAL;
  double gc = schedule == 0 ? sqrt(t_cur) : t_cur, gn = schedule == 0 ? sqrt(t_next) : t_next;
  hipLaunchKernelGGL(sampler_step_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, c_pred, n_pred, t_cur, t_next, gc,
                     gn, clip_x0, scale_input, last, n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_sampler_step_stochastic(double* x, const float* c_pred, const float* n_pred, const double* z,
                                           const double* t, const double* s, int schedule, int clip_x0,
                                           double scale_input, int last, int B, long n, hipStream_t stream) {
  if (!x || !c_pred || !n_pred || !z || !t || !s || B <= 0 || n <= 0 || (schedule != 0 && schedule != 1)) return ADM_EINVAL;
  hipLaunchKernelGGL(sampler_step_stochastic_kernel, dim3(ew_grid((long)B * n)), dim3(256), 0, stream, x, c_pred, n_pred,
                     z, t, s, schedule, clip_x0, scale_input, last, n, (long)B * n);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_sumsq_blocks(long n) { return n > 0 ? ew_grid(n, 16) : 0; }

extern "C" int adm_sumsq(const float* g, double* sumsq, double* partials, long n, hipStream_t stream) {
  if (!g || !sumsq || n <= 0 || ((uintptr_t)g & 15)) return ADM_EINVAL;
  const int blocks = ew_grid(n, 16);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, g, sumsq, partials, n);
  if (partials) hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partials, sumsq, blocks);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_adamw_step(float* p, const float* g, float* m, float* v, float* ema, const double* sumsq, long n,
                              float lr, float beta1, float beta2, float eps, float wd, float max_norm, int step,
                              float ema_decay, float grad_scale, hipStream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return ADM_EINVAL;
  float bc1 = 1.f - powf(beta1, (float)step);
  float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n, 4)), dim3(256), 0, stream, p, g, m, v, ema, sumsq, n, lr, beta1,
                     beta2, eps, wd, max_norm, bc1, bc2s, 1.f - ema_decay, grad_scale);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
