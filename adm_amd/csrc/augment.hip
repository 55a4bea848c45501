// Geometric augmentation of x_start (use_augment: True): the execution half of the reference's AugmentPipe
// (/root/reference/ddm/augment.py:236-276) for the transforms DDM enables (ddm_const.py:179-180, ddm_const_2.py:112-113).
//
//   flips -> reflect-pad by the batch-wide margin -> x2 up-sample with the sym6 low-pass (zero stuffing + 12 taps,
//   separable) -> bilinear affine resample (affine_grid / grid_sample, align_corners = False, zeros outside) onto a
//   2(H+6) x 2(W+6) grid -> x2 down-sample with sym6 (stride 2) -> crop 3 pixels per side.
//
// Three launches over NCHW fp32 planes; the images are 3 x 32 x 32 (a few MB per batch), so these kernels are
// latency-, not bandwidth-bound and are kept simple.  The padding margin is data dependent (the reference reads it
// back to the host: `.ceil().to(int32)` feeding F.pad, a device sync every step); here it STAYS ON THE DEVICE: the
// kernels read the four ints from memory and the intermediate buffer is sized for the worst case (margin <= W-1), so
// the training step keeps no host round trip.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

__constant__ float kSym6[12] = {0.015404109327027373f, 0.0034907120842174702f, -0.11799011114819057f, -0.048311742585633f,
                                0.4910559419267466f,   0.787641141030194f,     0.3379294217276218f,   -0.07263752278646252f,
                                -0.021060292512300564f, 0.04472490177066578f,  0.0017677118642428036f, -0.007800708325034148f};

__device__ __forceinline__ int reflect(int i, int n) {      // F.pad(mode='reflect'): edge not repeated; |margin| <= n-1
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// up[nc][2Hp][2Wp] from img[nc][H][W]; margin = {mx0, my0, mx1, my1}; flips[n] = {xflip, yflip}
__global__ __launch_bounds__(256) void aug_up_kernel(const float* __restrict__ img, const int* __restrict__ flips,
                                                     const int* __restrict__ margin, float* __restrict__ up, int NC, int C,
                                                     int H, int W) {
  const int mx0 = margin[0], my0 = margin[1], mx1 = margin[2], my1 = margin[3];
  const int Wp = W + mx0 + mx1, Hp = H + my0 + my1, W2 = 2 * Wp, H2 = 2 * Hp;
  const long total = (long)NC * H2 * W2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xu = (int)(i % W2);
    const long r = i / W2;
    const int yu = (int)(r % H2), nc = (int)(r / H2);
    const int n = nc / C;
    const bool fx = flips[2 * n] != 0, fy = flips[2 * n + 1] != 0;
    const float* plane = img + (long)nc * H * W;
    float acc = 0.f;
    // correlation with the flipped filter, padding 6, over the zero-stuffed signal: only even positions carry samples
#pragma unroll
    for (int ky = 0; ky < 12; ++ky) {
      const int jy = yu + ky - 6;
      if (jy < 0 || jy > 2 * Hp - 2 || (jy & 1)) continue;
      int sy = reflect((jy >> 1) - my0, H);
      if (fy) sy = H - 1 - sy;
      float row = 0.f;
#pragma unroll
      for (int kx = 0; kx < 12; ++kx) {
        const int jx = xu + kx - 6;
        if (jx < 0 || jx > 2 * Wp - 2 || (jx & 1)) continue;
        int sx = reflect((jx >> 1) - mx0, W);
        if (fx) sx = W - 1 - sx;
        row += kSym6[11 - kx] * plane[sy * W + sx];
      }
      acc += kSym6[11 - ky] * row;
    }
    up[i] = acc;
  }
}

// g[nc][Hs][Ws] = bilinear sample of up[nc][2Hp][2Wp] at theta[n] * (x_o, y_o, 1), normalised coordinates
__global__ __launch_bounds__(256) void aug_resample_kernel(const float* __restrict__ up, const int* __restrict__ margin,
                                                           const float* __restrict__ theta, float* __restrict__ g, int NC,
                                                           int C, int H, int W) {
  const int Wi = 2 * (W + margin[0] + margin[2]), Hi = 2 * (H + margin[1] + margin[3]);
  const int Ws = 2 * (W + 6), Hs = 2 * (H + 6);
  const long total = (long)NC * Hs * Ws;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Ws);
    const long r = i / Ws;
    const int y = (int)(r % Hs), nc = (int)(r / Hs);
    const float* t = theta + (long)(nc / C) * 6;
    const float xo = (2.f * x + 1.f) / Ws - 1.f, yo = (2.f * y + 1.f) / Hs - 1.f;
    const float gx = t[0] * xo + t[1] * yo + t[2], gy = t[3] * xo + t[4] * yo + t[5];
    const float ix = ((gx + 1.f) * Wi - 1.f) * 0.5f, iy = ((gy + 1.f) * Hi - 1.f) * 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float ax = ix - fx0, ay = iy - fy0;
    const float* plane = up + (long)nc * Hi * Wi;
    auto at = [&](int yy, int xx) -> float {
      return ((unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi) ? plane[(long)yy * Wi + xx] : 0.f;
    };
    g[i] = at(y0, x0) * (1.f - ax) * (1.f - ay) + at(y0, x0 + 1) * ax * (1.f - ay) + at(y0 + 1, x0) * (1.f - ax) * ay +
           at(y0 + 1, x0 + 1) * ax * ay;
  }
}

// out[nc][H][W]: stride-2 correlation with sym6 (padding 5) in x then y, cropped by 3 on every side
__global__ __launch_bounds__(256) void aug_down_kernel(const float* __restrict__ g, float* __restrict__ out, int NC, int H,
                                                       int W) {
  const int Ws = 2 * (W + 6), Hs = 2 * (H + 6);
  const long total = (long)NC * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long r = i / W;
    const int y = (int)(r % H), nc = (int)(r / H);
    const float* plane = g + (long)nc * Hs * Ws;
    const int bx = 2 * (x + 3) - 5, by = 2 * (y + 3) - 5;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 12; ++ky) {
      const int yy = by + ky;
      if ((unsigned)yy >= (unsigned)Hs) continue;
      float row = 0.f;
#pragma unroll
      for (int kx = 0; kx < 12; ++kx) {
        const int xx = bx + kx;
        if ((unsigned)xx < (unsigned)Ws) row += kSym6[kx] * plane[(long)yy * Ws + xx];
      }
      acc += kSym6[ky] * row;
    }
    out[i] = acc;
  }
}

inline int aug_grid(long n) { return (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256); }

}  // namespace

extern "C" long adm_aug_workspace_floats(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H < 2 || W < 2) return -1;
  return (long)N * C * (2L * (3 * H - 2)) * (2L * (3 * W - 2)) + (long)N * C * (2L * (H + 6)) * (2L * (W + 6));
}

extern "C" int adm_augment_geometric(const float* images, const int* flips, const int* margin, const float* theta,
                                     float* ws, float* out, int N, int C, int H, int W, hipStream_t stream) {
  if (!images || !flips || !margin || !theta || !ws || !out || N <= 0 || C <= 0 || H < 2 || W < 2) return ADM_EINVAL;
  const int NC = N * C;
  float* up = ws;
  float* g = ws + (long)NC * (2L * (3 * H - 2)) * (2L * (3 * W - 2));
  hipLaunchKernelGGL(aug_up_kernel, dim3(aug_grid((long)NC * 4 * (3 * H - 2) * (3 * W - 2))), dim3(256), 0, stream, images,
                     flips, margin, up, NC, C, H, W);
  hipLaunchKernelGGL(aug_resample_kernel, dim3(aug_grid((long)NC * 4 * (H + 6) * (W + 6))), dim3(256), 0, stream, up, margin,
                     theta, g, NC, C, H, W);
  hipLaunchKernelGGL(aug_down_kernel, dim3(aug_grid((long)NC * H * W)), dim3(256), 0, stream, g, out, NC, H, W);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
