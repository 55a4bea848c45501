// Weight / gradient layout transforms between the reference's parameter layout (OIHW,
// /root/reference/unet/uncond_unet.py:85) and the K-contiguous packed operands of the implicit-GEMM
// kernels.  Pure data movement over <= 11 MB per tensor; run once per optimiser step (or once per
// model for sampling), so simplicity beats tuning here.
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

// packed row (head, {q,k,v}, c) -> reference row (head, c, {q,k,v})   (uncond_unet.py:205)
__device__ __forceinline__ int qkv_to_ref(int n) {
  int h = n / 192, r = n - h * 192;
  int j = r >> 6, c = r & 63;
  return h * 192 + c * 3 + j;
}
__device__ __forceinline__ int qkv_to_packed(int n) {
  int h = n / 192, r = n - h * 192;
  int c = r / 3, j = r - c * 3;
  return h * 192 + j * 64 + c;
}

__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int Ci, int taps,
                                int Co_pad, int Ci_pad, int qkv) {
  long total = (long)Co_pad * taps * Ci_pad;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int ci = idx % Ci_pad;
    long t = idx / Ci_pad;
    int tap = t % taps;
    int cop = t / taps;
    float v = 0.f;
    if (cop < Co && ci < Ci) {
      int co = qkv ? qkv_to_ref(cop) : cop;
      v = w[((long)co * Ci + ci) * taps + tap];
    }
    out[idx] = v;
  }
}

// out[ci][taps-1-tap][co_packed] = w[co][ci][tap]
__global__ void pack_bwd_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int Ci, int taps,
                                int Co_pad, int Ci_pad, int qkv) {
  long total = (long)Ci_pad * taps * Co_pad;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int cop = idx % Co_pad;
    long t = idx / Co_pad;
    int tapf = t % taps;
    int ci = t / taps;
    float v = 0.f;
    if (cop < Co && ci < Ci) {
      int co = qkv ? qkv_to_ref(cop) : cop;
      v = w[((long)co * Ci + ci) * taps + (taps - 1 - tapf)];
    }
    out[idx] = v;
  }
}

__global__ void unpack_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int taps,
                              int Ci_pad, int qkv, int accumulate, int splits, long split_stride) {
  long total = (long)Co * Ci * taps;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int tap = idx % taps;
    long t = idx / taps;
    int ci = t % Ci;
    int co = t / Ci;
    int cop = qkv ? qkv_to_packed(co) : co;
    const long src = ((long)cop * taps + tap) * Ci_pad + ci;
    float v = dwp[src];
    for (int z = 1; z < splits; ++z) v += dwp[z * split_stride + src];      // fixed order: deterministic
    dw[idx] = accumulate ? dw[idx] + v : v;
  }
}

__global__ void permute_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int n_pad, int qkv,
                                   int inverse) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  float v = 0.f;
  if (i < n) {
    int src = i;
    if (qkv) src = inverse ? qkv_to_packed(i) : qkv_to_ref(i);
    v = in[src];
  }
  out[i] = v;
}

// out[n] = sum_m a[m][n].  One block per 64-column strip x row-chunk; fp32 partials, atomics across chunks.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, float* __restrict__ out, int M,
                                                     int N, int ld, int rows_per_block) {
  __shared__ float part[4][64];
  int col = blockIdx.x * 64 + (threadIdx.x & 63);
  int ry = threadIdx.x >> 6;
  int m0 = blockIdx.y * rows_per_block;
  int m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (col < N)
    for (int m = m0 + ry; m < m1; m += 4) s += a[(long)m * ld + col];
  part[ry][threadIdx.x & 63] = s;
  __syncthreads();
  if (ry == 0 && col < N) {
    s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    atomicAdd(&out[col], s);
  }
}

// One launch for ALL layers of a model.  table[e] = {src, dst_fwd, dst_bwd, Co, Ci, taps, Co_pad, Ci_pad, qkv, tile_begin,
// dst_wino_fwd, dst_wino_bwd, dst_wino2d_fwd, dst_wino2d_bwd, dst_x6_fwd, dst_x6_bwd, dst_gemm_x6_fwd, dst_gemm_x6_bwd} (int64 each; tile_begin = exclusive prefix sum of (Co_pad/32)*(Ci_pad/32) over the rows, in
// row order; the two Winograd destinations are 0 for layers that do not use conv_wino.hip).  One workgroup per 32
// (out-channel) x 32 (in-channel) tile of one layer: the tile's taps-interleaved source runs (32*taps contiguous floats per
// out-channel) go through LDS once and leave as 128-byte row segments of every operand layout, so reads and writes are
// coalesced (the previous element-per-thread gather ran at 0.7 TB/s and cost 3.8 ms per optimiser step).
constexpr int PT_COLS = 24;      // 18, 19: fp16-format images of the 2-D Winograd operands; 20: their scale (float bits); 21: overflow flag (int*); 22, 23: fp16-format images of the 1x1 operands (same scale and flag)
__global__ __launch_bounds__(256) void pack_table_kernel(const long* __restrict__ table, int n_entries) {
  __shared__ float tile[32][32 * 9 + 1];
  // layer of this tile: last row whose tile_begin <= blockIdx.x
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * PT_COLS + 9] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long* t = table + (long)lo * PT_COLS;
  const float* __restrict__ w = reinterpret_cast<const float*>(t[0]);
  float* __restrict__ fwd = reinterpret_cast<float*>(t[1]);
  float* __restrict__ bwd = reinterpret_cast<float*>(t[2]);
  const int Co = (int)t[3], Ci = (int)t[4], taps = (int)t[5], Co_pad = (int)t[6], Ci_pad = (int)t[7], qkv = (int)t[8];
  float* __restrict__ wf = reinterpret_cast<float*>(t[10]);
  float* __restrict__ wb = reinterpret_cast<float*>(t[11]);
  const int local = (int)((long)blockIdx.x - t[9]);
  const int tiles_ci = Ci_pad >> 5;
  const int co0 = (local / tiles_ci) << 5, ci0 = (local % tiles_ci) << 5;
  const int run = 32 * taps;                       // floats per out-channel row of the tile
  const int n = 32 * run;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = e / run, c = e - r * run;        // c = ci_l * taps + tap
    const int cop = co0 + r, ci = ci0 + c / taps;
    float v = 0.f;
    if (cop < Co && ci < Ci) v = w[((long)(qkv ? qkv_to_ref(cop) : cop) * Ci + ci0) * taps + c];
    tile[r][c] = v;
  }
  __syncthreads();
  // 1x1 layers on conv_gemm_x6.hip: the exact three-term bf16 split of both operands, [k/16][term][rows][16] (columns 16, 17)
  unsigned short* __restrict__ g6f = reinterpret_cast<unsigned short*>(t[16]);
  unsigned short* __restrict__ g6b = reinterpret_cast<unsigned short*>(t[17]);
  // ... and their two-term fp16 images [k/16][term(2)][rows][16] of scale * w (conv_gemm_x6.hip FMT 1, adm_split2_rows_f16; columns 22, 23)
  unsigned short* __restrict__ g6fh = reinterpret_cast<unsigned short*>(t[22]);
  unsigned short* __restrict__ g6bh = reinterpret_cast<unsigned short*>(t[23]);
  const float gscale = __uint_as_float((unsigned)t[20]);
  bool gbad = false;
  auto split2_store = [&](unsigned short* dh, long termh, float v) {
    const float a = v * gscale;
    gbad |= !(fabsf(a) < 65000.f);
    const _Float16 h0 = (_Float16)a, h1 = (_Float16)(a - (float)h0);
    dh[0] = __builtin_bit_cast(unsigned short, h0);
    dh[termh] = __builtin_bit_cast(unsigned short, h1);
  };
  auto split_store = [](unsigned short* d6, long term6, float v) {
    const unsigned b0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(b0 & 0xFFFF0000u);
    const unsigned b1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
    d6[0] = (unsigned short)(b0 >> 16);
    d6[term6] = (unsigned short)(b1 >> 16);
    d6[2 * term6] = (unsigned short)(__float_as_uint(r2) >> 16);
  };
  for (int e = threadIdx.x; e < n; e += 256) {     // forward operand [Co_pad][taps][Ci_pad]: ci fastest
    const int ci_l = e & 31, rest = e >> 5;
    const int tap = rest % taps, co_l = rest / taps;
    const float v = tile[co_l][ci_l * taps + tap];
    fwd[((long)(co0 + co_l) * taps + tap) * Ci_pad + ci0 + ci_l] = v;
    if (g6f && taps == 1) {
      const int k = ci0 + ci_l;
      split_store(g6f + ((((long)(k >> 4) * 3) * Co_pad + co0 + co_l) << 4) + (k & 15), (long)Co_pad << 4, v);
    }
    if (g6fh && taps == 1) {
      const int k = ci0 + ci_l;
      split2_store(g6fh + ((((long)(k >> 4) * 2) * Co_pad + co0 + co_l) << 4) + (k & 15), (long)Co_pad << 4, v);
    }
  }
  for (int e = threadIdx.x; e < n; e += 256) {     // data-gradient operand [Ci_pad][taps flipped][Co_pad]: co fastest
    const int co_l = e & 31, rest = e >> 5;
    const int tapf = rest % taps, ci_l = rest / taps;
    const float v = tile[co_l][ci_l * taps + (taps - 1 - tapf)];
    bwd[((long)(ci0 + ci_l) * taps + tapf) * Co_pad + co0 + co_l] = v;
    if (g6b && taps == 1) {
      const int k = co0 + co_l;
      split_store(g6b + ((((long)(k >> 4) * 3) * Ci_pad + ci0 + ci_l) << 4) + (k & 15), (long)Ci_pad << 4, v);
    }
    if (g6bh && taps == 1) {
      const int k = co0 + co_l;
      split2_store(g6bh + ((((long)(k >> 4) * 2) * Ci_pad + ci0 + ci_l) << 4) + (k & 15), (long)Ci_pad << 4, v);
    }
  }
  if (gbad && t[21]) *reinterpret_cast<int*>(t[21]) = 1;      // a scaled 1x1 weight left the fp16 range: the host falls back to the bf16 format
  if (taps != 9) return;
  // Winograd F(2,3) operands (conv_wino.hip): G g per filter row, u = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
  if (wf) {                                        // wf[xi][co][ky][ci]: ci fastest
    const long plane = (long)Co_pad * 3 * Ci_pad;
    for (int e = threadIdx.x; e < 32 * 3 * 32; e += 256) {
      const int ci_l = e & 31, rest = e >> 5;
      const int ky = rest % 3, co_l = rest / 3;
      const float* g = &tile[co_l][ci_l * 9 + ky * 3];
      const float g0 = g[0], g1 = g[1], g2 = g[2];
      const long o = ((long)(co0 + co_l) * 3 + ky) * Ci_pad + ci0 + ci_l;
      wf[o] = g0;
      wf[o + plane] = (g0 + g1 + g2) * 0.5f;
      wf[o + 2 * plane] = (g0 - g1 + g2) * 0.5f;
      wf[o + 3 * plane] = g2;
    }
  }
  // 2-D Winograd F(2x2, 3x3) operands (conv_wino2d.hip): U = G g G^T, plane ey * 4 + ex
  // ... and their exact three-term bf16 splits in adm_split3_bf16's K-chunk-tiled layout for conv_wino2d_x6.hip (columns 14, 15; 0 when unused)
  float* __restrict__ wf2 = reinterpret_cast<float*>(t[12]);
  float* __restrict__ wb2 = reinterpret_cast<float*>(t[13]);
  unsigned short* __restrict__ wf6 = reinterpret_cast<unsigned short*>(t[14]);
  unsigned short* __restrict__ wb6 = reinterpret_cast<unsigned short*>(t[15]);
  // ... and their two-term fp16 images [ey][cols/16][ex][term(2)][rows][16] of scale * U (conv_wino2d_x6.hip, X6Fmt<1>)
  unsigned short* __restrict__ wfh = reinterpret_cast<unsigned short*>(t[18]);
  unsigned short* __restrict__ wbh = reinterpret_cast<unsigned short*>(t[19]);
  const float hscale = __uint_as_float((unsigned)t[20]);
  int* __restrict__ hflag = reinterpret_cast<int*>(t[21]);
  bool hbad = false;
  if (wf2 || wb2 || wf6 || wb6 || wfh || wbh) {
    const long plane2 = (long)Co_pad * Ci_pad;
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
      // forward operand: ci fastest; data-gradient operand: co fastest -> two index maps over the same 32 x 32 tile
      for (int which = 0; which < 2; ++which) {
        float* __restrict__ dst = which ? wb2 : wf2;
        unsigned short* __restrict__ dst6 = which ? wb6 : wf6;
        unsigned short* __restrict__ dsth = which ? wbh : wfh;
        if (!dst && !dst6 && !dsth) continue;
        const int fast = e & 31, slow = e >> 5;
        const int co_l = which ? fast : slow, ci_l = which ? slow : fast;
        const float* g = &tile[co_l][ci_l * 9];
        float tr[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int ra = which ? 2 - a : a;            // flipped filter for the data gradient
          const float g0 = which ? g[ra * 3 + 2] : g[ra * 3], g1 = g[ra * 3 + 1], g2 = which ? g[ra * 3] : g[ra * 3 + 2];
          tr[a][0] = g0; tr[a][1] = (g0 + g1 + g2) * 0.5f; tr[a][2] = (g0 - g1 + g2) * 0.5f; tr[a][3] = g2;
        }
        const long o = which ? (long)(ci0 + ci_l) * Co_pad + co0 + co_l : (long)(co0 + co_l) * Ci_pad + ci0 + ci_l;
#pragma unroll
        for (int ex = 0; ex < 4; ++ex) {
          const float c0 = tr[0][ex], c1 = tr[1][ex], c2 = tr[2][ex];
          const float u[4] = {c0, (c0 + c1 + c2) * 0.5f, (c0 - c1 + c2) * 0.5f, c2};
#pragma unroll
          for (int ey = 0; ey < 4; ++ey) {
            if (dst) dst[(long)(ey * 4 + ex) * plane2 + o] = u[ey];
            if (dst6) {                            // a = a0 + a1 + a2 exactly (truncation splits, as adm_split3_bf16)
              const unsigned b0 = __float_as_uint(u[ey]);
              const float r1 = u[ey] - __uint_as_float(b0 & 0xFFFF0000u);
              const unsigned b1 = __float_as_uint(r1);
              const float r2 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
              // [ey][cols/16][ex][term][rows][16] with (rows, cols) = (Co_pad, Ci_pad) forward, (Ci_pad, Co_pad) data gradient
              const int rows6 = which ? Ci_pad : Co_pad, cols6 = which ? Co_pad : Ci_pad;
              const int n6 = which ? ci0 + ci_l : co0 + co_l, c6 = which ? co0 + co_l : ci0 + ci_l;
              unsigned short* d6 = dst6 + ((((long)(ey * (cols6 >> 4) + (c6 >> 4)) * 12 + ex * 3) * rows6 + n6) << 4) + (c6 & 15);
              const long term6 = (long)rows6 << 4;
              d6[0] = (unsigned short)(b0 >> 16);
              d6[term6] = (unsigned short)(b1 >> 16);
              d6[2 * term6] = (unsigned short)(__float_as_uint(r2) >> 16);
            }
            if (dsth) {                            // scale * u = h0 + h1, round to nearest (as adm_split2_f16)
              const float a = u[ey] * hscale;
              hbad |= !(fabsf(a) < 65000.f);
              const _Float16 h0 = (_Float16)a, h1 = (_Float16)(a - (float)h0);
              const int rowsh = which ? Ci_pad : Co_pad, colsh = which ? Co_pad : Ci_pad;
              const int nh = which ? ci0 + ci_l : co0 + co_l, ch = which ? co0 + co_l : ci0 + ci_l;
              unsigned short* dh = dsth + ((((long)(ey * (colsh >> 4) + (ch >> 4)) * 8 + ex * 2) * rowsh + nh) << 4) + (ch & 15);
              dh[0] = __builtin_bit_cast(unsigned short, h0);
              dh[(long)rowsh << 4] = __builtin_bit_cast(unsigned short, h1);
            }
          }
        }
      }
    }
    if (hbad && hflag) *hflag = 1;                 // a scaled weight left the fp16 range: the host falls back to the bf16 format
  }
  if (wb) {                                        // wb[xi][ci][ky'][co]: co fastest; g'(ky', kx') = w(2-ky', 2-kx')
    const long plane = (long)Ci_pad * 3 * Co_pad;
    for (int e = threadIdx.x; e < 32 * 3 * 32; e += 256) {
      const int co_l = e & 31, rest = e >> 5;
      const int kyf = rest % 3, ci_l = rest / 3;
      const float* g = &tile[co_l][ci_l * 9 + (2 - kyf) * 3];
      const float g0 = g[2], g1 = g[1], g2 = g[0];
      const long o = ((long)(ci0 + ci_l) * 3 + kyf) * Co_pad + co0 + co_l;
      wb[o] = g0;
      wb[o + plane] = (g0 + g1 + g2) * 0.5f;
      wb[o + 2 * plane] = (g0 - g1 + g2) * 0.5f;
      wb[o + 3 * plane] = g2;
    }
  }
}

}  // namespace

extern "C" int adm_pack_weight_table(const long* table, int n_entries, long total_tiles, hipStream_t stream) {
  if (!table || n_entries <= 0 || total_tiles <= 0 || total_tiles >= (1L << 31)) return ADM_EINVAL;
  hipLaunchKernelGGL(pack_table_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, table, n_entries);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_pack_weight(const float* w, float* wp_fwd, float* wp_bwd, int Co, int Ci, int ks, int Co_pad,
                               int Ci_pad, int qkv, hipStream_t stream) {
  if (!w || Co <= 0 || Ci <= 0 || ks < 1 || ks > 7 || Co_pad < Co || Ci_pad < Ci) return ADM_EINVAL;
  if (qkv && (Co % 192 != 0 || Co_pad != Co)) return ADM_EINVAL;
  int taps = ks * ks;
  long total = (long)Co_pad * taps * Ci_pad;
  int grid = (int)min((long)4096, (total + 255) / 256);
  if (wp_fwd) hipLaunchKernelGGL(pack_fwd_kernel, dim3(grid), dim3(256), 0, stream, w, wp_fwd, Co, Ci, taps, Co_pad, Ci_pad, qkv);
  if (wp_bwd) hipLaunchKernelGGL(pack_bwd_kernel, dim3(grid), dim3(256), 0, stream, w, wp_bwd, Co, Ci, taps, Co_pad, Ci_pad, qkv);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_unpack_wgrad(const float* dwp, float* dw, int Co, int Ci, int ks, int Co_pad, int Ci_pad, int qkv,
                                int accumulate, hipStream_t stream) {
  if (!dwp || !dw || Co <= 0 || Ci <= 0 || ks < 1 || ks > 7 || Co_pad < Co || Ci_pad < Ci) return ADM_EINVAL;
  if (qkv && (Co % 192 != 0)) return ADM_EINVAL;
  int taps = ks * ks;
  long total = (long)Co * Ci * taps;
  int grid = (int)min((long)4096, (total + 255) / 256);
  hipLaunchKernelGGL(unpack_kernel, dim3(grid), dim3(256), 0, stream, dwp, dw, Co, Ci, taps, Ci_pad, qkv, accumulate, 1, 0L);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

namespace {
__global__ void bias_splits_kernel(const float* __restrict__ bws, float* __restrict__ dbias, int n, int splits, long stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = bws[i];
  for (int z = 1; z < splits; ++z) v += bws[z * stride + i];
  dbias[i] += v;
}
}  // namespace

// Deterministic counterpart of the atomic split reduction: ws[splits][Co_pad][taps][Ci_pad] partial weight gradients (from
// adm_conv_wgrad_ws) are summed in split order and scattered to OIHW; bws[splits][Co_pad] bias partials are summed into
// dbias[0..Co_pad) (+=, the packed channel order the atomic path accumulates in).  bws / dbias may be NULL.
extern "C" int adm_unpack_wgrad_splits(const float* ws, int splits, float* dw, int Co, int Ci, int ks, int Co_pad, int Ci_pad,
                                       int qkv, int accumulate, const float* bws, float* dbias, hipStream_t stream) {
  if (!ws || !dw || splits < 1 || Co <= 0 || Ci <= 0 || ks < 1 || ks > 7 || Co_pad < Co || Ci_pad < Ci) return ADM_EINVAL;
  if (qkv && (Co % 192 != 0)) return ADM_EINVAL;
  if ((bws == nullptr) != (dbias == nullptr)) return ADM_EINVAL;
  int taps = ks * ks;
  long total = (long)Co * Ci * taps;
  int grid = (int)min((long)4096, (total + 255) / 256);
  hipLaunchKernelGGL(unpack_kernel, dim3(grid), dim3(256), 0, stream, ws, dw, Co, Ci, taps, Ci_pad, qkv, accumulate, splits,
                     (long)Co_pad * taps * Ci_pad);
  if (bws)
    hipLaunchKernelGGL(bias_splits_kernel, dim3((Co_pad + 255) / 256), dim3(256), 0, stream, bws, dbias, Co_pad, splits,
                       (long)Co_pad);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_permute_vec(const float* in, float* out, int n, int n_pad, int qkv, int inverse,
                               hipStream_t stream) {
  if (!in || !out || n <= 0 || n_pad < n) return ADM_EINVAL;
  if (qkv && n % 192 != 0) return ADM_EINVAL;
  hipLaunchKernelGGL(permute_vec_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, stream, in, out, n, n_pad, qkv, inverse);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

extern "C" int adm_colsum(const float* a, float* out, int M, int N, int ld, int accumulate, hipStream_t stream) {
  if (!a || !out || M <= 0 || N <= 0 || ld < N) return ADM_EINVAL;
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, sizeof(float) * N, stream) != hipSuccess) return ADM_ELAUNCH;
  }
  int strips = (N + 63) / 64;
  int chunks = max(1, min(adm_cdiv(M, 64), 2048 / strips));
  int rows = adm_cdiv(M, chunks);
  chunks = adm_cdiv(M, rows);
  hipLaunchKernelGGL(colsum_kernel, dim3(strips, chunks), dim3(256), 0, stream, a, out, M, N, ld, rows);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

namespace {
// out[(tap * Ci_pad + ci)][co] = w[co][ci][tap]: the B operand (rows = (tap, ci), K = co contiguous) of the GEMM form of a
// transposed convolution, col[m][(tap, ci)] = sum_co dy[m][co] w[co][ci][tap]
__global__ void pack_tconv_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int Ci, int taps, int Co_pad,
                                  int Ci_pad) {
  const long total = (long)taps * Ci_pad * Co_pad;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int co = idx % Co_pad;
    const long t = idx / Co_pad;
    const int ci = t % Ci_pad, tap = t / Ci_pad;
    out[idx] = (co < Co && ci < Ci) ? w[((long)co * Ci + ci) * taps + tap] : 0.f;
  }
}
}  // namespace

extern "C" int adm_pack_weight_tconv(const float* w, float* out, int Co, int Ci, int ks, int Co_pad, int Ci_pad,
                                     hipStream_t stream) {
  if (!w || !out || Co <= 0 || Ci <= 0 || ks < 1 || ks > 7 || Co_pad < Co || Ci_pad < Ci) return ADM_EINVAL;
  const long total = (long)ks * ks * Ci_pad * Co_pad;
  hipLaunchKernelGGL(pack_tconv_kernel, dim3((unsigned)min((long)4096, (total + 255) / 256)), dim3(256), 0, stream, w, out, Co, Ci,
                     ks * ks, Co_pad, Ci_pad);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

namespace {
// dw[co][ci][ky][kx] (+)= sum_z sum_ey Gt[ky][ey] * wx[z][co][ey][kx][ci],  Gt = G^T rows (1, 1/2, 1/2, 0) (0, 1/2, -1/2, 0) (0, 1/2, 1/2, 1)
__global__ void unpack_wino2d_kernel(const float* __restrict__ wx, float* __restrict__ dw, int Co, int Ci, int Ci_pad, int accumulate,
                                     int splits, long split_stride) {
  const long total = (long)Co * Ci * 3;      // one thread per (co, ci, kx): its three ky taps
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int kx = idx % 3;
    const long t = idx / 3;
    const int ci = t % Ci, co = t / Ci;
    float m[4];
#pragma unroll
    for (int ey = 0; ey < 4; ++ey) {
      const long src = (((long)co * 4 + ey) * 3 + kx) * Ci_pad + ci;
      float v = wx[src];
      for (int z = 1; z < splits; ++z) v += wx[z * split_stride + src];
      m[ey] = v;
    }
    const float h = 0.5f * (m[1] + m[2]);
    const float w[3] = {m[0] + h, 0.5f * (m[1] - m[2]), h + m[3]};
    float* o = dw + ((long)co * Ci + ci) * 9 + kx;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) o[ky * 3] = accumulate ? o[ky * 3] + w[ky] : w[ky];
  }
}
}  // namespace

// adm_unpack_wgrad (+ _splits) for the output of adm_conv_wgrad_wino2d / adm_conv_wgrad_ws(wino = 2): wx[splits][Co_pad][4][3][Ci_pad]
extern "C" int adm_unpack_wgrad_wino2d(const float* wx, int splits, float* dw, int Co, int Ci, int Co_pad, int Ci_pad, int accumulate,
                                       const float* bws, float* dbias, hipStream_t stream) {
  if (!wx || !dw || splits < 1 || Co <= 0 || Ci <= 0 || Co_pad < Co || Ci_pad < Ci) return ADM_EINVAL;
  if ((bws == nullptr) != (dbias == nullptr)) return ADM_EINVAL;
  const long total = (long)Co * Ci * 3;
  hipLaunchKernelGGL(unpack_wino2d_kernel, dim3((unsigned)min((long)4096, (total + 255) / 256)), dim3(256), 0, stream, wx, dw, Co, Ci,
                     Ci_pad, accumulate, splits, (long)Co_pad * 12 * Ci_pad);
  if (bws)
    hipLaunchKernelGGL(bias_splits_kernel, dim3((Co_pad + 255) / 256), dim3(256), 0, stream, bws, dbias, Co_pad, splits, (long)Co_pad);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

namespace {
// One launch for the weight gradients of ALL layers.  Row = 12 int64: {src, dst, Co, Ci, taps (0: 2-D Winograd planes), Ci_pad, qkv,
// accumulate, zero_src, block_begin, 0, 0}; block_begin = exclusive prefix sum of the rows' block counts (UT_ITEMS items per block;
// items = Co*Ci*taps, or Co*Ci*3 for the Winograd planes, as in unpack_kernel / unpack_wino2d_kernel).
constexpr int UT_ITEMS = 2048, UT_COLS = 12;
__global__ __launch_bounds__(256) void unpack_table_kernel(const long* __restrict__ table, int rows) {
  int lo = 0, hi = rows - 1;                        // last row whose block_begin <= blockIdx.x (uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * UT_COLS + 9] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long* r = table + (long)lo * UT_COLS;
  float* src = reinterpret_cast<float*>(r[0]);
  float* dw = reinterpret_cast<float*>(r[1]);
  const int Co = (int)r[2], Ci = (int)r[3], taps = (int)r[4], Ci_pad = (int)r[5], qkv = (int)r[6];
  const bool accumulate = r[7] != 0, zero = r[8] != 0;
  const long b0 = ((long)blockIdx.x - r[9]) * UT_ITEMS;
  if (taps > 0) {
    const long total = (long)Co * Ci * taps;
    for (long idx = b0 + threadIdx.x; idx < min(total, b0 + UT_ITEMS); idx += 256) {
      const int tap = idx % taps;
      const long t = idx / taps;
      const int ci = t % Ci, co = t / Ci;
      const int cop = qkv ? qkv_to_packed(co) : co;
      float* sp = src + ((long)cop * taps + tap) * Ci_pad + ci;
      const float v = *sp;
      if (zero) *sp = 0.f;
      dw[idx] = accumulate ? dw[idx] + v : v;
    }
  } else {
    const long total = (long)Co * Ci * 3;
    for (long idx = b0 + threadIdx.x; idx < min(total, b0 + UT_ITEMS); idx += 256) {
      const int kx = idx % 3;
      const long t = idx / 3;
      const int ci = t % Ci, co = t / Ci;
      float m[4];
#pragma unroll
      for (int ey = 0; ey < 4; ++ey) {
        float* sp = src + (((long)co * 4 + ey) * 3 + kx) * Ci_pad + ci;
        m[ey] = *sp;
        if (zero) *sp = 0.f;
      }
      const float h = 0.5f * (m[1] + m[2]);
      const float w[3] = {m[0] + h, 0.5f * (m[1] - m[2]), h + m[3]};
      float* o = dw + ((long)co * Ci + ci) * 9 + kx;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) o[ky * 3] = accumulate ? o[ky * 3] + w[ky] : w[ky];
    }
  }
}
}  // namespace

extern "C" int adm_unpack_wgrad_table(const long* table, int rows, long total_blocks, hipStream_t stream) {
  if (!table || rows <= 0 || total_blocks <= 0 || total_blocks >= (1L << 31)) return ADM_EINVAL;
  hipLaunchKernelGGL(unpack_table_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, table, rows);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
