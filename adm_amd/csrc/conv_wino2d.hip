// 3x3 stride-1 convolution (forward and data gradient) with the 2-D Winograd transform F(2x2, 3x3) on the fp32 MFMA.
//
// conv_wino.hip applies F(2,3) along x only: 6 multiply-adds per (pixel, cin, cout) instead of 9.  Nesting the same
// transform along y gives F(2x2, 3x3): a 2x2 output tile from a 4x4 input patch with 16 multiplies instead of 36, i.e. 4 per
// (pixel, cin, cout) -- 1.5x less MFMA work again (2.25x less than the direct kernel):
//     V = B^T d B          (4x4 patch d; B^T rows: (1,0,-1,0) (0,1,1,0) (0,-1,1,0) (0,1,0,-1); only +-1)
//     U = G g G^T          (3x3 filter g; G rows: (1,0,0) (.5,.5,.5) (.5,-.5,.5) (0,0,1); done at weight-pack time)
//     M[ey][ex] = sum_c V[ey][ex] * U[ey][ex]                      (16 GEMMs over K = Cin)
//     Y = A^T M A          (A^T rows: (1,1,1,0) (0,1,-1,-1); 2x2 outputs)
// Constants are +-1 on the data side and {1, 1/2, 1/4} on the weights: fp32 rounding stays at the direct kernel's level.
//
// Sixteen accumulator tiles do not fit a wave (256 AGPRs at one wave per SIMD), so the y index `ey` is a LOOP IN TIME: the
// workgroup keeps the 1-D kernel's shape -- 64 x-pairs x 64 couts, four ex accumulator tiles per wave -- and makes four passes
// over K = Cin, one per ey, each on the y-combined input rows
//     ey = 0: r0 - r2      ey = 1: r1 + r2      ey = 2: r2 - r1      ey = 3: r1 - r3          (r_i = input row 2ty - 1 + i)
// with the matching weight plane U[ey][.].  After each pass the x output transform (A^T along x) is applied to the MFMA
// accumulators and the result is folded into the two output rows kept in ordinary registers (row 2ty: passes 0, 1, 2; row
// 2ty+1: passes 1, -2, -3).  Per row pair that is 4 Cin/16 stages instead of the 1-D kernel's 6 Cin/16, for 8 instead of 4
// pixel loads per stage.  The 64 extra registers cost the third resident workgroup (two per CU), so both operands are double
// buffered here (64 KB of LDS, one barrier per stage).
//   A: thread = (tile = row pair x x-pair, 16-byte channel quad): 2 rows x 4 pixels raw buffer loads (range check = zero
//      padding), 4 + 4 float4 add/sub (y combination, then B^T along x), four ds_write_b128 into the ex planes
//   B: transformed weights Wq[ey][ex][n][cin] straight to LDS by LDS-DMA (wave w streams plane ex = w of the current ey)
// Used for 3x3 stride-1 convs / data gradients with even H and W and M >= the Winograd threshold (ADM_WINOGRAD2D=0 keeps the
// 1-D kernel); the fused nearest-x2 layers stay on conv_wino.hip.
// Replaces F.conv2d of Conv2d.forward and its autograd data gradient (/root/reference/unet/uncond_unet.py:98-110).
#include "common.h"
#include "../../include/adm_hip.h"

#ifndef W2_ABL
#define W2_ABL 0      // diagnostic builds (tools/bench_wino2d.cpp): 1 no A global loads, 2 no A transform / LDS stores, 4 no B DMA, 8 no LDS fragment reads
#endif

namespace {

struct Wino2P {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int Mt, N, H, W, Hh, Wh, Cin, ldx, ldy, ldr, wrows, tilesN, xbytes, wbytes, plane;   // Mt = B * H/2 * W/2 tiles; plane = wrows * Cin
  int splitk, chunks_per_split; float* ws;     // split-K over the input channels (small maps): partial outputs to ws[z][M][N]
};

typedef __attribute__((address_space(3))) void wino2_lds_void;
int g_w2_ws = -1;                                 // -1: per shape (default), 1: wave-specialised kernel (igemm_wino2d_ws_kernel), 0: symmetric kernel
constexpr int W2P = 64, W2N = 64, W2K = 16;       // tiles x couts x K-step

__global__ __launch_bounds__(256, 2) void igemm_wino2d_kernel(Wino2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                               // [2][4][W2P][W2K]
  float* Bs = smem + 2 * 4 * W2P * W2K;           // [2][4][W2N][W2K]
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
  const int mt0 = tm * W2P, n0 = tn * W2N;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.wbytes, 0x00020000);
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
  const int a_slot = (aq ^ ((pl >> 2) & 3)) << 2;       // swizzled float offset inside the 16-float row
  unsigned a_voff[2][4];          // [row A / row B of the current pass][pixel j]
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) a_voff[r][j] = OOB;

  // ---- B loader (LDS-DMA): wave w streams plane ex = w of the current ey; instruction i covers rows i*16 + (lane >> 2)
  unsigned b_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + (lane >> 2);
    const int ls = (lane & 3) ^ ((row >> 2) & 3);
    const int n = n0 + row;
    b_voff[i] = (n < p.wrows) ? (unsigned)(wid * p.plane + n * p.Cin + ls * 4) * 4u : OOB;
  }

  // split-K (gridDim.y > 1): this workgroup reduces over the channel chunks [c_begin, c_begin + chunks) only and writes a raw
  // partial tile (no bias / residual) to its slice of the workspace; splitk_reduce (conv_igemm.hip) sums the slices in order
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
    const int soff = (c_begin + ld_cc) << 6;      // 16 floats = 64 bytes per chunk
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (W2_ABL & 1) { dA[j] = f32x4{1.f, 2.f, 3.f, (float)soff}; dB[j] = f32x4{0.5f, 0.25f, (float)j, 1.f}; continue; }
      dA[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[0][j], soff, 0));
      dB[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[1][j], soff, 0));
    }
    st_ey = ld_ey;
    const int kb = (ld_ey * 4 * p.plane) * 4 + soff;       // plane block of this ey; the wave's ex plane is in b_voff
    float* lb = Bs + (buf * 4 + wid) * W2N * W2K;
    if (!(W2_ABL & 4)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (wino2_lds_void*)(lb + i * 16 * W2K), 16, (int)b_voff[i], kb, 0, 0);
    }
    if (++ld_cc == chunks) { ld_cc = 0; ++ld_ey; }
  };
  auto store_stage = [&](int buf) {               // y combination, then B^T along x, into the four ex planes
    if (W2_ABL & 2) return;
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
    float* la = As + buf * 4 * W2P * W2K + pl * W2K + a_slot;
    *reinterpret_cast<f32x4*>(la + 0 * W2P * W2K) = e[0] - e[2];
    *reinterpret_cast<f32x4*>(la + 1 * W2P * W2K) = e[1] + e[2];
    *reinterpret_cast<f32x4*>(la + 2 * W2P * W2K) = e[2] - e[1];
    *reinterpret_cast<f32x4*>(la + 3 * W2P * W2K) = e[1] - e[3];
  };

  f32x16 acc[4];
  f32x16 Y[2][2];                                 // [output row][output column of the pair]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[a][b][r] = 0.f;

  int foff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) foff[g] = lr * W2K + (((2 * g + lh) ^ ((lr >> 2) & 3)) << 2);

  // Two workgroups share a CU (one wave of each per SIMD) and run the same program; the workgroups of the second resident slot
  // start half a stage later so that the pair does not reach its barriers and LDS bursts in lockstep (MI355X_MICROARCH.md,
  // 'Two waves per SIMD', item 9).  Measured: +1-2 % only -- see DESIGN.md for what does limit this kernel.
  if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(16);
  issue_stage(0);
  store_stage(0);
  __syncthreads();
  int cc = 0, ey = 0;                             // (chunk, pass) of the stage being COMPUTED
  for (int s = 0; s < KT; ++s) {
    const int buf = s & 1;
    if (s + 1 < KT) issue_stage(buf ^ 1);
    const float* Ab = As + buf * 4 * W2P * W2K + wm * 32 * W2K;
    const float* Bb = Bs + buf * 4 * W2N * W2K + wn * 32 * W2K;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        if (W2_ABL & 8) { a[xi] = f32x4{1.f, (float)s, (float)lane, 2.f}; b[xi] = f32x4{(float)xi, 1.f, 0.5f, (float)g}; continue; }
        a[xi] = *reinterpret_cast<const f32x4*>(Ab + xi * W2P * W2K + foff[g]);
        b[xi] = *reinterpret_cast<const f32x4*>(Bb + xi * W2N * W2K + foff[g]);
      }
      if (g == 0 && cc == 0) {                    // first products of a pass: C = 0 (inline constant), no accumulator clearing
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[xi][0], b[xi][0], zero, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 4; ++k)
#pragma unroll
          for (int xi = 0; xi < 4; ++xi)
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[xi][k], b[xi][k], acc[xi], 0, 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int xi = 0; xi < 4; ++xi)
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[xi][k], b[xi][k], acc[xi], 0, 0, 0);
      }
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

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant: the same algorithm, tile and LDS images with the work of a stage divided between two kinds of waves
// (768 threads, one workgroup per CU):
//   * waves 8-11 PRODUCE the A operand: the loads of a stage (2 rows x 4 pixels per thread) are issued FOUR stages ahead into one of
//     four register sets -- these waves hold no accumulators, so they have the registers the symmetric kernel lacks -- then
//     transformed and written to LDS one stage ahead.  In the symmetric kernel a stage is as long as a global load takes (issued at
//     its start, needed at its end, and __syncthreads drains vmcnt besides): removing the A loads there gains 18 %.
//   * waves 0-7 CONSUME: weight DMA two stages ahead (their only global accesses, so the explicit vmcnt wait before a barrier never
//     touches the producers' prefetch), 32 MFMAs per stage, output transform at the end of each pass.
//   * barriers are s_barrier with LDS-scoped fences / explicit counters: global loads stay in flight across them.
constexpr int WS_RA = 2, WS_RB = 3, WS_D = 4;     // A slots, B slots, producer prefetch depth in stages

__device__ __forceinline__ void ws_barrier_lds() {          // waits for this wave's LDS traffic only
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(768) void igemm_wino2d_ws_kernel(Wino2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                               // [WS_RA][4][W2P][W2K]
  float* Bs = smem + WS_RA * 4 * W2P * W2K;       // [WS_RB][4][W2N][W2K]
  const int tid = threadIdx.x, lane = tid & 63, hw_wid = tid >> 6;
  const bool producer = hw_wid >= 8;              // waves 0-7 consume (two per SIMD), waves 8-11 produce (one per SIMD)
  const int wid = producer ? hw_wid - 8 : hw_wid;
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap, m-fastest inside an n-tile (see conv_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int mt0 = tm * W2P, n0 = tn * W2N;
  constexpr unsigned OOB = 0x80000000u;
  const int c_begin = (p.splitk > 1) ? (int)blockIdx.y * p.chunks_per_split : 0;
  const int chunks = (p.splitk > 1) ? min(p.chunks_per_split, (p.Cin >> 4) - c_begin) : (p.Cin >> 4);     // 16-channel chunks
  const int S = 4 * chunks;                       // stages: ey outer, chunk inner (a multiple of WS_D)

  if (producer) {
    // ================================================================ A operand
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
    const int ptid = wid * 64 + lane;
    const int pl = ptid >> 2, aq = ptid & 3;      // tile, channel quad
    unsigned a_base = 0, colmask = 0, rowmask = 0;
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
    float* la = As + pl * W2K + ((aq ^ ((pl >> 2) & 3)) << 2);      // swizzled float offset inside the 16-float row
    unsigned a_voff[2][4];
    int ld_ey = 0, ld_cc = 0;
    auto set_rows = [&]() {     // pass ey combines input rows (iA, iB): 0: +r0 -r2   1: +r1 +r2   2: -r1 +r2   3: +r1 -r3; past the end: nothing
      const int iA = (ld_ey == 0) ? 0 : 1, iB = (ld_ey == 3) ? 3 : 2;
      const bool vA = ((rowmask >> iA) & 1u) && ld_ey < 4, vB = ((rowmask >> iB) & 1u) && ld_ey < 4;
      const int offA = (iA - 1) * p.W * p.ldx * 4, offB = (iB - 1) * p.W * p.ldx * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cv = (colmask >> j) & 1u;
        a_voff[0][j] = (vA && cv) ? a_base + (unsigned)(offA + (j - 1) * p.ldx * 4) : OOB;
        a_voff[1][j] = (vB && cv) ? a_base + (unsigned)(offB + (j - 1) * p.ldx * 4) : OOB;
      }
    };
    set_rows();
    f32x4 dA[WS_D][4], dB[WS_D][4];
    int set_ey[WS_D];
    auto issue = [&](int d) {                     // next stage in (ey, chunk) order -> register set d
      const int soff = (c_begin + ld_cc) << 6;    // 16 floats = 64 bytes per chunk
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (W2_ABL & 1) { dA[d][j] = f32x4{1.f, 2.f, 3.f, (float)soff}; dB[d][j] = f32x4{0.5f, 0.25f, (float)j, 1.f}; continue; }
        dA[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[0][j], soff, 0));
        dB[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[1][j], soff, 0));
      }
      set_ey[d] = ld_ey;
      if (++ld_cc == chunks) { ld_cc = 0; ++ld_ey; set_rows(); }
    };
    auto store = [&](int d, int slot) {           // y combination, then B^T along x, into the four ex planes of A slot `slot`
      if (W2_ABL & 2) return;
      f32x4 e[4];
      if (set_ey[d] == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = dA[d][j] + dB[d][j];
      } else if (set_ey[d] == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = dB[d][j] - dA[d][j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = dA[d][j] - dB[d][j];
      }
      float* l = la + slot * 4 * W2P * W2K;
      *reinterpret_cast<f32x4*>(l + 0 * W2P * W2K) = e[0] - e[2];
      *reinterpret_cast<f32x4*>(l + 1 * W2P * W2K) = e[1] + e[2];
      *reinterpret_cast<f32x4*>(l + 2 * W2P * W2K) = e[2] - e[1];
      *reinterpret_cast<f32x4*>(l + 3 * W2P * W2K) = e[1] - e[3];
    };
#pragma unroll
    for (int d = 0; d < WS_D; ++d) issue(d);
    __builtin_amdgcn_sched_barrier(0);
    // barrier t separates "A(t) written" from compute(t); A(t) lives in slot t & 1; set t % WS_D is refilled with stage t + WS_D
    for (int t = 0; t < S; t += WS_D) {
#pragma unroll
      for (int d = 0; d < WS_D; ++d) {
        store(d, d & 1);
        __builtin_amdgcn_sched_barrier(0);
        issue(d);                                 // stages past the end read nothing (all offsets out of range)
        __builtin_amdgcn_sched_barrier(0);
        ws_barrier_lds();
      }
    }
    return;
  }

  // ================================================================== consumer waves: weight DMA, MFMA, output transform
  // Eight waves, two per SIMD, each 32 tiles x 16 couts on v_mfma_f32_16x16x4_f32 (two 16-tile blocks x four ex planes = eight
  // 4-register accumulators + the folded output rows: 64 registers where a 32x32 tile needs 128).  A lone MFMA wave per SIMD cannot
  // cover its own LDS fragment reads and barrier waits (measured: 27 % of such a kernel's time); a pair does.
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.wbytes, 0x00020000);
  const int wm = wid >> 2, wn = wid & 3;          // 32-tile block, 16-cout block
  const int lr = lane & 15, lq = lane >> 4;       // operand row / k quad: lane holds k = 4 lq + i in element i of its 16-byte read
  if (p.splitk > 1) {
    p.y = p.ws + (long)blockIdx.y * ((long)p.Mt * 4) * p.N;
    p.ldy = p.N; p.bias = nullptr; p.res = nullptr;
  }
  // B loader (LDS-DMA): 16 instructions per stage = plane ex x 16-row group; wave w issues plane w >> 1, groups 2 (w & 1) + {0, 1}
  unsigned b_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = ((wid & 1) * 2 + i) * 16 + (lane >> 2);
    const int ls = (lane & 3) ^ ((row >> 2) & 3);
    const int n = n0 + row;
    b_voff[i] = (n < p.wrows) ? (unsigned)((wid >> 1) * p.plane + n * p.Cin + ls * 4) * 4u : OOB;
  }
  int ld_ey = 0, ld_cc = 0, ld_slot = 0;
  auto issue_b = [&]() {                          // weights of the next stage in (ey, chunk) order -> ring slot
    const int kb = (ld_ey * 4 * p.plane) * 4 + ((c_begin + ld_cc) << 6);
    float* lb = Bs + (ld_slot * 4 + (wid >> 1)) * W2N * W2K + (wid & 1) * 2 * 16 * W2K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (!(W2_ABL & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (wino2_lds_void*)(lb + i * 16 * W2K), 16, (int)b_voff[i], kb, 0, 0);
    if (++ld_cc == chunks) { ld_cc = 0; ++ld_ey; }
    if (++ld_slot == WS_RB) ld_slot = 0;
  };

  f32x4 acc[4][2];                                // [ex plane][16-tile block]
  f32x4 Y[2][2][2];                               // [output row][output column of the pair][16-tile block]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int k = 0; k < 2; ++k) Y[a][b][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment offsets (floats): row * 16 + swizzled quad (physical quad = k quad ^ ((row >> 2) & 3), rows here are multiples of 16 + lr)
  const int fq = (lq ^ ((lr >> 2) & 3)) << 2;
  const int a_foff = (wm * 32 + lr) * W2K + fq;   // block 1: + 16 rows
  const int b_foff = (wn * 16 + lr) * W2K + fq;

  // The fragment reads of stage t+1 are issued right after barrier t+1 and BEFORE the MFMAs of stage t (two register sets): the
  // barrier re-aligns all waves every stage, so reads issued after it and consumed at once are exposed in every wave at the same
  // time (measured: 24 % of the kernel with everything else switched off).
  f32x4 fa0[2][4], fa1[2][4], fb[1][4];            // (168 registers at three waves per SIMD: the B fragments have one set only)
  int slot_b = 0;
  auto sync_and_read = [&](int t, int set) -> const float* {      // barrier t (A(t), B(t) ready), weights of t+2 on their way, A fragments of t -> set
    if (t + 1 < S) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 2 < S) issue_b();
    const float* Ab = As + (t & 1) * 4 * W2P * W2K + a_foff;
    const float* Bb = Bs + slot_b * 4 * W2N * W2K + b_foff;
    if (++slot_b == WS_RB) slot_b = 0;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      if (W2_ABL & 8) { fa0[set][xi] = f32x4{1.f, (float)t, (float)lane, 2.f}; fa1[set][xi] = fa0[set][xi]; continue; }
      fa0[set][xi] = *reinterpret_cast<const f32x4*>(Ab + xi * W2P * W2K);
      fa1[set][xi] = *reinterpret_cast<const f32x4*>(Ab + xi * W2P * W2K + 16 * W2K);
    }
    return Bb;
  };
  auto read_b = [&](const float* Bb) {
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      if (W2_ABL & 8) { fb[0][xi] = f32x4{(float)xi, 1.f, 0.5f, 2.f}; continue; }
      fb[0][xi] = *reinterpret_cast<const f32x4*>(Bb + xi * W2N * W2K);
    }
  };
  int cc = 0, ey = 0;
  auto compute = [&](int set) {
    if (cc == 0) {                                // first products of a pass: C = 0 (inline constant), no accumulator clearing
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[set][xi][0], fb[0][xi][0], zero, 0, 0, 0);
        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[set][xi][0], fb[0][xi][0], zero, 0, 0, 0);
      }
#pragma unroll
      for (int k = 1; k < 4; ++k)
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
          acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[set][xi][k], fb[0][xi][k], acc[xi][0], 0, 0, 0);
          acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[set][xi][k], fb[0][xi][k], acc[xi][1], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
          acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[set][xi][k], fb[0][xi][k], acc[xi][0], 0, 0, 0);
          acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[set][xi][k], fb[0][xi][k], acc[xi][1], 0, 0, 0);
        }
    }
    if (++cc == chunks) {
      // end of pass ey: A^T along x, then fold into the output rows (A^T along y: row 0 = Z0 + Z1 + Z2, row 1 = Z1 - Z2 - Z3)
      cc = 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x4 m1 = acc[1][k], m2 = acc[2][k];
        const f32x4 z0 = acc[0][k] + m1 + m2, z1 = m1 - m2 - acc[3][k];
        if (ey <= 2) { Y[0][0][k] += z0; Y[0][1][k] += z1; }
        if (ey == 1) { Y[1][0][k] += z0; Y[1][1][k] += z1; }
        if (ey >= 2) { Y[1][0][k] -= z0; Y[1][1][k] -= z1; }
      }
      ++ey;
    }
  };
  issue_b();
  issue_b();
  const float* bcur = sync_and_read(0, 0);
  for (int t = 0; t < S; t += 2) {                // S is a multiple of 4
    read_b(bcur);
    const float* bnext = sync_and_read(t + 1, 1);  // B(t) stays valid: its slot is refilled with B(t+3) after barrier t+1 ... see below
    compute(0);
    read_b(bnext);
    if (t + 2 < S) bcur = sync_and_read(t + 2, 0);
    compute(1);
  }

  // ---- epilogue.  C/D layout of the 16x16 tile: col = lane & 15 (cout), row = 4 (lane >> 4) + r (tile)
  const int n = n0 + wn * 16 + lr;
  if (n >= p.N) return;
  const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = mt0 + wm * 32 + k * 16 + 4 * lq + r;
      if (t >= p.Mt) continue;
      const int xp = t % p.Wh;
      const int u = t / p.Wh;                        // = b * Hh + ty
      const long px0 = ((long)u * 2) * p.W + 2 * xp; // pixel (b, 2ty, 2xp) in units of pixels: (b*H + 2ty) * W + 2xp
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const long px = px0 + (long)a * p.W;
        float y0 = Y[a][0][k][r] + bv, y1 = Y[a][1][k][r] + bv;
        if (p.res) { y0 += p.res[px * p.ldr + n]; y1 += p.res[(px + 1) * p.ldr + n]; }
        p.y[px * p.ldy + n] = y0;
        p.y[(px + 1) * p.ldy + n] = y1;
      }
    }
}

// U = G g G^T for both operand layouts, from the reference's OIHW parameter:
//   wf[ey][ex][co][ci]            (forward B operand)
//   wb[ey][ex][ci][co]            (data-gradient B operand: taps flipped in y and x, channels transposed)
__device__ __forceinline__ void wino2_G(const float g[3], float u[4]) {
  u[0] = g[0]; u[1] = (g[0] + g[1] + g[2]) * 0.5f; u[2] = (g[0] - g[1] + g[2]) * 0.5f; u[3] = g[2];
}

__global__ void pack_wino2d_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int Co, int Ci,
                                   int Co_pad, int Ci_pad) {
  const long total = (long)Co_pad * Ci_pad;
  const long planef = (long)Co_pad * Ci_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci_pad), co = (int)(i / Ci_pad);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = (co < Co && ci < Ci) ? w[(((long)co * Ci + ci) * 3 + a) * 3 + b] : 0.f;
    float t[3][4];                 // rows transformed along x: t[ky][ex]
#pragma unroll
    for (int a = 0; a < 3; ++a) wino2_G(g[a], t[a]);
    float tf[3][4];                // flipped filter g'(ky', kx') = w(2-ky', 2-kx'), transformed along x
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float gr[3] = {g[2 - a][2], g[2 - a][1], g[2 - a][0]};
      wino2_G(gr, tf[a]);
    }
#pragma unroll
    for (int ex = 0; ex < 4; ++ex) {
      const float cf[3] = {t[0][ex], t[1][ex], t[2][ex]}, cb[3] = {tf[0][ex], tf[1][ex], tf[2][ex]};
      float uf[4], ub[4];
      wino2_G(cf, uf); wino2_G(cb, ub);
#pragma unroll
      for (int ey = 0; ey < 4; ++ey) {
        if (wf) wf[(long)(ey * 4 + ex) * planef + (long)co * Ci_pad + ci] = uf[ey];
        if (wb) wb[(long)(ey * 4 + ex) * planef + (long)ci * Co_pad + co] = ub[ey];
      }
    }
  }
}

}  // namespace

// wf[16][Co_pad][Ci_pad] and / or wb[16][Ci_pad][Co_pad] (plane index ey * 4 + ex) from an OIHW 3x3 weight
extern "C" int adm_pack_weight_wino2d(const float* w, float* wf, float* wb, int Co, int Ci, int Co_pad, int Ci_pad,
                                      hipStream_t stream) {
  if (!w || (!wf && !wb) || Co <= 0 || Ci <= 0 || Co_pad < Co || Ci_pad < Ci) return ADM_EINVAL;
  const long total = (long)Co_pad * Ci_pad;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_wino2d_kernel, dim3(grid), dim3(256), 0, stream, w, wf, wb, Co, Ci, Co_pad, Ci_pad);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// y[B][H][W][ldy] = conv3x3(x[B][H][W][ldx], pad 1) (+ bias) (+ res); wq = adm_pack_weight_wino2d operand with `wrows` rows per
// plane (>= N) and K = Cin columns.  H and W even; Cin % 16 == 0.  ws (may be NULL) = workspace of ws_floats >=
// adm_wino2d_splitk(...) * B*H*W*N floats: small launches then split K and reduce deterministically (fixed order).
int adm_splitk_reduce(const float* ws, const float* bias, const float* res, float* y, long M, int N, int ldy, int ldr, int splitk,
                      hipStream_t stream);       // conv_igemm.hip

// Split count over the input channels for small launches: the 64-tile x 64-cout workgroups of an 8x8 map at batch 128 number
// 192, for 512 resident slots.  Splits are chosen to fill the slots (whole rounds), keeping >= 4 chunks (64 channels) per split.
extern "C" int adm_wino2d_splitk(int B, int H, int W, int Cin, int N) {
  const long wgs = (long)adm_cdiv((long)B * (H / 2) * (W / 2), W2P) * adm_cdiv(N, W2N);
  const int chunks = Cin >> 4;
  if (wgs >= 384 || chunks < 8) return 1;
  int s = (int)(512 / wgs);
  if (s > chunks / 4) s = chunks / 4;
  if (s > 4) s = 4;
  return s < 2 ? 1 : s;
}

// -1 (default): chosen per launch; 0: always the symmetric kernel; 1: always the wave-specialised one.  Returns the old setting.
extern "C" int adm_wino2d_variant(int ws) {
  const int old = g_w2_ws;
  if (ws >= -1 && ws <= 1) g_w2_ws = ws;
  return old;
}

extern "C" int adm_conv_fwd_wino2d(const float* x, const float* wq, const float* bias, const float* res, float* y, float* ws,
                                   long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                                   hipStream_t stream) {
  if (!x || !wq || !y || B <= 0 || H < 2 || W < 2 || (W & 1) || (H & 1)) return ADM_EINVAL;
  if ((Cin & 15) || (ldx & 3) || N <= 0 || wrows < N) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)wq) & 15) return ADM_EINVAL;
  Wino2P p;
  p.x = x; p.w = wq; p.bias = bias; p.res = res; p.y = y;
  const long Mt = (long)B * (H / 2) * (W / 2);
  const long xb = (long)B * H * W * ldx * 4, wb = 16L * wrows * Cin * 4;
  if (Mt >= (1L << 30) || xb >= (1L << 31) || wb >= (1L << 31)) return ADM_EINVAL;
  p.Mt = (int)Mt; p.N = N; p.H = H; p.W = W; p.Hh = H / 2; p.Wh = W / 2; p.Cin = Cin; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr;
  p.wrows = wrows; p.xbytes = (int)xb; p.wbytes = (int)wb; p.plane = wrows * Cin;
  p.tilesN = adm_cdiv(N, W2N);
  p.splitk = 1; p.chunks_per_split = 0; p.ws = nullptr;
  const long wgs = (long)adm_cdiv(Mt, W2P) * p.tilesN;
  const int sk = (ws && !(N & 3) && !(ldy & 3) && (!res || !(ldr & 3))) ? adm_wino2d_splitk(B, H, W, Cin, N) : 1;
  if (sk > 1 && ws_floats >= (long)sk * Mt * 4 * N) {
    const int chunks = Cin >> 4;
    p.chunks_per_split = (chunks + sk - 1) / sk;
    p.splitk = (chunks + p.chunks_per_split - 1) / p.chunks_per_split;
    p.ws = ws;
  }
  (void)wgs;
  // Measured (tools/bench_wino2d.cpp, bs=128): the two kernels are within 3 % of each other everywhere; the wave-specialised one
  // wins on the 16x16 maps (768 workgroups: 194 vs 188, 210 vs 198 TFLOP/s algorithmic), the symmetric one on the 32x32 maps
  // (3072 workgroups: 220 vs 217), and they tie on split-K launches.
  const long wgs_all = (long)adm_cdiv(Mt, W2P) * p.tilesN * p.splitk;
  const bool use_ws = g_w2_ws < 0 ? (wgs_all > 512 && wgs_all <= 1024) : g_w2_ws == 1;
  if (use_ws) {
    constexpr int smem_ws = (WS_RA * 4 * W2P + WS_RB * 4 * W2N) * W2K * (int)sizeof(float);
    static bool ws_attr_set = false;
    if (!ws_attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wino2d_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              smem_ws) != hipSuccess)
        return ADM_ELAUNCH;
      ws_attr_set = true;
    }
    const long grid_ws = (long)adm_cdiv(Mt, W2P) * p.tilesN;
    hipLaunchKernelGGL(igemm_wino2d_ws_kernel, dim3((unsigned)grid_ws, p.splitk), dim3(768), smem_ws, stream, p);
    ADM_CHECK_LAUNCH();
    if (p.splitk > 1) return adm_splitk_reduce(ws, bias, res, y, Mt * 4, N, ldy, ldr, p.splitk, stream);
    return ADM_OK;
  }
  constexpr int smem = 2 * (4 * W2P + 4 * W2N) * W2K * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wino2d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
        hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  const long grid = (long)adm_cdiv(Mt, W2P) * p.tilesN;
  hipLaunchKernelGGL(igemm_wino2d_kernel, dim3((unsigned)grid, p.splitk), dim3(256), smem, stream, p);
  ADM_CHECK_LAUNCH();
  if (p.splitk > 1) return adm_splitk_reduce(ws, bias, res, y, Mt * 4, N, ldy, ldr, p.splitk, stream);
  return ADM_OK;
}
