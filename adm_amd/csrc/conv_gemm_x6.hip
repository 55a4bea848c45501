// 1x1 convolution / pixel-wise linear map (forward and data gradient) with the f32 products carried on the bf16 MFMA through the exact
// three-term bf16 split of both operands (the arithmetic of conv_wino2d_x6.hip: a = a0 + a1 + a2 exactly, six bf16 products per f32
// product, f32 accumulation, error at the f32 MFMA's level):
//     y[m][n] = sum_k x[m][k] w[n][k] (+ bias[n]) (+ res[m][n])          m = pixel (NHWC row), k = input channel
// Structure = conv_wino2d_x6.hip's without the Winograd transforms (512 threads, one workgroup per CU, wave-specialised):
//   * workgroup tile 128 pixels x 128 couts, K step 32 channels (two 16-channel chunks) per stage and barrier;
//   * waves 4-7 PRODUCE the A operand: a wave instruction loads 8 pixel rows x 128 contiguous bytes; loads are issued four stages
//     ahead into one of four register sets; three-term split, twelve ds_write_b64 per stage;
//   * waves 0-3 CONSUME: 64 pixels x 64 couts each (four 32x32 accumulator tiles): 24 ds_read_b128 for 48 MFMAs per stage -- half the
//     fragment traffic per MFMA of the Winograd kernel, whose four ex planes leave room for one tile only.  The MFMA accumulators run
//     over two stages (24 matrix adds from C = 0) and are then added to the running totals with f32 adds;
//   * weights: split once per optimiser step into Wg6[chunk][term][n][16] (adm_split3_rows), one contiguous KB per LDS-DMA
//     instruction, three stages ahead; LDS: A 2 x 24 KB + B 4 x 24 KB.
// Used for 1x1 convs with M >= 8192, N >= 128 (a ragged last 128-cout tile reads zero rows), K % 32 == 0 (ADM_BF16X6=0 keeps them on conv_igemm.hip).
// Replaces F.conv2d (1x1) of Conv2d.forward and its data gradient (/root/reference/unet/uncond_unet.py:98-110).
#include "common.h"
#include "../../include/adm_hip.h"

namespace {

struct G6P {
  const float* x; const unsigned short* w; const float* bias; const float* res; float* y;
  int M, N, K, ldx, ldy, ldr, wrows, tilesN, xbytes, wbytes, ybytes, rbytes;
  const float* amax_x; float wscale;   // fp16 format only: bound vector (adm_hip.h) of |x|; the weights' (power-of-two) scale
  float* amax_y;                       // (may be null) bound vector raised to max |y| by the epilogue
};

typedef __attribute__((address_space(3))) void g6_lds_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int GM = 128, GN = 128, GCH = 2;         // pixels x couts per workgroup, 16-channel chunks per stage
// FMT 0: three bf16 terms by truncation, six products.  FMT 1: s a = h0 + h1 (two fp16 terms, round to nearest; s a power of two from a
// bound of max |a|: 16000 / max < s <= 32000 / max -- no Winograd sums here, the factor 4 of conv_wino2d_x6.hip is kept so that one
// rule serves all kernels), three products; the weights carry the fixed scale of the fp16 images (adm_split2_rows_f16).
template <int FMT> struct G6Fmt {
  static constexpr int TERMS = FMT ? 2 : 3;
  static constexpr int IMG = GCH * TERMS * 128 * 16;      // 16-bit elements of one operand image of a stage: [chunk][term][128 rows][16] = 24 / 16 KB
  static constexpr int QW = 2 * TERMS;                    // one-KB DMA instructions per consumer wave and stage
};
typedef _Float16 g6_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 g6_f16x2 __attribute__((ext_vector_type(2)));
__device__ inline float g6_scale(float amax) {      // = h3_scale of conv_wino2d_x6.hip
  if (!(amax > 0.f) || !(amax < 3e38f)) return 1.f;
  int e;
  frexpf(16000.f / amax, &e);
  return ldexpf(1.f, e - 1);
}
constexpr int G_RA = 2, G_RB = 4, G_D = 4;         // A slots, B slots, producer prefetch depth in stages

// plain v_sub_f32 (not the packed form): next to the bf16 MFMA of the consumer wave on the same SIMD plain VALU is ~93 % hidden,
// v_pk_add_f32 not at all (tools/overlap_probe2.hip; conv_wino2d_x6.hip)
__device__ __forceinline__ f32x4 g6_sub4(f32x4 a, f32x4 b) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) asm("v_sub_f32 %0, %1, %2" : "=v"(r[i]) : "v"(a[i]), "v"(b[i]));
  return r;
}
__device__ __forceinline__ void g6_split3(const f32x4 v, u32x2& t0, u32x2& t1, u32x2& t2) {
  f32x4 h, mh;
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = __uint_as_float(__float_as_uint(v[i]) & 0xFFFF0000u);
  const f32x4 r = g6_sub4(v, h);
#pragma unroll
  for (int i = 0; i < 4; ++i) mh[i] = __uint_as_float(__float_as_uint(r[i]) & 0xFFFF0000u);
  const f32x4 r2 = g6_sub4(r, mh);
  t0 = u32x2{__builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07060302u)};
  t1 = u32x2{__builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(r[3]), __float_as_uint(r[2]), 0x07060302u)};
  t2 = u32x2{__builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u),
             __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u)};
}
__device__ __forceinline__ void g6_split2(const f32x4 v, float s, u32x2& t0, u32x2& t1) {
  _Float16 h0[4], h1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float vs = v[i] * s;
    h0[i] = (_Float16)vs;
    h1[i] = (_Float16)(vs - (float)h0[i]);
  }
  t0 = u32x2{__builtin_bit_cast(unsigned, g6_f16x2{h0[0], h0[1]}), __builtin_bit_cast(unsigned, g6_f16x2{h0[2], h0[3]})};
  t1 = u32x2{__builtin_bit_cast(unsigned, g6_f16x2{h1[0], h1[1]}), __builtin_bit_cast(unsigned, g6_f16x2{h1[2], h1[3]})};
}
__device__ __forceinline__ void g6_barrier() {     // waits for this wave's LDS traffic only
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int FMT>
__global__ __launch_bounds__(512) void gemm_x6_kernel(G6P p) {
  constexpr int G_IMG = G6Fmt<FMT>::IMG, TERMS = G6Fmt<FMT>::TERMS, QW = G6Fmt<FMT>::QW;
  float sa = 1.f, inv_scale = 1.f;
  if (FMT) { sa = g6_scale(adm_amax_read(p.amax_x)); inv_scale = 1.f / (sa * p.wscale); }
  extern __shared__ __attribute__((aligned(16))) unsigned short smg[];
  unsigned short* As = smg;                        // [G_RA][chunk][term][128 pixels][16]
  unsigned short* Bs = smg + G_RA * G_IMG;         // [G_RB][chunk][term][128 couts][16]
  const int tid = threadIdx.x, lane = tid & 63, hw_wid = tid >> 6;
  const bool producer = hw_wid >= 4;
  const int wid = hw_wid & 3;
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap, m-fastest inside an n-tile (see conv_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int m0 = tm * GM, n0 = tn * GN;
  const int S = p.K >> 5;                          // stages of 32 channels
  constexpr unsigned OOB = 0x80000000u;

  if (producer) {
    // ================================================================ A operand
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
    const int ptid = wid * 64 + lane;
    // item j of a thread: pixel row (ptid >> 3) + 32 j, 16-byte quad ptid & 7 of the 128-byte stage row (chunk = quad >> 2)
    const int prow = ptid >> 3, quad = ptid & 7;
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + prow + 32 * j;
      voff[j] = (m < p.M) ? (unsigned)(((long)m * p.ldx + quad * 4) * 4) : OOB;
    }
    // LDS rows are 32 bytes with the halves of rows 8-15 (mod 16) swapped (conv_wino2d_x6.hip): physical half = logical ^ ((row >> 3) & 1)
    const int aq = quad & 3, ch = quad >> 2;
    unsigned short* la[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = prow + 32 * j;
      la[j] = As + (ch * TERMS * 128 + row) * 16 + ((((aq >> 1) ^ (row >> 3)) & 1) << 3) + (aq & 1) * 4;
    }
    f32x4 d[G_D][4];
    auto issue = [&](int set, int s) {             // loads of stage s (stages past the end read nothing)
      const int soff = s << 7;                     // 32 floats = 128 bytes per stage
#pragma unroll
      for (int j = 0; j < 4; ++j)
        d[set][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(s < S ? voff[j] : OOB), soff, 0));
    };
    auto store = [&](int set, int slot) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned short* l = la[j] + slot * G_IMG;
        if (FMT) {
          u32x2 h0, h1;
          g6_split2(d[set][j], sa, h0, h1);
          *reinterpret_cast<u32x2*>(l) = h0;
          *reinterpret_cast<u32x2*>(l + 128 * 16) = h1;
          continue;
        }
        u32x2 t0, t1, t2;
        g6_split3(d[set][j], t0, t1, t2);
        *reinterpret_cast<u32x2*>(l) = t0;
        *reinterpret_cast<u32x2*>(l + 128 * 16) = t1;
        *reinterpret_cast<u32x2*>(l + 2 * 128 * 16) = t2;
      }
    };
#pragma unroll
    for (int k = 0; k < G_D; ++k) issue(k, k);
    __builtin_amdgcn_sched_barrier(0);
    // barrier s separates "A(s) written" from compute(s); A(s) lives in slot s & 1; set s % 4 is refilled with stage s + 4
    for (int s0 = 0; s0 < S; s0 += G_D) {
#pragma unroll
      for (int k = 0; k < G_D; ++k) {
        if (s0 + k < S) {                          // (uniform)
          store(k, k & 1);
          __builtin_amdgcn_sched_barrier(0);
          issue(k, s0 + k + G_D);
          __builtin_amdgcn_sched_barrier(0);
          g6_barrier();
        }
      }
    }
    return;
  }

  // ================================================================== consumer waves: weight DMA, MFMA, epilogue
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.wbytes, 0x00020000);
  const int wm = wid >> 1, wn = wid & 1;           // 64-pixel block, 64-cout block
  const int lr = lane & 31, lh = lane >> 5;
  // B loader (LDS-DMA): 24 one-KB instructions per stage = (chunk, term) image x 32-row group; wave w issues q = 6w .. 6w+5.
  // Lane l of an instruction covers row 32 (q & 3) + (l >> 1), 16-byte half (l & 1) of the 32-byte row.
  unsigned b_voff[QW];
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int q = wid * QW + i, img = q >> 2;      // img = chunk * TERMS + term
    const int row = (q & 3) * 32 + (lane >> 1);
    const int n = n0 + row;
    const int half = (lane ^ (row >> 3)) & 1;       // logical half stored at physical half (lane & 1)
    b_voff[i] = (n < p.wrows) ? (unsigned)((((long)img * p.wrows + n) * 16 + half * 8) * 2) : OOB;
  }
  int ld_s = 0, ld_slot = 0;
  auto issue_b = [&]() {                           // weights of the next stage -> next ring slot
    const int kb = (ld_s * GCH * TERMS * p.wrows) << 5;   // stage block of six / four [chunk][term] images of wrows x 32 bytes
    unsigned short* dst = Bs + ld_slot * G_IMG + (wid * QW) * 512;
#pragma unroll
    for (int i = 0; i < QW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (g6_lds_void*)(dst + i * 512), 16, (int)b_voff[i], kb, 0, 0);
    ++ld_s;
    if (++ld_slot == G_RB) ld_slot = 0;
  };

  f32x16 acc[2][2], tot[2][2];                     // [pixel block][cout block] 32x32 tiles
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[a][b][r] = 0.f;
  const int fh = ((lh ^ (lr >> 3)) & 1) * 8;       // fragment: row = pixel / cout, 8 bf16 = 16 bytes at k = 8 (lane >> 5), swizzled half
  const int a_foff = (wm * 64 + lr) * 16 + fh;     // second block: + 32 rows
  const int b_foff = (wn * 64 + lr) * 16 + fh;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  issue_b();
  if (S > 1) issue_b();
  if (S > 2) issue_b();
  int slot_b = 0;
  for (int s = 0; s < S; ++s) {
    // B(s) was issued three stages ago; B(s+1), B(s+2) (six instructions each) may still be in flight
    if (s + 2 < S) { if (FMT) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    else if (s + 1 < S) { if (FMT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (s + 3 < S) issue_b();
    const unsigned short* Ab = As + (s & 1) * G_IMG + a_foff;
    const unsigned short* Bb = Bs + slot_b * G_IMG + b_foff;
    if (++slot_b == G_RB) slot_b = 0;
    const bool first = (s & 1) == 0;               // accumulator runs of two stages
#pragma unroll
    for (int ch = 0; ch < GCH; ++ch) {
      bf16x8 a[2][TERMS], b[2][TERMS];
#pragma unroll
      for (int k = 0; k < TERMS; ++k)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          a[blk][k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab + ((ch * TERMS + k) * 128 + blk * 32) * 16));
          b[blk][k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bb + ((ch * TERMS + k) * 128 + blk * 32) * 16));
        }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          f32x16 c;
          if (FMT) {                                 // three fp16 products, small ones first
            const g6_f16x8 a0 = __builtin_bit_cast(g6_f16x8, a[mi][0]), a1 = __builtin_bit_cast(g6_f16x8, a[mi][1]);
            const g6_f16x8 b0 = __builtin_bit_cast(g6_f16x8, b[ni][0]), b1 = __builtin_bit_cast(g6_f16x8, b[ni][1]);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, (first && ch == 0) ? zero : acc[mi][ni], 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
            continue;
          }
          if (first && ch == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][TERMS - 1], zero, 0, 0, 0);      // (uniform)
          else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][TERMS - 1], acc[mi][ni], 0, 0, 0);
#ifndef G6_HALF      // (diagnostic build of tools/bench_gemm_x6.cpp: three of the six products, to price a two-term format)
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][TERMS - 1], b[ni][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], c, 0, 0, 0);
#endif
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], c, 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], c, 0, 0, 0);
        }
    }
    if (!first || s + 1 == S) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) tot[mi][ni] += acc[mi][ni];
    }
  }

  // ---- epilogue.  C/D layout col = lane & 31 (cout), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel).  Branch-free: residual
  // loads and stores through buffer descriptors, masked lanes at an out-of-range offset (no residual = an empty descriptor).
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.rbytes : 0, 0x00020000);
  float am = 0.f;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + lr;
    const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int mb = m0 + wm * 64 + mi * 32 + 4 * lh;
      float rv[16];
      unsigned oy[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        const bool ok = m < p.M && n < p.N;
        oy[r] = ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)n) * 4u : OOB;
        const unsigned orr = ok ? ((unsigned)m * (unsigned)p.ldr + (unsigned)n) * 4u : OOB;
        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (int)orr, 0, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (FMT ? tot[mi][ni][r] * inv_scale : tot[mi][ni][r]) + bv + rv[r];
        if (oy[r] != OOB) am = fmaxf(am, fabsf(v));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (int)oy[r], 0, 0);
      }
    }
  }
  adm_amax_commit(am, p.amax_y);      // (all four consumer waves arrive here with all lanes)
}

// dst[k / 16][term][row][16] (bf16 bit patterns) <- the exact three-term split of src[row][k] (f32): the B operand of gemm_x6_kernel
__global__ void split3_rows_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols, int ld) {
  const long total = (long)rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / cols), c = (int)(i - (long)n * cols);
    const float a = src[(long)n * ld + c];
    const unsigned u = __float_as_uint(a);
    const float r = a - __uint_as_float(u & 0xFFFF0000u);
    const unsigned m = __float_as_uint(r);
    const float r2 = r - __uint_as_float(m & 0xFFFF0000u);
    unsigned short* d = dst + ((((long)(c >> 4) * 3) * rows + n) << 4) + (c & 15);
    const long term = (long)rows << 4;
    d[0] = (unsigned short)(u >> 16);
    d[term] = (unsigned short)(m >> 16);
    d[2 * term] = (unsigned short)(__float_as_uint(r2) >> 16);
  }
}

// fp16 format: dst[k / 16][term(2)][row][16] <- the two-term round-to-nearest split of scale * src[row][k]; *overflow is raised when a
// scaled value leaves the fp16 range
__global__ void split2_rows_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols, int ld, float scale,
                                   int* __restrict__ overflow) {
  const long total = (long)rows * cols;
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / cols), c = (int)(i - (long)n * cols);
    const float a = src[(long)n * ld + c] * scale;
    bad |= !(fabsf(a) < 65000.f);
    const _Float16 h0 = (_Float16)a, h1 = (_Float16)(a - (float)h0);
    unsigned short* d = dst + ((((long)(c >> 4) * 2) * rows + n) << 4) + (c & 15);
    d[0] = __builtin_bit_cast(unsigned short, h0);
    d[(long)rows << 4] = __builtin_bit_cast(unsigned short, h1);
  }
  if (bad && overflow) *overflow = 1;
}

}  // namespace

// dst (3 * rows * cols bf16, layout [cols/16][term][rows][16]) <- exact split a = a0 + a1 + a2 of src[rows][ld >= cols] (f32); cols % 16 == 0
extern "C" int adm_split3_rows(const float* src, void* dst, int rows, int cols, int ld, hipStream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 15) || ld < cols) return ADM_EINVAL;
  const long total = (long)rows * cols;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(split3_rows_kernel, dim3(grid), dim3(256), 0, stream, src, static_cast<unsigned short*>(dst), rows, cols, ld);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

static int gemm_x6_launch(const float* x, const void* w6, const float* bias, const float* res, float* y, long M, int K, int ldx, int N,
                          int wrows, int ldy, int ldr, hipStream_t stream, const float* amax_x, float wscale, float* amax_y) {
  const bool h3 = amax_x != nullptr;
  if (!x || !w6 || !y || M <= 0 || K <= 0 || (K & 31) || (ldx & 3) || N <= 0 || wrows < N) return ADM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)w6) & 15) return ADM_EINVAL;
  if (h3 && !(wscale > 0.f)) return ADM_EINVAL;
  G6P p;
  p.x = x; p.w = static_cast<const unsigned short*>(w6); p.bias = bias; p.res = res; p.y = y;
  p.amax_x = amax_x; p.wscale = wscale; p.amax_y = amax_y;
  const long xb = M * ldx * 4, wb = (h3 ? 2L : 3L) * wrows * K * 2, yb = M * ldy * 4, rb = res ? M * ldr * 4 : 0;
  if (M >= (1L << 30) || xb >= (1L << 31) || wb >= (1L << 31) || yb >= (1L << 31) || rb >= (1L << 31)) return ADM_EINVAL;
  p.M = (int)M; p.N = N; p.K = K; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr; p.wrows = wrows;
  p.xbytes = (int)xb; p.wbytes = (int)wb; p.ybytes = (int)yb; p.rbytes = (int)rb;
  p.tilesN = adm_cdiv(N, GN);
  constexpr int smem0 = (G_RA + G_RB) * G6Fmt<0>::IMG * (int)sizeof(unsigned short), smem1 = (G_RA + G_RB) * G6Fmt<1>::IMG * (int)sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem0) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem1) != hipSuccess)
      return ADM_ELAUNCH;
    attr_set = true;
  }
  const long grid = (long)adm_cdiv(M, GM) * p.tilesN;
  if (h3) hipLaunchKernelGGL(gemm_x6_kernel<1>, dim3((unsigned)grid), dim3(512), smem1, stream, p);
  else hipLaunchKernelGGL(gemm_x6_kernel<0>, dim3((unsigned)grid), dim3(512), smem0, stream, p);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}

// y[M][ldy] = x[M][ldx] (K channels) . w^T (+ bias) (+ res): w6 = adm_split3_rows of the packed operand [wrows >= N][K].
// K % 32 == 0, N % 4 == 0.
extern "C" int adm_gemm_x6(const float* x, const void* w6, const float* bias, const float* res, float* y, long M, int K, int ldx, int N,
                           int wrows, int ldy, int ldr, hipStream_t stream) {
  return gemm_x6_launch(x, w6, bias, res, y, M, K, ldx, N, wrows, ldy, ldr, stream, nullptr, 0.f, nullptr);
}
// adm_gemm_x6 that also raises the bound vector amax_y to max |y|
extern "C" int adm_gemm_x6_amax(const float* x, const void* w6, const float* bias, const float* res, float* y, long M, int K, int ldx, int N,
                                int wrows, int ldy, int ldr, float* amax_y, hipStream_t stream) {
  return gemm_x6_launch(x, w6, bias, res, y, M, K, ldx, N, wrows, ldy, ldr, stream, nullptr, 0.f, amax_y);
}
// ... on the fp16 format: wh = adm_split2_rows_f16 of the packed operand (scale wscale), amax_x = bound vector of |x| (include/adm_hip.h);
// amax_y (may be NULL): bound vector raised to max |y| (bias and residual included) for the kernels that consume y
extern "C" int adm_gemm_x6_h3(const float* x, const void* wh, const float* bias, const float* res, float* y, long M, int K, int ldx, int N,
                              int wrows, int ldy, int ldr, const float* amax_x, float wscale, float* amax_y, hipStream_t stream) {
  if (!amax_x) return ADM_EINVAL;
  return gemm_x6_launch(x, wh, bias, res, y, M, K, ldx, N, wrows, ldy, ldr, stream, amax_x, wscale, amax_y);
}
// dst (2 * rows * cols fp16, layout [cols/16][term(2)][rows][16]) <- two-term split of scale * src[rows][ld >= cols]; cols % 16 == 0
extern "C" int adm_split2_rows_f16(const float* src, void* dst, int rows, int cols, int ld, float scale, int* overflow, hipStream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 15) || ld < cols || !(scale > 0.f)) return ADM_EINVAL;
  const long total = (long)rows * cols;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(split2_rows_kernel, dim3(grid), dim3(256), 0, stream, src, static_cast<unsigned short*>(dst), rows, cols, ld, scale, overflow);
  ADM_CHECK_LAUNCH();
  return ADM_OK;
}
