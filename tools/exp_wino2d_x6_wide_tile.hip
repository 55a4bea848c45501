// EXPERIMENT 2 (not part of the product; round 3): the split-bf16 2-D Winograd forward kernel with a 128-COUT workgroup tile,
// made to fit the register file by taking the ex index out of the accumulators (stage = one (ey, ex, chunk pair); a consumer wave
// owns 32 tiles x 64 couts with ONE accumulator per 32 x 32 block and folds it into the four output tiles with its two signs).
// Results are correct (same sums as the product kernel, checked against it by this program); 224 registers, no spills.
//
// Measured (128 x 32 x 32 x 384 -> 384, one MI355X; the product kernel: 1.27-1.31 ms on the same box):
//   * fold in one piece after the group's last MFMA ............ 1.34 ms
//   * fold split around the stage boundary (this file) .......... 1.43 ms
//   * two accumulator sets (fold behind the next group's MFMAs) . does not fit (256 registers, 492-923 spills)
//   ablations of the first form: everything but MFMAs and the fold 0.80 ms; the fold adds 0.44 ms, the MFMAs 0.10 ms.
//   A cycle-counter trace says why: an empty stage of this kernel costs ~1050 cycles and a fold stage ~1640 against 768 cycles
//   of MFMA work per stage -- the per-stage fixed costs (barrier round, stage sequencing, DMA issue, the exposed first fragment
//   read) do not shrink with the tile, the number of stages per output doubles, and the fold (~70 fmas that need the group's last
//   MFMA results) drains the matrix pipe 96 times per workgroup.  Halving the producer work per MFMA and the fragment reads by 1/4
//   -- what the 128-cout tile is for -- is worth less than that.  The product kernel keeps its 64-cout tile.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops tools/exp_wino2d_x6_wide_tile.hip \
//         -Ladm_amd -ladm_hip -o tools/_expw && LD_LIBRARY_PATH=adm_amd tools/_expw
#include "../adm_amd/csrc/conv_wino2d_x6.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

namespace {
// Round 3: the same convolution with a 128-COUT workgroup tile -- half the input loads, transforms, splits and LDS stores per
// MFMA and 3/4 of the fragment reads -- made to fit the register file by taking the ex index out of the accumulators.
//
// The kernel above keeps four accumulator tiles (one per ex) next to the four output tiles Y of a 32 x 32 block: 8 x 16 registers
// per block, so a consumer wave (256 registers at two waves per SIMD) holds ONE block per ex.  Here a stage is one (ey, EX, chunk
// pair): the wave owns 32 tiles x 64 couts = two blocks, with one accumulator per block (32 registers) + their output tiles
// (128), and each A fragment feeds both blocks.  Stage order: K block of X6N_KBP chunk pairs > ey > ex > pair; an accumulator
// runs over the pairs of a block (<= 24 matrix adds from C = 0) and is then FOLDED into the output tiles with its two signs
// (column ex of A^T along x, row ey along y: Y[a][b] += sy[a] sx[b] m, plain f32 fma) -- the same sums as above, term by term.
// The producers see ex as the INNER index: the y-combined rows e[pair][chunk][4] of a whole (block, ey) group stay in their
// registers (64) for its 4 x pairs stages, each of which transforms / splits / stores one ex column of two chunks (six
// ds_write_b64), and the raw rows of the NEXT group are loaded in one burst eight stages ahead (128 registers).
// LDS: A[2][2 chunks][3 terms][64 tiles][16] (2 x 12 KB) + B[4][2 chunks][3 terms][128 couts][16] (4 x 24 KB) = 120 KB.
// Measured costs of the kernel above add up phase by phase (bookkeeping 0.25, MFMA 0.37, fragment reads 0.25, weight DMA 0.10,
// transform + stores 0.11, row loads 0.22 ms on 128 x 32 x 32 x 384 -> 384): this form halves the last two and cuts the reads by 1/4.
#ifndef X6N_DBG
#define X6N_DBG 0      // diagnostic builds: 1 no fold, 2 no group combine / loads, 4 no epilogue stores
#endif
#ifndef X6N_KBP_V
#define X6N_KBP_V 2
#endif
constexpr int X6N_KBP = X6N_KBP_V;                             // chunk pairs per K block (= X6_KB chunks)
constexpr int X6N_XN = 128;
constexpr int X6N_A_STAGE = 6 * X6P_T * X6K;           // bf16 elements: [chunk][term] images of 64 tiles
constexpr int X6N_B_STAGE = 6 * X6N_XN * X6K;
struct X6nSeq {
  int p0, len, ey, ex, pp, skip2;
  __device__ __forceinline__ void init(int pairs, int up) { p0 = 0; len = min(X6N_KBP, pairs); ey = 0; ex = 0; pp = 0; skip2 = up; }
  __device__ __forceinline__ int pair() const { return p0 + pp; }
  __device__ __forceinline__ bool done() const { return len <= 0; }
  __device__ __forceinline__ void next(int pairs) {
    if (++pp == len) {
      pp = 0;
      if (++ex == 4) {
        ex = 0;
        ++ey;
        if (skip2 && ey == 2) ++ey;
        if (ey == 4) { ey = 0; p0 += len; len = min(X6N_KBP, pairs - p0); }
      }
    }
  }
};

__global__ __launch_bounds__(512) void wino2d_x6n_kernel(X6P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem6[];
  unsigned short* As = smem6;                          // [2][2 chunks][3 terms][64][16]
  unsigned short* Bs = smem6 + 2 * X6N_A_STAGE;        // [4][2 chunks][3 terms][128][16]
  const int tid = threadIdx.x, lane = tid & 63, hw_wid = tid >> 6;
  const bool producer = hw_wid >= 4;
  const int wid = hw_wid & 3;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int mt0 = tm * X6P_T, n0 = tn * X6N_XN;
  constexpr unsigned OOB = 0x80000000u;
  const int c_begin = (p.splitk > 1) ? (int)blockIdx.y * p.chunks_per_split : 0;
  const int chunks = (p.splitk > 1) ? min(p.chunks_per_split, (p.Cin >> 4) - c_begin) : (p.Cin >> 4);     // even
  const int pairs = chunks >> 1;
  const int S = (p.up ? 3 : 4) * 4 * pairs;

  if (producer) {
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
    const int ptid = wid * 64 + lane;
    const int pl = ptid >> 2, aq = ptid & 3;
    unsigned a_base = 0, colmask = 0, rowmask = 0;
    {
      const int t = mt0 + pl;
      if (t < p.Mt) {
        const int xp = t % p.Wh;
        const int u = t / p.Wh;
        const int ty = u % p.Hh, b = u / p.Hh;
        a_base = p.up ? (unsigned)((((long)b * p.Hh + ty) * p.Wh + xp) * p.ldx + aq * 4) * 4u
                      : (unsigned)((((long)b * p.H + 2 * ty) * p.W + 2 * xp) * p.ldx + aq * 4) * 4u;
        colmask = (xp > 0 ? 1u : 0u) | 6u | (2 * xp + 2 < p.W ? 8u : 0u);
        rowmask = (ty > 0 ? 1u : 0u) | 6u | (2 * ty + 2 < p.H ? 8u : 0u);
      }
    }
    unsigned short* la = As + pl * X6K + ((((aq >> 1) ^ (pl >> 3)) & 1) << 3) + (aq & 1) * 4;
    // ---- group loader: the raw rows of one (K block, ey) group = X6N_KBP pairs x 2 chunks x (2 rows x 4 pixels)
    int g_p0 = 0, g_len = min(X6N_KBP, pairs), g_ey = 0;          // the NEXT group to load
    auto g_next = [&]() {
      ++g_ey;
      if (p.up && g_ey == 2) ++g_ey;
      if (g_ey == 4) { g_ey = 0; g_p0 += g_len; g_len = min(X6N_KBP, pairs - g_p0); }
    };
    f32x4 dA[X6N_KBP][2][4], dB[X6N_KBP][2][4];
    float raw_sg = -1.f;
    auto load_group = [&]() {                          // rows (iA, iB) of pass ey, combined later as A + sgn B: r0-r2, r1+r2, r2-r1, r1-r3
      const int ey = g_ey;
      const bool live = g_len > 0;
      const int iA = (ey == 0) ? 0 : (ey == 2) ? 2 : 1, iB = (ey == 3) ? 3 : (ey == 2) ? 1 : 2;
      const bool vA = ((rowmask >> iA) & 1u) && live, vB = ((rowmask >> iB) & 1u) && live;
      const int offA = (p.up ? ((iA + 1) >> 1) - 1 : iA - 1) * (p.up ? p.Wh : p.W) * p.ldx * 4;
      const int offB = (p.up ? ((iB + 1) >> 1) - 1 : iB - 1) * (p.up ? p.Wh : p.W) * p.ldx * 4;
      unsigned vo[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cv = (colmask >> j) & 1u;
        const int cj = (p.up ? ((j + 1) >> 1) - 1 : j - 1) * p.ldx * 4;
        vo[0][j] = (vA && cv) ? a_base + (unsigned)(offA + cj) : OOB;
        vo[1][j] = (vB && cv) ? a_base + (unsigned)(offB + cj) : OOB;
      }
#pragma unroll
      for (int pp = 0; pp < X6N_KBP; ++pp)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const bool have = pp < g_len;
          const int soff = (c_begin + 2 * (g_p0 + pp) + ch) << 6;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (X6_ABL & 1) { dA[pp][ch][j] = f32x4{1.f, 2.f, 3.f, (float)soff}; dB[pp][ch][j] = f32x4{0.5f, 0.25f, (float)j, 1.f}; continue; }
            dA[pp][ch][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, have ? (int)vo[0][j] : (int)OOB, soff, 0));
            dB[pp][ch][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, have ? (int)vo[1][j] : (int)OOB, soff, 0));
          }
        }
      raw_sg = ey == 1 ? 1.f : -1.f;
      if (live) g_next();
    };
    f32x4 e[X6N_KBP][2][4];
    auto combine = [&]() {                             // raw rows of the loaded group -> e (frees the raw set for the next burst)
      const float sg = raw_sg;
#pragma unroll
      for (int pp = 0; pp < X6N_KBP; ++pp)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              asm("v_fma_f32 %0, %1, %2, %3" : "=v"(e[pp][ch][j][i]) : "v"(dB[pp][ch][j][i]), "v"(sg), "v"(dA[pp][ch][j][i]));
    };
    auto store = [&](int pp, int ex, int slot) {       // column ex of B^T along x of both chunks of pair pp -> three terms -> LDS
      if (X6_ABL & 2) return;
      unsigned short* dst = la + slot * X6N_A_STAGE;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const f32x4 (&ee)[4] = e[pp][ch];
        const f32x4 v = ex == 0 ? p_sub4(ee[0], ee[2]) : ex == 1 ? p_add4(ee[1], ee[2]) : ex == 2 ? p_sub4(ee[2], ee[1]) : p_sub4(ee[1], ee[3]);
        u32x2 t0, t1, t2;
        split3_pack(v, t0, t1, t2);
        *reinterpret_cast<u32x2*>(dst + (ch * 3 + 0) * X6P_T * X6K) = t0;
        *reinterpret_cast<u32x2*>(dst + (ch * 3 + 1) * X6P_T * X6K) = t1;
        *reinterpret_cast<u32x2*>(dst + (ch * 3 + 2) * X6P_T * X6K) = t2;
      }
    };
    X6nSeq ps; ps.init(pairs, p.up);
    load_group();                                      // group 0
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < S; ++t) {
      if (ps.ex == 0 && ps.pp == 0 && !(X6N_DBG & 2)) {                  // first stage of a (block, ey) group (uniform)
        combine();
        __builtin_amdgcn_sched_barrier(0);
        load_group();                                  // the next group's rows, 4 x pairs stages ahead (past the end: nothing)
        __builtin_amdgcn_sched_barrier(0);
      }
      // (pp, ex) are uniform run-time values: the register arrays are indexed statically inside the switch
      const int sl = t & 1;
      if (ps.pp == 0) {
        switch (ps.ex) { case 0: store(0, 0, sl); break; case 1: store(0, 1, sl); break; case 2: store(0, 2, sl); break; default: store(0, 3, sl); }
      } else {
        switch (ps.ex) { case 0: store(1, 0, sl); break; case 1: store(1, 1, sl); break; case 2: store(1, 2, sl); break; default: store(1, 3, sl); }
      }
      ps.next(pairs);
      x6_barrier();
    }
    return;
  }

  // ================================================================== consumers
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.wbytes, 0x00020000);
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  if (p.splitk > 1) {
    p.y = p.ws + (long)blockIdx.y * ((long)p.Mt * 4) * p.N;
    p.ldy = p.N; p.bias = nullptr; p.res = nullptr;
    const long sb = (long)p.Mt * 4 * p.N * 4;
    p.ybytes = sb < (1L << 31) ? (int)sb : 0;
  }
  // weight DMA: 6 images [chunk][term] x 128 rows x 32 bytes per stage = 24 one-KB instructions; wave w issues q = 6w .. 6w + 5:
  // image q >> 2, rows (q & 3) * 32 + (lane >> 1)
  unsigned b_voff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = wid * 6 + i, img = q >> 2, ch = img / 3, term = img % 3;
    const int row = (q & 3) * 32 + (lane >> 1);
    const int n = n0 + row;
    const int half = (lane ^ (row >> 3)) & 1;
    b_voff[i] = (n < p.wrows) ? (unsigned)(((((long)ch * 12 + term) * p.wrows + n) * 16 + half * 8) * 2) : OOB;
  }
  X6nSeq lb; lb.init(pairs, p.up);
  int ld_slot = 0;
  auto issue_b = [&]() {
    // images (ex, term) of chunk c sit at ((ey * chunks_total + c) * 12 + ex * 3 + term) * wrows * 32 bytes
    const int kb = (((lb.ey * (p.Cin >> 4) + c_begin + 2 * lb.pair()) * 12 + lb.ex * 3) * p.wrows) << 5;
    unsigned short* dst = Bs + ld_slot * X6N_B_STAGE + (wid * 6) * 512;
    if (!(X6_ABL & 4)) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (x6_lds_void*)(dst + i * 512), 16, (int)b_voff[i], kb, 0, 0);
    }
    lb.next(pairs);
    if (++ld_slot == X6_RB) ld_slot = 0;
  };

  // The fold is SPLIT around the stage boundary: block 0's accumulator is folded right after the group's last MFMAs have been issued
  // (its own chain is done by then; block 1's still runs), block 1's after the next stage's barrier and first fragment reads have been
  // issued (its chain is done, the reads are in flight).  Folded in one piece after the group's last MFMA, the matrix pipe drains,
  // idles for the ~70 plain fmas and then waits out a barrier and an LDS round trip: measured 0.44 of 1.34 ms.
  f32x16 acc[2];                                       // [block]
  f32x16 Y[2][2][2];                                   // [output row][output column][block]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[a][b][k][r] = 0.f;
  const int a_foff = (wm * 32 + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;
  const int b_foff = (wn * 64 + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;
  X6nSeq cs; cs.init(pairs, p.up);
  int slot_b = 0;
  int pend_ey = -1, pend_ex = 0;                       // the group whose block 1 still waits to be folded (-1: none)
  // fold of one block: A^T along x (column ex: z0 takes ex = 0, 1, 2; z1 takes +ex 1, -ex 2, -ex 3) and along y (row ey likewise)
  auto fold = [&](auto blk_tag, int ey, int ex) {
    constexpr int K = decltype(blk_tag)::value;
    if (X6N_DBG & 1) return;
    const float sy[2] = {ey <= 2 ? 1.f : 0.f, ey == 1 ? 1.f : ey >= 2 ? -1.f : 0.f};
    const float sx[2] = {ex <= 2 ? 1.f : 0.f, ex == 1 ? 1.f : ex >= 2 ? -1.f : 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float sgn = sy[a] * sx[b];
        if (sgn != 0.f) {                              // (uniform)
#pragma unroll
          for (int r = 0; r < 16; ++r) Y[a][b][K][r] = __builtin_fmaf(acc[K][r], sgn, Y[a][b][K][r]);
        }
      }
  };
  issue_b();
  issue_b();
  issue_b();
  for (int t = 0; t < S; ++t) {
    if (t + 2 < S) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (t + 1 < S) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 3 < S) issue_b();
    const unsigned short* Ab = As + (t & 1) * X6N_A_STAGE + a_foff;
    const unsigned short* Bb = Bs + slot_b * X6N_B_STAGE + b_foff;
    if (++slot_b == X6_RB) slot_b = 0;
    const bool first = cs.pp == 0;
    const int ey = cs.ey, ex = cs.ex;
    const bool last = cs.pp + 1 == cs.len;
    cs.next(pairs);
    // four groups of six MFMAs: (chunk 0, block 0), (chunk 0, block 1), (chunk 1, block 0), (chunk 1, block 1); the A fragments of a
    // chunk serve both blocks.  One register set per operand: the next group's fragments are read right after this group's MFMAs
    // have been ISSUED (they take their operands at issue; the reads return while the chain executes).
    bf16x8 fa[3], fb[3];
    auto rd_a = [&](int ch) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (X6_ABL & 16) { fa[k] = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, (unsigned)t, 0x3f803f80u, (unsigned)ch}); continue; }
        fa[k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab + (ch * 3 + k) * X6P_T * X6K));
      }
    };
    auto rd_b = [&](int ch, int blk) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (X6_ABL & 16) { fb[k] = __builtin_bit_cast(bf16x8, u32x4{0x3f003f00u, (unsigned)k, 0x3f003f00u, (unsigned)lane}); continue; }
        fb[k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bb + ((ch * 3 + k) * X6N_XN + blk * 32) * X6K));
      }
    };
    rd_a(0);
    rd_b(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (pend_ey >= 0) {                                // block 1 of the previous group, while the first fragments are on their way
      fold(std::integral_constant<int, 1>{}, pend_ey, pend_ex);
      pend_ey = -1;
      __builtin_amdgcn_sched_barrier(0);
    }
    auto products = [&](auto first_tag) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = g >> 1, blk = g & 1;
        if (!(X6_ABL & 8)) {
          f32x16 c;
          if (decltype(first_tag)::value && ch == 0)   // first stage of a group: the chain starts from the inline constant 0
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(fa[0]), "v"(fb[2]));
          else
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc[blk], 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], c, 0, 0, 0);
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], c, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g == 1) rd_a(1);
        if (g < 3) rd_b((g + 1) >> 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (first) products(std::true_type{});
    else products(std::false_type{});
    if (last) {                                        // block 0 now (behind block 1's last chain), block 1 after the next barrier
      unsigned long long t0 = 0;
      if (X6N_DBG & 8) t0 = __builtin_readcyclecounter();
      fold(std::integral_constant<int, 0>{}, ey, ex);
      if ((X6N_DBG & 8) && p.ws && blockIdx.x == 0 && tid == 0 && t < 64) {
        __builtin_amdgcn_sched_barrier(0);
        reinterpret_cast<unsigned long long*>(p.ws)[t] = __builtin_readcyclecounter() - t0;
      }
      pend_ey = ey; pend_ex = ex;
    }
    if ((X6N_DBG & 8) && p.ws && blockIdx.x == 0 && tid == 0 && t < 64) reinterpret_cast<unsigned long long*>(p.ws)[64 + t] = __builtin_readcyclecounter();
  }
  if (pend_ey >= 0) fold(std::integral_constant<int, 1>{}, pend_ey, pend_ex);

  // ---- epilogue (per 32-cout block)
  const int tb = mt0 + wm * 32 + 4 * lh;
  if ((X6N_DBG & 4) && Y[0][0][0][0] != 12345.f) return;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int n = n0 + wn * 64 + k * 32 + lr;
    const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    if (p.ybytes > 0) {
      const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.ybytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.rbytes : 0, 0x00020000);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = tb + (r & 3) + 8 * (r >> 2);
        const bool ok = t < p.Mt && n < p.N;
        const int xp = t % p.Wh;
        const int u = t / p.Wh;
        const unsigned px0 = ((unsigned)u * 2u) * (unsigned)p.W + 2u * (unsigned)xp;
        unsigned oy[2][2], orr[2][2];
        float rv[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const unsigned px = px0 + (unsigned)a * (unsigned)p.W + (unsigned)c;
            oy[a][c] = ok ? (px * (unsigned)p.ldy + (unsigned)n) * 4u : OOB;
            orr[a][c] = ok ? (px * (unsigned)p.ldr + (unsigned)n) * 4u : OOB;
            rv[a][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (int)orr[a][c], 0, 0));
          }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, Y[a][c][k][r] + bv + rv[a][c]), rs_y, (int)oy[a][c], 0, 0);
      }
      continue;
    }
    if (n >= p.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tb + (r & 3) + 8 * (r >> 2);
      if (t >= p.Mt) continue;
      const int xp = t % p.Wh;
      const int u = t / p.Wh;
      const long px0 = ((long)u * 2) * p.W + 2 * xp;
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
}

}  // namespace

int main() {
  const int B = 128, H = 32, Cin = 384, N = 384;
  size_t nx = (size_t)B * H * H * Cin, nw = (size_t)16 * N * Cin, ny = (size_t)B * H * H * N;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.02f;
  float *x, *w, *y, *y2; void* w6;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); hipMalloc(&y2, ny * 4); hipMalloc(&w6, nw * 6);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  adm_split3_bf16(w, w6, N, Cin, 0);
  adm_conv_fwd_wino2d_x6(x, w6, nullptr, nullptr, y2, nullptr, 0, B, H, H, Cin, Cin, N, N, N, N, 0);
  X6P p;
  p.x = x; p.w = static_cast<const unsigned short*>(w6); p.bias = nullptr; p.res = nullptr; p.y = y;
  const long Mt = (long)B * (H / 2) * (H / 2);
  p.Mt = (int)Mt; p.N = N; p.H = H; p.W = H; p.Hh = H / 2; p.Wh = H / 2; p.Cin = Cin; p.ldx = Cin; p.ldy = N; p.ldr = N;
  p.wrows = N; p.xbytes = (int)((long)B * H * H * Cin * 4); p.wbytes = (int)(48L * N * Cin * 2); p.plane = N * Cin; p.up = 0;
  p.splitk = 1; p.chunks_per_split = 0; p.ws = nullptr;
  p.ybytes = (int)((long)B * H * H * N * 4); p.rbytes = 0;
  p.tilesN = adm_cdiv(N, X6N_XN);
  constexpr int smem_n = (2 * X6N_A_STAGE + X6_RB * X6N_B_STAGE) * (int)sizeof(unsigned short);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_x6n_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem_n);
  const unsigned grid = (unsigned)(adm_cdiv(Mt, X6P_T) * p.tilesN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(wino2d_x6n_kernel, dim3(grid), dim3(512), smem_n, 0, p);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(wino2d_x6n_kernel, dim3(grid), dim3(512), smem_n, 0, p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  std::vector<float> a(ny), b(ny);
  hipMemcpy(a.data(), y, ny * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y2, ny * 4, hipMemcpyDeviceToHost);
  double mx = 0;
  for (size_t i = 0; i < ny; ++i) mx = fmax(mx, fabs((double)a[i] - b[i]));
  printf("128-cout tile, ex in time: %.3f ms (%.1f TFLOP/s algorithmic); max |y - product kernel| = %.3e\n", ms,
         2.0 * B * H * H * (double)N * 9 * Cin / ms / 1e9, mx);
  return 0;
}
