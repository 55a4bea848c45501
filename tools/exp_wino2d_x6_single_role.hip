// EXPERIMENT (not part of the product; round 3): a single-role form of the split-bf16 2-D Winograd forward kernel.
// 256 threads = one wave per SIMD, every wave loads rows, transforms, splits, stores to LDS AND multiplies; NB = 1: 64-cout workgroup
// tiles (32 tiles x 32 couts per wave, as the product kernel), NB = 2: 128-cout tiles (32 x 64 per wave).
//
// Why it was tried: tools/overlap_probe2.hip shows that LDS operations of ANOTHER wave do not overlap with a wave's bf16 MFMAs on the
// same SIMD at all (0-3 %), while LDS operations the MFMA wave issues itself between its MFMAs cost ~6 cycles each -- and the
// ablation builds of the wave-specialised product kernel add up exactly (no phase hides another).
// What was measured (128 x 32 x 32 x 384 -> 384, one MI355X, same box as the product kernel's 1.30 ms):
//   NB = 1: 1.83 ms as first written; 1.63 ms with the group's other work interleaved MFMA by MFMA (sched_group_barrier).  Ablations:
//           bookkeeping + barriers + epilogue 0.375 (three stage sequencers and the row-offset logic in ONE instruction stream,
//           serial ahead of each stage's MFMAs), MFMA +0.348, fragment reads +0.135 (the specialised kernel: +0.254), weight DMA +0.152,
//           transform + LDS stores +0.18, global row loads +0.44 (two register sets = two stages of latency are not enough at 12 TB/s
//           of L1/L2 traffic; the specialised kernel's producers run four stages ahead).
//   NB = 2: does not fit: 128 accumulator + 128 output-row registers + two row sets + fragments need ~460 registers of which only 256
//           are addressable by VALU instructions (the rest are AGPRs: MFMA / load destinations only) -- 988 spills, 6.1 ms.
// Conclusion: the fragment reads do get cheaper inside the MFMA wave, but everything the producer waves did in parallel address
// arithmetic lands in front of the MFMAs, and the register file cannot hold the 128-cout tile that would pay for it.  Kept as a record.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops tools/exp_wino2d_x6_single_role.hip \
//         -Ladm_amd -ladm_hip -o tools/_exps && LD_LIBRARY_PATH=adm_amd tools/_exps
#include "../adm_amd/csrc/conv_wino2d_x6.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

namespace {
// Single-role version (round 3): 256 threads = ONE wave per SIMD, every wave loads, transforms, stores AND multiplies.
//
// Why (tools/overlap_probe2.hip, one MI355X): next to v_mfma_f32_32x32x16_bf16 of ANOTHER wave on the same SIMD
//     plain VALU (v_add_f32, v_and_b32, v_perm_b32) and buffer loads are 91-96 % hidden,
//     packed-f32 VALU (v_pk_add_f32, v_pk_fma_f32) 0-3 %,  ds_read_b128 / ds_write_b64 0-3 % (whatever the wave priorities),
// while LDS operations issued BY THE MFMA WAVE ITSELF between its MFMAs cost ~6 cycles each.  The ablation builds of the
// wave-specialised kernel above say the same thing from the other side: its phases add up exactly (0.248 bookkeeping + 0.368 MFMA +
// 0.254 fragment reads + 0.096 weight DMA + 0.113 transform / LDS stores + 0.223 global loads = 1.302 ms) -- a producer wave's LDS
// traffic does not overlap with the consumer wave of its SIMD.  So the LDS traffic moves into the MFMA waves' own instruction
// streams, the transform arithmetic is plain f32 (hidden in the MFMA shadow), and the 512 registers of a lone wave pay for
//   * a 128-cout workgroup tile: wave = 32 tiles x 64 couts (two 32-cout blocks per ex: 128 accumulator + 128 output-row registers),
//     i.e. half the A transforms / splits / LDS stores and 3/4 of the fragment reads per MFMA (9 reads per 12 MFMAs);
//   * two register sets of raw input rows (a set is y-combined at the top of a stage and reloaded at once: two full stages of
//     latency for every global load).
// Stage t: [barrier: A(t), B(t) visible] -> weight DMA B(t+1) -> y-combine A(t+1)'s rows, reload the set for stage t+3 -> four ex
// groups of {fragment reads for the next group, one quarter of A(t+1)'s transform + split + three ds_write_b64, twelve MFMAs}.
// LDS: A 2 x 24 KB + B 2 x 48 KB = 144 KB.  Arithmetic, stage order, weight layout and results are those of the kernel above.
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino2d_x6s_kernel(X6P p) {
  constexpr int XN = 64 * NB;                          // couts per workgroup
  constexpr int B_STAGE = 12 * XN * X6K;               // bf16 elements per B stage image
  extern __shared__ __attribute__((aligned(16))) unsigned short smem6[];
  unsigned short* As = smem6;                          // [2][4 ex][3 terms][64 tiles][16]
  unsigned short* Bs = smem6 + 2 * X6_A_STAGE;         // [2][4 ex][3 terms][XN couts][16]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int mt0 = tm * X6P_T, n0 = tn * XN;
  constexpr unsigned OOB = 0x80000000u;
  const int c_begin = (p.splitk > 1) ? (int)blockIdx.y * p.chunks_per_split : 0;
  const int chunks = (p.splitk > 1) ? min(p.chunks_per_split, (p.Cin >> 4) - c_begin) : (p.Cin >> 4);
  const int S = (p.up ? 3 : 4) * chunks;               // stages (even)

  // ---------------------------------------------------------------- A side (thread = tile, 16-byte channel quad)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
  const int pl = tid >> 2, aq = tid & 3;
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
  X6Seq ld; ld.init(chunks, p.up);
  unsigned a_voff[2][4];
  int voff_ey = -1;
  auto set_rows = [&]() {
    const int ey = ld.ey;
    const int iA = (ey == 0) ? 0 : (ey == 2) ? 2 : 1, iB = (ey == 3) ? 3 : (ey == 2) ? 1 : 2;     // A + sgn B: r0-r2, r1+r2, r2-r1, r1-r3
    const bool live = !ld.done();
    const bool vA = ((rowmask >> iA) & 1u) && live, vB = ((rowmask >> iB) & 1u) && live;
    const int offA = (p.up ? ((iA + 1) >> 1) - 1 : iA - 1) * (p.up ? p.Wh : p.W) * p.ldx * 4;
    const int offB = (p.up ? ((iB + 1) >> 1) - 1 : iB - 1) * (p.up ? p.Wh : p.W) * p.ldx * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool cv = (colmask >> j) & 1u;
      const int cj = (p.up ? ((j + 1) >> 1) - 1 : j - 1) * p.ldx * 4;
      a_voff[0][j] = (vA && cv) ? a_base + (unsigned)(offA + cj) : OOB;
      a_voff[1][j] = (vB && cv) ? a_base + (unsigned)(offB + cj) : OOB;
    }
    voff_ey = live ? ey : 4;
  };
  f32x4 dA[2][4], dB[2][4];
  int set_ey[2];
  auto issue_a = [&](int d) {                          // next stage of the load sequence -> register set d
    if (voff_ey != (ld.done() ? 4 : ld.ey)) set_rows();
    const int soff = (c_begin + ld.chunk()) << 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (X6_ABL & 1) { dA[d][j] = f32x4{1.f, 2.f, 3.f, (float)soff}; dB[d][j] = f32x4{0.5f, 0.25f, (float)j, 1.f}; continue; }
      dA[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[0][j], soff, 0));
      dB[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[1][j], soff, 0));
    }
    set_ey[d] = ld.ey;
    if (!ld.done()) ld.next(chunks);
  };
  auto ycombine = [&](int d, f32x4 (&e)[4]) {          // the pass's two input rows -> one row of the transformed patch: A + sgn B
    const float sg = set_ey[d] == 1 ? 1.f : -1.f;       // (pass 2 = r2 - r1: its rows are loaded in swapped roles, see set_rows)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(e[j][i]) : "v"(dB[d][j][i]), "v"(sg), "v"(dA[d][j][i]));
  };
  auto store_ex = [&](const f32x4 (&e)[4], int ex, unsigned short* dst) {   // column ex of B^T along x, split, three ds_write_b64
    if (X6_ABL & 2) return;
    const f32x4 v = ex == 0 ? p_sub4(e[0], e[2]) : ex == 1 ? p_add4(e[1], e[2]) : ex == 2 ? p_sub4(e[2], e[1]) : p_sub4(e[1], e[3]);
    u32x2 t0, t1, t2;
    split3_pack(v, t0, t1, t2);
    *reinterpret_cast<u32x2*>(dst + (ex * 3 + 0) * X6P_T * X6K) = t0;
    *reinterpret_cast<u32x2*>(dst + (ex * 3 + 1) * X6P_T * X6K) = t1;
    *reinterpret_cast<u32x2*>(dst + (ex * 3 + 2) * X6P_T * X6K) = t2;
  };

  // ---------------------------------------------------------------- B side (weights by LDS-DMA) and the MFMA tiles
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.wbytes, 0x00020000);
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  if (p.splitk > 1) {
    p.y = p.ws + (long)blockIdx.y * ((long)p.Mt * 4) * p.N;
    p.ldy = p.N; p.bias = nullptr; p.res = nullptr;
    const long sb = (long)p.Mt * 4 * p.N * 4;
    p.ybytes = sb < (1L << 31) ? (int)sb : 0;
  }
  // 12 images x (XN / 32) one-KB instructions per stage; wave w issues q = 6 NB w .. 6 NB (w + 1) - 1
  constexpr int QW = 6 * NB;
  unsigned b_voff[QW];
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int q = wid * QW + i, pt = q / (2 * NB), sub = q % (2 * NB);
    const int row = sub * 32 + (lane >> 1);
    const int n = n0 + row;
    const int half = (lane ^ (row >> 3)) & 1;
    b_voff[i] = (n < p.wrows) ? (unsigned)((((long)pt * p.wrows + n) * 16 + half * 8) * 2) : OOB;
  }
  X6Seq lb; lb.init(chunks, p.up);
  int ld_slot = 0;
  auto issue_b = [&]() {
    const int kb = ((lb.ey * (p.Cin >> 4) + c_begin + lb.chunk()) * 12 * p.wrows) << 5;
    unsigned short* dst = Bs + ld_slot * B_STAGE + (wid * QW) * 512;
    if (!(X6_ABL & 4)) {
#pragma unroll
      for (int i = 0; i < QW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (x6_lds_void*)(dst + i * 512), 16, (int)b_voff[i], kb, 0, 0);
    }
    if (!lb.done()) lb.next(chunks);
    ld_slot ^= 1;
  };      // (past the end the sequence stays on its last stage: a harmless re-read into the slot nobody uses)

  f32x16 acc[4][NB];
  f32x16 Y[2][2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[a][b][nb][r] = 0.f;
  const int a_foff = (wm * 32 + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;
  const int b_foff = (wn * 32 * NB + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;     // (+ nb * 32 rows: same swizzle phase)
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  X6Seq cs; cs.init(chunks, p.up);

  // ---------------------------------------------------------------- prologue: B(0), A(0) in LDS; sets 0 / 1 hold stages 1 / 2
  issue_b();
  issue_a(0);
  issue_a(1);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x4 e[4];
    ycombine(0, e);
#pragma unroll
    for (int ex = 0; ex < 4; ++ex) store_ex(e, ex, la);
    issue_a(0);                                         // stage 2 -> set 0 ... (set 1 holds stage 1)
  }
  // after the prologue: slot 0 = A(0); set 1 = rows of stage 1, set 0 = rows of stage 2.  Stage t consumes set (t + 1) & 1.
  auto stage = [&](int t, auto set_tag, auto first_tag) {
    constexpr int SET = decltype(set_tag)::value;
    // B(t) was issued one stage ago, the eight row loads of one set after it: vmcnt(8) covers the DMA
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned short* Ab = As + (t & 1) * X6_A_STAGE + a_foff;
    const unsigned short* Bb = Bs + (t & 1) * B_STAGE + b_foff;
    unsigned short* An = la + ((t + 1) & 1) * X6_A_STAGE;
    bf16x8 fa[2][3], fb[2][NB][3];
    auto frag = [&](int xi, int set) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (X6_ABL & 16) {
          fa[set][k] = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, (unsigned)t, 0x3f803f80u, (unsigned)xi});
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) fb[set][nb][k] = __builtin_bit_cast(bf16x8, u32x4{0x3f003f00u, (unsigned)k, 0x3f003f00u, (unsigned)lane});
          continue;
        }
        fa[set][k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab + (xi * 3 + k) * X6P_T * X6K));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          fb[set][nb][k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bb + ((xi * 3 + k) * XN + nb * 32) * X6K));
      }
    };
    // No branches from here to the end of the fourth ex group (one scheduling region per group): past the last stage the DMA and
    // the row loads read nothing (offsets out of range) and A(S) is stored into a slot nobody reads.
    frag(0, 0);
    issue_b();
    f32x4 e[4];
    ycombine(SET, e);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(SET);                                       // rows of stage t + 3
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      const int cur = xi & 1;
      if (xi < 3) frag(xi + 1, cur ^ 1);
      store_ex(e, xi, An);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (X6_ABL & 8) continue;
        const bf16x8 *a = fa[cur], *b = fb[cur][nb];
        f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], decltype(first_tag)::value ? zero : acc[xi][nb], 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        acc[xi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
      }
      // order inside the group: every MFMA is followed by its share of the group's other work -- the next group's fragment reads
      // first (they have the longest way to go), then the transform / split arithmetic, the three LDS stores last
#pragma unroll
      for (int i = 0; i < 6 * NB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // one MFMA
        if (xi < 3 && i < 3 * (1 + NB)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one fragment read
        __builtin_amdgcn_sched_group_barrier(0x002, (26 + 6 * NB - 1) / (6 * NB), 0);        // VALU share
        if (i >= 6 * NB - 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);              // one LDS store
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int ey = cs.ey;
    const bool last = cs.cc + 1 == cs.len;
    cs.next(chunks);
    if (last) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x16 z0 = acc[0][nb] + acc[1][nb] + acc[2][nb], z1 = sub16(sub16(acc[1][nb], acc[2][nb]), acc[3][nb]);
        if (ey <= 2) { Y[0][0][nb] += z0; Y[0][1][nb] += z1; }
        if (ey == 1) { Y[1][0][nb] += z0; Y[1][1][nb] += z1; }
        if (ey >= 2) { Y[1][0][nb] = sub16(Y[1][0][nb], z0); Y[1][1][nb] = sub16(Y[1][1][nb], z1); }
      }
    }
  };
  for (int t = 0; t < S; t += 2) {                      // S is even
    if (cs.cc == 0) stage(t, std::integral_constant<int, 1>{}, std::true_type{});
    else stage(t, std::integral_constant<int, 1>{}, std::false_type{});
    if (cs.cc == 0) stage(t + 1, std::integral_constant<int, 0>{}, std::true_type{});
    else stage(t + 1, std::integral_constant<int, 0>{}, std::false_type{});
  }

  // ---------------------------------------------------------------- epilogue (as above, per 32-cout block)
  const int tb = mt0 + wm * 32 + 4 * lh;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + wn * 32 * NB + nb * 32 + lr;
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
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, Y[a][c][nb][r] + bv + rv[a][c]), rs_y, (int)oy[a][c], 0, 0);
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
        float y0 = Y[a][0][nb][r] + bv, y1 = Y[a][1][nb][r] + bv;
        if (p.res) { y0 += p.res[px * p.ldr + n]; y1 += p.res[(px + 1) * p.ldr + n]; }
        p.y[px * p.ldy + n] = y0;
        p.y[(px + 1) * p.ldy + n] = y1;
      }
    }
  }
}

}  // namespace

template <int NB>
static int launch_s(const float* x, const void* wq6, float* y, int B, int H, int W, int Cin, int N) {
  X6P p;
  p.x = x; p.w = static_cast<const unsigned short*>(wq6); p.bias = nullptr; p.res = nullptr; p.y = y;
  const long Mt = (long)B * (H / 2) * (W / 2);
  p.Mt = (int)Mt; p.N = N; p.H = H; p.W = W; p.Hh = H / 2; p.Wh = W / 2; p.Cin = Cin; p.ldx = Cin; p.ldy = N; p.ldr = N;
  p.wrows = N; p.xbytes = (int)((long)B * H * W * Cin * 4); p.wbytes = (int)(48L * N * Cin * 2); p.plane = N * Cin; p.up = 0;
  p.splitk = 1; p.chunks_per_split = 0; p.ws = nullptr;
  p.ybytes = (int)((long)B * H * W * N * 4); p.rbytes = 0;
  constexpr int XN = 64 * NB;
  p.tilesN = adm_cdiv(N, XN);
  const int smem = (2 * X6_A_STAGE + 2 * 12 * XN * X6K) * (int)sizeof(unsigned short);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_x6s_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return -1;
  hipLaunchKernelGGL(wino2d_x6s_kernel<NB>, dim3((unsigned)(adm_cdiv(Mt, X6P_T) * p.tilesN)), dim3(256), smem, 0, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

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
  for (int nb = 1; nb <= 2; ++nb) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) nb == 1 ? launch_s<1>(x, w6, y, B, H, H, Cin, N) : launch_s<2>(x, w6, y, B, H, H, Cin, N);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) nb == 1 ? launch_s<1>(x, w6, y, B, H, H, Cin, N) : launch_s<2>(x, w6, y, B, H, H, Cin, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    std::vector<float> a(ny), b(ny);
    hipMemcpy(a.data(), y, ny * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y2, ny * 4, hipMemcpyDeviceToHost);
    double mx = 0;
    for (size_t i = 0; i < ny; ++i) mx = fmax(mx, fabs((double)a[i] - b[i]));
    printf("single-role NB=%d ABL=%d: %.3f ms (%.1f TFLOP/s algorithmic); max |y - product kernel| = %.3e\n", nb, X6_ABL, ms,
           2.0 * B * H * H * (double)N * 9 * Cin / ms / 1e9, mx);
  }
  return 0;
}
