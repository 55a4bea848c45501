// EXPERIMENT 3 (round 3): the split kernel on THREE fp16 products per f32 product instead of six bf16 products.
// a = a0 + a1 with a0 = rne_fp16(s a), a1 = rne_fp16(s a - a0) for a per-tensor power-of-two scale s that puts 4 max|a| just below the
// fp16 range: |s a - a0 - a1| <= 2^-24 |s a| while a1 is a normal fp16 number, <= 2^-25 absolutely below that (tools/fp16x3_accuracy.py:
// as accurate as the six-bf16 scheme against fp64 on normal, all-positive, heavy-tailed and outlier data); a b ~ a0 b0 + a0 b1 + a1 b0.
// Halves the MFMAs, the LDS fragment reads, the split stores and the weight DMA of a stage -- every phase of the product kernel that
// the ablations price -- at the cost of needing max|x| of the activation operand on the device (here: an absmax pass by this program).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops tools/exp_wino2d_h3.hip -Ladm_amd -ladm_hip \
//         -o tools/_exph3 && LD_LIBRARY_PATH=adm_amd tools/_exph3
#include "../adm_amd/csrc/conv_wino2d_x6.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int H3_A_STAGE = 4 * 2 * X6P_T * X6K;      // [4 ex][2 terms][64 tiles][16] fp16 = 16 KB
constexpr int H3_B_STAGE = 4 * 2 * X6N * X6K;

// power-of-two scale that puts 4 * amax (the Winograd input transform sums four values) below the fp16 range
__device__ __host__ inline float h3_scale(float amax) {
  if (!(amax > 0.f)) return 1.f;
  int e;
  frexpf(16000.f / amax, &e);                         // 16000 / amax = m 2^e, m in [0.5, 1)
  return ldexpf(1.f, e - 1);
}
// v * s = h0 + h1 (two fp16 terms, round to nearest), four channels -> two dwords per term
__device__ __forceinline__ void split2_pack(const f32x4 v, float s, u32x2& t0, u32x2& t1) {
  _Float16 h0[4], h1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float vs = v[i] * s;
    h0[i] = (_Float16)vs;
    h1[i] = (_Float16)(vs - (float)h0[i]);
  }
  t0 = u32x2{__builtin_bit_cast(unsigned, f16x2{h0[0], h0[1]}), __builtin_bit_cast(unsigned, f16x2{h0[2], h0[3]})};
  t1 = u32x2{__builtin_bit_cast(unsigned, f16x2{h1[0], h1[1]}), __builtin_bit_cast(unsigned, f16x2{h1[2], h1[3]})};
}
__device__ __forceinline__ void h3_store(const f32x4 (&e)[4], unsigned short* la, float s) {
  const f32x4 v[4] = {p_sub4(e[0], e[2]), p_add4(e[1], e[2]), p_sub4(e[2], e[1]), p_sub4(e[1], e[3])};
#pragma unroll
  for (int ex = 0; ex < 4; ++ex) {
    u32x2 t0, t1;
    split2_pack(v[ex], s, t0, t1);
    *reinterpret_cast<u32x2*>(la + (ex * 2 + 0) * X6P_T * X6K) = t0;
    *reinterpret_cast<u32x2*>(la + (ex * 2 + 1) * X6P_T * X6K) = t1;
  }
}

__global__ __launch_bounds__(512) void wino2d_h3_kernel(X6P p, const float* __restrict__ amax_x, const float* __restrict__ wscale) {
  const float sa = h3_scale(*amax_x), inv_scale = 1.f / (sa * *wscale);
  extern __shared__ __attribute__((aligned(16))) unsigned short smem6[];
  unsigned short* As = smem6;                          // [X6_RA][4 ex][3 terms][X6P_T][X6K]
  unsigned short* Bs = smem6 + X6_RA * H3_A_STAGE;     // [X6_RB][4 ex][3 terms][X6N][X6K]
  const int tid = threadIdx.x, lane = tid & 63, hw_wid = tid >> 6;
#ifdef X6_INTERLEAVE_ROLES
  const bool producer = hw_wid & 1;
  const int wid = hw_wid >> 1;                         // role-local wave index 0..3
#else
  const bool producer = hw_wid >= 4;
  const int wid = hw_wid & 3;
#endif
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap, m-fastest inside an n-tile (see conv_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tilesM = gridDim.x / p.tilesN;
  const int tm = bid % tilesM, tn = bid / tilesM;
  const int mt0 = tm * X6P_T, n0 = tn * X6N;
  constexpr unsigned OOB = 0x80000000u;

  const int c_begin = (p.splitk > 1) ? (int)blockIdx.y * p.chunks_per_split : 0;
  const int chunks = (p.splitk > 1) ? min(p.chunks_per_split, (p.Cin >> 4) - c_begin) : (p.Cin >> 4);     // 16-channel chunks (even)
  const int S = (p.up ? 3 : 4) * chunks;              // stages (an even number: Cin is a multiple of 32)

  if (producer) {
    // ================================================================ producer waves: A operand
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
    const int ptid = wid * 64 + lane;
    const int pl = ptid >> 2, aq = ptid & 3;          // tile, channel quad
    unsigned a_base = 0, colmask = 0, rowmask = 0;
    {
      const int t = mt0 + pl;
      if (t < p.Mt) {
        const int xp = t % p.Wh;
        const int u = t / p.Wh;
        const int ty = u % p.Hh, b = u / p.Hh;
        a_base = p.up ? (unsigned)((((long)b * p.Hh + ty) * p.Wh + xp) * p.ldx + aq * 4) * 4u        // source pixel (b, ty, xp)
                      : (unsigned)((((long)b * p.H + 2 * ty) * p.W + 2 * xp) * p.ldx + aq * 4) * 4u;
        colmask = (xp > 0 ? 1u : 0u) | 6u | (2 * xp + 2 < p.W ? 8u : 0u);
        rowmask = (ty > 0 ? 1u : 0u) | 6u | (2 * ty + 2 < p.H ? 8u : 0u);
      }
    }
    // rows are 32 bytes; a 16-lane group of a fragment read covers 16 rows at one 16-byte half, i.e. only half of the banks, unless
    // the halves of rows 8-15 (mod 16) are swapped: physical half = logical half ^ ((row >> 3) & 1), for A and B alike
    unsigned short* la = As + pl * X6K + ((((aq >> 1) ^ (pl >> 3)) & 1) << 3) + (aq & 1) * 4;
    X6Seq ld; ld.init(chunks, p.up);
    unsigned a_voff[2][4];
    int voff_ey = -1;
    auto set_rows = [&]() {     // pass ey combines input rows (iA, iB): 0: +r0 -r2   1: +r1 +r2   2: -r1 +r2   3: +r1 -r3; past the end: nothing
      const int ey = ld.ey;
      const int iA = (ey == 0) ? 0 : 1, iB = (ey == 3) ? 3 : 2;
      const bool live = !ld.done();
      const bool vA = ((rowmask >> iA) & 1u) && live, vB = ((rowmask >> iB) & 1u) && live;
      // up-sampled row 2ty - 1 + i reads source row ty + (i + 1) / 2 - 1 = ty - 1, ty, ty, ty + 1 (columns likewise)
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
    constexpr int D = 4;                              // stages in flight
    f32x4 dA[D][4], dB[D][4];
    int set_ey[D];
    auto issue = [&](int d) {                         // next stage of the sequence -> register set d
      if (voff_ey != (ld.done() ? 4 : ld.ey)) set_rows();
      const int soff = (c_begin + ld.chunk()) << 6;   // 16 floats = 64 bytes per chunk
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (X6_ABL & 1) { dA[d][j] = f32x4{1.f, 2.f, 3.f, (float)soff}; dB[d][j] = f32x4{0.5f, 0.25f, (float)j, 1.f}; continue; }
        dA[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[0][j], soff, 0));
        dB[d][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)a_voff[1][j], soff, 0));
      }
      set_ey[d] = ld.ey;
      if (!ld.done()) ld.next(chunks);
    };
    auto store = [&](int d, int slot) {
      f32x4 e[4];
      if (set_ey[d] == 1) {                           // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = p_add4(dA[d][j], dB[d][j]);
      } else if (set_ey[d] == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = p_sub4(dB[d][j], dA[d][j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = p_sub4(dA[d][j], dB[d][j]);
      }
      h3_store(e, la + slot * H3_A_STAGE, sa);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d);
    __builtin_amdgcn_sched_barrier(0);
    // barrier t separates "A(t) written" from compute(t); A(t) lives in slot t & 1; set t % D is refilled with stage t + D
    for (int t = 0; t < S; t += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (t + d < S) {                              // (uniform; S is even, a multiple of 4 without the up-sampling)
          const bool tl = X6_TL && p.ws && blockIdx.x == 0 && tid == 256 && t + d < 48;
          unsigned long long* TL = reinterpret_cast<unsigned long long*>(p.ws) + 1024 + (t + d) * 4;
          if (tl) TL[0] = __builtin_readcyclecounter();
          store(d, d & 1);
          __builtin_amdgcn_sched_barrier(0);
          if (tl) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TL[1] = __builtin_readcyclecounter(); }
          issue(d);                                   // stages past the end read nothing (all offsets out of range)
          __builtin_amdgcn_sched_barrier(0);
          if (tl) TL[2] = __builtin_readcyclecounter();
          x6_barrier();
          if (tl) TL[3] = __builtin_readcyclecounter();
        }
      }
    }
    return;
  }

  // ================================================================== consumer waves: weight DMA, MFMA, output transform
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, p.wbytes, 0x00020000);
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  if (p.splitk > 1) {
    p.y = p.ws + (long)blockIdx.y * ((long)p.Mt * 4) * p.N;
    p.ldy = p.N; p.bias = nullptr; p.res = nullptr;
    const long sb = (long)p.Mt * 4 * p.N * 4;
    p.ybytes = sb < (1L << 31) ? (int)sb : 0;
  }
  // B loader (LDS-DMA): 24 one-KB instructions per stage = (ex, term) image pt x 32-row half; wave w issues q = 6w .. 6w+5.
  // Lane l of an instruction covers row (q & 1) * 32 + (l >> 1), 16-byte half (l & 1) of the 32-byte row.
  unsigned b_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = wid * 4 + i, pt = q >> 1;
    const int row = (q & 1) * 32 + (lane >> 1);
    const int n = n0 + row;
    const int half = (lane ^ (row >> 3)) & 1;           // logical half stored at physical half (lane & 1)
    b_voff[i] = (n < p.wrows) ? (unsigned)((((long)pt * p.wrows + n) * 16 + half * 8) * 2) : OOB;
  }
  X6Seq lb; lb.init(chunks, p.up);
  int ld_slot = 0;
  auto issue_b = [&]() {                              // weights of the next stage of the sequence -> next ring slot
    const int kb = ((lb.ey * (p.Cin >> 4) + c_begin + lb.chunk()) * 8 * p.wrows) << 5;   // (ey, chunk) block of twelve [ex][term] images of wrows x 32 bytes
    unsigned short* dst = Bs + ld_slot * H3_B_STAGE + (wid * 4) * 512;           // 512 elements = one KB per instruction
    if (!(X6_ABL & 4)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (x6_lds_void*)(dst + i * 512), 16, (int)b_voff[i], kb, 0, 0);
    }
    lb.next(chunks);
    if (++ld_slot == X6_RB) ld_slot = 0;
  };

  f32x16 acc[4];
  f32x16 Y[2][2];                                     // [output row][output column of the pair]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[a][b][r] = 0.f;
  const int a_foff = (wm * 32 + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;   // fragment: row = tile / cout, 8 bf16 = 16 bytes at k = 8 (lane >> 5)
  const int b_foff = (wn * 32 + lr) * X6K + ((lh ^ (lr >> 3)) & 1) * 8;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  X6Seq cs; cs.init(chunks, p.up);
  int slot_b = 0;
  issue_b();
  issue_b();
  issue_b();
  for (int t = 0; t < S; ++t) {
    const bool tl = X6_TL && p.ws && blockIdx.x == 0 && tid == 0 && t < 48;
    unsigned long long* TL = reinterpret_cast<unsigned long long*>(p.ws) + t * 4;
    if (tl) TL[0] = __builtin_readcyclecounter();
    // B(t) was issued three stages ago; B(t+1) and B(t+2) (six instructions each) may still be in flight.  Plain s_barrier +
    // explicit counters: a release fence would drain the weight prefetch (vmcnt(0)).
    if (t + 2 < S) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (t + 1 < S) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tl) TL[1] = __builtin_readcyclecounter();
    if (t + 3 < S) issue_b();
    const unsigned short* Ab = As + (t & 1) * H3_A_STAGE + a_foff;
    const unsigned short* Bb = Bs + slot_b * H3_B_STAGE + b_foff;
    if (++slot_b == X6_RB) slot_b = 0;
    const bool first = cs.cc == 0;                    // first stage of a (block, ey) group: C = 0, no accumulator clearing
    // Fragment reads are issued ONE ex GROUP AHEAD of the MFMAs that use them (two register sets): left to itself the compiler
    // reads a group's six fragments right before its six MFMAs, so every group starts with an exposed LDS round trip (~130 cycles,
    // four times per stage) -- a read returns while the matrix pipe works only if it was issued before the chain it follows
    // (tools/overlap_probe2.hip: ds_read_b128 interleaved with MFMAs of the same wave costs ~6 cycles each, not a latency).
    f16x8 fa[2][2], fb[2][2];
    auto frag = [&](int xi, int set) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        fa[set][k] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Ab + (xi * 2 + k) * X6P_T * X6K));
        fb[set][k] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Bb + (xi * 2 + k) * X6N * X6K));
      }
    };
    auto products = [&](auto first_tag) {
      frag(0, 0);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const int cur = xi & 1;
        if (xi < 3) frag(xi + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);            // (keeps the next group's reads above this group's chain)
        if (X6_ABL & 8) continue;
        const f16x8 *a = fa[cur], *b = fb[cur];
        // small products first: a0 b1 + a1 b0, then a0 b0 (a1 b1 <= 2^-24 |a b| is dropped)
        f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], decltype(first_tag)::value ? zero : acc[xi], 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], c, 0, 0, 0);
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (first) products(std::true_type{});            // (wave-uniform)
    else products(std::false_type{});
    if (tl) TL[2] = __builtin_readcyclecounter();
    const int ey = cs.ey;
    const bool last = cs.cc + 1 == cs.len;
    cs.next(chunks);
    if (last) {
      // end of a (block, ey) group: A^T along x (z0 = m0 + m1 + m2, z1 = m1 - m2 - m3), then A^T along y into the output rows
      // (row 0 += Z(ey = 0, 1, 2); row 1 += Z(1) - Z(2) - Z(3)) with f32 adds
      const f32x16 z0 = acc[0] + acc[1] + acc[2], z1 = sub16(sub16(acc[1], acc[2]), acc[3]);
      if (ey <= 2) { Y[0][0] += z0; Y[0][1] += z1; }
      if (ey == 1) { Y[1][0] += z0; Y[1][1] += z1; }
      if (ey >= 2) { Y[1][0] = sub16(Y[1][0], z0); Y[1][1] = sub16(Y[1][1], z1); }
    }
  }

  // ---- epilogue.  C/D layout col = lane&31 (cout), row = (r&3) + 8 (r>>2) + 4 (lane>>5) (tile)
  const int n = n0 + wn * 32 + lr;
  const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
  const int tb = mt0 + wm * 32 + 4 * lh;
  if (p.ybytes > 0) {
    // branch-free: residual loads and output stores through buffer descriptors, masked lanes at an out-of-range offset (a missing
    // residual = an empty descriptor: reads return 0).  With `if`s per element the compiler serialises the 64 residual loads of a
    // thread, each waiting for the previous one -- tens of microseconds per workgroup.
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.rbytes : 0, 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tb + (r & 3) + 8 * (r >> 2);
      const bool ok = t < p.Mt && n < p.N;
      const int xp = t % p.Wh;
      const int u = t / p.Wh;                      // = b * Hh + ty
      const unsigned px0 = ((unsigned)u * 2u) * (unsigned)p.W + 2u * (unsigned)xp;     // pixel (b, 2ty, 2xp): (b*H + 2ty) * W + 2xp
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
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, Y[a][c][r] * inv_scale + bv + rv[a][c]), rs_y, (int)oy[a][c], 0, 0);
    }
    return;
  }
  if (n >= p.N) return;
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
      float y0 = Y[a][0][r] * inv_scale + bv, y1 = Y[a][1][r] * inv_scale + bv;
      if (p.res) { y0 += p.res[px * p.ldr + n]; y1 += p.res[(px + 1) * p.ldr + n]; }
      p.y[px * p.ldy + n] = y0;
      p.y[(px + 1) * p.ldy + n] = y1;
    }
  }
}


// weights: planes src[ey*4+ex][rows][cols] (f32) -> dst[ey][cols/16][ex][term(2)][rows][16] fp16, scaled by *scale
__global__ void split2_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols, const float* __restrict__ scale) {
  const long per = (long)rows * cols, total = per * 16;
  const int chunks = cols >> 4;
  const float s = *scale;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int img = (int)(i / per);
    const long rc = i - img * per;
    const int n = (int)(rc / cols), c = (int)(rc - (long)n * cols);
    const int ey = img >> 2, ex = img & 3;
    const float a = src[i] * s;
    const _Float16 h0 = (_Float16)a, h1 = (_Float16)(a - (float)h0);
    unsigned short* d = dst + ((((long)(ey * chunks + (c >> 4)) * 8 + ex * 2) * rows + n) << 4) + (c & 15);
    d[0] = __builtin_bit_cast(unsigned short, h0);
    d[(long)rows << 4] = __builtin_bit_cast(unsigned short, h1);
  }
}
__global__ void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out, float margin) {
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m * margin));
}
__global__ void wscale_kernel(const float* amax, float* scale) { *scale = h3_scale(*amax * 0.25f); }   // (weights are not summed: the x4 margin is not needed)
}  // namespace

static void run(int B, int H, int Cin, int N) {
  size_t nx = (size_t)B * H * H * Cin, nw = (size_t)16 * N * Cin, ny = (size_t)B * H * H * N;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.02f;
  float *x, *w, *y, *y2, *y3, *sc; void *w6, *wh;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); hipMalloc(&y2, ny * 4); hipMalloc(&y3, ny * 4);
  hipMalloc(&w6, nw * 6); hipMalloc(&wh, nw * 4); hipMalloc(&sc, 16); hipMemset(sc, 0, 16);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  adm_split3_bf16(w, w6, N, Cin, 0);
  adm_conv_fwd_wino2d_x6(x, w6, nullptr, nullptr, y2, nullptr, 0, B, H, H, Cin, Cin, N, N, N, N, 0);       // six bf16 products
  adm_conv_fwd_wino2d(x, w, nullptr, nullptr, y3, nullptr, 0, B, H, H, Cin, Cin, N, N, N, N, 0);             // f32 MFMA
  hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, 0, x, (long)nx, sc, 1.f);                      // sc[0] = max |x|
  hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, 0, w, (long)nw, sc + 1, 1.f);                  // sc[1] = max |U|
  hipLaunchKernelGGL(wscale_kernel, dim3(1), dim3(1), 0, 0, sc + 1, sc + 2);                                 // sc[2] = weight scale
  hipLaunchKernelGGL(split2_kernel, dim3(4096), dim3(256), 0, 0, w, (unsigned short*)wh, N, Cin, sc + 2);
  X6P p;
  p.x = x; p.w = static_cast<const unsigned short*>(wh); p.bias = nullptr; p.res = nullptr; p.y = y;
  const long Mt = (long)B * (H / 2) * (H / 2);
  p.Mt = (int)Mt; p.N = N; p.H = H; p.W = H; p.Hh = H / 2; p.Wh = H / 2; p.Cin = Cin; p.ldx = Cin; p.ldy = N; p.ldr = N;
  p.wrows = N; p.xbytes = (int)((long)B * H * H * Cin * 4); p.wbytes = (int)(32L * N * Cin * 2); p.plane = N * Cin; p.up = 0;
  p.splitk = 1; p.chunks_per_split = 0; p.ws = nullptr;
  p.ybytes = (int)((long)B * H * H * N * 4); p.rbytes = 0;
  p.tilesN = adm_cdiv(N, X6N);
  constexpr int smem = (X6_RA * H3_A_STAGE + X6_RB * H3_B_STAGE) * (int)sizeof(unsigned short);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&wino2d_h3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const unsigned grid = (unsigned)(adm_cdiv(Mt, X6P_T) * p.tilesN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(wino2d_h3_kernel, dim3(grid), dim3(512), smem, 0, p, sc, sc + 2);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(wino2d_h3_kernel, dim3(grid), dim3(512), smem, 0, p, sc, sc + 2);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  float ms6;
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) adm_conv_fwd_wino2d_x6(x, w6, nullptr, nullptr, y2, nullptr, 0, B, H, H, Cin, Cin, N, N, N, N, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms6, e0, e1); ms6 /= 10;
  std::vector<float> a(ny), b(ny), c(ny);
  hipMemcpy(a.data(), y, ny * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y2, ny * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), y3, ny * 4, hipMemcpyDeviceToHost);
  double d36 = 0, d3f = 0, d6f = 0, sc_ = 0;
  for (size_t i = 0; i < ny; ++i) {
    d36 = fmax(d36, fabs((double)a[i] - b[i])); d3f = fmax(d3f, fabs((double)a[i] - c[i])); d6f = fmax(d6f, fabs((double)b[i] - c[i]));
    sc_ = fmax(sc_, fabs((double)c[i]));
  }
  printf("B=%d H=%d Cin=%d N=%d: 3 x fp16 %.3f ms (%.1f TFLOP/s algorithmic) | 6 x bf16 %.3f ms | max diff: fp16x3 vs bf16x6 %.2e, fp16x3 vs f32 MFMA %.2e, bf16x6 vs f32 MFMA %.2e (max |y| %.2e)\n",
         B, H, Cin, N, ms, 2.0 * B * H * H * (double)N * 9 * Cin / ms / 1e9, ms6, d36, d3f, d6f, sc_);
  hipFree(x); hipFree(w); hipFree(y); hipFree(y2); hipFree(y3); hipFree(w6); hipFree(wh); hipFree(sc);
}

int main() {
  run(2, 8, 32, 64);
  run(128, 32, 384, 384);
  run(128, 32, 192, 192);
  run(128, 16, 384, 384);
  run(128, 16, 768, 384);
  run(128, 8, 384, 384);
  return 0;
}
