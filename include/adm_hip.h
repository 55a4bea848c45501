/* adm_hip.h -- C ABI of libadm_hip.so: the MI355X (gfx950) hot path of DDM's UNet + analytic schedule.
 *
 * The reference (zacz08/ADM, a DDM fork) has no FFI / plugin ABI: its hot path is PyTorch ATen calls
 * made from Python modules.  This header therefore defines the boundary a
 * maintainer binds instead of those ATen calls; each entry point cites the reference lines whose
 * arithmetic it replaces (paths relative to /root/reference).  INTEGRATION.md shows the ctypes
 * binding and the two-line change in the reference's modules.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; activations are NHWC
 *     ([B][H][W][C], C contiguous; "ld*" = floats between consecutive pixels, >= C)
 *   - no allocation, no synchronisation, no host<->device copies inside any call: work is enqueued on
 *     `stream` (a hipStream_t) and the call returns; graph-capture safe
 *   - return 0 on success, -22 (EINVAL) for a shape/alignment the kernels do not support,
 *     -5 if the launch itself failed.  Nothing falls back to another implementation.
 *   - channel counts seen by the GEMM-shaped kernels are multiples of 32 (the host pads 3 -> 32 for
 *     the stem / heads; packed weights carry the zero padding)
 */
#ifndef ADM_HIP_H
#define ADM_HIP_H
#include <stdint.h>
/* identical to HIP's own typedef, so this header needs no HIP include (C11 allows the repeat) */
typedef struct ihipStream_t* hipStream_t;

#ifdef __cplusplus
/* A BOUND VECTOR (the `amax` arguments below): ADM_AMAX_SLOTS floats at a stride of ADM_AMAX_STRIDE floats (one cache line each), zeroed by
 * the caller; the bound is the maximum of the slots.  The kernels that produce a tensor raise the slots with atomicMax, every wave
 * on its own slot (tens of thousands of waves raising ONE address serialise in the L2: 185 us for a 22 us launch); the kernels that
 * consume the tensor read all slots once per workgroup. */
#define ADM_AMAX_SLOTS 64
#define ADM_AMAX_STRIDE 32
#define ADM_AMAX_FLOATS (ADM_AMAX_SLOTS * ADM_AMAX_STRIDE)

extern "C" {
#endif

int adm_version(void);

/* ---------------- convolution / Linear: fp32-MFMA implicit GEMM ------------------------------ */

/* y[B,H,W,N] = conv_{ks x ks, pad ks/2}(x) + bias (+ res).   Replaces Conv2d.forward
 * (unet/uncond_unet.py:91-113: F.conv2d, the up branch's conv_transpose2d-with-ones == nearest x2
 * when up=1, bias add) and Linear.forward (:62-66, as ks=1,H=W=1).  x is [B,Hin,Win,ldx] with
 * (Hin,Win) = (H,W) or (H/2,W/2) when up.  wp = adm_pack_weight output [wrows][ks*ks*Cin].
 * res (optional) [B*H*W][ldr] is added in the epilogue (the block's residual / skip add, :201, :209).
 * tile: -1 = heuristic, 0..3 force a tile shape (tests). */
int adm_conv_fwd(const float* x, const float* wp, const float* bias, const float* res, float* y,
                 int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                 int ks, int up, int tile, hipStream_t stream);

/* adm_conv_fwd with a stride and explicit top/left zero padding: tap (ky, kx) of output (oy, ox) reads input
 * (oy*stride + ky - pad_lo, ox*stride + kx - pad_lo); anything outside the Hin x Win image is zero (this is the
 * bottom/right padding).  The KL-f4 autoencoder's Downsample -- F.pad(x, (0,1,0,1)) + Conv2d(3x3, stride 2, padding 0),
 * /root/reference/ddm/encoder_decoder.py:78-96 -- is stride 2, pad_lo 0, Hout = Hin/2.  Forward only (the first stage is
 * frozen: /root/reference/ddm/ddm_const_2.py:436-440). */
int adm_conv_fwd_strided(const float* x, const float* wp, const float* bias, const float* res, float* y,
                         int B, int Hin, int Win, int Hout, int Wout, int Cin, int ldx, int N, int wrows,
                         int ldy, int ldr, int ks, int stride, int pad_lo, hipStream_t stream);

/* 3x3 stride-1 conv (forward / data gradient) through a 1-D Winograd F(2,3) transform along x: 1.5x fewer MFMA flops than
 * adm_conv_fwd, same NHWC contract (up = 0, ks = 3), W even, Cin % 16 == 0.  wq = adm_pack_weight_wino's operand
 * [4][wrows][3][Cin] (G g applied at pack time: 1, (g0+g1+g2)/2, (g0-g1+g2)/2, 1).  fp32 throughout; not bit-identical to the
 * direct kernel (different summation), same tolerance.  F.conv2d of Conv2d.forward, uncond_unet.py:98-110. */
int adm_conv_fwd_wino(const float* x, const float* wq, const float* bias, const float* res, float* y, int B, int H, int W,
                      int Cin, int ldx, int N, int wrows, int ldy, int ldr, hipStream_t stream);
/* adm_conv_fwd_wino with the nearest-x2 up-sampling of Conv2d(up=True) fused into the loader: x is [B][H/2][W/2][ldx],
 * H x W is the OUTPUT grid (both even).  uncond_unet.py:98-104. */
int adm_conv_fwd_wino_up(const float* x, const float* wq, const float* bias, const float* res, float* y, int B, int H, int W,
                         int Cin, int ldx, int N, int wrows, int ldy, int ldr, hipStream_t stream);
/* OIHW [Co][Ci][3][3] -> wf [4][Co_pad][3][Ci_pad] (forward) and wb [4][Ci_pad][3][Co_pad] (data gradient: taps flipped,
 * channels transposed); either may be NULL. */
int adm_pack_weight_wino(const float* w, float* wf, float* wb, int Co, int Ci, int Co_pad, int Ci_pad, hipStream_t stream);

/* Deterministic split-K for the small-M layers (4x4 resolution, embedding Linears): adm_conv_splitk(M, N, K) is the
 * number of K slices the library would use (1 = none); adm_conv_fwd_ws is adm_conv_fwd with a workspace of at least
 * splitk*M*N floats: partial tiles are written there and summed (+ bias, + res) in a fixed order by a second launch. */
int adm_conv_splitk(int M, int N, int K);
int adm_conv_fwd_ws(const float* x, const float* wp, const float* bias, const float* res, float* y, float* ws,
                    long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr, int ks,
                    int up, hipStream_t stream);

/* dwp[Cout][ks*ks][Cin] = sum_pixels dy[p][co] * x[p+tap][ci]  (atomic fp32 accumulation over
 * `splits` pixel ranges; dwp is overwritten: the call zero-fills it when it splits).  The weight-gradient of the conv above
 * (autograd of F.conv2d in the reference).  Cout, Cin multiples of 32. */
int adm_conv_wgrad(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx,
                   int Cout, int lddy, int ks, int up, int splits, hipStream_t stream);

/* adm_conv_wgrad that also accumulates the conv's bias gradient, dbias[co] += sum_pixels dy[p][co] (fp32 atomics; the
 * caller zero-fills or passes the gradient buffer to accumulate into): the dy tiles already pass through registers, so
 * no separate reduction pass over dy is needed.  dbias == NULL is adm_conv_wgrad. */
int adm_conv_wgrad_bias(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin, int ldx,
                        int Cout, int lddy, int ks, int up, int splits, hipStream_t stream);

/* adm_conv_wgrad_bias for 3x3 stride-1 convs through the transposed Winograd form F(3,2) along x (1.5x fewer MFMA flops;
 * fp32; power-of-two H and W >= 2).  Same outputs: dwp[Cout][9][Cin] (zero-filled by the call when it splits) and the
 * optional dbias accumulation.  Autograd of F.conv2d's weight, uncond_unet.py:98-110. */
int adm_conv_wgrad_wino(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin, int ldx,
                        int Cout, int lddy, int splits, hipStream_t stream);

/* adm_conv_wgrad_wino for Conv2d(up=True): x is the HALF-resolution input [B][H/2][W/2][ldx], H x W is dy's grid. */
int adm_conv_wgrad_wino_up(const float* x, const float* dy, float* dwp, float* dbias, int B, int H, int W, int Cin, int ldx,
                           int Cout, int lddy, int splits, hipStream_t stream);

/* 2-D Winograd F(2x2, 3x3) forward / data gradient (conv_wino2d.hip): 2.25x fewer MFMA flops than adm_conv_fwd, 1.5x fewer
 * than adm_conv_fwd_wino; H and W even, Cin % 16 == 0.  wq = adm_pack_weight_wino2d operand: wf[16][Co_pad][Ci_pad] (forward)
 * or wb[16][Ci_pad][Co_pad] (data gradient, taps flipped), plane index ey * 4 + ex of U = G g G^T. */
int adm_pack_weight_wino2d(const float* w, float* wf, float* wb, int Co, int Ci, int Co_pad, int Ci_pad, hipStream_t stream);
int adm_conv_fwd_wino2d(const float* x, const float* wq, const float* bias, const float* res, float* y, float* ws, long ws_floats,
                        int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr, hipStream_t stream);
/* f32 products on the bf16 MFMA through the exact three-term bf16 split (conv_wino2d_x6.hip): same contract as
 * adm_conv_fwd_wino2d with wq6 = adm_split3_bf16(adm_pack_weight_wino2d operand).
 * Replaces F.conv2d of Conv2d.forward + its data gradient (/root/reference/unet/uncond_unet.py:98-110). */
int adm_conv_fwd_wino2d_x6(const float* x, const void* wq6, const float* bias, const float* res, float* y, float* ws,
                           long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                           hipStream_t stream);
/* ... on the nearest x2 up-sampling of x[B][H/2][W/2][ldx] (Conv2d(up=True), uncond_unet.py:105-108); H x W = the output grid */
int adm_conv_fwd_wino2d_x6_up(const float* x, const void* wq6, const float* bias, const float* res, float* y, float* ws,
                              long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                              hipStream_t stream);
int adm_wino2d_x6_splitk(int B, int H, int W, int Cin, int N);
/* The same convolution on THREE fp16 products per f32 product (conv_wino2d_x6.hip, X6Fmt<1>): operands are split into two fp16 terms
 * by round-to-nearest after a power-of-two scaling, s a = h0 + h1 with |s a - h0 - h1| <= 2^-24 |s a| (h1 normal), and
 * a b ~ (h0 h0' + h0 h1' + h1 h0') / (s s').  wqh = adm_split2_f16 of the adm_pack_weight_wino2d planes with scale `wscale` (a power
 * of two; *overflow is raised if a scaled weight leaves the fp16 range), layout [ey][cols/16][ex][term(2)][rows][16].  amax_x: a bound
 * vector (above) of |x| over the whole input (e.g. written by adm_gn_fwd_amax, which produced x); the kernel derives the activation scale
 * from it so that the Winograd input transform (sums of four values) stays inside the fp16 range.  Error against fp64: that of the
 * six-bf16-product form (tools/fp16x3_accuracy.py, tests/test_hip_ops.py).  up != 0: Conv2d(up=True) as adm_conv_fwd_wino2d_x6_up.
 * Replaces F.conv2d of uncond_unet.py:98-110 like the other forms. */
int adm_split2_f16(const float* src, void* dst, int rows, int cols, float scale, int* overflow, hipStream_t stream);
int adm_conv_fwd_wino2d_h3(const float* x, const void* wqh, const float* bias, const float* res, float* y, float* ws,
                           long ws_floats, int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr,
                           const float* amax_x, float wscale, int up, hipStream_t stream);
/* dst (48 * rows * cols bf16, layout [ey][cols/16][ex][term][rows][16]) <- exact split a = a0 + a1 + a2 of the sixteen Winograd
 * planes src[ey * 4 + ex][rows][cols] (f32) */
int adm_split3_bf16(const float* src, void* dst, int rows, int cols, hipStream_t stream);
/* 1x1 conv / pixel-wise linear map with the f32 products on the bf16 MFMA by exact three-term splitting (conv_gemm_x6.hip):
 * y[M][ldy] = x[M][ldx] (K channels) . w^T (+ bias) (+ res), w6 = adm_split3_rows of the packed operand [wrows >= N][K] (f32, row
 * stride ld) = [K/16][3][wrows][16] bf16.  K % 32 == 0.  Replaces F.conv2d (1x1) + its data gradient (uncond_unet.py:98-110). */
int adm_gemm_x6(const float* x, const void* w6, const float* bias, const float* res, float* y, long M, int K, int ldx, int N, int wrows,
                int ldy, int ldr, hipStream_t stream);
int adm_split3_rows(const float* src, void* dst, int rows, int cols, int ld, hipStream_t stream);
/* adm_gemm_x6 that also raises the bound vector amax_y (above) to max |y| */
int adm_gemm_x6_amax(const float* x, const void* w6, const float* bias, const float* res, float* y, long M, int K, int ldx, int N,
                     int wrows, int ldy, int ldr, float* amax_y, hipStream_t stream);
/* ... on the fp16 format (two fp16 terms per operand, three MFMAs per f32 product, as adm_conv_fwd_wino2d_h3): wh = adm_split2_rows_f16 of
 * the packed operand = [K/16][2][wrows][16] fp16 of wscale * w (*overflow raised if a scaled weight leaves the fp16 range), amax_x = bound
 * vector of |x|, amax_y (may be NULL) = bound vector raised to max |y|.  Same replacement (uncond_unet.py:98-110). */
int adm_gemm_x6_h3(const float* x, const void* wh, const float* bias, const float* res, float* y, long M, int K, int ldx, int N, int wrows,
                   int ldy, int ldr, const float* amax_x, float wscale, float* amax_y, hipStream_t stream);
int adm_split2_rows_f16(const float* src, void* dst, int rows, int cols, int ld, float scale, int* overflow, hipStream_t stream);
/* The 2-D Winograd weight gradient with the f32 products on the bf16 MFMA by exact three-term splitting (conv_wgrad_x6.hip): same
 * contract as adm_conv_wgrad_wino2d (dwp2[Cout][4 ey][3 kx][Cin] -> adm_unpack_wgrad_wino2d; dbias += column sums of dy; splits = 0:
 * chosen by the launcher; H, W powers of two >= 2).  _ws: deterministic mode, split z stores its partial planes at ws[z][Cout][12][Cin]
 * and its bias partial at bws[z][Cout]; splits must be adm_conv_wgrad_x6_plan(...).
 * Replaces the autograd weight gradient of Conv2d.forward (/root/reference/unet/uncond_unet.py:98-110). */
int adm_conv_wgrad_x6(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx, int Cout,
                      int lddy, int splits, hipStream_t stream);
/* _up: weight gradient of Conv2d(up=True): x is the conv's half-resolution input [B][H/2][W/2][ldx], H x W is dy's grid */
int adm_conv_wgrad_x6_up(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx, int Cout,
                         int lddy, int splits, hipStream_t stream);
int adm_conv_wgrad_x6_ws(const float* x, const float* dy, float* ws, float* bws, int B, int H, int W, int Cin, int ldx, int Cout,
                         int lddy, int splits, int up, hipStream_t stream);
int adm_conv_wgrad_x6_plan(int B, int H, int W, int Cin, int Cout);
/* Weight gradient of a 1x1 conv on the same kernel (conv_wgrad_x6.hip, MODE 1): dwp[Cout][Cin] (+)= sum over P pixels of
 * dy[p][co] x[p][ci] (-> adm_unpack_wgrad with ks = 1), dbias += column sums of dy; _ws / _plan as for adm_conv_wgrad_x6. */
int adm_gemm_wgrad_x6(const float* x, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout, int lddy,
                      int splits, hipStream_t stream);
int adm_gemm_wgrad_x6_ws(const float* x, const float* dy, float* ws, float* bws, long P, int Cin, int ldx, int Cout, int lddy,
                         int splits, hipStream_t stream);
int adm_gemm_wgrad_x6_plan(long P, int Cin, int Cout);
/* The same two weight gradients with x stored as bf16 (the opt-in bf16 mode's GroupNorm outputs, adm_gn_fwd_bf16out): x16 =
 * [B][H][W][ldx] bf16 ([B][H/2][W/2][ldx] when up); products exact in x, three-term exact in dy, f32 accumulate. */
int adm_conv_wgrad_x6_bf16a(const void* x16, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx,
                            int Cout, int lddy, int splits, int up, hipStream_t stream);
int adm_gemm_wgrad_x6_bf16a(const void* x16, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout, int lddy,
                            int splits, hipStream_t stream);
/* ... and on the fp16 format of adm_conv_fwd_wino2d_h3 (two fp16 terms per operand, three products): amax_x / amax_dy = bound vectors
 * (above) of |x| / |dy|, written by the kernels that produced the tensors; det != 0 = the _ws contract (dwp = ws, dbias = bws,
 * splits from the _plan call).  Replaces the same autograd weight gradient (/root/reference/unet/uncond_unet.py:98-110). */
int adm_conv_wgrad_x6_h3(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx, int Cout,
                         int lddy, int splits, int up, int det, const float* amax_x, const float* amax_dy, hipStream_t stream);
int adm_gemm_wgrad_x6_h3(const float* x, const float* dy, float* dwp, float* dbias, long P, int Cin, int ldx, int Cout, int lddy,
                         int splits, int det, const float* amax_x, const float* amax_dy, hipStream_t stream);
/* kernel variant of adm_conv_fwd_wino2d: -1 (default) chosen per launch, 1 wave-specialised (producer / consumer waves), 0 symmetric;
 * returns the old value */
int adm_wino2d_variant(int ws);
/* form of adm_conv_fwd_wino2d_h3's kernel: -1 (default) chosen per launch, 1 the 128-cout workgroups (eight consumer waves) / 3 the 96-cout
 * workgroups (six) wherever the launch qualifies (no split-K), 0 always 64-cout workgroups; returns the old value */
int adm_wino2d_h3_wide(int v);
/* form of adm_conv_wgrad_x6_h3 / adm_gemm_wgrad_x6_h3's kernel: -1 (default) / 2: 128 couts per workgroup (sixteen waves) where the launch
 * qualifies (more than 64 couts, not the deterministic workspace mode), 1: always 64; returns the old value */
int adm_wgrad_h3_blocks(int v);
/* split count (>= 1) adm_conv_fwd_wino2d uses when given a workspace of that many B*H*W*N-float slices (small maps) */
int adm_wino2d_splitk(int B, int H, int W, int Cin, int N);

/* 2-D Winograd F(3x3, 2x2) weight gradient (conv_wgrad_wino.hip, MODE 2): 1.5x fewer MFMA flops than adm_conv_wgrad_wino.  H, W
 * powers of two, H >= 2.  dwp2[Cout][4 ey][3 kx][Cin] holds the x-folded transform planes; adm_unpack_wgrad_wino2d applies the
 * y half of G^T while scattering to OIHW (splits = 1, bws = dbias = NULL for the atomic path; the workspace form of
 * adm_conv_wgrad_ws(wino = 2) otherwise). */
int adm_conv_wgrad_wino2d(const float* x, const float* dy, float* dwp2, float* dbias, int B, int H, int W, int Cin, int ldx,
                          int Cout, int lddy, int splits, hipStream_t stream);
int adm_unpack_wgrad_wino2d(const float* wx, int splits, float* dw, int Co, int Ci, int Co_pad, int Ci_pad, int accumulate,
                            const float* bws, float* dbias, hipStream_t stream);

/* Deterministic weight gradients (bitwise reproducible backward; SURVEY.md section 5.2 "run twice, bit-compare").  The
 * kernels above combine their pixel-range splits with fp32 atomics.  Here split z writes its partial tile with plain stores
 * to ws[z][Cout][ks*ks][Cin] (and its bias partial to bws[z][Cout], bws may be NULL) and adm_unpack_wgrad_splits sums the
 * splits in a fixed order.  adm_conv_wgrad_plan returns the split count the launcher picks (>= 1) for the direct
 * (wino = 0), Winograd F(3,2) (wino = 1: ks = 3, power-of-two H and W) or 2-D F(3x3,2x2) (wino = 2; ws planes are then
 * [Cout][12][Cin], see adm_conv_wgrad_wino2d) kernel; pass it as `splits`. */
int adm_conv_wgrad_plan(int B, int H, int W, int Cin, int Cout, int ks, int up, int wino);
int adm_conv_wgrad_ws(const float* x, const float* dy, float* ws, float* bws, int B, int H, int W, int Cin, int ldx, int Cout,
                      int lddy, int ks, int up, int splits, int wino, hipStream_t stream);

/* ---- reduced-precision option (BASELINE.json configs[2], "bf16"): same contracts as adm_conv_fwd / adm_conv_wgrad,
 * tensors stay fp32 in HBM, the contraction runs on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16 on their way
 * into LDS, fp32 accumulation).  wp16 = the adm_pack_weight layouts converted with adm_f32_to_bf16.  Cin % 64 == 0
 * for adm_conv_fwd_bf16.  Never selected implicitly: the host opts in (adm_amd.ops.set_compute_precision). */
int adm_conv_fwd_bf16(const float* x, const unsigned short* wp16, const float* bias, const float* res, float* y,
                      int B, int H, int W, int Cin, int ldx, int N, int wrows, int ldy, int ldr, int ks, int up,
                      int tile, hipStream_t stream);
int adm_conv_wgrad_bf16(const float* x, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx, int Cout,
                        int lddy, int ks, int up, int splits, hipStream_t stream);
/* bf16-STORAGE variants (BASELINE configs[2]): the activation operand is already bf16 in HBM (written by adm_gn_fwd_bf16out), so
 * the kernels read half the bytes and skip the conversion; results are bit-identical to the f32-activation entry points above. */
int adm_conv_fwd_bf16a(const void* x16, const unsigned short* wp16, const float* bias, const float* res, float* y, int B, int H, int W,
                       int Cin, int ldx, int N, int wrows, int ldy, int ldr, int ks, int up, int tile, hipStream_t stream);
int adm_conv_wgrad_bf16a(const void* x16, const float* dy, float* dwp, int B, int H, int W, int Cin, int ldx, int Cout, int lddy, int ks,
                         int up, int splits, hipStream_t stream);
/* adm_gn_fwd with y stored as bf16 (round to nearest even): y16[B][HW][C] */
int adm_gn_fwd_bf16out(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
                       long ss_bstride, void* y16, int B, int HW, int C, int G, float eps, int silu, float drop_p, uint64_t seed,
                       hipStream_t stream);
int adm_f32_to_bf16(const float* src, unsigned short* dst, long n, hipStream_t stream);

/* OIHW [Co][Ci][ks][ks] (the reference's parameter layout, uncond_unet.py:85) ->
 *   wp_fwd [Co_pad][ks*ks][Ci_pad]            (B operand of adm_conv_fwd)
 *   wp_bwd [Ci_pad][ks*ks flipped][Co_pad]    (B operand of adm_conv_fwd computing dL/dx)
 * zero padded.  qkv != 0: rows are permuted from the reference's (head, c, {q,k,v}) interleave
 * (uncond_unet.py:205) to (head, {q,k,v}, c) so q/k/v are contiguous 64-float runs.
 * Either output may be NULL. */
int adm_pack_weight(const float* w, float* wp_fwd, float* wp_bwd, int Co, int Ci, int ks, int Co_pad, int Ci_pad,
                    int qkv, hipStream_t stream);
/* adm_pack_weight (+ adm_pack_weight_wino) for every layer of a model in ONE launch (used after each optimiser step).
 * table = device array of n_entries rows of 12 int64: {src, dst_fwd, dst_bwd, Co, Ci, ks*ks, Co_pad, Ci_pad, qkv, tile_begin,
 * dst_wino_fwd, dst_wino_bwd, dst_wino2d_fwd, dst_wino2d_bwd}; tile_begin = the exclusive prefix sum of (Co_pad/32)*(Ci_pad/32) in row order; total_tiles =
 * the sum; the Winograd destinations may be 0.  dst_fwd / dst_bwd are required. */
int adm_pack_weight_table(const long* table, int n_entries, long total_tiles, hipStream_t stream);
/* inverse of the fwd packing for gradients: dw OIHW = (accumulate ? dw : 0) + dwp */
int adm_unpack_wgrad(const float* dwp, float* dw, int Co, int Ci, int ks, int Co_pad, int Ci_pad, int qkv,
                     int accumulate, hipStream_t stream);
/* adm_unpack_wgrad over the `splits` partial gradients of adm_conv_wgrad_ws, summed in split order; bias partials
 * bws[splits][Co_pad] are summed into dbias[0..Co_pad) (+=, packed channel order); bws and dbias may both be NULL. */
int adm_unpack_wgrad_splits(const float* ws, int splits, float* dw, int Co, int Ci, int ks, int Co_pad, int Ci_pad, int qkv,
                            int accumulate, const float* bws, float* dbias, hipStream_t stream);
/* adm_unpack_wgrad / adm_unpack_wgrad_wino2d (splits = 1) for the weight gradients of ALL layers in ONE launch, issued once at the
 * end of the backward pass.  table = device array of `rows` rows of 12 int64: {src, dst, Co, Ci, taps, Ci_pad, qkv, accumulate,
 * zero_src, block_begin, 0, 0}; taps = ks*ks for the plain packed layout [Co_pad][taps][Ci_pad], 0 for the 2-D Winograd planes
 * [Co_pad][4 ey][3 kx][Ci_pad]; a row owns ceil(items / 2048) blocks (items = Co*Ci*taps, or Co*Ci*3), block_begin = exclusive
 * prefix sum, total_blocks = the sum.  zero_src != 0 clears every workspace element after reading it: with workspaces that are zero
 * at rest the weight-gradient entry points take splits = -1 ("chosen by the launcher, workspace already zero") and enqueue no memset. */
int adm_unpack_wgrad_table(const long* table, int rows, long total_blocks, hipStream_t stream);
/* out[i] = in[perm(i)] over n_pad entries (zero beyond n); qkv permutation of bias vectors.
 * inverse=1 maps packed order back to the reference order. */
int adm_permute_vec(const float* in, float* out, int n, int n_pad, int qkv, int inverse, hipStream_t stream);
/* out[n] (+)= sum_m a[m][n]: bias gradients. */
int adm_colsum(const float* a, float* out, int M, int N, int ld, int accumulate, hipStream_t stream);

/* ---------------- GroupNorm (+ adaptive scale/shift) + SiLU (+ dropout) ----------------------- */

/* stats[b][g] = {mean, rstd} of x[b, :, g*cpg:(g+1)*cpg]; ws: scratch of B*splits*G*2 doubles
 * (splits = adm_gn_splits(HW, C)).  F.group_norm's moments (uncond_unet.py:128). */
/* 1 (default): one-launch register-resident GroupNorm where a plan exists; 0: always the multi-pass kernels.  Returns the old value. */
int adm_gn_fused(int on);
int adm_gn_splits(int HW, int C);
int adm_gn_stats(const float* x, float* stats, double* ws, int B, int HW, int C, int G, float eps, hipStream_t stream);
/* y = act( (xhat*gamma + beta) * (1 + scale[b,c]) + shift[b,c] ) * dropmask
 * xhat = (x - mean) * rstd.  ss = [.., 2C] rows of (scale | shift) or NULL (uncond_unet.py:191-196);
 * ss_bstride = floats between batch rows (0 = broadcast one row: sampling runs the embedding at batch 1).
 * silu: apply SiLU.  drop_p > 0: inverted dropout with the stateless mask (seed, element index)
 * (uncond_unet.py:200). */
int adm_gn_apply(const float* x, const float* stats, const float* gamma, const float* beta, const float* ss,
                 long ss_bstride, float* y, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed,
                 hipStream_t stream);
/* adm_gn_stats + adm_gn_apply in one call (same arguments); feature maps of HW <= 256 pixels run as ONE launch that keeps
 * its slab in registers between the reduction and the apply pass (x is read once).  stats is written either way. */
int adm_gn_fwd(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
               long ss_bstride, float* y, int B, int HW, int C, int G, float eps, int silu, float drop_p, uint64_t seed,
               hipStream_t stream);
/* adm_gn_fwd that also raises the bound vector amax (above; zeroed by the caller) to max |y| with one atomicMax per wave: the scale basis
 * of the fp16-format convolution that consumes y (adm_conv_fwd_wino2d_h3). */
int adm_gn_fwd_amax(const float* x, float* stats, double* ws, const float* gamma, const float* beta, const float* ss,
                    long ss_bstride, float* y, float* amax, int B, int HW, int C, int G, float eps, int silu, float drop_p,
                    uint64_t seed, hipStream_t stream);
/* Backward of adm_gn_apply.  Pass 1 reduces per (b,c): r1 = sum du, r2 = sum du*xhat into red[B][C][2]
 * (du = dy * mask * act'(u)); pass 2 writes dx and, when the pointers are non-NULL, dss[B][2C]
 * (d scale | d shift; row stride = ss_bstride when that is non-zero: ss and dss may be column slices of one wide buffer),
 * dgamma[C], dbeta[C] (accumulated: caller zero-fills dgamma/dbeta). */
int adm_gn_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
               const float* ss, long ss_bstride, float* dx, float* dss, float* dgamma, float* dbeta, float* red,
               int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed, hipStream_t stream);

/* The batch reduction of dgamma / dbeta for EVERY GroupNorm layer of a backward pass in one launch: adm_gn_bwd / adm_gn_bwd_add
 * called with dgamma = dbeta = NULL leave the per-image sums tot[B][C][2] at red + B*S*C*2 floats (S = adm_gn_splits(HW, C)); the
 * caller keeps `red` (and ss) alive and queues a row.  table = device array of `rows` rows of 8 int64: {tot, ss (0: none),
 * ss_bstride, dgamma, dbeta, B, C, block_begin}; a row owns ceil(C / 32) blocks, block_begin = exclusive prefix sum, total_blocks =
 * the sum.  dgamma[c] += sum_b (1 + scale[b][c]) tot[b][c][1], dbeta[c] += ... tot[b][c][0]: the same order as the per-layer pass. */
int adm_gn_bwd_param_table(const long* table, int rows, long total_blocks, hipStream_t stream);

/* adm_gn_bwd with dx = (GroupNorm input gradient) + addend[B][HW][C] (addend may be NULL): the block input also feeds the
 * residual branch (uncond_unet.py:189, 201), and adding that branch's gradient here saves autograd's separate
 * read-read-write pass over the activation. */
int adm_gn_bwd_add(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                   const float* ss, long ss_bstride, const float* addend, float* dx, float* dss, float* dgamma, float* dbeta,
                   float* red, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed, hipStream_t stream);
/* ... that also raises the bound vector amax (zeroed by the caller) to max |dx|: the data-gradient convolution that consumes dx
 * can then run on the fp16 format (adm_conv_fwd_wino2d_h3 with the data-gradient weight image). */
int adm_gn_bwd_add_amax(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                        const float* ss, long ss_bstride, const float* addend, float* dx, float* dss, float* dgamma, float* dbeta,
                        float* red, float* amax, int B, int HW, int C, int G, int silu, float drop_p, uint64_t seed,
                        hipStream_t stream);

/* ---------------- KL autoencoder (first stage) helpers ---------------------------------------- */

/* In-place s[r][0:cols] = softmax(scale * s[r][0:cols]) for `rows` rows of stride ld: the single-head
 * (d = C = 512) attention of the autoencoder's mid block, whose QK^T and PV products run on adm_conv_fwd
 * (/root/reference/ddm/encoder_decoder.py:190-213).  cols % 4 == 0, cols <= 8192. */
int adm_softmax_rows(float* s, long rows, int cols, long ld, float scale, hipStream_t stream);
/* z[m][c] = zscale * (mean + exp(0.5*clamp(logvar,-30,20)) * eps[m*C+c]); moments rows = (mean[0:C] | logvar[C:2C]);
 * eps == NULL gives the mode.  DiagonalGaussianDistribution.sample (/root/reference/ddm/encoder_decoder.py:855-867). */
int adm_posterior_sample(const float* moments, int ldm, const float* eps, float* z, int ldz, long M, int C,
                         float zscale, hipStream_t stream);

/* ---------------- self-attention core -------------------------------------------------------- */

/* qkv [B][L][heads*192] packed (head, {q,k,v}, 64); out [B][L][heads*64].
 * out = softmax_k(q.k / 8) v per (b, head): uncond_unet.py:205-208.  L <= 32 or a multiple of 32; sequences beyond
 * 256 (L = 1024 at the 32x32 level of the 64x64-latent configs) run in 256-row chunks with an online softmax. */
int adm_attn_fwd(const float* qkv, float* out, float* lse, int B, int L, int heads, hipStream_t stream);
/* lse [B*heads][L] = log-sum-exp per query saved by the forward (may be NULL there when no backward
 * follows); delta [B*heads][L] scratch.  dqkv has the qkv layout.  Autograd of :205-208. */
int adm_attn_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta,
                 int B, int L, int heads, hipStream_t stream);
/* adm_attn_fwd with both matrix products on the fp16 split format (attention_h3.hip: three fp16 MFMAs per f32 product, as
 * adm_conv_fwd_wino2d_h3): amax = bound vector of |qkv|; L in {32, 64, 128, 256} or a multiple of 256 (ADM_EINVAL otherwise).  Same replacement (:205-208). */
int adm_attn_fwd_h3(const float* qkv, float* out, float* lse, const float* amax, int B, int L, int heads, hipStream_t stream);
/* adm_attn_bwd on the same format: amax_qkv / amax_dout = bound vectors of |qkv| and |dout|, amax_dqkv (may be NULL) = bound vector
 * raised to max |dqkv|; L as above.  Autograd of :205-208. */
int adm_attn_bwd_h3(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta,
                    const float* amax_qkv, const float* amax_dout, float* amax_dqkv, int B, int L, int heads, hipStream_t stream);
/* ... that also raises the bound vector amax to max |dqkv| (the qkv conv's gradients then run on the fp16 format) */
int adm_attn_bwd_amax(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta, float* amax,
                      int B, int L, int heads, hipStream_t stream);

/* ---------------- resampling / layout / elementwise ------------------------------------------ */

/* mode 0: y[B,H/2,W/2,C] = scale * sum_{2x2} x   (down: scale .25, uncond_unet.py:107-108;
 *         also the backward of up with scale 1)
 * mode 1: y[B,2H,2W,C]   = scale * x[y/2,x/2]    (up: uncond_unet.py:105-106; backward of down with .25)
 * acc != 0: y += ... */
int adm_resample2x(const float* x, float* y, int B, int H, int W, int C, int mode, float scale, int acc,
                   hipStream_t stream);
/* y[B,H,W,Cpad] = mul[b] * x_nchw (channels >= C zero); x is fp32 or fp64 (x_is_f64).
 * EDMPrecond's `c_in * x` + .to(float32) (uncond_unet.py:616, 628) fused with NCHW->NHWC. */
int adm_nchw_to_nhwc(const void* x, int x_is_f64, const float* mul, long mul_bstride, float* y, int B, int C, int HW,
                     int Cpad, hipStream_t stream);
/* ... that also raises the bound vector amax to max |y| (the stem conv then runs on the fp16 format) */
int adm_nchw_to_nhwc_amax(const void* x, int x_is_f64, const float* mul, long mul_bstride, float* y, float* amax, int B, int C, int HW,
                          int Cpad, hipStream_t stream);
/* out_nchw[b,c,p] = a[b] * x_nchw[b,c,p] + s[b] * f_nhwc[b,p,c]   (D = c_skip x + c_out F, :631-632).
 * x == NULL (then a may be NULL): out = s[b] * f -- the adjoint of adm_nchw_to_nhwc, i.e. dL/dx through `c_in * x` (the tensors
 * EDMPrecond.forward returns take part in autograd w.r.t. x, uncond_unet.py:614-635). */
int adm_precond_out(const void* x, int x_is_f64, const float* f, int ldf, const float* a, const float* s,
                    long coef_bstride, float* out, int B, int C, int HW, hipStream_t stream);
/* backward of the two above w.r.t. f:  df_nhwc[b,p,c<C] = s[b] * dout_nchw[b,c,p], zero for c >= C */
int adm_precond_out_bwd(const float* dout, const float* s, long coef_bstride, float* df, int ldf, int B, int C,
                        int HW, hipStream_t stream);
/* ... that also raises the bound vector amax to max |df| (the output conv's gradients then run on the fp16 format) */
int adm_precond_out_bwd_amax(const float* dout, const float* s, long coef_bstride, float* df, int ldf, float* amax, int B, int C,
                             int HW, hipStream_t stream);
/* emb[b][0:C/2] = cos(t[b] f_i), emb[b][C/2:] = sin(t[b] f_i), f_i = (1/10000)^(i/(C/2))
 * (PositionalEmbedding, uncond_unet.py:224-230) */
int adm_pos_embedding(const float* t, float* emb, int B, int C, hipStream_t stream);
/* y = silu(x) ; dx = dy * silu'(x) */
int adm_silu_fwd(const float* x, float* y, long n, hipStream_t stream);
int adm_silu_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t stream);
/* y = a + b (n floats); y may alias a */
int adm_add(const float* a, const float* b, float* y, long n, hipStream_t stream);
/* y = a + b (+ c when non-NULL), n % 4 == 0, 16-byte aligned: the sum of the gradients autograd would otherwise add pairwise for a
 * tensor with several consumers -- the encoder outputs feed the next block and both decoders' concatenations
 * (uncond_unet.py:548-571). */
int adm_add3(const float* a, const float* b, const float* c, float* y, float* amax, long n, hipStream_t stream);     /* amax (may be NULL): a bound vector, raised to max |y| */
/* dst[m][dst_off + c] (+)= scale * src[m][src_off + c], c < C: channel concat / slice copies
 * (torch.cat, :571, :578; `scale` carries uncond_unet_sd_3's skip-tuning ratio) */
int adm_copy_channels(const float* src, int lds, int src_off, float* dst, int ldd, int dst_off, long M, int C,
                      float scale, int acc, hipStream_t stream);
/* torch.cat((a, scale_b * b), channel axis) of two NHWC tensors in ONE launch (uncond_unet.py:571; `scale_b` = uncond_unet_sd_3's
 * skip-tuning ratio), amax (may be NULL) = bound vector raised to max |y|; adm_split2 = its adjoint (da = dy[:, :Ca], db = scale_b dy[:, Ca:]) */
int adm_concat2(const float* a, int Ca, const float* b, int Cb, float* y, long M, float scale_b, float* amax, hipStream_t stream);
int adm_split2(const float* dy, float* da, int Ca, float* db, int Cb, long M, float scale_b, hipStream_t stream);
/* out[b,i] = a[b] * x[b,i] + s[b] * y[b,i]  (x may be NULL -> s*y only; x fp32 or fp64): the
 * single-decoder variants' D_y = (x - (sigma-1) D_x) / g(sigma)  (uncond_unet_sd.py:602) and its backward */
int adm_axpby_b(const void* x, int x_is_f64, const float* y, const float* a, const float* s, long coef_bstride,
                float* out, int B, long n, hipStream_t stream);

/* SpatialAtt (uncond_unet.py:27-37) after the 384->1 `map` conv:  att [B][HW][ldatt] (channel 0),
 * y[b,p,c] = softsign( sum_j softmax_j(q_p k_j) att_j ) * h[b,p,c] + xres[b,p,c]
 * q = qw*att+qb, k = kw*att+kb; qk = {qw,qb,kw,kb} on device.  HW <= 64. */
int adm_spatial_att_fwd(const float* att, int ldatt, const float* qk, const float* h, const float* xres, float* y,
                        int B, int HW, int C, hipStream_t stream);
/* dh, datt (channel 0 of [B][HW][ldatt], other channels zeroed), dqk[4] (accumulated).  dqk_part = [B][4] workspace:
 * per-image partials summed in image order (deterministic); NULL = fp32 atomics. */
int adm_spatial_att_bwd(const float* att, int ldatt, const float* qk, const float* h, const float* dy, float* dh,
                        float* datt, float* dqk, float* dqk_part, int B, int HW, int C, hipStream_t stream);

/* ---------------- analytic schedule (ddm/ddm_const.py, ddm/ddm_const_2.py) -------------------- */

/* schedule 0 = 'const' (sqrt(t) noise gain), 1 = 'const_2' (t).
 * x_t = x0 + C t + g(t) eps with C = -x0   (ddm_const.py:284-287 / ddm_const_2.py:173-176); NCHW, n per image */
int adm_q_sample(const float* x0, const float* noise, const float* t, float* xt, int B, long n, int schedule,
                 hipStream_t stream);
/* Weighted SSE loss and its gradients (ddm_const.py:335-358 with ddm/loss.py MSE_Loss 'sum'):
 * per_sample[b] = w1[b] sum (Cp+x0)^2 + w2[b] sum (Np-noise)^2 ; dCp = gscale*2 w1 (Cp + x0) ; dNp likewise.
 * w = [B][2] weights precomputed on device. */
int adm_ddm_loss(const float* c_pred, const float* n_pred, const float* x0, const float* noise, const float* w,
                 float* per_sample, float* d_c, float* d_n, float gscale, int B, long n, hipStream_t stream);
/* Latent-space loss (LatentDiffusion.p_losses, ddm_const_2.py:527-596, schedule const_2): the two weighted SSE terms
 * of adm_ddm_loss plus w3[b] * sum |x_rec - x0| with x_rec = xt - Cp t - t Np.  w = [B][3] = (w1, w2, w3).
 * per_sample[b] = SSE part, per_l1[b] = un-weighted L1 sum; dCp = gscale (2 w1 (Cp + x0) - w3 t sign(x_rec - x0)),
 * dNp = gscale (2 w2 (Np - noise) - w3 t sign(x_rec - x0)).  Both outputs are zero-filled by the call. */
int adm_ddm_loss_latent(const float* c_pred, const float* n_pred, const float* x0, const float* noise, const float* xt,
                        const float* t, const float* w, float* per_sample, float* per_l1, float* d_c, float* d_n,
                        float gscale, int B, long n, int schedule, int use_l1,
                        hipStream_t stream);
/* One deterministic sampler update in fp64 (ddm_const.py:450-455 / ddm_const_2.py:363-368):
 * x0 = x - C t - eps g(t); [clamp]; x_next = x0 + C t' + eps g(t');  if last: clamp, /scale, (x+1)/2 */
int adm_sampler_step(double* x, const float* c_pred, const float* n_pred, double t_cur, double t_next, int schedule,
                     int clip_x0, double scale_input, int last, long n, hipStream_t stream);

/* One stochastic sampler update in fp64 (ddm_const.py:296-303, 410-414 / ddm_const_2.py:185-197, 324-328):
 * x0 = x - C t - g(t) eps; [clamp]; C' = -x0; x <- mean(x, C', eps, t, s) + sigma(t, s) z, per-image t[B], s[B];
 * z = the N(0,1) draw (device fp64).  last: final clamp, /scale, (x+1)/2. */
int adm_sampler_step_stochastic(double* x, const float* c_pred, const float* n_pred, const double* z, const double* t,
                                const double* s, int schedule, int clip_x0, double scale_input, int last, int B, long n,
                                hipStream_t stream);

/* ---------------- augmentation of x_start (use_augment: True) ------------------------------- */

/* Execution half of AugmentPipe for the transforms DDM enables (/root/reference/ddm/augment.py:161-172, 236-276;
 * instantiated at ddm_const.py:179-180 / ddm_const_2.py:112-113): per-image x/y flips, reflect padding by the batch-wide
 * `margin` = {mx0, my0, mx1, my1} (DEVICE ints: no host read-back), sym6 x2 up-sampling, bilinear affine resampling with
 * theta[N][2][3] (normalised coordinates, align_corners = False, zeros outside), sym6 x2 down-sampling and crop.
 * images / out: NCHW fp32 [N][C][H][W]; flips: int32 [N][2]; ws: adm_aug_workspace_floats(N, C, H, W) floats. */
long adm_aug_workspace_floats(int N, int C, int H, int W);
int adm_augment_geometric(const float* images, const int* flips, const int* margin, const float* theta, float* ws,
                          float* out, int N, int C, int H, int W, hipStream_t stream);

/* ---------------- optimiser (train_uncond_dpm.py:292-310, ddm/ema.py:158-188) ---------------- */

/* sumsq[0] += sum g^2.  partials = workspace of adm_sumsq_blocks(n) doubles: per-workgroup partial sums combined in a
 * fixed order by a second launch (the norm, hence the clip factor and the update, are bitwise reproducible); NULL = one
 * launch with fp64 atomics. */
int adm_sumsq_blocks(long n);
int adm_sumsq(const float* g, double* sumsq, double* partials, long n, hipStream_t stream);
/* AdamW step on flat buffers with the clip factor computed on device from sumsq[0]:
 * g *= min(1, max_norm/(sqrt(sumsq)+1e-6)); decoupled weight decay; optional EMA lerp
 * (ema += (1-decay)(p-ema)) fused into the same pass when ema != NULL. */
int adm_adamw_step(float* p, const float* g, float* m, float* v, float* ema, const double* sumsq, long n, float lr,
                   float beta1, float beta2, float eps, float wd, float max_norm, int step, float ema_decay,
                   float grad_scale, hipStream_t stream);


/* ================================================================================================
 * Conditional super-resolution denoiser (BASELINE configs[4]):
 * the operators /root/reference/unet/cond_unet_sd.py needs beyond the unconditional UNet's.  NHWC fp32.
 * ================================================================================================ */

/* adm_conv_wgrad for a strided conv with explicit top/left padding (Downsample = Conv2d(C, C', 4, 2, 1), cond_unet_sd.py:341-342;
 * also the 7x7 stem with stride 1, pad_lo 3): tap (ky, kx) of output (oy, ox) reads x(oy*stride + ky - pad_lo, ...).  The forward
 * is adm_conv_fwd_strided (ks up to 7). */
int adm_conv_wgrad_strided(const float* x, const float* dy, float* dwp, float* dbias, int B, int Hin, int Win, int Hout, int Wout,
                           int Cin, int ldx, int Cout, int lddy, int ks, int stride, int pad_lo, hipStream_t stream);
/* B operand of the GEMM form of the transposed conv (data gradient of a strided conv): out[(tap*Ci_pad + ci)][co] = w[co][ci][tap];
 * col[m][(tap, ci)] = adm_conv_fwd(dy as a 1x1 conv with this operand), then adm_col2im gathers dx. */
int adm_pack_weight_tconv(const float* w, float* out, int Co, int Ci, int ks, int Co_pad, int Ci_pad, hipStream_t stream);
int adm_col2im(const float* col, float* dx, int B, int Hin, int Win, int Ho, int Wo, int C, int ks, int stride, int pad_lo,
               hipStream_t stream);

/* WeightStandardizedConv2d (cond_unet_sd.py:344-357): wn[o][:] = (w[o][:] - mean_o) * rsqrt(var_o + eps) over the K = Cin*kh*kw
 * values of output channel o; stats[o] = (mean, rstd).  Backward: dw (+)= rstd (dwn - mean(dwn) - wn mean(dwn wn)). */
int adm_ws_fwd(const float* w, float* wn, float* stats, int O, int K, float eps, hipStream_t stream);
int adm_ws_bwd(const float* w, const float* stats, const float* dwn, float* dw, int O, int K, int accumulate, hipStream_t stream);

/* LayerNorm over the channels of every pixel with gain g and no bias (cond_unet_sd.py:359-368): x, y [M][C]. */
int adm_lnc_blocks(long M);
int adm_lnc_fwd(const float* x, const float* g, float* y, long M, int C, float eps, hipStream_t stream);
/* part = workspace of adm_lnc_blocks(M) * C doubles */
int adm_lnc_bwd(const float* x, const float* dy, const float* g, float* dx, float* dg, double* part, long M, int C, float eps,
                int accumulate, hipStream_t stream);

/* nn.BatchNorm2d of RelationNet.input_conv{1,2} (cond_unet_sd.py:247-254) on [M][C] rows.  training: batch statistics, running
 * statistics updated in place (momentum, unbiased variance), mr[c] = (mean, rstd) saved for the backward; else mr from the running
 * statistics.  part = adm_bn_blocks(M) * 2 C doubles; sums = 2 C floats. */
int adm_bn_blocks(long M);
int adm_bn_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float* mr, float* y,
               double* part, long M, int C, float eps, float momentum, int training, hipStream_t stream);
int adm_bn_bwd(const float* x, const float* dy, const float* mr, const float* gamma, float* dx, float* dgamma, float* dbeta,
               double* part, float* sums, long M, int C, int training, int accumulate, hipStream_t stream);

/* F.interpolate(mode='bilinear', align_corners=...) (cond_unet_sd.py:196, 235, 824): x [B][Hi][Wi][C] -> channels
 * [coff, coff + C) of y [B][Ho][Wo][ldy]; the backward is the exact adjoint in gather form (no atomics). */
int adm_bilinear_fwd(const float* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int ldy, int coff, int align_corners,
                     hipStream_t stream);
int adm_bilinear_bwd(const float* dy, float* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int lddy, int coff,
                     int align_corners, hipStream_t stream);

/* act: 0 identity, 1 ReLU, 2 GELU (erf form), followed by dropout(p) with the stateless hash of the GroupNorm kernels. */
int adm_act_fwd(const float* x, float* y, long n, int act, float drop_p, uint64_t seed, hipStream_t stream);
int adm_act_bwd(const float* x, const float* dy, float* dx, long n, int act, float drop_p, uint64_t seed, hipStream_t stream);
/* GaussianFourierProjection (cond_unet_sd.py:396-405): out[b] = [sin(2 pi x_b W), cos(2 pi x_b W)], W [D] */
int adm_fourier_features(const float* x, const float* W, float* out, int B, int D, hipStream_t stream);

/* adm_spatial_att_fwd / _bwd for maps of up to 2560 pixels (the 16x16 bottleneck of the SR denoiser): nothing of the HW x HW
 * softmax is stored; gate [B][HW][2] carries (pooled value, log-sum-exp) per row to the backward. */
int adm_spatial_att_big_fwd(const float* att, int ldatt, const float* qk, const float* h, const float* xres, float* y, float* gate,
                            int B, int HW, int C, hipStream_t stream);
int adm_spatial_att_big_bwd(const float* att, int ldatt, const float* qk, const float* h, const float* dy, const float* gate,
                            float* dh, float* datt, float* dqk, float* dqk_part, int B, int HW, int C, hipStream_t stream);

/* softmax(scale q k^T) v per (image, head) with separate query / key lengths; head h = columns [h*D, (h+1)*D) of rows with
 * strides ldq / ldk / ldv / ldo; D in {4, 8, 16, 32, 64}.  RelationNet's windowed cross-attention (cond_unet_sd.py:221-231,
 * scale 1) and the bottleneck Attention (:532-554, scale 32^-0.5).  lse [B*H][Lq]; delta = workspace [B*H][Lq]. */
int adm_mha_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int Lq, int Lk, int H, int D, int ldq,
                int ldk, int ldv, int ldo, float scale, hipStream_t stream);
int adm_mha_bwd(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse, float* dq,
                float* dk, float* dv, float* delta, int B, int Lq, int Lk, int H, int D, int ldq, int ldk, int ldv, int ldo, int lddq,
                int lddk, int lddv, float scale, hipStream_t stream);

/* LinearAttention core (cond_unet_sd.py:516-529): qkv [B][N][384] = (q | k | v) of 4 heads x 32 -> out [B][N][128];
 * ctx [B][4][32][32] and kst [B][128][2] are saved for the backward; ws = adm_linattn_ws_floats(B, N) floats;
 * dctx [B][4][32][32] and S [B][128] are workspaces of the backward. */
long adm_linattn_ws_floats(int B, int N);
int adm_linattn_fwd(const float* qkv, float* out, float* ctx, float* kst, float* ws, int B, int N, hipStream_t stream);
int adm_linattn_bwd(const float* qkv, const float* dout, const float* ctx, const float* kst, float* dqkv, float* dctx, float* S,
                    float* ws, int B, int N, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADM_HIP_H */
