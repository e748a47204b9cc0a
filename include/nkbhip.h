/* libnkbhip — C ABI of the MI355X (gfx950) hot path behind nkb_classification's
 * model.get_model() / engine.train_epoch() / losses.get_loss() / utils.get_optimizer().
 *
 * The reference (nkb-tech/nkb-classification) has no FFI of its own: every FLOP of its train step is
 * issued by torch/timm from these Python call sites, which the entry points below replace:
 *   nkb_classification/engine.py:48      preds = model(img)                 -> conv_gemm / bn_* / pool / im2row
 *   nkb_classification/engine.py:51      loss = criterion(preds, target)    -> loss_forward
 *   nkb_classification/engine.py:55-58   scaler.scale(loss).backward()      -> loss_backward / conv_gemm(mode=1) /
 *                                                                              conv_wgrad / bn_backward / pool bwd
 *   nkb_classification/engine.py:59      scaler.step(optimizer)             -> optim_step
 *   nkb_classification/engine.py:62      epoch_logger.log_iter (softmax/argmax, logging.py:268-281) -> loss_forward
 *   nkb_classification/engine.py:66-71   per-parameter grad.norm()          -> segment_sumsq
 *   nkb_classification/model.py:41-43,114-116  classifier head(s)           -> conv_gemm (R=S=1) / colsum
 *
 * Conventions: all pointers are BORROWED device pointers (never freed or retained); no allocation and
 * no synchronisation inside any call; kernels are enqueued on the caller's `stream`; every function
 * returns 0 on success or a non-zero hipError_t-style code, with text in nkb_last_error().
 * dtype codes: 0 = fp32 (parity mode, exact-fp32 MFMA), 1 = bf16 (fp32 accumulate).
 * Activations are NHWC (channel-contiguous rows with leading dimension ld*); filters are
 * [Cout][R][S][Cin] — the physical layout of a torch channels_last weight.
 */
#ifndef NKBHIP_H
#define NKBHIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* nkb_stream_t; /* hipStream_t */

#define NKB_DT_F32 0
#define NKB_DT_BF16 1

const char* nkb_last_error(void);
int nkb_version(void);
/* Launch counters of the specialised kernels since process start (or the last reset): which = 0 eight-phase GEMM (gemm8p), 1 eight-phase
 * weight gradient (wgrad8p / wgrad256), 2 shared-strip 3x3 weight gradient, 3 fp8 weight gradient, 4 Gram-form closing convolution,
 * 5 bn_apply fused with the Gram matrix, 6 row-balanced 3x3 core (convp), 7 pixel-resident 1x1 expansion (conv1p), 8 ring-buffered stem
 * (stemp: forward and weight gradient), 9 streamed g^T a (gramr), 10 row-streaming 256 x 128-tile 1x1 weight gradient (wgradr).  Tests use them to prove that a benchmark configuration took the path it is priced on. */
long long nkb_kernel_launches(int which, int reset);

/* Implicit-GEMM convolution / linear layer on MFMA.
 * mode 0 (forward):  y[n,p,q,co] = sum_{r,s,ci} x[n, p*stride+r-pad, q*stride+s-pad, ci] * w[co,r,s,ci]
 * mode 1 (dgrad):    y[n,h,w,co] = sum_{r,s,ci} x[n, (h+pad-r)/stride, (w+pad-s)/stride, ci] * w[co,r,s,ci]
 *                    (terms with a non-integral source coordinate are skipped; pass the [Cin][R][S][Cout] filter)
 * Epilogue: + bias[co], + add[m][co], relu (1: ReLU, 2: ReLU6), optional fp32 output, optional per-row-tile channel sums
 * stats[tile][0][co] = sum y, stats[tile][1][co] = sum y^2 (tile count: nkb_conv_gemm_stat_tiles).
 * Cin must be a multiple of 64 (bf16) / 32 (fp32); stride in {1,2}. */
int nkb_conv_gemm(int dtype, int mode, const void* x, const void* w, void* y, const void* add, const float* bias,
                  float* stats, int N, int H, int W, int Cin, int ldx, int P, int Q, int Cout, int ldy, int ldadd,
                  int R, int S, int stride, int pad, int relu, int out_f32, int add_h, int add_w,
                  const unsigned char* add_bits, nkb_stream_t stream);
/* add_h/add_w > 0: `add` is an [N][add_h][add_w] tensor living on the even (h, w) positions of the output grid (the
 * gradient of a stride-2 1x1 shortcut conv, folded in without materialising its zero-dilated form). */
/* add_bits (optional): ReLU bit mask of the `add` operand as written by nkb_bn_apply(relu_bits): add[m][c] only counts
 * where its bit is set (the gradient of a residual block's output, masked on the fly instead of in a separate pass). */
int nkb_conv_gemm_stat_tiles(int dtype, int M, int Cout);

/* Weight gradient: dw[co][r][s][ci] += sum_{n,p,q} dy[n,p,q,co] * x[n, p*stride+r-pad, q*stride+s-pad, ci] (fp32 atomics);
 * optional bias gradient dbias[co] += sum_{n,p,q} dy[n,p,q,co] from the same pass over dy. */
int nkb_conv_wgrad(int dtype, const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W, int Cin, int ldx,
                   int P, int Q, int Cout, int lddy, int R, int S, int stride, int pad, float* workspace,
                   long long workspace_floats, nkb_stream_t stream);
/* The same product with "=" instead of "+=" (dw / dbias overwritten; deterministic form only): scratch products such as the
 * Gram-form R = g^T a need no memset in front of them. */
int nkb_conv_wgrad_assign(int dtype, const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W, int Cin, int ldx,
                          int P, int Q, int Cout, int lddy, int R, int S, int stride, int pad, float* workspace,
                          long long workspace_floats, nkb_stream_t stream);
/* Workspace that makes nkb_conv_wgrad deterministic (bit-identical across runs): every (tile, pixel-split) workgroup stores
 * its fp32 partial tile into its own slab and a second launch adds the slabs to dw (and the bias partials to dbias) in
 * split order.  workspace == NULL keeps the single-launch form that accumulates with fp32 atomics. */
long long nkb_conv_wgrad_workspace_floats(int dtype, int N, int P, int Q, int Cin, int Cout, int R, int S, int stride, int pad,
                                          int has_bias);

/* BatchNorm2d (torch semantics: biased var to normalise, unbiased var into running_var, momentum blend). */
int nkb_bn_finalize(const float* partials, int tiles, int C, long long count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, int training, float* scale,
                    float* shift, float* save_mean, float* save_invstd, nkb_stream_t stream);
/* relu_bits (optional out): one byte per 16-byte chunk of y (8 bf16 / 4 fp32 channels), bit e = y[chunk*n + e] > 0 */
/* res_scale/res_shift (optional): `res` is itself a raw conv output (the projection shortcut of a block's first unit);
 * it enters as bn(res) = res*res_scale + res_shift without a separate pass that would write and re-read it */
int nkb_bn_apply(int dtype, const void* x, const void* res, void* y, const float* scale, const float* shift,
                 long long rows, int C, int relu, unsigned char* relu_bits, const float* res_scale,
                 const float* res_shift, nkb_stream_t stream);
/* ReLU mask: from relu_bits when given, else yact (> 0), else recomputed as x*fscale+fshift > 0 when fscale is given,
 * else none. */
int nkb_bn_backward(int dtype, const void* dy, const void* x, const void* yact, const unsigned char* relu_bits,
                    const float* fscale,
                    const float* fshift, const float* mean, const float* invstd, const float* gamma, long long rows,
                    int C, float* dgamma, float* dbeta, void* dx, void* dy_masked, float* workspace,
                    size_t workspace_floats, nkb_stream_t stream);
/* Data gradient of a convolution whose input was relu(bn(c)) (engine.py:55-58 backward through timm's conv -> bn -> act
 * chains): nkb_conv_gemm(mode 1) whose epilogue also applies the ReLU mask recomputed from (c, scale, shift), stores the
 * masked gradient and leaves per-row-tile sums of g' and g'*(c-mean) in stats (nkb_conv_gemm_stat_tiles(dtype, N*P*Q,
 * Cout) tiles, nkb_bn_stats_floats floats); nkb_bn_backward_from_stats then finishes that stage's BatchNorm backward
 * without a reduction pass over g and c.  sums: 2*C floats of scratch. */
int nkb_conv_dgrad_bn(int dtype, const void* dy, const void* w, void* g_masked, const void* c, const float* scale,
                      const float* shift, const float* mean, float* stats, const unsigned char* relu_bits,
                      const void* add, int ldadd, const unsigned char* add_bits, int add_h, int add_w, int N, int H,
                      int W, int Cin, int ldx, int P, int Q, int Cout, int ldy, int R, int S, int stride, int pad,
                      nkb_stream_t stream);
/* Row-balanced, DMA-pipelined core for 3x3 / stride-1 / pad-1 convolutions in bf16 (csrc/convp.hip) — timm Bottleneck / BasicBlock
 * conv2 built at /root/reference/nkb_classification/model.py:82, forward from engine.py:48, data gradient from engine.py:55-58.
 * One 512-thread workgroup per CU owns M / #workgroups consecutive output pixels; BatchNorm partial sums leave as ONE row per
 * workgroup: stats[tiles][2][Cout] with tiles = nkb_convp_tiles(...) (0: shape not eligible -> use nkb_conv_gemm / nkb_conv_dgrad_bn;
 * kind 0 forward, 1 data gradient).  nkb_convp_fwd: y = conv(x, w), stats = sums of y and y^2 (nkb_conv_gemm(mode 0) with stats).
 * nkb_convp_dgrad_bn: nkb_conv_dgrad_bn's recomputed-mask form (no residual operand); w is the [Cin][3][3][Cout] data-gradient filter,
 * Cin / ldx describe dY, Cout / ldy the produced gradient.  Both feed nkb_bn_finalize / nkb_bn_backward_from_stats with `tiles`. */
int nkb_convp_tiles(int dtype, int kind, int N, int H, int W, int Cin, int ldx, int Cout, int ldy, int R, int S, int stride, int pad);
/* Pixel-resident 1x1 / stride-1 convolution in bf16 for the EXPANSION stage of a bottleneck (csrc/conv1p.hip) — timm Bottleneck conv3,
 * Cin = 256 -> Cout >= 2 Cin (multiple of 256), forward from /root/reference/nkb_classification/engine.py:48.  One workgroup per CU
 * keeps its M / #CUs pixels x Cin in LDS and walks all output channels with the filter streamed from L2 into registers; y = conv(x, w),
 * stats[tiles][2][Cout] = per-workgroup sums of y and y^2 (nkb_conv_gemm(mode 0) with stats; feeds nkb_bn_finalize with `tiles`).
 * nkb_conv1p_tiles: 0 = shape not eligible -> use nkb_conv_gemm.  x is [M][ldx], w [Cout][Cin], y [M][ldy]. */
int nkb_conv1p_tiles(int dtype, long long M, int Cin, int ldx, int Cout, int ldy);
int nkb_conv1p_fwd(int dtype, const void* x, const void* w, void* y, float* stats, long long M, int Cin, int ldx, int Cout, int ldy,
                   nkb_stream_t stream);
/* The envelope of the row-resident kernels (convp, conv1p, stemp; NKB_CONVP=0 switches the family off).  on = 0 / 1 (default 1);
 * narrow bit 0: 3x3 also for Cout % 256 == 128 as 128-channel tiles (default 0: measured level with / slower than the 128 x 128 kernel
 * in the ResNet-50 step); bit 1: the 64 -> 64 channel 3x3 form with the filter resident in registers / LDS (default 1: forward), bit 2:
 * that form for the data gradient as well (default 0); bit 4 / 5 / 6: conv1p / stemp / gramr OFF (default on).  Tests and A/B timing. */
void nkb_convp_config(int on, int narrow);
/* Data-parallel runs: the family's BACKWARD kernels size their one-workgroup-per-CU grids for #CUs - cus, leaving room for the
 * collective's resident workgroups: nkb_convp_dgrad_bn (its partial-sum row count), nkb_gramr (its slab count) and the weight
 * gradients that run on wgradr (csrc/wgradr.hip: the split count behind nkb_conv_wgrad_workspace_floats).  0 = default.  The value
 * changes what nkb_convp_tiles(kind 1) / nkb_gramr_workspace_floats / nkb_conv_wgrad_workspace_floats answer, so buffers sized under
 * another value are stale: every launch of the three checks the capacity it is handed (`tiles`, workspace floats) against the
 * geometry it is about to launch and fails instead of writing past it; the Python binding also drops recorded launch plans. */
void nkb_rowres_reserve_cus(int cus);
int nkb_rowres_reserved_cus(void);
/* tiles: the partial-sum rows `stats` has room for = nkb_convp_tiles(dtype, kind, ...) at the time the buffer was sized; a launch
 * whose geometry differs (nkb_rowres_reserve_cus / nkb_convp_config changed in between) is refused. */
int nkb_convp_fwd(int dtype, const void* x, const void* w, void* y, float* stats, int N, int H, int W, int Cin, int ldx, int Cout,
                  int ldy, int tiles, nkb_stream_t stream);
int nkb_convp_dgrad_bn(int dtype, const void* dy, const void* w, void* g_masked, const void* c, const float* scale,
                       const float* shift, const float* mean, float* stats, int N, int H, int W, int Cin, int ldx, int Cout,
                       int ldy, int tiles, nkb_stream_t stream);
/* relu_bits != NULL: the stage closes a residual block — its mask comes from nkb_bn_apply's bit array (scale/shift
 * unused; c and mean may then be NULL as well: only the sums of g' are produced, the Gram form takes sum g'c from R = g'^T a) and the residual operand `add` (optionally under add_bits, or on the sub-grid add_h x add_w) is still added
 * before masking, so the block-output gradient is produced, masked and reduced in the one epilogue. */
/* One parity class (ph, pw) of the data gradient of a 3x3 / stride-2 / pad-1 convolution: output pixels (2h'+ph, 2w'+pw)
 * as a stride-1 gather over dY with (1|2) x (1|2) taps; w_class = [C][Rc][Sc][K] from nkb_wprep modes 2..5 (= 2 + 2*ph + pw).
 * The four classes together replace nkb_conv_gemm(mode 1, stride 2), which multiplies 3 taps out of 4 by zero.
 * add: optional residual ([N][Hout][Wout][ldadd], or the sub-grid form of nkb_conv_gemm with add_h/add_w);
 * c != NULL: fused BN-backward epilogue of nkb_conv_dgrad_bn, `stats` pointing at this class's own tile range. */
int nkb_conv_dgrad_s2class(int dtype, const void* dy, const void* w_class, void* y, const void* add, const void* c,
                           const float* scale, const float* shift, const float* mean, float* stats, int N, int Hdy,
                           int Wdy, int K, int ldx, int Hout, int Wout, int C, int ldy, int ldadd, int ph, int pw,
                           int add_h, int add_w, nkb_stream_t stream);
int nkb_bn_backward_from_stats(int dtype, const void* g, const void* x, float* stats, int tiles, const float* mean,
                               const float* invstd, const float* gamma, long long rows, int C, float* dgamma,
                               float* dbeta, void* dx, float* sums, nkb_stream_t stream);
size_t nkb_bn_stats_floats(int tiles, int C); /* size of the `partials` buffer nkb_bn_finalize expects */

/* ---- Gram form of a bottleneck's closing stage: conv3 (1x1) -> bn3 -> += shortcut -> ReLU ------------------------------
 * (timm resnet.py Bottleneck.forward, built by timm.create_model at /root/reference/nkb_classification/model.py:82 and driven
 * from engine.py:48 forward / engine.py:55-58 backward).  For c = a W^T every statistic BatchNorm needs of c follows from the
 * Cin x Cin Gram matrix of a (nkb_conv_wgrad(dy = a, x = a, dbias = column sums)): mean = W mu, var = diag(W Cov W^T).  So
 *   nkb_gram_bn_stats         scale / shift / mean / invstd / running statistics BEFORE the convolution runs (+ T = W Cov, mu kept),
 *   nkb_conv_affine_residual  y = relu(conv(a, W) * scale + shift + res') with the ReLU bit mask; c is never written or re-read,
 * and backward, with g = masked output gradient and R = g^T a (nkb_conv_wgrad into a zeroed scratch):
 *   nkb_gram_bn_backward      sum g c = rowdot(W, R) -> dgamma, dbeta, dW = k1 R + M k2 T - gamma r dbeta mu^T, and the filter
 *                             [k1 W ; W^T diag(k2) W] + bias k3 W of the concatenated data gradient,
 *   nkb_conv_dgrad_bn_cat     da = [g | a] . wcat^T + cbias with nkb_conv_dgrad_bn's fused BN-backward epilogue for the stage before,
 * so neither c nor dc = k1 g + k2 c + k3 exists in HBM (the 4x-wide tensor's bn_apply / bn_backward passes disappear).
 * tests/test_gram_bn_math.py pins the algebra against autograd in float64. */
int nkb_gram_bn_stats(int dtype, const void* w, const float* gram, const float* colsum, long long count, int Cin, int Cout,
                      const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      float* cov, float* mu, float* T, float* scale, float* shift, float* mean, float* invstd, nkb_stream_t stream);
/* w: [Cout][Cin] in the compute dtype (what the MFMA multiplies by); gram [Cin][Cin], colsum [Cin] fp32; cov: Cin*Cin floats of
 * scratch; mu [Cin], T [Cout][Cin] are outputs kept for nkb_gram_bn_backward.  Cin <= 512. */
int nkb_bn_apply_gram(int dtype, const void* c, void* y, const float* scale, const float* shift, long long rows, int C, float* gram,
                      float* work, size_t work_floats, nkb_stream_t stream);
size_t nkb_bn_apply_gram_workspace_floats(long long rows, int C);
/* nkb_bn_apply(relu) of the stage before the closing convolution fused with the Gram matrix of its output (bf16, C = 64 or 128):
 * y is written once, bit-identical to nkb_bn_apply, and gram[0 .. C*C) = y^T y, gram[C*C .. C*C + C) = column sums of y come out of
 * the same pass (per-workgroup fp32 slabs in `work`, added in a fixed order) — no second read of the activations. */
int nkb_conv_affine_residual(int dtype, const void* x, const void* w, void* y, const float* scale, const float* shift,
                             const void* res, int ldres, const float* res_scale, const float* res_shift,
                             unsigned char* relu_bits, int N, int H, int W, int Cin, int ldx, int P, int Q, int Cout, int ldy,
                             int R, int S, int stride, int pad, nkb_stream_t stream);
/* bf16, Cout > 64.  res_scale / res_shift (optional): res is the raw output of a projection shortcut and enters as
 * rnd(res * res_scale + res_shift), as in nkb_bn_apply.  relu_bits: layout of nkb_bn_apply's relu_bits. */
int nkb_gram_bn_backward(int dtype, const void* w, const float* R, const float* T, const float* mu, float* gstats, int tiles,
                         long long count, int Cin, int Cout, const float* gamma, const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, float* dw, void* wcat, void* q, float* cbias, float* work, nkb_stream_t stream);
size_t nkb_gram_bn_backward_workspace_floats(int Cin, int Cout);
/* gstats: per-row-tile sums of g as nkb_conv_dgrad_bn leaves them (first plane used; nkb_bn_stats_floats floats); dgamma, dbeta,
 * dw accumulate (+=); wcat: [Cin][Cout + Cin] in the compute dtype, cbias [Cin]; work: nkb_gram_bn_backward_workspace_floats floats
 * of scratch (tile sums, the partial sums of cbias = k3 W, and V = k2 .* W whose product V^T W — Q — runs on the MFMA
 * transposed-A GEMM of nkb_gemm_tn_batched).  64 | Cin <= 512. */
/* Two-launch form of that data gradient (exactly one of wcat / q is given to nkb_gram_bn_backward): nkb_gram_k1w builds
 * wk1[j][k] = k1_k W[k][j] from the forward scale alone, t = g . wk1^T is a plain nkb_conv_gemm that can start before anything of the
 * backward algebra exists (which then runs beside it on the weight-gradient stream, writing q = Q [Cin][Cin] and cbias), and
 * nkb_conv_dgrad_bn_add finishes da = t + a . Q + cbias with the fused BN-backward epilogue. */
int nkb_gram_k1w(int dtype, const void* w, const float* k1, int Cin, int Cout, void* out, nkb_stream_t stream);
int nkb_conv_dgrad_bn_add(int dtype, const void* a, int lda, int K, const void* q, const float* cbias, const void* t, int ldt,
                          void* g_masked, const void* c_prev, const float* scale, const float* shift, const float* mean, float* stats,
                          long long M, int Cout, int ldy, nkb_stream_t stream);
/* Gram form of a closing stage WITH a stride-1 projection shortcut (timm Bottleneck.downsample = conv1x1 + bn, layer1.0): both
 * BatchNorms' statistics come from Gram matrices (of the main branch's input a and of the block input x), their scales are folded
 * into one filter [scale3 .* W3 | scale_d .* Wd] (nkb_gram_fold2) and y = relu([a | x] . wf^T + shift3 + shift_d) is ONE launch with the
 * ReLU bits — neither raw conv output exists; backward: two R products (g^T a, g^T x), two nkb_gram_bn_backward calls, the main
 * branch's nkb_conv_dgrad_bn_cat and dx = [g | x] . [k1 Wd ; Qd]^T + cbias_d (nkb_conv_cat_bias). */
int nkb_gram_fold2(int dtype, const void* w1, const float* s1, int K1, const void* w2, const float* s2, int K2, int Cout, void* out,
                   const float* shift1, const float* shift2, float* shift_out, nkb_stream_t stream);
int nkb_conv_cat_relu_bits(int dtype, const void* a, int lda, int K1, const void* x, int ldx, int K2, const void* wf, const float* shift,
                           void* y, unsigned char* relu_bits, long long M, int Cout, int ldy, nkb_stream_t stream);
int nkb_conv_cat_bias(int dtype, const void* a, int lda, int K1, const void* x, int ldx, int K2, const void* w, const float* bias, void* y,
                      long long M, int Cout, int ldy, nkb_stream_t stream);
int nkb_conv_dgrad_bn_cat(int dtype, const void* g, int ldg, int K1, const void* a, int lda, int K2, const void* wcat,
                          const float* cbias, void* g_masked, const void* c_prev, const float* scale, const float* shift,
                          const float* mean, float* stats, long long M, int Cout, int ldy, nkb_stream_t stream);
size_t nkb_bn_backward_workspace_floats(long long rows, int C);

/* MaxPool2d(3, 2, 1) and global average pool, forward (backward=0) / backward (backward=1), NHWC. */
int nkb_maxpool3x3s2(int dtype, int backward, const void* in, void* out, unsigned char* idx, int N, int H, int W, int C,
                     nkb_stream_t stream);
int nkb_avgpool(int dtype, int backward, const void* in, void* out, int N, int HW, int C, nkb_stream_t stream);
/* ResNet stem tail in one pass (timm resnet.py forward_features: bn1 -> act1 -> maxpool, reached from
 * nkb_classification/model.py:84).  c is the raw conv1 output [N][H][W][C].
 * backward=0: y_or_g (out) = maxpool3x3s2(relu(c*scale+shift)) [N][P][Q][C], idx = winning window slot.
 * backward=1: y_or_g (in) is the pooled gradient; dc = gradient w.r.t. c through pool routing, ReLU mask and batch-norm
 *             backward; dgamma/dbeta accumulate; workspace >= nkb_bn_relu_maxpool_workspace_floats floats. */
int nkb_bn_relu_maxpool(int dtype, int backward, const void* c, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* gamma, void* y_or_g, unsigned char* idx,
                        void* dc, float* dgamma, float* dbeta, float* workspace, size_t workspace_floats, int N, int H,
                        int W, int C, nkb_stream_t stream);
/* The same with xsel [N][P][Q][C] (compute dtype, may be NULL = nkb_bn_relu_maxpool): forward also writes the raw conv output behind
 * every pooled winner, backward's reduction reads that instead of gathering two bytes per channel from c. */
int nkb_bn_relu_maxpool_sel(int dtype, int backward, const void* c, const float* scale, const float* shift,
                            const float* mean, const float* invstd, const float* gamma, void* y_or_g, unsigned char* idx,
                            void* dc, float* dgamma, float* dbeta, float* workspace, size_t workspace_floats, void* xsel,
                            int N, int H, int W, int C, nkb_stream_t stream);
size_t nkb_bn_relu_maxpool_workspace_floats(int N, int H, int W, int C);

/* NCHW fp32 image -> [N*P*Q][Kp] rows, k = (r*S+s)*Cin + c (stem conv / patch embedding as a GEMM). */
int nkb_im2row(int dtype, const float* x, void* col, int N, int Cin, int H, int W, int R, int S, int stride, int pad,
               int Kp, nkb_stream_t stream);

/* Packed ResNet stem: Conv2d(C<=4 -> Cout, 7x7, stride 2, pad 3) (timm resnet.py conv1, reached from
 * nkb_classification/model.py:84) computed as an implicit GEMM straight from the channel-padded NHWC image instead of
 * an im2row matrix.
 *   nkb_stem_pack   NCHW fp32 image -> xp[N][H][Wp][4] in the compute dtype, Wp = W rounded up to even
 *   nkb_stem_wprep  fp32 master [Cout][7][7][C] -> wp[Cout][nkb_stem_weight_cols(dtype)] in the compute dtype
 *   nkb_stem_conv   y[N*P*Q][ldy] = conv(xp, wp), optional per-tile BN partial sums (as nkb_conv_gemm, mode 0)
 *   nkb_stemp_conv  the same product for bf16 / 64 output channels with image rows streamed ONCE through an LDS ring (csrc/stemp.hip:
 *                   one workgroup per image or band of output rows); stats[tiles][2][64] with tiles = nkb_stemp_tiles(...) (0: not
 *                   eligible -> nkb_stem_conv) — one partial-sum row per workgroup, feeds nkb_bn_finalize like nkb_stem_conv's
 *   nkb_stem_wgrad  dwp[Cout][224] (fp32, zeroed by the caller) += dY^T x window(xp)
 *   nkb_stem_wfold  dw[Cout][7][7][C] += dwp (drops the padding columns) */
int nkb_stem_pack(int dtype, const float* x, void* out, int N, int C, int H, int W, nkb_stream_t stream);
int nkb_stem_wprep(int dtype, const float* w, void* wp, int Cout, int C, nkb_stream_t stream);
int nkb_stem_weight_cols(int dtype);
int nkb_stem_conv(int dtype, const void* xp, const void* wp, void* y, float* stats, int N, int H, int W, int Cout,
                  int ldy, nkb_stream_t stream);
/* R[co][ci] (fp32, OVERWRITTEN) = g^T a over M pixels (csrc/gramr.hip): the Gram-form closing stage's backward product on the main stream
 * (nkb_conv_wgrad_assign's 1x1 form) for bf16, (co, ci) = (256, 64) or (512, 128): one workgroup per CU streams its pixels through LDS and
 * keeps the whole result in registers; one fp32 slab per workgroup in `workspace` (>= nkb_gramr_workspace_floats floats; 0 = not
 * eligible -> nkb_conv_wgrad_assign), summed in workgroup order.  g: [M][ldg], a: [M][lda].  mode bit 0: R overwritten (else accumulated
 * into); bit 1: R stored [ci][co] — the weight gradient of a 1x1 convolution with Cin = co (g = its input), Cout = ci (a = dY). */
long long nkb_gramr_workspace_floats(int dtype, long long M, int co, int ci);
int nkb_gramr(int dtype, const void* g, int ldg, const void* a, int lda, float* R, long long M, int co, int ci, int mode,
              float* workspace, long long workspace_floats, nkb_stream_t stream);
int nkb_stemp_tiles(int dtype, int N, int H, int W, int Cout);
/* nkb_stem_wgrad's product on the same ring (dwp[64][224] fp32 += dY^T x window(xp); one fp32 slab per workgroup in `workspace`, summed in
 * workgroup order): for the shapes nkb_stemp_tiles admits, workspace >= nkb_stemp_wgrad_workspace_floats floats. */
long long nkb_stemp_wgrad_workspace_floats(int dtype, int N, int H, int W, int Cout);
int nkb_stemp_wgrad(int dtype, const void* dy, const void* xp, float* dwp, int N, int H, int W, int Cout, int lddy, float* workspace,
                    long long workspace_floats, nkb_stream_t stream);
int nkb_stemp_conv(int dtype, const void* xp, const void* wp, void* y, float* stats, int N, int H, int W, int Cout, int ldy,
                   nkb_stream_t stream);
int nkb_stem_wgrad(int dtype, const void* dy, const void* xp, float* dwp, int N, int H, int W, int Cout, int lddy,
                   float* workspace, long long workspace_floats, nkb_stream_t stream);
long long nkb_stem_wgrad_workspace_floats(int dtype, int N, int H, int W, int Cout);
int nkb_stem_wfold(int dtype, const float* dwp, float* dw, int Cout, int C, nkb_stream_t stream);

/* fp32 master filter [A][B][C] -> compute-dtype copy (mode 0: rows padded to ld; mode 1: transposed [C][B][ld];
 * modes 2..5: transposed parity class 2*ph+pw of a 3x3 stride-2 filter, [C][Rc*Sc][ld], see nkb_conv_dgrad_s2class). */
int nkb_wprep(int dtype, const float* src, void* dst, int A, int B, int C, int ld, int mode, nkb_stream_t stream);
/* All weight re-layouts of a step in one launch: jobs[j] = {src element offset from `base`, dst device pointer, A, B, C,
 * ld, mode (as nkb_wprep), index of the job's first block}, 8 x int64 per job on the device; a job occupies
 * nkb_wprep_job_blocks(...) consecutive blocks, total_blocks = sum over jobs. */
int nkb_wprep_multi(int dtype, const float* base, const long long* jobs, int njobs, int total_blocks, const void* shadow, nkb_stream_t stream);
int nkb_wprep_block_elems(void);
long long nkb_wprep_job_blocks(int A, int B, int C, int ld, int mode); /* blocks a job occupies in nkb_wprep_multi */
int nkb_add2d(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, nkb_stream_t stream);
int nkb_colsum(int dtype, const void* x, float* out, int rows, int C, int ld, nkb_stream_t stream);
int nkb_pad_cast(int dtype, const float* src, void* dst, int rows, int C, int ld_src, int ld_dst, float mul,
                 nkb_stream_t stream);

/* ---- transformer (timm VisionTransformer) ------------------------------------------------------------------ */
/* Batched GEMM y[z][m][n] = sum_k x[z][m][k] w[z][n][k] and its transposed-A form out[z][a][b] = sum_m A[z][m][a] B[z][m][b];
 * z = zo*inner + zi, element offsets zo*s?o + zi*s?i (attention products over all image x head pairs). */
int nkb_gemm_batched(int dtype, const void* x, const void* w, void* y, int M, int N, int K, int ldx, int ldw, int ldy,
                     int outer, int inner, long long sxo, long long sxi, long long swo, long long swi, long long syo,
                     long long syi, int out_f32, nkb_stream_t stream);
int nkb_gemm_tn_batched(int dtype, const void* a, const void* b, void* out, int M, int Na, int Nb, int lda, int ldb,
                        int ldo, int outer, int inner, long long sao, long long sai, long long sbo, long long sbi,
                        long long soo, long long soi, nkb_stream_t stream);
/* Linear layer with a fused activation epilogue. act 1: y2 = xW^T+b, y = gelu(y2) (exact erf). act 2: y = (xW^T) * gelu'(aux).
 * act 3: y = (xW^T) where 0 < aux < 6, else 0 (ReLU6 backward; aux = the ReLU6 output, whose forward is nkb_conv_gemm(relu = 2)).
 * act 4: y = (xW^T) * aux (GELU backward with aux = gelu'(pre) kept by nkb_gelu_fwd_dgelu or by act 5).
 * act 5: y = gelu(xW^T+b), y2 = gelu'(xW^T+b): fc1 forward that leaves exactly what act 4 needs and never stores the pre-activation
 * (bf16 shapes of the 256 x 256 eight-phase core only: nkb_linear_gelu_fused_ok() == 1; erf by Abramowitz-Stegun 7.1.26,
 * |error| <= 1.5e-7, below the bf16 rounding of both outputs). */
int nkb_linear_gelu(int dtype, int act, const void* x, const void* w, const float* bias, const void* aux, void* y, void* y2,
                    int M, int K, int N, nkb_stream_t stream);
int nkb_linear_gelu_fused_ok(int dtype, int M, int K, int N);
/* y = add + row_scale[m / rows_per_sample] * (xW^T + b): a residual branch under per-sample stochastic depth (timm-style DropPath of
 * the unicom blocks: the reference reaches it through unicom's Block.forward from /root/reference/nkb_classification/model.py:77-79)
 * in one launch; bf16 shapes of the eight-phase core only (nkb_linear_gelu_fused_ok() == 1). */
int nkb_linear_residual_scaled(int dtype, const void* x, const void* w, const float* bias, const void* add, const float* row_scale,
                               int rows_per_sample, void* y, int M, int K, int N, nkb_stream_t stream);
/* LayerNorm over the last dim (biased variance). backward=0: in = x -> out = y, writes mean/rstd.
 * backward=1: in = dy, x = saved input -> out = dx (+ add), dgamma/dbeta accumulated (workspace: per-block partials added in a
 * fixed order; NULL: atomics). Strides in elements.
 * yq / q_state / q_kind (optional, D % 256 == 0, out_stride == D): fp8 copy of the output rows for the fp8 GEMM that consumes
 * them (see nkb_fp8_quantize).  Backward (workspace form only): the copy is of row_scale[row / rows_per_sample] * dx (row_scale
 * optional: the stochastic-depth factor of the branch the gradient enters) and colsum[D] += its column sums — the operand and
 * the bias gradient nkb_fp8_quantize_colsum would make of dx for the Linear backward that consumes it.  q_kind 2 (backward, bf16):
 * yq is a bf16 [rows][D] tensor that receives row_scale[row / rows_per_sample] * dx (nkb_scale_rows of the stored dx; no state, no colsum). */
int nkb_layernorm(int dtype, int backward, const void* in, long long in_stride, const void* x, long long x_stride,
                  const float* gamma, const float* beta, float* mean, float* rstd, const void* add, void* out,
                  long long out_stride, float* dgamma, float* dbeta, int rows, int D, float eps, float* workspace,
                  void* yq, float* q_state, int q_kind, const float* row_scale, int rows_per_sample, float* colsum,
                  nkb_stream_t stream);
/* (yq / q_state / q_kind, forward only, optional: an fp8 copy of the output rows — packed [rows][D] bytes, scale q_state[0], amax
 * into q_state[2], kind as in nkb_fp8_quantize — for the fp8 GEMM that consumes the normalised rows; D % 256 == 0, out_stride == D.) */
size_t nkb_layernorm_workspace_floats(int D); /* backward: optional scratch for the deterministic dgamma/dbeta reduction */
/* Workspace-form backward with dgamma = dbeta = NULL leaves only the per-block partial rows in `workspace`; this call then adds their
 * ordered sums to dgamma / dbeta (and colsum when planes == 3: the launch wrote an fp8 copy) — on any stream ordered after that
 * launch, so that the two small reduction launches need not sit in the backward chain.  rows, D as in that launch. */
int nkb_layernorm_param_reduce(float* workspace, int rows, int D, int planes, float* dgamma, float* dbeta, float* colsum,
                               nkb_stream_t stream);
/* exact-erf GELU: dy == NULL -> out = gelu(x); else out = dy * gelu'(x) */
/* forward that also stores gelu'(x) (timm Mlp.act, backward then is the act-4 epilogue of nkb_linear_gelu) */
int nkb_gelu_fwd_dgelu(int dtype, const void* x, void* y, void* gp, long long n, nkb_stream_t stream);
int nkb_gelu(int dtype, const void* x, const void* dy, void* out, long long n, nkb_stream_t stream);
/* softmax over fp32 score rows (forward: p = softmax(scale*s); backward: ds = scale*p*(dp - sum dp*p)), zero padded to ldp */
int nkb_attn_softmax(int dtype, int backward, const float* s, int lds, const void* p_in, void* out, int ldp,
                     long long rows, int cols, float scale, nkb_stream_t stream);
/* Fused attention (bf16, head dim 64, T <= 256): O = softmax(QK^T*scale)V with per-row log-sum-exp; the backward half
 * recomputes P from Q, K, LSE and writes P and dS = P o (dP - rowsum(P o dP)) * scale as [B*H][T][ldp] bf16. */
int nkb_attn_forward(int dtype, const void* qkv, void* out, float* lse, int B, int T, int H, int dh, float scale,
                     void* outq, float* q_state, nkb_stream_t stream);
/* dq != NULL: the same pass also forms dQ = dS K and stores it at dq[(b*T + q) * ld_dq + h*64 ..] (e.g. the Q third of the
 * d_qkv matrix, ld_dq = 3*H*64), so that only dV = P^T dO and dK = dS^T Q remain as separate GEMMs. */
int nkb_attn_backward_ds(int dtype, const void* qkv, const void* dout, const float* lse, void* P, void* dS, int ldp, int B,
                         int T, int H, int dh, float scale, void* dq, long long ld_dq, nkb_stream_t stream);
/* Whole attention backward in one pass per (image, head): dQ, dK, dV written into the three thirds of dqkv
 * ([B*T][3*H*64], same layout as qkv).  Reads qkv, dout, the forward output `out` (for delta = rowsum(dout o out)) and lse;
 * no score matrix reaches memory (bf16, head dim 64, T <= 256). */
/* (colsum / colsum_work, optional: colsum[3 H dh] += the column sums of the stored d_qkv rows = the qkv projection's bias gradient,
 * from per-image sums in colsum_work[B][3 H dh] added in image order — no separate pass over d_qkv) */
int nkb_attn_backward(int dtype, const void* qkv, const void* dout, const void* out, const float* lse, void* dqkv, int B, int T,
                      int H, int dh, float scale, void* dqkv_q, float* q_state, float* colsum, float* colsum_work, nkb_stream_t stream);
/* (outq / dqkv_q with q_state, optional: fp8 copies of the outputs — e4m3 of `out`, e5m2 of `dqkv`, packed rows, scale q_state[0],
 * amax into q_state[2] as nkb_fp8_quantize would — for the fp8 projection / qkv-gradient GEMMs that consume them.) */
int nkb_head_transpose(int dtype, const void* in, int ld_in, long long sio, long long sii, int outer, int inner, void* out,
                       int T, int dh, int ldt, nkb_stream_t stream);
/* token assembly: forward x[b][t] = (t == 0 ? cls : tok[b][t-1]) + pos[t]; cls == NULL (unicom layout, no class token):
 * x[b][t] = tok[b][t] + pos[t].  backward (class-token layout only): tok := x[:, 1:]. */
int nkb_vit_assemble(int dtype, int backward, void* tok, const float* cls, const float* pos, void* x, int B, int Tn, int D,
                     nkb_stream_t stream);
/* ReLU6 (unicom Mlp.act, reached through unicom.load at model.py:77-79): dy == NULL -> out = min(max(x,0),6);
 * else out = dy where 0 < x < 6, 0 elsewhere */
int nkb_relu6(int dtype, const void* x, const void* dy, void* out, long long n, nkb_stream_t stream);
/* stochastic depth (unicom Block.drop_path): out[r][i] = x[r][i] * scale[r] (+ add[r][i]); also its own backward */
int nkb_scale_rows(int dtype, const void* x, const void* add, void* out, const float* scale, int rows, long long inner,
                   nkb_stream_t stream);
/* Split-K Linear layer, second half: partial[z][m][n] (fp32, from nkb_gemm_batched with one batch entry per K slice) ->
 * y[m][n] = sum_z partial + bias[n] in the compute dtype, and the per-128-row-tile channel sums nkb_bn_finalize expects
 * (stats may be NULL).  For skinny GEMMs with a very long reduction (unicom feature[0]: 128 x 262144 -> 1024). */
int nkb_splitk_reduce(int dtype, const float* partial, int splits, int M, int N, void* y, int ldy, const float* bias,
                      float* stats, nkb_stream_t stream);
/* Eval-mode BatchNorm folding (val_epoch, engine.py:88-117): dst[Cout][K] = w[Cout][K] * scale[Cout] in the compute dtype;
 * the folded filter + bias = shift + the conv epilogue's residual add / ReLU replace conv -> bn -> act in eval mode. */
int nkb_wfold(int dtype, const float* w, const float* scale, void* dst, int Cout, int K, nkb_stream_t stream);
/* Input pipeline on the device (replaces the PadIfNeeded -> Horizontal/VerticalFlip -> Normalize -> ToTensorV2 tail of
 * the albumentations stack, configs/singletask_config.py:162-219, and lets engine.py:40's H2D copy move uint8):
 * src [B][Hs][Ws][3] uint8 (device), sizes [B][2] int32 (h, w of the valid top-left region; NULL = Hs x Ws),
 * flags [B] uint8 (bit 0 horizontal, bit 1 vertical flip; NULL = none), out [B][3][Ho][Wo] fp32 (device);
 * mean / stdev: 3 HOST floats each (fractions of 255 as in A.Normalize); fill: pad value in 0..255. */
int nkb_image_prep(const unsigned char* src, const int* sizes, const unsigned char* flags, float* out, int B, int Hs, int Ws,
                   int Ho, int Wo, const float* mean, const float* stdev, float fill, nkb_stream_t stream);
/* out = keep ? in/(1-p) : 0 (+ add); forward draws keep from a hash of (seed, index) and stores it in mask */
int nkb_dropout(int dtype, int backward, const void* in, const void* add, void* out, unsigned char* mask, long long n,
                float p, unsigned long long seed, nkb_stream_t stream);
int nkb_colsum2d(int dtype, const void* x, float* out, long long rows, int C, long long ld, float* workspace,
                 nkb_stream_t stream);   /* workspace: NULL (atomics) or >= 256 * C floats (ordered two-stage sum) */

/* Losses (kind 0: CrossEntropyLoss(weight); kind 1: FocalLoss(alpha, gamma) over un-ignored rows, losses.py:59-94).
 * reduction 0 "mean": out2[0] = loss, out2[1] = 1/normaliser; 1 "sum" / 2 "none": out2[0] = sum, out2[1] = 1 (the per-row
 * losses of "none" are row_state[i].loss, three floats per row).  A label outside [0, C) that is not ignore_index makes the
 * loss NaN (torch asserts on the device there).  probs/argmax double as the logger's softmax/argmax. */
int nkb_loss_forward(int kind, const float* logits, int ld, const long long* target, int B, int C,
                     const float* class_weight, float gamma, long long ignore_index, float* probs, int ldp,
                     int* argmax, void* row_state, float* out2, int reduction, nkb_stream_t stream);
size_t nkb_loss_row_state_bytes(int B);
int nkb_loss_backward(const float* probs, int ldp, const long long* target, const void* row_state, const float* out2,
                      const float* grad_out, int grad_out_per_row, int B, int C, float* dlogits, int ldd,
                      nkb_stream_t stream);

/* Fused flat-arena optimizer step. kind: 0 adam, 1 nadam (decoupled wd), 2 radam, 3 sgd.
 * skip_flag (device float, may be NULL): the launch does nothing when *skip_flag != 0 — the skipped step of
 * GradScaler.step (engine.py:59) decided on the device. */
int nkb_optim_step(int kind, float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n, float lr,
                   float wd, float beta1, float beta2, float eps, float grad_scale, float c0, float c1, float c2,
                   float c3, const float* skip_flag, nkb_stream_t stream);
/* Data-parallel gradient exchange in bf16 with fp32 accumulation (new functionality: the reference is single-device,
 * /root/reference/train.py:98): out[i] = sum_{p < nparts} parts[p * stride + i] accumulated in fp32 in part order, optional bf16 copy of
 * the sum; nparts = 1 widens a received bf16 bucket back into the fp32 gradient arena. */
int nkb_bucket_sum_bf16(const void* parts, long long stride, int nparts, float* out, void* out_bf16, long long n, nkb_stream_t stream);
/* Gradient scaler (train.py:37 torch.cuda.amp.GradScaler; engine.py:55-60): in-place g *= 1 / *scale with an inf/nan check
 * (*found_inf = 1 when any unscaled value is not finite), and the scale / growth-tracker update of GradScaler.update();
 * the update also copies found_inf to *last_found_inf and clears found_inf.  scale, found_inf: device floats. */
int nkb_grad_unscale_check(float* g, long long n, const float* scale, float* found_inf, nkb_stream_t stream);
int nkb_scaler_update(float* scale, int* growth_tracker, float* found_inf, float* last_found_inf, float growth,
                      float backoff, int interval, nkb_stream_t stream);
int nkb_segment_sumsq(const float* x, const long long* offsets, int nseg, float* out, nkb_stream_t stream);

/* fp8 path (BASELINE configs[4] "unicom ViT-L/14 ... fp8"; the reference itself has no fp8 mode — engine.py:43-47 is fp16 autocast):
 * per-tensor scaled OCP fp8 operands for the Linear contractions, fp32 accumulation, bf16 result.
 *   nkb_fp8_quantize: dst[i] = fp8(src[i] * state[0]) (kind 0: e4m3, saturating at 448; kind 1: e5m2, 57344), and
 *                     state[2] = max(state[2], max |src|) — the amax the NEXT scale is derived from (delayed scaling).
 *   nkb_fp8_amax:     state[2] = max(state[2], max |src|) only (just-in-time scaling of weights: amax, update, quantize).
 *   nkb_fp8_scale_update: state[0] = fp8_max / state[2], state[1] = 1 / state[0] (unchanged when no value was seen), state[2] = 0.
 *   nkb_gemm_fp8:     y[M][N] = (xq . wq^T) * *deq_x * *deq_w (+ bias) (+ add), ReLU (1) / ReLU6 (2); mode 0: both operands
 *                     e4m3 (forward), mode 1: xq is e5m2 (a gradient) and wq e4m3 (data gradient).  K % 128 == 0, N % 256 == 0.
 *                     aux (optional, [M][ldy] in bf16): aux_mode 0 multiplies the result by it (saved activation derivative),
 *                     aux_mode 1 keeps the result where 0 < aux < 6 (ReLU6 backward, aux = the clamped forward output).
 *                     yq / q_state / q_kind (optional): second output yq[M][N] bytes = fp8(y * q_state[0]) (kind as in
 *                     nkb_fp8_quantize) with q_state[2] = max(q_state[2], max |y|) — the operand of the next fp8 GEMM without a
 *                     separate quantisation pass over y.
 *                     mask_out / mask_in (optional, with yq): the ReLU6 mask as one bit per element ([M][N / 8] bytes): a relu == 2
 *                     launch writes it (bit = 0 < value < 6) and may then omit the bf16 output (y == NULL); a data-gradient
 *                     launch keeps its result where the bit is set — 1/16 of the bytes of the bf16 `aux` form.
 *                     colsum / colsum_work (optional, with mask_in and yq; M % 256 == 0): colsum[N] += the column sums of the
 *                     (bf16-rounded) result — the bias gradient of the Linear this data gradient feeds — from per-tile partial
 *                     sums in colsum_work ([M / 256][N] floats, added in tile order); y may then be NULL as well: the gradient
 *                     exists only as the fp8 operand of the next GEMMs.
 *                     row_scale / rows_per_sample (optional, with add): y = add + row_scale[m / rows_per_sample] * (product + bias):
 *                     per-sample stochastic depth of the residual branch (unicom blocks) inside the epilogue.
 * state: three device floats {scale, 1 / scale, running amax}. */
int nkb_fp8_quantize(int dtype, int kind, const void* src, long long n, float* state, void* dst, nkb_stream_t stream);
int nkb_fp8_amax(int dtype, const void* src, long long n, float* state, nkb_stream_t stream);
/* nkb_fp8_quantize of a [rows][C] bf16 matrix (row stride ld) that also adds the matrix's column sums to colsum[C] — the bias
 * gradient of a Linear layer when src = dY, in the one pass that reads the unquantised values.  C % 512 == 0; workspace of
 * nkb_fp8_quantize_colsum_workspace_floats(rows, C) floats (partial sums per row block, added in block order).  With row_scale
 * (per sample, rows_per_sample rows each) the matrix quantised and summed is row_scale[row / rows_per_sample] * src. */
long long nkb_fp8_quantize_colsum_workspace_floats(long long rows, int C);
int nkb_fp8_quantize_colsum(int kind, const void* src, long long rows, int C, long long ld, float* state, void* dst, float* colsum,
                            float* workspace, const float* row_scale, int rows_per_sample, nkb_stream_t stream);
int nkb_fp8_scale_update(float* state, int kind, nkb_stream_t stream);
/* Many tensors in one launch (a model's weight matrices): jobs = device array of njobs x 6 int64 {src (bf16), dst (bytes),
 * n (multiple of 8), state (3 floats), kind, first block}; blocks per job from nkb_fp8_job_blocks(n).  pass 0: amax only;
 * 1: quantise with state[0] (+ amax); 2: scale from amax; 3: amax <- 0; 4: scale from amax, then amax <- 0 (the per-step update
 * of delayed-scaling sites, all of them in one launch). */
long long nkb_fp8_job_blocks(long long n);
int nkb_fp8_multi(int pass, const long long* jobs, int njobs, long long total_blocks, nkb_stream_t stream);
/* fp8 weight gradient: dw[Cout][Cin] += (*deq_g * *deq_x) * sum_m gq[m][cout] * xq[m][cin] with gq = the e5m2 bytes [M][ldg] the
 * data gradient consumed and xq = the e4m3 bytes [M][ldx] the forward GEMM consumed (reference: the autograd weight gradient of
 * the nn.Linear layers reached from /root/reference/nkb_classification/engine.py:55-58).  M % 128 == 0, Cin % 256 == 0,
 * Cout % 256 == 0; workspace of nkb_wgrad_fp8_workspace_floats() floats (-1: shape not supported); per-split partial tiles are
 * added in split order (deterministic).  Bias gradients are not part of it. */
long long nkb_wgrad_fp8_workspace_floats(int M, int Cin, int Cout);
int nkb_wgrad_fp8(const void* gq, const void* xq, float* dw, const float* deq_g, const float* deq_x, int M, int Cin, int ldx,
                  int Cout, int ldg, float* workspace, long long workspace_floats, nkb_stream_t stream);
int nkb_gemm_fp8(int mode, const void* xq, const void* wq, void* y, const float* bias, const void* add, const void* aux,
                 int aux_mode, void* yq, float* q_state, int q_kind, const float* row_scale, int rows_per_sample,
                 void* mask_out, const void* mask_in, float* colsum, float* colsum_work, const float* deq_x, const float* deq_w,
                 int M, int K, int N, int ldx, int ldw, int ldy, int ldadd, int relu, nkb_stream_t stream);

/* Envelope of the 256 x 256 eight-phase GEMM core that nkb_conv_gemm / nkb_linear_gelu use for wide plain 1x1 / Linear launches
 * (csrc/gemm8p.hip): on = 0 / 1; min_tiles, min_k > 0 replace the defaults (192 tiles, K >= 768).  Tests and A/B timing. */
void nkb_gemm8p_config(int on, int min_tiles, int min_k);
/* the ragged last row block (M % 256 rows) of a persistent 256 x 256 GEMM launch on its own small kernel where that saves the launch a
 * round of tiles (csrc/gemm8p.hip, gemm8p_ragged_kernel): 1 (default) = on, 0 = every row block on the persistent kernel.  Tests / A-B timing. */
void nkb_gemm8p_ragged(int on);

/* Host replay of a recorded step (csrc/plan.hip).  The reference has no counterpart: its step is re-traced by the Python
 * interpreter every iteration (engine.py:36-75 -> torch dispatcher).  nkb_classification/hip.py records the entry points one
 * forward / backward / optimizer pass calls — every buffer lives in the persistent workspace, so the argument lists are constant
 * — and nkb_plan_run re-issues them from a flat table: entry.fn >= 0 is an index into the recordable entry points
 * (nkb_plan_fn_name / nkb_plan_fn_args: name and argument count; pointers and streams in .p, every integer type in .i, floats
 * in .f, in declaration order), entry.fn < 0 one of the stream operations that sit between launches:
 *   NKB_PLAN_EVENT_RECORD       a[0].p = hipEvent_t, a[1].p = stream
 *   NKB_PLAN_STREAM_WAIT_EVENT  a[0].p = stream,     a[1].p = hipEvent_t
 *   NKB_PLAN_MEMSET             a[0].p = device pointer, a[1].i = bytes (set to zero), a[2].p = stream
 * Returns 0, or the code of the first failing entry (index in *failed, text in nkb_last_error()); later entries are not issued. */
typedef union { void* p; long long i; float f; } NkbPlanArg;
#define NKB_PLAN_MAX_ARGS 32
typedef struct { int fn; int nargs; NkbPlanArg a[NKB_PLAN_MAX_ARGS]; } NkbPlanEntry;
#define NKB_PLAN_EVENT_RECORD (-1)
#define NKB_PLAN_STREAM_WAIT_EVENT (-2)
#define NKB_PLAN_MEMSET (-3)
int nkb_plan_fn_count(void);
const char* nkb_plan_fn_name(int id);
int nkb_plan_fn_args(int id);
int nkb_plan_max_args(void);
size_t nkb_plan_entry_bytes(void);
int nkb_plan_run(const void* entries, int n, int* failed);

/* Per-launch HIP-event profiler (bench.py's roofline leg). */
void nkb_prof_enable(int on);
int nkb_prof_collect(double* ms, long long* launches, double* work, double* bytes, int slots);
int nkb_prof_collect_raw(int* kernel_id, double* ms, double* work, double* bytes, int cap);
const char* nkb_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif
