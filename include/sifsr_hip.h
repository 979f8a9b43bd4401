/* sifsr_hip.h -- C ABI of libsifsr_hip.so: the SIF-CNN-SR hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  Plain pointers and sizes only: every `float*`
 * below is a DEVICE pointer (fp32), `stream` is a hipStream_t passed as void*, all tensors are
 * dense.  Functions only enqueue work on `stream` (no allocation, no synchronisation) and
 * return 0 on success, a SIFSR_ERR_* code (1001..1003) for bad arguments, or the hipError_t of a
 * failed launch.  The reference is pure Python on PyTorch; each entry point names the reference
 * interface (path relative to the reference repo root, file:line) whose ATen dispatch it replaces.
 * The ctypes binding a maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Layouts: model input  x  (B,2,H,W) NCHW  == torch.cat((lst_up, ndvi), 1), train_model_B_gradFTM.py:94
 *          model output sr (B,1,H,W) NCHW, model.py:645
 *          params: ONE flat fp32 buffer of 282,705 floats in `model.parameters()` order
 *                  (conv OIHW weights / BN weight / BN bias per layer, then outlay.weight, outlay.bias)
 *          running: flat fp32 [per BN layer: running_mean(C), running_var(C)], nbt: int64[17]
 *          activations between kernels: NHWC fp32 (internal).
 * Sizes: the model entry points take H, W = any multiples of 8 that are >= 24 (three 2x poolings, the reference's own
 * constraint, model.py:597-603; tiles are masked where a level is not a multiple of the 16x16 conv tile); the loss
 * operators take any H, W >= 10, multiples of 4 where a /4 decimation is involved.  Other sizes: SIFSR_ERR_SHAPE.
 */
#ifndef SIFSR_HIP_H
#define SIFSR_HIP_H
#include <stddef.h>

#ifdef __cplusplus
#define SIFSR_API extern "C" __attribute__((visibility("default")))
#else
#define SIFSR_API
#endif

/* ---- introspection (host only, no GPU touched) -------------------------------------------- */
SIFSR_API int sifsr_abi_version(void);   /* 3 since round 3 (sifsr_conv3x3_bwd16 added, the split-bf16 entry points removed); 2 = round 2 */
SIFSR_API int sifsr_num_params(void);   /* 282705 */
SIFSR_API int sifsr_num_running(void);  /* 1184 = 2 * 592 channels */
/* out[17][8] = {cin, cout, level, w_off, gamma_off, beta_off, run_off, ch_off}; returns 17 */
SIFSR_API int sifsr_layer_table(int* out, int capacity_rows);

/* ---- ModelB_2 (model.py:533-645) ------------------------------------------------------------ */
SIFSR_API size_t sifsr_model_workspace_bytes(int B, int H, int W, int training);
/* Debug/test introspection: float offsets of the named workspace regions, in this order:
 * y[17] (raw conv outputs, NHWC), P[3], R[3], U[3], g[17] (dL/d relu(bn(y)); dL/dy is not stored; g[0] leaves the backward multiplied
 * by its ReLU mask where the first layer's weight gradient runs in its linear form), dyB[3] (unused), gP[3], gU[3],
 * mean, invstd, scale, shift (per-channel vectors of all layers, indexed by ch_off).  Returns the count (56). */
SIFSR_API int sifsr_model_workspace_regions(int B, int H, int W, size_t* out, int capacity);
/* ModelB_2.forward, model.py:608-645.  training != 0: batch statistics, running-stat update
 * (momentum, unbiased var) and nbt += 1, activations kept in `workspace` for sifsr_model_backward;
 * training == 0: model.eval() semantics (running statistics), predict.py:68,100. */
SIFSR_API int sifsr_model_forward(const float* x, float* sr, const float* params, float* running, long long* nbt,
                                  void* workspace, size_t workspace_bytes, int B, int H, int W, int training,
                                  float momentum, float eps, void* stream);
/* autograd transpose of the above (loss.backward(), train_model_B_gradFTM.py:119): dsr = dL/dsr;
 * writes ALL 282,705 gradients ("=" semantics) into grads (same layout as params). */
SIFSR_API int sifsr_model_backward(const float* x, const float* dsr, const float* params, float* grads,
                                   void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
/* The same two calls with a compute mode: 0 = fp32 (identical to the calls above), 1 = BASELINE.json config 5,
 * "bf16 mixed precision, MFMA-bf16 conv tiles": every activation-like tensor INSIDE the network (raw conv outputs, pooled /
 * residual / upsampled tensors and the gradients with respect to them) is STORED as bf16 -- half the HBM bytes of every pass --
 * and the operands of the sixteen 3x3 MFMA convs (activations after BatchNorm+ReLU, weights, dy) are rounded to bf16 while
 * staging and contracted with v_mfma_f32_16x16x32_bf16, two taps per MFMA (the weight-gradient pass uses the K = 16 form over 16
 * pixels).  Accumulation, BatchNorm statistics and coefficients, the arithmetic of every non-conv kernel, x, sr, dsr, master
 * weights, gradients of the parameters and the optimizer stay fp32.  The workspace size is that of the fp32 mode (the bf16
 * tensors use the first half of their regions).  Forward and backward of one step must use the same mode. */
SIFSR_API int sifsr_model_forward_ex(const float* x, float* sr, const float* params, float* running, long long* nbt,
                                     void* workspace, size_t workspace_bytes, int B, int H, int W, int training,
                                     float momentum, float eps, int compute, void* stream);
SIFSR_API int sifsr_model_backward_ex(const float* x, const float* dsr, const float* params, float* grads, void* workspace,
                                      size_t workspace_bytes, int B, int H, int W, int compute, void* stream);

/* ---- 3x3 convolution pieces (nn.Conv2d(k=3,padding=1,padding_mode='replicate'), model.py:135,138,507) */
/* OIHW -> MFMA fragment order: wfwd 9*cin*cout floats (forward operand); wdgrad 4*9*cin*cout floats: the
 * transposed+flipped fp32 dgrad operand (n = 9*cin*cout floats) followed by the two bf16 packs of n/2 floats each,
 * [fwd | dgrad] (config 5), and 2n unused floats (they held the packs of the split-bf16 mode removed in round 3). */
SIFSR_API int sifsr_pack_conv_weights(const float* w_oihw, int cin, int cout, float* wfwd, float* wdgrad, void* stream);
/* y = conv(cat([a0, a1], C)), a_i = relu(src_i*scale_i+shift_i) if scale_i != NULL else src_i (NHWC, C_i % 16 == 0;
 * src1 may be NULL).  stat_partials: NULL or [sifsr_conv3x3_stat_blocks()][cout][2] per-workgroup (sum, sumsq) of y. */
SIFSR_API int sifsr_conv3x3_stat_blocks(int B, int H, int W, int cout);
SIFSR_API int sifsr_conv3x3_fwd(const float* src0, int C0, const float* scale0, const float* shift0,
                                const float* src1, int C1, const float* scale1, const float* shift1,
                                const float* wfwd, float* y, int cout, float* stat_partials, int B, int H, int W,
                                void* stream);
/* g_in = conv^T(dy) incl. replicate-border fold; channels [0,C0) -> g0, [C0,cin) -> g1 (NULL if unused);
 * addend (cin channels, may be NULL) is added to g0 (needs g1 == NULL). */
SIFSR_API int sifsr_conv3x3_dgrad(const float* dy, int cout, const float* wdgrad, const float* w_oihw, int cin,
                                  float* g0, int C0, float* g1, int C1, const float* addend, int B, int H, int W,
                                  void* stream);
/* The input gradient as ModelB_2's backward runs it since round 2: the layer's BatchNorm+ReLU backward (model.py:136-137)
 * is applied while the operand is staged, dy = sc*g*[z>0] + k1*z + k0 with z = y*sc + sh, so dL/dy is never written to
 * HBM.  g = dL/d relu(bn(y)) and y = the layer's raw conv output (both NHWC, cout channels), coef_f = [sc|sh|k1|k0] from
 * sifsr_bn_relu_bwd_coef.  border: cout-channel NHWC scratch of the same size, only its image-border pixels are written
 * (dL/dy there, for the replicate-padding fold). */
SIFSR_API int sifsr_conv3x3_dgrad_fused(const float* g, const float* y, const float* coef_f, int cout, const float* wdgrad,
                                        const float* wwd, int cin, float* g0, int C0, float* g1, int C1, const float* addend,
                                        float* border, int B, int H, int W, void* stream);   /* wwd: Winograd pack or NULL */
/* Winograd F(2x2,3x3) forms of the forward and input-gradient convolutions (what ModelB_2 runs in fp32 for layers with
 * <= 64 output channels and even H, W): the channel contraction runs on the 16 transform-domain positions of a 2x2 output
 * patch instead of the 9 taps of each pixel -- 2.25x fewer matrix-core products, same fp32 accuracy class.  wwf / wwd =
 * the transform-domain weights G g G^T in fragment order (16*cin*cout floats each) from sifsr_pack_conv_weights_wino; the
 * tap packs are still passed (shapes the Winograd kernel does not take fall back to them; the border fold uses wdgrad).
 * stat_partials rows: sifsr_conv3x3_stat_blocks_wino(). */
SIFSR_API int sifsr_pack_conv_weights_wino(const float* w_oihw, int cin, int cout, float* wwf, float* wwd, void* stream);
SIFSR_API int sifsr_conv3x3_stat_blocks_wino(int B, int H, int W, int cin, int cout);
SIFSR_API int sifsr_conv3x3_fwd_wino(const float* src0, int C0, const float* scale0, const float* shift0,
                                     const float* src1, int C1, const float* scale1, const float* shift1,
                                     const float* wfwd, const float* wwf, float* y, int cout, float* stat_partials,
                                     int B, int H, int W, void* stream);
SIFSR_API int sifsr_conv3x3_dgrad_wino(const float* dy, int cout, const float* wdgrad, const float* wwd, int cin,
                                       float* g0, int C0, float* g1, int C1, const float* addend, int B, int H, int W,
                                       void* stream);
/* bf16 forms of the two calls above (BASELINE.json config 5, "bf16 mixed precision, MFMA-bf16 conv tiles"): every activation
 * tensor (src*, y, dy, g*, addend) is NHWC **bf16** -- the pointers are typed float* only for uniformity of the interface --,
 * values are widened on load, the folded BatchNorm+ReLU is applied in fp32, operands are rounded to bf16 for
 * v_mfma_f32_16x16x32_bf16 (two taps per MFMA), accumulation is fp32 and results are rounded to bf16 on store (statistics
 * are those of the rounded values).  Both take the `wdgrad` buffer of sifsr_pack_conv_weights, whose second half holds
 * the bf16 fragment packs [forward | dgrad]; scale / shift / stat_partials stay fp32. */
SIFSR_API int sifsr_conv3x3_fwd_bf16(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1,
                                     int C1, const float* scale1, const float* shift1, const float* wdgrad, float* y,
                                     int cout, float* stat_partials, int B, int H, int W, void* stream);
SIFSR_API int sifsr_conv3x3_dgrad_bf16(const float* dy, int cout, const float* wdgrad, int cin, float* g0, int C0, float* g1,
                                       int C1, const float* addend, int B, int H, int W, void* stream);
SIFSR_API size_t sifsr_conv3x3_wgrad_scratch_floats(int cin, int cout, int nblk);
/* dw (OIHW) = sum_pixels dy (x) a_in; deterministic 2-stage reduction through `scratch`. */
SIFSR_API int sifsr_conv3x3_wgrad(const float* src0, int C0, const float* scale0, const float* shift0,
                                  const float* src1, int C1, const float* scale1, const float* shift1,
                                  const float* dy, int cout, float* scratch, int nblk, float* dw, int B, int H, int W,
                                  void* stream);
/* The weight gradient with dL/dy formed while its tile is staged (see sifsr_conv3x3_dgrad_fused): g, y, coef_f as there. */
SIFSR_API int sifsr_conv3x3_wgrad_fused(const float* src0, int C0, const float* scale0, const float* shift0,
                                        const float* src1, int C1, const float* scale1, const float* shift1,
                                        const float* g, const float* y, const float* coef_f, int cout, float* scratch,
                                        int nblk, float* dw, int B, int H, int W, void* stream);
/* Winograd F(3x3, 2x2) form of the weight gradient (fp32, even H and W): 16 instead of 36 matrix-core products per 2x2
 * output patch and channel pair; every lane transforms its own (patch, channel) operand in registers from channel-plane
 * tiles in LDS.  y == coef_f == NULL: g is dL/dy itself; otherwise as sifsr_conv3x3_wgrad_fused.  Results differ from
 * sifsr_conv3x3_wgrad by fp32 rounding only (each workgroup applies the output transform A^T M A to its accumulators and writes a
 * tap-domain slab of 9 values per weight pair; the slabs are reduced in float64 in a fixed order). */
SIFSR_API size_t sifsr_conv3x3_wgrad_wino_scratch_floats(int cin, int cout, int nblk);
SIFSR_API int sifsr_conv3x3_wgrad_wino(const float* src0, int C0, const float* scale0, const float* shift0,
                                       const float* src1, int C1, const float* scale1, const float* shift1,
                                       const float* g, const float* y, const float* coef_f, int cout, float* scratch,
                                       int nblk, float* dw, int B, int H, int W, void* stream);
/* Input gradient AND weight gradient of a 16 -> 16 channel layer (the DoubleConvolution layers at full and half resolution,
 * model.py:135,138) from ONE read of its operands -- what ModelB_2's backward runs for inbloc.bloc.3, ub3.convbloc.bloc.3 and
 * db1.resblock.doubleconv.bloc.0/.3 since round 3: the two passes consume the same staged tiles (dL/dy formed from (g, y) while
 * staging; the forward input a_in = relu(x*x_scale + x_shift), or x itself with x_scale == NULL), so the layer's backward moves
 * 4 tensors through HBM instead of 7.  Every tensor NHWC with exactly 16 channels; H, W multiples of 16, >= 32 (returns the
 * shape error otherwise: use sifsr_conv3x3_dgrad_fused / _dgrad_wino + sifsr_conv3x3_wgrad_wino).
 *   y == coef_f == NULL: g is dL/dy itself (border unused); otherwise g, y, coef_f, border as for sifsr_conv3x3_dgrad_fused.
 *   gin = conv^T(dL/dy) incl. the replicate-border fold (+ addend);  dw = OIHW weight gradient (16,16,3,3).
 *   bn_partials != NULL: gin is the gradient w.r.t. relu(bn(bn_y)) of the layer below -- bn_y, bn_scale, bn_shift must BE x, x_scale,
 *   x_shift (the layer below is the layer whose raw output is this layer's input) -- and its BatchNorm-backward sums are emitted as sifsr_conv3x3_bwd16_stat_rows() rows of [16][2]
 *   (sum dz, sum dz*y per channel; dz = gin*[bn_y*scale+shift > 0]); add the rows up.  Not together with addend.
 *   scratch: sifsr_conv3x3_bwd16_scratch_floats() floats (one tap-domain weight-gradient slab per workgroup, reduced in float64).
 *   After sifsr_set_op_storage_bf16(1) the activation tensors (x, g, y, border, gin, addend, bn_y) are NHWC bf16 -- what the model's
 *   bf16 mode runs for these layers: half the bytes, fp32 arithmetic (fp32 MFMAs in the Winograd domain; the border fold rounds
 *   its operands to bf16 as the bf16 input-gradient kernel does). */
SIFSR_API int sifsr_conv3x3_bwd16_stat_rows(int B, int H, int W);
SIFSR_API size_t sifsr_conv3x3_bwd16_scratch_floats(int B, int H, int W);
SIFSR_API int sifsr_conv3x3_bwd16(const float* x, const float* x_scale, const float* x_shift, const float* g, const float* y,
                                  const float* coef_f, float* border, const float* wdgrad, const float* wwd, float* gin,
                                  const float* addend, const float* bn_y, const float* bn_scale, const float* bn_shift,
                                  float* bn_partials, float* scratch, float* dw, int B, int H, int W, void* stream);
/* The same for a layer whose output also feeds a pooling stage (inbloc.bloc.3 -> AvgPool2d(2,2), model.py:597, :528): its upstream
 * gradient is g (the decoder skip's) + the pooling adjoint of the half-resolution gradient pool_gp ((B, H/2, W/2, 16) NHWC), i.e.
 * g_eff[y][x] = g[y][x] + 0.25 * pool_gp[y/2][x/2], formed while staging -- the BatchNorm-backward reduction that precedes it in
 * ModelB_2's backward takes its sums from the same g_eff on the fly and no longer stores it (a 16-channel tensor write; the single-op
 * entry sifsr_bn_relu_bwd_coef with gpool still writes g_eff back into g).  y, coef_f, border required; coef_f must come from g_eff. */
SIFSR_API int sifsr_conv3x3_bwd16_pool(const float* x, const float* x_scale, const float* x_shift, const float* g,
                                       const float* pool_gp, const float* y, const float* coef_f, float* border,
                                       const float* wdgrad, const float* wwd, float* gin, const float* bn_y,
                                       const float* bn_scale, const float* bn_shift, float* bn_partials, float* scratch,
                                       float* dw, int B, int H, int W, void* stream);
/* The same for the LAST 16 -> 16 layer, ub3.convbloc.bloc.3, whose output feeds outlay (Conv2d(16, 1, 3, padding_mode="replicate"),
 * model.py:605): the upstream gradient g = outlay^T(dsr) is never stored -- each staged pixel recomputes it from a 20x20 tile of
 * dsr = d loss / d sr ([B][H][W] fp32, in every storage mode) and w_out ([1][16][3][3]), replicate-padding adjoint included; the rest
 * (y, coef_f, border, gin, dw, the BatchNorm sums of the layer below) as sifsr_conv3x3_bwd16 with y != NULL and no addend.
 * Replaces tail_bwd_apply's write + re-read of dL/dy (2 tensor passes of 16 x 256^2 x batch). */
SIFSR_API int sifsr_conv3x3_bwd16_tail(const float* x, const float* x_scale, const float* x_shift, const float* dsr,
                                       const float* w_out, const float* y, const float* coef_f, float* border,
                                       const float* wdgrad, const float* wwd, float* gin, const float* bn_y,
                                       const float* bn_scale, const float* bn_shift, float* bn_partials, float* scratch,
                                       float* dw, int B, int H, int W, void* stream);
/* bf16 form of the weight gradient (BASELINE.json config 5): src* and dy are NHWC bf16 tensors; the staged x and dy are
 * rounded to bf16 when read from LDS and contracted 16 pixels at a time with v_mfma_f32_16x16x16_bf16; fp32 accumulation,
 * slabs and dw. */
SIFSR_API int sifsr_conv3x3_wgrad_bf16(const float* src0, int C0, const float* scale0, const float* shift0,
                                  const float* src1, int C1, const float* scale1, const float* shift1,
                                  const float* dy, int cout, float* scratch, int nblk, float* dw, int B, int H, int W,
                                  void* stream);
/* inbloc.bloc.0, Conv2d(2,16): x NCHW -> y NHWC (model.py:596) */
/* Activation storage of the single-operator entry points below that are not 3x3 convolutions (the thin first / last convs,
 * BatchNorm reductions, pooling / upsampling and their adjoints, the fused tail): on = 1 makes the calling thread's later calls
 * treat their activation tensors (y, g, dy, pooled / upsampled tensors; NOT x, sr, dsr, parameters, statistics) as NHWC bf16,
 * as ModelB_2's bf16 mode does; 0 (default) = fp32.  sifsr_model_forward_ex / _backward_ex choose it from `compute`. */
SIFSR_API int sifsr_set_op_storage_bf16(int on);
/* stat_partials: NULL or [sifsr_conv_in_stat_blocks()][16][2] per-workgroup (sum, sumsq) of y */
SIFSR_API int sifsr_conv_in_stat_blocks(int B, int H, int W);
SIFSR_API int sifsr_conv_in_fwd(const float* x, const float* w, float* y, float* stat_partials, int B, int H, int W, void* stream);
SIFSR_API int sifsr_conv_in_wgrad(const float* x, const float* dy, float* scratch, int nblk, float* dw, int B, int H, int W, void* stream);
/* outlay, Conv2d(16,1)+bias: y NHWC (BN+ReLU folded) -> sr NCHW (model.py:605); dwb = [144 weight | 1 bias] */
SIFSR_API int sifsr_conv_out_fwd(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                                 float* sr, int B, int H, int W, void* stream);
SIFSR_API int sifsr_conv_out_dgrad(const float* dsr, const float* w, float* g, int B, int H, int W, void* stream);
SIFSR_API int sifsr_conv_out_wgrad(const float* y, const float* scale, const float* shift, const float* dsr, float* scratch,
                                   int nblk, float* dwb, int B, int H, int W, void* stream);
/* Fused tail of the backward pass (model.py:605 <- :139-141): outlay dW|db (145 floats -> dwb), and the BatchNorm+ReLU
 * backward of the layer feeding outlay with the outlay input gradient recomputed from dsr instead of stored:
 * y = raw conv output (B,H,W,16) -> dgamma, dbeta, dy.  scratch: >= 64 + nblk*(145+32) floats; coef: 48 float64. */
SIFSR_API int sifsr_conv_out_bn_relu_bwd(const float* y, const float* scale, const float* shift, const float* mean,
                                         const float* invstd, const float* dsr, const float* w, float* scratch, int nblk,
                                         float* dwb, float* dgamma, float* dbeta, double* coef, float* dy, int B, int H,
                                         int W, void* stream);
/* Fused head of the backward pass (model.py:596 / :135-137): BatchNorm+ReLU backward of inbloc.bloc.0-2 + the input
 * conv's weight gradient; dy = dL/dy is formed inside the wgrad staging and never stored (the model input needs no
 * gradient).  g = dL/d relu(bn(y)).  scratch: >= nblk*288 floats, nblk <= 1024; coef: 48 float64. */
SIFSR_API int sifsr_conv_in_bn_relu_bwd(const float* x, const float* g, const float* y, const float* scale,
                                        const float* shift, const float* mean, const float* invstd, float* scratch,
                                        int nblk, float* dw, float* dgamma, float* dbeta, double* coef, int B, int H, int W,
                                        void* stream);
/* The weight gradient of inbloc.bloc.0 (Conv2d(2, 16, 3, bias=False) + BatchNorm + ReLU, model.py:596) WITHOUT reading (g, y): what
 * ModelB_2's backward runs since round 3 where the fused 16 -> 16 kernel serves inbloc.bloc.3.  dL/dy = sd*dz + k1*y + k0 is affine in
 * dz = g*[y*scale+shift > 0] and y = W p (p = the 18 replicate-padded inputs under a pixel), so
 *   dW = sd * D + k1 * (W G) + k0 * X,   D = sum dz p^T (one read of dz),  G = sum p p^T and X = sum p (functions of x alone).
 * x (B,2,H,W) NCHW; dz (B,H,W,16) NHWC (bf16 after sifsr_set_op_storage_bf16(1)); w (16,2,3,3); coef = the 48 float64
 * [sd | k1 | k0] that sifsr_conv_in_bn_relu_bwd / sifsr_bn_relu_bwd_coef return; scratch: _scratch_floats(nblk) floats; dw (16,2,3,3). */
SIFSR_API size_t sifsr_conv_in_bwd_linear_scratch_floats(int nblk);
SIFSR_API int sifsr_conv_in_bwd_linear(const float* x, const float* dz, const float* w, const double* coef, float* scratch,
                                       int nblk, float* dw, int B, int H, int W, void* stream);

/* ---- BatchNorm2d (model.py:136,139,508; eps 1e-5, momentum 0.1) ------------------------------ */
SIFSR_API int sifsr_bn_finalize(const float* stat_partials, int nblk, int C, double count, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                float* mean, float* invstd, float* scale, float* shift, void* stream);
/* g = dL/d relu(bn(y)) -> dgamma, dbeta, dy = dL/dy; partials: >= nblk*C*2 floats; coef: 3*C float64 scratch.
 * gpool != NULL: the activation also feeds AvgPool2d(2,2) (model.py:504); gpool = gradient of the pooled tensor
 * (B,H/2,W/2,C) and g_eff = g + 0.25*gpool[y/2][x/2] is formed on the fly (npix = B*H*W); else H, W are ignored. */
SIFSR_API int sifsr_bn_relu_bwd(const float* g, const float* y, const float* scale, const float* shift, const float* mean,
                                const float* invstd, int C, size_t npix, float* partials, int nblk, float* dgamma,
                                float* dbeta, double* coef, float* dy, const float* gpool, int H, int W, void* stream);
/* The statistics half alone, as the model's backward uses it: dgamma, dbeta and coef_f = 4*C floats [sc | sh | k1 | k0]
 * with which sifsr_conv3x3_dgrad_fused / _wgrad_fused form dL/dy = sc*g*[z>0] + k1*z + k0 (z = y*sc + sh) themselves; beta =
 * the BatchNorm bias.  gpool != NULL: g is completed IN PLACE to g + 0.25*gpool[y/2][x/2] (the AvgPool2d adjoint). */
SIFSR_API int sifsr_bn_relu_bwd_coef(float* g, const float* y, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, const float* beta, int C, size_t npix, float* partials, int nblk,
                                     float* dgamma, float* dbeta, double* coef, float* coef_f, const float* gpool, int H, int W,
                                     void* stream);

/* ---- resampling (NHWC; scale == NULL: input used as stored) ---------------------------------- */
SIFSR_API int sifsr_bnrelu_pool2(const float* y, const float* scale, const float* shift, float* out, int B, int H, int W, int C, void* stream); /* AvgPool2d(2,2), model.py:504 */
SIFSR_API int sifsr_bnrelu_add(const float* p, const float* y, const float* scale, const float* shift, float* out, int C, size_t npix, void* stream); /* model.py:311-312 */
SIFSR_API int sifsr_bnrelu_up2x(const float* y, const float* scale, const float* shift, float* out, int B, int Hin, int Win, int C, void* stream); /* Upsample(x2,bilinear,align_corners=True), model.py:207 */
SIFSR_API int sifsr_pool2_bwd(const float* gp, float* g, int B, int H, int W, int C, int accumulate, void* stream);
SIFSR_API int sifsr_up2x_bwd(const float* gu, float* g, int B, int Hin, int Win, int C, void* stream);
/* The same adjoint when g is the complete gradient w.r.t. relu(bn(y)) of the low-resolution layer (the decoder inputs of
 * ModelB_2): also leaves that layer's BatchNorm-backward sums, one row [C][2] = (sum dz, sum dz*y) per workgroup,
 * dz = g*[y*scale + shift > 0], in partials (sifsr_up2x_bwd_stat_rows() rows; 0 = shape not served, C in {16, 32, 64}). */
SIFSR_API int sifsr_up2x_bwd_stat_rows(int B, int Hin, int Win, int C);
SIFSR_API int sifsr_up2x_bwd_bn_sums(const float* gu, float* g, int B, int Hin, int Win, int C, const float* y,
                                     const float* scale, const float* shift, float* partials, void* stream);

/* ---- SIF loss operators on (B,1,H,W) images --------------------------------------------------- */
/* taps9: 9 HOST floats, the separable factor of generate_psf_kernel (utils.py:1615-1639) */
SIFSR_API int sifsr_gauss9_reflect_fwd(const float* x, const float* taps9, float* out, int B, int H, int W, void* stream);   /* get_output_ftm, utils.py:1833-1860 */
SIFSR_API int sifsr_gauss9_reflect_bwd(const float* g, const float* taps9, float* gx, int B, int H, int W, void* stream);
SIFSR_API int sifsr_gauss9_decimate4_fwd(const float* x, const float* taps9, float* out_lr, int B, int H, int W, void* stream); /* downscale_LST_SR_to_LR, utils.py:1671-1706 */
SIFSR_API int sifsr_gauss9_decimate4_bwd(const float* g_lr, const float* taps9, float* gx, int B, int H, int W, void* stream);
SIFSR_API int sifsr_sobel4_fwd(const float* x, float* out_b4hw, int B, int H, int W, void* stream);   /* train_model_B_predef_filters.py:120-128 */
SIFSR_API int sifsr_sobel4_bwd(const float* g_b4hw, float* gx, int B, int H, int W, void* stream);
SIFSR_API int sifsr_huber_partial_blocks(size_t n);
/* out[0] = mean huber_1(a - bscale*b), nn.HuberLoss(delta=1), train_model_B_gradFTM.py:454 */
SIFSR_API int sifsr_huber_fwd(const float* a, const float* b, float bscale, size_t n, float* partials, float* out, void* stream);
SIFSR_API int sifsr_huber_bwd(const float* a, const float* b, float bscale, const float* gout, size_t n, float* ga, void* stream);
SIFSR_API size_t sifsr_sif_loss_workspace_bytes(int kind, int B, int H, int W);
/* kind 2: SR2 loss, train_model_B_gradFTM.py:99-117; kind 1: SR1, train_model_B_predef_filters.py:111-133.
 * losses3 (device) = {ds_loss, percep_loss, alpha*ds + (1-alpha)*percep}; dsr (may be NULL) = d loss / d sr. */
SIFSR_API int sifsr_sif_loss(int kind, const float* sr, const float* lst, const float* ndvi, int B, int H, int W,
                             float mean, float std, float alpha, float gamma, const float* taps_ds9,
                             const float* taps_ftm9, void* workspace, size_t workspace_bytes, float* losses3,
                             float* dsr, void* stream);

/* ---- optimizer: torch.optim.Adam on the flat buffer (train_model_B_gradFTM.py:453,121) -------- */
SIFSR_API int sifsr_adam_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream);
/* The same update with the step count in device memory (int64, incremented by the call) and a 2-float device scratch for
 * the bias-correction coefficients: nothing depends on a host value that changes per step, so a whole training step
 * (forward, loss, backward, Adam) can be captured into a hipGraph and replayed. */
SIFSR_API int sifsr_adam_flat_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, long long* step_dev, float* coef2,
                                  float grad_scale, void* stream);

/* ---- the step before the path and the metrics after it --------------------
 * sifsr_tiles_prepare: per tile, what dataset.py:134-142 / predict.py:84-100 do on the host: z-score of the
 * win x win LST tile, bicubic x4 (us.upsampling = cv2.resize INTER_CUBIC, utils.py:163-180; A = -0.75, half-pixel
 * centres, edge clamp), NDVI optional clip to [-1,1] + z-score, torch.cat((lst_up, ndvi), 1) -> x (T,2,4win,4win).
 *   granule = 1: lst is ONE raster (lst_h, lst_w), ndvi (4 lst_h, 4 lst_w); tiles (ty,tx) at (win*ty, win*tx)
 *   granule = 0: lst (T,1,win,win), ndvi (T,1,4win,4win) with T = tiles_y*tiles_x (lst_h, lst_w ignored)
 * Pass mean 0 / std 1 for already normalised inputs.  win % 4 == 0, win <= 64.
 * sifsr_tiles_paste: predict.py:101-103, out[4win*ty + Y][4win*tx + X] = sr*std_lst + mean_lst, out (4 lst_h, 4 lst_w). */
SIFSR_API int sifsr_tiles_prepare(const float* lst, const float* ndvi, float* x, int tiles_y, int tiles_x, int win, int lst_h,
                                  int lst_w, int granule, float mean_lst, float std_lst, float mean_ndvi, float std_ndvi,
                                  int clip_ndvi, void* stream);
SIFSR_API int sifsr_tiles_paste(const float* sr, float* out, int tiles_y, int tiles_x, int win, int lst_w, float mean_lst,
                                float std_lst, void* stream);
/* us.psnr_skimage / us.ssim_skimage (utils.py:548-578) of (B,1,H,W) batches: out2[0] = mean_i PSNR_i,
 * out2[1] = mean_i SSIM_i with scikit-image 0.22 defaults (7x7 uniform window, sample covariance, K1 0.01, K2 0.03)
 * and data_range = max - min of the whole TARGET batch, as the reference passes it. */
SIFSR_API size_t sifsr_psnr_ssim_scratch_bytes(int B, int H, int W);
SIFSR_API int sifsr_psnr_ssim(const float* pred, const float* targ, int B, int H, int W, void* scratch, size_t scratch_bytes,
                              float* out2, void* stream);

/* us.downsampling (utils.py:183-213), the 'norm-L4' decimation used by the scale-invariance baseline's dataset
 * (dataset.py:258): out[b][i][j] = (mean over the 4x4 block of x^4)^(1/4); x (B,H,W), H and W multiples of 4. */
SIFSR_API int sifsr_l4pool4(const float* x, float* out, int B, int H, int W, void* stream);

/* Fourier-domain evaluation of (B,H,W) images, H and W powers of two (4..2048):
 *   mag      (optional, (B,H,W))  = np.fft.fftshift(np.abs(sp.fft.fft2(img)))            compare_methods.py:312-324
 *   spectrum (optional, (B,nr+1)) = us.compute_2D_attenuation_spectra(mag), nr = min(H/2, W/2) - 1   utils.py:598-636
 * hand-written radix-2 FFT in LDS, float64 throughout (ring means in float64, stored as float32 dB values). */
SIFSR_API size_t sifsr_fft2_attenuation_scratch_bytes(int B, int H, int W);
SIFSR_API int sifsr_fft2_attenuation(const float* img, int B, int H, int W, void* scratch, size_t scratch_bytes, float* mag,
                                     float* spectrum, void* stream);

/* ---- schedule knob -----------------------------------------------------------------------------
 * sifsr_model_backward runs the 16 MFMA weight gradients on a second, lower-priority stream owned by the library
 * (forked from / joined to `stream` with events, so the call is still stream-ordered and hipGraph-capturable).
 * on = 0 keeps everything on `stream`; on = 1 forces the second stream; -1 restores the default (environment variable
 * SIFSR_WGRAD_STREAM, else on).  The results are bit-identical either way.  No reference counterpart. */
SIFSR_API int sifsr_set_wgrad_stream(int on);

/* ---- measurement hook (bench.py roofline) ------------------------------------------------------
 * Time ONE kernel of the model schedule with HIP events on its launch stream, inside normal steps:
 * layer = row of sifsr_layer_table, phase 1 = forward conv, 2 = dgrad, 3 = wgrad; layer < 0 disables.
 * sifsr_profile_read synchronises the recorded events and returns their summed duration and count. */
SIFSR_API int sifsr_profile_select(int layer, int phase);
SIFSR_API int sifsr_profile_read(float* total_ms, int* count);
/* Several kernels side by side: sifsr_profile_select(l, p) makes slot 0, each sifsr_profile_add one more (returns the
 * slot, < 0 when the 8 slots are used); all event pairs are created by these calls, outside the caller's timed region. */
SIFSR_API int sifsr_profile_add(int layer, int phase);
SIFSR_API int sifsr_profile_read_slot(int slot, float* total_ms, int* count);

#endif /* SIFSR_HIP_H */
