// Launchers of the VALU edge convolutions, BatchNorm, resampling, loss and optimizer kernels
// (internal; the C-ABI is include/sifsr_hip.h).
#pragma once
#include "common.h"

// ---- edge_conv.hip ----
int conv_in_fwd_blocks(int B, int H, int W);   // rows of `partials` ([blocks][16][2]) launch_conv_in_fwd writes
int launch_conv_in_fwd(const float* x, const float* w, float* y, float* partials, int B, int H, int W, hipStream_t s);
int launch_conv_in_wgrad(const float* x, const float* dy, float* partials, int nblk, float* dw, int B, int H, int W, hipStream_t s);
int launch_conv_out_fwd(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                        float* out, int B, int H, int W, hipStream_t s);
int launch_conv_out_dgrad(const float* dsr, const float* w, float* g, int B, int H, int W, hipStream_t s);
int launch_conv_out_wgrad(const float* y, const float* scale, const float* shift, const float* dsr, float* partials,
                          int nblk, float* dw, float* db, int B, int H, int W, hipStream_t s);
int launch_conv_in_wgrad_fused(const float* x, const float* g, const float* y, const float* scale, const float* shift,
                               const double* coef, float* partials, int nblk, float* dw, int B, int H, int W, hipStream_t s);
// "head" without a second read of (g, y): dW = sd * D + k1 * (W G) + k0 * X (edge_conv.hip); gram scratch holds G | X as float64
size_t conv_in_gram_scratch_floats();
int launch_conv_in_gram(const float* x, float* scratch, int B, int H, int W, hipStream_t s);
const double* conv_in_gram_result(const float* scratch);
int launch_conv_in_dz_wgrad(const float* x, const float* dz, float* partials, int nblk, int B, int H, int W, hipStream_t s);
int launch_conv_in_dw_combine(const float* partials, int nblk, const double* gram, const float* w, const double* coef, float* dw,
                              hipStream_t s);
// ---- fused_edges.hip ---- (outlay backward + BatchNorm/ReLU backward of its producer)
int launch_tail_bwd_reduce(const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                           const float* dsr, const float* w, float* wpart, float* bnpart, int nblk, int B, int H, int W,
                           hipStream_t s);
int launch_tail_bwd_apply(const float* y, const float* scale, const float* shift, const double* coef, const float* dsr,
                          const float* w, float* dy, int B, int H, int W, hipStream_t s);
int launch_sum_partials(const float* partials, int nblk, int n, float* out, hipStream_t s);

// ---- bn.hip ----
// training: per-workgroup (sum, sumsq) partials -> batch mean / invstd, folded scale/shift, running-stat update
int launch_bn_finalize(const float* partials, int nblk, int C, double count, const float* gamma, const float* beta,
                       float* run_mean, float* run_var, float momentum, float eps, float* mean, float* invstd,
                       float* scale, float* shift, hipStream_t s);
// eval: scale/shift from the running statistics of all 17 layers in one launch
int launch_bn_eval_coeffs(const float* params, const float* running, float eps, float* scale, float* shift, hipStream_t s);
// gp != nullptr: the AvgPool2d(2,2) adjoint of the half-resolution gradient gp (image H x W at full resolution) is
// added to g on the fly in both passes:  g_eff = g + 0.25 * gp[y/2][x/2]
// g_out (== g, only with gp): the completed gradient g_eff is also written back in place, for consumers that form
// dL/dy from (g, y) themselves (bn_bwd4, common.h) instead of reading the output of launch_bn_bwd_apply.
int launch_bn_bwd_reduce(const float* g, const float* y, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int C, size_t npix, float* partials, int nblk, hipStream_t s,
                         const float* gp = nullptr, int H = 0, int W = 0, float* g_out = nullptr);
// coef: 3 x C float64 (scale, k1, k0 of dy = scale*dz + k1*y + k0; the fused head / tail kernels and launch_bn_bwd_apply);
// coef_f (optional, needs shift and beta): 4 x C fp32 [sc | sh | k1 | k0] of bn_bwd4 (dy = sc*dz + k1*z + k0 on z = y*sc + sh)
int launch_bn_bwd_finalize(const float* partials, int nblk, int C, double count, const float* scale, const float* mean,
                           const float* invstd, float* dgamma, float* dbeta, double* coef, hipStream_t s,
                           const float* shift = nullptr, const float* beta = nullptr, float* coef_f = nullptr,
                           const float* xs_partials = nullptr, int xs_n = 0, float* xs_out = nullptr);   // xs_*: a rider, xs_out[e] =
                           // the column sums of xs_partials[nblk][xs_n] (same row count) -- one launch less on the chain
int launch_bn_bwd_finalize2(const float* pa, int na, const float* pb, int nb, int C, double count, const float* scale,
                            const float* mean, const float* invstd, float* dgamma, float* dbeta, double* coef, hipStream_t s,
                            const float* shift = nullptr, const float* beta = nullptr, float* coef_f = nullptr);
int launch_bn_bwd_apply(const float* g, const float* y, const float* scale, const float* shift, const double* coef,
                        int C, size_t npix, float* dy, hipStream_t s, const float* gp = nullptr, int H = 0, int W = 0);
int launch_nbt_increment(long long* nbt, int n, hipStream_t s);

// ---- resample.hip ---- (all NHWC, C % 4 == 0; scale == nullptr => input used as stored)
int launch_bnrelu_pool2(const float* y, const float* scale, const float* shift, float* out, int B, int H, int W, int C, hipStream_t s);
int launch_bnrelu_add(const float* p, const float* y, const float* scale, const float* shift, float* out, int C, size_t npix, hipStream_t s);
int launch_bnrelu_up2x(const float* y, const float* scale, const float* shift, float* out, int B, int Hin, int Win, int C, hipStream_t s);
int launch_pool2_bwd(const float* gp, float* g, int B, int H, int W, int C, int accumulate, hipStream_t s);
// bn_partials != nullptr: also the BatchNorm-backward sums (sum dz, sum dz*y) of the low-resolution layer whose gradient g is,
// one row of [C][2] per workgroup, up2x_bwd_stat_rows() rows (0: the tiled kernel does not serve this shape)
int launch_up2x_bwd(const float* gu, float* g, int B, int Hin, int Win, int C, hipStream_t s, const float* bn_y = nullptr,
                    const float* bn_scale = nullptr, const float* bn_shift = nullptr, float* bn_partials = nullptr);
int up2x_bwd_stat_rows(int B, int Hin, int Win, int C);

// ---- adam.hip ----
int launch_adam_flat(float* p, const float* g, float* m, float* v, int n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, hipStream_t s);

int launch_adam_flat_dev(float* p, const float* g, float* m, float* v, int n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, long long* step_dev, float* coef2, float grad_scale, hipStream_t s);

// ---- pipeline.hip ---- (input pipeline before the model, metrics after it: SURVEY.md §8 f2 / f1)
int launch_tiles_prepare(const float* lst, const float* ndvi, float* x, int T, int tiles_x, int win, long long lst_step_y,
                         long long lst_step_x, int lst_row, long long ndvi_step_y, long long ndvi_step_x, int ndvi_row,
                         float mean_lst, float std_lst, float mean_ndvi, float std_ndvi, int clip_ndvi, hipStream_t s);
int launch_tiles_paste(const float* sr, float* out, int T, int tiles_x, int hr, long long out_row, float mean, float std,
                       hipStream_t s);
int launch_l4pool4(const float* x, float* out, int B, int H, int W, hipStream_t s);
size_t psnr_ssim_scratch_bytes(int B, int H, int W);
int launch_psnr_ssim(const float* pred, const float* targ, int B, int H, int W, void* scratch, float* out2, hipStream_t s);

// ---- fourier.hip ---- (Fourier-domain evaluation)
size_t fourier_scratch_bytes(int B, int H, int W);
int launch_fft2_attenuation(const float* img, int B, int H, int W, void* scratch, float* mag, float* spectrum, hipStream_t s);
