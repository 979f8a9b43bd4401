// Model-level schedule (internal; the C-ABI is include/sifsr_hip.h).
#pragma once
#include "common.h"
#include "conv.h"
#include "edge_conv.h"
#include "loss.h"

// All offsets in floats from the workspace base; every region is 256-byte aligned.
struct WsLayout {
  size_t npix[4];                         // B*H*W at the four resolution levels
  size_t mean, invstd, scale, shift;      // per-channel vectors, all 17 BN layers back to back
  size_t wfwd, wdg;                       // fragment-ordered conv weights (forward / dgrad)
  size_t wwf, wwd;                        // the same in the Winograd F(2x2,3x3) domain (16 instead of 9 values per weight pair)
  size_t y[SIFSR_NUM_BN_LAYERS];          // raw conv outputs (pre-BN), NHWC
  size_t P[3], R[3], U[3];                // pooled inputs, residual sums, upsampled decoder inputs
  size_t partials;                        // BN statistic partials (scratch)
  size_t partials_cap;                    // its size in floats
  size_t fwd_end;
  size_t coef;                            // BN-backward affine coefficients (3 x C float64, current layer: fused head / tail)
  size_t coef_f;                          // bn_bwd4 coefficients of every layer, [4][C] fp32 at 4 * ch_off (kept for the whole backward:
                                          // the weight gradients on the second stream read them long after the chain moved on)
  size_t dy_border;                       // dL/dy on the image border of the layer being processed (NHWC indexing, border pixels only)
  size_t bpart;                           // BN-backward sums from the dgrad border kernel ([wave][16][2])
  size_t g[SIFSR_NUM_BN_LAYERS];          // grad w.r.t. relu(bn(y_l)); dL/dy_l is formed from (g_l, y_l) by its consumers (bn_bwd4),
                                          // stored only for ub3.convbloc.bloc.3 (fused tail)
  size_t dyB[3];                          // (unused since dL/dy is no longer stored; = dy_border, kept for the regions table)
  size_t gP[3], gU[3];
  size_t slabs;                           // thin-layer per-workgroup partials (scratch)
  size_t gram;                            // Gram matrix / sums of the input patches for the first layer's weight gradient (edge_conv.hip)
  size_t slab_l[SIFSR_NUM_BN_LAYERS];     // wgrad per-workgroup partial dW of every MFMA layer (reduced together at the end)
  size_t slab_cap[SIFSR_NUM_BN_LAYERS];   // ... and the size of each region in floats (checked against the grid actually launched)
  size_t total;
};

int sifsr_layout(int B, int H, int W, int training, WsLayout* out);

int sifsr_engine_forward(const float* x, float* sr, const float* params, float* running, long long* nbt, float* ws,
                         size_t ws_floats, int B, int H, int W, int training, float momentum, float eps, hipStream_t s,
                         int bf16 = 0);
int sifsr_engine_backward(const float* x, const float* dsr, const float* params, float* grads, float* ws,
                          size_t ws_floats, int B, int H, int W, hipStream_t s, int bf16 = 0);
int sifsr_engine_set_wgrad_stream(int on);   // 1 / 0: weight gradients on their own stream or not; -1: SIFSR_WGRAD_STREAM / default (on)
int sifsr_engine_profile_select(int layer, int phase);
int sifsr_engine_profile_add(int layer, int phase);
int sifsr_engine_profile_read(int slot, float* total_ms, int* count);
