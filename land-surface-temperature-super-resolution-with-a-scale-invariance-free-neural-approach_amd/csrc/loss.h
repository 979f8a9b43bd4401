// Launchers of the SIF loss kernels (internal; the C-ABI is include/sifsr_hip.h).
#pragma once
#include "common.h"

int launch_blur_fwd(const float* x, const float* taps9, float* out, int B, int H, int W, hipStream_t s);
int launch_blur_bwd(const float* g, const float* taps9, float* out, int B, int H, int W, hipStream_t s);
int launch_blurdec_fwd(const float* x, const float* taps9, float* out, int B, int H, int W, hipStream_t s);
int launch_blurdec_bwd(const float* glr, const float* taps9, float* out, int B, int H, int W, hipStream_t s);
int launch_sobel_fwd(const float* x, float* out, int B, int H, int W, hipStream_t s);
int launch_sobel_bwd(const float* g, float* out, int B, int H, int W, hipStream_t s);
int huber_partial_blocks(size_t n);
int launch_huber_fwd(const float* a, const float* b, float bscale, size_t n, float* partials, float* out, hipStream_t s);
int launch_huber_bwd(const float* a, const float* b, float bscale, const float* gout, size_t n, float* ga, hipStream_t s);
size_t sif_loss_workspace_floats(int kind, int B, int H, int W);
int launch_sif_loss(int kind, const float* sr, const float* lst, const float* ndvi, int B, int H, int W, float mean,
                    float std, float alpha, float gamma, const float* taps_ds, const float* taps_ftm, float* ws,
                    float* losses3, float* dsr, hipStream_t s);
