// Argument blocks of the 3x3 convolution kernels (internal; the C-ABI is include/sifsr_hip.h).
#pragma once
#include "common.h"
#include "diag.h"

// One input operand of a conv: channels [coff, coff + 16*nq) of an NHWC tensor with C channels.
// scale/shift != nullptr: the tensor is a RAW conv output and relu(x*scale[c]+shift[c]) (the folded
// BatchNorm + ReLU of the producing layer) is applied while staging; nullptr: used as stored.
struct ConvSrc {
  const float* ptr;
  const float* scale;
  const float* shift;
  int C;
  int coff;
  int nq;   // number of 16-channel blocks taken from this operand
};

struct ConvDst {
  float* ptr;
  int C;
  int coff;
};

struct ConvArgs {
  ConvSrc src[2];       // channel concat [src0 | src1] (torch.cat([up, skip], 1), model.py:247)
  ConvDst dst[2];       // output channel blocks [0, dst_split) -> dst[0], the rest -> dst[1]
  const float* wpack;   // fragment-ordered weights, see pack_weights_kernel
  const float* wpack_wino = nullptr;   // optional: the Winograd-domain pack (pack_wino_kernel) of the same weights; used instead
                                       // of wpack where conv3x3_use_wino() holds (fp32, cout <= 64, even H and W)
  const float* addend;  // optional tensor added to dst[0] in the epilogue (residual gradient), C = addC
  float* stat_partials; // optional [conv3x3_grid_blocks()][Cout][2] per-workgroup (sum, sum of squares) of the output
  int addC;
  int dst_split;        // in 16-channel blocks
  int B, H, W;
  int NQ;               // 16-channel blocks of the contraction (Cin / 16)
  int bf16 = 0;         // 1: wpack is the bf16 fragment pack, operands are rounded to bf16 (fp32 accumulate/outputs)
  // dgrad of a 16-channel layer whose output IS the gradient w.r.t. relu(bn(y)) of the previous layer: the epilogue
  // also produces that layer's BatchNorm-backward sums (dz = g*[y*scale+shift > 0]; sum dz, sum dz*y per channel) into
  // stat_partials, saving the separate reduce pass over (g, y).  bn_y = that layer's raw conv output (C = 16).
  const float* bn_y = nullptr;
  const float* bn_scale = nullptr;
  const float* bn_shift = nullptr;
  // dgrad with the BatchNorm+ReLU backward of ITS OWN layer applied while staging (bn_bwd4, common.h): src[0].ptr is
  // g = dL/d relu(bn(y)) (C = the layer's cout), bw_y the layer's raw conv output, bw_coef = [sc | sh | k1 | k0], C floats
  // each (bn_bwd_finalize*).  dL/dy is then never stored -- except on the image border, where the producers also write it
  // to bw_border (same NHWC indexing, only border pixels are touched) for the replicate-border fold kernel.
  const float* bw_y = nullptr;
  const float* bw_coef = nullptr;
  float* bw_border = nullptr;
};

// conv3x3, stride 1, NHWC fp32, MFMA implicit GEMM.  zero_pad = 0: replicate padding (forward,
// nn.Conv2d(padding_mode='replicate'), model.py:135); zero_pad = 1: zero padding (the interior part
// of the transposed conv used by dgrad; the replicate-border fold is dgrad_border_fix).
int launch_conv3x3_mfma(const ConvArgs& a, int cout, int zero_pad, hipStream_t s);
bool conv3x3_use_wino(const ConvArgs& a, int cout);
// conv_wino8.hip: Winograd kernels for 32 / 64 output channels whose eight waves all stage and contract (a.wpack = the Winograd pack)
bool conv3x3_wino8_applies(int nb, int nq);
int launch_conv3x3_wino8(const ConvArgs& a, int nb, int zero_pad, bool dyf, int grid, int ntiles, int lgx, int lgy, hipStream_t s);
int conv3x3_wino_kind(const ConvArgs& a, int cout, int zero_pad);   // 0 tap-domain, 1 / 2 Winograd variants (residency differs)
// workgroups launched == stat_partials rows written; wino = conv3x3_wino_kind() of the same call
int conv3x3_grid_blocks(int B, int H, int W, int cout, int wino = 0);

struct WgradArgs {
  ConvSrc src[2];      // the conv's forward input (same description as in the forward call)
  const float* dy;     // gradient w.r.t. the raw conv output, NHWC, Cout channels
  float* slabs;        // [nblk][chunks][9*cin_chunk*Cout] per-block partial dW in fragment order
  int B, H, W;
  int NQ;              // Cin / 16 (total)
  int ntiles;          // B * ceil(H/8) * ceil(W/16)
  int bf16 = 0;        // 1: operands rounded to bf16 when read from LDS, v_mfma_f32_16x16x16_bf16 (config 5)
  // dy_y != nullptr: `dy` is g = dL/d relu(bn(y)) and dL/dy is formed while staging (bn_bwd4): dy_y = the layer's raw
  // conv output, dy_coef = [sc | sh | k1 | k0] (Cout floats each)
  const float* dy_y = nullptr;
  const float* dy_coef = nullptr;
};
// returns the number of slab blocks used through *nblk_out
int launch_conv3x3_wgrad(const WgradArgs& a, int cin, int cout, int nblk, hipStream_t s);
int wgrad_nbi_chunk(const WgradArgs& a, int cin);   // 16-channel blocks per Cin chunk (blockIdx.y)
int launch_wgrad_reduce(const float* slabs, int nblk, int cin, int cout, int nbi_chunk, float* dw_oihw, hipStream_t s);
size_t wgrad_slab_floats(int cin, int cout);   // floats per block
// All layers' slab reductions in ONE launch (end of the backward pass) instead of one latency-bound launch per layer.
struct WgradReduceJob { size_t slab_off; int nblk, cin, cout, nbi_chunk, w_off; };
int launch_wgrad_reduce_batched(const float* ws, const WgradReduceJob* jobs, int njobs, float* grads, hipStream_t s);

// Winograd F(3x3, 2x2) form of the weight gradient (conv_wgrad_wino.hip): same arguments and grid as launch_conv3x3_wgrad,
// slabs of 16 * cin * cout floats per block; the finish call reduces the slabs (float64) and applies the output transform.
bool conv3x3_wgrad_use_wino(const WgradArgs& a, int cin, int cout);
int wgrad_wino_nbi_chunk(const WgradArgs& a, int cin);   // its Cin chunking (a 32-channel chunk may span the two sources)
int launch_conv3x3_wgrad_wino(const WgradArgs& a, int cin, int cout, int nblk, hipStream_t s);
// Output transform of the Winograd F(3x3, 2x2) weight gradient, dW = A^T M A per (cout, cin) pair with A^T = [[1, .5, .5, 0],
// [0, .5, -.5, 0], [0, .5, .5, 1]]: lane-local (a lane holds the same pairs for all 16 xi), applied by the producing kernels to their
// accumulators before the slab is written -- slabs are then 9 instead of 16 values per pair, in the tap-domain kernel's layout
// ([chunk][nbo][nbi][tap][lane][4]), and the tap-domain slab reduction (launch_wgrad_reduce*) serves both.
template <typename V4, typename GET>
static __device__ __forceinline__ void wino_wgrad_taps(GET&& m, V4 (&tap)[9]) {   // m(xi) -> V4 (f32x4), xi = 4 u + v
  V4 t[3][4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const V4 m1 = m(4 + v), m2 = m(8 + v);
    t[0][v] = m(v) + 0.5f * (m1 + m2);
    t[1][v] = 0.5f * (m1 - m2);
    t[2][v] = 0.5f * (m1 + m2) + m(12 + v);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    tap[3 * i + 0] = t[i][0] + 0.5f * (t[i][1] + t[i][2]);
    tap[3 * i + 1] = 0.5f * (t[i][1] - t[i][2]);
    tap[3 * i + 2] = 0.5f * (t[i][1] + t[i][2]) + t[i][3];
  }
}

// Input gradient AND weight gradient of a 16 -> 16 channel layer from one read of its operands (conv_bwd16.hip): H, W multiples
// of 16, every tensor NHWC with exactly 16 channels.
struct Bwd16Args {
  const float* x;                  // the layer's forward input: a raw conv output (x_scale / x_shift = the folded BatchNorm of the
  const float* x_scale = nullptr;  //   layer below, relu(x * scale + shift) applied while staging) or, with nullptr, a stored tensor
  const float* x_shift = nullptr;
  const float* g = nullptr;        // dL/d relu(bn(y)) of this layer -- or dL/dy itself when y == nullptr
  const float* tail_dsr = nullptr; // instead of g ("tail", ub3.convbloc.bloc.3): g is the input gradient of outlay (Conv2d 16 -> 1, replicate
  const float* tail_w = nullptr;   //   padding), recomputed while staging from d loss / d sr [B][H][W] (fp32) and the outlay weight [1][16][3][3]
  const float* y = nullptr;        // this layer's raw conv output: dL/dy is formed while staging (bn_bwd4) ...
  const float* coef = nullptr;     // ... from [sc | sh | k1 | k0], 16 floats each (bn_bwd_finalize*)
  float* dy_border = nullptr;      // with y: dL/dy of the image-border pixels goes here (NHWC indexing) for the border-fold kernel
  const float* wpack_wino;         // Winograd-domain input-gradient weights of the layer (pack_wino_kernel, wwd), 16 * 256 floats
  float* gin;                      // out: gradient w.r.t. the layer's input (zero-padded part; the border fold is a separate kernel)
  const float* addend = nullptr;   // optional tensor added to gin (residual gradient)
  const float* bn_y = nullptr;     // optional: raw conv output of the layer BELOW + its (scale, shift): the epilogue also emits that
  const float* bn_scale = nullptr; //   layer's BatchNorm-backward sums (sum dz, sum dz * y per channel) ...
  const float* bn_shift = nullptr;
  float* stat_partials = nullptr;  // ... as [conv3x3_bwd16_grid()][16][2]
  float* slabs;                    // out: conv3x3_bwd16_grid() weight-gradient slabs of 16 * 256 floats (Winograd domain,
                                   //   tap domain after the in-kernel output transform: 9 * 256 floats each; reduce with launch_wgrad_reduce*, nblk = grid)
  int B, H, W;
  const float* pool_gp = nullptr;  // with y (not the tail): g_eff = g + 0.25 * pool_gp[b][y/2][x/2] is formed while staging -- the AvgPool2d(2,2)
                                   //   adjoint of the half-resolution gradient (16 channels, H/2 x W/2) of a layer that also feeds a pooling stage
                                   //   (inbloc.bloc.3); the BatchNorm-backward reduction then need not write the completed gradient back
  int store_dz = 0;                // with stat_partials: gin is stored MASKED, dz = gin * [bn_y * scale + shift > 0] -- what the layer below's
                                   //   BatchNorm backward consumes; for a layer below whose only other consumer is the linear form of its
                                   //   weight gradient (inbloc.bloc.0: edge_conv.hip conv_in_dz_wgrad_kernel)
  int half = 0;                    // 1: the activation tensors above are stored as bf16 (the bf16 compute mode; the arithmetic stays
                                   //    fp32 -- fp32 MFMAs in the Winograd domain: these layers are HBM-bound, what counts is the bytes)
};
bool conv3x3_bwd16_applies(int B, int H, int W);
int conv3x3_bwd16_grid(int B, int H, int W);           // workgroups launched = stat_partials rows = slabs
int launch_conv3x3_bwd16(const Bwd16Args& a, hipStream_t s);

// dgrad: replicate-padding adjoint fold for the border pixels (adds to g_in).  wdg_layer = the layer's
// dgrad weight pack [fragment order | tap-major] written by pack_weights.
int launch_dgrad_border_fix(const float* dy, int Cout, const float* wdg_layer, int Cin, float* g0, int C0,
                            int split_ch, float* g1, int C1, int B, int H, int W, hipStream_t s, int bf16 = 0,
                            const float* bn_y = nullptr, const float* bn_scale = nullptr, const float* bn_shift = nullptr,
                            float* bn_partials = nullptr, int masked = 0);   // masked: g0 holds g * [bn_y*scale+shift > 0] (with bn_partials)
int dgrad_border_waves(int B, int H, int W, int Cin);   // rows of bn_partials ([wave][16][2]) the border kernel writes

// wwf / wwd (optional): Winograd-domain packs of all layers, 16/9 of the size and offsets of wfwd
// nbt (optional, with the Winograd packs): nbt_n int64 counters incremented by the same launch (BatchNorm num_batches_tracked)
int launch_pack_weights(const float* params, float* wfwd, float* wdgrad, hipStream_t s, float* wwf = nullptr, float* wwd = nullptr,
                        long long* nbt = nullptr, int nbt_n = 0);
