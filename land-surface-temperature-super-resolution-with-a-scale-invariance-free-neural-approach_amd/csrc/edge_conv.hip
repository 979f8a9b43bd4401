// The two thin convolutions -- thin in one GEMM dimension only, so they still run on the matrix cores:
//   * inbloc.bloc.0 : Conv2d(2 -> 16), reads the model's NCHW input (model.py:596, :135)   K = 18
//   * outlay        : Conv2d(16 -> 1) + bias, writes the NCHW output (model.py:605)        N = 1
// 16x16-pixel tiles, replicate padding; the forward kernels are MFMA products (see each kernel).
#include "edge_conv.h"

namespace {

// ------------------------------------------------------------------------------------------
// input conv forward: x (B,2,H,W) NCHW -> y (B,H,W,16) NHWC raw, + BatchNorm partial statistics
// ------------------------------------------------------------------------------------------
// On the matrix cores: D[co][px] = sum_k A[co][k] * B[k][px] with k = ci*9 + t (18, padded to 20 = five
// v_mfma_f32_16x16x4_f32 per 16-pixel row): A = the lane's five weights, held for the whole kernel; B = one ds_read_b32
// per MFMA from the x halo planes at a per-lane constant (ci, tap) offset.  The D fragment (4 consecutive output
// channels of one pixel per lane) is exactly one 16-byte NHWC store, and the BatchNorm sums fall out of the same
// registers.  (The scalar version: 288 FMAs per pixel, statistics transposed through LDS, 102-110 us against a 55 us
// write floor.)
template <bool HS>   // HS: y stored as bf16 (the bf16 compute mode, common.h); the statistics are those of the stored values
__global__ __launch_bounds__(256) void conv_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, float* __restrict__ partials,
                                                          int B, int H, int W) {
  // PERSISTENT since round 3: conv_in_fwd_blocks() workgroups walk the 16x16 tiles, so the BatchNorm statistics come out as
  // <= 2048 partial rows instead of one per tile (16,384 at batch 64: the finalize launch behind it took 28 us, now 7)
  __shared__ float tile[2][2 * 324];
  __shared__ float red[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ntiles = B * tiles_x * tiles_y;
  auto stage = [&](int t, float* dst) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int x0 = tx * 16, y0 = ty * 16;
    for (int e = tid; e < 2 * 324; e += 256) {
      const int c = e / 324, p = e - c * 324;
      const int py = p / 18, px = p - py * 18;
      const int gy = clampi(y0 - 1 + py, 0, H - 1), gx = clampi(x0 - 1 + px, 0, W - 1);
      dst[e] = x[((size_t)(b * 2 + c) * H + gy) * W + gx];
    }
  };
  if ((int)blockIdx.x < ntiles) stage(blockIdx.x, tile[0]);
  float wa[5];      // A[co = i][k = 4j + kq]: w is OIHW = [co][ci*9 + t]; k >= 18 is padding (weight 0, any finite B)
  int off[5];       // B[k][px = i]: plane ci, tap (t/3, t%3) of the halo tile, row added per MFMA
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int kk = 4 * j + kq;
    wa[j] = kk < 18 ? w[i * 18 + kk] : 0.f;
    const int c = kk >= 9 ? 1 : 0, t = kk < 18 ? kk - 9 * c : 0;
    off[j] = (kk < 18 ? c * 324 : 0) + (t / 3) * 18 + t % 3 + i;
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  int buf = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x, buf ^= 1) {
    __syncthreads();                                   // tile[buf] staged; everybody is past the previous use of tile[buf ^ 1]
    if (t + (int)gridDim.x < ntiles) stage(t + gridDim.x, tile[buf ^ 1]);
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int x0 = tx * 16, y0 = ty * 16;
    const float* T = tile[buf];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * wave + r;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 5; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], T[off[j] + row * 18], acc, 0, 0, 0);
      // lane: output channels 4*kq .. 4*kq+3 of pixel (row, i); partial tiles when H or W is not a multiple of 16
      if (y0 + row < H && x0 + i < W) {
        const float4 o = as_stored4<HS>(make_float4(acc[0], acc[1], acc[2], acc[3]));
        stA4<HS>(y, ((size_t)(b * H + y0 + row) * W + x0 + i) * 16 + 4 * kq, o);
        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] += ov[q]; s2[q] = fmaf(ov[q], ov[q], s2[q]); }
      }
    }
  }
  if (partials != nullptr) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { s1[q] += __shfl_xor(s1[q], m); s2[q] += __shfl_xor(s2[q], m); }
      if (i == 0) { red[wave][4 * kq + q][0] = s1[q]; red[wave][4 * kq + q][1] = s2[q]; }
    }
    __syncthreads();
    if (tid < 32) {
      const int co = tid >> 1, j = tid & 1;
      const float s = red[0][co][j] + red[1][co][j] + red[2][co][j] + red[3][co][j];
      partials[((size_t)blockIdx.x * 16 + co) * 2 + j] = s;
    }
  }
}

// input conv weight gradient: dW[co][ci][t] = sum_p dy[p][co] * x[clamp(p+t)][ci]  (288 outputs) on the
// matrix cores: M = co (16), N = n = ci*9+t (18, padded to two 16-wide tiles), K = pixels.
//   A lane (i = co, k = pixel)  <- dy tile [256 px][16]        (conflict-free: pixel stride 16 floats)
//   B lane (j = n,  k = pixel)  <- x halo planes at a per-lane constant (ci, tap) offset
// Persistent workgroups; each wave takes every 4th k-step; partial [blk][288] summed by sum_partials_kernel.
// FUSED: dy is not read but computed while staging, dy = scale*g*[z>0] + k1*y + k0 (the BatchNorm+ReLU backward
// of inbloc.bloc.1/2; `dy` then holds g) -- the first layer has no input gradient, so this wgrad is dy's only
// consumer and dy never goes to HBM.
template <bool FUSED, bool HS>
__global__ __launch_bounds__(256) void conv_in_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ yraw,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const double* __restrict__ coef,
                                                            float* __restrict__ partials, int B, int H, int W) {
  SIFSR_CHAIN_PRIO();
  __shared__ float tile[2 * 324 + 8];
  __shared__ float dyt[256 * 16];
  __shared__ float red[4][2][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ntiles = B * tiles_x * tiles_y;
  const int i16 = lane & 15, k = lane >> 4;
  int offn[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = nt * 16 + i16;
    offn[nt] = n < 18 ? (n / 9) * 324 + ((n % 9) / 3) * 18 + (n % 9) % 3 : 0;   // n >= 18: columns never stored
  }
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  float4 sc4, sh4;
  double sd[4], k1[4], k0[4];
  if (FUSED) {
    const int q4 = tid & 3;   // the staging loop below gives every thread a fixed channel quad
    sc4 = ld4(scale + 4 * q4); sh4 = ld4(shift + 4 * q4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sd[j] = coef[4 * q4 + j]; k1[j] = coef[16 + 4 * q4 + j]; k0[j] = coef[32 + 4 * q4 + j]; }
  }
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int x0 = tx * 16, y0 = ty * 16;
    __syncthreads();
    for (int e = tid; e < 2 * 324; e += 256) {
      const int c = e / 324, p = e - c * 324;
      const int py = p / 18, px = p - py * 18;
      const int gy = clampi(y0 - 1 + py, 0, H - 1), gx = clampi(x0 - 1 + px, 0, W - 1);
      tile[e] = x[((size_t)(b * 2 + c) * H + gy) * W + gx];
    }
    for (int e = tid; e < 256 * 4; e += 256) {
      const int p = e >> 2, c4 = e & 3;
      if (y0 + (p >> 4) >= H || x0 + (p & 15) >= W) {     // outside the image: contributes nothing
        *reinterpret_cast<float4*>(&dyt[p * 16 + 4 * c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      const size_t off = ((size_t)(b * H + y0 + (p >> 4)) * W + x0 + (p & 15)) * 16 + 4 * c4;
      float4 v = ldA4<HS>(dy, off);
      if (FUSED) {
        const float4 yv = ldA4<HS>(yraw, off);
        v.x = (float)fma(sd[0], (double)(fmaf(yv.x, sc4.x, sh4.x) > 0.f ? v.x : 0.f), fma(k1[0], (double)yv.x, k0[0]));
        v.y = (float)fma(sd[1], (double)(fmaf(yv.y, sc4.y, sh4.y) > 0.f ? v.y : 0.f), fma(k1[1], (double)yv.y, k0[1]));
        v.z = (float)fma(sd[2], (double)(fmaf(yv.z, sc4.z, sh4.z) > 0.f ? v.z : 0.f), fma(k1[2], (double)yv.z, k0[2]));
        v.w = (float)fma(sd[3], (double)(fmaf(yv.w, sc4.w, sh4.w) > 0.f ? v.w : 0.f), fma(k1[3], (double)yv.w, k0[3]));
      }
      *reinterpret_cast<float4*>(&dyt[p * 16 + 4 * c4]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = wave; ks < 64; ks += 4) {
      const int r = ks >> 2, qd = ks & 3;
      const float av = dyt[(r * 16 + 4 * qd + k) * 16 + i16];
      const int base = r * 18 + 4 * qd + k;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tile[offn[0] + base], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tile[offn[1] + base], acc[1], 0, 0, 0);
    }
  }
  // D[row = co = 4*(lane>>4) + r][col = n]  ->  red[wave][nt][co*16 + col]; sum the 4 waves; e = co*18 + n
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][nt][(4 * k + r) * 16 + i16] = acc[nt][r];
  __syncthreads();
  for (int e = tid; e < 288; e += 256) {
    const int co = e / 18, n = e % 18;
    const int nt = n >> 4, col = n & 15;
    partials[(size_t)blockIdx.x * 288 + e] =
        red[0][nt][co * 16 + col] + red[1][nt][co * 16 + col] + red[2][nt][co * 16 + col] + red[3][nt][co * 16 + col];
  }
}

// ------------------------------------------------------------------------------------------
// output conv forward: a = relu(y*scale+shift) (B,H,W,16) -> sr (B,1,H,W) = conv(a) + bias
// ------------------------------------------------------------------------------------------
constexpr int OCS = 20;   // LDS pixel stride (floats): 5 slots of 16 B -> conflict-free b128 reads

// 16 -> 1 output convolution on the matrix cores, in two steps per 16x16 output tile:
//   1. P[p'][t] = sum_c a[p'][c] * w[c][t] for every pixel p' of the 18x18 halo tile and every tap t: a
//      (pixels x 16 channels) x (16 channels x 9 taps, padded to 16) product = 4 v_mfma_f32_16x16x4_f32 per 16 pixels.
//      A operand = the lane's own float4 of 4 channels of one pixel (a coalesced 64 B / pixel read, BatchNorm + ReLU
//      applied in registers), B operand = 4 weights per lane held for the whole kernel; no LDS staging of the input.
//   2. out[p] = bias + sum_t P[p + t][t]: nine LDS reads per output pixel.
// The scalar version (144 FMAs and 36 ds_read_b128 per pixel) ran at 114-127 us against a 50 us HBM floor.
template <bool HS>
__global__ __launch_bounds__(256) void conv_out_fwd_kernel(const float* __restrict__ y, const float* scale,
                                                           const float* shift, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int H, int W) {
  constexpr int NGRP = 21;                      // ceil(324 / 16) groups of 16 halo pixels
  __shared__ float P[NGRP * 16 * 9];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16, b = blockIdx.z;
  float wb[4];                                   // B[k = kq][n = tap i] of MFMA j: w is (1,16,3,3) = [ci][t]
#pragma unroll
  for (int j = 0; j < 4; ++j) wb[j] = i < 9 ? w[(4 * kq + j) * 9 + i] : 0.f;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale != nullptr) { sc = ld4(scale + 4 * kq); sh = ld4(shift + 4 * kq); }
  constexpr int NPW = (NGRP + 3) / 4;            // groups per wave (wave w owns groups w, w+4, ...)
  float4 v[NPW];
#pragma unroll
  for (int n = 0; n < NPW; ++n) {                // all loads of this wave first
    const int g = wave + 4 * n;
    int p = g * 16 + i;
    if (p > 323) p = 323;                        // the last group is partial; group 21+ does not exist (clamped, unused)
    const int r = p / 18, c = p - r * 18;
    const int gy = clampi(y0 - 1 + r, 0, H - 1), gx = clampi(x0 - 1 + c, 0, W - 1);   // replicate padding
    v[n] = ldA4<HS>(y, ((size_t)(b * H + gy) * W + gx) * 16 + 4 * kq);
  }
#pragma unroll
  for (int n = 0; n < NPW; ++n) {
    const int g = wave + 4 * n;
    if (g >= NGRP) break;                        // wave-uniform
    float4 a = v[n];
    if (scale != nullptr) a = bn_relu4(a, sc, sh);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wb[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wb[3], acc, 0, 0, 0);
    if (i < 9) {                                 // D row = pixel 4*kq + r of the group, column = tap i
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(g * 16 + 4 * kq + r) * 9 + i] = acc[r];
    }
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  float s = bias[0];
#pragma unroll
  for (int t = 0; t < 9; ++t) s += P[((ty + t / 3) * 18 + tx + t % 3) * 9 + t];
  if (y0 + ty < H && x0 + tx < W) out[(size_t)(b * H + y0 + ty) * W + x0 + tx] = s;
}

// output conv input-gradient: g[q][ci] = sum_t w[ci][t] * sum_{p: clamp(p+t)=q} dsr[p]  (replicate adjoint)
__global__ __launch_bounds__(256) void conv_out_dgrad_kernel(const float* __restrict__ dsr, const float* __restrict__ w,
                                                             float* __restrict__ g, int B, int H, int W) {
  const size_t n = (size_t)B * H * W;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const int qx = q % W, qy = (q / W) % H;
    const size_t img = q - (size_t)qy * W - qx;
    float S[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ty = t / 3 - 1, tx = t % 3 - 1;
      int ys[2], ny = 0, xs[2], nx = 0;
      if (qy - ty >= 0 && qy - ty < H) ys[ny++] = qy - ty;
      if ((qy == 0 && ty == -1) || (qy == H - 1 && ty == 1)) ys[ny++] = qy;
      if (qx - tx >= 0 && qx - tx < W) xs[nx++] = qx - tx;
      if ((qx == 0 && tx == -1) || (qx == W - 1 && tx == 1)) xs[nx++] = qx;
      float s = 0.f;
      for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix) s += dsr[img + (size_t)ys[iy] * W + xs[ix]];
      S[t] = s;
    }
    float* gp = g + q * 16;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) s = fmaf(w[(4 * c4 + j) * 9 + t], S[t], s);
        o[j] = s;
      }
      st4(gp + 4 * c4, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
}

// output conv weight/bias gradient: dW[ci][t] = sum_p dsr[p] * a[clamp(p+t)][ci]; db = sum dsr (145 outputs).
// On the matrix cores with the HALO pixel p' = p + t as contraction index, so the A operand does not depend
// on the tap:  D[ci][t] = sum_{p'} a[p'][ci] * dsr[p' - t]   (dsr = 0 outside the 16x16 tile).
//   A lane (i = ci, k = pixel) <- a halo tile [18 rows][20 cols (2 zero pad)][OCS]
//   B lane (j = t,  k = pixel) <- dsr tile at the per-lane tap shift
__global__ __launch_bounds__(256) void conv_out_wgrad_kernel(const float* __restrict__ y, const float* scale,
                                                             const float* shift, const float* __restrict__ dsr,
                                                             float* __restrict__ partials, int B, int H, int W) {
  __shared__ float tile[18 * 20 * OCS];
  __shared__ float dt[256];
  __shared__ float red[4][256];
  __shared__ float bsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ntiles = B * tiles_x * tiles_y;
  const int i16 = lane & 15, k = lane >> 4;
  const int tty = i16 < 9 ? i16 / 3 : 100, ttx = i16 < 9 ? i16 % 3 : 100;   // lanes j >= 9: always out of range -> 0
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;
  for (int e = tid; e < 18 * 20 * OCS; e += 256) tile[e] = 0.f;     // pad columns 18,19 stay zero
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int x0 = tx * 16, y0 = ty * 16;
    __syncthreads();
    for (int e = tid; e < 324 * 4; e += 256) {
      const int p = e >> 2, c4 = e & 3;
      const int py = p / 18, px = p - py * 18;
      const int gy = clampi(y0 - 1 + py, 0, H - 1), gx = clampi(x0 - 1 + px, 0, W - 1);
      float4 v = ld4(y + ((size_t)(b * H + gy) * W + gx) * 16 + 4 * c4);
      if (scale != nullptr) v = bn_relu4(v, ld4(scale + 4 * c4), ld4(shift + 4 * c4));
      *reinterpret_cast<float4*>(&tile[(py * 20 + px) * OCS + 4 * c4]) = v;
    }
    const float dv = (y0 + (tid >> 4) < H && x0 + (tid & 15) < W) ? dsr[(size_t)(b * H + y0 + (tid >> 4)) * W + x0 + (tid & 15)] : 0.f;
    dt[tid] = dv;
    bacc += dv;
    __syncthreads();
    for (int ks = wave; ks < 90; ks += 4) {          // 18 rows x 5 quads of the padded halo tile
      const int row = ks / 5, col = (ks - row * 5) * 4 + k;
      const float av = tile[(row * 20 + col) * OCS + i16];
      const int oy = row - tty, ox = col - ttx;
      const float bv = (oy >= 0 && oy < 16 && ox >= 0 && ox < 16) ? dt[oy * 16 + ox] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * k + r) * 16 + i16] = acc[r];   // [ci][t]
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) bacc += __shfl_xor(bacc, m);
  if (lane == 0) bsum[wave] = bacc;
  __syncthreads();
  if (tid < 144) {
    const int ci = tid / 9, t = tid % 9;
    partials[(size_t)blockIdx.x * 145 + tid] = red[0][ci * 16 + t] + red[1][ci * 16 + t] + red[2][ci * 16 + t] + red[3][ci * 16 + t];
  } else if (tid == 144) {
    partials[(size_t)blockIdx.x * 145 + 144] = bsum[0] + bsum[1] + bsum[2] + bsum[3];
  }
}

// ------------------------------------------------------------------------------------------
// input conv weight gradient WITHOUT reading (g, y) again ("head", round 3).  dL/dy of inbloc.bloc.0 is affine in dz = g*[z>0] and y,
//   dy = sd*dz + k1*y + k0  (per channel; BatchNorm + ReLU backward, model.py:136-137),  and y = W p (Conv2d(2, 16, 3, bias=False)),
// p(q) = the 18 replicate-padded input values under pixel q.  So
//   dW[c][k] = sum_q dy[q][c] p(q)[k] = sd[c] * D[c][k] + k1[c] * (W G)[c][k] + k0[c] * X[k]
//   D = sum_q dz[q] p(q)^T   (16 x 18: the only part that needs a gradient tensor -- ONE 16-channel read, of dz, which the
//                             input-gradient kernel of inbloc.bloc.3 stores in place of g: it has the mask in its epilogue)
//   G = sum_q p(q) p(q)^T    (18 x 18 Gram matrix of the input patches),  X = sum_q p(q):  functions of the network INPUT alone,
//                             computed on the second stream while the backward chain runs.
// The fused form (conv_in_wgrad_kernel<FUSED>) reads g and y = 2 tensors of 16 channels at full resolution for a 2 -> 16 layer.
// ------------------------------------------------------------------------------------------
constexpr int GRAM_N = 189, GRAM_PITCH = 192;            // 171 upper-triangle entries of G (row-major, k >= k') + 18 of X
__host__ __device__ constexpr int gram_idx(int kr, int k) { return kr * 18 - kr * (kr - 1) / 2 + (k - kr); }

// row kr of the upper triangle = entries (kr, kr .. 17), as packed pairs (k, k + 1) from k = kr (pv[18] = pv[19] = 0 pad an odd row)
template <int WV> static __device__ __forceinline__ void gram_rows(const float (&pv)[20], f32x2 (&acc)[26]) {
  int n = 0;
#pragma unroll
  for (int kr = WV; kr < 18; kr += 4)
#pragma unroll
    for (int k = kr; k < 18; k += 2) { acc[n] = __builtin_elementwise_fma((f32x2){pv[kr], pv[kr]}, (f32x2){pv[k], pv[k + 1]}, acc[n]); ++n; }
}
template <int WV> static __device__ __forceinline__ void gram_store(const f32x2 (&acc)[26], float* __restrict__ row, int lane) {
  int n = 0;
#pragma unroll
  for (int kr = WV; kr < 18; kr += 4)
#pragma unroll
    for (int k = kr; k < 18; k += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v = acc[n][h];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane == 0 && k + h < 18) row[gram_idx(kr, k + h)] = v;
      }
      ++n;
    }
}

// persistent workgroups over 16x16 tiles; wave w accumulates the rows k' = w, w + 4, ... of the upper triangle for all 256 pixels
// of the tile (4 per lane), wave 3 (the shortest rows) also X.  partials: [gridDim.x][GRAM_PITCH]
template <bool FULL>   // FULL: H and W are multiples of 16 (no partial tiles)
__global__ __launch_bounds__(256, 2) void conv_in_gram_kernel(const float* __restrict__ x, float* __restrict__ partials, int B, int H, int W) {
  __shared__ float tile[2][2 * 324 + 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ntiles = B * tiles_x * tiles_y;
  // the next tile's halo planes travel through registers (3 values per thread): the loads are in flight while this tile is processed
  float px_[3];
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int x0 = tx * 16, y0 = ty * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = min(tid + 256 * i, 2 * 324 - 1);
      const int c = e / 324, p = e - c * 324;
      const int py = p / 18, pxx = p - py * 18;
      px_[i] = x[((size_t)(b * 2 + c) * H + clampi(y0 - 1 + py, 0, H - 1)) * W + clampi(x0 - 1 + pxx, 0, W - 1)];
    }
  };
  f32x2 acc[26], xs[9];
#pragma unroll
  for (int i = 0; i < 26; ++i) acc[i] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) xs[i] = (f32x2){0.f, 0.f};
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  int buf = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x, buf ^= 1) {
    tile[buf][tid] = px_[0];
    tile[buf][tid + 256] = px_[1];
    if (tid + 512 < 2 * 324) tile[buf][tid + 512] = px_[2];
    __syncthreads();          // one barrier per tile (the other buffer was last read before the previous one)
    fetch(t + (int)gridDim.x < ntiles ? t + (int)gridDim.x : t);
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
    const int x0 = tx * 16, y0 = ty * 16;
    const float* T = tile[buf];
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int r = (lane >> 4) + 4 * i, col = lane & 15;
      const bool inside = FULL || (y0 + r < H && x0 + col < W);    // partial tiles: pixels past the image contribute nothing
      float pv[20];
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        const float v = T[(k / 9) * 324 + (r + (k % 9) / 3) * 18 + col + (k % 9) % 3];
        pv[k] = inside ? v : 0.f;
      }
      pv[18] = pv[19] = 0.f;
      switch (wave) {
        case 0: gram_rows<0>(pv, acc); break;
        case 1: gram_rows<1>(pv, acc); break;
        case 2: gram_rows<2>(pv, acc); break;
        default:
          gram_rows<3>(pv, acc);
#pragma unroll
          for (int k = 0; k < 9; ++k) xs[k] += (f32x2){pv[2 * k], pv[2 * k + 1]};
      }
    }
  }
  float* row = partials + (size_t)blockIdx.x * GRAM_PITCH;
  switch (wave) {
    case 0: gram_store<0>(acc, row, lane); break;
    case 1: gram_store<1>(acc, row, lane); break;
    case 2: gram_store<2>(acc, row, lane); break;
    default:
      gram_store<3>(acc, row, lane);
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        float v = xs[k >> 1][k & 1];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane == 0) row[171 + k] = v;
      }
  }
}

// gram[e] = sum over the workgroup rows, float64, fixed order: one 64-thread workgroup per entry
__global__ __launch_bounds__(64) void conv_in_gram_reduce_kernel(const float* __restrict__ partials, int nblk, double* __restrict__ gram) {
  const int e = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int k = lane; k < nblk; k += 64) s += (double)partials[(size_t)k * GRAM_PITCH + e];
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) gram[e] = s;
}

// D partials: work item = (pixel, channel quad), dz as one coalesced float4 straight from HBM, the pixel's 18 patch values from the
// LDS halo planes of x; 72 FMAs (36 packed) per item into per-lane accumulators.  Persistent, next tile's dz quads and x halo
// in flight (as tail_bwd_reduce_kernel, fused_edges.hip).  partials: [gridDim.x][288], e = co * 18 + ci * 9 + t
template <bool HS>   // HS: dz stored as bf16
__global__ __launch_bounds__(256, 3) void conv_in_dz_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                  float* __restrict__ partials, int B, int H, int W) {
  SIFSR_CHAIN_PRIO();
  __shared__ float tile[2][2 * 324 + 8];
  __shared__ float red[4][4][72];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ntiles = B * tiles_x * tiles_y;
  const int c4 = tid & 3, pl = tid >> 2;
  const int lx = pl & 15, ly0 = pl >> 4;
  f32x2 acc[4][9];                                       // D[4 c4 + j][2 u, 2 u + 1]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int u = 0; u < 9; ++u) acc[j][u] = (f32x2){0.f, 0.f};
  float4 pz[4];
  float px_[3];
  size_t nbase = 0;
  int nrow[4] = {0, 0, 0, 0};
  bool nin[4] = {false, false, false, false};
  auto fetch_z = [&](int k) __attribute__((always_inline)) { pz[k] = ldA4<HS>(dz, nbase + (size_t)nrow[k]); };
  auto fetch = [&](int tl) __attribute__((always_inline)) {
    const int tx = tl % tiles_x, r = tl / tiles_x, ty = r % tiles_y, b = r / tiles_y;
    const int x0 = tx * 16, y0 = ty * 16;
    const int gx = min(x0 + lx, W - 1);
    nbase = ((size_t)(b * H) * W + gx) * 16 + 4 * c4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      nrow[k] = min(y0 + ly0 + 4 * k, H - 1) * W * 16;
      nin[k] = y0 + ly0 + 4 * k < H && x0 + lx < W;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = min(tid + 256 * i, 2 * 324 - 1);
      const int c = e / 324, p = e - c * 324;
      const int py = p / 18, pxx = p - py * 18;
      px_[i] = x[((size_t)(b * 2 + c) * H + clampi(y0 - 1 + py, 0, H - 1)) * W + clampi(x0 - 1 + pxx, 0, W - 1)];
    }
  };
  int buf = 0;
  bool cin_[4] = {false, false, false, false};
  if ((int)blockIdx.x < ntiles) {
    fetch(blockIdx.x);
#pragma unroll
    for (int k = 0; k < 4; ++k) fetch_z(k);
  }
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x, buf ^= 1) {
    tile[buf][tid] = px_[0];
    tile[buf][tid + 256] = px_[1];
    if (tid + 512 < 2 * 324) tile[buf][tid + 512] = px_[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) cin_[k] = nin[k];
    __syncthreads();          // one barrier per tile (the other buffer was last read before the previous one)
    fetch(tl + (int)gridDim.x < ntiles ? tl + (int)gridDim.x : tl);
    const float* T = tile[buf];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 zv = pz[k];
      fetch_z(k);
      __builtin_amdgcn_sched_barrier(0);
      if (!cin_[k]) zv = make_float4(0.f, 0.f, 0.f, 0.f);   // partial tiles
      const int ly = ly0 + 4 * k;
      f32x2 pp[9];                                         // the 18 patch values as pairs (n, n + 1), n = ci * 9 + t
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int n0 = 2 * u, n1 = 2 * u + 1;
        pp[u] = (f32x2){T[(n0 / 9) * 324 + (ly + (n0 % 9) / 3) * 18 + lx + (n0 % 9) % 3],
                        T[(n1 / 9) * 324 + (ly + (n1 % 9) / 3) * 18 + lx + (n1 % 9) % 3]};
      }
      const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 9; ++u) acc[j][u] = __builtin_elementwise_fma((f32x2){zz[j], zz[j]}, pp[u], acc[j][u]);
    }
  }
  // across the 16 lanes of a channel quad, then the 4 waves (fp32: a workgroup sums a few thousand terms; float64 across workgroups)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v = acc[j][u][h];
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane < 4) red[wave][c4][j * 18 + 2 * u + h] = v;
      }
  __syncthreads();
  for (int e = tid; e < 288; e += 256) {
    const int co = e / 18, n = e - co * 18;
    const int q = co >> 2, i = (co & 3) * 18 + n;
    partials[(size_t)blockIdx.x * 288 + e] = red[0][q][i] + red[1][q][i] + red[2][q][i] + red[3][q][i];
  }
}

// dW[c][k] = sd[c] * D[c][k] + k1[c] * (W G)[c][k] + k0[c] * X[k], float64; one 64-thread workgroup per entry e = c * 18 + k
__global__ __launch_bounds__(64) void conv_in_dw_combine_kernel(const float* __restrict__ partials, int nblk,
                                                                const double* __restrict__ gram, const float* __restrict__ w,
                                                                const double* __restrict__ coef, float* __restrict__ dw) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const int c = e / 18, k = e - c * 18;
  double s = 0.0;
  for (int r = lane; r < nblk; r += 64) s += (double)partials[(size_t)r * 288 + e];
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) {
    double a = 0.0;
    for (int kr = 0; kr < 18; ++kr) a += (double)w[c * 18 + kr] * gram[kr <= k ? gram_idx(kr, k) : gram_idx(k, kr)];
    dw[e] = (float)(coef[c] * s + coef[16 + c] * a + coef[32 + c] * gram[171 + k]);
  }
}

// out[e] = sum_k partials[k][e] in float64, fixed order (deterministic): 4 outputs x 64 row lanes per block
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partials, int nblk, int n,
                                                           float* __restrict__ out) {
  __shared__ double part[64][4];
  const int el = threadIdx.x & 3, grp = threadIdx.x >> 2;
  const int e = blockIdx.x * 4 + el;
  double s = 0.0;
  if (e < n)
    for (int k = grp; k < nblk; k += 64) s += (double)partials[(size_t)k * n + e];
  part[grp][el] = s;
  __syncthreads();
  for (int st = 32; st > 0; st >>= 1) {
    if (grp < st) part[grp][el] += part[grp + st][el];
    __syncthreads();
  }
  if (grp == 0 && e < n) out[e] = (float)part[0][el];
}

}  // namespace

int conv_in_fwd_blocks(int B, int H, int W) {   // workgroups launched == statistic rows written
  const int ntiles = B * ((H + 15) / 16) * ((W + 15) / 16);
  return ntiles < 2048 ? ntiles : 2048;
}

int launch_conv_in_fwd(const float* x, const float* w, float* y, float* partials, int B, int H, int W, hipStream_t s) {
  if (H < 1 || W < 1 || B < 1) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL(conv_in_fwd_kernel<true>, dim3(conv_in_fwd_blocks(B, H, W)), dim3(256), 0, s, x, w, y, partials, B, H, W);
  else hipLaunchKernelGGL(conv_in_fwd_kernel<false>, dim3(conv_in_fwd_blocks(B, H, W)), dim3(256), 0, s, x, w, y, partials, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_conv_in_wgrad(const float* x, const float* dy, float* partials, int nblk, float* dw, int B, int H, int W,
                         hipStream_t s) {
  if (H < 1 || W < 1) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL((conv_in_wgrad_kernel<false, true>), dim3(nblk), dim3(256), 0, s, x, dy, nullptr, nullptr, nullptr, nullptr, partials, B, H, W);
  else hipLaunchKernelGGL((conv_in_wgrad_kernel<false, false>), dim3(nblk), dim3(256), 0, s, x, dy, nullptr, nullptr, nullptr, nullptr, partials, B, H, W);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(72), dim3(256), 0, s, partials, nblk, 288, dw);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_conv_in_wgrad_fused(const float* x, const float* g, const float* y, const float* scale, const float* shift,
                               const double* coef, float* partials, int nblk, float* dw, int B, int H, int W,
                               hipStream_t s) {
  if (H < 1 || W < 1) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL((conv_in_wgrad_kernel<true, true>), dim3(nblk), dim3(256), 0, s, x, g, y, scale, shift, coef, partials, B, H, W);
  else hipLaunchKernelGGL((conv_in_wgrad_kernel<true, false>), dim3(nblk), dim3(256), 0, s, x, g, y, scale, shift, coef, partials, B, H, W);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(72), dim3(256), 0, s, partials, nblk, 288, dw);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

size_t conv_in_gram_scratch_floats() { return (size_t)1024 * GRAM_PITCH + 2 * GRAM_PITCH; }
// scratch (conv_in_gram_scratch_floats() floats): [1024][192] partial rows, then 189 float64 = the result (G upper triangle | X)
int launch_conv_in_gram(const float* x, float* scratch, int B, int H, int W, hipStream_t s) {
  if (H < 1 || W < 1 || B < 1) return SIFSR_ERR_SHAPE;
  const int ntiles = B * ((H + 15) / 16) * ((W + 15) / 16);
  const int nblk = ntiles < 512 ? ntiles : 512;   // two workgroups per CU are resident (registers): one round
  if (H % 16 == 0 && W % 16 == 0) hipLaunchKernelGGL(conv_in_gram_kernel<true>, dim3(nblk), dim3(256), 0, s, x, scratch, B, H, W);
  else hipLaunchKernelGGL(conv_in_gram_kernel<false>, dim3(nblk), dim3(256), 0, s, x, scratch, B, H, W);
  hipLaunchKernelGGL(conv_in_gram_reduce_kernel, dim3(GRAM_N), dim3(64), 0, s, scratch, nblk, reinterpret_cast<double*>(scratch + (size_t)1024 * GRAM_PITCH));
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
const double* conv_in_gram_result(const float* scratch) { return reinterpret_cast<const double*>(scratch + (size_t)1024 * GRAM_PITCH); }

// D partials from the stored dz (nblk workgroups; three per CU are resident: nblk <= 768 keeps it to one round)
int launch_conv_in_dz_wgrad(const float* x, const float* dz, float* partials, int nblk, int B, int H, int W, hipStream_t s) {
  if (H < 1 || W < 1 || nblk < 1) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL(conv_in_dz_wgrad_kernel<true>, dim3(nblk), dim3(256), 0, s, x, dz, partials, B, H, W);
  else hipLaunchKernelGGL(conv_in_dz_wgrad_kernel<false>, dim3(nblk), dim3(256), 0, s, x, dz, partials, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_conv_in_dw_combine(const float* partials, int nblk, const double* gram, const float* w, const double* coef, float* dw,
                              hipStream_t s) {
  hipLaunchKernelGGL(conv_in_dw_combine_kernel, dim3(288), dim3(64), 0, s, partials, nblk, gram, w, coef, dw);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_conv_out_fwd(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                        float* out, int B, int H, int W, hipStream_t s) {
  if (H < 1 || W < 1) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL(conv_out_fwd_kernel<true>, dim3((W + 15) / 16, (H + 15) / 16, B), dim3(256), 0, s, y, scale, shift, w, bias, out, H, W);
  else hipLaunchKernelGGL(conv_out_fwd_kernel<false>, dim3((W + 15) / 16, (H + 15) / 16, B), dim3(256), 0, s, y, scale, shift, w, bias, out, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_conv_out_dgrad(const float* dsr, const float* w, float* g, int B, int H, int W, hipStream_t s) {
  const size_t n = (size_t)B * H * W;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv_out_dgrad_kernel, dim3(blocks), dim3(256), 0, s, dsr, w, g, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_conv_out_wgrad(const float* y, const float* scale, const float* shift, const float* dsr, float* partials,
                          int nblk, float* dw, float* db, int B, int H, int W, hipStream_t s) {
  if (H < 1 || W < 1) return SIFSR_ERR_SHAPE;
  hipLaunchKernelGGL(conv_out_wgrad_kernel, dim3(nblk), dim3(256), 0, s, y, scale, shift, dsr, partials, B, H, W);
  // dw (144 floats) and db (1 float) are adjacent in the flat gradient buffer (outlay.weight, outlay.bias)
  if (db != dw + 144) return SIFSR_ERR_ARG;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(37), dim3(256), 0, s, partials, nblk, 145, dw);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_sum_partials(const float* partials, int nblk, int n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3((n + 3) / 4), dim3(256), 0, s, partials, nblk, n, out);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
