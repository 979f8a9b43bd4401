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
