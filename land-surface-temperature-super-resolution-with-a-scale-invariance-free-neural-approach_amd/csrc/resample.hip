// Resampling glue of the U-Net (all NHWC, fp32 or -- HS, the bf16 compute mode -- bf16 storage with fp32 arithmetic; one thread per
// (pixel, channel quad)):
//   * AvgPool2d(2,2) of relu(bn(y))                        -- DownBlock_pool, model.py:504
//   * residual sum  x + relu(bn(DoubleConv(x)))            -- ResidualConnection, model.py:311-312
//   * Upsample(x2, bilinear, align_corners=True) of relu(bn(y)) -- UpBlock, model.py:207
// and their adjoints.  The BatchNorm+ReLU of the producing layer is folded in (scale/shift), so the
// normalised activation itself is never written to HBM.
#include "edge_conv.h"

namespace {

__device__ __forceinline__ float4 maybe_bnrelu(float4 v, const float* scale, const float* shift, int c) {
  return scale != nullptr ? bn_relu4(v, ld4(scale + c), ld4(shift + c)) : v;
}

template <bool HS>
__global__ void bnrelu_pool2_kernel(const float* __restrict__ y, const float* scale, const float* shift,
                                    float* __restrict__ out, int B, int H, int W, int C) {
  // 32-bit index arithmetic (element counts < 2^30, checked by the launcher): five 64-bit divisions per element made this
  // kernel instruction-bound (62 us for 335 MB at batch 64)
  const unsigned Q = C / 4, Ho = H / 2, Wo = W / 2;
  const unsigned n = (unsigned)B * Ho * Wo * Q;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const unsigned p = e / Q;
    const int c = (int)(e - p * Q) * 4;
    const unsigned rr = p / Wo, ox = p - rr * Wo, b = rr / Ho, oy = rr - b * Ho;
    const size_t s = (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const float4 v00 = maybe_bnrelu(ldA4<HS>(y, s), scale, shift, c), v01 = maybe_bnrelu(ldA4<HS>(y, s + C), scale, shift, c);
    const float4 v10 = maybe_bnrelu(ldA4<HS>(y, s + (size_t)W * C), scale, shift, c);
    const float4 v11 = maybe_bnrelu(ldA4<HS>(y, s + (size_t)W * C + C), scale, shift, c);
    float4 o;
    o.x = (v00.x + v01.x + v10.x + v11.x) * 0.25f;
    o.y = (v00.y + v01.y + v10.y + v11.y) * 0.25f;
    o.z = (v00.z + v01.z + v10.z + v11.z) * 0.25f;
    o.w = (v00.w + v01.w + v10.w + v11.w) * 0.25f;
    stA4<HS>(out, (size_t)p * C + c, o);
  }
}

template <bool HS>
__global__ void bnrelu_add_kernel(const float* __restrict__ p, const float* __restrict__ y, const float* scale,
                                  const float* shift, float* __restrict__ out, int C, size_t nquads) {
  const unsigned Q = C / 4;   // a power of two for the model's channel counts: the modulo is a mask
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nquads; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((unsigned)e % Q) * 4;
    const float4 a = ldA4<HS>(p, e * 4), v = maybe_bnrelu(ldA4<HS>(y, e * 4), scale, shift, c);
    stA4<HS>(out, e * 4, make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w));
  }
}

// source index / weights of the align_corners=True bilinear x2 upsample for output coordinate o
// (ATen upsample_bilinear2d: ratio = (in-1)/(out-1); src = ratio*o; i0 = (int)src; lambda = src - i0)
__device__ __forceinline__ void up_coord(int o, int in, int out, int& i0, int& i1, float& l0, float& l1) {
  const float ratio = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = ratio * (float)o;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <bool HS>
__global__ void bnrelu_up2x_kernel(const float* __restrict__ y, const float* scale, const float* shift,
                                   float* __restrict__ out, int B, int Hin, int Win, int C) {
  const int Q = C / 4, Ho = 2 * Hin, Wo = 2 * Win;
  const size_t n = (size_t)B * Ho * Wo * Q;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % Q) * 4;
    const size_t p = e / Q;
    const int ox = p % Wo, oy = (p / Wo) % Ho, b = p / ((size_t)Wo * Ho);
    int y0, y1, x0, x1; float hy0, hy1, hx0, hx1;
    up_coord(oy, Hin, Ho, y0, y1, hy0, hy1);
    up_coord(ox, Win, Wo, x0, x1, hx0, hx1);
    const size_t base = (size_t)b * Hin * Win * C + c;
    const float4 v00 = maybe_bnrelu(ldA4<HS>(y, base + ((size_t)y0 * Win + x0) * C), scale, shift, c);
    const float4 v01 = maybe_bnrelu(ldA4<HS>(y, base + ((size_t)y0 * Win + x1) * C), scale, shift, c);
    const float4 v10 = maybe_bnrelu(ldA4<HS>(y, base + ((size_t)y1 * Win + x0) * C), scale, shift, c);
    const float4 v11 = maybe_bnrelu(ldA4<HS>(y, base + ((size_t)y1 * Win + x1) * C), scale, shift, c);
    float4 o;
    o.x = hy0 * (hx0 * v00.x + hx1 * v01.x) + hy1 * (hx0 * v10.x + hx1 * v11.x);
    o.y = hy0 * (hx0 * v00.y + hx1 * v01.y) + hy1 * (hx0 * v10.y + hx1 * v11.y);
    o.z = hy0 * (hx0 * v00.z + hx1 * v01.z) + hy1 * (hx0 * v10.z + hx1 * v11.z);
    o.w = hy0 * (hx0 * v00.w + hx1 * v01.w) + hy1 * (hx0 * v10.w + hx1 * v11.w);
    stA4<HS>(out, p * C + c, o);
  }
}

// adjoint of AvgPool2d(2,2): g[y][x] (+)= 0.25 * gp[y/2][x/2]
template <bool HS>
__global__ void pool2_bwd_kernel(const float* __restrict__ gp, float* __restrict__ g, int B, int H, int W, int C,
                                 int accumulate) {
  const int Q = C / 4, Ho = H / 2, Wo = W / 2;
  const size_t n = (size_t)B * H * W * Q;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % Q) * 4;
    const size_t p = e / Q;
    const int x = p % W, yy = (p / W) % H, b = p / ((size_t)W * H);
    const float4 v = ldA4<HS>(gp, (((size_t)b * Ho + yy / 2) * Wo + x / 2) * C + c);
    float4 o = make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w);
    if (accumulate) {
      const float4 a = ldA4<HS>(g, p * C + c);
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    stA4<HS>(g, p * C + c, o);
  }
}

// adjoint of the bilinear x2 upsample, gather form (deterministic): every low-res pixel collects the
// high-res pixels whose two source taps include it (output rows 2i-2 .. 2i+2 are the only candidates
// because ratio < 1/2).
template <bool HS>
__global__ void up2x_bwd_kernel(const float* __restrict__ gu, float* __restrict__ g, int B, int Hin, int Win, int C) {
  SIFSR_CHAIN_PRIO();
  const int Q = C / 4, Ho = 2 * Hin, Wo = 2 * Win;
  const size_t n = (size_t)B * Hin * Win * Q;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % Q) * 4;
    const size_t p = e / Q;
    const int ix = p % Win, iy = (p / Win) % Hin, b = p / ((size_t)Win * Hin);
    float wy[5], wx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int oy = 2 * iy - 2 + k, ox = 2 * ix - 2 + k;
      wy[k] = 0.f; wx[k] = 0.f;
      if (oy >= 0 && oy < Ho) {
        int a0, a1; float l0, l1;
        up_coord(oy, Hin, Ho, a0, a1, l0, l1);
        wy[k] = (a0 == iy ? l0 : 0.f) + (a1 == iy ? l1 : 0.f);
      }
      if (ox >= 0 && ox < Wo) {
        int a0, a1; float l0, l1;
        up_coord(ox, Win, Wo, a0, a1, l0, l1);
        wx[k] = (a0 == ix ? l0 : 0.f) + (a1 == ix ? l1 : 0.f);
      }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      if (wy[ky] == 0.f) continue;
      const int oy = 2 * iy - 2 + ky;
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        if (wx[kx] == 0.f) continue;
        const int ox = 2 * ix - 2 + kx;
        const float w = wy[ky] * wx[kx];
        const float4 v = ldA4<HS>(gu, (((size_t)b * Ho + oy) * Wo + ox) * C + c);
        acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
      }
    }
    stA4<HS>(g, p * C + c, acc);
  }
}

// ---- tiled forms for C in {16, 32, 64} (the model's channel counts) ----
// Upsample: one workgroup = 16x16 output pixels.  Their <= 10x10 source pixels are loaded, BatchNorm+ReLU'd ONCE and
// kept in LDS (the per-output form above transforms every source pixel four times and spends most of its time on
// 64-bit index arithmetic); the blend itself is the same expression, so results are bit-identical.
template <int C, bool HS>
__global__ __launch_bounds__(256) void bnrelu_up2x_tile_kernel(const float* __restrict__ y, const float* scale,
                                                               const float* shift, float* __restrict__ out, int Hin,
                                                               int Win) {
  constexpr int Q = C / 4, SR = 10;
  __shared__ float4 src[SR * SR * Q];
  const int tid = threadIdx.x;
  const int Ho = 2 * Hin, Wo = 2 * Win;
  const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16, b = blockIdx.z;
  int sy0, sx0, t1; float f0, f1;
  up_coord(oy0, Hin, Ho, sy0, t1, f0, f1);
  up_coord(ox0, Win, Wo, sx0, t1, f0, f1);
  for (int e = tid; e < SR * SR * Q; e += 256) {
    const int c4 = e % Q, p = e / Q;
    const int py = p / SR, px = p - py * SR;
    const int gy = min(sy0 + py, Hin - 1), gx = min(sx0 + px, Win - 1);
    src[e] = maybe_bnrelu(ldA4<HS>(y, ((size_t)(b * Hin + gy) * Win + gx) * C + 4 * c4), scale, shift, 4 * c4);
  }
  __syncthreads();
  for (int e = tid; e < 256 * Q; e += 256) {
    const int c4 = e % Q, p = e / Q;
    const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
    if (oy >= Ho || ox >= Wo) continue;
    int y0, y1, x0, x1; float hy0, hy1, hx0, hx1;
    up_coord(oy, Hin, Ho, y0, y1, hy0, hy1);
    up_coord(ox, Win, Wo, x0, x1, hx0, hx1);
    const float4 v00 = src[((y0 - sy0) * SR + (x0 - sx0)) * Q + c4], v01 = src[((y0 - sy0) * SR + (x1 - sx0)) * Q + c4];
    const float4 v10 = src[((y1 - sy0) * SR + (x0 - sx0)) * Q + c4], v11 = src[((y1 - sy0) * SR + (x1 - sx0)) * Q + c4];
    float4 o;
    o.x = hy0 * (hx0 * v00.x + hx1 * v01.x) + hy1 * (hx0 * v10.x + hx1 * v11.x);
    o.y = hy0 * (hx0 * v00.y + hx1 * v01.y) + hy1 * (hx0 * v10.y + hx1 * v11.y);
    o.z = hy0 * (hx0 * v00.z + hx1 * v01.z) + hy1 * (hx0 * v10.z + hx1 * v11.z);
    o.w = hy0 * (hx0 * v00.w + hx1 * v01.w) + hy1 * (hx0 * v10.w + hx1 * v11.w);
    stA4<HS>(out, ((size_t)(b * Ho + oy) * Wo + ox) * C + 4 * c4, o);
  }
}

// Adjoint, separable through LDS: one workgroup = 8x8 low-res pixels x 16 channels; the 19x19 high-res pixels that
// can reach them (rows 2i-2 .. 2i+2 per low-res row i) are read ONCE (the gather form above reads every high-res
// pixel from ~4 threads), reduced horizontally, then vertically.  Deterministic (gather, fixed order).
// bn_y != nullptr: g IS the complete gradient w.r.t. relu(bn(y)) of the low-resolution layer (ub1 / ub2 / ub3's input
// comes from ONE producer), so the workgroup also leaves that layer's BatchNorm-backward sums of its 8x8 pixels --
// (sum dz, sum dz*y) per channel, dz = g*[y*scale + shift > 0] -- in row (b, by, bx) of bn_partials ([rows][C][2]); the
// separate reduce pass over (g, y) is not launched (bn_bwd_finalize2 takes these sums).
template <int C, bool HS>
__global__ __launch_bounds__(256) void up2x_bwd_tile_kernel(const float* __restrict__ gu, float* __restrict__ g, int Hin,
                                                            int Win, const float* __restrict__ bn_y,
                                                            const float* __restrict__ bn_scale,
                                                            const float* __restrict__ bn_shift,
                                                            float* __restrict__ bn_partials) {
  SIFSR_CHAIN_PRIO();
  constexpr int TL = 8, HR = 2 * TL + 3, NCH = C / 16;
  __shared__ float4 hi[HR * HR * 4];
  __shared__ float4 tmp[HR * TL * 4];
  __shared__ float wxs[TL][5], wys[TL][5];
  const int tid = threadIdx.x;
  const int Ho = 2 * Hin, Wo = 2 * Win;
  const int ix0 = blockIdx.x * TL, iy0 = blockIdx.y * TL;
  const int b = blockIdx.z / NCH, ch0 = (blockIdx.z % NCH) * 16;
  if (tid < 2 * TL * 5) {
    const bool isy = tid >= TL * 5;
    const int t = isy ? tid - TL * 5 : tid;
    const int il = t / 5, k = t - il * 5;
    const int i = (isy ? iy0 : ix0) + il, in = isy ? Hin : Win, on = isy ? Ho : Wo;
    const int o = 2 * i - 2 + k;
    float w = 0.f;
    if (i < in && o >= 0 && o < on) {
      int a0, a1; float l0, l1;
      up_coord(o, in, on, a0, a1, l0, l1);
      w = (a0 == i ? l0 : 0.f) + (a1 == i ? l1 : 0.f);
    }
    if (isy) wys[il][k] = w; else wxs[il][k] = w;
  }
  for (int e = tid; e < HR * HR * 4; e += 256) {
    const int q = e & 3, p = e >> 2;
    const int py = p / HR, px = p - py * HR;
    const int oy = 2 * iy0 - 2 + py, ox = 2 * ix0 - 2 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) v = ldA4<HS>(gu, ((size_t)(b * Ho + oy) * Wo + ox) * C + ch0 + 4 * q);
    hi[e] = v;
  }
  __syncthreads();
  for (int e = tid; e < HR * TL * 4; e += 256) {
    const int q = e & 3, p = e >> 2;
    const int py = p / TL, il = p - py * TL;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float w = wxs[il][k];
      const float4 v = hi[(py * HR + 2 * il + k) * 4 + q];
      acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
    tmp[e] = acc;
  }
  __syncthreads();
  {
    const int q = tid & 3, p = tid >> 2;
    const int il_y = p / TL, il_x = p - il_y * TL;
    const int iy = iy0 + il_y, ix = ix0 + il_x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float w = wys[il_y][k];
      const float4 v = tmp[((2 * il_y + k) * TL + il_x) * 4 + q];
      acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
    const bool in = iy < Hin && ix < Win;
    acc = as_stored4<HS>(acc);             // the sums below are those of the stored gradient
    if (in) stA4<HS>(g, ((size_t)(b * Hin + iy) * Win + ix) * C + ch0 + 4 * q, acc);
    if (bn_partials != nullptr) {
      float4 d1 = make_float4(0.f, 0.f, 0.f, 0.f), d2 = d1;
      if (in) {
        const float4 yv = ldA4<HS>(bn_y, ((size_t)(b * Hin + iy) * Win + ix) * C + ch0 + 4 * q);
        const float4 sc = ld4(bn_scale + ch0 + 4 * q), sh = ld4(bn_shift + ch0 + 4 * q);
        d1.x = fmaf(yv.x, sc.x, sh.x) > 0.f ? acc.x : 0.f; d2.x = d1.x * yv.x;
        d1.y = fmaf(yv.y, sc.y, sh.y) > 0.f ? acc.y : 0.f; d2.y = d1.y * yv.y;
        d1.z = fmaf(yv.z, sc.z, sh.z) > 0.f ? acc.z : 0.f; d2.z = d1.z * yv.z;
        d1.w = fmaf(yv.w, sc.w, sh.w) > 0.f ? acc.w : 0.f; d2.w = d1.w * yv.w;
      }
      // 64 pixels x 4 quads -> per (quad, component, which sum): a fixed-order tree.  Lanes of a wave hold 16 pixels x 4 quads
      // (quad = lane & 3): xor-shuffles over lane bits 2..5 add the 16 pixels of each quad, then the four waves through LDS.
      // (round 3: this was a 64-step serial sum by 32 threads -- 1.7 us per workgroup of a kernel that otherwise takes 5)
      float v8[8] = {d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w};
#pragma unroll
      for (int m = 4; m < 64; m <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] += __shfl_xor(v8[j], m);
      __syncthreads();                    // tmp[] / hi[] have been consumed by everybody
      float* const red = reinterpret_cast<float*>(hi);
      if ((tid & 63) < 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[((tid >> 6) * 4 + (tid & 3)) * 8 + j] = v8[j];
      }
      __syncthreads();
      if (tid < 32) {                     // (quad qq, component r, which sum)
        const int qq = tid & 3, r = (tid >> 2) & 3, which = tid >> 4;
        const float sum = (red[(0 * 4 + qq) * 8 + which * 4 + r] + red[(1 * 4 + qq) * 8 + which * 4 + r]) +
                          (red[(2 * 4 + qq) * 8 + which * 4 + r] + red[(3 * 4 + qq) * 8 + which * 4 + r]);
        const size_t row = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        bn_partials[(row * C + ch0 + 4 * qq + r) * 2 + which] = sum;
      }
    }
  }
}

inline int grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

int launch_bnrelu_pool2(const float* y, const float* scale, const float* shift, float* out, int B, int H, int W, int C, hipStream_t s) {
  if (H % 2 || W % 2 || C % 4 || (size_t)B * H * W * C >= ((size_t)1 << 32)) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL(bnrelu_pool2_kernel<true>, dim3(grid_for((size_t)B * H / 2 * W / 2 * C / 4)), dim3(256), 0, s, y, scale, shift, out, B, H, W, C);
  else hipLaunchKernelGGL(bnrelu_pool2_kernel<false>, dim3(grid_for((size_t)B * H / 2 * W / 2 * C / 4)), dim3(256), 0, s, y, scale, shift, out, B, H, W, C);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_bnrelu_add(const float* p, const float* y, const float* scale, const float* shift, float* out, int C, size_t npix, hipStream_t s) {
  if (C % 4) return SIFSR_ERR_SHAPE;
  const size_t nq = npix * C / 4;
  if (sifsr_half_storage()) hipLaunchKernelGGL(bnrelu_add_kernel<true>, dim3(grid_for(nq)), dim3(256), 0, s, p, y, scale, shift, out, C, nq);
  else hipLaunchKernelGGL(bnrelu_add_kernel<false>, dim3(grid_for(nq)), dim3(256), 0, s, p, y, scale, shift, out, C, nq);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_bnrelu_up2x(const float* y, const float* scale, const float* shift, float* out, int B, int Hin, int Win, int C, hipStream_t s) {
  if (C % 4) return SIFSR_ERR_SHAPE;
  if ((C == 16 || C == 32 || C == 64) && B <= 65535 && Hin >= 2 && Win >= 2) {
    const dim3 grid((2 * Win + 15) / 16, (2 * Hin + 15) / 16, B);
#define SIFSR_UP(CV, HV) hipLaunchKernelGGL((bnrelu_up2x_tile_kernel<CV, HV>), grid, dim3(256), 0, s, y, scale, shift, out, Hin, Win)
    const bool hs = sifsr_half_storage();
    if (C == 16) { if (hs) SIFSR_UP(16, true); else SIFSR_UP(16, false); }
    else if (C == 32) { if (hs) SIFSR_UP(32, true); else SIFSR_UP(32, false); }
    else { if (hs) SIFSR_UP(64, true); else SIFSR_UP(64, false); }
#undef SIFSR_UP
    SIFSR_LAUNCH_CHECK();
    return SIFSR_OK;
  }
  if (sifsr_half_storage()) hipLaunchKernelGGL(bnrelu_up2x_kernel<true>, dim3(grid_for((size_t)B * Hin * Win * C)), dim3(256), 0, s, y, scale, shift, out, B, Hin, Win, C);
  else hipLaunchKernelGGL(bnrelu_up2x_kernel<false>, dim3(grid_for((size_t)B * Hin * Win * C)), dim3(256), 0, s, y, scale, shift, out, B, Hin, Win, C);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_pool2_bwd(const float* gp, float* g, int B, int H, int W, int C, int accumulate, hipStream_t s) {
  if (H % 2 || W % 2 || C % 4) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL(pool2_bwd_kernel<true>, dim3(grid_for((size_t)B * H * W * C / 4)), dim3(256), 0, s, gp, g, B, H, W, C, accumulate);
  else hipLaunchKernelGGL(pool2_bwd_kernel<false>, dim3(grid_for((size_t)B * H * W * C / 4)), dim3(256), 0, s, gp, g, B, H, W, C, accumulate);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int up2x_bwd_stat_rows(int B, int Hin, int Win, int C) {
  if (!((C == 16 || C == 32 || C == 64) && (size_t)B * (C / 16) <= 65535)) return 0;
  return B * ((Hin + 7) / 8) * ((Win + 7) / 8);
}

int launch_up2x_bwd(const float* gu, float* g, int B, int Hin, int Win, int C, hipStream_t s, const float* bn_y,
                    const float* bn_scale, const float* bn_shift, float* bn_partials) {
  if (C % 4) return SIFSR_ERR_SHAPE;
  if (bn_partials != nullptr && (!bn_y || !bn_scale || !bn_shift || up2x_bwd_stat_rows(B, Hin, Win, C) == 0)) return SIFSR_ERR_ARG;
  if ((C == 16 || C == 32 || C == 64) && (size_t)B * (C / 16) <= 65535) {
    const dim3 grid((Win + 7) / 8, (Hin + 7) / 8, B * (C / 16));
#define SIFSR_UPB(CV, HV) hipLaunchKernelGGL((up2x_bwd_tile_kernel<CV, HV>), grid, dim3(256), 0, s, gu, g, Hin, Win, bn_y, bn_scale, bn_shift, bn_partials)
    const bool hs = sifsr_half_storage();
    if (C == 16) { if (hs) SIFSR_UPB(16, true); else SIFSR_UPB(16, false); }
    else if (C == 32) { if (hs) SIFSR_UPB(32, true); else SIFSR_UPB(32, false); }
    else { if (hs) SIFSR_UPB(64, true); else SIFSR_UPB(64, false); }
#undef SIFSR_UPB
    SIFSR_LAUNCH_CHECK();
    return SIFSR_OK;
  }
  if (sifsr_half_storage()) hipLaunchKernelGGL(up2x_bwd_kernel<true>, dim3(grid_for((size_t)B * Hin * Win * C / 4)), dim3(256), 0, s, gu, g, B, Hin, Win, C);
  else hipLaunchKernelGGL(up2x_bwd_kernel<false>, dim3(grid_for((size_t)B * Hin * Win * C / 4)), dim3(256), 0, s, gu, g, B, Hin, Win, C);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
