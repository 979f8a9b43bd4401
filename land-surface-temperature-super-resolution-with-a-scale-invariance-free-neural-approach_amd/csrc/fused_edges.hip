// Backward of the network's tail in two passes instead of four kernels and five tensor round trips:
//   outlay (Conv2d 16->1 + bias, model.py:605) weight/bias gradient, its input gradient, and the BatchNorm+ReLU
//   backward of the layer that feeds it (ub3.convbloc.bloc.3/4/5).
// The outlay input gradient g[q][ci] = sum_t w[ci][t] * S_t(q) costs 144 FMAs per pixel from 9 values of
// d loss/d sr, so it is recomputed where it is needed instead of being written to and re-read from HBM:
//   pass 1 (tail_bwd_reduce): one read of y  -> outlay dW/db partials (matrix cores) + BN (sum dz, sum dz*xhat) partials
//   pass 2 (tail_bwd_apply):  one read of y  -> dy = scale*dz + k1*y + k0   (one write)
#include "edge_conv.h"

namespace {

#define TAIL_ABL 0
constexpr int OCS = 20;   // LDS pixel stride of the y halo tile (floats): conflict-free b128 reads

// S_t(q) = sum of dsr[p] over the output pixels p whose (clamped) tap t reads input pixel q -- the adjoint of
// replicate padding.  dt = 18x18 LDS tile of dsr around the 16x16 tile of q, ZERO outside the image;
// (ly, lx) = q inside the tile; ym/yp/xm/xp: q lies on the first/last image row/column.
__device__ __forceinline__ void outlay_gather(const float* dt, int ly, int lx, bool ym, bool yp, bool xm, bool xp,
                                              float S[9]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ty = t / 3 - 1, tx = t % 3 - 1;
    const int ry = ly + 1 - ty, rx = lx + 1 - tx;
    float s = dt[ry * 18 + rx];
    const bool cy = (ty == -1 && ym) || (ty == 1 && yp);
    const bool cx = (tx == -1 && xm) || (tx == 1 && xp);
    if (cx) s += dt[ry * 18 + lx + 1];
    if (cy) s += dt[(ly + 1) * 18 + rx];
    if (cx && cy) s += dt[(ly + 1) * 18 + lx + 1];
    S[t] = s;
  }
}

__device__ __forceinline__ void stage_dsr_halo(float* dt, const float* __restrict__ dsr, int b, int y0, int x0, int H,
                                               int W, int tid) {
  for (int e = tid; e < 324; e += 256) {
    const int py = e / 18, px = e - py * 18;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    dt[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? dsr[(size_t)(b * H + gy) * W + gx] : 0.f;
  }
}

// One S value for a runtime tap t -- the B operand of the matrix-core form of the outlay input gradient.
// `off` = the interior offset (1 - ty) * 18 + (1 - tx) of tap t, precomputed per lane.
__device__ __forceinline__ float outlay_S1(const float* dt, int ly, int lx, int t, int off, bool border, int gy, int gx,
                                           int H, int W) {
  float s = dt[ly * 18 + lx + off];
  if (border) {
    const int ty = t / 3 - 1, tx = t - (ty + 1) * 3 - 1;
    const int ry = ly + 1 - ty, rx = lx + 1 - tx;
    const bool cy = (ty == -1 && gy == 0) || (ty == 1 && gy == H - 1);
    const bool cx = (tx == -1 && gx == 0) || (tx == 1 && gx == W - 1);
    if (cx) s += dt[ry * 18 + lx + 1];
    if (cy) s += dt[(ly + 1) * 18 + rx];
    if (cx && cy) s += dt[(ly + 1) * 18 + lx + 1];
  }
  return s;
}

// Pass 1.  Persistent workgroups over 16x16 tiles, software-pipelined (next tile's halo in registers while the
// current one is processed out of LDS; <= 128 VGPRs so four workgroups per CU keep enough bytes in flight).
// Two matrix-core contractions per tile:
//   outlay dW[ci][t]  = sum_{halo px p'} a[p'][ci] * dsr[p' - t]        (A lane (ci, p'), B lane (t, p'))
//   g[ci][q]          = sum_t w[ci][t] * S_t(q)                          (A lane (ci, t),  B lane (q, t))
// the D fragment of the second (4 channels x 1 pixel per lane) is consumed in registers by the BatchNorm
// reduction: dz = g*[z>0]; sum dz and sum dz*y per channel (fp32 per lane over the workgroup's tiles -- a few
// hundred terms -- float64 across lanes, waves and workgroups).
template <bool HS>   // HS: y stored as bf16 (the bf16 compute mode, common.h)
__global__ __launch_bounds__(256, 2) void tail_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ dsr, const float* __restrict__ w,
                                                                 float* __restrict__ wpart, float* __restrict__ bnpart,
                                                                 int B, int H, int W) {
  __shared__ float tile[18 * 20 * OCS];   // RAW y halo tile, [row][col (18 + 2 pad)][OCS]
  __shared__ float dt[324];
  __shared__ float red[4][256];
  __shared__ float bsum[4];
  __shared__ double dred[4][4][8];        // [wave][channel quad][sum dz x4 | sum dz*y x4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ntiles = B * tiles_x * tiles_y;   // last ones may be partial
  const int i16 = lane & 15, k = lane >> 4;

  // ---- outlay dW operands
  const int tty = i16 < 9 ? i16 / 3 : 100, ttx = i16 < 9 ? i16 % 3 : 100;
  const float sci = scale[i16], shi = shift[i16];
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;

  // ---- g operands: A = w[ci = i16][t = 4*kk + k] (t >= 9 -> 0); B tap of this lane per k-step kk
  float wa[3];
  int boff[3];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int t = 4 * kk + k;
    wa[kk] = t < 9 ? w[i16 * 9 + t] : 0.f;
    boff[kk] = t < 9 ? (2 - t / 3) * 18 + (2 - t % 3) : 19;
  }
  // BatchNorm: this lane owns channels 4k..4k+3 (the D rows) of pixel column i16
  const float4 sc = ld4(scale + 4 * k), sh = ld4(shift + 4 * k);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};

  // ---- staging: element e = tid + 256*i of the 18 x 18 x 4 float4 halo (i < 6), e of the 18 x 18 dsr halo (i < 2)
  float4 py[6];
  float pd[2];
  auto fetch = [&](int tl) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int x0 = tx * 16, y0 = ty * 16;
    const bool border = tx == 0 || ty == 0 || tx == tiles_x - 1 || ty == tiles_y - 1;
    if (!border) {
      const size_t yb = ((size_t)(b * H + y0 - 1) * W + x0 - 1) * 16;
      const float* db = dsr + (size_t)(b * H + y0 - 1) * W + x0 - 1;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int e = tid + 256 * i;
        if (e < 324 * 4) {
          const int hy = (e >> 2) / 18;
          py[i] = ldA4<HS>(y, yb + (size_t)(hy * (W - 18) * 16 + e * 4));   // ((hy*W + hx)*16 + 4*q4), e = (hy*18 + hx)*4 + q4
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        if (e < 324) pd[i] = db[(e / 18) * (W - 18) + e];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int e = tid + 256 * i;
        if (e < 324 * 4) {
          const int p = e >> 2, hy = p / 18, hx = p - hy * 18;
          const int gy = clampi(y0 - 1 + hy, 0, H - 1), gx = clampi(x0 - 1 + hx, 0, W - 1);
          py[i] = ldA4<HS>(y, ((size_t)(b * H + gy) * W + gx) * 16 + 4 * (e & 3));
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        if (e < 324) {
          const int hy = e / 18, hx = e - hy * 18;
          const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
          pd[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? dsr[(size_t)(b * H + gy) * W + gx] : 0.f;
        }
      }
    }
  };

  for (int e = tid; e < 18 * 20 * OCS; e += 256) tile[e] = 0.f;
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y;
    const int x0 = tx * 16, y0 = ty * 16;
    const bool border = tx == 0 || ty == 0 || tx == tiles_x - 1 || ty == tiles_y - 1;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int e = tid + 256 * i;
      if (e < 324 * 4) {
        const int hy = (e >> 2) / 18;
        // LDS offset (hy*20 + hx)*OCS + 4*q4 with e = (hy*18 + hx)*4 + q4:  5*e - q4 + 40*hy
        *reinterpret_cast<float4*>(&tile[5 * e - (e & 3) + 40 * hy]) = py[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (tid + 256 * i < 324) dt[tid + 256 * i] = pd[i];
    __syncthreads();
    if (tl + (int)gridDim.x < ntiles) fetch(tl + gridDim.x);

    bacc += dt[((tid >> 4) + 1) * 18 + (tid & 15) + 1];
    // outlay dW: this wave takes halo rows wave, wave+4, ...
#pragma unroll 1
    for (int row = (TAIL_ABL & 1) ? 100 : wave; row < 18; row += 4) {
      const int oy = row - tty;
      const bool vy = oy >= 0 && oy < 16;
      const float* arow = &tile[row * 20 * OCS + k * OCS + i16];
      const float* brow = &dt[(oy + 1) * 18 + k - ttx + 1];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const float av = fmaxf(fmaf(arow[4 * q * OCS], sci, shi), 0.f);
        const int ox = 4 * q + k - ttx;
        const bool vx = (q == 0) ? ox >= 0 : (q == 4 ? (ox < 16 && ttx < 3) : ttx < 3);
        const float bv = (vy && vx) ? brow[4 * q] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
      }
    }
    // g on the matrix cores, BatchNorm sums from the D fragment: this wave takes tile rows 4*wave .. 4*wave+3
#pragma unroll 2
    for (int gi = (TAIL_ABL & 2) ? 4 : 0; gi < 4; ++gi) {
      const int ly = 4 * wave + gi, lx = i16;
      f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        float bv = outlay_S1(dt, ly, lx, 4 * kk + k, boff[kk], border && 4 * kk + k < 9, y0 + ly, x0 + lx, H, W);
        if (kk == 2 && k != 0) bv = 0.f;
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kk], bv, g, 0, 0, 0);
      }
      const float4 yv = *reinterpret_cast<const float4*>(&tile[((ly + 1) * 20 + lx + 1) * OCS + 4 * k]);
      const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
      const bool inside = y0 + ly < H && x0 + lx < W;      // partial tiles: pixels past the image stay out of the sums
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dz = (inside && fmaf(yy[j], scv[j], shv[j]) > 0.f) ? g[j] : 0.f;
        t1[j] += dz;
        t2[j] = fmaf(dz, yy[j], t2[j]);
      }
    }
  }

  // outlay dW [ci][t] and db
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * k + r) * 16 + i16] = acc[r];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) bacc += __shfl_xor(bacc, m);
  if (lane == 0) bsum[wave] = bacc;
  // BatchNorm sums: reduce over the 16 pixel lanes of each channel quad, then over the 4 waves
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double a1 = (double)t1[j], a2 = (double)t2[j];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
    if (i16 == 0) { dred[wave][k][j] = a1; dred[wave][k][4 + j] = a2; }
  }
  __syncthreads();
  if (tid < 144) {
    const int ci = tid / 9, t = tid % 9;
    wpart[(size_t)blockIdx.x * 145 + tid] = red[0][ci * 16 + t] + red[1][ci * 16 + t] + red[2][ci * 16 + t] + red[3][ci * 16 + t];
  } else if (tid == 144) {
    wpart[(size_t)blockIdx.x * 145 + 144] = bsum[0] + bsum[1] + bsum[2] + bsum[3];
  } else if (tid >= 192 && tid < 208) {
    // channel c: sum dz and sum dz*xhat = invstd * (sum dz*y - mean * sum dz), in float64
    const int c = tid - 192, q = c >> 2, j = c & 3;
    const double s1 = dred[0][q][j] + dred[1][q][j] + dred[2][q][j] + dred[3][q][j];
    const double s2 = dred[0][q][4 + j] + dred[1][q][4 + j] + dred[2][q][4 + j] + dred[3][q][4 + j];
    bnpart[((size_t)blockIdx.x * 16 + c) * 2 + 0] = (float)s1;
    bnpart[((size_t)blockIdx.x * 16 + c) * 2 + 1] = (float)((double)invstd[c] * (s2 - (double)mean[c] * s1));
  }
}

template <bool HS>   // HS: y read and dy written as bf16
__global__ __launch_bounds__(256) void tail_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const double* __restrict__ coef,
                                                             const float* __restrict__ dsr, const float* __restrict__ w,
                                                             float* __restrict__ dy, int H, int W) {
  __shared__ float dt[324];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16, b = blockIdx.z;
  stage_dsr_halo(dt, dsr, b, y0, x0, H, W, tid);
  const int c4 = tid & 3, pl = tid >> 2;
  const float4 sc = ld4(scale + 4 * c4), sh = ld4(shift + 4 * c4);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  double sd[4], k1[4], k0[4];
  float wr[4][9];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sd[j] = coef[4 * c4 + j]; k1[j] = coef[16 + 4 * c4 + j]; k0[j] = coef[32 + 4 * c4 + j];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = w[(4 * c4 + j) * 9 + t];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int p = pass * 64 + pl, ly = p >> 4, lx = p & 15;
    if (y0 + ly >= H || x0 + lx >= W) continue;
    const size_t off = ((size_t)(b * H + y0 + ly) * W + x0 + lx) * 16 + 4 * c4;
    const float4 yv = ldA4<HS>(y, off);
    float S[9];
    outlay_gather(dt, ly, lx, y0 + ly == 0, y0 + ly == H - 1, x0 + lx == 0, x0 + lx == W - 1, S);
    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) g = fmaf(wr[j][t], S[t], g);
      const float dz = fmaf(yy[j], scv[j], shv[j]) > 0.f ? g : 0.f;
      o[j] = (float)fma(sd[j], (double)dz, fma(k1[j], (double)yy[j], k0[j]));
    }
    stA4<HS>(dy, off, make_float4(o[0], o[1], o[2], o[3]));
  }
}

}  // namespace

int launch_tail_bwd_reduce(const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                           const float* dsr, const float* w, float* wpart, float* bnpart, int nblk, int B, int H, int W,
                           hipStream_t s) {
  if (H < 3 || W < 3 || nblk < 1) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL(tail_bwd_reduce_kernel<true>, dim3(nblk), dim3(256), 0, s, y, scale, shift, mean, invstd, dsr, w, wpart, bnpart, B, H, W);
  else hipLaunchKernelGGL(tail_bwd_reduce_kernel<false>, dim3(nblk), dim3(256), 0, s, y, scale, shift, mean, invstd, dsr, w, wpart, bnpart, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_tail_bwd_apply(const float* y, const float* scale, const float* shift, const double* coef, const float* dsr,
                          const float* w, float* dy, int B, int H, int W, hipStream_t s) {
  if (H < 3 || W < 3) return SIFSR_ERR_SHAPE;
  if (sifsr_half_storage()) hipLaunchKernelGGL(tail_bwd_apply_kernel<true>, dim3((W + 15) / 16, (H + 15) / 16, B), dim3(256), 0, s, y, scale, shift, coef, dsr, w, dy, H, W);
  else hipLaunchKernelGGL(tail_bwd_apply_kernel<false>, dim3((W + 15) / 16, (H + 15) / 16, B), dim3(256), 0, s, y, scale, shift, coef, dsr, w, dy, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
