// Backward of the network's tail in two passes instead of four kernels and five tensor round trips:
//   outlay (Conv2d 16->1 + bias, model.py:605) weight/bias gradient, its input gradient, and the BatchNorm+ReLU
//   backward of the layer that feeds it (ub3.convbloc.bloc.3/4/5).
// The outlay input gradient g[q][ci] = sum_t w[ci][t] * S_t(q) costs 144 FMAs per pixel from 9 values of
// d loss/d sr, so it is recomputed where it is needed instead of being written to and re-read from HBM:
//   pass 1 (tail_bwd_reduce): one read of y  -> outlay dW/db partials (matrix cores) + BN (sum dz, sum dz*xhat) partials
//   pass 2 (tail_bwd_apply):  one read of y  -> dy = scale*dz + k1*y + k0   (one write)
#include "edge_conv.h"

namespace {

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

__global__ __launch_bounds__(256) void tail_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ scale,
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
  __shared__ double dred[256][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = W / 16, tiles_y = H / 16, ntiles = B * tiles_x * tiles_y;
  // matrix-core part (outlay dW): A lane (i = ci, k = halo pixel), B lane (j = tap, k = halo pixel)
  const int i16 = lane & 15, k = lane >> 4;
  const int tty = i16 < 9 ? i16 / 3 : 100, ttx = i16 < 9 ? i16 % 3 : 100;
  const float sci = scale[i16], shi = shift[i16];
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;
  // BatchNorm part: thread = (channel quad c4, pixel lane pl); four passes of 64 pixels per tile
  const int c4 = tid & 3, pl = tid >> 2;
  const float4 sc = ld4(scale + 4 * c4), sh = ld4(shift + 4 * c4), mu = ld4(mean + 4 * c4), is = ld4(invstd + 4 * c4);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
  float wr[4][9];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = w[(4 * c4 + j) * 9 + t];
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};

  for (int e = tid; e < 18 * 20 * OCS; e += 256) tile[e] = 0.f;
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int x0 = tx * 16, y0 = ty * 16;
    __syncthreads();
    for (int e = tid; e < 324 * 4; e += 256) {
      const int p = e >> 2, q4 = e & 3;
      const int py = p / 18, px = p - py * 18;
      const int gy = clampi(y0 - 1 + py, 0, H - 1), gx = clampi(x0 - 1 + px, 0, W - 1);
      *reinterpret_cast<float4*>(&tile[(py * 20 + px) * OCS + 4 * q4]) = ld4(y + ((size_t)(b * H + gy) * W + gx) * 16 + 4 * q4);
    }
    stage_dsr_halo(dt, dsr, b, y0, x0, H, W, tid);
    __syncthreads();
    bacc += dt[((tid >> 4) + 1) * 18 + (tid & 15) + 1];
    for (int ks = wave; ks < 90; ks += 4) {          // 18 rows x 5 quads of the padded halo tile
      const int row = ks / 5, col = (ks - row * 5) * 4 + k;
      const float av = fmaxf(fmaf(tile[(row * 20 + col) * OCS + i16], sci, shi), 0.f);
      const int oy = row - tty, ox = col - ttx;
      const float bv = (oy >= 0 && oy < 16 && ox >= 0 && ox < 16) ? dt[(oy + 1) * 18 + ox + 1] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int p = pass * 64 + pl, ly = p >> 4, lx = p & 15;
      float S[9];
      outlay_gather(dt, ly, lx, y0 + ly == 0, y0 + ly == H - 1, x0 + lx == 0, x0 + lx == W - 1, S);
      const float4 yv = *reinterpret_cast<const float4*>(&tile[((ly + 1) * 20 + lx + 1) * OCS + 4 * c4]);
      const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float g = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) g = fmaf(wr[j][t], S[t], g);
        const float dz = fmaf(yy[j], scv[j], shv[j]) > 0.f ? g : 0.f;
        a1[j] += (double)dz;
        a2[j] += (double)dz * (double)((yy[j] - muv[j]) * isv[j]);
      }
    }
  }
  // outlay dW [ci][t] and db
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * k + r) * 16 + i16] = acc[r];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) bacc += __shfl_xor(bacc, m);
  if (lane == 0) bsum[wave] = bacc;
#pragma unroll
  for (int j = 0; j < 4; ++j) { dred[tid][j] = a1[j]; dred[tid][4 + j] = a2[j]; }
  __syncthreads();
  if (tid < 144) {
    const int ci = tid / 9, t = tid % 9;
    wpart[(size_t)blockIdx.x * 145 + tid] = red[0][ci * 16 + t] + red[1][ci * 16 + t] + red[2][ci * 16 + t] + red[3][ci * 16 + t];
  } else if (tid == 144) {
    wpart[(size_t)blockIdx.x * 145 + 144] = bsum[0] + bsum[1] + bsum[2] + bsum[3];
  }
  for (int st = 32; st > 0; st >>= 1) {
    if (pl < st) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dred[tid][j] += dred[tid + st * 4][j];
    }
    __syncthreads();
  }
  if (pl == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bnpart[((size_t)blockIdx.x * 16 + 4 * c4 + j) * 2 + 0] = (float)dred[tid][j];
      bnpart[((size_t)blockIdx.x * 16 + 4 * c4 + j) * 2 + 1] = (float)dred[tid][4 + j];
    }
  }
}

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
    const size_t off = ((size_t)(b * H + y0 + ly) * W + x0 + lx) * 16 + 4 * c4;
    const float4 yv = ld4(y + off);
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
    st4(dy + off, make_float4(o[0], o[1], o[2], o[3]));
  }
}

}  // namespace

int launch_tail_bwd_reduce(const float* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                           const float* dsr, const float* w, float* wpart, float* bnpart, int nblk, int B, int H, int W,
                           hipStream_t s) {
  if (H % 16 || W % 16 || nblk < 1) return SIFSR_ERR_SHAPE;
  hipLaunchKernelGGL(tail_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, s, y, scale, shift, mean, invstd, dsr, w, wpart,
                     bnpart, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_tail_bwd_apply(const float* y, const float* scale, const float* shift, const double* coef, const float* dsr,
                          const float* w, float* dy, int B, int H, int W, hipStream_t s) {
  if (H % 16 || W % 16) return SIFSR_ERR_SHAPE;
  hipLaunchKernelGGL(tail_bwd_apply_kernel, dim3(W / 16, H / 16, B), dim3(256), 0, s, y, scale, shift, coef, dsr, w, dy, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
