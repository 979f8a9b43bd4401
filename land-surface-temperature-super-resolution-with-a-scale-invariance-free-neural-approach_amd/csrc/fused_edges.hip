// Backward of the network's tail in two passes instead of four kernels and five tensor round trips:
//   outlay (Conv2d 16->1 + bias, model.py:605) weight/bias gradient, its input gradient, and the BatchNorm+ReLU
//   backward of the layer that feeds it (ub3.convbloc.bloc.3/4/5).
// The outlay input gradient g[q][ci] = sum_t w[ci][t] * S_t(q) costs 144 FMAs per pixel from 9 values of
// d loss/d sr, so it is recomputed where it is needed instead of being written to and re-read from HBM:
//   pass 1 (tail_bwd_reduce): one read of y  -> outlay dW/db partials (matrix cores) + BN (sum dz, sum dz*xhat) partials
//   pass 2 (tail_bwd_apply):  one read of y  -> dy = scale*dz + k1*y + k0   (one write)
#include "edge_conv.h"

namespace {

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

// Pass 1.  With S_t(q) (above) both contractions are sums over the INPUT pixels q of the same nine values:
//   outlay dW[ci][t] = sum_q a[q][ci] * S_t(q)          g[q][ci] = sum_t w[ci][t] * S_t(q)
// so a work item = (pixel q, channel quad) needs y[q][4 channels] -- one coalesced float4 straight from HBM, no halo of y, no
// LDS tile of it -- and the 3x3 neighbourhood of d loss / d sr around q (LDS, 18x18 per 16x16 tile, double-buffered: one barrier
// per tile).  Per item 20 + 18 packed FMAs (dW accumulators [4][9 (+1)] per lane; g) and the BatchNorm sums dz = g*[z>0]: sum dz,
// sum dz*y.  (Rounds 1-2 ran both contractions on the matrix cores -- 138 v_mfma_f32_16x16x4 per tile with 7 of 16 columns of
// the dW product empty, ~75 us each; an fp32 MFMA moves 32 FMAs per cycle and SIMD, exactly what v_pk_fma_f32 does, and the
// operands of the VALU form need no staging: 169 -> see DESIGN.md section 4.3.)
// Persistent workgroups over 16x16 tiles (the last ones may be partial); the next tile's y quads and d loss / d sr halo are in
// flight while the current tile is processed.  fp32 per lane over the workgroup's tiles (a few dozen terms), float64 across
// lanes, waves and workgroups.
// HS: y stored as bf16 (the bf16 compute mode, common.h).  FULL: H and W are multiples of 16 (no partial tiles: `inside` is true)
template <bool HS, bool FULL>
__global__ __launch_bounds__(256, 3) void tail_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ dsr, const float* __restrict__ w,
                                                                 float* __restrict__ wpart, float* __restrict__ bnpart,
                                                                 int B, int H, int W) {
  __shared__ float dt[2][328];
  __shared__ double dred[4][4][48];       // [wave][channel quad][dW 4 x 9 | db | pad | sum dz x4 | sum dz*y x4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16, ntiles = B * tiles_x * tiles_y;
  const int c4 = tid & 3, pl = tid >> 2;                 // channel quad; pixel slot: items = pixels pl + 64 * k of the tile, k < 4
  const int lx = pl & 15, ly0 = pl >> 4;                 // ... = (row ly0 + 4 k, column lx)

  const float4 sc4 = ld4(scale + 4 * c4), sh4 = ld4(shift + 4 * c4);
  const f32x2 scl = {sc4.x, sc4.y}, sch = {sc4.z, sc4.w}, shl = {sh4.x, sh4.y}, shh = {sh4.z, sh4.w};
  f32x2 wl[9], wh[9];                                    // outlay weight [1][16][3][3]: (w[4 c4 + 0..1][t]), (w[4 c4 + 2..3][t])
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* wo = w + (4 * c4) * 9 + t;
    wl[t] = (f32x2){wo[0], wo[9]}; wh[t] = (f32x2){wo[18], wo[27]};
  }
  f32x2 acc[4][5];                                       // dW[4 c4 + j][2 u, 2 u + 1]  (u = 4: tap 8 | unused)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int u = 0; u < 5; ++u) acc[j][u] = (f32x2){0.f, 0.f};
  f32x2 t1l = {0.f, 0.f}, t1h = t1l, t2l = t1l, t2h = t1l;
  float bacc = 0.f;

  float4 py[4];
  float pd[2];
  size_t nbase = 0;                                      // next tile: element offset of its item 0, row step (0 past the last image row)
  int nrow[4] = {0, 0, 0, 0};
  auto fetch_y = [&](int k) __attribute__((always_inline)) { py[k] = ldA4<HS>(y, nbase + (size_t)nrow[k]); };
  auto fetch = [&](int tl) __attribute__((always_inline)) {
    const int tx = tl % tiles_x, r = tl / tiles_x, ty = r % tiles_y, b = r / tiles_y;
    const int x0 = tx * 16, y0 = ty * 16;
    const int gx = min(x0 + lx, W - 1);
    nbase = ((size_t)(b * H) * W + gx) * 16 + 4 * c4;
#pragma unroll
    for (int k = 0; k < 4; ++k) nrow[k] = min(y0 + ly0 + 4 * k, H - 1) * W * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i;
      const int hy = e / 18, hx = e - hy * 18;
      const int qy = y0 - 1 + hy, qx = x0 - 1 + hx;
      const float v = dsr[(size_t)(b * H + clampi(qy, 0, H - 1)) * W + clampi(qx, 0, W - 1)];   // (no branch: select after the load)
      pd[i] = (e < 324 && qy >= 0 && qy < H && qx >= 0 && qx < W) ? v : 0.f;
    }
  };

  int buf = 0;
  if ((int)blockIdx.x < ntiles) {
    fetch(blockIdx.x);
#pragma unroll
    for (int k = 0; k < 4; ++k) fetch_y(k);
  }
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x, buf ^= 1) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y;
    const int x0 = tx * 16, y0 = ty * 16;
    const float xmf = (x0 + lx == 0) ? 1.f : 0.f, xpf = (x0 + lx == W - 1) ? 1.f : 0.f;
    dt[buf][tid] = pd[0];
    if (tid + 256 < 324) dt[buf][tid + 256] = pd[1];
    __syncthreads();          // (one barrier per tile: the other buffer was last read before the previous barrier)
    // (past the last tile the same tile is fetched again: straight-line code, the compiler keeps the items apart)
    fetch(tl + (int)gridDim.x < ntiles ? tl + (int)gridDim.x : tl);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 yv = py[k];                           // this tile's item k; its register takes the next tile's right away
      fetch_y(k);
      __builtin_amdgcn_sched_barrier(0);                 // (items one after the other: interleaved they need 204 registers)
      const int ly = ly0 + 4 * k;
      const float* D = &dt[buf][(ly + 1) * 18 + lx + 1];
      // S_t(q), t = (ty, tx): the adjoint of the clamp is separable -- tap tx = -1 reads column qx from output columns qx + 1 and,
      // on the first image column, qx itself; tx = +1 from qx - 1 and, on the last column, qx; likewise the rows.  n = the 3x3
      // neighbourhood of d loss / d sr around q (zero outside the image); 12 FMAs, no branch, on every tile.
      float n[3][3], cx[3][3], S[10];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) n[dy][dx] = D[(dy - 1) * 18 + dx - 1];
      const float ymf = (y0 + ly == 0) ? 1.f : 0.f, ypf = (y0 + ly == H - 1) ? 1.f : 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        cx[dy][0] = fmaf(xmf, n[dy][1], n[dy][2]);       // tx = -1
        cx[dy][1] = n[dy][1];
        cx[dy][2] = fmaf(xpf, n[dy][1], n[dy][0]);       // tx = +1
      }
#pragma unroll
      for (int tx_ = 0; tx_ < 3; ++tx_) {
        S[0 + tx_] = fmaf(ymf, cx[1][tx_], cx[2][tx_]);  // ty = -1
        S[3 + tx_] = cx[1][tx_];
        S[6 + tx_] = fmaf(ypf, cx[1][tx_], cx[0][tx_]);  // ty = +1
      }
      S[9] = 0.f;
      const bool inside = FULL || (y0 + ly < H && x0 + lx < W);    // partial tiles: pixels past the image stay out of every sum
      const f32x2 yl = {yv.x, yv.y}, yh = {yv.z, yv.w};
      const f32x2 zl = __builtin_elementwise_fma(yl, scl, shl), zh = __builtin_elementwise_fma(yh, sch, shh);
      const float a[4] = {inside ? fmaxf(zl[0], 0.f) : 0.f, inside ? fmaxf(zl[1], 0.f) : 0.f, inside ? fmaxf(zh[0], 0.f) : 0.f,
                          inside ? fmaxf(zh[1], 0.f) : 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 5; ++u) acc[j][u] = __builtin_elementwise_fma((f32x2){a[j], a[j]}, (f32x2){S[2 * u], S[2 * u + 1]}, acc[j][u]);
      f32x2 gl = {0.f, 0.f}, gh = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        gl = __builtin_elementwise_fma(wl[t], (f32x2){S[t], S[t]}, gl);
        gh = __builtin_elementwise_fma(wh[t], (f32x2){S[t], S[t]}, gh);
      }
      const f32x2 dl = {(inside && zl[0] > 0.f) ? gl[0] : 0.f, (inside && zl[1] > 0.f) ? gl[1] : 0.f};
      const f32x2 dh = {(inside && zh[0] > 0.f) ? gh[0] : 0.f, (inside && zh[1] > 0.f) ? gh[1] : 0.f};
      t1l += dl; t1h += dh;
      t2l = __builtin_elementwise_fma(dl, yl, t2l); t2h = __builtin_elementwise_fma(dh, yh, t2h);
      bacc += D[0];                                      // (every channel quad sums d loss / d sr: quad 0's sum is the bias gradient)
    }
  }

  // ---- across the 16 lanes of a channel quad (lane = 4 * pixel slot + c4), in float64; then waves, then workgroups
  auto quad_sum = [&](float v) __attribute__((always_inline)) {
    double d = (double)v;
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) d += __shfl_xor(d, m);
    __builtin_amdgcn_sched_barrier(0);                   // (one reduction after the other: 45 interleaved ones set the kernel's register count)
    return d;
  };
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const double d = quad_sum(acc[j][t >> 1][t & 1]);
      if (lane < 4) dred[wave][c4][j * 9 + t] = d;
    }
  {
    const double d = quad_sum(bacc);
    if (lane < 4) dred[wave][c4][36] = d;
    const float t1[4] = {t1l[0], t1l[1], t1h[0], t1h[1]}, t2[4] = {t2l[0], t2l[1], t2h[0], t2h[1]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double d1 = quad_sum(t1[j]), d2 = quad_sum(t2[j]);
      if (lane < 4) { dred[wave][c4][40 + j] = d1; dred[wave][c4][44 + j] = d2; }
    }
  }
  __syncthreads();
  auto wsum = [&](int q, int i) __attribute__((always_inline)) { return dred[0][q][i] + dred[1][q][i] + dred[2][q][i] + dred[3][q][i]; };
  if (tid < 144) {
    const int ci = tid / 9, t = tid % 9;
    wpart[(size_t)blockIdx.x * 145 + tid] = (float)wsum(ci >> 2, (ci & 3) * 9 + t);
  } else if (tid == 144) {
    wpart[(size_t)blockIdx.x * 145 + 144] = (float)wsum(0, 36);
  } else if (tid >= 192 && tid < 208) {
    // channel c: sum dz and sum dz*xhat = invstd * (sum dz*y - mean * sum dz), in float64
    const int c = tid - 192, q = c >> 2, j = c & 3;
    const double s1 = wsum(q, 40 + j), s2 = wsum(q, 44 + j);
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
  const bool full = H % 16 == 0 && W % 16 == 0;
#define TAIL_REDUCE(HS_, FULL_) hipLaunchKernelGGL((tail_bwd_reduce_kernel<HS_, FULL_>), dim3(nblk), dim3(256), 0, s, y, scale, shift, mean, invstd, dsr, w, wpart, bnpart, B, H, W)
  if (sifsr_half_storage()) { if (full) TAIL_REDUCE(true, true); else TAIL_REDUCE(true, false); }
  else { if (full) TAIL_REDUCE(false, true); else TAIL_REDUCE(false, false); }
#undef TAIL_REDUCE
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
