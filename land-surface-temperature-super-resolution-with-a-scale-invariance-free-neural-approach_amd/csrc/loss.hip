// SIF loss operators on (B,1,H,W) fp32 images (NCHW with one channel == plain 2-D images):
//   * get_output_ftm           (utils.py:1833-1860): reflect-pad 4, 9x9 Gaussian PSF, crop
//                              == separable 9-tap Gaussian with reflect border;
//   * downscale_LST_SR_to_LR   (utils.py:1671-1706, deci_type='bic'): the same blur, then
//                              F.interpolate(scale 1/4, bicubic, A=-0.75) and crop [1:65]
//                              == 4-tap [-3/32, 19/32, 19/32, -3/32] per axis at stride 4 from pixel 0;
//   * the 4-filter Sobel bank  (train_model_B_predef_filters.py:38-42,120-128), zero padding;
//   * nn.HuberLoss(delta=1, mean) (train_model_B_gradFTM.py:454) and its gradient;
//   * the fused SR2 / SR1 losses with their gradient w.r.t. the prediction
//     (train_model_B_gradFTM.py:99-117, train_model_B_predef_filters.py:111-133).
// One workgroup = 256 threads = one 32x32 tile; halo tiles live in LDS; every adjoint is written in
// gather form, so all results are order-stable (no float atomics).
#include "loss.h"

namespace {

constexpr int T = 32;          // tile edge
constexpr int R = 4;           // Gaussian half width
constexpr int TP = T + 2 * R;  // 40
constexpr int LS = TP + 1;     // LDS row stride

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }
__device__ __forceinline__ float huber_val(float e) { const float a = fabsf(e); return a < 1.f ? 0.5f * e * e : a - 0.5f; }
__device__ __forceinline__ float clamp1(float e) { return fminf(fmaxf(e, -1.f), 1.f); }

struct Taps { float w[9]; };
__device__ __forceinline__ float k4(int a) { return (a == 0 || a == 3) ? -0.09375f : 0.59375f; }

// L[py][px] = a*img[refl(y0-4+py)][refl(x0-4+px)] + b   (reflect-padded halo tile)
__device__ __forceinline__ void load_tile_reflect(float* L, const float* __restrict__ img, int H, int W, int y0,
                                                  int x0, float a, float b, int tid) {
  for (int e = tid; e < TP * TP; e += 256) {
    const int py = e / TP, px = e - py * TP;
    // partial tiles reach past the image: clamp to the last halo coordinate (n-1+R) before reflecting; the LDS entries
    // beyond it only feed outputs that are never stored
    const int gy = refl(min(y0 - R + py, H - 1 + R), H), gx = refl(min(x0 - R + px, W - 1 + R), W);
    L[py * LS + px] = fmaf(a, img[(size_t)gy * W + gx], b);
  }
}
// zero-padded halo tile
__device__ __forceinline__ void load_tile_zero(float* L, const float* __restrict__ img, int H, int W, int y0, int x0,
                                               int tid) {
  for (int e = tid; e < TP * TP; e += 256) {
    const int py = e / TP, px = e - py * TP;
    const int gy = y0 - R + py, gx = x0 - R + px;
    L[py * LS + px] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(size_t)gy * W + gx] : 0.f;
  }
}

// horizontal correlation of all TP rows: M[py][x] = sum_t w_t L[py][x + 4 + t],  x in [0,T)
__device__ __forceinline__ void hpass(const float* L, float* M, const Taps& k, int tid) {
  for (int e = tid; e < TP * T; e += 256) {
    const int py = e / T, x = e - py * T;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) s = fmaf(k.w[t], L[py * LS + x + t], s);
    M[py * LS + x] = s;
  }
}
// vertical correlation: value at tile pixel (y, x)
__device__ __forceinline__ float vpass_at(const float* M, const Taps& k, int y, int x) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) s = fmaf(k.w[t], M[(y + t) * LS + x], s);
  return s;
}

// ---- adjoint of the reflect-border blur, gather form ------------------------------------------
// 1-D: out[q] = sum_t w_t v[q-t] + [1<=q<=4] c[-q] + [n-5<=q<=n-2] c[2(n-1)-q],  c[u] = sum_t w_t v[u-t]
// (v zero outside [0,n)).  `line` points at tile coordinate 0 of a zero-padded halo line (global
// coordinate g0 - 4), `stride` is the element stride along the line.
__device__ __forceinline__ float adj_extra(const float* line, int stride, const Taps& k, int q, int g0, int n) {
  float s = 0.f;
  int u = 0;
  bool has = false;
  if (q >= 1 && q <= R) { u = -q; has = true; }
  else if (q >= n - 1 - R && q <= n - 2) { u = 2 * (n - 1) - q; has = true; }
  if (has) {
    const int lo = max(0, u - R), hi = min(n - 1, u + R);
    for (int p = lo; p <= hi; ++p) s = fmaf(k.w[u - p + R], line[(p - g0 + R) * stride], s);
  }
  return s;
}
// horizontal adjoint pass on all TP rows of a zero-padded tile V -> M
__device__ __forceinline__ void hpass_adj(const float* V, float* M, const Taps& k, int x0, int W, int tid) {
  for (int e = tid; e < TP * T; e += 256) {
    const int py = e / T, x = e - py * T;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) s = fmaf(k.w[t], V[py * LS + x + t], s);   // symmetric taps
    s += adj_extra(V + py * LS, 1, k, x0 + x, x0, W);
    M[py * LS + x] = s;
  }
}
__device__ __forceinline__ float vpass_adj_at(const float* M, const Taps& k, int y, int x, int y0, int H) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) s = fmaf(k.w[t], M[(y + t) * LS + x], s);
  s += adj_extra(M + x, LS, k, y0 + y, y0, H);
  return s;
}

// ------------------------------------------------------------------------------------------------
// unfused operators
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blur_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, Taps k,
                                                       int H, int W) {
  __shared__ float L[TP * LS], M[TP * LS];
  const int tid = threadIdx.x, x0 = blockIdx.x * T, y0 = blockIdx.y * T;
  const size_t img = (size_t)blockIdx.z * H * W;
  load_tile_reflect(L, x + img, H, W, y0, x0, 1.f, 0.f, tid);
  __syncthreads();
  hpass(L, M, k, tid);
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = ty + 8 * r;
    if (y0 + y < H && x0 + tx < W) out[img + (size_t)(y0 + y) * W + x0 + tx] = vpass_at(M, k, y, tx);
  }
}

__global__ __launch_bounds__(256) void blur_bwd_kernel(const float* __restrict__ g, float* __restrict__ out, Taps k,
                                                       int H, int W) {
  __shared__ float L[TP * LS], M[TP * LS];
  const int tid = threadIdx.x, x0 = blockIdx.x * T, y0 = blockIdx.y * T;
  const size_t img = (size_t)blockIdx.z * H * W;
  load_tile_zero(L, g + img, H, W, y0, x0, tid);
  __syncthreads();
  hpass_adj(L, M, k, x0, W, tid);
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = ty + 8 * r;
    if (y0 + y < H && x0 + tx < W) out[img + (size_t)(y0 + y) * W + x0 + tx] = vpass_adj_at(M, k, y, tx, y0, H);
  }
}

// lr[i][j] = sum_{a,b} k4[a] k4[b] G(x)[4i+a][4j+b]
__global__ __launch_bounds__(256) void blurdec_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, Taps k,
                                                          int H, int W) {
  __shared__ float L[TP * LS], M[TP * LS];
  const int tid = threadIdx.x, x0 = blockIdx.x * T, y0 = blockIdx.y * T;
  const size_t img = (size_t)blockIdx.z * H * W;
  load_tile_reflect(L, x + img, H, W, y0, x0, 1.f, 0.f, tid);
  __syncthreads();
  hpass(L, M, k, tid);
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
  float* Bt = L;   // reuse: blurred tile [T][LS]
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = vpass_at(M, k, ty + 8 * r, tx);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) Bt[(ty + 8 * r) * LS + tx] = v[r];
  __syncthreads();
  if (tid < 64) {
    const int i = tid >> 3, j = tid & 7;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s = fmaf(k4(a) * k4(b), Bt[(4 * i + a) * LS + 4 * j + b], s);
    if (y0 / 4 + i < H / 4 && x0 / 4 + j < W / 4)
      out[(size_t)blockIdx.z * (H / 4) * (W / 4) + (size_t)(y0 / 4 + i) * (W / 4) + x0 / 4 + j] = s;
  }
}

// zero-padded halo tile of D^T glr:  v[y][x] = k4[y%4] k4[x%4] glr[y/4][x/4]
__device__ __forceinline__ void load_tile_dect(float* L, const float* __restrict__ glr, int H, int W, int y0, int x0,
                                               int tid) {
  for (int e = tid; e < TP * TP; e += 256) {
    const int py = e / TP, px = e - py * TP;
    const int gy = y0 - R + py, gx = x0 - R + px;
    L[py * LS + px] = (gy >= 0 && gy < H && gx >= 0 && gx < W)
                          ? k4(gy & 3) * k4(gx & 3) * glr[(size_t)(gy >> 2) * (W / 4) + (gx >> 2)] : 0.f;
  }
}

__global__ __launch_bounds__(256) void blurdec_bwd_kernel(const float* __restrict__ glr, float* __restrict__ out,
                                                          Taps k, int H, int W) {
  __shared__ float L[TP * LS], M[TP * LS];
  const int tid = threadIdx.x, x0 = blockIdx.x * T, y0 = blockIdx.y * T;
  load_tile_dect(L, glr + (size_t)blockIdx.z * (H / 4) * (W / 4), H, W, y0, x0, tid);
  __syncthreads();
  hpass_adj(L, M, k, x0, W, tid);
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
  const size_t img = (size_t)blockIdx.z * H * W;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = ty + 8 * r;
    if (y0 + y < H && x0 + tx < W) out[img + (size_t)(y0 + y) * W + x0 + tx] = vpass_adj_at(M, k, y, tx, y0, H);
  }
}

// Sobel bank, cross-correlation with zero padding.  filters f0..f3 (row-major 3x3):
//  f0 = [1 2 1; 0 0 0; -1 -2 -1], f1 = [1 0 -1; 2 0 -2; 1 0 -1], f2 = [2 1 0; 1 0 -1; 0 -1 -2], f3 = [0 1 2; -1 0 1; -2 -1 0]
__device__ __forceinline__ void sobel4(const float n[9], float o[4]) {
  o[0] = (n[0] + 2.f * n[1] + n[2]) - (n[6] + 2.f * n[7] + n[8]);
  o[1] = (n[0] + 2.f * n[3] + n[6]) - (n[2] + 2.f * n[5] + n[8]);
  o[2] = (2.f * n[0] + n[1] + n[3]) - (n[5] + n[7] + 2.f * n[8]);
  o[3] = (n[1] + 2.f * n[2] + n[5]) - (n[3] + 2.f * n[6] + n[7]);
}
__device__ __forceinline__ void neigh9_zero(const float* __restrict__ img, int H, int W, int y, int x, float n[9]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    n[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(size_t)yy * W + xx] : 0.f;
  }
}

__global__ void sobel_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W) {
  const size_t n = (size_t)B * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int xx = e % W, yy = (e / W) % H;
    const size_t b = e / ((size_t)W * H);
    float nb[9], o[4];
    neigh9_zero(x + b * H * W, H, W, yy, xx, nb);
    sobel4(nb, o);
#pragma unroll
    for (int f = 0; f < 4; ++f) out[((b * 4 + f) * H + yy) * W + xx] = o[f];
  }
}

// adjoint: gx[q] = sum_f sum_t F_f[t] g_f[q - t]   (g is (B,4,H,W))
__device__ __forceinline__ float sobel_adj_at(const float* __restrict__ g, size_t plane, int H, int W, int y, int x) {
  const float F[4][9] = {{1, 2, 1, 0, 0, 0, -1, -2, -1}, {1, 0, -1, 2, 0, -2, 1, 0, -1},
                         {2, 1, 0, 1, 0, -1, 0, -1, -2}, {0, 1, 2, -1, 0, 1, -2, -1, 0}};
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
        if (F[f][t] != 0.f) s = fmaf(F[f][t], g[f * plane + (size_t)yy * W + xx], s);
    }
  }
  return s;
}

__global__ void sobel_bwd_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int H, int W) {
  const size_t n = (size_t)B * H * W, plane = (size_t)H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int xx = e % W, yy = (e / W) % H;
    const size_t b = e / plane;
    out[e] = sobel_adj_at(g + b * 4 * plane, plane, H, W, yy, xx);
  }
}

// block-wide sum of one float per thread (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  v = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return v;
}

__global__ __launch_bounds__(256) void huber_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float bscale, size_t n, float* __restrict__ partials) {
  __shared__ float sh[4];
  float s = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
    s += huber_val(a[e] - bscale * b[e]);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// out[j] = scale[j] * sum_k partials[k*stride + j]  (float64, fixed order); j < nout <= 4
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partials, int nblk, int stride,
                                                            int nout, float sc0, float sc1, float alpha,
                                                            float* __restrict__ out) {
  __shared__ double r[2][256];
  const int tid = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int k = tid; k < nblk; k += 256) {
    s0 += (double)partials[(size_t)k * stride];
    if (nout > 1) s1 += (double)partials[(size_t)k * stride + 1];
  }
  r[0][tid] = s0; r[1][tid] = s1;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) { r[0][tid] += r[0][tid + st]; r[1][tid] += r[1][tid + st]; }
    __syncthreads();
  }
  if (tid == 0) {
    const float l0 = (float)(r[0][0] * (double)sc0);
    out[0] = l0;
    if (nout > 1) {
      const float l1 = (float)(r[1][0] * (double)sc1);
      out[1] = l1;
      out[2] = alpha * l0 + (1.f - alpha) * l1;   // loss = alpha*ds + (1-alpha)*percep
    }
  }
}

// ga = gout[0] * clamp(a - bscale*b, -1, 1) / n
__global__ void huber_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float bscale,
                                 const float* __restrict__ gout, float inv_n, size_t n, float* __restrict__ ga) {
  const float go = gout[0] * inv_n;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    ga[e] = go * clamp1(a[e] - bscale * b[e]);
}

// ------------------------------------------------------------------------------------------------
// fused losses.  Pass A (per 32x32 tile): loss partials and the two residual-gradient maps
//   r1 (B,H/4,W/4) = alpha    /N1 * clamp(dn - lst)         dn = (D G1(sr*std+mean) - mean)/std
//   r2             = (1-alpha)/N2 * clamp(e2)               SR2: e2 = (sr - G2 sr) - gamma (ndvi - G2 ndvi), (B,H,W)
//                                                           SR1: e2 = sobel(sr) - gamma sobel(ndvi),        (B,H,W,4)
// Pass B: dsr = G1^T D^T r1 + (r2 - G2^T r2)   |   dsr = G1^T D^T r1 + sobel^T r2
// ------------------------------------------------------------------------------------------------
template <int KIND>   // 2: SR2 (gradFTM), 1: SR1 (predef filters)
__global__ __launch_bounds__(256) void sif_loss_fwd_kernel(const float* __restrict__ sr, const float* __restrict__ lst,
                                                           const float* __restrict__ ndvi, Taps k1, Taps k2, float mean,
                                                           float std, float gamma, float w1, float w2,
                                                           float* __restrict__ r1, float* __restrict__ r2,
                                                           float* __restrict__ partials, int H, int W) {
  __shared__ float L[TP * LS], M[TP * LS], N[TP * LS];
  __shared__ float sh[4];
  const int tid = threadIdx.x, x0 = blockIdx.x * T, y0 = blockIdx.y * T;
  const int tx = tid & 31, ty = tid >> 5;
  const size_t img = (size_t)blockIdx.z * H * W;
  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float h1 = 0.f, h2 = 0.f;

  // ---- consistency term: blur(mtf 0.1) of the de-normalised prediction, decimate, re-normalise ----
  load_tile_reflect(L, sr + img, H, W, y0, x0, std, mean, tid);
  __syncthreads();
  hpass(L, M, k1, tid);
  __syncthreads();
  {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = vpass_at(M, k1, ty + 8 * r, tx);
#pragma unroll
    for (int r = 0; r < 4; ++r) N[(ty + 8 * r) * LS + tx] = v[r];
  }
  __syncthreads();
  if (tid < 64) {
    const int i = tid >> 3, j = tid & 7;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s = fmaf(k4(a) * k4(b), N[(4 * i + a) * LS + 4 * j + b], s);
    if (y0 / 4 + i < H / 4 && x0 / 4 + j < W / 4) {          // partial tiles: only pixels of the image count
      const size_t o = (size_t)blockIdx.z * (H / 4) * (W / 4) + (size_t)(y0 / 4 + i) * (W / 4) + x0 / 4 + j;
      const float e = (s - mean) / std - lst[o];
      h1 = huber_val(e);
      r1[o] = w1 * clamp1(e);
    }
  }
  __syncthreads();

  if (KIND == 2) {
    // ---- high-frequency term: (sr - G2 sr) vs gamma (ndvi - G2 ndvi) ----
    load_tile_reflect(L, sr + img, H, W, y0, x0, 1.f, 0.f, tid);
    __syncthreads();
    hpass(L, M, k2, tid);
    __syncthreads();
    float hs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hs[r] = L[(ty + 8 * r + R) * LS + tx + R] - vpass_at(M, k2, ty + 8 * r, tx);
    __syncthreads();
    load_tile_reflect(L, ndvi + img, H, W, y0, x0, 1.f, 0.f, tid);
    __syncthreads();
    hpass(L, M, k2, tid);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = ty + 8 * r;
      const float hn = L[(y + R) * LS + tx + R] - vpass_at(M, k2, y, tx);
      const float e = hs[r] - gamma * hn;
      if (y0 + y < H && x0 + tx < W) {
        h2 += huber_val(e);
        r2[img + (size_t)(y0 + y) * W + x0 + tx] = w2 * clamp1(e);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = y0 + ty + 8 * r, x = x0 + tx;
      if (y >= H || x >= W) continue;
      float nb[9], a[4], c[4];
      neigh9_zero(sr + img, H, W, y, x, nb);
      sobel4(nb, a);
      neigh9_zero(ndvi + img, H, W, y, x, nb);
      sobel4(nb, c);
      float4 o;
      float e;
      e = a[0] - gamma * c[0]; h2 += huber_val(e); o.x = w2 * clamp1(e);
      e = a[1] - gamma * c[1]; h2 += huber_val(e); o.y = w2 * clamp1(e);
      e = a[2] - gamma * c[2]; h2 += huber_val(e); o.z = w2 * clamp1(e);
      e = a[3] - gamma * c[3]; h2 += huber_val(e); o.w = w2 * clamp1(e);
      st4(r2 + (img + (size_t)y * W + x) * 4, o);
    }
  }
  h1 = block_sum(h1, sh);
  h2 = block_sum(h2, sh);
  if (tid == 0) { partials[blk * 2] = h1; partials[blk * 2 + 1] = h2; }
}

// fin_*: the loss values (the job of loss_finalize_kernel) are reduced by workgroup (0,0,0) of THIS launch after its tile -- the
// gradient does not depend on them (w1, w2 are constants), and a separate 7 us launch between the two passes sat on the serial chain
template <int KIND>
__global__ __launch_bounds__(256) void sif_loss_bwd_kernel(const float* __restrict__ r1, const float* __restrict__ r2,
                                                           Taps k1, Taps k2, float* __restrict__ dsr, int H, int W,
                                                           const float* __restrict__ fin_partials, int fin_nblk, float fin_sc0,
                                                           float fin_sc1, float fin_alpha, float* __restrict__ fin_out) {
  __shared__ float L[TP * LS], M[TP * LS];
  __shared__ double fin_w[4][2];
  const int tid = threadIdx.x, x0 = blockIdx.x * T, y0 = blockIdx.y * T;
  const int tx = tid & 31, ty = tid >> 5;
  const size_t img = (size_t)blockIdx.z * H * W;
  float acc[4];
  // G1^T D^T r1   (d dn / d sr = G1^T D^T exactly: the std of the de-normalisation cancels the 1/std)
  load_tile_dect(L, r1 + (size_t)blockIdx.z * (H / 4) * (W / 4), H, W, y0, x0, tid);
  __syncthreads();
  hpass_adj(L, M, k1, x0, W, tid);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = vpass_adj_at(M, k1, ty + 8 * r, tx, y0, H);
  __syncthreads();
  if (KIND == 2) {
    load_tile_zero(L, r2 + img, H, W, y0, x0, tid);
    __syncthreads();
    hpass_adj(L, M, k2, x0, W, tid);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = ty + 8 * r;
      acc[r] += L[(y + R) * LS + tx + R] - vpass_adj_at(M, k2, y, tx, y0, H);
    }
  } else {
    const float F[4][9] = {{1, 2, 1, 0, 0, 0, -1, -2, -1}, {1, 0, -1, 2, 0, -2, 1, 0, -1},
                           {2, 1, 0, 1, 0, -1, 0, -1, -2}, {0, 1, 2, -1, 0, 1, -2, -1, 0}};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = y0 + ty + 8 * r, x = x0 + tx;
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
        if (y < H && x < W && yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const float4 g = ld4(r2 + (img + (size_t)yy * W + xx) * 4);
          s += F[0][t] * g.x + F[1][t] * g.y + F[2][t] * g.z + F[3][t] * g.w;
        }
      }
      acc[r] += s;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (y0 + ty + 8 * r < H && x0 + tx < W) dsr[img + (size_t)(y0 + ty + 8 * r) * W + x0 + tx] = acc[r];
  if (fin_out != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {   // (workgroup-uniform)
    double s0 = 0.0, s1 = 0.0;
    for (int k = tid; k < fin_nblk; k += 256) { s0 += (double)fin_partials[2 * (size_t)k]; s1 += (double)fin_partials[2 * (size_t)k + 1]; }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
    if ((tid & 63) == 0) { fin_w[tid >> 6][0] = s0; fin_w[tid >> 6][1] = s1; }
    __syncthreads();
    if (tid == 0) {
      const float l0 = (float)((fin_w[0][0] + fin_w[1][0] + fin_w[2][0] + fin_w[3][0]) * (double)fin_sc0);
      const float l1 = (float)((fin_w[0][1] + fin_w[1][1] + fin_w[2][1] + fin_w[3][1]) * (double)fin_sc1);
      fin_out[0] = l0; fin_out[1] = l1;
      fin_out[2] = fin_alpha * l0 + (1.f - fin_alpha) * l1;   // loss = alpha*ds + (1-alpha)*percep
    }
  }
}

inline Taps make_taps(const float* t9) { Taps k; for (int i = 0; i < 9; ++i) k.w[i] = t9[i]; return k; }
inline int ew_grid(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace


// The four unfused operators take any image of at least 2R+2 = 10 pixels a side (the two reflect zones of the adjoint must not overlap; the
// decimating pair and the fused training loss need multiples of 4): partial tiles are masked.
#define SIFSR_CHECK_ANY(H, W, M) if ((H) < 2 * R + 2 || (W) < 2 * R + 2 || (H) % (M) || (W) % (M) || B < 1 || B > 65535) return SIFSR_ERR_SHAPE
#define SIFSR_TGRID(H, W, B) dim3(((W) + T - 1) / T, ((H) + T - 1) / T, B)

int launch_blur_fwd(const float* x, const float* taps9, float* out, int B, int H, int W, hipStream_t s) {
  SIFSR_CHECK_ANY(H, W, 1);
  hipLaunchKernelGGL(blur_fwd_kernel, SIFSR_TGRID(H, W, B), dim3(256), 0, s, x, out, make_taps(taps9), H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_blur_bwd(const float* g, const float* taps9, float* out, int B, int H, int W, hipStream_t s) {
  SIFSR_CHECK_ANY(H, W, 1);
  hipLaunchKernelGGL(blur_bwd_kernel, SIFSR_TGRID(H, W, B), dim3(256), 0, s, g, out, make_taps(taps9), H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_blurdec_fwd(const float* x, const float* taps9, float* out, int B, int H, int W, hipStream_t s) {
  SIFSR_CHECK_ANY(H, W, 4);
  hipLaunchKernelGGL(blurdec_fwd_kernel, SIFSR_TGRID(H, W, B), dim3(256), 0, s, x, out, make_taps(taps9), H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_blurdec_bwd(const float* glr, const float* taps9, float* out, int B, int H, int W, hipStream_t s) {
  SIFSR_CHECK_ANY(H, W, 4);
  hipLaunchKernelGGL(blurdec_bwd_kernel, SIFSR_TGRID(H, W, B), dim3(256), 0, s, glr, out, make_taps(taps9), H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_sobel_fwd(const float* x, float* out, int B, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(sobel_fwd_kernel, dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, s, x, out, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_sobel_bwd(const float* g, float* out, int B, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(sobel_bwd_kernel, dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, s, g, out, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int huber_partial_blocks(size_t n) { size_t b = (n + 4095) / 4096; return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); }
int launch_huber_fwd(const float* a, const float* b, float bscale, size_t n, float* partials, float* out, hipStream_t s) {
  const int nblk = huber_partial_blocks(n);
  hipLaunchKernelGGL(huber_partial_kernel, dim3(nblk), dim3(256), 0, s, a, b, bscale, n, partials);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, partials, nblk, 1, 1, (float)(1.0 / (double)n), 0.f, 0.f, out);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
int launch_huber_bwd(const float* a, const float* b, float bscale, const float* gout, size_t n, float* ga, hipStream_t s) {
  hipLaunchKernelGGL(huber_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, s, a, b, bscale, gout, (float)(1.0 / (double)n), n, ga);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

size_t sif_loss_workspace_floats(int kind, int B, int H, int W) {
  const size_t hr = (size_t)B * H * W, lr = hr / 16, nblk = (size_t)B * ((H + T - 1) / T) * ((W + T - 1) / T);
  return lr + (kind == 1 ? 4 * hr : hr) + 2 * nblk;
}

int launch_sif_loss(int kind, const float* sr, const float* lst, const float* ndvi, int B, int H, int W, float mean,
                    float std, float alpha, float gamma, const float* taps_ds, const float* taps_ftm, float* ws,
                    float* losses3, float* dsr, hipStream_t s) {
  SIFSR_CHECK_ANY(H, W, 4);
  if (kind != 1 && kind != 2) return SIFSR_ERR_ARG;
  const size_t hr = (size_t)B * H * W, lr = hr / 16;
  const int nblk = B * ((H + T - 1) / T) * ((W + T - 1) / T);
  float* r1 = ws;
  float* r2 = r1 + lr;
  float* partials = r2 + (kind == 1 ? 4 * hr : hr);
  const double n1 = (double)lr, n2 = kind == 1 ? 4.0 * (double)hr : (double)hr;
  const float w1 = (float)((double)alpha / n1), w2 = (float)((1.0 - (double)alpha) / n2);
  const Taps k1 = make_taps(taps_ds), k2 = make_taps(taps_ftm);
  const dim3 grid = SIFSR_TGRID(H, W, B);
  if (kind == 2) {
    hipLaunchKernelGGL((sif_loss_fwd_kernel<2>), grid, dim3(256), 0, s, sr, lst, ndvi, k1, k2, mean, std, gamma, w1, w2, r1, r2, partials, H, W);
  } else {
    hipLaunchKernelGGL((sif_loss_fwd_kernel<1>), grid, dim3(256), 0, s, sr, lst, ndvi, k1, k2, mean, std, gamma, w1, w2, r1, r2, partials, H, W);
  }
  if (dsr != nullptr) {
    if (kind == 2) hipLaunchKernelGGL((sif_loss_bwd_kernel<2>), grid, dim3(256), 0, s, r1, r2, k1, k2, dsr, H, W, partials, nblk, (float)(1.0 / n1), (float)(1.0 / n2), alpha, losses3);
    else hipLaunchKernelGGL((sif_loss_bwd_kernel<1>), grid, dim3(256), 0, s, r1, r2, k1, k2, dsr, H, W, partials, nblk, (float)(1.0 / n1), (float)(1.0 / n2), alpha, losses3);
  } else {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, partials, nblk, 2, 2, (float)(1.0 / n1), (float)(1.0 / n2), alpha, losses3);
  }
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
