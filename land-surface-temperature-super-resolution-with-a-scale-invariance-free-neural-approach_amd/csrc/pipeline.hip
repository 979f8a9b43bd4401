// The step before the hot path and the per-batch metrics after it (f1), on the device.
//
//  * tiles_prepare: what ModisDatasetB.__getitem__ (dataset.py:134-142) / predict.py:84-100 do on the host per
//    tile -- z-score of the 64x64 LST tile, bicubic x4 upsample (us.upsampling = cv2.resize INTER_CUBIC,
//    utils.py:163-180: A = -0.75, half-pixel centres, edge-clamped), NDVI clip to [-1,1] + z-score,
//    torch.cat((lst_up, ndvi), 1) -- fused into one kernel that reads tiles straight out of a granule (or a
//    batch) and writes the model input (T,2,4w,4w).
//  * tiles_paste: predict.py:101-103, `* std + mean` and the write into the 4x granule.
//  * psnr / ssim: us.psnr_skimage / us.ssim_skimage (utils.py:548-578; scikit-image 0.22 defaults: 7x7 uniform
//    window, sample covariance, K1 = 0.01, K2 = 0.03, data_range = max - min of the TARGET BATCH, mean over
//    the window-valid interior), batch means as two device scalars -- no D2H of the images, no host stall.
#include "edge_conv.h"

namespace {

// ---------------------------------------------------------------------------------------------
// bicubic x4 (ATen upsample_bicubic2d / OpenCV INTER_CUBIC coefficients, A = -0.75)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  c[0] = cc2(t + 1.f, A); c[1] = cc1(t, A); c[2] = cc1(1.f - t, A); c[3] = cc2(2.f - t, A);
}

struct TileGeom {
  int tiles_x;                        // tile t -> (ty, tx) = (t / tiles_x, t % tiles_x)
  long long lst_step_y, lst_step_x;   // element offset of tile (ty, tx) = ty*step_y + tx*step_x
  int lst_row;                        // row stride inside a tile (elements)
  long long ndvi_step_y, ndvi_step_x;
  int ndvi_row;
};

// one workgroup = 16 output rows x (4*win) columns of one tile; thread = output column (win = 64 -> 256 threads)
__global__ __launch_bounds__(256) void tiles_prepare_kernel(const float* __restrict__ lst, const float* __restrict__ ndvi,
                                                            float* __restrict__ x, const TileGeom gm, int win,
                                                            float mean_lst, float istd_lst, float mean_ndvi,
                                                            float istd_ndvi, int clip_ndvi) {
  __shared__ float src[8][64 + 1];
  const int hr = 4 * win;
  const int t = blockIdx.x, Y0 = blockIdx.y * 16, X = threadIdx.x;
  const int ty = t / gm.tiles_x, tx = t - ty * gm.tiles_x;
  const float* lt = lst + ty * gm.lst_step_y + tx * gm.lst_step_x;
  const float* nt = ndvi + ty * gm.ndvi_step_y + tx * gm.ndvi_step_x;
  const int sy0 = Y0 / 4 - 2;   // first of the 8 source rows the 16 output rows touch
  for (int e = threadIdx.x; e < 8 * win; e += 256) {
    const int r = e / win, cidx = e - r * win;
    const int gy = clampi(sy0 + r, 0, win - 1);
    src[r][cidx] = (lt[(size_t)gy * gm.lst_row + cidx] - mean_lst) * istd_lst;
  }
  __syncthreads();
  float* o0 = x + ((size_t)t * 2 + 0) * hr * hr;
  float* o1 = x + ((size_t)t * 2 + 1) * hr * hr;
  if (X < hr) {
    // horizontal pass: source index of output column X (scale 1/4, half-pixel centres)
    const float sx = 0.25f * ((float)X + 0.5f) - 0.5f;
    const float fx = floorf(sx);
    const int ix = (int)fx;
    float cx[4];
    cubic_coeffs(sx - fx, cx);
    int xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xs[j] = clampi(ix - 1 + j, 0, win - 1);
    float hrow[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      hrow[r] = src[r][xs[0]] * cx[0] + src[r][xs[1]] * cx[1] + src[r][xs[2]] * cx[2] + src[r][xs[3]] * cx[3];
#pragma unroll
    for (int dy = 0; dy < 16; ++dy) {
      const int Y = Y0 + dy;
      const float sy = 0.25f * ((float)Y + 0.5f) - 0.5f;
      const float fy = floorf(sy);
      const int iy = (int)fy;
      float cy[4];
      cubic_coeffs(sy - fy, cy);
      float v = 0.f;
      // staged row r <-> source row clamp(sy0 + r): the clamp is already applied; Y0 is a multiple of 16, so
      // iy - sy0 depends on dy alone (compile time): iy = Y0/4 + ((dy + 2) >> 2) - 1
      (void)iy;
      const int rb = ((dy + 2) >> 2) + 0;   // = iy - 1 - sy0
#pragma unroll
      for (int i = 0; i < 4; ++i) v = (i == 0) ? hrow[rb + i] * cy[0] : v + hrow[rb + i] * cy[i];
      o0[(size_t)Y * hr + X] = v;
      float nv = nt[(size_t)Y * gm.ndvi_row + X];
      if (clip_ndvi) nv = fminf(fmaxf(nv, -1.f), 1.f);
      o1[(size_t)Y * hr + X] = (nv - mean_ndvi) * istd_ndvi;
    }
  }
}

__global__ __launch_bounds__(256) void tiles_paste_kernel(const float* __restrict__ sr, float* __restrict__ out, int T,
                                                          int tiles_x, int hr, long long out_row, float mean, float std) {
  const size_t n = (size_t)T * hr * hr;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int X = (int)(e % hr);
    const size_t r = e / hr;
    const int Y = (int)(r % hr), t = (int)(r / hr);
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    out[((size_t)ty * hr + Y) * out_row + (size_t)tx * hr + X] = sr[e] * std + mean;
  }
}

// us.downsampling (utils.py:183-213), the 'norm-L4' decimation: (mean of x^4 over each 4x4 block)^(1/4)
__global__ __launch_bounds__(256) void l4pool4_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W) {
  const int Ho = H / 4, Wo = W / 4;
  const size_t n = (size_t)B * Ho * Wo;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int ox = (int)(e % Wo);
    const size_t r = e / Wo;
    const int oy = (int)(r % Ho), b = (int)(r / Ho);
    const float* src = x + ((size_t)b * H + 4 * oy) * W + 4 * ox;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = ld4(src + (size_t)i * W);
      s += v.x * v.x * v.x * v.x; s += v.y * v.y * v.y * v.y; s += v.z * v.z * v.z * v.z; s += v.w * v.w * v.w * v.w;
    }
    out[e] = powf(s / 16.f, 0.25f);
  }
}

// ---------------------------------------------------------------------------------------------
// PSNR / SSIM
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ t, size_t n, float* __restrict__ part) {
  __shared__ float smin[256], smax[256];
  float lo = INFINITY, hi = -INFINITY;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const float v = t[e];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + st]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}

__global__ void minmax_final_kernel(float* __restrict__ part, int n) {
  if (threadIdx.x == 0) {
    float lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < n; ++i) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
    part[512] = lo; part[513] = hi;
  }
}

constexpr int MT = 32;          // SSIM output tile
constexpr int MH = MT + 6;      // + 3-pixel halo of the 7x7 window

// per (image, 32x32 tile): sum of the SSIM map over the window-valid pixels of the tile, and sum of squared error
__global__ __launch_bounds__(256) void psnr_ssim_tile_kernel(const float* __restrict__ pred, const float* __restrict__ targ,
                                                             const float* __restrict__ mm, int H, int W,
                                                             double* __restrict__ part) {
  __shared__ float a[MH][MH + 1], b[MH][MH + 1];   // a = target (im1 in skimage's call), b = prediction
  __shared__ double hs[5][MH][MT];                  // horizontal 7-sums of a, b, a*a, b*b, a*b
  __shared__ double red[256][2];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * MT, y0 = blockIdx.y * MT, img = blockIdx.z;
  const float R = mm[513] - mm[512];
  const float C1 = (0.01f * R) * (0.01f * R), C2 = (0.03f * R) * (0.03f * R);
  const float* tp = targ + (size_t)img * H * W;
  const float* pp = pred + (size_t)img * H * W;
  for (int e = tid; e < MH * MH; e += 256) {
    const int r = e / MH, c = e - r * MH;
    const int gy = y0 - 3 + r, gx = x0 - 3 + c;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    a[r][c] = in ? tp[(size_t)gy * W + gx] : 0.f;
    b[r][c] = in ? pp[(size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < MH * MT; e += 256) {
    const int r = e / MT, c = e - r * MT;
    double s[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float u = a[r][c + k], v = b[r][c + k];
      s[0] += (double)u; s[1] += (double)v; s[2] += (double)(u * u); s[3] += (double)(v * v); s[4] += (double)(u * v);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) hs[q][r][c] = s[q];
  }
  __syncthreads();
  double ssum = 0.0, esum = 0.0;
  for (int e = tid; e < MT * MT; e += 256) {
    const int r = e / MT, c = e - r * MT;
    const int gy = y0 + r, gx = x0 + c;
    if (gy < H && gx < W) {
      const float d = a[r + 3][c + 3] - b[r + 3][c + 3];   // skimage: float32 difference, squared, float64 mean
      esum += (double)(d * d);
    }
    if (gy >= 3 && gy < H - 3 && gx >= 3 && gx < W - 3) {
      double s[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) s[q] += hs[q][r + k][c];
      // scipy.ndimage.uniform_filter: float64 accumulation, float32 result; the rest in float32 like skimage
      const float ux = (float)(s[0] / 49.0), uy = (float)(s[1] / 49.0);
      const float uxx = (float)(s[2] / 49.0), uyy = (float)(s[3] / 49.0), uxy = (float)(s[4] / 49.0);
      const float cov_norm = 49.f / 48.f;
      const float vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
      const float A1 = 2.f * ux * uy + C1, A2 = 2.f * vxy + C2;
      const float B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
      ssum += (double)((A1 * A2) / (B1 * B2));
    }
  }
  red[tid][0] = ssum; red[tid][1] = esum;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) { red[tid][0] += red[tid + st][0]; red[tid][1] += red[tid + st][1]; }
    __syncthreads();
  }
  if (tid == 0) {
    const size_t blk = ((size_t)img * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    part[2 * blk] = red[0][0]; part[2 * blk + 1] = red[0][1];
  }
}

// out[0] = mean_i 10*log10(R^2 / mse_i), out[1] = mean_i mean(SSIM map_i over the valid interior)
__global__ __launch_bounds__(256) void psnr_ssim_final_kernel(const double* __restrict__ part, int tiles_per_img, int B,
                                                              int H, int W, const float* __restrict__ mm,
                                                              float* __restrict__ out) {
  __shared__ double r1[256], r2[256];
  const double R = (double)(mm[513] - mm[512]);
  double ps = 0.0, ss = 0.0;
  for (int img = threadIdx.x; img < B; img += 256) {
    double s = 0.0, e = 0.0;
    for (int k = 0; k < tiles_per_img; ++k) { s += part[2 * ((size_t)img * tiles_per_img + k)]; e += part[2 * ((size_t)img * tiles_per_img + k) + 1]; }
    const double mse = e / ((double)H * W);
    ps += 10.0 * log10(R * R / mse);
    ss += s / ((double)(H - 6) * (W - 6));
  }
  r1[threadIdx.x] = ps; r2[threadIdx.x] = ss;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = (float)(r1[0] / B); out[1] = (float)(r2[0] / B); }
}

}  // namespace

int launch_tiles_prepare(const float* lst, const float* ndvi, float* x, int T, int tiles_x, int win, long long lst_step_y,
                         long long lst_step_x, int lst_row, long long ndvi_step_y, long long ndvi_step_x, int ndvi_row,
                         float mean_lst, float std_lst, float mean_ndvi, float std_ndvi, int clip_ndvi, hipStream_t s) {
  if (T < 1 || tiles_x < 1 || win < 4 || win > 64 || win % 4 || std_lst == 0.f || std_ndvi == 0.f) return SIFSR_ERR_SHAPE;
  TileGeom gm{tiles_x, lst_step_y, lst_step_x, lst_row, ndvi_step_y, ndvi_step_x, ndvi_row};
  hipLaunchKernelGGL(tiles_prepare_kernel, dim3(T, (4 * win) / 16), dim3(256), 0, s, lst, ndvi, x, gm, win, mean_lst,
                     1.f / std_lst, mean_ndvi, 1.f / std_ndvi, clip_ndvi);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_tiles_paste(const float* sr, float* out, int T, int tiles_x, int hr, long long out_row, float mean, float std,
                       hipStream_t s) {
  if (T < 1 || tiles_x < 1 || hr < 1) return SIFSR_ERR_SHAPE;
  const size_t n = (size_t)T * hr * hr;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(tiles_paste_kernel, dim3((int)blocks), dim3(256), 0, s, sr, out, T, tiles_x, hr, out_row, mean, std);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

size_t psnr_ssim_scratch_bytes(int B, int H, int W) {
  const size_t tiles = (size_t)((W + MT - 1) / MT) * ((H + MT - 1) / MT);
  return 528 * sizeof(float) + (size_t)B * tiles * 2 * sizeof(double) + 64;
}

int launch_psnr_ssim(const float* pred, const float* targ, int B, int H, int W, void* scratch, float* out2, hipStream_t s) {
  if (B < 1 || H < 7 || W < 7) return SIFSR_ERR_SHAPE;
  float* mm = reinterpret_cast<float*>(scratch);
  double* part = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + 528 * sizeof(float));
  const size_t n = (size_t)B * H * W;
  int mmb = (int)((n + 256 * 64 - 1) / (256 * 64));
  if (mmb > 256) mmb = 256;
  if (mmb < 1) mmb = 1;
  const dim3 grid((W + MT - 1) / MT, (H + MT - 1) / MT, B);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(mmb), dim3(256), 0, s, targ, n, mm);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, s, mm, mmb);
  hipLaunchKernelGGL(psnr_ssim_tile_kernel, grid, dim3(256), 0, s, pred, targ, mm, H, W, part);
  hipLaunchKernelGGL(psnr_ssim_final_kernel, dim3(1), dim3(256), 0, s, part, (int)(grid.x * grid.y), B, H, W, mm, out2);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_l4pool4(const float* x, float* out, int B, int H, int W, hipStream_t s) {
  if (B < 1 || H < 4 || W < 4 || H % 4 || W % 4) return SIFSR_ERR_SHAPE;
  const size_t n = (size_t)B * (H / 4) * (W / 4);
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(l4pool4_kernel, dim3((int)blocks), dim3(256), 0, s, x, out, B, H, W);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
