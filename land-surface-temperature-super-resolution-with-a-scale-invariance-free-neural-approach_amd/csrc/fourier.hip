// Fourier-domain evaluation: the only place the reference uses an FFT.
//   fourier_dict[m] = np.fft.fftshift(np.abs(sp.fft.fft2(LST)))          compare_methods.py:312-324
//   us.compute_2D_attenuation_spectra(fourier_dict[m])                   utils.py:598-636
// Hand-written 2-D FFT (radix-2, in LDS, float64 -- an evaluation-time operator on a handful of 256x256
// images; accuracy matters more than rate here: the attenuation spectra go down to -60 dB) + magnitude written
// at the fftshift-ed position, then the ring means of the magnitude around the DC bin.
#include "edge_conv.h"

namespace {

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// In-place radix-2 decimation-in-time FFT of length N = 1 << lg in LDS (input already bit-reversed);
// N/2 threads, tw[k] = exp(-2 pi i k / N), k < N/2.
__device__ __forceinline__ void fft_lds(cplx* x, const cplx* tw, int lg, int tid) {
  const int N = 1 << lg;
  for (int s = 1; s <= lg; ++s) {
    const int half = 1 << (s - 1);
    const int j = tid & (half - 1), base = (tid >> (s - 1)) << s;
    const cplx w = tw[j << (lg - s)];
    const cplx a = x[base + j], b = cmul(x[base + j + half], w);   // a butterfly only touches its own two slots
    x[base + j] = {a.re + b.re, a.im + b.im};
    x[base + j + half] = {a.re - b.re, a.im - b.im};
    __syncthreads();
  }
  (void)N;
}

__device__ __forceinline__ int bitrev(int v, int lg) { return (int)(__brev((unsigned)v) >> (32 - lg)); }

// pass 1: one workgroup per (row, image): real row -> complex spectrum row, written to `spec` (B,H,W)
__global__ void fft_rows_kernel(const float* __restrict__ img, cplx* __restrict__ spec, int H, int W, int lgW) {
  extern __shared__ double smem[];
  cplx* x = reinterpret_cast<cplx*>(smem);
  cplx* tw = x + W;
  const int tid = threadIdx.x;   // W/2 threads
  const size_t row = (size_t)blockIdx.y * H + blockIdx.x;
  const float* src = img + row * W;
  for (int i = tid; i < W; i += blockDim.x) x[bitrev(i, lgW)] = {(double)src[i], 0.0};
  for (int k = tid; k < W / 2; k += blockDim.x) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)W, &sn, &cs);
    tw[k] = {cs, sn};
  }
  __syncthreads();
  fft_lds(x, tw, lgW, tid);
  cplx* dst = spec + row * W;
  for (int i = tid; i < W; i += blockDim.x) dst[i] = x[i];
}

// pass 2: one workgroup per (column, image): column FFT, magnitude, store at the fftshift-ed position
__global__ void fft_cols_mag_kernel(const cplx* __restrict__ spec, float* __restrict__ mag, double* __restrict__ mag64,
                                    int H, int W, int lgH) {
  extern __shared__ double smem[];
  cplx* x = reinterpret_cast<cplx*>(smem);
  cplx* tw = x + H;
  const int tid = threadIdx.x;   // H/2 threads
  const int u = blockIdx.x;
  const size_t ib = (size_t)blockIdx.y * H * W;
  for (int i = tid; i < H; i += blockDim.x) x[bitrev(i, lgH)] = spec[ib + (size_t)i * W + u];
  for (int k = tid; k < H / 2; k += blockDim.x) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)H, &sn, &cs);
    tw[k] = {cs, sn};
  }
  __syncthreads();
  fft_lds(x, tw, lgH, tid);
  const int us = (u + W / 2) & (W - 1);
  for (int v = tid; v < H; v += blockDim.x) {
    const int vs = (v + H / 2) & (H - 1);
    const double m = sqrt(x[v].re * x[v].re + x[v].im * x[v].im);
    if (mag) mag[ib + (size_t)vs * W + us] = (float)m;
    if (mag64) mag64[ib + (size_t)vs * W + us] = m;
  }
}

// ring r (0 <= r < nr) = pixels with r^2 < d^2 <= (r+1)^2 around the centre (H/2, W/2): utils.py:621-631.
// One workgroup per (image, 16-row band); thread t owns rings t, t+256, ... and walks the two arcs of each ring
// inside the band (exact integer ring test), so every (band, ring) partial is one thread's fixed-order float64
// sum: deterministic, no atomics.
__global__ __launch_bounds__(256) void ring_partial_kernel(const double* __restrict__ mag64, int H, int W, int nr,
                                                           int rows_per_blk, double* __restrict__ part) {
  const int img = blockIdx.y, y0 = blockIdx.x * rows_per_blk;
  const int cy = H / 2, cx = W / 2;
  const double* m = mag64 + (size_t)img * H * W;
  for (int r = threadIdx.x; r < nr; r += 256) {
    double s = 0.0; double cnt = 0.0;
    const long long lo = (long long)r * r, hi = (long long)(r + 1) * (r + 1);
    for (int y = y0; y < y0 + rows_per_blk && y < H; ++y) {
      const long long dy2 = (long long)(y - cy) * (y - cy);
      if (dy2 > hi) continue;
      // columns with lo < dy2 + dx2 <= hi: |dx| in (sqrt(lo - dy2), sqrt(hi - dy2)]
      const long long rem_hi = hi - dy2, rem_lo = lo - dy2;
      int dxmax = (int)sqrt((double)rem_hi);
      while ((long long)(dxmax + 1) * (dxmax + 1) <= rem_hi) ++dxmax;
      while ((long long)dxmax * dxmax > rem_hi) --dxmax;
      int dxmin = 0;                                   // smallest |dx| with dx^2 > rem_lo
      if (rem_lo >= 0) {
        dxmin = (int)sqrt((double)rem_lo);
        while ((long long)dxmin * dxmin > rem_lo) --dxmin;
        while ((long long)(dxmin + 1) * (dxmin + 1) <= rem_lo) ++dxmin;
        dxmin += 1;
      }
      for (int dx = dxmin; dx <= dxmax; ++dx) {
        const int xr = cx + dx, xl = cx - dx;
        if (xr < W) { s += m[(size_t)y * W + xr]; cnt += 1.0; }
        if (dx != 0 && xl >= 0) { s += m[(size_t)y * W + xl]; cnt += 1.0; }
      }
    }
    double* o = part + (((size_t)img * gridDim.x + blockIdx.x) * nr + r) * 2;
    o[0] = s; o[1] = cnt;
  }
}

// spectrum[img][0] = 1 (intensity_f0 / intensity_f0, utils.py:619); [1 + r] = 10 (log10(mean_r) - log10(f0))
__global__ void ring_final_kernel(const double* __restrict__ part, const double* __restrict__ mag64, int nblk, int nr,
                                  int H, int W, float* __restrict__ spectrum) {
  const int img = blockIdx.x;
  const double f0 = mag64[(size_t)img * H * W + (size_t)(H / 2) * W + W / 2];
  float* out = spectrum + (size_t)img * (nr + 1);
  if (threadIdx.x == 0) out[0] = 1.f;
  for (int r = threadIdx.x; r < nr; r += blockDim.x) {
    double s = 0.0, c = 0.0;
    for (int k = 0; k < nblk; ++k) {
      const double* p = part + (((size_t)img * nblk + k) * nr + r) * 2;
      s += p[0]; c += p[1];
    }
    out[1 + r] = (float)(10.0 * (log10(s / c) - log10(f0)));
  }
}

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

size_t fourier_scratch_bytes(int B, int H, int W) {
  const size_t npx = (size_t)B * H * W;
  const int nr = (H / 2 < W / 2 ? H / 2 : W / 2) - 1;
  const int nblk = (H + 15) / 16;
  return npx * 16 + npx * 8 + (size_t)B * nblk * (nr > 0 ? nr : 1) * 16 + 256;
}

// mag (optional, float32 (B,H,W)) = fftshift(|fft2(img)|); spectrum (optional, (B, nr+1)), nr = min(H/2, W/2) - 1
int launch_fft2_attenuation(const float* img, int B, int H, int W, void* scratch, float* mag, float* spectrum, hipStream_t s) {
  if (B < 1 || !pow2(H) || !pow2(W) || H < 4 || W < 4 || H > 2048 || W > 2048) return SIFSR_ERR_SHAPE;
  const size_t npx = (size_t)B * H * W;
  cplx* spec = reinterpret_cast<cplx*>(scratch);
  double* mag64 = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + npx * 16);
  double* part = mag64 + npx;
  hipLaunchKernelGGL(fft_rows_kernel, dim3(H, B), dim3(W / 2), (size_t)(W + W / 2) * 16, s, img, spec, H, W, ilog2(W));
  hipLaunchKernelGGL(fft_cols_mag_kernel, dim3(W, B), dim3(H / 2), (size_t)(H + H / 2) * 16, s, spec, mag, mag64, H, W, ilog2(H));
  if (spectrum != nullptr) {
    const int nr = (H / 2 < W / 2 ? H / 2 : W / 2) - 1;
    if (nr < 1) return SIFSR_ERR_SHAPE;
    const int nblk = (H + 15) / 16;
    hipLaunchKernelGGL(ring_partial_kernel, dim3(nblk, B), dim3(256), 0, s, mag64, H, W, nr, 16, part);
    hipLaunchKernelGGL(ring_final_kernel, dim3(B), dim3(128), 0, s, part, mag64, nblk, nr, H, W, spectrum);
  }
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
