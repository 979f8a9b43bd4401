// Training / eval BatchNorm2d pieces (nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1, biased batch
// variance for normalisation, unbiased for running_var -- model.py:136,139,508).
//
// Forward statistics are produced as per-workgroup (sum, sumsq) partials by the conv kernels'
// epilogues; bn_finalize reduces them in float64 (order-stable, no float atomics), folds
// gamma/beta/mean/invstd into scale/shift for the consumer's load path and updates running stats.
// Backward: dz = g * [z > 0];  dgamma = sum dz*xhat;  dbeta = sum dz;
//           dy = gamma*invstd * (dz - dbeta/N - xhat*dgamma/N) = scale*dz + c1*y + c0.
#include "edge_conv.h"

namespace {

// (s1, s2) summed over the workgroup in float64, fixed order: butterfly inside each wave, then the waves in order -- ONE barrier
// (the 8-level LDS tree that stood here cost 8 barriers in ~30 launches per step that do nothing else; they sit on the serial chain)
template <int NT>
static __device__ __forceinline__ void block_sum2(double& s1, double& s2, double (*wsum)[2]) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { wsum[wave][0] = s1; wsum[wave][1] = s2; }
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { s1 += wsum[w][0]; s2 += wsum[w][1]; }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int nblk, int C,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* run_mean,
                                                          float* run_var, float momentum, float eps, float* mean,
                                                          float* invstd, float* scale, float* shift) {
  __shared__ double wsum[4][2];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k = tid; k < nblk; k += 256) {
    const float2 v = *reinterpret_cast<const float2*>(partials + ((size_t)k * C + c) * 2);
    s1 += (double)v.x; s2 += (double)v.y;
  }
  block_sum2<256>(s1, s2, wsum);
  if (tid == 0) {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * istd;
    mean[c] = (float)m;
    invstd[c] = istd;
    scale[c] = sc;
    shift[c] = fmaf(-(float)m, sc, beta[c]);
    if (run_mean != nullptr) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
  }
}

struct BnEvalTable { int gamma_off[SIFSR_NUM_BN_LAYERS], beta_off[SIFSR_NUM_BN_LAYERS], run_off[SIFSR_NUM_BN_LAYERS],
                         ch_off[SIFSR_NUM_BN_LAYERS], cout[SIFSR_NUM_BN_LAYERS]; };

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ params, const float* __restrict__ running, float eps,
                                      float* scale, float* shift, const BnEvalTable tb) {
  const int l = blockIdx.x, c = threadIdx.x;
  if (c >= tb.cout[l]) return;
  const float rm = running[tb.run_off[l] + c], rv = running[tb.run_off[l] + tb.cout[l] + c];
  const float sc = params[tb.gamma_off[l] + c] / sqrtf(rv + eps);
  scale[tb.ch_off[l] + c] = sc;
  shift[tb.ch_off[l] + c] = fmaf(-rm, sc, params[tb.beta_off[l] + c]);
}

// Optional second gradient source: the AvgPool2d(2,2) adjoint of a half-resolution gradient gp, added on the fly,
//   g_eff[y][x] = g[y][x] + 0.25 * gp[y/2][x/2]
// (the skip layers inbloc.bloc.3 / db1,2.lastconv feed both the decoder skip and the next pooling stage; this
// replaces a separate read-modify-write pass over g).  W2 = W/2 etc. describe the full-resolution image.
struct PoolAdj { const float* gp; int H, W; };
template <int C, bool HS = false>
__device__ __forceinline__ float4 pool_adj4(const PoolAdj pa, size_t p, int c4) {
  // 32-bit index arithmetic (p < 2^28: the launchers keep every tensor below 2^32 bytes): the 64-bit divisions that stood here
  // cost ~300 instructions per element and made the kernel compute-bound once bf16 storage halved its bytes
  const unsigned W = (unsigned)pa.W, H = (unsigned)pa.H, p32 = (unsigned)p;
  const unsigned r = p32 / W, x = p32 - r * W;
  const unsigned b = r / H, yy = r - b * H;
  const float4 v = ldA4<HS>(pa.gp, (((size_t)b * (H / 2) + yy / 2) * (W / 2) + x / 2) * C + 4 * c4);
  return make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w);
}

// per-workgroup partial (sum dz, sum dz*xhat) per channel
template <int C, bool HS>   // HS: g, y, gp stored as bf16 (the bf16 compute mode, common.h)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* g, const float* __restrict__ y,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, size_t npix,
                                                            float* __restrict__ partials, const PoolAdj pa,
                                                            float* __restrict__ g_out) {
  SIFSR_CHAIN_PRIO();
  constexpr int Q = C / 4;          // channel quads
  constexpr int PP = 256 / Q;       // pixels per pass per workgroup
  __shared__ double red[256][8];   // float64 accumulation: dy = scale*(dz - mean(dz) - ...) cancels heavily
  const int tid = threadIdx.x, c4 = tid % Q, pl = tid / Q;
  const float4 sc = ld4(scale + 4 * c4), sh = ld4(shift + 4 * c4), mu = ld4(mean + 4 * c4), is = ld4(invstd + 4 * c4);
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  for (size_t p = (size_t)blockIdx.x * PP + pl; p < npix; p += (size_t)gridDim.x * PP) {
    const float4 yv = ldA4<HS>(y, p * C + 4 * c4);
    float4 gv = ldA4<HS>(g, p * C + 4 * c4);
    if (pa.gp != nullptr) {
      const float4 q = pool_adj4<C, HS>(pa, p, c4); gv.x += q.x; gv.y += q.y; gv.z += q.z; gv.w += q.w;
      // the completed gradient goes back in place (g_out == g; each element is read and written by this thread only):
      // the input- and weight-gradient convolutions of this layer form dL/dy from (g, y) while staging (bn_bwd4)
      if (g_out != nullptr) { gv = as_stored4<HS>(gv); stA4<HS>(g_out, p * C + 4 * c4, gv); }   // sums = those of the stored values
    }
    const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dz = fmaf(yy[j], scv[j], shv[j]) > 0.f ? gg[j] : 0.f;
      a1[j] += (double)dz;
      a2[j] += (double)dz * (double)((yy[j] - muv[j]) * isv[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid][j] = a1[j]; red[tid][4 + j] = a2[j]; }
  __syncthreads();
  // tree over the PP pixel lanes that share a channel quad (threads c4, c4+Q, c4+2Q, ...)
  for (int st = PP / 2; st > 0; st >>= 1) {
    if (pl < st) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[tid][j] += red[tid + st * Q][j];
    }
    __syncthreads();
  }
  if (pl == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      partials[((size_t)blockIdx.x * C + 4 * c4 + j) * 2 + 0] = (float)red[tid][j];
      partials[((size_t)blockIdx.x * C + 4 * c4 + j) * 2 + 1] = (float)red[tid][4 + j];
    }
  }
}

// fp32 coefficients of bn_bwd4 (common.h), [sc | sh | k1 | k0] with C floats each: dy = sc*dz + k1*z + k0 on the
// forward's own pre-activation z = fma(y, sc, sh); k1 = -invstd*dgamma/N, k0 = -sc*dbeta/N - k1*beta (float64, rounded once).
static __device__ __forceinline__ void write_coef_f(float* coef_f, int C, int c, const float* scale, const float* shift,
                                                    const float* invstd, const float* beta, double db, double dg, double count) {
  if (coef_f == nullptr) return;
  const double k1 = -(double)invstd[c] * dg / count;
  coef_f[c] = scale[c];
  coef_f[C + c] = shift[c];
  coef_f[2 * C + c] = (float)k1;
  coef_f[3 * C + c] = (float)(-(double)scale[c] * db / count - k1 * (double)beta[c]);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, int C,
                                                              double count, const float* __restrict__ scale,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* dgamma,
                                                              float* dbeta, double* coef, const float* __restrict__ shift,
                                                              const float* __restrict__ beta, float* coef_f,
                                                              const float* __restrict__ xs_partials, int xs_n, float* __restrict__ xs_out) {
  SIFSR_CHAIN_PRIO();
  __shared__ double wsum[4][2];
  const int c = blockIdx.x, tid = threadIdx.x;
  if (c >= C) {
    // rider (the tail of the backward): xs_out[e] = sum over the nblk rows of xs_partials[row][xs_n] -- the outlay weight / bias
    // gradient partials of the same pass; four outputs per extra workgroup, float64, fixed order (as sum_partials_kernel)
    __shared__ double part[64][4];
    const int el = tid & 3, grp = tid >> 2, e = (c - C) * 4 + el;
    double s = 0.0;
    if (e < xs_n)
      for (int k = grp; k < nblk; k += 64) s += (double)xs_partials[(size_t)k * xs_n + e];
    part[grp][el] = s;
    __syncthreads();
    if (grp == 0 && e < xs_n) {
      double t = 0.0;
      for (int g = 0; g < 64; ++g) t += part[g][el];
      xs_out[e] = (float)t;
    }
    return;
  }
  double s1 = 0.0, s2 = 0.0;
  for (int k = tid; k < nblk; k += 256) {
    const float2 v = *reinterpret_cast<const float2*>(partials + ((size_t)k * C + c) * 2);
    s1 += (double)v.x; s2 += (double)v.y;
  }
  block_sum2<256>(s1, s2, wsum);
  if (tid == 0) {
    const double db = s1, dg = s2;
    dbeta[c] = (float)db;
    dgamma[c] = (float)dg;
    // dy = scale*dz + k1*y + k0, kept in float64: the three terms cancel to << |scale*dz| when the
    // incoming gradient is smooth (ATen's CPU kernel also evaluates this in acc_type<float> = double)
    const double k1 = -(double)scale[c] * (double)invstd[c] * dg / count;
    coef[c] = (double)scale[c];
    coef[C + c] = k1;
    coef[2 * C + c] = -(double)scale[c] * db / count - k1 * (double)mean[c];
    write_coef_f(coef_f, C, c, scale, shift, invstd, beta, db, dg, count);
  }
}

// Same, for sums produced by the dgrad epilogue + border kernel of the layer above (conv_mfma.hip): two partial arrays,
// and the second sum is sum dz*y (not dz*xhat): sum dz*xhat = invstd * (sum dz*y - mean * sum dz), in float64.
template <int NT>   // threads per channel: 1024 when there are thousands of partial rows (the upsample adjoint writes one per 8x8 pixels)
__global__ __launch_bounds__(NT) void bn_bwd_finalize2_kernel(const float* __restrict__ pa, int na,
                                                               const float* __restrict__ pb, int nb, int C, double count,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float* dgamma,
                                                               float* dbeta, double* coef, const float* __restrict__ shift,
                                                               const float* __restrict__ beta, float* coef_f) {
  SIFSR_CHAIN_PRIO();
  __shared__ double wsum[NT / 64][2];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k = tid; k < na; k += NT) { const float2 v = *reinterpret_cast<const float2*>(pa + ((size_t)k * C + c) * 2); s1 += (double)v.x; s2 += (double)v.y; }
  for (int k = tid; k < nb; k += NT) { const float2 v = *reinterpret_cast<const float2*>(pb + ((size_t)k * C + c) * 2); s1 += (double)v.x; s2 += (double)v.y; }
  block_sum2<NT>(s1, s2, wsum);
  if (tid == 0) {
    const double db = s1;
    const double dg = (double)invstd[c] * (s2 - (double)mean[c] * db);
    dbeta[c] = (float)db;
    dgamma[c] = (float)dg;
    const double k1 = -(double)scale[c] * (double)invstd[c] * dg / count;
    coef[c] = (double)scale[c];
    coef[C + c] = k1;
    coef[2 * C + c] = -(double)scale[c] * db / count - k1 * (double)mean[c];
    write_coef_f(coef_f, C, c, scale, shift, invstd, beta, db, dg, count);
  }
}

template <int C>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const double* __restrict__ coef,
                                                           size_t nquads, float* __restrict__ dy, const PoolAdj pa) {
  SIFSR_CHAIN_PRIO();
  constexpr int Q = C / 4;
  const int c4 = threadIdx.x % Q;   // 256 % Q == 0 and grid stride is a multiple of 256
  const float4 sc = ld4(scale + 4 * c4), sh = ld4(shift + 4 * c4);
  double sd[4], k1[4], k0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sd[j] = coef[4 * c4 + j]; k1[j] = coef[C + 4 * c4 + j]; k0[j] = coef[2 * C + 4 * c4 + j]; }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nquads; e += (size_t)gridDim.x * 256) {
    const float4 yv = ld4(y + e * 4);
    float4 gv = ld4(g + e * 4);
    if (pa.gp != nullptr) { const float4 q = pool_adj4<C>(pa, e / Q, c4); gv.x += q.x; gv.y += q.y; gv.z += q.z; gv.w += q.w; }
    float4 o;
    o.x = (float)fma(sd[0], (double)(fmaf(yv.x, sc.x, sh.x) > 0.f ? gv.x : 0.f), fma(k1[0], (double)yv.x, k0[0]));
    o.y = (float)fma(sd[1], (double)(fmaf(yv.y, sc.y, sh.y) > 0.f ? gv.y : 0.f), fma(k1[1], (double)yv.y, k0[1]));
    o.z = (float)fma(sd[2], (double)(fmaf(yv.z, sc.z, sh.z) > 0.f ? gv.z : 0.f), fma(k1[2], (double)yv.z, k0[2]));
    o.w = (float)fma(sd[3], (double)(fmaf(yv.w, sc.w, sh.w) > 0.f ? gv.w : 0.f), fma(k1[3], (double)yv.w, k0[3]));
    st4(dy + e * 4, o);
  }
}

__global__ void nbt_increment_kernel(long long* nbt, int n) {
  const int i = threadIdx.x;
  if (i < n) nbt[i] += 1;
}

}  // namespace

int launch_bn_finalize(const float* partials, int nblk, int C, double count, const float* gamma, const float* beta,
                       float* run_mean, float* run_var, float momentum, float eps, float* mean, float* invstd,
                       float* scale, float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, partials, nblk, C, count, gamma, beta, run_mean,
                     run_var, momentum, eps, mean, invstd, scale, shift);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_bn_eval_coeffs(const float* params, const float* running, float eps, float* scale, float* shift, hipStream_t s) {
  const NetTable& nt = sifsr_net();
  BnEvalTable tb;
  for (int l = 0; l < SIFSR_NUM_BN_LAYERS; ++l) {
    tb.gamma_off[l] = nt.L[l].gamma_off; tb.beta_off[l] = nt.L[l].beta_off; tb.run_off[l] = nt.L[l].run_off;
    tb.ch_off[l] = nt.L[l].ch_off; tb.cout[l] = nt.L[l].cout;
  }
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(SIFSR_NUM_BN_LAYERS), dim3(64), 0, s, params, running, eps, scale, shift, tb);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_bn_bwd_reduce(const float* g, const float* y, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int C, size_t npix, float* partials, int nblk, hipStream_t s, const float* gp,
                         int H, int W, float* g_out) {
  if (gp != nullptr && (H < 2 || W < 2 || H % 2 || W % 2 || npix % ((size_t)H * W))) return SIFSR_ERR_SHAPE;
  if (g_out != nullptr && (g_out != g || gp == nullptr)) return SIFSR_ERR_ARG;   // in place, and only with the pooling adjoint
  const PoolAdj pa{gp, H, W};
  const bool hs = sifsr_half_storage();
  switch (C) {
    case 16:
      if (hs) hipLaunchKernelGGL((bn_bwd_reduce_kernel<16, true>), dim3(nblk), dim3(256), 0, s, g, y, scale, shift, mean, invstd, npix, partials, pa, g_out);
      else hipLaunchKernelGGL((bn_bwd_reduce_kernel<16, false>), dim3(nblk), dim3(256), 0, s, g, y, scale, shift, mean, invstd, npix, partials, pa, g_out);
      break;
    case 32:
      if (hs) hipLaunchKernelGGL((bn_bwd_reduce_kernel<32, true>), dim3(nblk), dim3(256), 0, s, g, y, scale, shift, mean, invstd, npix, partials, pa, g_out);
      else hipLaunchKernelGGL((bn_bwd_reduce_kernel<32, false>), dim3(nblk), dim3(256), 0, s, g, y, scale, shift, mean, invstd, npix, partials, pa, g_out);
      break;
    case 64:
      if (hs) hipLaunchKernelGGL((bn_bwd_reduce_kernel<64, true>), dim3(nblk), dim3(256), 0, s, g, y, scale, shift, mean, invstd, npix, partials, pa, g_out);
      else hipLaunchKernelGGL((bn_bwd_reduce_kernel<64, false>), dim3(nblk), dim3(256), 0, s, g, y, scale, shift, mean, invstd, npix, partials, pa, g_out);
      break;
    default: return SIFSR_ERR_SHAPE;
  }
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_bn_bwd_finalize(const float* partials, int nblk, int C, double count, const float* scale, const float* mean,
                           const float* invstd, float* dgamma, float* dbeta, double* coef, hipStream_t s, const float* shift,
                           const float* beta, float* coef_f, const float* xs_partials, int xs_n, float* xs_out) {
  if (coef_f != nullptr && (!shift || !beta)) return SIFSR_ERR_ARG;
  if (xs_partials != nullptr && (xs_n < 1 || !xs_out)) return SIFSR_ERR_ARG;
  const int extra = xs_partials != nullptr ? (xs_n + 3) / 4 : 0;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C + extra), dim3(256), 0, s, partials, nblk, C, count, scale, mean, invstd,
                     dgamma, dbeta, coef, shift, beta, coef_f, xs_partials, xs_n, xs_out);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_bn_bwd_finalize2(const float* pa, int na, const float* pb, int nb, int C, double count, const float* scale,
                            const float* mean, const float* invstd, float* dgamma, float* dbeta, double* coef, hipStream_t s,
                            const float* shift, const float* beta, float* coef_f) {
  if (coef_f != nullptr && (!shift || !beta)) return SIFSR_ERR_ARG;
  // NOTE: the result depends on the thread count (summation order), so the choice is a function of the row count alone
  if (na + nb > 4096)
    hipLaunchKernelGGL((bn_bwd_finalize2_kernel<1024>), dim3(C), dim3(1024), 0, s, pa, na, pb, nb, C, count, scale, mean, invstd, dgamma,
                       dbeta, coef, shift, beta, coef_f);
  else
    hipLaunchKernelGGL((bn_bwd_finalize2_kernel<256>), dim3(C), dim3(256), 0, s, pa, na, pb, nb, C, count, scale, mean, invstd, dgamma,
                       dbeta, coef, shift, beta, coef_f);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_bn_bwd_apply(const float* g, const float* y, const float* scale, const float* shift, const double* coef,
                        int C, size_t npix, float* dy, hipStream_t s, const float* gp, int H, int W) {
  if (gp != nullptr && (H < 2 || W < 2 || H % 2 || W % 2 || npix % ((size_t)H * W))) return SIFSR_ERR_SHAPE;
  const PoolAdj pa{gp, H, W};
  const size_t nquads = npix * (size_t)C / 4;
  size_t blocks = (nquads + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  switch (C) {
    case 16: hipLaunchKernelGGL((bn_bwd_apply_kernel<16>), dim3((int)blocks), dim3(256), 0, s, g, y, scale, shift, coef, nquads, dy, pa); break;
    case 32: hipLaunchKernelGGL((bn_bwd_apply_kernel<32>), dim3((int)blocks), dim3(256), 0, s, g, y, scale, shift, coef, nquads, dy, pa); break;
    case 64: hipLaunchKernelGGL((bn_bwd_apply_kernel<64>), dim3((int)blocks), dim3(256), 0, s, g, y, scale, shift, coef, nquads, dy, pa); break;
    default: return SIFSR_ERR_SHAPE;
  }
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_nbt_increment(long long* nbt, int n, hipStream_t s) {
  hipLaunchKernelGGL(nbt_increment_kernel, dim3(1), dim3(64), 0, s, nbt, n);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
