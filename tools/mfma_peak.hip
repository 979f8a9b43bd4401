// Sustained fp32 MFMA rate of the device: 256 CUs x 8 waves, each issuing independent v_mfma_f32_16x16x4_f32
// back to back from registers only (no LDS, no memory).  Build+run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
// RANDOM = 0: every MFMA multiplies the same two small per-lane constants (low switching activity);
// RANDOM = 1: operands are 8 + 8 registers of uniformly random floats in [-1, 1) (what a real layer feeds).
template <int RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x4 a[8];
  for (int i = 0; i < 8; ++i) a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float x[8], y[8];
  unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u; x[i] = RANDOM ? (float)(int)s * (1.0f / 2147483648.0f) : threadIdx.x * 1e-3f;
    s = s * 1664525u + 1013904223u; y[i] = RANDOM ? (float)(int)s * (1.0f / 2147483648.0f) : threadIdx.x * 2e-3f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[(i + r) & 7], y[(i + 3 * r) & 7], a[i], 0, 0, 0);
    // no rescaling of the accumulators: with |x|, |y| < 1 they random-walk to ~1e3 over a launch, far from overflow,
    // and any VALU instruction in this loop would serialise with the MFMAs (tools/mfma_valu_coexec.hip) and read as
    // a lower "peak" (an earlier version scaled them by 0.5 per iteration and measured 141 TFLOP/s for that reason)
  }
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) sum += a[i][0] + a[i][1] + a[i][2] + a[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
int main() {
  float* d; hipMalloc(&d, 2048 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int random = 0; random < 2; ++random) {
    for (int rep = 0; rep < 4; ++rep) {
      const int iters = 100000, wgs = 512;   // ~170 ms per launch: long enough for the clock to settle
      hipEventRecord(e0, 0);
      if (random) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(512), 0, 0, d, iters);
      else hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(512), 0, 0, d, iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)wgs * 8 * iters * 32 * 2048.0;
      printf("%s operands: %.2f ms  %.1f TFLOP/s fp32 MFMA (%.1f%% of 157.3)\n", random ? "random  " : "constant", ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573);
    }
  }
  return 0;
}
