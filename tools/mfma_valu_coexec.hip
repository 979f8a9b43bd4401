// Do fp32 MFMAs and ordinary fp32 VALU work overlap on a SIMD?  512-thread workgroups: waves 0-3 (one per SIMD)
// stream v_mfma_f32_16x16x4_f32 (or bf16 16x16x16), waves 4-7 stream dependent-free v_fma_f32.  Reports the time of
// MFMA alone, VALU alone and both together:  together ~= max -> they overlap;  ~= sum -> they serialise.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int MODE, int KIND>   // MODE bit0: MFMA waves active, bit1: VALU waves active; KIND 0 fp32 MFMA, 1 bf16 MFMA
__global__ __launch_bounds__(512) void k(float* out, int iters, int valu_per_iter) {
  const int wave = threadIdx.x >> 6;
  float res = 0.f;
  if (wave < 4) {
    if (MODE & 1) {
      f32x4 a[8];
      for (int i = 0; i < 8; ++i) a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float x = threadIdx.x * 1e-3f, y = threadIdx.x * 2e-3f;
      s16x4 xs = {1, 2, 3, 4}, ys = {5, 6, 7, 8};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[i], 0, 0, 0);
            else a[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xs, ys, a[i], 0, 0, 0);
          }
      }
      for (int i = 0; i < 8; ++i) res += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    }
  } else {
    if (MODE & 2) {
      float v[16];
      for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-4f + i;
      const float c = 1.0001f, d = 1e-6f;
      for (int it = 0; it < iters; ++it)
        for (int r = 0; r < valu_per_iter; ++r) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], c, d);
        }
      for (int i = 0; i < 16; ++i) res += v[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int MODE, int KIND>
float run(float* d, int iters, int vpi) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, d, 100, vpi);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, d, iters, vpi);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int iters = 20000;
  // 32 MFMAs per iteration: fp32 = 32*32 = 1024 cycles; VALU: vpi*16 v_fma at 4 cycles = vpi*64 cycles per iteration
  for (int vpi = 4; vpi <= 16; vpi *= 2) {
    const float m = run<1, 0>(d, iters, vpi), v = run<2, 0>(d, iters, vpi), b = run<3, 0>(d, iters, vpi);
    printf("fp32 MFMA: alone %.2f ms | %3d v_fma/iter alone %.2f ms | together %.2f ms  (max %.2f, sum %.2f)\n", m, vpi * 16, v, b,
           m > v ? m : v, m + v);
  }
  for (int vpi = 4; vpi <= 16; vpi *= 2) {
    const float m = run<1, 1>(d, iters, vpi), v = run<2, 1>(d, iters, vpi), b = run<3, 1>(d, iters, vpi);
    printf("bf16 MFMA: alone %.2f ms | %3d v_fma/iter alone %.2f ms | together %.2f ms  (max %.2f, sum %.2f)\n", m, vpi * 16, v, b,
           m > v ? m : v, m + v);
  }
  return 0;
}
