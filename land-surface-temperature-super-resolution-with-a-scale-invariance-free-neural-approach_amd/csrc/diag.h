// Diagnostic build flavours of the MFMA kernels -- ONE place for every switch that changes what a kernel does in a
// non-shipping build.  The shipped library is built with none of them defined; tests/test_diag_builds.py compiles each flavour
// once (CPU box, no GPU needed) so that they keep building, and tools/build_ab.sh NAME -D... makes an A/B library of one.
//
//   -DSIFSR_DIAG_NOMFMA    the conv kernels without their window reads, transforms and MFMAs (results are zeros): what a kernel's
//                          data movement alone costs (tools/nomfma_sweep.sh, profiles/r02_nomfma_sweep.txt)
//   -DSIFSR_DIAG_CLOCK     shader-clock / 100 MHz stamps around the item loop of conv3x3_wino8_kernel, read back through
//                          sifsr_debug_timers() (tools/clock_probe.py: the in-kernel clock under load), and per-phase / per-role
//                          clock sums of conv3x3_bwd16_kernel through sifsr_debug_timers16() (tools/clock_probe16.py)
//   -DSIFSR_PK_MODE=0..3   form of the packed add / subtract of the Winograd transforms (common.h; 0 ships)
#pragma once

//   -DSIFSR_DIAG_W8_ABL=n  conv3x3_wino8_kernel with parts of its non-matrix work removed (wrong results; what each part costs):
//                          1 one window row read instead of four, 2 no input transform, 4 a quarter of the output transform
#ifndef SIFSR_DIAG_W8_ABL
#define SIFSR_DIAG_W8_ABL 0
#endif

#ifdef SIFSR_DIAG_NOMFMA
#define SIFSR_DIAG_SKIP_MATRIX_WORK(never_true) if (never_true)
#else
#define SIFSR_DIAG_SKIP_MATRIX_WORK(never_true)
#endif

#ifdef SIFSR_DIAG_CLOCK
#define SIFSR_DIAG_CLOCK_DECL __device__ unsigned long long sifsr_clk[4];
#define SIFSR_DIAG_CLOCK_BEGIN const unsigned long long ck0__ = __builtin_amdgcn_s_memtime(), cr0__ = __builtin_amdgcn_s_memrealtime();
#define SIFSR_DIAG_CLOCK_END(tid)                                                                       \
  {                                                                                                     \
    const unsigned long long ck1__ = __builtin_amdgcn_s_memtime(), cr1__ = __builtin_amdgcn_s_memrealtime(); \
    if ((tid) == 0) { atomicAdd(&sifsr_clk[0], ck1__ - ck0__); atomicAdd(&sifsr_clk[1], cr1__ - cr0__); }   \
  }
#define SIFSR_DIAG_CLOCK_READER                                                                                          \
  extern "C" __attribute__((visibility("default"))) int sifsr_debug_timers(unsigned long long* out4, int reset) {        \
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(sifsr_clk), 32) != hipSuccess) return 1;                                    \
    if (reset) { unsigned long long z[4] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sifsr_clk), z, 32) != hipSuccess) return 2; } \
    return 0;                                                                                                            \
  }
// conv3x3_bwd16_kernel: shader-clock ticks per phase and wave role, summed over workgroups (wave 0 of each role reports):
//   [0..3] input-gradient role: staging (incl. the wait for the tile's loads), contraction (+ epilogue), barrier wait, tile count;
//   [4..7] the same for the weight-gradient role;  [8], [9]: of the staging time, the wait for the prefetched loads per role
#define SIFSR_DIAG_CLOCK16_DECL                                                                                             \
  __device__ unsigned long long sifsr_clk16[10];                                                                          \
  extern "C" __attribute__((visibility("default"))) int sifsr_debug_timers16(unsigned long long* out10, int reset) {      \
    if (hipMemcpyFromSymbol(out10, HIP_SYMBOL(sifsr_clk16), 80) != hipSuccess) return 1;                                  \
    if (reset) { unsigned long long z[10] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sifsr_clk16), z, 80) != hipSuccess) return 2; } \
    return 0;                                                                                                             \
  }
// conv3x3_wino8_kernel: per item and wave team (waves 0 and 4 report; [0..5] the contract-first team, [6..11] the stage-first one):
//   staging, window reads (incl. the wait for them), input transform, MFMAs + output transform, epilogue + barrier wait, item count
#define SIFSR_DIAG_CLOCK8_DECL                                                                                              \
  __device__ unsigned long long sifsr_clk8[12];                                                                           \
  extern "C" __attribute__((visibility("default"))) int sifsr_debug_timers8(unsigned long long* out12, int reset) {       \
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(sifsr_clk8), 96) != hipSuccess) return 1;                                   \
    if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sifsr_clk8), z, 96) != hipSuccess) return 2; } \
    return 0;                                                                                                             \
  }
#define SIFSR_DIAG_ACC8_LOCALS unsigned long long dacc8__[6] = {0, 0, 0, 0, 0, 0};
#define SIFSR_DIAG_ACC8(i, d) dacc8__[(i)] += (unsigned long long)(d)   /* registers; one atomic per counter at the end of the kernel */
#define SIFSR_DIAG_ACC8_FLUSH(ro)                                                                          \
  if (lane == 0 && (wave8 & 3) == 0) {                                                                     \
    for (int i__ = 0; i__ < 6; ++i__) atomicAdd(&sifsr_clk8[(ro) + i__], dacc8__[i__]);                    \
  }
#define SIFSR_DIAG_ADD(var, d) var += (d)
#define SIFSR_DIAG_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SIFSR_DIAG_ACC16(i, d) do { if (lane == 0 && wave == 0) atomicAdd(&sifsr_clk16[(i)], (unsigned long long)(d)); } while (0)
#define SIFSR_DIAG_WAIT_LOADS(var)                                   \
  unsigned long long var = 0;                                        \
  {                                                                  \
    const unsigned long long w0__ = __builtin_amdgcn_s_memtime();    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
    var = __builtin_amdgcn_s_memtime() - w0__;                       \
  }
#else
#define SIFSR_DIAG_CLOCK_DECL
#define SIFSR_DIAG_CLOCK_BEGIN
#define SIFSR_DIAG_CLOCK_END(tid)
#define SIFSR_DIAG_CLOCK_READER
#define SIFSR_DIAG_CLOCK16_DECL
#define SIFSR_DIAG_CLOCK8_DECL
#define SIFSR_DIAG_ACC8_LOCALS
#define SIFSR_DIAG_ACC8(i, d)
#define SIFSR_DIAG_ACC8_FLUSH(ro)
#define SIFSR_DIAG_ADD(var, d)
#define SIFSR_DIAG_T(var)
#define SIFSR_DIAG_ACC16(i, d)
#define SIFSR_DIAG_WAIT_LOADS(var)
#endif
