// Shared declarations for the gfx950 (MI355X / CDNA4) SIF-CNN-SR kernels.
//
// Data layout: every activation *inside* the model is NHWC fp32
// ([B][H][W][C], C in {16,32,64}); the model input (B,2,H,W) and output (B,1,H,W) keep the
// reference's NCHW layout (model.py:608-645).  Parameters use the reference's own layouts
// (conv OIHW, BN vectors) inside one flat fp32 buffer in `model.parameters()` order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SIFSR_OK 0
#define SIFSR_ERR_SHAPE 1001      // unsupported / inconsistent shape
#define SIFSR_ERR_ARG 1002        // null pointer, bad enum
#define SIFSR_ERR_WORKSPACE 1003  // workspace too small

typedef __attribute__((ext_vector_type(4))) float f32x4;

// Kernels on the serial chain of the backward pass raise their waves' instruction-issue priority: a weight-gradient
// kernel of the previous layer may be running beside them on the library's second stream (engine.hip SideLane) at the
// default priority 0, and whatever it delays here delays the whole chain.  No effect when nothing runs beside them.
#define SIFSR_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)

#define SIFSR_LAUNCH_CHECK()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- activation storage: fp32, or bf16 in the bf16 compute mode (BASELINE.json config 5) ----
// Every activation-like tensor INSIDE the model (raw conv outputs, pooled / residual / upsampled tensors, the gradients with
// respect to them, the image-border scratch of dL/dy) is stored as NHWC bf16 in that mode: half the HBM bytes of every pass.
// Kernels are templated on HS ("half storage"); pointers stay `float*` in the interfaces (the tensors live in the same
// workspace regions, using the first half of each), element offsets are in ELEMENTS.  Arithmetic stays fp32: values are
// widened on load (exact) and rounded to nearest-even on store.
typedef __attribute__((ext_vector_type(2))) __bf16 sifsr_bf16x2;
static __device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  sifsr_bf16x2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;          // v_cvt_pk_bf16_f32, RNE
  return __builtin_bit_cast(unsigned, p);
}
static __device__ __forceinline__ uint2 pack_bf16x4(float4 v) { return make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
static __device__ __forceinline__ float4 unpack_bf16x4(uint2 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}
static __device__ __forceinline__ float round_bf16(float a) { return (float)(__bf16)a; }
static __device__ __forceinline__ float4 round_bf16x4(float4 v) { return unpack_bf16x4(pack_bf16x4(v)); }
template <bool HS> static __device__ __forceinline__ float4 ldA4(const float* base, size_t e) {   // elements e .. e+3 (e % 4 == 0)
  if constexpr (HS) return unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + e));
  else return *reinterpret_cast<const float4*>(base + e);
}
template <bool HS> static __device__ __forceinline__ void stA4(float* base, size_t e, float4 v) {
  if constexpr (HS) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + e) = pack_bf16x4(v);
  else *reinterpret_cast<float4*>(base + e) = v;
}
template <bool HS> static __device__ __forceinline__ float ldA1(const float* base, size_t e) {
  if constexpr (HS) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[e] << 16);
  else return base[e];
}
// what a consumer of a stored value will read back: the value itself (fp32 storage) or its bf16 rounding
template <bool HS> static __device__ __forceinline__ float4 as_stored4(float4 v) { if constexpr (HS) return round_bf16x4(v); else return v; }

// Host side: which storage the launch functions of the NON-convolution kernels (BatchNorm reductions, pooling / upsampling and
// their adjoints, the thin first / last convs, the fused tail) instantiate.  The 3x3 convolution launches carry the compute mode
// in their argument blocks; everything else asks this thread-local switch, which sifsr_engine_forward / _backward (and the
// `_bf16` C-ABI entry points of single operators) set for the duration of one call through a HalfStorageScope.
bool sifsr_half_storage();
struct HalfStorageScope {
  bool prev;
  explicit HalfStorageScope(bool on);
  ~HalfStorageScope();
};

// relu(fma(v, scale, shift)) on 4 channels: BatchNorm (folded to scale/shift) + ReLU applied when a
// consumer loads a raw conv output (model.py:136-137 / :139-140).
// Written on 2-vectors so that hipcc emits v_pk_fma_f32 / v_pk_max_f32 (two lanes of fp32 per instruction): on gfx950
// every VALU instruction of ANY wave on a SIMD stalls that SIMD's matrix pipe for its duration
// (tools/mfma_valu_coexec.hip: MFMA + VALU time = the sum, not the max), so the staging transform's instruction count
// is paid in full inside the MFMA kernels.  Same fp32 results as the scalar form (fma then max, per element).
typedef __attribute__((ext_vector_type(2))) float f32x2;
static __device__ __forceinline__ float4 bn_relu4(float4 v, float4 sc, float4 sh) {
  const f32x2 z = {0.f, 0.f};
  f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
  lo = __builtin_elementwise_max(__builtin_elementwise_fma(lo, (f32x2){sc.x, sc.y}, (f32x2){sh.x, sh.y}), z);
  hi = __builtin_elementwise_max(__builtin_elementwise_fma(hi, (f32x2){sc.z, sc.w}, (f32x2){sh.z, sh.w}), z);
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// a - b / a + b on a register pair as ONE packed instruction.  (hipcc emits v_pk_add_f32 for most sums of two float2 values but
// scalarises every float2 difference into two v_sub_f32 -- twice the issue slots on a SIMD whose matrix pipe waits for every
// vector instruction.)  Exact, like the operations they replace.  SIFSR_PK_MODE selects the form (A/B builds, tools/build_ab.sh):
//   0  inline assembly v_pk_add_f32 (neg modifiers for the difference).  An asm statement is opaque to LLVM's hazard
//      recognizer: nothing pads a matrix-pipe hazard between it and the MFMAs around it; tools/isa_lint.py checks the
//      compiled code instead (tests/test_isa_hazards.py).
//   1  the same followed by `s_nop 1`: two wait states before ANY later instruction can read the result -- what LLVM itself
//      inserts between a vector write and an MFMA that reads it -- so the result -> MFMA hazard is padded by construction.
//   2  compiler-visible: fma(b, -+1, a) = one v_pk_fma_f32, the +-1 out of a scalar asm move so that LLVM cannot fold the product
//      away (with literal constants it does, and falls back to scalarised subtractions).  Round-3 finding: correct and fully
//      padded by the compiler, but it reschedules the transforms and the 64-channel Winograd kernels go from 255 registers to
//      256 + 85..136 spilled -- kept for A/B only.
//   3  two scalar adds per pair (round-2 A/B: 7-8 % slower per layer beside v_mfma_f32_16x16x4_f32).
// Measured on one MI355X, same device, interleaved (round 3, whole SR2 step at batch 64): mode 0 9,905 patches/s, mode 1 9,570
// (-3.4 %: the two wait states after each of ~120 packed adds per work item are not hidden), mode 2 8,645 (-12.7 %: spills in
// the MFMA loops).  Mode 0 ships; the lint covers fall-through order AND every branch edge (loop back-edges included) of the
// exact objects that are linked into the library.
#ifndef SIFSR_PK_MODE
#define SIFSR_PK_MODE 0
#endif
#if SIFSR_PK_MODE == 3
static __device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return (f32x2){a[0] - b[0], a[1] - b[1]}; }
static __device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return (f32x2){a[0] + b[0], a[1] + b[1]}; }
#elif SIFSR_PK_MODE == 2
static __device__ __forceinline__ f32x2 pk_unit(bool minus) {
  float r;
  if (minus) asm("s_mov_b32 %0, -1.0" : "=s"(r)); else asm("s_mov_b32 %0, 1.0" : "=s"(r));
  return (f32x2){r, r};
}
static __device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return __builtin_elementwise_fma(b, pk_unit(true), a); }
static __device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return __builtin_elementwise_fma(b, pk_unit(false), a); }
#else
#if SIFSR_PK_MODE == 1
#define SIFSR_PK_PAD "\n\ts_nop 1"
#else
#define SIFSR_PK_PAD ""
#endif
static __device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" SIFSR_PK_PAD : "=v"(r) : "v"(a), "v"(b));
  return r;
}
static __device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {   // same reason: not every float2 sum comes out packed
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" SIFSR_PK_PAD : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#endif

// BatchNorm + ReLU BACKWARD applied while a consumer stages its operand (the input-gradient and weight-gradient
// convolutions of layer l read g_l = dL/d relu(bn(y_l)) and y_l instead of a stored dL/dy_l):
//   z  = fma(y, sc, sh)                      the forward's pre-activation, bit for bit (same fma) -> same ReLU mask
//   dy = sc * (z > 0 ? g : 0) + k1 * z + k0
// with sc = gamma*invstd, sh = beta - mean*sc, k1 = -invstd * sum(dz*xhat) / N, k0 = -sc * sum(dz) / N - k1 * beta
// (bn_bwd_finalize*_kernel; nn.BatchNorm2d backward, model.py:136-137: dy = sc * (dz - mean(dz) - xhat * mean(dz*xhat)),
// xhat = (z - beta) / gamma -- written on z so that no division by gamma appears and gamma = 0 gives dy = 0).
static __device__ __forceinline__ float4 bn_bwd4(float4 g, float4 y, float4 sc, float4 sh, float4 k1, float4 k0) {
  f32x2 zl = __builtin_elementwise_fma((f32x2){y.x, y.y}, (f32x2){sc.x, sc.y}, (f32x2){sh.x, sh.y});
  f32x2 zh = __builtin_elementwise_fma((f32x2){y.z, y.w}, (f32x2){sc.z, sc.w}, (f32x2){sh.z, sh.w});
  const f32x2 tl = __builtin_elementwise_fma((f32x2){k1.x, k1.y}, zl, (f32x2){k0.x, k0.y});
  const f32x2 th = __builtin_elementwise_fma((f32x2){k1.z, k1.w}, zh, (f32x2){k0.z, k0.w});
  const f32x2 dl = {zl.x > 0.f ? g.x : 0.f, zl.y > 0.f ? g.y : 0.f};
  const f32x2 dh = {zh.x > 0.f ? g.z : 0.f, zh.y > 0.f ? g.w : 0.f};
  const f32x2 ol = __builtin_elementwise_fma((f32x2){sc.x, sc.y}, dl, tl);
  const f32x2 oh = __builtin_elementwise_fma((f32x2){sc.z, sc.w}, dh, th);
  return make_float4(ol.x, ol.y, oh.x, oh.y);
}

// ---------------------------------------------------------------------------------------------
// Network table (== reference state_dict order, model.py:596-605; checked by tests/test_capi_symbols.py)
// ---------------------------------------------------------------------------------------------
#define SIFSR_NUM_BN_LAYERS 17
#define SIFSR_NUM_PARAMS 282705

struct LayerInfo {
  int cin, cout;
  int level;       // 0: HxW, 1: /2, 2: /4, 3: /8
  int w_off;       // offset of conv weight (OIHW) in the flat parameter buffer
  int gamma_off;   // BN weight
  int beta_off;    // BN bias
  int run_off;     // offset in the flat running-stat buffer: [mean C][var C]
  int ch_off;      // offset in per-channel scratch vectors (sum of cout of previous layers)
  int wpack_off;   // offset in packed-weight buffers (fwd and dgrad share the offset table)
};

enum {
  L_IN0 = 0, L_IN3, L_D1A, L_D1B, L_D1C, L_D2A, L_D2B, L_D2C, L_D3A, L_D3B, L_D3C,
  L_U1A, L_U1B, L_U2A, L_U2B, L_U3A, L_U3B
};

struct NetTable {
  LayerInfo L[SIFSR_NUM_BN_LAYERS];
  int out_w_off, out_b_off;  // outlay weight (1,16,3,3) and bias
  int total_params;          // 282705
  int total_running;         // sum 2*cout
  int total_channels;        // sum cout
  int total_wpack;           // sum 9*cin*cout over MFMA layers (all but L_IN0)
};

const NetTable& sifsr_net();
