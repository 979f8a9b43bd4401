// Input gradient AND weight gradient of a 16 -> 16 channel replicate-padded 3x3 convolution from ONE read of its operands
// (nn.Conv2d backward, model.py:135,138 -- the DoubleConvolution layers at full and half resolution: inbloc.bloc.3,
// ub3.convbloc.bloc.3, db1.resblock.doubleconv.bloc.0/.3).
//
// Why: with 4/9 of the matrix work gone (Winograd) these layers are HBM-bound in every pass, and the separate kernels read the
// same tensors again and again -- per layer the input gradient reads (g, y, y_below) and writes g_below (4 tensors), the weight
// gradient reads (y_below, g, y) once more (3 tensors).  Both passes consume exactly the same staged data:
//   dL/dy tile (18x18 halo, formed from (g, y) by the BatchNorm+ReLU backward while staging, never stored)  -> dgrad B operand
//   dL/dy tile (16x16 core) and a_in = relu(bn(y_below)) tile (18x18 halo)                                 -> wgrad operands
// so this kernel stages them once (4 tensors of HBM traffic instead of 7) and splits the matrix work by WAVE ROLE:
//   waves 0-3  input gradient,  Winograd F(2x2,3x3): the consumer of conv_mfma.hip (lane = (patch, cout quad); 16 ds_read_b128 of
//              the patch's 4x4 window, input transform in registers, 64 MFMAs per 16-patch group, output transform per xi-row),
//              transform-domain weights in LDS; epilogue = NHWC stores (+ residual addend) and the BatchNorm-backward sums of the
//              layer below (its raw y is the staged input tile in LDS);
//   waves 4-7  weight gradient, Winograd F(3x3,2x2): the per-lane register transforms of conv_wgrad_wino.hip (lane = (channel,
//              patch of a 4-patch k-step); 2 + 8 ds_read_b64 from channel planes, 16 MFMAs per k-step, 4 k-steps per wave and
//              tile), accumulators live across all tiles, added up through LDS into one slab per workgroup at the end.
// One of each role per SIMD: the two MFMA streams (64 MFMAs per wave and tile each) interleave on the matrix pipe and each
// role's transform / epilogue instructions issue under the other's MFMAs.  All eight waves stage: thread = (channel quad,
// 3 of the 324 halo pixels), the next tile's 9 float4 loads in flight during the contraction, LDS double-buffered -> ONE barrier
// per tile, and the two roles stage at different times (see tile_loop).  The replicate-border fold of the input gradient stays the separate kernel (dgrad_border_kernel).
#include "conv.h"

#include <stdlib.h>

SIFSR_DIAG_CLOCK16_DECL   // (diag.h: nothing in the shipped build; tools/clock_probe16.py reads the per-phase clock sums)

namespace {

constexpr int WPITCH = 20, WHALF = 10, WPLANE = 364;   // dgrad operand layout, as conv_mfma.hip (even / odd columns split)
constexpr int DPS = 260;                               // wgrad dy channel-plane stride (floats): >= 256, = 4 (mod 64)
constexpr int XPS = 324;                               // wgrad input channel-plane stride: 18*18 = 324 = 4 (mod 64)
constexpr int XPW = 18;
constexpr int DYQ_F4 = 4 * WPLANE;                     // float4 slots of the dgrad layout
constexpr int DYP_F = 16 * DPS, XP_F = 16 * XPS;       // floats
constexpr unsigned OOB = 0xFFFFFF00u;

typedef __attribute__((ext_vector_type(4))) unsigned u32x4b;
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ float4 bl4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4b v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
static __device__ __forceinline__ void bs4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v) {
  u32x4b u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, (int)soff, 0);
}
// activation access: byte offsets of the fp32 layout; HS (bf16 storage, the bf16 compute mode -- common.h) halves them
template <bool HS> static __device__ __forceinline__ float4 al4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (HS) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2b;
    const u32x2b v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(voff >> 1), (int)(soff >> 1), 0);
    return unpack_bf16x4(make_uint2(v.x, v.y));
  } else return bl4(r, voff, soff);
}
template <bool HS> static __device__ __forceinline__ void as4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v) {
  if constexpr (HS) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2b;
    const uint2 p = pack_bf16x4(v);
    u32x2b u; u.x = p.x; u.y = p.y;
    __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)(voff >> 1), (int)(soff >> 1), 0);
  } else bs4(r, voff, soff, v);
}
static __device__ __forceinline__ int xslot(int i) { return (i >> 2) + 4 * (i & 3); }

// DYM: where dL/dy comes from.  0: stored (a.g).  1: formed while staging from (g, y) by the BatchNorm+ReLU backward (bn_bwd4).
// 2 ("tail", ub3.convbloc.bloc.3): as 1, but g itself -- the input gradient of outlay (Conv2d 16 -> 1, model.py:605) -- is
// recomputed per staged pixel from a 20x20 tile of d loss / d sr (9 LDS reads, 36 FMAs per pixel and channel quad) instead of
// being written to and read back from HBM by tail_bwd_apply_kernel (fused_edges.hip).
// HS: every activation tensor (x, g, y, gin, addend, the border scratch) is stored as bf16
// POOL (DYM == 1): the upstream gradient is g + 0.25 * pool_gp[y/2][x/2] (Bwd16Args::pool_gp), added while staging
template <int DYM, bool HS, bool POOL = false>
__global__ __launch_bounds__(512, 2) void conv3x3_bwd16_kernel(const Bwd16Args a, const int ntiles, const int lgx, const int lgy) {
  constexpr bool DYF = DYM != 0, TAIL = DYM == 2;
  static_assert(!POOL || DYM == 1, "the pooling adjoint joins the (g, y) form");
  __shared__ float4 dyq[2][DYQ_F4];                       // dL/dy halo, [cout quad][pixel (even | odd columns)][4]
  __shared__ __align__(16) float dyp[2][DYP_F + 16];      // dL/dy core, channel planes
  __shared__ __align__(16) float xp[2][XP_F + 16];        // a_in halo, channel planes
  __shared__ float4 wlds[16 * 64];                        // transform-domain dgrad weights, [xi][lane]
  __shared__ float red[4][16][2];
  __shared__ float dtl[TAIL ? 2 * 400 : 4];               // tail: d loss / d sr around the tile, 20 x 20, zero outside the image
  __shared__ float4 wout[TAIL ? 36 : 1];                  // tail: outlay weights, [tap][channel quad]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_amdgcn_s_setprio(2);                          // chain kernel: above a weight gradient of another layer on the second stream
  const int H = a.H, W = a.W;
  const int tiles_x = W / 16, tiles_y = H / 16;           // H, W multiples of 16 (checked by the launcher)
  const unsigned npix = (unsigned)a.B * (unsigned)H * (unsigned)W;

  // ---- persistent XCD-aware tile walk (conv_mfma.hip)
  const int G = gridDim.x;
  const bool xcd_map = (G % 8 == 0) && (ntiles % 8 == 0);
  const int t_lo = xcd_map ? (blockIdx.x % 8) * (ntiles / 8) : 0;
  const int t_hi = xcd_map ? t_lo + ntiles / 8 : ntiles;
  const int t_step = xcd_map ? G / 8 : G;
  const int t_first = t_lo + (xcd_map ? blockIdx.x / 8 : blockIdx.x);
  auto tile_pos = [&](int tt, int& tb, int& txi, int& tyi) __attribute__((always_inline)) {
    if (lgx >= 0) { tyi = (tt >> lgx) & (tiles_y - 1); tb = tt >> (lgx + lgy); txi = (tt + tyi + tb) & (tiles_x - 1); }
    else { txi = tt % tiles_x; const int r = tt / tiles_x; tyi = r % tiles_y; tb = r / tiles_y; }
  };

  for (int i = tid; i < 16 * 64; i += 512) wlds[i] = ld4(a.wpack_wino + 4 * (size_t)i);
  if (TAIL && tid < 36) {
    const float* wo = a.tail_w + (4 * (tid & 3)) * 9 + (tid >> 2);   // outlay weight [1][16][3][3]
    wout[tid] = make_float4(wo[0], wo[9], wo[18], wo[27]);
  }

  constexpr unsigned PXB = HS ? 32u : 64u;                // bytes per pixel (16 channels)
  const __amdgpu_buffer_rsrc_t rg = mk_rsrc(TAIL ? a.y : a.g, npix * PXB);
  const __amdgpu_buffer_rsrc_t ry = mk_rsrc(DYF ? a.y : a.g, npix * PXB);
  const __amdgpu_buffer_rsrc_t rx = mk_rsrc(a.x, npix * PXB);
  const __amdgpu_buffer_rsrc_t rbd = mk_rsrc(DYF && a.dy_border ? a.dy_border : const_cast<float*>(a.g), npix * PXB);

  // ---- staging map: thread -> (channel quad cg, halo pixels pslot + 128 it, it < 3)
  const int cg = tid & 3, pslot = tid >> 2;
  int spy[3], spx[3];
  unsigned rel[3];                                        // byte offset of the slot relative to the halo origin (interior tiles)
  int lq[3], lp[3], lx[3];                                // LDS indices: dgrad layout (float4), dy plane (float, -1: not core), x plane
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    int p = pslot + 128 * it;
    if (p >= 324) p = 323;                                // lanes past the tile re-load the last pixel and store nothing
    spy[it] = p / 18; spx[it] = p - spy[it] * 18;
    rel[it] = (unsigned)(spy[it] * W + spx[it]) * 64u + (unsigned)cg * 16u;
    lq[it] = cg * WPLANE + spy[it] * WPITCH + (spx[it] & 1) * WHALF + (spx[it] >> 1);
    const bool core = spy[it] >= 1 && spy[it] <= 16 && spx[it] >= 1 && spx[it] <= 16;
    lp[it] = core ? cg * DPS + (spy[it] - 1) * 16 + (spx[it] - 1) : -1;
    lx[it] = cg * XPS + p;
  }
  const bool slot2 = pslot + 256 < 324;
  // POOL: low-resolution pixel of the slot relative to (tile row * 8 - 1, tile column * 8 - 1): ((spy + 1) >> 1, (spx + 1) >> 1)
  const int H2 = H >> 1, W2 = W >> 1;
  const __amdgpu_buffer_rsrc_t rgp = mk_rsrc(POOL ? a.pool_gp : a.x, POOL ? (npix >> 2) * PXB : 4u);
  unsigned relp[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) relp[it] = (unsigned)(((spy[it] + 1) >> 1) * W2 + ((spx[it] + 1) >> 1)) * 64u + (unsigned)cg * 16u;
  float4 pgp[POOL ? 3 : 1];
  float4 csc = make_float4(0.f, 0.f, 0.f, 0.f), csh = csc, ck1 = csc, ck0 = csc;
  if (DYF) { csc = ld4(a.coef + 4 * cg); csh = ld4(a.coef + 16 + 4 * cg); ck1 = ld4(a.coef + 32 + 4 * cg); ck0 = ld4(a.coef + 48 + 4 * cg); }
  const bool xraw = a.x_scale == nullptr;

  float4 pg[TAIL ? 1 : 3], py[DYF ? 3 : 1], px_[3];      // the tile in flight
  int st_b = 0, st_tx = 0, st_ty = 0;                     // ... and its position
  // tail: the d loss / d sr tile runs ONE TILE AHEAD of the others (its LDS copy has to be complete -- a barrier -- before
  // write_stage forms g from it): element tid of the 20 x 20 tile, fp32 in every mode
  const __amdgpu_buffer_rsrc_t rds = mk_rsrc(TAIL ? a.tail_dsr : a.x, TAIL ? npix * 4u : 4u);
  float pd = 0.f;
  const int dey = tid / 20, dex = tid - dey * 20;
  auto issue_d = [&](int tt) __attribute__((always_inline)) {
    int b, txi, tyi;
    tile_pos(tt, b, txi, tyi);
    const int gy = tyi * 16 - 2 + dey, gx = txi * 16 - 2 + dex;
    const bool inside = tid < 400 && gy >= 0 && gy < H && gx >= 0 && gx < W;
    pd = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rds, inside ? (int)((unsigned)((b * H + gy) * W + gx) * 4u) : (int)OOB, 0, 0));
  };
  auto write_d = [&](int buf) __attribute__((always_inline)) {
    if (tid < 400) dtl[buf * 400 + tid] = pd;
  };
  auto issue = [&](int tt) __attribute__((always_inline)) {
    tile_pos(tt, st_b, st_tx, st_ty);
    const int x0 = st_tx * 16 - 1, y0 = st_ty * 16 - 1;
    const bool interior = st_tx > 0 && st_ty > 0 && st_tx + 1 < tiles_x && st_ty + 1 < tiles_y;
    if (interior) {
      const unsigned soff = (unsigned)((st_b * H + y0) * W + x0) * 64u;
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        if (!TAIL) pg[it] = al4<HS>(rg, rel[it], soff);
        if (DYF) py[it] = al4<HS>(ry, rel[it], soff);
        px_[it] = al4<HS>(rx, rel[it], soff);
        if (POOL) pgp[it] = al4<HS>(rgp, relp[it], (unsigned)((st_b * H2 + st_ty * 8 - 1) * W2 + st_tx * 8 - 1) * 64u);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int gy = y0 + spy[it], gx = x0 + spx[it];
        const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const unsigned pc = (unsigned)((st_b * H + clampi(gy, 0, H - 1)) * W + clampi(gx, 0, W - 1)) * 64u + (unsigned)cg * 16u;
        if (!TAIL) pg[it] = al4<HS>(rg, inside ? pc : OOB, 0u);   // zero padding of dL/dy
        if (DYF) py[it] = al4<HS>(ry, inside ? pc : OOB, 0u);
        px_[it] = al4<HS>(rx, pc, 0u);                        // replicate padding of the forward input
        if (POOL) pgp[it] = al4<HS>(rgp, inside ? (unsigned)((st_b * H2 + (gy >> 1)) * W2 + (gx >> 1)) * 64u + (unsigned)cg * 16u : OOB, 0u);
      }
    }
  };
  auto write_stage = [&](int buf) __attribute__((always_inline)) {
    const int x0 = st_tx * 16 - 1, y0 = st_ty * 16 - 1;
    const bool interior = st_tx > 0 && st_ty > 0 && st_tx + 1 < tiles_x && st_ty + 1 < tiles_y;
    float4* const Q = dyq[buf];
    float* const P = dyp[buf];
    float* const X = xp[buf];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      if (it == 2 && !slot2) continue;
      float4 v = pg[TAIL ? 0 : it];
      if (TAIL) {
        // g[c](q) = sum_t w[c][t] * S_t(q), S_t(q) = sum of d loss / d sr over the output pixels whose (clamped) tap t reads q
        // (the adjoint of replicate padding, fused_edges.hip outlay_gather)
        // (the adjoint of the clamp is separable: 12 FMAs on the 3x3 neighbourhood n of d loss / d sr, no branch -- fused_edges.hip)
        const float* D = dtl + buf * 400 + (spy[it] + 1) * 20 + spx[it] + 1;
        const int gy = y0 + spy[it], gx = x0 + spx[it];
        const float ymf = gy == 0 ? 1.f : 0.f, ypf = gy == H - 1 ? 1.f : 0.f, xmf = gx == 0 ? 1.f : 0.f, xpf = gx == W - 1 ? 1.f : 0.f;
        float n[3][3], cx[3][3], S[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) n[dy][dx] = D[(dy - 1) * 20 + dx - 1];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          cx[dy][0] = fmaf(xmf, n[dy][1], n[dy][2]);
          cx[dy][1] = n[dy][1];
          cx[dy][2] = fmaf(xpf, n[dy][1], n[dy][0]);
        }
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          S[0 + tx] = fmaf(ymf, cx[1][tx], cx[2][tx]);
          S[3 + tx] = cx[1][tx];
          S[6 + tx] = fmaf(ypf, cx[1][tx], cx[0][tx]);
        }
        f32x2 gl = {0.f, 0.f}, gh = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float4 wq = wout[t * 4 + cg];
          gl = __builtin_elementwise_fma((f32x2){wq.x, wq.y}, (f32x2){S[t], S[t]}, gl);
          gh = __builtin_elementwise_fma((f32x2){wq.z, wq.w}, (f32x2){S[t], S[t]}, gh);
        }
        v = make_float4(gl[0], gl[1], gh[0], gh[1]);
      }
      if (POOL) {
        const float4 q = pgp[it];
        v.x = fmaf(0.25f, q.x, v.x); v.y = fmaf(0.25f, q.y, v.y); v.z = fmaf(0.25f, q.z, v.z); v.w = fmaf(0.25f, q.w, v.w);
      }
      if (DYF) {
        v = bn_bwd4(v, py[it], csc, csh, ck1, ck0);
        if (!interior) {
          const int gy = y0 + spy[it], gx = x0 + spx[it];
          const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
          if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);   // bn_bwd4 of the zeros that were loaded is not 0
          const bool edge = gy == 0 || gy == H - 1 || gx == 0 || gx == W - 1;
          if (a.dy_border != nullptr && inside && lp[it] >= 0 && edge)   // dL/dy on the image border, for the border-fold kernel
            as4<HS>(rbd, (unsigned)((st_b * H + gy) * W + gx) * 64u + (unsigned)cg * 16u, 0u, v);
        }
      }
      Q[lq[it]] = v;
      if (lp[it] >= 0) {
        float* d = P + lp[it];
        d[0] = v.x; d[4 * DPS] = v.y; d[8 * DPS] = v.z; d[12 * DPS] = v.w;   // channel 4 cg + r -> plane cg + 4 r
      }
      const float4 xv = px_[it];   // RAW: the weight-gradient waves apply the folded BatchNorm + ReLU when they read their windows (they
      float* e = X + lx[it];      // have the issue slots to spare), the input-gradient waves read y_below itself for the BatchNorm sums
      e[0] = xv.x; e[4 * XPS] = xv.y; e[8 * XPS] = xv.z; e[12 * XPS] = xv.w;
    }
  };

  const bool dgrad_role = wave8 < 4;
  const int wave = wave8 & 3;
  const int kq = lane >> 4, i16 = lane & 15;
  // ---- input-gradient state
  const int pxp = lane & 7, pyl = (lane >> 3) & 1;
  const int g0 = wave * 4;
  const int lbase = kq * WPLANE + (g0 + 2 * pyl) * WPITCH + pxp;
  const __amdgpu_buffer_rsrc_t rd = mk_rsrc(a.gin, npix * PXB);
  const __amdgpu_buffer_rsrc_t rad = mk_rsrc(a.addend ? a.addend : a.gin, npix * PXB);
  const bool bn_stats = a.bn_y != nullptr;
  float4 bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsh = bsc;
  if (bn_stats) { bsc = ld4(a.bn_scale + 4 * kq); bsh = ld4(a.bn_shift + 4 * kq); }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  // ---- weight-gradient state (the accumulators are declared in the weight-gradient branch)
  const int pa_off = xslot(i16) * DPS + 2 * kq;           // my dy plane, my patch of a k-step (kq = patch of the 4)
  const int pb_off = xslot(i16) * XPS + 2 * kq;
  // folded BatchNorm of the layer below for MY input channel (lane i16 of the B operand): a_in = relu(x * xs + xb)
  const f32x2 xs2 = xraw ? (f32x2){1.f, 1.f} : (f32x2){a.x_scale[i16], a.x_scale[i16]};
  const f32x2 xb2 = xraw ? (f32x2){0.f, 0.f} : (f32x2){a.x_shift[i16], a.x_shift[i16]};

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto lo2 = [](f32x4 v) { return (f32x2){v[0], v[1]}; };
  auto hi2 = [](f32x4 v) { return (f32x2){v[2], v[3]}; };
  auto acc2 = [](f32x4& y, f32x2 l, f32x2 h, bool minus) {
    const f32x2 yl = minus ? pk_sub((f32x2){y[0], y[1]}, l) : pk_add((f32x2){y[0], y[1]}, l);
    const f32x2 yh = minus ? pk_sub((f32x2){y[2], y[3]}, h) : pk_add((f32x2){y[2], y[3]}, h);
    y = (f32x4){yl[0], yl[1], yh[0], yh[1]};
  };

  // The tile loop, instantiated once per role (two loops, not one loop with a role branch inside: the weight-gradient
  // accumulators are then live only in the weight-gradient waves' code and the input-gradient waves' transforms do not spill).
  // Both roles execute exactly one barrier per tile, so the workgroup's barrier counts match.
  // The two roles of a SIMD are OUT OF PHASE: the input-gradient waves stage their share of tile j+1 BEFORE they contract
  // tile j, the weight-gradient waves after -- each role's staging (vector / LDS instructions) issues under the other role's
  // MFMAs instead of both staging while the matrix pipe idles (conv_wino8.hip does the same with its two wave teams).
  //   iteration j:   A: stage(j+1) -> loads(j+2) -> contract(j) -> barrier      B: contract(j) -> stage(j+1) -> loads(j+2) -> barrier
  // The barrier of iteration j publishes tile j+1 (other buffer) and retires every read of tile j's buffer.
  auto tile_loop = [&](const bool stage_first, auto&& contract) __attribute__((always_inline)) {
    int t = t_first, buf = 0;
    if (t >= t_hi) { __syncthreads(); return; }           // (wlds barrier; workgroups without tiles exist only for tiny problems)
    if (TAIL) issue_d(t);
    issue(t);
    if (TAIL) {
      write_d(0);
      if (t + t_step < t_hi) issue_d(t + t_step);
    }
    __syncthreads();                                      // wlds (tail: wout, d loss / d sr of tile 0)
    write_stage(0);
    if (TAIL) write_d(1);
    int cb = st_b, txi = st_tx, tyi = st_ty;              // the tile being contracted
    if (t + t_step < t_hi) issue(t + t_step);
    if (TAIL && t + 2 * t_step < t_hi) issue_d(t + 2 * t_step);
    __syncthreads();                                      // tile 0 staged
    while (true) {
      const int t_next = t + t_step;
      const bool more = t_next < t_hi;
      const int nb_ = st_b, ntx_ = st_tx, nty_ = st_ty;   // position of tile j+1 (in flight)
      const int ro_ = stage_first ? 0 : 4;
      (void)ro_;
      SIFSR_DIAG_T(c0);
      if (stage_first && more) {
        SIFSR_DIAG_WAIT_LOADS(cw);
        SIFSR_DIAG_ACC16(8, cw);
        write_stage(buf ^ 1);
        if (TAIL) write_d(buf);                           // d loss / d sr of tile j+2 (tile j's copy was last read before the previous barrier)
        if (t_next + t_step < t_hi) issue(t_next + t_step);
        if (TAIL && t_next + 2 * t_step < t_hi) issue_d(t_next + 2 * t_step);
      }
      SIFSR_DIAG_T(c1);
      contract(buf, cb, txi, tyi);
      SIFSR_DIAG_T(c2);
      if (!stage_first && more) {
        SIFSR_DIAG_WAIT_LOADS(cw);
        SIFSR_DIAG_ACC16(9, cw);
        write_stage(buf ^ 1);
        if (TAIL) write_d(buf);
        if (t_next + t_step < t_hi) issue(t_next + t_step);
        if (TAIL && t_next + 2 * t_step < t_hi) issue_d(t_next + 2 * t_step);
      }
      SIFSR_DIAG_T(c3);
      if (!more) break;
      __syncthreads();
      SIFSR_DIAG_T(c4);
      SIFSR_DIAG_ACC16(ro_ + 0, (c1 - c0) + (c3 - c2)); SIFSR_DIAG_ACC16(ro_ + 1, c2 - c1); SIFSR_DIAG_ACC16(ro_ + 2, c4 - c3); SIFSR_DIAG_ACC16(ro_ + 3, 1);
      cb = nb_; txi = ntx_; tyi = nty_;
      t = t_next;
      buf ^= 1;
    }
  };

  if (dgrad_role) {
    tile_loop(true, [&](const int buf, const int cb, const int txi, const int tyi) {
      // ======================= input gradient: Winograd F(2x2,3x3), one 16-patch group (4 tile rows) per wave =======================
      f32x4 Y[2][2] = {{zero4, zero4}, {zero4, zero4}};
      SIFSR_DIAG_SKIP_MATRIX_WORK(a.B < 0) {   // (diag.h: nothing in the shipped build)
      f32x2 dl[4][4], dh[4][4];
      const float4* Lg = dyq[buf] + lbase;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 v0 = Lg[r * WPITCH], v1 = Lg[r * WPITCH + WHALF], v2 = Lg[r * WPITCH + 1], v3 = Lg[r * WPITCH + WHALF + 1];
        dl[r][0] = (f32x2){v0.x, v0.y}; dh[r][0] = (f32x2){v0.z, v0.w};
        dl[r][1] = (f32x2){v1.x, v1.y}; dh[r][1] = (f32x2){v1.z, v1.w};
        dl[r][2] = (f32x2){v2.x, v2.y}; dh[r][2] = (f32x2){v2.z, v2.w};
        dl[r][3] = (f32x2){v3.x, v3.y}; dh[r][3] = (f32x2){v3.z, v3.w};
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {     // rows of B^T d: [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
        const f32x2 l0 = pk_sub(dl[0][c], dl[2][c]), l1 = pk_add(dl[1][c], dl[2][c]), l2 = pk_sub(dl[2][c], dl[1][c]), l3 = pk_sub(dl[1][c], dl[3][c]);
        const f32x2 h0 = pk_sub(dh[0][c], dh[2][c]), h1 = pk_add(dh[1][c], dh[2][c]), h2 = pk_sub(dh[2][c], dh[1][c]), h3 = pk_sub(dh[1][c], dh[3][c]);
        dl[0][c] = l0; dl[1][c] = l1; dl[2][c] = l2; dl[3][c] = l3;
        dh[0][c] = h0; dh[1][c] = h1; dh[2][c] = h2; dh[3][c] = h3;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {     // columns, same pattern
        const f32x2 l0 = pk_sub(dl[r][0], dl[r][2]), l1 = pk_add(dl[r][1], dl[r][2]), l2 = pk_sub(dl[r][2], dl[r][1]), l3 = pk_sub(dl[r][1], dl[r][3]);
        const f32x2 h0 = pk_sub(dh[r][0], dh[r][2]), h1 = pk_add(dh[r][1], dh[r][2]), h2 = pk_sub(dh[r][2], dh[r][1]), h3 = pk_sub(dh[r][1], dh[r][3]);
        dl[r][0] = l0; dl[r][1] = l1; dl[r][2] = l2; dl[r][3] = l3;
        dh[r][0] = h0; dh[r][1] = h1; dh[r][2] = h2; dh[r][3] = h3;
      }
#pragma unroll
      for (int ar = 0; ar < 4; ++ar) {
        f32x4 Mc[4];
        float4 wr[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) wr[b] = wlds[(4 * ar + b) * 64 + lane];
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].x, dl[ar][b][0], zero4, 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].y, dl[ar][b][1], Mc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].z, dh[ar][b][0], Mc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].w, dh[ar][b][1], Mc[b], 0, 0, 0);
        // t = M A: t0 = M0 + M1 + M2, t1 = M1 - (M2 + M3); the first reads of fresh MFMA results are compiler-visible adds
        // (it pads the matrix-pipe -> VALU hazard for those, not for inline assembly -- conv_mfma.hip)
        const f32x4 t0 = Mc[0] + Mc[1] + Mc[2], u = Mc[2] + Mc[3];
        const f32x2 t0l = lo2(t0), t0h = hi2(t0);
        const f32x2 t1l = pk_sub(lo2(Mc[1]), lo2(u)), t1h = pk_sub(hi2(Mc[1]), hi2(u));
        if (ar <= 2) { acc2(Y[0][0], t0l, t0h, false); acc2(Y[0][1], t1l, t1h, false); }
        if (ar == 1) { acc2(Y[1][0], t0l, t0h, false); acc2(Y[1][1], t1l, t1h, false); }
        if (ar >= 2) { acc2(Y[1][0], t0l, t0h, true); acc2(Y[1][1], t1l, t1h, true); }
      }
      }
      // ---- epilogue: the lane's 2x2 output pixels x 4 input channels
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int oy = o >> 1, ox = o & 1;
        const int yy = tyi * 16 + g0 + 2 * pyl + oy, xx = txi * 16 + 2 * pxp + ox;
        f32x4 v = Y[oy][ox];
        const unsigned pixo = (unsigned)((cb * H + yy) * W + xx) * 64u + (unsigned)kq * 16u;
        if (a.addend != nullptr) {
          const float4 ad = al4<HS>(rad, pixo, 0u);
          v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
        }
        if (HS) {   // the BatchNorm sums below are those of the STORED (bf16-rounded) gradient
          const float4 vr = round_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
          v = (f32x4){vr.x, vr.y, vr.z, vr.w};
        }
        if (!(bn_stats && a.store_dz)) as4<HS>(rd, pixo, 0u, make_float4(v[0], v[1], v[2], v[3]));
        if (bn_stats) {   // dz = g_in * [y_below * scale + shift > 0]; sum dz, sum dz * y_below.  y_below = the staged x tile (raw) in LDS:
          // channel 4 kq + r lives in plane kq + 4 r, pixel (row + 1, column + 1) of the halo.  (Round 3: these were four global
          // loads per lane requested before the MFMAs -- under this kernel's memory load they came back after the contraction
          // and the input-gradient waves spent 2/3 of their phase waiting for them, tools/clock_probe16.py)
          const float* yp_ = xp[buf] + kq * XPS + (g0 + 2 * pyl + oy + 1) * XPW + 2 * pxp + ox + 1;
          const float yy4[4] = {yp_[0], yp_[4 * XPS], yp_[8 * XPS], yp_[12 * XPS]};
          const float scv[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, shv[4] = {bsh.x, bsh.y, bsh.z, bsh.w};
          float dzv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dz = fmaf(yy4[r], scv[r], shv[r]) > 0.f ? v[r] : 0.f;
            dzv[r] = dz;
            s1[r] += dz; s2[r] = fmaf(dz, yy4[r], s2[r]);
          }
          if (a.store_dz) as4<HS>(rd, pixo, 0u, make_float4(dzv[0], dzv[1], dzv[2], dzv[3]));
        }
      }
    });
    for (int w = 0; w < 7; ++w) __syncthreads();          // the weight-gradient waves' slab reduction (below): 1 + 3 * 2 barriers
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile_loop(false, [&](const int buf, const int cb, const int txi, const int tyi) {
      (void)cb; (void)txi; (void)tyi;
      // ======================= weight gradient: Winograd F(3x3,2x2), 4 k-steps (of 4 patches) per wave =======================
      const float* const pa = dyp[buf] + pa_off;
      const float* const pb = xp[buf] + pb_off;
      SIFSR_DIAG_SKIP_MATRIX_WORK(a.B < 0)   // (diag.h: nothing in the shipped build)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int h = wave + 4 * j;                       // k-step: patch row h >> 1 (0..7), patch columns 4 (h & 1) + (0..3)
        const int pr = h >> 1, pcb = 4 * (h & 1);
        float ug[16], vv[16];
        {
          const float* q = pa + (2 * pr) * 16 + 2 * pcb;
          const f32x2 d0 = *reinterpret_cast<const f32x2*>(q), d1 = *reinterpret_cast<const f32x2*>(q + 16);
          const f32x2 r1 = pk_add(d0, d1), r2 = pk_sub(d0, d1);               // rows of G g: [g0, g0 + g1, g0 - g1, g1]
          const f32x2 rows[4] = {d0, r1, r2, d1};
#pragma unroll
          for (int u = 0; u < 4; ++u) {                                       // columns: (x, y) -> [x, x + y, x - y, y]
            ug[4 * u + 0] = rows[u][0]; ug[4 * u + 1] = rows[u][0] + rows[u][1];
            ug[4 * u + 2] = rows[u][0] - rows[u][1]; ug[4 * u + 3] = rows[u][1];
          }
        }
        {
          const float* q = pb + (2 * pr) * XPW + 2 * pcb;
          f32x2 wl[4], wh[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { wl[u] = *reinterpret_cast<const f32x2*>(q + u * XPW); wh[u] = *reinterpret_cast<const f32x2*>(q + u * XPW + 2); }
          if (!xraw) {
            const f32x2 z2 = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              wl[u] = __builtin_elementwise_max(__builtin_elementwise_fma(wl[u], xs2, xb2), z2);
              wh[u] = __builtin_elementwise_max(__builtin_elementwise_fma(wh[u], xs2, xb2), z2);
            }
          }
          // rows: [d0 - d2, d1 + d2, d2 - d1, d3 - d1] on both pixel pairs
          const f32x2 rl[4] = {pk_sub(wl[0], wl[2]), pk_add(wl[1], wl[2]), pk_sub(wl[2], wl[1]), pk_sub(wl[3], wl[1])};
          const f32x2 rh[4] = {pk_sub(wh[0], wh[2]), pk_add(wh[1], wh[2]), pk_sub(wh[2], wh[1]), pk_sub(wh[3], wh[1])};
#pragma unroll
          for (int u = 0; u < 4; ++u) {                                       // columns: same pattern on (x0, x1 | x2, x3)
            vv[4 * u + 0] = rl[u][0] - rh[u][0]; vv[4 * u + 1] = rl[u][1] + rh[u][0];
            vv[4 * u + 2] = rh[u][0] - rl[u][1]; vv[4 * u + 3] = rh[u][1] - rl[u][1];
          }
        }
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ug[tt], vv[tt], acc[tt], 0, 0, 0);
      }
    });
    // ---- the four weight-gradient waves' accumulators are added up through LDS in a fixed order (waves 1, 2, 3 into wave 0;
    // the input-gradient waves execute the matching barriers), then ONE slab per workgroup, [tap][lane][4] (the layout of
    // both weight-gradient kernel families for one block pair)
    float4* const comb = dyq[0];
    __syncthreads();                                      // every read of the tile buffers is done
    for (int w = 1; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) comb[tt * 64 + lane] = make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) { const float4 v = comb[tt * 64 + lane]; acc[tt][0] += v.x; acc[tt][1] += v.y; acc[tt][2] += v.z; acc[tt][3] += v.w; }
      }
      __syncthreads();
    }
    if (wave == 0) {   // the output transform per lane (conv.h wino_wgrad_taps): a tap-domain slab, 9 * 256 floats
      float* slab = a.slabs + (size_t)blockIdx.x * (9 * 256);
      f32x4 tap[9];
      wino_wgrad_taps([&](int xi) { return acc[xi]; }, tap);
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) st4(slab + tt * 256 + lane * 4, make_float4(tap[tt][0], tap[tt][1], tap[tt][2], tap[tt][3]));
    }
  }
  // ---- BatchNorm-backward partials of the layer below: one row per workgroup
  if (a.stat_partials != nullptr) {
    if (dgrad_role) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = s1[r], v = s2[r];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
        if (i16 == 0) { red[wave][4 * kq + r][0] = u; red[wave][4 * kq + r][1] = v; }
      }
    }
    __syncthreads();
    if (tid < 16) {
      float u = 0.f, v = 0.f;
      for (int w = 0; w < 4; ++w) { u += red[w][tid][0]; v += red[w][tid][1]; }
      float* o = a.stat_partials + ((size_t)blockIdx.x * 16 + tid) * 2;
      o[0] = u; o[1] = v;
    }
  }
}

}  // namespace

bool conv3x3_bwd16_applies(int B, int H, int W) {
  static const int off = getenv("SIFSR_NO_BWD16") ? atoi(getenv("SIFSR_NO_BWD16")) : 0;   // 1: separate input- / weight-gradient kernels (A/B)
  return !off && B >= 1 && H >= 32 && W >= 32 && H % 16 == 0 && W % 16 == 0 && (size_t)B * H * W * 64 < ((size_t)1 << 32) - 4096;
}

int conv3x3_bwd16_grid(int B, int H, int W) {
  const int ntiles = B * (H / 16) * (W / 16);
  static const int dbg = getenv("SIFSR_DBG_BWD16_GRID") ? atoi(getenv("SIFSR_DBG_BWD16_GRID")) : 256;   // one workgroup per CU
  if (ntiles <= dbg) return ntiles;
  const int rounds = (ntiles + dbg - 1) / dbg;
  int g = (ntiles + rounds - 1) / rounds;
  g = (g + 7) & ~7;
  return g < ntiles ? g : ntiles;
}

int launch_conv3x3_bwd16(const Bwd16Args& a, hipStream_t s) {
  if (!conv3x3_bwd16_applies(a.B, a.H, a.W)) return SIFSR_ERR_SHAPE;
  if (!a.x || !a.wpack_wino || !a.gin || !a.slabs) return SIFSR_ERR_ARG;
  const bool tail = a.tail_dsr != nullptr;
  if (tail ? (!a.tail_w || !a.y || !a.coef) : !a.g) return SIFSR_ERR_ARG;
  if ((a.y != nullptr) != (a.coef != nullptr) || (a.x_scale != nullptr) != (a.x_shift != nullptr)) return SIFSR_ERR_ARG;
  if (a.stat_partials != nullptr && (!a.bn_y || !a.bn_scale || !a.bn_shift)) return SIFSR_ERR_ARG;
  // the BatchNorm sums are taken from the staged input tile: the layer below IS the layer whose raw output is the input
  if (a.stat_partials != nullptr && (a.bn_y != a.x || a.bn_scale != a.x_scale || a.bn_shift != a.x_shift)) return SIFSR_ERR_ARG;
  if (a.store_dz && (a.stat_partials == nullptr || a.addend != nullptr)) return SIFSR_ERR_ARG;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
  const int tx_ = a.W / 16, ty_ = a.H / 16, ntiles = a.B * tx_ * ty_;
  const int lgx = (pow2(tx_) && pow2(ty_)) ? lg(tx_) : -1, lgy = lgx >= 0 ? lg(ty_) : -1;
  const dim3 grid(conv3x3_bwd16_grid(a.B, a.H, a.W)), block(512);
  if (a.pool_gp != nullptr && (tail || a.y == nullptr)) return SIFSR_ERR_ARG;
  if (a.pool_gp != nullptr) {
    if (a.half) hipLaunchKernelGGL((conv3x3_bwd16_kernel<1, true, true>), grid, block, 0, s, a, ntiles, lgx, lgy);
    else hipLaunchKernelGGL((conv3x3_bwd16_kernel<1, false, true>), grid, block, 0, s, a, ntiles, lgx, lgy);
    SIFSR_LAUNCH_CHECK();
    return SIFSR_OK;
  }
  if (a.half) {
    if (tail) hipLaunchKernelGGL((conv3x3_bwd16_kernel<2, true>), grid, block, 0, s, a, ntiles, lgx, lgy);
    else if (a.y != nullptr) hipLaunchKernelGGL((conv3x3_bwd16_kernel<1, true>), grid, block, 0, s, a, ntiles, lgx, lgy);
    else hipLaunchKernelGGL((conv3x3_bwd16_kernel<0, true>), grid, block, 0, s, a, ntiles, lgx, lgy);
  } else {
    if (tail) hipLaunchKernelGGL((conv3x3_bwd16_kernel<2, false>), grid, block, 0, s, a, ntiles, lgx, lgy);
    else if (a.y != nullptr) hipLaunchKernelGGL((conv3x3_bwd16_kernel<1, false>), grid, block, 0, s, a, ntiles, lgx, lgy);
    else hipLaunchKernelGGL((conv3x3_bwd16_kernel<0, false>), grid, block, 0, s, a, ntiles, lgx, lgy);
  }
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
