// 3x3 convolution / input gradient in the Winograd F(2x2, 3x3) domain for 32 and 64 output channels: the variant in which
// ALL EIGHT waves of the workgroup stage and contract.
//
// conv_mfma.hip splits the workgroup into four producer and four consumer waves.  Its Winograd consumers need 210-244
// registers (64 for the window / its transform, 64 for the next item's weights, 32 + 16 per patch group for accumulators), and
// a kernel's register allocation is uniform over its waves: one workgroup = 2 waves per SIMD is all a CU holds, of which only
// ONE issues MFMAs -- nothing covers that wave's LDS latency, transform and epilogue (matrix pipe 42-47 % busy, timers: a
// third of a consumer's time is not MFMA).  Here every wave does both jobs:
//   * staging: thread -> (channel quad, 3 halo pixels); the loads of item j + 1 are issued when item j has gone to LDS and land
//     during the MFMA phase of item j; they are transformed (BatchNorm+ReLU of the producing layer, or the BatchNorm+ReLU
//     backward of this layer: DYF) and written to the other LDS buffer when the wave has finished its MFMAs -- one barrier per item;
//   * contraction: wave = (cout block, half of the tile's patch groups) for 64 output channels, (cout block, one of the
//     four patch groups) for 32 -- half the accumulators per wave of conv_mfma.hip's split, which pays for the staging registers.
// Two MFMA-issuing waves per SIMD, out of phase by whatever the barrier leaves them, overlap one's window reads, transform
// and output transform with the other's MFMAs.
// Same LDS layout (parity-split halo rows, conv_mfma.hip), weight pack (pack_wino_kernel), lane maps, per-item output transform
// and epilogue (split destinations, residual addend, BatchNorm statistics, dL/dy border scratch) as the consumer there.
#include "conv.h"

#include <stdlib.h>

SIFSR_DIAG_CLOCK_DECL   // (diag.h: nothing in the shipped build)
SIFSR_DIAG_CLOCK8_DECL
namespace {

constexpr int PW = 18;
constexpr int WPITCH = 20, WHALF = 10, WPLANE = 364;   // see conv_mfma.hip

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
static __device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v) {
  u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, (int)soff, 0);
}
constexpr unsigned OOB = 0xFFFFFF00u;   // per-lane offset beyond any tensor: buffer loads return 0, stores are dropped

// DEPTH: items in flight per thread (register sets); 1 everywhere (see the launcher)
template <int NB, bool ZERO_PAD, bool DYF, int DEPTH>
__global__ __launch_bounds__(512, 2) void conv3x3_wino8_kernel(const ConvArgs a, const int ntiles, const int lgx, const int lgy) {
  static_assert(NB == 2 || NB == 4, "32 or 64 output channels");
  static_assert(!DYF || ZERO_PAD, "the fused BatchNorm backward belongs to the input-gradient pass");
  constexpr int NGRP = NB == 4 ? 2 : 1;      // 16-patch groups (4 tile rows each) per wave and item

  __shared__ float4 lds[2][4 * WPLANE];
  __shared__ float red[8][16][2];
  // per-channel staging coefficients of every channel block: [q][sc | sh | k1 | k0][channel quad] (the k's: DYF only).  A load
  // from global memory inside write_stage would expose an L2 round trip per item; carried with the item they would hold 8-16
  // registers across the MFMA phase.
  __shared__ float4 coef[8][4][4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_amdgcn_s_setprio(2);             // above a co-running weight-gradient kernel (see conv_mfma.hip)
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
  const int NQ = a.NQ;
  const unsigned npix = (unsigned)a.B * (unsigned)H * (unsigned)W;

  // ---- persistent tile walk (as conv_mfma.hip: XCD-contiguous ranges, column rotated per row)
  const int G = gridDim.x;
  const bool xcd_map = (G % 8 == 0) && (ntiles % 8 == 0);
  const int t_lo = xcd_map ? (blockIdx.x % 8) * (ntiles / 8) : 0;
  const int t_hi = xcd_map ? t_lo + ntiles / 8 : ntiles;
  const int t_step = xcd_map ? G / 8 : G;
  int t = t_lo + (xcd_map ? blockIdx.x / 8 : blockIdx.x);
  auto tile_pos = [&](int tt, int& tb, int& txi, int& tyi) {
    if (lgx >= 0) { tyi = (tt >> lgx) & (tiles_y - 1); tb = tt >> (lgx + lgy); txi = (tt + tyi + tb) & (tiles_x - 1); }
    else { txi = tt % tiles_x; const int r = tt / tiles_x; tyi = r % tiles_y; tb = r / tiles_y; }
  };

  // =========================================== staging ===========================================
  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.src[0].ptr, npix * a.src[0].C * 4u);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.src[1].ptr ? a.src[1].ptr : a.src[0].ptr, npix * (a.src[1].ptr ? a.src[1].C : a.src[0].C) * 4u);
  const __amdgpu_buffer_rsrc_t rsy = make_rsrc(DYF ? a.bw_y : a.src[0].ptr, npix * a.src[0].C * 4u);
  const __amdgpu_buffer_rsrc_t rbd = make_rsrc(DYF && a.bw_border ? a.bw_border : const_cast<float*>(a.src[0].ptr), npix * a.src[0].C * 4u);
  const int cg = tid & 3, pslot = tid >> 2;         // channel quad, halo pixels pslot + 128 * it (18 x 18 = 324 pixels)
  auto hpix = [&](int it, int& py, int& px_) {      // halo coordinates of my it-th pixel (lanes past the plane: the last pixel)
    int p = pslot + 128 * it;
    if (p >= PW * PW) p = PW * PW - 1;
    py = (p * 57) >> 10;                            // p / 18 for p < 324
    px_ = p - py * PW;
  };
  int prel[3], lslot[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    int py, px_;
    hpix(it, py, px_);
    prel[it] = py * W + px_;
    lslot[it] = cg * WPLANE + py * WPITCH + (px_ & 1) * WHALF + (px_ >> 1);
  }
  const bool st2 = pslot + 256 < PW * PW;           // my third pixel exists
  int pixv[3];                  // per-lane pixel index of the tile being fetched (-1: outside, dgrad only)
  int pix_base = 0;             // scalar pixel offset added to pixv (interior tiles)
  auto set_tile = [&](int tt) {
    int tb, txi, tyi;
    tile_pos(tt, tb, txi, tyi);
    const int tx0 = txi * 16, ty0 = tyi * 16;
    const bool interior = txi > 0 && tyi > 0 && txi + 1 < tiles_x && tyi + 1 < tiles_y;
    if (interior) {
      pix_base = (tb * H + ty0 - 1) * W + tx0 - 1;
#pragma unroll
      for (int it = 0; it < 3; ++it) pixv[it] = prel[it];
    } else {
      pix_base = 0;
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        int py, px_;
        hpix(it, py, px_);
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + px_;
        const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int pc = (tb * H + clampi(gy, 0, H - 1)) * W + clampi(gx, 0, W - 1);
        pixv[it] = (ZERO_PAD && !inside) ? -1 : pc;
      }
    }
  };
  float4 stgS[DEPTH][3], styS[DYF ? DEPTH : 1][DYF ? 3 : 1];
  int it_t = t, it_q = 0;       // next item to fetch; past the end the last item is fetched again (never used)
  int tSS[DEPTH], qSS[DEPTH];   // (tile, channel block) of the item held in register set k
  auto issue_loads = [&](const int k) {
    float4 (&stg)[3] = stgS[k];
    float4 (&sty)[DYF ? 3 : 1] = styS[DYF ? k : 0];
    const int q = it_q;
    tSS[k] = it_t; qSS[k] = it_q;
    const bool first = q < a.src[0].nq;
    const int C = first ? a.src[0].C : a.src[1].C;
    const int lgc = 31 - __builtin_clz((unsigned)C) + 2;                       // log2(C * 4 bytes)
    const int ch = (first ? a.src[0].coff + 16 * q : a.src[1].coff + 16 * (q - a.src[0].nq));
    const unsigned soff = ((unsigned)pix_base << lgc) + (unsigned)ch * 4u;
    const __amdgpu_buffer_rsrc_t r = first ? rs0 : rs1;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const unsigned voff = (ZERO_PAD && pixv[it] < 0) ? OOB : (((unsigned)pixv[it] << lgc) + (unsigned)cg * 16u);
      stg[it] = bload4(r, voff, soff);
      if (DYF) sty[it] = bload4(rsy, voff, soff);
    }
    if (it_q + 1 < NQ) ++it_q;
    else if (it_t + t_step < t_hi) { it_t += t_step; it_q = 0; set_tile(it_t); }
  };
  if (tid < NQ * 16) {   // (NQ <= 8, checked by the launcher)
    const int q = tid >> 4, kind = (tid >> 2) & 3, c4 = tid & 3;
    const bool first = q < a.src[0].nq;
    const int C = first ? a.src[0].C : a.src[1].C;
    const int ch = (first ? a.src[0].coff + 16 * q : a.src[1].coff + 16 * (q - a.src[0].nq));
    const float* scp = first ? a.src[0].scale : a.src[1].scale;
    const float* shp = first ? a.src[0].shift : a.src[1].shift;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DYF) v = ld4(a.bw_coef + kind * C + ch + 4 * c4);          // [sc | sh | k1 | k0], C floats each
    else if (scp != nullptr && kind < 2) v = ld4((kind ? shp : scp) + ch + 4 * c4);
    coef[q][kind][c4] = v;
  }
  auto write_stage = [&](float4* Lb, const int k) {
    const float4 (&stg)[3] = stgS[k];
    const float4 (&sty)[DYF ? 3 : 1] = styS[DYF ? k : 0];
    const int q = qSS[k], tS = tSS[k];
    const bool first = q < a.src[0].nq;
    const bool praw = !DYF && (first ? a.src[0].scale : a.src[1].scale) == nullptr;
    const float4 psc = coef[q][0][cg], psh = coef[q][1][cg];
    float4 k1A = psc, k0A = psc;
    if (DYF) { k1A = coef[q][2][cg]; k0A = coef[q][3][cg]; }
    int wb = 0, wy0 = 0, wx0 = 0;
    bool winterior = true;
    if (DYF) {
      int txi, tyi;
      tile_pos(tS, wb, txi, tyi);
      wy0 = tyi * 16 - 1; wx0 = txi * 16 - 1;
      winterior = txi > 0 && tyi > 0 && txi + 1 < tiles_x && tyi + 1 < tiles_y;
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      float4 v = stg[it];
      const bool mine = it < 2 || st2;
      if (DYF) {
        v = bn_bwd4(v, sty[it], psc, psh, k1A, k0A);
        if (!winterior) {
          // halo pixels outside the image are zero padding of dL/dy (bn_bwd4 of the zeros they loaded is not 0); the image-border
          // pixels of the tile's own 16x16 core go to bw_border for the border-fold kernel
          int py, px_;
          hpix(it, py, px_);
          const int gy = wy0 + py, gx = wx0 + px_;
          const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
          if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
          const bool core = py >= 1 && py <= 16 && px_ >= 1 && px_ <= 16;
          const bool edge = gy == 0 || gy == H - 1 || gx == 0 || gx == W - 1;
          if (a.bw_border != nullptr && inside && core && edge && mine)
            bstore4(rbd, (unsigned)((wb * H + gy) * W + gx) * (unsigned)a.src[0].C * 4u + (unsigned)(16 * q + 4 * cg) * 4u, 0u, v);
        }
      } else if (!praw) v = bn_relu4(v, psc, psh);
      if (mine) Lb[lslot[it]] = v;
    }
  };

  // =========================================== contraction ===========================================
  const int nb0 = NB == 4 ? (wave8 & 3) : (wave8 & 1);                 // my cout block
  const int g0 = NB == 4 ? (wave8 >> 2) * 8 : (wave8 >> 1) * 4;        // my first tile row
  const int kq = lane >> 4;
  const int pxp = lane & 7, pyl = (lane >> 3) & 1;
  const int lbase = kq * WPLANE + (g0 + 2 * pyl) * WPITCH + pxp;
  const __amdgpu_buffer_rsrc_t rd0 = make_rsrc(a.dst[0].ptr, npix * a.dst[0].C * 4u);
  const __amdgpu_buffer_rsrc_t rd1 = make_rsrc(a.dst[1].ptr, npix * a.dst[1].C * 4u);
  const __amdgpu_buffer_rsrc_t rad = make_rsrc(a.addend ? a.addend : a.dst[0].ptr, npix * (a.addend ? a.addC : a.dst[0].C) * 4u);
  const __amdgpu_buffer_rsrc_t rww = make_rsrc(a.wpack, (unsigned)(NB * 16) * (unsigned)(NQ * 16) * 64u);
  float4 wq[16];
  auto wsoff = [&](int q_, int xi) { return (unsigned)(((nb0 * NQ + q_) * 16 + xi)) * 1024u; };
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) wq[xi] = bload4(rww, (unsigned)lane * 16u, wsoff(0, xi));

  float s1[4], s2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) s1[r] = s2[r] = 0.f;
  f32x4 Y[NGRP][2][2];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NGRP; ++g)
#pragma unroll
    for (int o = 0; o < 4; ++o) Y[g][o >> 1][o & 1] = zero4;
  auto lo2 = [](f32x4 v) { return (f32x2){v[0], v[1]}; };
  auto hi2 = [](f32x4 v) { return (f32x2){v[2], v[3]}; };
  auto acc2 = [](f32x4& y, f32x2 l, f32x2 h, bool minus) {
    const f32x2 yl = minus ? pk_sub((f32x2){y[0], y[1]}, l) : pk_add((f32x2){y[0], y[1]}, l);
    const f32x2 yh = minus ? pk_sub((f32x2){y[2], y[3]}, h) : pk_add((f32x2){y[2], y[3]}, h);
    y = (f32x4){yl[0], yl[1], yh[0], yh[1]};
  };

  // ---- prologue: item 0 to LDS, item 1 in flight
  const int n_items = ((t_hi - t + t_step - 1) / t_step) * NQ;
  set_tile(t);
  issue_loads(0);
  __syncthreads();   // the coefficient table is complete
  write_stage(lds[0], 0);
#pragma unroll
  for (int i = 1; i <= DEPTH; ++i) issue_loads(i % DEPTH);   // item i lives in set i % DEPTH
  __syncthreads();

  SIFSR_DIAG_CLOCK_BEGIN
  SIFSR_DIAG_ACC8_LOCALS
  int buf = 0, q = 0;
  auto do_item = [&](const int j, const int ks) {   // ks = (j + 1) % DEPTH: the register set of item j + 1
    const bool last_q = q + 1 == NQ;
    const int t_next = t + t_step;
    const bool more = j + 1 < n_items;
    const int qn = last_q ? 0 : q + 1;
    const float4* L = lds[buf];
    // Waves w and w + 4 share a SIMD.  The upper four stage item j + 1 BEFORE they contract item j, the lower four after: each
    // team's staging (vector / LDS work, no MFMA) then runs under the other team's MFMAs instead of both SIMD-mates staging at
    // the same time with the matrix pipe idle.  (lds[buf ^ 1] is free since the barrier that closed item j - 1; either way a
    // wave's loads have had one full item to land.)
    const bool stage_first = wave8 >= 4;
    unsigned long long dt_rd = 0, dt_tr = 0, dt_mm = 0;
    (void)dt_rd; (void)dt_tr; (void)dt_mm;
    SIFSR_DIAG_T(c0);
    // staging at the higher wave priority: it is a few dozen instructions that decide when the next loads go out; contraction one
    // step below (still above a co-running weight-gradient kernel at 0).  +0.4 % on the step against the opposite order.
    __builtin_amdgcn_s_setprio(3);
    if (stage_first && more) {
      write_stage(lds[buf ^ 1], ks);
      issue_loads(ks);          // item j + 1 + DEPTH
    }
    __builtin_amdgcn_s_setprio(2);
    SIFSR_DIAG_T(c1);
    SIFSR_DIAG_SKIP_MATRIX_WORK(a.B < 0)   // (diag.h: nothing in the shipped build)
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
      // ---- the patch's 4x4 input window -> V = B^T d B (in place), as conv_mfma.hip's Winograd consumer
      f32x2 dl[4][4], dh[4][4];
      const float4* Lg = L + lbase + g * 4 * WPITCH;
      SIFSR_DIAG_T(g0_);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (diag.h SIFSR_DIAG_W8_ABL & 1: one window row read instead of four -- the cost of the window reads)
        const float4* Lr = (SIFSR_DIAG_W8_ABL & 1) ? Lg : Lg + r * WPITCH;
        const float4 v0 = Lr[0], v1 = Lr[WHALF], v2 = Lr[1], v3 = Lr[WHALF + 1];
        dl[r][0] = (f32x2){v0.x, v0.y}; dh[r][0] = (f32x2){v0.z, v0.w};
        dl[r][1] = (f32x2){v1.x, v1.y}; dh[r][1] = (f32x2){v1.z, v1.w};
        dl[r][2] = (f32x2){v2.x, v2.y}; dh[r][2] = (f32x2){v2.z, v2.w};
        dl[r][3] = (f32x2){v3.x, v3.y}; dh[r][3] = (f32x2){v3.z, v3.w};
      }
      SIFSR_DIAG_T(g1_);
      if (!(SIFSR_DIAG_W8_ABL & 2)) {   // (diag.h: & 2 skips the input transform)
#pragma unroll
      for (int c = 0; c < 4; ++c) {     // rows: [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
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
      }
      SIFSR_DIAG_T(g2_);
      // ---- per xi-row: 16 MFMAs (4 chains), the output transform of the previous row behind them (software-pipelined)
      f32x4 M[2][4];
      auto out_row = [&](const int ar, const f32x4 (&Mr)[4]) {
        if (SIFSR_DIAG_W8_ABL & 4) { if (ar == 3) { Y[g][0][0] += Mr[0]; Y[g][0][1] += Mr[1]; Y[g][1][0] += Mr[2]; Y[g][1][1] += Mr[3]; } return; }   // (diag.h: & 4: the four xi-rows chained into one accumulator set, one sum at the end)
        // the first reads of fresh MFMA results are plain vector sums: the compiler pads that hazard, not the one of an asm statement
        const f32x4 t0 = Mr[0] + Mr[1] + Mr[2], u = Mr[2] + Mr[3];
        const f32x2 t0l = lo2(t0), t0h = hi2(t0);
        const f32x2 t1l = pk_sub(lo2(Mr[1]), lo2(u)), t1h = pk_sub(hi2(Mr[1]), hi2(u));
        if (ar <= 2) { acc2(Y[g][0][0], t0l, t0h, false); acc2(Y[g][0][1], t1l, t1h, false); }
        if (ar == 1) { acc2(Y[g][1][0], t0l, t0h, false); acc2(Y[g][1][1], t1l, t1h, false); }
        if (ar >= 2) { acc2(Y[g][1][0], t0l, t0h, true); acc2(Y[g][1][1], t1l, t1h, true); }
      };
#pragma unroll
      for (int ar = 0; ar < 4; ++ar) {
        f32x4 (&Mc)[4] = M[(SIFSR_DIAG_W8_ABL & 4) ? 1 : (ar & 1)];
        float4 wr[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) wr[b] = wq[4 * ar + b];
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].x, dl[ar][b][0], ((SIFSR_DIAG_W8_ABL & 4) && ar > 0) ? Mc[b] : zero4, 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].y, dl[ar][b][1], Mc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].z, dh[ar][b][0], Mc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].w, dh[ar][b][1], Mc[b], 0, 0, 0);
        if (g == NGRP - 1) {
          __builtin_amdgcn_sched_barrier(0);   // the next item's weights, xi-row by xi-row behind the MFMAs that used this row's
#pragma unroll
          for (int b = 0; b < 4; ++b) wq[4 * ar + b] = bload4(rww, (unsigned)lane * 16u, wsoff(qn, 4 * ar + b));
        }
        if (ar > 0) out_row(ar - 1, M[(ar - 1) & 1]);
      }
      out_row(3, M[1]);
      SIFSR_DIAG_T(g3_);
      SIFSR_DIAG_ADD(dt_rd, g1_ - g0_); SIFSR_DIAG_ADD(dt_tr, g2_ - g1_); SIFSR_DIAG_ADD(dt_mm, g3_ - g2_);
    }
    __builtin_amdgcn_s_setprio(3);
    SIFSR_DIAG_T(c2);

    // ---- item j + 1 -> the other buffer (everybody finished reading it before the previous barrier), item j + 2 in flight
    if (!stage_first && more) {
      write_stage(lds[buf ^ 1], ks);
      issue_loads(ks);
    }

    if (last_q) {
      // ---- tile epilogue: the lane's 2x2 output pixels x 4 channels per group
      int cb, txi, tyi;
      tile_pos(t, cb, txi, tyi);
      const bool do_stats = a.stat_partials != nullptr;
      const bool d0 = nb0 < a.dst_split;
      const int dC = d0 ? a.dst[0].C : a.dst[1].C;
      const __amdgpu_buffer_rsrc_t rd = d0 ? rd0 : rd1;
      const unsigned chb = (unsigned)((d0 ? a.dst[0].coff + 16 * nb0 : a.dst[1].coff + 16 * (nb0 - a.dst_split)) + 4 * kq) * 4u;
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int oy = o >> 1, ox = o & 1;
          const int yy = tyi * 16 + g0 + 4 * g + 2 * pyl + oy, xx = txi * 16 + 2 * pxp + ox;
          const bool ok = yy < H && xx < W;
          f32x4 v = Y[g][oy][ox];
          Y[g][oy][ox] = zero4;
          const unsigned pixo = (unsigned)((cb * H + yy) * W + xx);
          if (a.addend != nullptr) {
            const float4 ad = bload4(rad, ok ? pixo * (unsigned)a.addC * 4u + (unsigned)(16 * nb0 + 4 * kq) * 4u : OOB, 0u);
            v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
          }
          bstore4(rd, ok ? pixo * (unsigned)dC * 4u + chb : OOB, 0u, make_float4(v[0], v[1], v[2], v[3]));
          if (do_stats) {
            if (!ok) v = zero4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[r] += v[r]; s2[r] = fmaf(v[r], v[r], s2[r]); }
          }
        }
      }
    }
    SIFSR_DIAG_T(c3);
    __syncthreads();   // item j + 1 is staged; lds[buf] may be refilled
    SIFSR_DIAG_T(c4);
    SIFSR_DIAG_ACC8(0, (c1 - c0) + (c3 - c2)); SIFSR_DIAG_ACC8(1, dt_rd); SIFSR_DIAG_ACC8(2, dt_tr); SIFSR_DIAG_ACC8(3, dt_mm);
    SIFSR_DIAG_ACC8(4, c4 - c3); SIFSR_DIAG_ACC8(5, 1);
    buf ^= 1;
    if (last_q) { t = t_next; q = 0; } else ++q;
  };
  for (int j = 0; j < n_items; j += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
      if (j + k < n_items) do_item(j + k, (k + 1) % DEPTH);   // uniform over the workgroup
  }

  SIFSR_DIAG_CLOCK_END(tid)
  SIFSR_DIAG_ACC8_FLUSH(wave8 >= 4 ? 6 : 0)
  // ---- per-workgroup BatchNorm partials (sum, sum of squares) over all tiles this workgroup produced ----
  if (a.stat_partials != nullptr) {
    const int px = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float u = s1[r], v = s2[r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
      if (px == 0) { red[wave8][4 * kq + r][0] = u; red[wave8][4 * kq + r][1] = v; }
    }
    __syncthreads();
    if (tid < NB * 16) {
      const int nb = tid >> 4, cc = tid & 15;
      float u = 0.f, v = 0.f;
      // waves of cout block nb: wave8 & (NB - 1) == nb, in increasing order
      for (int w = nb; w < 8; w += NB) { u += red[w][cc][0]; v += red[w][cc][1]; }
      float* o = a.stat_partials + ((size_t)blockIdx.x * (NB * 16) + tid) * 2;
      o[0] = u; o[1] = v;
    }
  }
}

}  // namespace

bool conv3x3_wino8_applies(int nb, int nq) {
  static const int off = getenv("SIFSR_NO_WINO8") ? atoi(getenv("SIFSR_NO_WINO8")) : 0;   // 1: conv_mfma.hip's producer / consumer kernels (A/B)
  return !off && (nb == 2 || nb == 4) && nq <= 8;
}

// a.wpack must already point at the Winograd-domain pack; grid as for the other Winograd variants (one workgroup per CU)
int launch_conv3x3_wino8(const ConvArgs& a, int nb, int zero_pad, bool dyf, int grid, int ntiles, int lgx, int lgy, hipStream_t s) {
  const dim3 g(grid), block(512);
  static const int dbg_depth = getenv("SIFSR_DBG_WINO8_DEPTH") ? atoi(getenv("SIFSR_DBG_WINO8_DEPTH")) : 0;   // 1 / 2: force (A/B)
  // measured with two items in flight on the one-channel-block layers (where an item is shortest): 1 % slower on the step --
  // as in conv_mfma.hip, load latency is not what these kernels wait for.  Kept as an A/B knob.
  const bool deep = nb == 2 && dbg_depth == 2;
#define SIFSR_W8L(NBV, DV)                                                                                                        \
  {                                                                                                                               \
    if (dyf) hipLaunchKernelGGL((conv3x3_wino8_kernel<NBV, true, true, DV>), g, block, 0, s, a, ntiles, lgx, lgy);               \
    else if (zero_pad) hipLaunchKernelGGL((conv3x3_wino8_kernel<NBV, true, false, DV>), g, block, 0, s, a, ntiles, lgx, lgy);    \
    else hipLaunchKernelGGL((conv3x3_wino8_kernel<NBV, false, false, DV>), g, block, 0, s, a, ntiles, lgx, lgy);                 \
    SIFSR_LAUNCH_CHECK();                                                                                                         \
    return SIFSR_OK;                                                                                                              \
  }
  if (nb == 2 && deep) SIFSR_W8L(2, 2)
  if (nb == 2) SIFSR_W8L(2, 1)
  if (nb == 4) SIFSR_W8L(4, 1)
#undef SIFSR_W8L
  return SIFSR_ERR_SHAPE;
}

SIFSR_DIAG_CLOCK_READER
