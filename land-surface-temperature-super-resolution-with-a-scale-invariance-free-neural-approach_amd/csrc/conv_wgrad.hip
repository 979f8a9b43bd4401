// Weight gradient of the replicate-padded 3x3 convolution on the fp32 matrix cores:
//   dW[co][ci][t] = sum_{b,y,x} dy[b,y,x,co] * a_in[b, clamp(y+ty), clamp(x+tx), ci]
// (the reduction ATen/cuDNN performs for nn.Conv2d backward-weight, model.py:135,138,507).
//
// GEMM view per tap: dW_t (Cout x Cin) = dY^T (Cout x P) * A_t (P x Cin), P = B*H*W pixels.
// v_mfma_f32_16x16x4_f32 with the 4-deep K index = 4 consecutive pixels of a row:
//   A operand: lane (i = co, k = pixel)  <- dy   tile  in LDS  [pixel][Cout + pad]
//   B operand: lane (j = ci, k = pixel)  <- a_in halo tile in LDS [pixel][Cin + pad], shifted by tap
// Row strides are = 16 (mod 32) floats so both ds_read_b32 patterns are bank-conflict free.
// A workgroup walks 8x16-pixel tiles (persistent grid), keeps its dW partial in MFMA accumulators,
// and writes ONE slab per workgroup; wgrad_reduce_kernel sums the slabs in a fixed order
// (deterministic, no float atomics) and scatters to the OIHW gradient.
#include "conv.h"

namespace {

constexpr int WT_ROWS = 8;                 // tile rows
constexpr int WPW = 18, WPH = WT_ROWS + 2; // halo tile
constexpr int WPIX_IN = WPW * WPH;         // 180
constexpr int WPIX_OUT = 16 * WT_ROWS;     // 128

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
// activation loads: offsets are those of the fp32 layout; HS (bf16 storage, the bf16 compute mode -- common.h) halves them
template <bool HS> static __device__ __forceinline__ float4 wg_aload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (HS) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2w;
    const u32x2w v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(voff >> 1), (int)(soff >> 1), 0);
    return unpack_bf16x4(make_uint2(v.x, v.y));
  } else {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4w;
    const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
}
static __device__ __forceinline__ float4 wg_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// bf16 operand mode (BASELINE.json config 5): the LDS tiles stay fp32; a wave reads the four 4-pixel k-steps of a
// tile row (the same conflict-free ds_read_b32 pattern as the fp32 loop), rounds them to bf16 (RNE) and issues ONE
// v_mfma_f32_16x16x16_bf16 over the row's 16 pixels where the fp32 mode issues four v_mfma_f32_16x16x4_f32.
// K index 4*kg + j of the bf16 MFMA <-> pixel kg + 4*j of the row (any bijection works as long as A and B agree).
typedef __attribute__((ext_vector_type(4))) short wg_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 wg_bf16x2;
static __device__ __forceinline__ wg_s16x4 wg_pack4(float a, float b, float c, float d) {
  wg_bf16x2 lo, hi;
  lo[0] = (__bf16)a; lo[1] = (__bf16)b; hi[0] = (__bf16)c; hi[1] = (__bf16)d;
  const uint2 u = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
  return __builtin_bit_cast(wg_s16x4, u);
}

// DYF: a.dy holds g = dL/d relu(bn(y)); dL/dy is formed while the dy tile is staged (bn_bwd4, common.h) from g, the
// layer's raw conv output a.dy_y and the coefficients a.dy_coef -- the BatchNorm-backward elementwise pass is gone.
template <int NBO, int NBI, bool BF16, bool DYF>   // cout blocks, cin blocks handled by one workgroup (cin chunk = blockIdx.y)
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const WgradArgs a, const int lgx, const int lgy) {
  constexpr int WO = NBO >= 2 ? 2 : 1, WI = NBI >= 2 ? 2 : 1, WP = 4 / (WO * WI);
  constexpr int NBO_W = NBO / WO, NBI_W = NBI / WI;
  constexpr int CSO = NBO * 16 + (NBO % 2 == 0 ? 16 : 0);   // = 16 mod 32
  constexpr int CSI = NBI * 16 + (NBI % 2 == 0 ? 16 : 0);
  constexpr int NT = NBO_W * NBI_W * 9;                       // accumulator tiles per wave
  constexpr int QO = NBO * 4, QI = NBI * 4;                   // channel quads per pixel (dy / input chunk)
  constexpr int PPO = 256 / QO, PPI = 256 / QI;               // pixels staged per pass by the workgroup
  constexpr int NIO = WPIX_OUT / PPO;                         // prefetch float4s per thread, dy tile
  constexpr int NII = (WPIX_IN + PPI - 1) / PPI;              // ... input halo tile (last pass partial)

  __shared__ float smem[WPIX_OUT * CSO + WPIX_IN * CSI];
  float* const lds_dy = smem;
  float* const lds_in = smem + WPIX_OUT * CSO;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave % WO, wi = (wave / WO) % WI, wp = wave / (WO * WI);
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + WT_ROWS - 1) / WT_ROWS;   // last row / column of tiles may be partial
  const int q0 = blockIdx.y * NBI;   // first 16-channel block of my cin chunk
  constexpr int Cout = NBO * 16;
  const unsigned npix = (unsigned)a.B * (unsigned)H * (unsigned)W;

  // the cin chunk lies entirely in one of the two concatenated sources (chunks are <= 32 channels)
  const bool first = q0 < a.src[0].nq;
  const ConvSrc& src = first ? a.src[0] : a.src[1];
  const int ch0 = src.coff + 16 * (first ? q0 : q0 - a.src[0].nq);
  const int lgc = 31 - __builtin_clz((unsigned)src.C) + 2;                 // log2(C * 4 bytes)
  constexpr bool HS = BF16;                                               // bf16 mode: activations stored as bf16
  constexpr unsigned ESZ = HS ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t rin = wg_rsrc(src.ptr, npix * (unsigned)src.C * ESZ);
  const __amdgpu_buffer_rsrc_t rdy = wg_rsrc(a.dy, npix * (unsigned)Cout * ESZ);
  const __amdgpu_buffer_rsrc_t rdyy = wg_rsrc(DYF ? a.dy_y : a.dy, npix * (unsigned)Cout * ESZ);

  // ---- per-thread staging constants (tile independent) ----
  const int c4o = tid % QO, po0 = tid / QO;      // dy: pixel po0 + i*PPO, channels 4*c4o..
  const int c4i = tid % QI, pi0 = tid / QI;      // input halo: pixel pi0 + i*PPI
  unsigned vo_dy[NIO];
#pragma unroll
  for (int i = 0; i < NIO; ++i) {
    const int p = po0 + i * PPO;
    vo_dy[i] = (unsigned)(((p >> 4) * W + (p & 15)) * Cout * 4 + c4o * 16);
  }
  int ipy[NII], ipx[NII];
#pragma unroll
  for (int i = 0; i < NII; ++i) {
    int p = pi0 + i * PPI;
    if (p >= WPIX_IN) p = WPIX_IN - 1;
    ipy[i] = p / WPW;
    ipx[i] = p - ipy[i] * WPW;
  }
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool raw = src.scale == nullptr;
  if (!raw) { psc = ld4(src.scale + ch0 + 4 * c4i); psh = ld4(src.shift + ch0 + 4 * c4i); }

  f32x4 acc[NBO_W][NBI_W][9];
#pragma unroll
  for (int o = 0; o < NBO_W; ++o)
#pragma unroll
    for (int i = 0; i < NBI_W; ++i)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[o][i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // DYF: the thread's four coefficient quads (its channels 4*c4o.. of the layer's Cout never change)
  float4 dsc = make_float4(0.f, 0.f, 0.f, 0.f), dsh = dsc, dk1 = dsc, dk0 = dsc;
  if (DYF) {
    dsc = ld4(a.dy_coef + 4 * c4o); dsh = ld4(a.dy_coef + Cout + 4 * c4o);
    dk1 = ld4(a.dy_coef + 2 * Cout + 4 * c4o); dk0 = ld4(a.dy_coef + 3 * Cout + 4 * c4o);
  }
  float4 pdy[NIO], pin[NII], pyy[DYF ? NIO : 1];
  auto issue = [&](int tile) {
    int txi, tyi, b;
    if (lgx >= 0) { txi = tile & (tiles_x - 1); tyi = (tile >> lgx) & (tiles_y - 1); b = tile >> (lgx + lgy); }
    else { txi = tile % tiles_x; const int r = tile / tiles_x; tyi = r % tiles_y; b = r / tiles_y; }
    const int x0 = txi * 16, y0 = tyi * WT_ROWS;
    const unsigned base = (unsigned)((b * H + y0) * W + x0);
    if (x0 + 16 <= W && y0 + WT_ROWS <= H) {
#pragma unroll
      for (int i = 0; i < NIO; ++i) {
        pdy[i] = wg_aload4<HS>(rdy, vo_dy[i], base * (unsigned)(Cout * 4));
        if (DYF) pyy[i] = wg_aload4<HS>(rdyy, vo_dy[i], base * (unsigned)(Cout * 4));
      }
    } else {
      // partial tile: dy of the pixels outside the image must read as 0 (they contribute nothing to dW):
      // a per-lane offset beyond the descriptor's range makes the buffer load return 0
#pragma unroll
      for (int i = 0; i < NIO; ++i) {
        const int p = po0 + i * PPO;
        const bool in = y0 + (p >> 4) < H && x0 + (p & 15) < W;
        pdy[i] = wg_aload4<HS>(rdy, in ? vo_dy[i] : 0xFFFFFF00u, base * (unsigned)(Cout * 4));
        if (DYF) pyy[i] = wg_aload4<HS>(rdyy, in ? vo_dy[i] : 0xFFFFFF00u, base * (unsigned)(Cout * 4));
      }
    }
    const bool interior = txi > 0 && tyi > 0 && txi + 1 < tiles_x && tyi + 1 < tiles_y;
    const unsigned chb = (unsigned)(ch0 + 4 * c4i) * 4u;
    if (interior) {
      const unsigned soff = ((base - (unsigned)W - 1u) << lgc);
#pragma unroll
      for (int i = 0; i < NII; ++i) pin[i] = wg_aload4<HS>(rin, ((unsigned)(ipy[i] * W + ipx[i]) << lgc) + chb, soff);
    } else {
#pragma unroll
      for (int i = 0; i < NII; ++i) {
        const int gy = clampi(y0 - 1 + ipy[i], 0, H - 1), gx = clampi(x0 - 1 + ipx[i], 0, W - 1);
        pin[i] = wg_aload4<HS>(rin, ((unsigned)((b * H + gy) * W + gx) << lgc) + chb, 0u);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) issue(tile);
  const int i16 = lane & 15, k = lane >> 4;
  const float* const a_base = lds_dy + (4 * wp + k) * CSO + 16 * (wo * NBO_W) + i16;
  const float* const b_base = lds_in + (4 * wp + k) * CSI + 16 * (wi * NBI_W) + i16;
  const float* const a16_base = lds_dy + (wp * 16 + k) * CSO + 16 * (wo * NBO_W) + i16;       // bf16 mode: row wp, pixel k
  const float* const b16_base = lds_in + (wp * WPW + k) * CSI + 16 * (wi * NBI_W) + i16;

  while (tile < a.ntiles) {
    __syncthreads();          // previous tile's MFMA reads are done
    if (DYF) {
      // partial tiles: pixels outside the image loaded zeros, and bn_bwd4(0, 0) is not 0 -> mask them
      int txi, tyi;
      if (lgx >= 0) { txi = tile & (tiles_x - 1); tyi = (tile >> lgx) & (tiles_y - 1); }
      else { txi = tile % tiles_x; tyi = (tile / tiles_x) % tiles_y; }
      const int x0 = txi * 16, y0 = tyi * WT_ROWS;
      const bool full = x0 + 16 <= W && y0 + WT_ROWS <= H;
#pragma unroll
      for (int i = 0; i < NIO; ++i) {
        float4 v = bn_bwd4(pdy[i], pyy[i], dsc, dsh, dk1, dk0);
        if (!full) {
          const int p = po0 + i * PPO;
          if (!(y0 + (p >> 4) < H && x0 + (p & 15) < W)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        pdy[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NIO; ++i)
      *reinterpret_cast<float4*>(&lds_dy[(po0 + i * PPO) * CSO + 4 * c4o]) = pdy[i];
#pragma unroll
    for (int i = 0; i < NII; ++i) {
      const int p = pi0 + i * PPI;
      float4 v = pin[i];
      if (!raw) v = bn_relu4(v, psc, psh);
      if (i + 1 < NII || p < WPIX_IN) *reinterpret_cast<float4*>(&lds_in[p * CSI + 4 * c4i]) = v;
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) issue(next);     // in flight during the MFMA phase below

    if (BF16) {
      // whole rows per wave: rows wp, wp + WP, ...; per row 4 reads per operand -> one bf16 MFMA per (o, i, tap)
#pragma unroll
      for (int jj = 0; jj < WT_ROWS / WP; ++jj) {
        wg_s16x4 ah[NBO_W];
#pragma unroll
        for (int o = 0; o < NBO_W; ++o) {
          const float* pa = a16_base + (WP * jj * 16) * CSO + 16 * o;
          ah[o] = wg_pack4(pa[0], pa[4 * CSO], pa[8 * CSO], pa[12 * CSO]);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int tyy = t / 3, txx = t - 3 * (t / 3);
          wg_s16x4 bh[NBI_W];
#pragma unroll
          for (int i = 0; i < NBI_W; ++i) {
            const float* pb = b16_base + ((WP * jj + tyy) * WPW + txx) * CSI + 16 * i;
            bh[i] = wg_pack4(pb[0], pb[4 * CSI], pb[8 * CSI], pb[12 * CSI]);
          }
#pragma unroll
          for (int o = 0; o < NBO_W; ++o)
#pragma unroll
            for (int i = 0; i < NBI_W; ++i)
              acc[o][i][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[o], bh[i], acc[o][i][t], 0, 0, 0);
        }
      }
    } else {
    // k-steps ks = (row r, pixel quad qd) = j * WP + wp, fully unrolled: the wave-dependent part (4 * wp pixels)
      // sits in the two base pointers, everything else is an immediate ds_read offset (no address arithmetic
      // between the MFMAs, and the reads can be hoisted across k-steps).
  #pragma unroll
      for (int j = 0; j < WT_ROWS * 4 / WP; ++j) {
        const int r = WP == 4 ? j : (WP == 2 ? (j >> 1) : (j >> 2));
        const int qc = WP == 4 ? 0 : (WP == 2 ? 2 * (j & 1) : (j & 3));
        float av[NBO_W];
  #pragma unroll
        for (int o = 0; o < NBO_W; ++o) av[o] = a_base[(r * 16 + 4 * qc) * CSO + 16 * o];
  #pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int tyy = t / 3, txx = t - 3 * (t / 3);
          float bv[NBI_W];
  #pragma unroll
          for (int i = 0; i < NBI_W; ++i) bv[i] = b_base[((r + tyy) * WPW + 4 * qc + txx) * CSI + 16 * i];
  #pragma unroll
          for (int o = 0; o < NBO_W; ++o)
  #pragma unroll
            for (int i = 0; i < NBI_W; ++i)
              acc[o][i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[o], bv[i], acc[o][i][t], 0, 0, 0);
        }
      }
    }
    tile = next;
  }

  // ---- combine the WP pixel-split waves through LDS, then write the slab ----
  // slab layout (floats): [chunk][nbo][nbi][tap][lane][4]
  const size_t slab_floats = (size_t)gridDim.y * NBO * NBI * 9 * 256;
  float* slab = a.slabs + (size_t)blockIdx.x * slab_floats + (size_t)blockIdx.y * NBO * NBI * 9 * 256;
  if (WP > 1) {
    // one round per extra pixel-split wave group: it parks its tiles in LDS, group 0 adds them
    static_assert(WP == 1 || WO * WI * NT * 256 <= WPIX_OUT * CSO + WPIX_IN * CSI, "wgrad reduction scratch too small");
    float* const mine = smem + ((size_t)(wi * WO + wo) * NT) * 256;
    for (int w = 1; w < WP; ++w) {
      __syncthreads();
      if (wp == w) {
#pragma unroll
        for (int o = 0; o < NBO_W; ++o)
#pragma unroll
          for (int i = 0; i < NBI_W; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
              const f32x4 v = acc[o][i][t];
              *reinterpret_cast<float4*>(mine + ((o * NBI_W + i) * 9 + t) * 256 + lane * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
      }
      __syncthreads();
      if (wp == 0) {
#pragma unroll
        for (int o = 0; o < NBO_W; ++o)
#pragma unroll
          for (int i = 0; i < NBI_W; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
              const float4 v = *reinterpret_cast<const float4*>(mine + ((o * NBI_W + i) * 9 + t) * 256 + lane * 4);
              acc[o][i][t][0] += v.x; acc[o][i][t][1] += v.y; acc[o][i][t][2] += v.z; acc[o][i][t][3] += v.w;
            }
      }
    }
  }
  if (wp == 0) {
#pragma unroll
    for (int o = 0; o < NBO_W; ++o)
#pragma unroll
      for (int i = 0; i < NBI_W; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int nbo = wo * NBO_W + o, nbi = wi * NBI_W + i;
          const f32x4 v = acc[o][i][t];
          st4(slab + ((size_t)((nbo * NBI + nbi) * 9 + t)) * 256 + lane * 4, make_float4(v[0], v[1], v[2], v[3]));
        }
  }
}

// dW[co][ci][t] = sum over workgroups of slab[blk][chunk][nbo][nbi][t][lane][r]
//   co = 16 nbo + 4 (lane>>4) + r,  ci = 16 (chunk*NBI + nbi) + (lane & 15)
// EL consecutive slab elements x (256 / EL) slab groups per workgroup, tree-reduced through LDS in a fixed order
// and scattered once to the OIHW gradient.  Small layers use a small EL so that the (latency-bound) walk over
// the slabs is spread over >= ~500 workgroups.
template <int EL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int nblk, int cin, int cout,
                                                           int nbi_chunk, float* __restrict__ dw) {
  constexpr int GR = 256 / EL;
  __shared__ double part[GR][EL];
  const int n = 9 * cin * cout;
  const int jl = threadIdx.x % EL, grp = threadIdx.x / EL;
  const int j = blockIdx.x * EL + jl;
  double s = 0.0;
  if (j < n)
    for (int k = grp; k < nblk; k += GR) s += (double)slabs[(size_t)k * n + j];
  part[grp][jl] = s;
  __syncthreads();
  for (int st = GR / 2; st > 0; st >>= 1) {
    if (grp < st) part[grp][jl] += part[grp + st][jl];
    __syncthreads();
  }
  if (grp == 0 && j < n) {
    const int r = j & 3, lane = (j >> 2) & 63;
    const int rest = j >> 8;
    const int t = rest % 9, r2 = rest / 9;
    const int NBO = cout / 16;
    const int nbi = r2 % nbi_chunk, r3 = r2 / nbi_chunk;
    const int nbo = r3 % NBO, chunk = r3 / NBO;
    const int co = 16 * nbo + 4 * (lane >> 4) + r, ci = 16 * (chunk * nbi_chunk + nbi) + (lane & 15);
    dw[(co * cin + ci) * 9 + t] = (float)part[0][jl];
  }
}

struct ReduceTable {
  unsigned long long slab_off[16];
  int nblk[16], cin[16], cout[16], nbi_chunk[16], w_off[16], blk_start[17];
  int njobs;
};

// Batched form of the slab reduction (the slabs come from HBM here, not from a warm L2): block -> (layer, 32
// consecutive slab elements = one 128-byte line per slab); 8 lanes x float4 cover the line, 32 lane groups walk
// the slabs; float64 sums, fixed order; n = 9*cin*cout is a multiple of 32 for every layer.
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const float* __restrict__ ws, const ReduceTable tb,
                                                                   float* __restrict__ grads) {
  __shared__ double part[32][8][4];
  int l = 0;
  while (l + 1 < tb.njobs && (int)blockIdx.x >= tb.blk_start[l + 1]) ++l;
  const int cin = tb.cin[l], cout = tb.cout[l], nblk = tb.nblk[l], nbi_chunk = tb.nbi_chunk[l];
  const int n = 9 * cin * cout;
  const float* slabs = ws + tb.slab_off[l];
  const int jl = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int j4 = ((int)blockIdx.x - tb.blk_start[l]) * 32 + 4 * jl;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
  for (int k = grp; k < nblk; k += 32) {
    const float4 v = ld4(slabs + (size_t)k * n + j4);
    s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
  }
  part[grp][jl][0] = s0; part[grp][jl][1] = s1; part[grp][jl][2] = s2; part[grp][jl][3] = s3;
  __syncthreads();
  for (int st = 16; st > 0; st >>= 1) {
    if (grp < st) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[grp][jl][r] += part[grp + st][jl][r];
    }
    __syncthreads();
  }
  if (threadIdx.x < 32) {
    const int j = ((int)blockIdx.x - tb.blk_start[l]) * 32 + threadIdx.x;
    const int r = j & 3, lane = (j >> 2) & 63;
    const int rest = j >> 8;
    const int t = rest % 9, r2 = rest / 9;
    const int NBO = cout / 16;
    const int nbi = r2 % nbi_chunk, r3 = r2 / nbi_chunk;
    const int nbo = r3 % NBO, chunk = r3 / NBO;
    const int co = 16 * nbo + 4 * (lane >> 4) + r, ci = 16 * (chunk * nbi_chunk + nbi) + (lane & 15);
    grads[tb.w_off[l] + (co * cin + ci) * 9 + t] = (float)part[0][threadIdx.x >> 2][threadIdx.x & 3];
  }
}

template <int NBO, int NBI, bool BF16, bool DYF>
int launch_wgrad_t(const WgradArgs& a, int chunks, int nblk, hipStream_t s) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
  const int tx_ = (a.W + 15) / 16, ty_ = (a.H + WT_ROWS - 1) / WT_ROWS;
  const int lgx = (pow2(tx_) && pow2(ty_)) ? lg(tx_) : -1, lgy = lgx >= 0 ? lg(ty_) : -1;
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<NBO, NBI, BF16, DYF>), dim3(nblk, chunks), dim3(256), 0, s, a, lgx, lgy);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

}  // namespace

size_t wgrad_slab_floats(int cin, int cout) { return (size_t)9 * cin * cout; }

// Cin is processed in chunks of <= 32 channels (blockIdx.y): keeps LDS <= 76 KB so two workgroups fit a CU.
// A chunk never straddles the two concatenated sources (16-channel chunks when the first has an odd
// number of 16-channel blocks, e.g. ub3.convbloc.bloc.0 = cat(16, 16)).
int wgrad_nbi_chunk(const WgradArgs& a, int cin) {
  if (cin < 32) return 1;
  if (a.src[1].ptr != nullptr && (a.src[0].nq % 2)) return 1;
  return 2;
}

int launch_conv3x3_wgrad(const WgradArgs& a, int cin, int cout, int nblk, hipStream_t s) {
  if (a.H < 1 || a.W < 1 || cin % 16 || cout % 16 || nblk < 1) return SIFSR_ERR_SHAPE;
  if (a.ntiles != a.B * ((a.H + WT_ROWS - 1) / WT_ROWS) * ((a.W + 15) / 16)) return SIFSR_ERR_ARG;
  {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const size_t npix = (size_t)a.B * a.H * a.W;
    int cmax = a.src[0].C > a.src[1].C ? a.src[0].C : a.src[1].C;
    cmax = cmax > cout ? cmax : cout;
    if (npix * cmax * 4 >= ((size_t)1 << 32) - 4096) return SIFSR_ERR_SHAPE;      // 32-bit buffer offsets
    if (!pow2(a.src[0].C) || (a.src[1].ptr && !pow2(a.src[1].C))) return SIFSR_ERR_SHAPE;
  }
  const int nbi = wgrad_nbi_chunk(a, cin), chunks = (cin / 16) / nbi, nbo = cout / 16;
  const bool dyf = a.dy_y != nullptr;
  if (dyf && !a.dy_coef) return SIFSR_ERR_ARG;
#define SIFSR_WG(NBOV, NBIV)                                                                                                   \
  if (nbo == NBOV && nbi == NBIV)                                                                                              \
    return a.bf16 ? (dyf ? launch_wgrad_t<NBOV, NBIV, true, true>(a, chunks, nblk, s) : launch_wgrad_t<NBOV, NBIV, true, false>(a, chunks, nblk, s)) \
                  : (dyf ? launch_wgrad_t<NBOV, NBIV, false, true>(a, chunks, nblk, s) : launch_wgrad_t<NBOV, NBIV, false, false>(a, chunks, nblk, s));
  SIFSR_WG(1, 1) SIFSR_WG(1, 2) SIFSR_WG(2, 1) SIFSR_WG(2, 2) SIFSR_WG(4, 2)
#undef SIFSR_WG
  return SIFSR_ERR_SHAPE;
}

int launch_wgrad_reduce(const float* slabs, int nblk, int cin, int cout, int nbi_chunk, float* dw_oihw, hipStream_t s) {
  const int n = 9 * cin * cout;
  if (n <= 2304)
    hipLaunchKernelGGL((wgrad_reduce_kernel<4>), dim3((n + 3) / 4), dim3(256), 0, s, slabs, nblk, cin, cout, nbi_chunk, dw_oihw);
  else if (n <= 4608)
    hipLaunchKernelGGL((wgrad_reduce_kernel<8>), dim3((n + 7) / 8), dim3(256), 0, s, slabs, nblk, cin, cout, nbi_chunk, dw_oihw);
  else if (n <= 9216)
    hipLaunchKernelGGL((wgrad_reduce_kernel<16>), dim3((n + 15) / 16), dim3(256), 0, s, slabs, nblk, cin, cout, nbi_chunk, dw_oihw);
  else
    hipLaunchKernelGGL((wgrad_reduce_kernel<32>), dim3((n + 31) / 32), dim3(256), 0, s, slabs, nblk, cin, cout, nbi_chunk, dw_oihw);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_wgrad_reduce_batched(const float* ws, const WgradReduceJob* jobs, int njobs, float* grads, hipStream_t s) {
  if (njobs < 1 || njobs > 16) return SIFSR_ERR_ARG;
  ReduceTable tb;
  int blk = 0;
  for (int i = 0; i < njobs; ++i) {
    const int n = 9 * jobs[i].cin * jobs[i].cout;
    if (n % 32 || jobs[i].slab_off % 4) return SIFSR_ERR_SHAPE;
    tb.slab_off[i] = jobs[i].slab_off; tb.nblk[i] = jobs[i].nblk; tb.cin[i] = jobs[i].cin; tb.cout[i] = jobs[i].cout;
    tb.nbi_chunk[i] = jobs[i].nbi_chunk; tb.w_off[i] = jobs[i].w_off;
    tb.blk_start[i] = blk;
    blk += n / 32;
  }
  tb.blk_start[njobs] = blk;
  tb.njobs = njobs;
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blk), dim3(256), 0, s, ws, tb, grads);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
