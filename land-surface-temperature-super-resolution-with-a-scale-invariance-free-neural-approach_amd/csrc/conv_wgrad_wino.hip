// Weight gradient of the replicate-padded 3x3 convolution in the Winograd F(3x3, 2x2) domain, fp32 matrix cores.
//
//   dW[co][ci][i][j] = sum over 2x2 output patches P of  sum_{a,b in {0,1}} d_P[i+a][j+b][ci] * g_P[a][b][co]
// with d_P the 4x4 input window of the patch (replicate-clamped) and g_P the 2x2 patch of dL/dy -- the correlation of a
// 4x4 tile with a 2x2 kernel giving 3x3 outputs, i.e. F(3x3, 2x2):
//   dW = A^T [ sum_P (G g_P G^T) . (B^T d_P B) ] A,      16 products per patch and channel pair instead of 36
//   B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]],  G = [[1,0],[1,1],[1,-1],[0,1]],
//   A^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]          (the halves of the textbook G moved into A^T: G stays exact)
// (nn.Conv2d backward-weight, model.py:135,138,507.)
//
// GEMM view per transform-domain position xi = (u, v): M_xi (Cout x Cin) += Ug_xi^T (Cout x patches) * V_xi (patches x Cin),
// v_mfma_f32_16x16x4_f32 with the 4-deep K index = 4 neighbouring patches of a patch row:
//   A operand: lane (i = cout, k = patch)  needs ONE value per xi: Ug_xi of (patch k, channel i)
//   B operand: lane (j = cin,  k = patch)  needs ONE value per xi: V_xi  of (patch k, channel j)
// The transforms act per (patch, channel), so the lane that feeds the MFMA computes them itself, in registers, from the raw
// tiles: LDS holds the dy tile and the input halo tile as CHANNEL PLANES [channel][pixel] (plane strides = 4 mod 64 dwords),
// a lane reads its 2x2 patch (2 ds_read_b64) and its 4x4 window (8 ds_read_b64: two horizontally adjacent pixels of one
// channel are one 8-byte word) without bank conflicts, transforms (row stage as v_pk_add_f32 on the pixel pairs, column
// stage scalar: 34 vector instructions), and issues the 16 MFMAs of the k-step.  Nothing transform-domain ever goes through
// LDS (an earlier form that staged Ug and V in LDS for the MFMAs to read was 2x slower than the tap-domain kernel: 64-96 KB
// of LDS per workgroup, two barriers per 0.4 us of matrix work).  Same skeleton as conv_wgrad.hip otherwise: persistent
// workgroups over 8x16-pixel tiles, the next tile's operands prefetched into registers, BatchNorm+ReLU of the producing
// layer and the BatchNorm+ReLU backward of this layer (DYF, bn_bwd4) applied while staging, one slab per workgroup.
// Slabs hold 16 transform-domain values per weight pair; wgrad_wino_reduce sums them in float64 (fixed order) and
// wgrad_wino_finish applies A^T . A and scatters to the OIHW gradient.
#include "conv.h"

#include <stdlib.h>

namespace {

constexpr int XT_ROWS = 8;                 // tile rows (4 patch rows x 8 patch columns = 32 patches = 8 k-steps)
constexpr int XPW = 18, XPH = XT_ROWS + 2; // halo tile
constexpr int XPIX_IN = XPW * XPH;         // 180
constexpr int XPIX_OUT = 16 * XT_ROWS;     // 128
// Channel planes in LDS: channel ch = 16 b + i lives at b * BS + slot(i) * PS with slot(i) = (i >> 2) + 4 * (i & 3).
//   reads  (ds_read_b64, banks mod 64 per 32-lane half = 16 channels x 2 patches): PS = 4 (mod 64) -> 4 slot + 2 k + {0,1}: all 64 banks
//   writes (ds_write_b32, banks mod 32 per 32-lane half = channel quads x consecutive pixels, one r = ch & 3 per instruction):
//          16 b + 4 (quad & 3) + 16 r + pixel with BS = 16 (mod 32): conflict-free for 8 quads x 4 pixels, 2-way (free for a store) otherwise
constexpr int XPSO = 132, XPSI = 196;
constexpr int XBSO = 16 * XPSO + 16, XBSI = 16 * XPSI + 16;
static __device__ __forceinline__ int xslot(int i) { return (i >> 2) + 4 * (i & 3); }

typedef __attribute__((ext_vector_type(4))) unsigned u32x4x;
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t xw_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ float4 xw_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4x v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int NBO, int NBI, bool DYF>   // cout blocks, cin blocks handled by one workgroup (cin chunk = blockIdx.y)
__global__ __launch_bounds__(256) void conv3x3_wgrad_wino_kernel(const WgradArgs a, const int lgx, const int lgy, const int Cout) {
  // Cout = all output channels of the layer; this workgroup takes the NBO 16-channel blocks from block blockIdx.z * NBO on
  // (64 output channels run as two halves of 32: the 32-channel variant keeps two workgroups per CU resident, the 64-channel
  // one -- 436 registers -- only one)
  constexpr int WO = NBO >= 2 ? 2 : 1, WI = NBI >= 2 ? 2 : 1, WP = 4 / (WO * WI);
  constexpr int NBO_W = NBO / WO, NBI_W = NBI / WI;
  constexpr int QO = NBO * 4, QI = NBI * 4;                   // channel quads per pixel (dy / input chunk)
  constexpr int PPO = 256 / QO, PPI = 256 / QI;               // pixels staged per pass by the workgroup
  constexpr int NIO = XPIX_OUT / PPO;                         // prefetch float4s per thread, dy tile
  constexpr int NII = (XPIX_IN + PPI - 1) / PPI;              // ... input halo tile (last pass partial)
  constexpr int KSW = 8 / WP;                                 // k-steps per wave per tile

  __shared__ __align__(16) float smem[NBO * XBSO + NBI * XBSI];
  float* const lds_dy = smem;
  float* const lds_in = smem + NBO * XBSO;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave % WO, wi = (wave / WO) % WI, wp = wave / (WO * WI);
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + XT_ROWS - 1) / XT_ROWS;   // last row / column of tiles may be partial
  const int q0 = blockIdx.y * NBI;   // first 16-channel block of my cin chunk
  const int co0 = blockIdx.z * NBO * 16;      // my first output channel
  const unsigned npix = (unsigned)a.B * (unsigned)H * (unsigned)W;

  // A cin chunk normally lies in one of the two concatenated sources.  A 32-channel chunk that straddles them (an odd number
  // of 16-channel blocks in the first source: ub3.convbloc.bloc.0 = cat(16, 16)) is staged with its two blocks on alternate
  // waves, so that the source (a buffer resource, wave-uniform) is per wave: dy / (g, y) are then read once, not per block.
  const bool straddle = NBI == 2 && a.src[1].ptr != nullptr && q0 < a.src[0].nq && q0 + NBI > a.src[0].nq;
  const int qb = q0 + (straddle ? (wave & 1) : 0);          // the 16-channel block this wave stages from (first one if not straddling)
  const bool first = qb < a.src[0].nq;
  const ConvSrc& src = first ? a.src[0] : a.src[1];
  const int ch0 = src.coff + 16 * (first ? qb : qb - a.src[0].nq);
  const int lgc = 31 - __builtin_clz((unsigned)src.C) + 2;                 // log2(C * 4 bytes)
  const __amdgpu_buffer_rsrc_t rin = xw_rsrc(src.ptr, npix * (unsigned)src.C * 4u);
  const __amdgpu_buffer_rsrc_t rdy = xw_rsrc(a.dy, npix * (unsigned)Cout * 4u);
  const __amdgpu_buffer_rsrc_t rdyy = xw_rsrc(DYF ? a.dy_y : a.dy, npix * (unsigned)Cout * 4u);

  // ---- per-thread staging constants (tile independent) ----
  const int c4o = tid % QO, po0 = tid / QO;      // dy: pixel po0 + i*PPO, channels 4*c4o..
  // input halo: pixel pi0 + i*PPI, LDS channels 4*c4i.., source channels ch0 + 4*c4s..
  const int c4i = straddle ? (tid & 3) + 4 * (wave & 1) : tid % QI;
  const int pi0 = straddle ? ((tid >> 2) & 15) + 16 * (tid >> 7) : tid / QI;
  const int c4s = straddle ? (tid & 3) : c4i;
  unsigned vo_dy[NIO];
#pragma unroll
  for (int i = 0; i < NIO; ++i) {
    const int p = po0 + i * PPO;
    vo_dy[i] = (unsigned)(((p >> 4) * W + (p & 15)) * Cout * 4 + co0 * 4 + c4o * 16);
  }
  int ipy[NII], ipx[NII];
#pragma unroll
  for (int i = 0; i < NII; ++i) {
    int p = pi0 + i * PPI;
    if (p >= XPIX_IN) p = XPIX_IN - 1;
    ipy[i] = p / XPW;
    ipx[i] = p - ipy[i] * XPW;
  }
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool raw = src.scale == nullptr;
  if (!raw) { psc = ld4(src.scale + ch0 + 4 * c4s); psh = ld4(src.shift + ch0 + 4 * c4s); }
  float4 dsc = make_float4(0.f, 0.f, 0.f, 0.f), dsh = dsc, dk1 = dsc, dk0 = dsc;
  if (DYF) {
    const float* cf = a.dy_coef + co0 + 4 * c4o;
    dsc = ld4(cf); dsh = ld4(cf + Cout);
    dk1 = ld4(cf + 2 * Cout); dk0 = ld4(cf + 3 * Cout);
  }

  f32x4 acc[NBO_W][NBI_W][16];
#pragma unroll
  for (int o = 0; o < NBO_W; ++o)
#pragma unroll
    for (int i = 0; i < NBI_W; ++i)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[o][i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 pdy[NIO], pin[NII], pyy[DYF ? NIO : 1];
  auto issue = [&](int tile) {
    int txi, tyi, b;
    if (lgx >= 0) { txi = tile & (tiles_x - 1); tyi = (tile >> lgx) & (tiles_y - 1); b = tile >> (lgx + lgy); }
    else { txi = tile % tiles_x; const int r = tile / tiles_x; tyi = r % tiles_y; b = r / tiles_y; }
    const int x0 = txi * 16, y0 = tyi * XT_ROWS;
    const unsigned base = (unsigned)((b * H + y0) * W + x0);
    const bool full = x0 + 16 <= W && y0 + XT_ROWS <= H;
#pragma unroll
    for (int i = 0; i < NIO; ++i) {
      const int p = po0 + i * PPO;
      // partial tile: dy of the pixels outside the image must read as 0 (they contribute nothing to dW)
      const bool in = full || (y0 + (p >> 4) < H && x0 + (p & 15) < W);
      pdy[i] = xw_bload4(rdy, in ? vo_dy[i] : 0xFFFFFF00u, base * (unsigned)(Cout * 4));
      if (DYF) pyy[i] = xw_bload4(rdyy, in ? vo_dy[i] : 0xFFFFFF00u, base * (unsigned)(Cout * 4));
    }
    const bool interior = txi > 0 && tyi > 0 && txi + 1 < tiles_x && tyi + 1 < tiles_y;
    const unsigned chb = (unsigned)(ch0 + 4 * c4s) * 4u;
    if (interior) {
      const unsigned soff = ((base - (unsigned)W - 1u) << lgc);
#pragma unroll
      for (int i = 0; i < NII; ++i) pin[i] = xw_bload4(rin, ((unsigned)(ipy[i] * W + ipx[i]) << lgc) + chb, soff);
    } else {
#pragma unroll
      for (int i = 0; i < NII; ++i) {
        const int gy = clampi(y0 - 1 + ipy[i], 0, H - 1), gx = clampi(x0 - 1 + ipx[i], 0, W - 1);
        pin[i] = xw_bload4(rin, ((unsigned)((b * H + gy) * W + gx) << lgc) + chb, 0u);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) issue(tile);
  const int i16 = lane & 15, k = lane >> 4;
  // my channel planes; my patch of k-step h: patch row h >> 1, patch column 4 * (h & 1) + k
  const float* const pa = lds_dy + (wo * NBO_W) * XBSO + xslot(i16) * XPSO + 2 * k;
  const float* const pb = lds_in + (wi * NBI_W) * XBSI + xslot(i16) * XPSI + 2 * k;

  while (tile < a.ntiles) {
    __syncthreads();          // previous tile's reads are done
    {
      int txi, tyi;
      if (lgx >= 0) { txi = tile & (tiles_x - 1); tyi = (tile >> lgx) & (tiles_y - 1); }
      else { txi = tile % tiles_x; tyi = (tile / tiles_x) % tiles_y; }
      const int x0 = txi * 16, y0 = tyi * XT_ROWS;
      const bool full = x0 + 16 <= W && y0 + XT_ROWS <= H;
#pragma unroll
      for (int i = 0; i < NIO; ++i) {
        const int p = po0 + i * PPO;
        float4 v = pdy[i];
        if (DYF) {
          v = bn_bwd4(v, pyy[i], dsc, dsh, dk1, dk0);
          if (!full && !(y0 + (p >> 4) < H && x0 + (p & 15) < W)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float* d = lds_dy + (c4o >> 2) * XBSO + (c4o & 3) * XPSO + p;      // channel 4 c4o + r -> slot (c4o & 3) + 4 r
        d[0] = v.x; d[4 * XPSO] = v.y; d[8 * XPSO] = v.z; d[12 * XPSO] = v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < NII; ++i) {
      const int p = pi0 + i * PPI;
      float4 v = pin[i];
      if (!raw) v = bn_relu4(v, psc, psh);
      if (i + 1 < NII || p < XPIX_IN) {
        float* d = lds_in + (c4i >> 2) * XBSI + (c4i & 3) * XPSI + p;
        d[0] = v.x; d[4 * XPSI] = v.y; d[8 * XPSI] = v.z; d[12 * XPSI] = v.w;
      }
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) issue(next);     // in flight during the MFMA phase below

    SIFSR_DIAG_SKIP_MATRIX_WORK(a.B < 0)   // (diag.h: nothing in the shipped build)
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      const int h = wp + WP * j;                 // k-step: patch row h >> 1, patch columns 4 * (h & 1) + (0..3)
      const int pr = h >> 1, pcb = 4 * (h & 1);
      // ---- A side: Ug = G g G^T of my (patch, cout) for each of my cout blocks
      float ug[NBO_W][16];
#pragma unroll
      for (int o = 0; o < NBO_W; ++o) {
        const float* q = pa + o * XBSO + (2 * pr) * 16 + 2 * pcb;
        const f32x2 g0 = *reinterpret_cast<const f32x2*>(q), g1 = *reinterpret_cast<const f32x2*>(q + 16);
        const f32x2 r1 = pk_add(g0, g1), r2 = pk_sub(g0, g1);                 // rows of G g: [g0, g0 + g1, g0 - g1, g1]
        const f32x2 rows[4] = {g0, r1, r2, g1};
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                         // columns: (x, y) -> [x, x + y, x - y, y]
          ug[o][4 * u + 0] = rows[u][0]; ug[o][4 * u + 1] = rows[u][0] + rows[u][1];
          ug[o][4 * u + 2] = rows[u][0] - rows[u][1]; ug[o][4 * u + 3] = rows[u][1];
        }
      }
      // ---- B side: V = B^T d B of my (patch, cin) for each of my cin blocks
      float vv[NBI_W][16];
#pragma unroll
      for (int n = 0; n < NBI_W; ++n) {
        const float* q = pb + n * XBSI + (2 * pr) * XPW + 2 * pcb;
        f32x2 dl[4], dh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { dl[u] = *reinterpret_cast<const f32x2*>(q + u * XPW); dh[u] = *reinterpret_cast<const f32x2*>(q + u * XPW + 2); }
        // rows: [d0 - d2, d1 + d2, d2 - d1, d3 - d1] on both pixel pairs
        const f32x2 rl[4] = {pk_sub(dl[0], dl[2]), pk_add(dl[1], dl[2]), pk_sub(dl[2], dl[1]), pk_sub(dl[3], dl[1])};
        const f32x2 rh[4] = {pk_sub(dh[0], dh[2]), pk_add(dh[1], dh[2]), pk_sub(dh[2], dh[1]), pk_sub(dh[3], dh[1])};
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                         // columns: same pattern on (x0, x1 | x2, x3)
          vv[n][4 * u + 0] = rl[u][0] - rh[u][0]; vv[n][4 * u + 1] = rl[u][1] + rh[u][0];
          vv[n][4 * u + 2] = rh[u][0] - rl[u][1]; vv[n][4 * u + 3] = rh[u][1] - rl[u][1];
        }
      }
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int o = 0; o < NBO_W; ++o)
#pragma unroll
          for (int n = 0; n < NBI_W; ++n)
            acc[o][n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ug[o][t], vv[n][t], acc[o][n][t], 0, 0, 0);
    }
    tile = next;
  }

  // ---- combine the WP pixel-split waves through LDS, then write the slab ----
  // slab layout (floats): [chunk][nbo][nbi][tap][lane][4] -- the output transform A^T M A is applied here, per lane, so a slab holds
  // 9 values per weight pair (the tap-domain kernel's layout) instead of the 16 of the transform domain: 44 % fewer slab bytes
  // written, read back by the reduction, and one reduction kernel for both families
  constexpr int NT = NBO_W * NBI_W * 16;
  const size_t slab_floats = (size_t)gridDim.y * gridDim.z * NBO * NBI * 9 * 256;   // = 9 * Cin * Cout
  float* slab = a.slabs + (size_t)blockIdx.x * slab_floats + (size_t)(blockIdx.y * gridDim.z + blockIdx.z) * NBO * NBI * 9 * 256;
  if (WP > 1) {
    // the accumulators of one extra wave group do not all fit the tile buffers at once: park / add them 4 xi at a time
    static_assert(WO * WI * NBO_W * NBI_W * 4 * 256 <= NBO * XBSO + NBI * XBSI, "wgrad reduction scratch too small");
    float* const mine = smem + ((size_t)(wi * WO + wo) * NBO_W * NBI_W * 4) * 256;
    for (int w = 1; w < WP; ++w) {
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        __syncthreads();
        if (wp == w) {
#pragma unroll
          for (int o = 0; o < NBO_W; ++o)
#pragma unroll
            for (int i = 0; i < NBI_W; ++i)
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const f32x4 v = acc[o][i][4 * tq + t];
                *reinterpret_cast<float4*>(mine + ((o * NBI_W + i) * 4 + t) * 256 + lane * 4) = make_float4(v[0], v[1], v[2], v[3]);
              }
        }
        __syncthreads();
        if (wp == 0) {
#pragma unroll
          for (int o = 0; o < NBO_W; ++o)
#pragma unroll
            for (int i = 0; i < NBI_W; ++i)
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const float4 v = *reinterpret_cast<const float4*>(mine + ((o * NBI_W + i) * 4 + t) * 256 + lane * 4);
                f32x4& r = acc[o][i][4 * tq + t];
                r[0] += v.x; r[1] += v.y; r[2] += v.z; r[3] += v.w;
              }
        }
      }
    }
  }
  (void)NT;
  if (wp == 0) {
#pragma unroll
    for (int o = 0; o < NBO_W; ++o)
#pragma unroll
      for (int i = 0; i < NBI_W; ++i) {
        const int nbo = wo * NBO_W + o, nbi = wi * NBI_W + i;
        f32x4 tap[9];
        wino_wgrad_taps([&](int xi) { return acc[o][i][xi]; }, tap);
#pragma unroll
        for (int t = 0; t < 9; ++t)
          st4(slab + ((size_t)((nbo * NBI + nbi) * 9 + t)) * 256 + lane * 4, make_float4(tap[t][0], tap[t][1], tap[t][2], tap[t][3]));
      }
  }
}

template <int NBO, int NBI, bool DYF>
int launch_xw(const WgradArgs& a, int chunks, int halves, int nblk, hipStream_t s) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
  const int tx_ = (a.W + 15) / 16, ty_ = (a.H + XT_ROWS - 1) / XT_ROWS;
  const int lgx = (pow2(tx_) && pow2(ty_)) ? lg(tx_) : -1, lgy = lgx >= 0 ? lg(ty_) : -1;
  hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<NBO, NBI, DYF>), dim3(nblk, chunks, halves), dim3(256), 0, s, a, lgx, lgy, NBO * 16 * halves);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

}  // namespace

// 16-channel blocks per cin chunk (blockIdx.y): 32-channel chunks from 32 input channels on, also across the two sources
int wgrad_wino_nbi_chunk(const WgradArgs& a, int cin) {
  (void)a;
  return cin < 32 ? 1 : 2;
}

bool conv3x3_wgrad_use_wino(const WgradArgs& a, int cin, int cout) {
  static const int off = getenv("SIFSR_NO_WINO_WGRAD") ? atoi(getenv("SIFSR_NO_WINO_WGRAD")) : 0;   // 1: tap-domain weight gradients (A/B)
  const int nbo = cout / 16, nbi = wgrad_wino_nbi_chunk(a, cin);
  const bool shape = (nbo == 1 || nbo == 2 || nbo == 4) && (nbi == 1 || nbi == 2) && !(nbo == 4 && nbi == 1);
  return !off && a.bf16 == 0 && a.H % 2 == 0 && a.W % 2 == 0 && shape;
}

int launch_conv3x3_wgrad_wino(const WgradArgs& a, int cin, int cout, int nblk, hipStream_t s) {
  if (a.H < 2 || a.W < 2 || cin % 16 || cout % 16 || nblk < 1) return SIFSR_ERR_SHAPE;
  if (a.ntiles != a.B * ((a.H + XT_ROWS - 1) / XT_ROWS) * ((a.W + 15) / 16)) return SIFSR_ERR_ARG;
  if (!conv3x3_wgrad_use_wino(a, cin, cout)) return SIFSR_ERR_SHAPE;
  {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const size_t npix = (size_t)a.B * a.H * a.W;
    int cmax = a.src[0].C > a.src[1].C ? a.src[0].C : a.src[1].C;
    cmax = cmax > cout ? cmax : cout;
    if (npix * cmax * 4 >= ((size_t)1 << 32) - 4096) return SIFSR_ERR_SHAPE;      // 32-bit buffer offsets
    if (!pow2(a.src[0].C) || (a.src[1].ptr && !pow2(a.src[1].C))) return SIFSR_ERR_SHAPE;
  }
  const bool dyf = a.dy_y != nullptr;
  if (dyf && !a.dy_coef) return SIFSR_ERR_ARG;
  const int nbi = wgrad_wino_nbi_chunk(a, cin), chunks = (cin / 16) / nbi, nbo = cout / 16;
  static const int split64 = getenv("SIFSR_DBG_WGRAD_WINO_SPLIT64") ? atoi(getenv("SIFSR_DBG_WGRAD_WINO_SPLIT64")) : 1;   // A/B knob
#define SIFSR_XW(NBOV, NBIV, HV)                                                                                     \
  if (nbo == NBOV * HV && nbi == NBIV)                                                                              \
    return dyf ? launch_xw<NBOV, NBIV, true>(a, chunks, HV, nblk, s) : launch_xw<NBOV, NBIV, false>(a, chunks, HV, nblk, s);
  SIFSR_XW(1, 1, 1) SIFSR_XW(1, 2, 1) SIFSR_XW(2, 1, 1) SIFSR_XW(2, 2, 1)
  if (split64) { SIFSR_XW(2, 2, 2) }
  SIFSR_XW(4, 2, 1)
#undef SIFSR_XW
  return SIFSR_ERR_SHAPE;
}

