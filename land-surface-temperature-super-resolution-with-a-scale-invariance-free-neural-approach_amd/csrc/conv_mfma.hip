// 3x3 convolution (stride 1) as an implicit GEMM on the fp32 matrix cores of gfx950.
//
// Replaces what the reference dispatches to ATen/MKLDNN/cuDNN for nn.Conv2d(k=3, padding=1,
// padding_mode='replicate', bias=False) -- model.py:135,138,507 -- and its input-gradient.
//
// Mapping (one workgroup = 512 threads = 8 waves, four PRODUCERS and four CONSUMERS; PERSISTENT: it walks 16x16-pixel output
// tiles in an XCD-aware order; see the role comment above the kernel):
//   * the (16+2)x(16+2) input halo tile of one 16-channel block is staged in LDS by the producers as
//     [channel-quad k][pixel][4 channels], with the producing layer's BatchNorm+ReLU folded into the staging
//     (relu(x*scale+shift)) -- or, in the input-gradient pass, the BatchNorm+ReLU BACKWARD of the layer itself (DYF: dL/dy is
//     formed from (g, y) here and never stored);
//   * tap-domain consumers: v_mfma_f32_16x16x4_f32, A = weights (16 cout x 4 cin), B = activations (4 cin x 16 pixels of
//     one image row), D = 16 cout x 16 pixels.  Lane (i = lane&15, k = lane>>4) reads ONE ds_read_b128 = channels 4k..4k+3 of
//     pixel i and feeds 4 MFMAs (k-step j uses channel 4k+j on both operands -- the contraction order inside a 16-channel
//     block is a free permutation); this read is bank-conflict free;
//   * Winograd consumers (WINO; what the fp32 model runs for <= 64 output channels): the same lane map with the 16
//     transform-domain positions of a 2x2 output patch in the place of the 9 taps of a pixel -- 4/9 of the MFMAs, the input
//     and output transforms as packed adds on the operand / accumulator registers;
//   * weights come pre-packed in fragment order (pack_weights_kernel / pack_wino_kernel): one coalesced
//     buffer_load_dwordx4 per (cout block, cin block, tap or xi) per wave, L2 resident;
//   * D rows are 4 consecutive cout per lane -> one 16-byte NHWC store per lane per pixel;
//   * optional epilogue: per-channel (sum, sumsq) of the tile for training-mode BatchNorm statistics (or, for 16-channel
//     input gradients, the BatchNorm-backward sums of the layer below), reduced with wave shuffles + LDS and written per
//     workgroup (deterministic 2-stage reduction), residual addend, split destinations.
#include "conv.h"

#include <stdlib.h>

namespace {

constexpr int PW = 18;        // plane width  (16 + 2 halo)
constexpr int PLANE = 336;    // 18*18 = 324 pixels, padded to a multiple of 16 (bank rule)
// Winograd variant (WINO, see the consumer): halo rows of WPITCH 16-byte slots, even columns in slots [0, 9), odd ones
// in [WHALF, WHALF + 9), so that the 4 consecutive pixels of a patch row are slots {x/2, WHALF + x/2, x/2 + 1, WHALF + x/2 + 1}
// and the 16 lanes of a read (8 patches of one patch row | 8 of the next, 2 halo rows = 2 * WPITCH slots = 640 B = 128 (mod
// 256) further) cover all 64 banks exactly once.
// The plane (channel quad) stride is = 4 or 12 (mod 16) slots, so that the four quads of a pixel -- written by four neighbouring
// producer lanes -- start 64 B apart in the 256-byte bank row and a 16-lane ds_write_b128 (4 pixels x 4 quads) is conflict-free.
constexpr int WPITCH = 20, WHALF = 10, WPLANE = 364;   // >= 18 * WPITCH = 360 slots per channel quad; 364 = 12 (mod 16)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// Raw buffer access (T8 of the CDNA guide): 128-bit descriptor in SGPRs + 32-bit per-lane byte offset +
// scalar byte offset.  Per-lane offsets at or beyond num_records read as 0 (used for dgrad's zero padding);
// the scalar offset is not part of that check.
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
constexpr unsigned OOB = 0xFFFFFF00u;   // per-lane offset beyond any tensor: buffer loads return 0

// ---- bf16 operand mode (BASELINE.json config 5: "bf16 mixed precision, MFMA-bf16 conv tiles") ----
// Activations are STORED as bf16 (round 3; common.h "activation storage"), weights as fp32 master copies with bf16 fragment packs;
// the producers widen the loaded values, apply the folded BatchNorm + ReLU in fp32, round the result to bf16 (RNE,
// v_cvt_pk_bf16_f32) into LDS, and ONE v_mfma_f32_16x16x32_bf16 (two taps at a time, see the consumer loop) contracts the 16
// channels that take four v_mfma_f32_16x16x4_f32 in the fp32 mode -- same lane map (lane (i, kq) holds channels 4kq..4kq+3),
// fp32 accumulation; the epilogue rounds the outputs to bf16 for the store (statistics are taken of the rounded values: they
// describe what the consumers will read).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
static __device__ __forceinline__ uint2 bload2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  return make_uint2(v.x, v.y);
}
static __device__ __forceinline__ void bstore2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, uint2 v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  u32x2 u; u.x = v.x; u.y = v.y;
  __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)voff, (int)soff, 0);
}
// Activation access through a buffer resource: byte offsets are those of the fp32 layout; with half storage (HS) the resource
// was made with half the size and the offsets are halved (OOB >> 1 = 0x7FFFFF80 stays beyond any bf16 tensor: fp32-sized
// tensors are limited to 2^32 - 4096 bytes by the launchers).
template <bool HS> static __device__ __forceinline__ float4 aload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (HS) return unpack_bf16x4(bload2(r, voff >> 1, soff >> 1));
  else return bload4(r, voff, soff);
}
template <bool HS> static __device__ __forceinline__ void astore4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v) {
  if constexpr (HS) bstore2(r, voff >> 1, soff >> 1, pack_bf16x4(v));
  else bstore4(r, voff, soff, v);
}

// 512 threads = 8 waves, specialised by role (one of each per SIMD):
//   waves 0-3  CONSUMERS: ds_read_b128 + MFMA over the staged buffer, then the tile epilogue (stores, statistics);
//   waves 4-7  PRODUCERS: global loads of the next (tile, channel block) halo, BatchNorm+ReLU transform, LDS writes.
// One s_barrier per work item hands a filled buffer to the consumers and a drained one back to the producers
// (double-buffered LDS), so the producers' VALU / VMEM / LDS-write instructions issue in the shadow of the
// consumers' 32-cycle MFMAs on the same SIMD instead of in a separate phase of the same wave.
// MODE 0: fp32 MFMA.  1: bf16 operands (config 5).  (A third mode -- fp32 emulated exactly on the bf16 matrix cores by a
// three-term operand split -- existed in rounds 1-2 and was removed in round 3: the Winograd consumers reach the same goal,
// fewer matrix-pipe cycles per fp32 result, without operand splitting)
// DYF (dgrad only): the operand is dL/dy of the layer, formed while staging from g = dL/d relu(bn(y)) and y
// (bn_bwd4) -- the BatchNorm-backward elementwise pass and its tensor round trip do not exist (ConvArgs::bw_*).
// WINO (fp32 only): the consumers contract in the Winograd F(2x2, 3x3) domain -- per 2x2 output patch and channel
// 16 products instead of 36 -- see the consumer branch.  One workgroup per CU (the transforms want registers).
// WINO forward with NB == 1 (16 output channels: the layers with the least matrix work per byte): the transform-domain weights of
// the layer (<= 2 channel blocks, 16 KB each) live in LDS instead of 64 registers per lane and the output transform is not
// software-pipelined, which brings the kernel under 128 registers -> TWO workgroups per CU, i.e. a second consumer wave
// per SIMD to issue while the first one waits (measured at one workgroup per CU: matrix pipe busy 32 %, SIMD idle half the time).
template <int NB, bool ZERO_PAD, int MODE, bool DYF, bool WINO>
__global__ __launch_bounds__(512, (WINO ? (NB == 1 && !ZERO_PAD ? 4 : 2) : (NB <= 2 ? 4 : 2))) void conv3x3_mfma_kernel(const ConvArgs a, const int ntiles, const int lgx,
                                                           const int lgy) {
  static_assert(!DYF || ZERO_PAD, "the fused BatchNorm backward belongs to the input-gradient pass");
  static_assert(!WINO || (MODE == 0 && NB <= 4), "the Winograd consumer is fp32, up to 64 output channels");
  static_assert(MODE == 0 || MODE == 1, "fp32 or bf16 operands");
  constexpr bool BF16 = MODE != 0;
  constexpr bool HS = BF16;                                  // bf16 mode: activations are stored as bf16 (common.h)
  constexpr unsigned ESZ = HS ? 2u : 4u;                     // bytes per stored activation element
  constexpr int CBW = NB >= 4 ? NB / 4 : 1;                  // cout blocks per consumer wave
  constexpr int CST = 4;                                     // ... block nb0 + CST * c
  constexpr int NG = NB == 1 ? 4 : (NB == 2 ? 8 : 16);       // tile rows per consumer wave

  __shared__ float4 lds[2][WINO ? 4 * WPLANE : 4 * PLANE];
  __shared__ float red[4][CBW][16][2];
  // (forward only: the input-gradient variants carry the fused BatchNorm work of two layers and measured slower this way)
  constexpr bool WLDS = WINO && NB == 1 && !ZERO_PAD;
  __shared__ float4 wlds[WLDS ? 2 * 16 * 64 : 1];   // WLDS: [channel block q < 2][xi][lane] fragment-ordered weights

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  // priorities: 2 for every wave of this kernel (above a co-running weight-gradient kernel at 0, see SIFSR_CHAIN_PRIO),
  // 3 inside the consumers' MFMA loop (above this kernel's own producers, as before)
  __builtin_amdgcn_s_setprio(2);
  const bool producer = wave8 >= 4;
  const int wave = wave8 & 3;
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;   // the last tile row / column may be partial
  const int NQ = a.NQ;
  const unsigned npix = (unsigned)a.B * (unsigned)H * (unsigned)W;

  // ---- persistent tile walk (identical in both roles).  Workgroups b, b+8, ... are observed to share an XCD
  // (private L2), so each residue class gets one contiguous range of tiles, walked with stride G/8 so that
  // concurrently running workgroups touch neighbouring tiles (speed only; any placement is correct).
  const int G = gridDim.x;
  const bool xcd_map = (G % 8 == 0) && (ntiles % 8 == 0);
  const int t_lo = xcd_map ? (blockIdx.x % 8) * (ntiles / 8) : 0;
  const int t_hi = xcd_map ? t_lo + ntiles / 8 : ntiles;
  const int t_step = xcd_map ? G / 8 : G;
  int t = t_lo + (xcd_map ? blockIdx.x / 8 : blockIdx.x);
  auto tile_pos = [&](int tt, int& tb, int& txi, int& tyi) {
    if (lgx >= 0) { tyi = (tt >> lgx) & (tiles_y - 1); tb = tt >> (lgx + lgy); txi = (tt + tyi + tb) & (tiles_x - 1); }   // column rotated per row: no workgroup is pinned to a border column
    else { txi = tt % tiles_x; const int r = tt / tiles_x; tyi = r % tiles_y; tb = r / tiles_y; }
  };

  if (WLDS) {   // (a.NQ <= 2, checked by the launcher)
    for (int i = tid; i < a.NQ * 16 * 64; i += 512) wlds[i] = ld4(a.wpack + 4 * (size_t)i);
    __syncthreads();
  }

  if (producer) {
    // =========================================== PRODUCER ===========================================
    // 16-output-channel Winograd layers (the HBM-heaviest ones): the PRODUCERS take the higher priority.  Without the MFMA work
    // the same data movement runs at 5.5 TB/s (100 us for 16->16 @256^2, SIFSR_DIAG_NOMFMA build, diag.h); with it 150 us although the
    // matrix work alone is ~75 us -- the consumers' back-to-back MFMAs keep the producers' few staging instructions and the next
    // item's loads from issuing on time.  Producers at 3, consumers at 2: -4 % forward, -6 % input gradient, +0.9 % on the step.
    // (the bf16-operand kernels, HBM-bound throughout, take the same arrangement: +0.8 % on the bf16 step)
    if ((WINO && NB == 1) || MODE == 1) __builtin_amdgcn_s_setprio(3);
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.src[0].ptr, npix * a.src[0].C * ESZ);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.src[1].ptr ? a.src[1].ptr : a.src[0].ptr, npix * (a.src[1].ptr ? a.src[1].C : a.src[0].C) * ESZ);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(DYF ? a.bw_y : a.src[0].ptr, npix * a.src[0].C * ESZ);
    const __amdgpu_buffer_rsrc_t rbd = make_rsrc(DYF && a.bw_border ? a.bw_border : const_cast<float*>(a.src[0].ptr), npix * a.src[0].C * ESZ);
    // staging map: thread -> (channel quad cg, 6 halo pixels).  Interior tiles: the 6 pixel offsets relative to
    // the halo origin are tile-independent constants; the tile position is a SCALAR offset.
    const int ptid = tid - 256;
    // fp32: (channel quad fastest) -- measured equal to the conflict-free order, kept.  bf16: ds_write_b64 is
    // serviced in groups of 16 contiguous lanes on a 32-dword bank row -> 16 consecutive pixels of ONE quad.
    const int cg = BF16 ? (ptid >> 4) & 3 : ptid & 3;
    const int pslot = BF16 ? ((ptid & 15) | ((ptid >> 6) << 4)) : ptid >> 2;
    int spy[6], spx[6], prel[6], lslot[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      int p = pslot + 64 * it;
      if (p >= PW * PW) p = PW * PW - 1;           // lanes past the plane re-load the last pixel and do not store it
      spy[it] = p / PW;
      spx[it] = p - spy[it] * PW;
      prel[it] = spy[it] * W + spx[it];
      lslot[it] = WINO ? cg * WPLANE + spy[it] * WPITCH + (spx[it] & 1) * WHALF + (spx[it] >> 1) : cg * PLANE + pslot + 64 * it;
    }
    int pixv[6];                  // per-lane pixel index of the tile being fetched (-1: outside, dgrad only)
    int pix_base = 0;             // scalar pixel offset added to pixv (interior tiles)
    auto set_tile = [&](int tt) {
      int tb, txi, tyi;
      tile_pos(tt, tb, txi, tyi);
      const int tx0 = txi * 16, ty0 = tyi * 16;
      const bool interior = txi > 0 && tyi > 0 && txi + 1 < tiles_x && tyi + 1 < tiles_y;
      if (interior) {
        pix_base = (tb * H + ty0 - 1) * W + tx0 - 1;
#pragma unroll
        for (int it = 0; it < 6; ++it) pixv[it] = prel[it];
      } else {
        pix_base = 0;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
          const int gy = ty0 - 1 + spy[it], gx = tx0 - 1 + spx[it];
          const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
          const int pc = (tb * H + clampi(gy, 0, H - 1)) * W + clampi(gx, 0, W - 1);
          pixv[it] = (ZERO_PAD && !inside) ? -1 : pc;
        }
      }
    };
    // One item in flight: the loads of item j+1 are issued right after item j went to LDS and land while the
    // consumers work on item j.  (Two register sets / two items in flight measured 4 % slower: the consumers,
    // not the load latency, are the critical path.)
    // DEPTH register sets = DEPTH items in flight: the loads of item j + DEPTH are issued when item j goes to LDS.  One is
    // enough everywhere: with two or three sets the one-workgroup-per-CU Winograd variants (whose items are half as long)
    // measured the same to 0.2 % -- what they wait for is not load latency.
    constexpr int DEPTH = 1;
    float4 stgS[DEPTH][6], scS[DEPTH], shS[DEPTH];
    float4 styS[DYF ? DEPTH : 1][DYF ? 6 : 1], k1S[DEPTH], k0S[DEPTH];   // DYF: y of the same slots, two more coefficient quads
    bool rawS[DEPTH];
    int it_t = t, it_q = 0;       // next item to fetch; past the end the last item is fetched again (never used)
    int tS[DEPTH], qS[DEPTH];     // DYF: (tile, channel block) of the item held in set k (for the masks of write_stage)
    auto issue_loads = [&](const int k) {
      float4 (&stg)[6] = stgS[k];
      float4 (&styA)[DYF ? 6 : 1] = styS[DYF ? k : 0];
      float4 &psc = scS[k], &psh = shS[k], &k1A = k1S[k], &k0A = k0S[k];
      bool& praw = rawS[k];
      const int q = it_q;
      tS[k] = it_t; qS[k] = it_q;
      const bool first = q < a.src[0].nq;
      const int C = first ? a.src[0].C : a.src[1].C;
      const int lgc = 31 - __builtin_clz((unsigned)C) + 2;                       // log2(C * 4 bytes)
      const int ch = (first ? a.src[0].coff + 16 * q : a.src[1].coff + 16 * (q - a.src[0].nq));
      const unsigned soff = ((unsigned)pix_base << lgc) + (unsigned)ch * 4u;
      const __amdgpu_buffer_rsrc_t r = first ? rs0 : rs1;
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const unsigned voff = (ZERO_PAD && pixv[it] < 0) ? OOB : (((unsigned)pixv[it] << lgc) + (unsigned)cg * 16u);
        stg[it] = aload4<HS>(r, voff, soff);
        if (DYF) styA[it] = aload4<HS>(rsy, voff, soff);
      }
      const float* scp = first ? a.src[0].scale : a.src[1].scale;
      const float* shp = first ? a.src[0].shift : a.src[1].shift;
      praw = scp == nullptr;
      const int chs = praw ? 0 : ch + 4 * cg;
      if (DYF) {   // [sc | sh | k1 | k0], C floats each
        psc = ld4(a.bw_coef + ch + 4 * cg); psh = ld4(a.bw_coef + C + ch + 4 * cg);
        k1A = ld4(a.bw_coef + 2 * C + ch + 4 * cg); k0A = ld4(a.bw_coef + 3 * C + ch + 4 * cg);
      } else {
        psc = ld4((praw ? a.wpack : scp) + chs); psh = ld4((praw ? a.wpack : shp) + chs);   // unconditional: fixed load count per set
      }
      // advance to the next item
      if (it_q + 1 < NQ) ++it_q;
      else if (it_t + t_step < t_hi) { it_t += t_step; it_q = 0; set_tile(it_t); }
    };
    auto write_stage = [&](float4* Lb, const int k) {
      const float4 (&stg)[6] = stgS[k];
      const float4 (&styA)[DYF ? 6 : 1] = styS[DYF ? k : 0];
      const float4 psc = scS[k], psh = shS[k], k1A = k1S[k], k0A = k0S[k];
      const bool praw = rawS[k];
      const int tA = tS[k], qA = qS[k];
      // DYF: tile of THIS item (the tile state above already belongs to the next one).  Halo pixels outside the image
      // are zero padding of dL/dy (bn_bwd4 of the zeros they loaded is not 0), and the image-border pixels of the tile's
      // own 16x16 core go to bw_border for the border-fold kernel.  Interior tiles need neither.
      int wb = 0, wy0 = 0, wx0 = 0;
      bool winterior = true;
      if (DYF) {
        int txi, tyi;
        tile_pos(tA, wb, txi, tyi);
        wy0 = tyi * 16 - 1; wx0 = txi * 16 - 1;
        winterior = txi > 0 && tyi > 0 && txi + 1 < tiles_x && tyi + 1 < tiles_y;
      }
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        float4 v = stg[it];
        if (DYF) {
          v = bn_bwd4(v, styA[it], psc, psh, k1A, k0A);
          if (!winterior) {
            const int gy = wy0 + spy[it], gx = wx0 + spx[it];
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool core = spy[it] >= 1 && spy[it] <= 16 && spx[it] >= 1 && spx[it] <= 16;
            const bool edge = gy == 0 || gy == H - 1 || gx == 0 || gx == W - 1;
            if (a.bw_border != nullptr && inside && core && edge && (it < 5 || pslot < PW * PW - 320))
              astore4<HS>(rbd, (unsigned)((wb * H + gy) * W + gx) * (unsigned)a.src[0].C * 4u + (unsigned)(16 * qA + 4 * cg) * 4u, 0u, v);
          }
        } else if (!praw) v = bn_relu4(v, psc, psh);
        if (it < 5 || pslot < PW * PW - 320) {
          if (BF16) reinterpret_cast<uint2*>(Lb)[cg * PLANE + pslot + 64 * it] = pack_bf16x4(v);
          else Lb[lslot[it]] = v;
        }
      }
    };

    const int n_items = ((t_hi - t + t_step - 1) / t_step) * NQ;
    set_tile(t);
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) issue_loads(k);
    for (int j = 0; j < n_items; j += DEPTH) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        if (j + k < n_items) {                // uniform over the workgroup's producers: one barrier per item, as the consumers
          write_stage(lds[(j + k) & 1], k);
          issue_loads(k);                     // item j + k + DEPTH (past the end: the last one again, never used)
          __syncthreads();                    // item j + k staged; the consumers have drained the other buffer
        }
      }
    }
    if (a.stat_partials != nullptr) __syncthreads();   // matches the consumers' barrier in the statistics tail
    return;
  }

  // ============================================= CONSUMER =============================================
  const int nb0 = NB == 1 ? 0 : (NB == 2 ? (wave & 1) : wave);
  const int g0 = NB == 1 ? wave * 4 : (NB == 2 ? (wave >> 1) * 8 : 0);
  const int kq = lane >> 4, px = lane & 15;
  const __amdgpu_buffer_rsrc_t rd0 = make_rsrc(a.dst[0].ptr, npix * a.dst[0].C * ESZ);
  const __amdgpu_buffer_rsrc_t rd1 = make_rsrc(a.dst[1].ptr, npix * a.dst[1].C * ESZ);
  const __amdgpu_buffer_rsrc_t rad = make_rsrc(a.addend ? a.addend : a.dst[0].ptr, npix * (a.addend ? a.addC : a.dst[0].C) * ESZ);
  const bool bn_stats = NB == 1 && ZERO_PAD && a.bn_y != nullptr;     // BatchNorm-backward sums of the previous layer
  const __amdgpu_buffer_rsrc_t rby = make_rsrc(bn_stats ? a.bn_y : a.dst[0].ptr, npix * 16u * ESZ);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wpack, (unsigned)(NB * 16) * (unsigned)(NQ * 16) * (BF16 ? 18u : 36u));

  float4 wf[BF16 ? 1 : CBW][BF16 ? 1 : 9];
  uint2 wh[BF16 ? CBW : 1][BF16 ? 9 : 1];     // bf16 mode: 4 bf16 per lane per (cout block, tap)
  auto load_weights = [&](int q) {
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
      const int nb = nb0 + CST * c;
      if (BF16) {
        const unsigned soff = (unsigned)((nb * NQ + q) * 9) * 512u;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          wh[c][tp] = bload2(rw, (unsigned)lane * 8u, soff + tp * 512u);
        }
      } else {
        const unsigned soff = (unsigned)((nb * NQ + q) * 9) * 1024u;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) wf[c][tp] = bload4(rw, (unsigned)lane * 16u, soff + tp * 1024u);
      }
    }
  };

  float s1[CBW][4], s2[CBW][4];
#pragma unroll
  for (int c = 0; c < CBW; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[c][r] = s2[c][r] = 0.f;


  if constexpr (WINO) {
    // =================================== Winograd F(2x2, 3x3) consumer ===================================
    // Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output patch: the channel contraction runs on the 16 transform-domain
    // positions xi = (a, b) instead of the 9 taps -- 16 products per 4 outputs instead of 36 (2.25x fewer MFMAs).
    //   * B operand: lane (i, kq) owns patch i of its wave's 16-patch group (8 patch columns x 2 patch rows = 16 x 4 output
    //     pixels) and channels 4kq..4kq+3: 16 ds_read_b128 of the staged halo (the 4x4 input window of the patch), the input
    //     transform V = B^T d B on float4 registers (rows, then columns: 32 packed-pair adds each), and V[xi] feeds 4 MFMAs
    //     (k-step j contracts channel 4kq+j) -- the same lane map as the direct kernel, with xi in place of the tap;
    //   * A operand: the transform-domain weights U[xi] = G g G^T, packed per (cout block, cin block, xi) in fragment order by
    //     pack_weights_kernel; resident in registers, the next item's fetched xi-row by xi-row behind the MFMAs that used it;
    //   * the output transform A^T M A is linear, so it is applied per channel block (item) to the xi-row's four
    //     accumulators as soon as they complete and summed into the patch's 2x2 outputs Y: only 4 + 4 accumulators
    //     (+ the software-pipelined second set) are live instead of 16 per patch.
    // Waves split the tile as in the direct kernel (cout blocks for NB >= 2, 4-row groups otherwise); a wave that shares a
    // patch group with another cout block's wave repeats the input transform (VALU in exchange for no LDS round trip of V).
    constexpr int NGRP = NG / 4;                       // 16-patch groups (4 tile rows each) per wave
    static_assert(CBW == 1, "one cout block per wave");
    const int pxp = lane & 7, pyl = (lane >> 3) & 1;
    const int lbase = kq * WPLANE + (g0 + 2 * pyl) * WPITCH + pxp;
    const __amdgpu_buffer_rsrc_t rww = make_rsrc(a.wpack, (unsigned)(NB * 16) * (unsigned)(NQ * 16) * 64u);
    float4 wq[WLDS ? 1 : 16];
    auto wsoff = [&](int q_, int xi) { return (unsigned)(((nb0 * NQ + q_) * 16 + xi)) * 1024u; };
    if (!WLDS) {
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) wq[xi] = bload4(rww, (unsigned)lane * 16u, wsoff(0, xi));
    }

    f32x4 Y[NGRP][2][2];
#pragma unroll
    for (int g = 0; g < NGRP; ++g)
#pragma unroll
      for (int o = 0; o < 4; ++o) Y[g][o >> 1][o & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto lo2 = [](f32x4 v) { return (f32x2){v[0], v[1]}; };
    auto hi2 = [](f32x4 v) { return (f32x2){v[2], v[3]}; };
    auto acc2 = [](f32x4& y, f32x2 l, f32x2 h, bool minus) {
      const f32x2 yl = minus ? pk_sub((f32x2){y[0], y[1]}, l) : pk_add((f32x2){y[0], y[1]}, l);
      const f32x2 yh = minus ? pk_sub((f32x2){y[2], y[3]}, h) : pk_add((f32x2){y[2], y[3]}, h);
      y = (f32x4){yl[0], yl[1], yh[0], yh[1]};
    };
    int buf = 0, q = 0;

    while (true) {
      __syncthreads();   // lds[buf] holds this item; the producers may now refill the other buffer
      const bool last_q = q + 1 == NQ;
      const int t_next = t + t_step;
      const bool more = !last_q || t_next < t_hi;
      const int qn = last_q ? 0 : q + 1;

      // fused BatchNorm-backward sums of the previous layer (NB == 1 dgrad): its y at this lane's 2x2 output pixels,
      // requested before the MFMA work of the tile's last channel block (see the direct kernel)
      // (the two-workgroups-per-CU variant has no registers to hold it across the MFMA work: it loads y in the epilogue,
      // where the other workgroup covers the wait)
      float4 yq[NB == 1 ? 4 : 1];
      float4 bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsh = bsc;
      if (NB == 1 && !WLDS && bn_stats && last_q) {
        int cb_, txi_, tyi_;
        tile_pos(t, cb_, txi_, tyi_);
        const int y0_ = tyi_ * 16 + g0 + 2 * pyl, x0_ = txi_ * 16 + 2 * pxp;
        bsc = ld4(a.bn_scale + 4 * kq); bsh = ld4(a.bn_shift + 4 * kq);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int yy_ = y0_ + (o >> 1), xx_ = x0_ + (o & 1);
          yq[o] = bload4(rby, (yy_ < H && xx_ < W) ? (unsigned)((cb_ * H + yy_) * W + xx_) * 64u + (unsigned)kq * 16u : OOB, 0u);
        }
      }

      const float4* L = lds[buf];
      __builtin_amdgcn_s_setprio(NB == 1 ? 2 : 3);   // NB == 1: below this kernel's producers (see there)
      SIFSR_DIAG_SKIP_MATRIX_WORK(a.B < 0)   // (diag.h: nothing in the shipped build)
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        // ---- the patch's 4x4 input window -> V = B^T d B (in place).  Every float4 is handled as its two aligned
        // register pairs so that each add / subtract is ONE v_pk_add_f32 (left to itself the compiler pairs components of
        // different float4s and spends more v_mov than adds assembling the operands)
        f32x2 dl[4][4], dh[4][4];
        const float4* Lg = L + lbase + g * 4 * WPITCH;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 v0 = Lg[r * WPITCH], v1 = Lg[r * WPITCH + WHALF], v2 = Lg[r * WPITCH + 1], v3 = Lg[r * WPITCH + WHALF + 1];
          dl[r][0] = (f32x2){v0.x, v0.y}; dh[r][0] = (f32x2){v0.z, v0.w};
          dl[r][1] = (f32x2){v1.x, v1.y}; dh[r][1] = (f32x2){v1.z, v1.w};
          dl[r][2] = (f32x2){v2.x, v2.y}; dh[r][2] = (f32x2){v2.z, v2.w};
          dl[r][3] = (f32x2){v3.x, v3.y}; dh[r][3] = (f32x2){v3.z, v3.w};
        }
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
        // ---- per xi-row a: M[b] = U[a][b] * V[a][b] over the 16 channels (4 MFMAs each, 4 independent chains), then
        // t = M A (2 columns) and Y += A^T rows: Y[0] += t for a = 0, 1, 2;  Y[1] += t, -t, -t for a = 1, 2, 3
        // (software-pipelined by one xi-row: the output transform of row a - 1 is issued behind the MFMAs of row a, so that it
        // does not sit waiting for results that are still in the matrix pipe)
        f32x4 M[WLDS ? 1 : 2][4];
        auto out_row = [&](const int ar, const f32x4 (&Mr)[4]) {
          // t = M A: t0 = M0 + M1 + M2, t1 = M1 - (M2 + M3).  The FIRST instruction that reads freshly written MFMA results
          // must be one the compiler knows as a vector-ALU instruction -- it pads the matrix-pipe -> VALU read hazard with
          // s_nop itself, which it does not do for inline assembly -- hence the plain vector sums (they compile to
          // v_pk_add_f32); the subtraction then reads M1, whose chain finished before the M2 / M3 ones the sums waited for.
          const f32x4 t0 = Mr[0] + Mr[1] + Mr[2], u = Mr[2] + Mr[3];
          const f32x2 t0l = lo2(t0), t0h = hi2(t0);
          const f32x2 t1l = pk_sub(lo2(Mr[1]), lo2(u)), t1h = pk_sub(hi2(Mr[1]), hi2(u));
          if (ar <= 2) { acc2(Y[g][0][0], t0l, t0h, false); acc2(Y[g][0][1], t1l, t1h, false); }
          if (ar == 1) { acc2(Y[g][1][0], t0l, t0h, false); acc2(Y[g][1][1], t1l, t1h, false); }
          if (ar >= 2) { acc2(Y[g][1][0], t0l, t0h, true); acc2(Y[g][1][1], t1l, t1h, true); }
        };
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) {
          f32x4 (&Mc)[4] = M[WLDS ? 0 : (ar & 1)];
          float4 wr[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) wr[b] = WLDS ? wlds[(q * 16 + 4 * ar + b) * 64 + lane] : wq[WLDS ? 0 : 4 * ar + b];
#pragma unroll
          for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].x, dl[ar][b][0], zero4, 0, 0, 0);
#pragma unroll
          for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].y, dl[ar][b][1], Mc[b], 0, 0, 0);
#pragma unroll
          for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].z, dh[ar][b][0], Mc[b], 0, 0, 0);
#pragma unroll
          for (int b = 0; b < 4; ++b) Mc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b].w, dh[ar][b][1], Mc[b], 0, 0, 0);
          if (!WLDS && g == NGRP - 1) {
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch behind this xi-row's MFMAs (see the direct kernel)
#pragma unroll
            for (int b = 0; b < 4; ++b) wq[WLDS ? 0 : 4 * ar + b] = bload4(rww, (unsigned)lane * 16u, wsoff(qn, 4 * ar + b));
          }
          if (WLDS) out_row(ar, Mc);
          else if (ar > 0) out_row(ar - 1, M[WLDS ? 0 : ((ar - 1) & 1)]);
        }
        if (!WLDS) out_row(3, M[WLDS ? 0 : 1]);
      }
      __builtin_amdgcn_s_setprio(2);
      buf ^= 1;
      if (!last_q) { ++q; continue; }

      // ---- tile epilogue: the lane's 2x2 output pixels x 4 channels per group
      int cb, txi, tyi;
      tile_pos(t, cb, txi, tyi);
      const bool do_stats = a.stat_partials != nullptr && !bn_stats;
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
          if (NB == 1 && bn_stats) {
            if (WLDS) {
              yq[o] = bload4(rby, ok ? pixo * 64u + (unsigned)kq * 16u : OOB, 0u);
              bsc = ld4(a.bn_scale + 4 * kq); bsh = ld4(a.bn_shift + 4 * kq);
            }
            const float yy4[4] = {yq[o].x, yq[o].y, yq[o].z, yq[o].w};
            const float scv[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, shv[4] = {bsh.x, bsh.y, bsh.z, bsh.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float dz = (ok && fmaf(yy4[r], scv[r], shv[r]) > 0.f) ? v[r] : 0.f;
              s1[0][r] += dz; s2[0][r] = fmaf(dz, yy4[r], s2[0][r]);
            }
          } else if (do_stats) {
            if (!ok) v = zero4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[0][r] += v[r]; s2[0][r] = fmaf(v[r], v[r], s2[0][r]); }
          }
        }
      }
      if (!more) break;
      t = t_next; q = 0;
    }
  } else {
  load_weights(0);   // NQ == 1 (single 16-channel block): the weights stay in registers for every tile
  int buf = 0, q = 0;

  f32x4 acc[CBW][NG];
#pragma unroll
  for (int c = 0; c < CBW; ++c)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[c][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  while (true) {
    __syncthreads();   // lds[buf] holds this item; the producers may now refill the other buffer

    const bool last_q = q + 1 == NQ;
    const int t_next = t + t_step;
    const bool more = !last_q || t_next < t_hi;

    // fused BatchNorm-backward sums (NB == 1 dgrad): y of the previous layer for this tile's 4 rows per wave is
    // requested BEFORE the MFMA loop of the tile's last channel block and consumed in the epilogue, so the epilogue
    // does not park the wave on a fresh HBM round trip (SQ_WAIT_ANY 0.44 -> see DESIGN.md section 10)
    float4 yq[NB == 1 ? 4 : 1];
    float4 bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsh = bsc;
    if (NB == 1 && bn_stats && last_q) {
      int cb_, txi_, tyi_;
      tile_pos(t, cb_, txi_, tyi_);
      const unsigned tp_ = (unsigned)((cb_ * H + tyi_ * 16 + g0) * W + txi_ * 16);
      const bool colok_ = txi_ * 16 + px < W;
      const int rows_ok_ = H - (tyi_ * 16 + g0);
      bsc = ld4(a.bn_scale + 4 * kq); bsh = ld4(a.bn_shift + 4 * kq);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        yq[g] = aload4<HS>(rby, (colok_ && g < rows_ok_) ? (unsigned)px * 64u + (unsigned)kq * 16u : OOB, (tp_ + (unsigned)(g * W)) * 64u);
    }
    const float4* L = lds[buf];
    // The weights of the NEXT item (channel block q+1, or block 0 of the next tile; NQ == 1: the same ones again) are
    // fetched tap by tap as soon as the current tap's MFMAs are issued (tap-outer loop order), so their L2 latency hides
    // under the other taps and no second register set is needed.  Unconditional (no branch inside the MFMA stream);
    // the sched_barrier keeps the compiler from hoisting the loads to the top, which doubles the live weights and
    // spills (measured: +1.3 % on the step, and the NB = 8 variant no longer spills).
    const int qn = last_q ? 0 : q + 1;
    __builtin_amdgcn_s_setprio(MODE == 1 ? 2 : 3);   // bf16 operands: below this kernel's producers
    if (BF16) {
      // v_mfma_f32_16x16x32_bf16 (16 cycles for K = 32; the K = 16 form of gfx90a takes the same 16): one MFMA contracts the
      // 16 channels of TWO taps.  Lane (i, kq) holds k = 8*kq .. 8*kq+7 = channels 4kq..4kq+3 of the first tap, then of the
      // second -- exactly the two 8-byte words the per-tap layout already has, for the weights and for the staged operand.
      // Pairs (0,1) (2,3) (4,5) (6,7) (8,-): the last one runs with zero weights in its upper half.
      const uint2* L16 = reinterpret_cast<const uint2*>(L);
      auto cat = [](uint2 lo, uint2 hi) { return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)); };
      const uint2 z2 = make_uint2(0u, 0u);
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        const int t0 = 2 * p, t1 = 2 * p + 1 < 9 ? 2 * p + 1 : 8;
        const bool two = 2 * p + 1 < 9;
        const int ty0 = t0 / 3, tx0 = t0 - 3 * (t0 / 3), ty1 = t1 / 3, tx1 = t1 - 3 * (t1 / 3);
#pragma unroll
        for (int gb = 0; gb < NG / 4; ++gb) {
          bf16x8 bh[4];
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            const int r = g0 + gb * 4 + gi;
            const int o0 = kq * PLANE + (r + ty0) * PW + tx0 + px, o1 = kq * PLANE + (r + ty1) * PW + tx1 + px;
            bh[gi] = cat(L16[o0], L16[o1]);
          }
#pragma unroll
          for (int c = 0; c < CBW; ++c) {
            const bf16x8 w = cat(wh[c][t0], two ? wh[c][t1] : z2);
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
              acc[c][gb * 4 + gi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, bh[gi], acc[c][gb * 4 + gi], 0, 0, 0);
          }
        }
        {
          __builtin_amdgcn_sched_barrier(0);   // keep the prefetch behind this pair's MFMAs: hoisted, it doubles the live weights
#pragma unroll
          for (int c = 0; c < CBW; ++c) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              if (k == 1 && !two) continue;
              const int tp = k == 0 ? t0 : t1;
              const unsigned so = (unsigned)(((nb0 + CST * c) * NQ + qn) * 9 + tp) * 512u;
              wh[c][tp] = bload2(rw, (unsigned)lane * 8u, so);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int ty = tp / 3, tx = tp - 3 * (tp / 3);
#pragma unroll
        for (int gb = 0; gb < NG / 4; ++gb) {
          float4 bf[4];
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            const int r = g0 + gb * 4 + gi;
            bf[gi] = L[kq * PLANE + (r + ty) * PW + tx + px];
          }
#pragma unroll
          for (int c = 0; c < CBW; ++c) {
            const float4 w = wf[c][tp];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
              acc[c][gb * 4 + gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, bf[gi].x, acc[c][gb * 4 + gi], 0, 0, 0);
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
              acc[c][gb * 4 + gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, bf[gi].y, acc[c][gb * 4 + gi], 0, 0, 0);
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
              acc[c][gb * 4 + gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, bf[gi].z, acc[c][gb * 4 + gi], 0, 0, 0);
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
              acc[c][gb * 4 + gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, bf[gi].w, acc[c][gb * 4 + gi], 0, 0, 0);
          }
        }
        {
          __builtin_amdgcn_sched_barrier(0);   // keep the prefetch behind this tap's MFMAs: hoisted, it doubles the live weights
#pragma unroll
          for (int c = 0; c < CBW; ++c)
            wf[c][tp] = bload4(rw, (unsigned)lane * 16u, (unsigned)(((nb0 + CST * c) * NQ + qn) * 9 + tp) * 1024u);
        }
      }
    }
    __builtin_amdgcn_s_setprio(2);
    buf ^= 1;
    if (!last_q) { ++q; continue; }

    // ---- tile epilogue: NHWC stores (+ residual addend), running per-channel statistics.
    // scalar offset = tile row start; per-lane offset = (pixel column, 4-channel group) constant.
    int cb, txi, tyi;
    tile_pos(t, cb, txi, tyi);
    const unsigned tile_pix = (unsigned)((cb * H + tyi * 16 + g0) * W + txi * 16);
    // partial tiles (H or W not a multiple of 16 at this level): columns past W get an out-of-range offset (the
    // store is dropped, the addend reads 0) and stay out of the statistics; rows past H are skipped.
    const bool colok = txi * 16 + px < W;
    const int rows_ok = H - (tyi * 16 + g0);
    const bool do_stats = a.stat_partials != nullptr && !bn_stats;
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
      const int nb = nb0 + CST * c;
      const bool d0 = nb < a.dst_split;
      const int dC = d0 ? a.dst[0].C : a.dst[1].C;
      const __amdgpu_buffer_rsrc_t rd = d0 ? rd0 : rd1;
      const unsigned chb = (unsigned)((d0 ? a.dst[0].coff + 16 * nb : a.dst[1].coff + 16 * (nb - a.dst_split)) + 4 * kq) * 4u;
      const unsigned vo = colok ? (unsigned)px * (unsigned)dC * 4u + chb : OOB;
      const unsigned va = colok ? (unsigned)px * (unsigned)a.addC * 4u + (unsigned)(16 * nb + 4 * kq) * 4u : OOB;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const unsigned rowpix = tile_pix + (unsigned)(g * W);
        f32x4 v = acc[c][g];
        acc[c][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g >= rows_ok) continue;
        if (a.addend != nullptr) {
          const float4 ad = aload4<HS>(rad, va, rowpix * (unsigned)a.addC * 4u);
          v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
        }
        if (HS) {   // the statistics below are those of the STORED (bf16-rounded) values
          const float4 vr = round_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
          v = (f32x4){vr.x, vr.y, vr.z, vr.w};
        }
        astore4<HS>(rd, vo, rowpix * (unsigned)dC * 4u, make_float4(v[0], v[1], v[2], v[3]));
        if (NB == 1 && bn_stats) {    // dz = g * [y*scale + shift > 0]; sum dz, sum dz*y (masked lanes read y = 0, v ignored)
          const float yy[4] = {yq[g].x, yq[g].y, yq[g].z, yq[g].w};
          const float scv[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, shv[4] = {bsh.x, bsh.y, bsh.z, bsh.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dz = (colok && fmaf(yy[r], scv[r], shv[r]) > 0.f) ? v[r] : 0.f;
            s1[c][r] += dz; s2[c][r] = fmaf(dz, yy[r], s2[c][r]);
          }
        } else if (do_stats) {        // uniform: forward in training mode only (VALU work stalls the matrix pipe)
          if (!colok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[c][r] += v[r]; s2[c][r] = fmaf(v[r], v[r], s2[c][r]); }
        }
      }
    }
    if (!more) break;
    t = t_next; q = 0;
  }

  }   // direct (non-Winograd) consumer

  // ---- per-workgroup BatchNorm partials (sum, sumsq) over all tiles this workgroup produced ----
  if (a.stat_partials != nullptr) {
#pragma unroll
    for (int c = 0; c < CBW; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = s1[c][r], v = s2[c][r];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
        if (px == 0) { red[wave][c][4 * kq + r][0] = u; red[wave][c][4 * kq + r][1] = v; }
      }
    __syncthreads();
    if (tid < NB * 16) {
      const int nb = tid >> 4, cc = tid & 15;
      float u = 0.f, v = 0.f;
      if (NB == 1) {
        for (int w = 0; w < 4; ++w) { u += red[w][0][cc][0]; v += red[w][0][cc][1]; }
      } else if (NB == 2) {
        for (int w = nb; w < 4; w += 2) { u += red[w][0][cc][0]; v += red[w][0][cc][1]; }
      } else {
        u = red[nb & 3][nb >> 2][cc][0]; v = red[nb & 3][nb >> 2][cc][1];
      }
      float* o = a.stat_partials + ((size_t)blockIdx.x * (NB * 16) + tid) * 2;
      o[0] = u; o[1] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing: OIHW parameters -> MFMA fragment order, for forward and for dgrad
//   fwd  : wf[nb][q][tap][lane][j] = W[co = 16nb + (lane&15)][ci = 16q + 4(lane>>4) + j][tap]
//   dgrad: wd[nb][q][tap][lane][j] = W[co = 16q + 4(lane>>4) + j][ci = 16nb + (lane&15)][8 - tap]
//          (transposed and spatially flipped: dx[p] = sum_t W_t^T dy[p - t])
//   bf16 : the same two packs rounded to bf16 (config 5), [fwd16 | dgrad16] = 2 * 9*cin*cout bf16
// The dgrad buffer of a layer holds [wd | fwd16 dgrad16 | unused] = 4 * 9*cin*cout floats (the size the C ABI documents).
// ---------------------------------------------------------------------------------------------
struct PackTable { int w_off[16], cin[16], cout[16], p_off[16]; };

static __device__ __forceinline__ void pack_weights_body(const float* __restrict__ params, float* __restrict__ wfwd,
                                                         float* __restrict__ wdg, const PackTable& tb, int bx, int gdx) {
  const int l = blockIdx.y;
  const int cin = tb.cin[l], cout = tb.cout[l];
  const int n = 9 * cin * cout;
  const float* W = params + tb.w_off[l];
  for (int e = bx * blockDim.x + threadIdx.x; e < n; e += gdx * blockDim.x) {
    const int j = e & 3, lane = (e >> 2) & 63;
    const int rest = e >> 8;
    const int tap = rest % 9, r2 = rest / 9;
    {
      const int NQ = cin / 16;
      const int q = r2 % NQ, nb = r2 / NQ;
      const int co = 16 * nb + (lane & 15), ci = 16 * q + 4 * (lane >> 4) + j;
      wfwd[tb.p_off[l] + e] = W[(co * cin + ci) * 9 + tap];
    }
    {
      const int NQ = cout / 16;
      const int q = r2 % NQ, nb = r2 / NQ;
      const int co = 16 * q + 4 * (lane >> 4) + j, ci = 16 * nb + (lane & 15);
      wdg[4 * tb.p_off[l] + e] = W[(co * cin + ci) * 9 + (8 - tap)];
    }
    {
      // bf16 fragment packs (same element order), 2 bytes each, behind the layer's fp32 dgrad pack (4n floats in all):
      //   [fp32 dgrad n | fwd bf16 n/2 | dgrad bf16 n/2 | 2n unused (held the mid / lo terms of the removed split-bf16 mode)]
      __bf16* h = reinterpret_cast<__bf16*>(wdg + 4 * tb.p_off[l] + n);
      const float wv[2] = {wfwd[tb.p_off[l] + e], wdg[4 * tb.p_off[l] + e]};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        h[k * n + e] = (__bf16)wv[k];
      }
    }
  }
}

// Winograd-domain weights U = G g G^T (4x4 per (cout, cin) pair; G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]) in
// fragment order, for the forward (g = W[co][ci]) and for the input gradient (g = W[co][ci] transposed and flipped):
//   ww[nb][q][xi][lane][j], xi = 4a + b, same (lane, j) -> (row, k) map as the tap packs above; 16*cin*cout floats each.
__global__ void pack_weights_kernel(const float* __restrict__ params, float* __restrict__ wfwd,
                                    float* __restrict__ wdg, const PackTable tb) {
  pack_weights_body(params, wfwd, wdg, tb, blockIdx.x, gridDim.x);
}

static __device__ __forceinline__ void pack_wino_body(const float* __restrict__ params, float* __restrict__ wwf,
                                                      float* __restrict__ wwd, const PackTable& tb, int bx, int gdx) {
  const int l = blockIdx.y;
  const int cin = tb.cin[l], cout = tb.cout[l];
  const int n = 16 * cin * cout;
  const float* W = params + tb.w_off[l];
  const size_t off = (size_t)tb.p_off[l] / 9 * 16;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  for (int e = bx * blockDim.x + threadIdx.x; e < n; e += gdx * blockDim.x) {
    const int j = e & 3, lane = (e >> 2) & 63;
    const int rest = e >> 8;
    const int xi = rest & 15, r2 = rest >> 4;
    const int xa = xi >> 2, xb = xi & 3;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int NQ = (k == 0 ? cin : cout) / 16;
      const int q = r2 % NQ, nb = r2 / NQ;
      const int co = k == 0 ? 16 * nb + (lane & 15) : 16 * q + 4 * (lane >> 4) + j;
      const int ci = k == 0 ? 16 * q + 4 * (lane >> 4) + j : 16 * nb + (lane & 15);
      const float* g = W + (size_t)(co * cin + ci) * 9;
      float u = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float row = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) row += g[k == 0 ? 3 * r + c : 8 - (3 * r + c)] * G[xb][c];
        u += G[xa][r] * row;
      }
      (k == 0 ? wwf : wwd)[off + e] = u;
    }
  }
}
__global__ void pack_wino_kernel(const float* __restrict__ params, float* __restrict__ wwf, float* __restrict__ wwd,
                                 const PackTable tb) {
  pack_wino_body(params, wwf, wwd, tb, blockIdx.x, gridDim.x);
}
// all four packs of all layers (+ the BatchNorm num_batches_tracked counters of a training-mode forward) in ONE launch: the three
// launches this replaces were 20 us at the head of every forward for < 2 us of work.  blockIdx.x < gx: tap packs; the rest: Winograd.
__global__ void pack_all_kernel(const float* __restrict__ params, float* __restrict__ wfwd, float* __restrict__ wdg,
                                float* __restrict__ wwf, float* __restrict__ wwd, const PackTable tb, int gx, long long* nbt, int nbt_n) {
  if ((int)blockIdx.x < gx) pack_weights_body(params, wfwd, wdg, tb, blockIdx.x, gx);
  else pack_wino_body(params, wwf, wwd, tb, (int)blockIdx.x - gx, (int)gridDim.x - gx);
  if (nbt != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < nbt_n) nbt[threadIdx.x] += 1;
}

// ---------------------------------------------------------------------------------------------
// dgrad border fold.  Forward reads x[clamp(p + t)] (replicate padding); its adjoint sends the
// gradient of every out-of-range read back to the clamped pixel:
//   g[q] = sum_t W_t^T * sum_{p : clamp(p+t) = q} dy[p]
// The MFMA kernel (zero_pad) produced the p = q - t terms; this kernel adds the rest, which exist only
// for q on the image border.  Per axis the extra source is p = q itself, for (q = 0, t = -1) and
// (q = N-1, t = +1); the 2-D set is the product of the per-axis sets minus the (base, base) pair.
// One thread per (border pixel, 4 input channels); weights tap-major so the ci quad is one float4.
// ---------------------------------------------------------------------------------------------
// One WAVE per (image, side, 16-pixel border segment, 16 input channels):
//   D[ci][pixel] += sum over (tap, source) pairs of  Wd_tap[ci][co] * dy[source(pixel)][co]
// with v_mfma_f32_16x16x4_f32: A = fragment-ordered dgrad weights (same pack as the main kernel; tap index
// flipped: forward tap t <-> pack index 8 - t), B = dy rows gathered straight from global (one float4 per lane
// = 4 k-steps), no LDS, no barrier.  Pairs per side (forward tap (ty,tx), source p; q = border pixel):
//   top    (qy=0)   : ((-1,tx), (0, qx-tx)) tx=-1..1 ;  bottom (qy=H-1): ((+1,tx), (H-1, qx-tx))
//   left   (qx=0,   1<=qy<=H-2): ((ty,-1), (qy-ty, 0)) ; right (qx=W-1): ((ty,+1), (qy-ty, W-1))
//   corners belong to the top/bottom passes and add, on the corner lane only,
//     top-left: ((0,-1),(0,0)), ((-1,-1),(1,0)+(0,0))      top-right: ((0,+1),(0,W-1)), ((-1,+1),(1,W-1)+(0,W-1))
//     bottom-left: ((0,-1),(H-1,0)), ((+1,-1),(H-2,0)+(H-1,0))   bottom-right: mirrored.
__global__ __launch_bounds__(256, 8) void dgrad_border_kernel(const float* __restrict__ dy, int Cout,
                                                           const float* __restrict__ wd, int Cin, float* g0, int C0,
                                                           int split_ch, float* g1, int C1, int B, int H, int W,
                                                           int bf16, const float* __restrict__ bn_y,
                                                           const float* __restrict__ bn_scale,
                                                           const float* __restrict__ bn_shift,
                                                           float* __restrict__ bn_partials, int masked) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int NBI = Cin / 16, NQ = Cout / 16;
  const int seg_tb = (W + 15) / 16, seg_lr = (H - 2 + 15) / 16;
  const int per_img = (2 * seg_tb + 2 * seg_lr) * NBI;
  if (wave_g >= B * per_img) return;                       // wave-uniform
  SIFSR_CHAIN_PRIO();
  const int b = wave_g / per_img;
  int r = wave_g - b * per_img;
  const int nb = r % NBI; r /= NBI;
  int side, seg;                                           // 0 top, 1 bottom, 2 left, 3 right
  if (r < 2 * seg_tb) { side = r / seg_tb; seg = r - side * seg_tb; }
  else { r -= 2 * seg_tb; side = 2 + r / seg_lr; seg = r % seg_lr; }
  const int px = lane & 15, kq = lane >> 4;
  const float* wbase = wd + (size_t)nb * NQ * 9 * 256 + lane * 4;       // wd[nb][q][tap'][lane][4]
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // w: fragment-ordered weights of one (channel block, tap); bv: 4 k-steps of dy for this lane's pixel
  auto mac = [&](float4 w, float4 bv, bool round_b) {
    if (bf16) { w = round_bf16x4(w); if (round_b) bv = round_bf16x4(bv); }   // same products as the bf16 MFMA of the main kernel
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, bv.w, acc, 0, 0, 0);
  };
  auto add4 = [](float4 u, float4 v) { return make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w); };

  int qy, qx;
  bool valid = true;
  if (side < 2) { qy = side == 0 ? 0 : H - 1; qx = seg * 16 + px; valid = qx < W; }   // partial last segment when W % 16 != 0
  else { qx = side == 2 ? 0 : W - 1; qy = 1 + seg * 16 + px; valid = qy <= H - 2; }
  // The kernel runs beside a weight-gradient kernel that keeps the HBM queues full, so every dependent round trip
  // costs several microseconds.  Everything is therefore requested up front and branch-free: the read-modify-write
  // operand here, and per channel block all dy rows (buffer loads; masked lanes take the out-of-range offset and
  // read 0) and weights before the first MFMA.
  const size_t pix_o = valid ? (size_t)(b * H + qy) * W + qx : 0;
  const int ci_o = 16 * nb + 4 * kq;
  // bf16 != 0: the compute mode whose activations are STORED as bf16 (common.h): dy, g0 / g1 and bn_y are bf16 tensors
  float* const dbase = ci_o < split_ch ? g0 : g1;
  const size_t de = ci_o < split_ch ? pix_o * C0 + ci_o : pix_o * C1 + (ci_o - split_ch);   // element index
  const float4 g_old = valid ? (bf16 ? ldA4<true>(dbase, de) : ldA4<false>(dbase, de)) : z4;

  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (unsigned)B * (unsigned)H * (unsigned)W * (unsigned)Cout * (bf16 ? 2u : 4u));
  auto dyload = [&](unsigned voff, unsigned soff) { return bf16 ? aload4<true>(rdy, voff, soff) : aload4<false>(rdy, voff, soff); };
  auto poff = [&](int y, int x, bool ok) {
    return ok ? (unsigned)((b * H + y) * W + x) * (unsigned)Cout * 4u + (unsigned)kq * 16u : OOB;
  };
  // ro / wt: the three taps every border pixel has.  Corner lanes (top/bottom passes, first/last segment) add
  //   e_left, n_left, e_right, n_right with e = the corner pixel, n = its vertical neighbour, under the taps
  //   (0,-1), (ty,-1) / (0,+1), (ty,+1); the (ty,+-1) tap multiplies e + n.  Their offsets are formed where they are
  //   used (registers, see below).
  unsigned ro[3];
  int wt[3];
  bool cl = false, cr = false;
  const int ty_tb = side == 0 ? -1 : 1;                    // top / bottom passes
  const int yin = side == 0 ? 1 : H - 2;                   // the row next to the border row
  if (side < 2) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tx = i - 1, sx = qx - tx;
      ro[i] = poff(qy, sx < 0 ? 0 : (sx >= W ? W - 1 : sx), valid && sx >= 0 && sx < W);
      wt[i] = 8 - ((ty_tb + 1) * 3 + (tx + 1));
    }
    cl = seg == 0; cr = seg == seg_tb - 1;
  } else {
    const int tx = side == 2 ? -1 : 1;
    const int yc = valid ? qy : H - 2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ty = i - 1;
      ro[i] = poff(yc - ty, qx, valid);
      wt[i] = 8 - ((ty + 1) * 3 + (tx + 1));
    }
  }
  const bool corner = cl || cr;                            // wave-uniform
  // Register budget <= 64: beside a resident weight-gradient kernel (96 registers x 4 waves per SIMD) only 128
  // registers per SIMD are free, so the number of border waves in flight -- and with it this kernel's duration on the
  // serial chain -- is set by its register count (measured beside wgrad<1,1>: 155 VGPRs 70-120 us, 99 VGPRs 25-30 us).
  // Hence one channel block at a time, and the corner sources reuse the registers of the three main taps.
  for (int q = 0; q < NQ; ++q) {
    float4 rv[3], wv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      rv[i] = dyload(ro[i], (unsigned)q * 64u);
      wv[i] = ld4(wbase + ((size_t)q * 9 + wt[i]) * 256);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) mac(wv[i], rv[i], true);
    if (corner) {
#pragma unroll
      for (int sd = 0; sd < 2; ++sd) {                       // left corner, then right corner
        const int xc = sd == 0 ? 0 : W - 1;
        const bool on = (sd == 0 ? cl : cr) && valid && qx == xc;
        rv[0] = dyload(poff(qy, xc, on), (unsigned)q * 64u);
        rv[1] = dyload(poff(yin, xc, on), (unsigned)q * 64u);
        wv[0] = ld4(wbase + ((size_t)q * 9 + (8 - (1 * 3 + 2 * sd))) * 256);
        wv[1] = ld4(wbase + ((size_t)q * 9 + (8 - ((ty_tb + 1) * 3 + 2 * sd))) * 256);
        if (bf16) { rv[0] = round_bf16x4(rv[0]); rv[1] = round_bf16x4(rv[1]); }
        mac(wv[0], rv[0], false);
        mac(wv[1], add4(rv[0], rv[1]), false);
      }
    }
  }
  if (valid && !masked) {
    // D rows = ci 4*kq + r, col = pixel  ->  one float4 read-modify-write per lane (read issued at the top)
    float4 v = g_old;
    v.x += acc[0]; v.y += acc[1]; v.z += acc[2]; v.w += acc[3];
    if (bf16) stA4<true>(dbase, de, v); else stA4<false>(dbase, de, v);
  }
  if (bn_partials != nullptr) {
    // the BatchNorm-backward sums are linear in g: this wave adds (delta*mask, delta*mask*y) of its 16 border pixels
    // for its 16 channels (Cin == 16 here, nb == 0); rows of bn_partials = global wave index
    float d1[4] = {0.f, 0.f, 0.f, 0.f}, d2[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {   // loaded here, not at the top: 12 live registers through the MFMA loop would break the 64-register budget
      const float4 yv = bf16 ? ldA4<true>(bn_y, pix_o * 16 + 4 * kq) : ldA4<false>(bn_y, pix_o * 16 + 4 * kq);
      const float4 bsc = ld4(bn_scale + 4 * kq), bsh = ld4(bn_shift + 4 * kq);
      const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, scv[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, shv[4] = {bsh.x, bsh.y, bsh.z, bsh.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dz = fmaf(yy[r], scv[r], shv[r]) > 0.f ? acc[r] : 0.f;
        d1[r] = dz; d2[r] = dz * yy[r];
      }
      if (masked) {   // the main kernel stored dz = g * [z > 0] instead of g (conv_bwd16.hip, store_dz): the fold adds its masked part
        const float4 v = make_float4(g_old.x + d1[0], g_old.y + d1[1], g_old.z + d1[2], g_old.w + d1[3]);
        if (bf16) stA4<true>(dbase, de, v); else stA4<false>(dbase, de, v);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { d1[r] += __shfl_xor(d1[r], m); d2[r] += __shfl_xor(d2[r], m); }
    if (px == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bn_partials[((size_t)wave_g * 16 + 4 * kq + r) * 2 + 0] = d1[r];
        bn_partials[((size_t)wave_g * 16 + 4 * kq + r) * 2 + 1] = d2[r];
      }
    }
  }
}

}  // namespace

// Number of persistent workgroups (== rows of stat_partials written) for a B x H x W conv with cout outputs:
// at most 256 CUs x the residency the kernel variant reaches, and an even split of the tiles.
bool conv3x3_use_wino(const ConvArgs& a, int cout) {
  static const int off = getenv("SIFSR_NO_WINO") ? atoi(getenv("SIFSR_NO_WINO")) : 0;   // 1: direct kernels everywhere (A/B, debugging)
  // (16 output channels: the kernel keeps the layer's transform-domain weights in LDS, two channel blocks at most)
  return !off && a.wpack_wino != nullptr && a.bf16 == 0 && cout <= 64 && a.H % 2 == 0 && a.W % 2 == 0 && (cout > 16 || a.NQ <= 2);
}

// 0: tap-domain kernel; 1: Winograd, one workgroup per CU; 2: Winograd forward with 16 output channels, two per CU
int conv3x3_wino_kind(const ConvArgs& a, int cout, int zero_pad) {
  if (!conv3x3_use_wino(a, cout)) return 0;
  return (cout == 16 && !zero_pad) ? 2 : 1;
}

int conv3x3_grid_blocks(int B, int H, int W, int cout, int wino) {
  const int ntiles = B * ((H + 15) / 16) * ((W + 15) / 16);
  const int per_cu = wino ? (wino == 2 ? 2 : 1) : (cout >= 64 ? 1 : 2);   // residency of the kernel variants (VGPR-limited); wino == 2: the
                                                                            // two-workgroups-per-CU forward variant for 16 output channels
  static const int dbg_grid = getenv("SIFSR_DBG_CONV_GRID") ? atoi(getenv("SIFSR_DBG_CONV_GRID")) : 0;   // tuning knob
  const int gmax = dbg_grid > 0 ? dbg_grid : 256 * per_cu;
  if (ntiles <= gmax) return ntiles;
  const int rounds = (ntiles + gmax - 1) / gmax;
  int g = (ntiles + rounds - 1) / rounds;
  g = (g + 7) & ~7;                      // keep the XCD-contiguous mapping available
  return g < ntiles ? g : ntiles;
}

int launch_conv3x3_mfma(const ConvArgs& a, int cout, int zero_pad, hipStream_t s) {
  if (a.H < 1 || a.W < 1 || cout % 16 || a.NQ < 1 || a.src[0].nq + a.src[1].nq != a.NQ) return SIFSR_ERR_SHAPE;
  if (a.bf16 < 0 || a.bf16 > 1) return SIFSR_ERR_ARG;
  if (!a.src[0].ptr || !a.dst[0].ptr || !a.wpack) return SIFSR_ERR_ARG;
  const int ntiles = a.B * ((a.H + 15) / 16) * ((a.W + 15) / 16);
  const dim3 grid(conv3x3_grid_blocks(a.B, a.H, a.W, cout)), block(512);
  const int nb = cout / 16;
  // 32-bit byte offsets (buffer addressing): every tensor must stay below 4 GiB; channel counts powers of two
  const size_t npix = (size_t)a.B * a.H * a.W;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
  int cmax = a.src[0].C > a.src[1].C ? a.src[0].C : a.src[1].C;
  cmax = cmax > a.dst[0].C ? cmax : a.dst[0].C;
  cmax = cmax > a.dst[1].C ? cmax : a.dst[1].C;
  if (npix * cmax * 4 >= ((size_t)1 << 32) - 4096) return SIFSR_ERR_SHAPE;
  if (!pow2(a.src[0].C) || (a.src[1].ptr && !pow2(a.src[1].C))) return SIFSR_ERR_SHAPE;
  const int tx_ = (a.W + 15) / 16, ty_ = (a.H + 15) / 16;
  const int lgx = (pow2(tx_) && pow2(ty_)) ? lg(tx_) : -1, lgy = lgx >= 0 ? lg(ty_) : -1;
  const bool dyf = a.bw_y != nullptr;
  if (dyf && (!zero_pad || !a.bw_coef || a.src[1].ptr || a.src[0].scale || a.src[0].coff)) return SIFSR_ERR_ARG;
  if (nb == 8 && conv3x3_use_wino(a, 64) && a.stat_partials == nullptr && a.dst_split % 4 == 0) {
    // 128 output channels (the input gradient of ub1.convbloc.bloc.0): two Winograd launches of 64 channels each -- the
    // weight pack is [cout block][cin block][xi], so the second half is a pointer offset; each half lies in one destination
    for (int h = 0; h < 2; ++h) {
      ConvArgs w = a;
      w.wpack_wino = a.wpack_wino + (size_t)h * 4 * a.NQ * 16 * 256;
      const bool first = 4 * h < a.dst_split;
      w.dst[0] = first ? a.dst[0] : a.dst[1];
      w.dst[0].coff += first ? 64 * h : 16 * (4 * h - a.dst_split);
      w.dst[1] = w.dst[0];
      w.dst_split = 4;
      if (a.addend != nullptr) w.addend = a.addend + 64 * h;
      const int rc = launch_conv3x3_mfma(w, 64, zero_pad, s);
      if (rc != SIFSR_OK) return rc;
    }
    return SIFSR_OK;
  }
  if (conv3x3_use_wino(a, cout)) {
    // Winograd F(2x2,3x3) consumers (fp32, <= 64 output channels, even image sizes): a.wpack_wino replaces a.wpack
    ConvArgs w = a;
    w.wpack = a.wpack_wino;
    const dim3 wgrid(conv3x3_grid_blocks(a.B, a.H, a.W, cout, conv3x3_wino_kind(a, cout, zero_pad)));
    // 32 / 64 output channels: the kernel in which all eight waves stage and contract (conv_wino8.hip)
    if (conv3x3_wino8_applies(nb, a.NQ)) return launch_conv3x3_wino8(w, nb, zero_pad, dyf, (int)wgrid.x, ntiles, lgx, lgy, s);
#define SIFSR_WINO_LAUNCH(NBV, ZP, DY) hipLaunchKernelGGL((conv3x3_mfma_kernel<NBV, ZP, 0, DY, true>), wgrid, block, 0, s, w, ntiles, lgx, lgy)
#define SIFSR_WINO_CASE(NBV)                                                                              \
  case NBV:                                                                                               \
    if (dyf) SIFSR_WINO_LAUNCH(NBV, true, true);                                                          \
    else if (zero_pad) SIFSR_WINO_LAUNCH(NBV, true, false);                                               \
    else SIFSR_WINO_LAUNCH(NBV, false, false);                                                            \
    break;
    switch (nb) {
      SIFSR_WINO_CASE(1)
      SIFSR_WINO_CASE(2)
      SIFSR_WINO_CASE(4)
      default: return SIFSR_ERR_SHAPE;
    }
#undef SIFSR_WINO_CASE
#undef SIFSR_WINO_LAUNCH
    SIFSR_LAUNCH_CHECK();
    return SIFSR_OK;
  }
#define SIFSR_CONV_LAUNCH(NBV, ZP, MD, DY) hipLaunchKernelGGL((conv3x3_mfma_kernel<NBV, ZP, MD, DY, false>), grid, block, 0, s, a, ntiles, lgx, lgy)
#define SIFSR_CONV_MODE(NBV, MD)                                                                          \
    if (dyf) SIFSR_CONV_LAUNCH(NBV, true, MD, true);                                                      \
    else if (zero_pad) SIFSR_CONV_LAUNCH(NBV, true, MD, false);                                           \
    else SIFSR_CONV_LAUNCH(NBV, false, MD, false);
#define SIFSR_CONV_CASE(NBV)                                                                              \
  case NBV:                                                                                               \
    if (a.bf16) { SIFSR_CONV_MODE(NBV, 1) }                                                          \
    else { SIFSR_CONV_MODE(NBV, 0) }                                                                      \
    break;
  switch (nb) {
    SIFSR_CONV_CASE(1)
    SIFSR_CONV_CASE(2)
    SIFSR_CONV_CASE(4)
    SIFSR_CONV_CASE(8)
    default: return SIFSR_ERR_SHAPE;
  }
#undef SIFSR_CONV_CASE
#undef SIFSR_CONV_MODE
#undef SIFSR_CONV_LAUNCH
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_pack_weights(const float* params, float* wfwd, float* wdgrad, hipStream_t s, float* wwf, float* wwd, long long* nbt, int nbt_n) {
  const NetTable& nt = sifsr_net();
  PackTable tb;
  int maxn = 0;
  for (int l = 1; l < SIFSR_NUM_BN_LAYERS; ++l) {
    tb.w_off[l - 1] = nt.L[l].w_off;
    tb.cin[l - 1] = nt.L[l].cin;
    tb.cout[l - 1] = nt.L[l].cout;
    tb.p_off[l - 1] = nt.L[l].wpack_off;
    const int n = 9 * nt.L[l].cin * nt.L[l].cout;
    maxn = n > maxn ? n : maxn;
  }
  const dim3 grid((maxn + 255) / 256 > 64 ? 64 : (maxn + 255) / 256, 16);
  if (nbt != nullptr && (nbt_n < 1 || nbt_n > 256)) return SIFSR_ERR_ARG;
  if (wwf != nullptr && wwd != nullptr) {
    hipLaunchKernelGGL(pack_all_kernel, dim3(grid.x * 3, 16), dim3(256), 0, s, params, wfwd, wdgrad, wwf, wwd, tb, (int)grid.x, nbt, nbt_n);
  } else {
    if (nbt != nullptr) return SIFSR_ERR_ARG;
    hipLaunchKernelGGL(pack_weights_kernel, grid, dim3(256), 0, s, params, wfwd, wdgrad, tb);
  }
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_pack_weights_one(const float* w, int cin, int cout, float* wfwd, float* wdg, hipStream_t s, float* wwf, float* wwd) {
  PackTable tb;
  tb.w_off[0] = 0; tb.cin[0] = cin; tb.cout[0] = cout; tb.p_off[0] = 0;
  const int n = 9 * cin * cout;
  const dim3 grid((n + 255) / 256 > 64 ? 64 : (n + 255) / 256, 1);
  if (wfwd != nullptr && wdg != nullptr) {
    hipLaunchKernelGGL(pack_weights_kernel, grid, dim3(256), 0, s, w, wfwd, wdg, tb);
    SIFSR_LAUNCH_CHECK();
  }
  if (wwf != nullptr && wwd != nullptr) {
    hipLaunchKernelGGL(pack_wino_kernel, dim3(grid.x * 2, 1), dim3(256), 0, s, w, wwf, wwd, tb);
    SIFSR_LAUNCH_CHECK();
  }
  return SIFSR_OK;
}

int dgrad_border_waves(int B, int H, int W, int Cin) {
  return B * (2 * ((W + 15) / 16) + 2 * ((H - 2 + 15) / 16)) * (Cin / 16);
}

int launch_dgrad_border_fix(const float* dy, int Cout, const float* wdg_layer, int Cin, float* g0, int C0,
                            int split_ch, float* g1, int C1, int B, int H, int W, hipStream_t s, int bf16,
                            const float* bn_y, const float* bn_scale, const float* bn_shift, float* bn_partials, int masked) {
  if (bn_partials != nullptr && (Cin != 16 || !bn_y || !bn_scale || !bn_shift)) return SIFSR_ERR_ARG;
  if (masked && bn_partials == nullptr) return SIFSR_ERR_ARG;
  if (H < 3 || W < 2 || Cin % 16 || Cout % 16) return SIFSR_ERR_SHAPE;
  const int waves = B * (2 * ((W + 15) / 16) + 2 * ((H - 2 + 15) / 16)) * (Cin / 16);
  hipLaunchKernelGGL(dgrad_border_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, dy, Cout, wdg_layer, Cin, g0, C0,
                     split_ch, g1, C1, B, H, W, bf16, bn_y, bn_scale, bn_shift, bn_partials, masked);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
