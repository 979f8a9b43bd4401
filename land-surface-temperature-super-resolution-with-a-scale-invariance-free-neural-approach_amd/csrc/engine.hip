// ModelB_2 forward / backward as an explicit launch schedule on one HIP stream
// (model.py:608-645 and its autograd transpose).  No allocation, no host sync: every buffer is a
// fixed offset into a caller-provided workspace, so a whole step is hipGraph-capturable.
//
// Dataflow per Conv-BN-ReLU unit l (training):
//   conv kernel: y_l = conv(a_in)            a_in = relu(y_prev*scale+shift) applied on load
//                + per-workgroup (sum,sumsq) -> bn_finalize -> mean/invstd/scale/shift, running stats
//   backward:    bn_bwd_reduce(g_l, y_l) [or sums from the dgrad above] -> bn_bwd_finalize -> dgamma, dbeta, (sc, sh, k1, k0)
//                dy_l = sc*g_l*[z>0] + k1*z + k0, z = y_l*sc + sh, is NEVER stored: formed while staging (bn_bwd4) by
//                wgrad(a_in, g_l, y_l) -> slabs -> reduce -> dW  and  dgrad(g_l, y_l) (+ border fold) -> g of the inputs
#include "engine.h"

#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------------------------
// activation storage of the current call (common.h)
// ---------------------------------------------------------------------------------------------
static thread_local bool t_half_storage = false;
bool sifsr_half_storage() { return t_half_storage; }
HalfStorageScope::HalfStorageScope(bool on) : prev(t_half_storage) { t_half_storage = on; }
HalfStorageScope::~HalfStorageScope() { t_half_storage = prev; }

// ---------------------------------------------------------------------------------------------
// network table
// ---------------------------------------------------------------------------------------------
static NetTable build_net() {
  static const int cin[SIFSR_NUM_BN_LAYERS] = {2, 16, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 64, 64, 32, 32, 16};
  static const int cout[SIFSR_NUM_BN_LAYERS] = {16, 16, 16, 16, 32, 32, 32, 64, 64, 64, 64, 64, 32, 32, 16, 16, 16};
  static const int level[SIFSR_NUM_BN_LAYERS] = {0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 2, 2, 1, 1, 0, 0};
  NetTable t;
  int p = 0, r = 0, c = 0, wp = 0;
  for (int l = 0; l < SIFSR_NUM_BN_LAYERS; ++l) {
    LayerInfo& L = t.L[l];
    L.cin = cin[l]; L.cout = cout[l]; L.level = level[l];
    L.w_off = p; p += cout[l] * cin[l] * 9;
    L.gamma_off = p; p += cout[l];
    L.beta_off = p; p += cout[l];
    L.run_off = r; r += 2 * cout[l];
    L.ch_off = c; c += cout[l];
    L.wpack_off = wp;
    if (l > 0) wp += 9 * cin[l] * cout[l];
  }
  t.out_w_off = p; p += 144;
  t.out_b_off = p; p += 1;
  t.total_params = p; t.total_running = r; t.total_channels = c; t.total_wpack = wp;
  return t;
}
const NetTable& sifsr_net() {
  static const NetTable t = build_net();
  return t;
}

// ---------------------------------------------------------------------------------------------
// workspace layout
// ---------------------------------------------------------------------------------------------
static size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }   // in floats (256 B)

static int wgrad_blocks(int cin, int cout, int chunks, int ntiles, bool wino = false) {
  // persistent, software-pipelined workgroups: two per CU are resident (four for the 16->16 variant: 20 KB LDS,
  // 66 VGPRs), so launch that many in total (x-dim = total / Cin chunks in blockIdx.y), but keep >= 4 tiles per
  // workgroup so that the slab write + slab reduction stay small next to the MFMA work (measured per layer).
  // (re-swept with the weight gradients on the second stream: smaller grids leave the chain more room but lose more
  // than they give -- 1024 / 512 stay)
  static const int dbg_scale = getenv("SIFSR_DBG_WGRAD_GRID_PCT") ? atoi(getenv("SIFSR_DBG_WGRAD_GRID_PCT")) : 100;   // tuning knob
  // (round 2, beside the Winograd chain -- one workgroup per CU, registers to spare: 1.5x the round-1 grids, +1 %; flat to 2.5x)
  // Winograd-domain kernels (conv_wgrad_wino.hip): two workgroups per CU are resident (registers), every workgroup walks the same
  // number of tiles -> exactly one round of resident workgroups
  static const int dbg_wino = getenv("SIFSR_DBG_WGRAD_WINO_GRID") ? atoi(getenv("SIFSR_DBG_WGRAD_WINO_GRID")) : 512;
  static const int dbg_wino11 = getenv("SIFSR_DBG_WGRAD_WINO_GRID11") ? atoi(getenv("SIFSR_DBG_WGRAD_WINO_GRID11")) : 512;
  static const int dbg_wino42 = getenv("SIFSR_DBG_WGRAD_WINO_GRID42") ? atoi(getenv("SIFSR_DBG_WGRAD_WINO_GRID42")) : 256;   // 64 output channels: one resident workgroup per CU
  const int wino_total = (cin == 16 && cout == 16) ? dbg_wino11 : cout == 64 ? dbg_wino42 : dbg_wino;
  const int total = wino ? (wino_total > 768 ? 768 : wino_total) : ((cin == 16 && cout == 16) ? 1536 : 768) * dbg_scale / 100;
  int n = total / chunks;
  if (n < 64) n = 64;
  const int cap = ntiles / 4 > 0 ? ntiles / 4 : 1;
  if (n > cap) n = cap;
  return n;
}

int sifsr_layout(int B, int H, int W, int training, WsLayout* o) {
  // three 2x poolings + exact x2 upsamplings back: H, W multiples of 8 (model.py:597-603), as in the reference; the
  // deepest level (H/8 x W/8) must still have a 3x3 neighbourhood
  if (B < 1 || H < 24 || W < 24 || H % 8 || W % 8) return SIFSR_ERR_SHAPE;
  const NetTable& nt = sifsr_net();
  WsLayout& w = *o;
  size_t off = 0;
  auto take = [&](size_t n) { const size_t r = off; off += align64(n); return r; };
  size_t N[4];
  N[0] = (size_t)B * H * W; N[1] = N[0] / 4; N[2] = N[0] / 16; N[3] = N[0] / 64;
  for (int i = 0; i < 4; ++i) w.npix[i] = N[i];

  w.mean = take(nt.total_channels); w.invstd = take(nt.total_channels);
  w.scale = take(nt.total_channels); w.shift = take(nt.total_channels);
  w.wfwd = take(nt.total_wpack); w.wdg = take(4 * (size_t)nt.total_wpack);
  w.wwf = take((size_t)nt.total_wpack / 9 * 16); w.wwd = take((size_t)nt.total_wpack / 9 * 16);   // Winograd-domain packs
  for (int l = 0; l < SIFSR_NUM_BN_LAYERS; ++l) w.y[l] = take(nt.L[l].cout * N[nt.L[l].level]);
  static const int pc[3] = {16, 32, 64};
  for (int k = 0; k < 3; ++k) {
    w.P[k] = take(pc[k] * N[k + 1]);
    w.R[k] = take(pc[k] * N[k + 1]);
  }
  w.U[0] = take(64 * N[2]); w.U[1] = take(32 * N[1]); w.U[2] = take(16 * N[0]);

  // scratch: BN statistic partials (forward: one entry per conv workgroup; backward: <= 1024 blocks)
  size_t maxpart = 0;
  for (int l = 0; l < SIFSR_NUM_BN_LAYERS; ++l) {
    const int lh = H >> nt.L[l].level, lw = W >> nt.L[l].level;
    const size_t nblk = (size_t)B * ((lh + 15) / 16) * ((lw + 15) / 16);   // 16x16 tiles, the last row / column partial
    const size_t n = (nblk > 1024 ? nblk : 1024) * nt.L[l].cout * 2;
    maxpart = n > maxpart ? n : maxpart;
  }
  // ... and the upsample adjoints' rows (one per 8x8 low-resolution pixels and image) for the three decoder inputs
  for (int k = 0; k < 3; ++k) {
    const int lh = H >> (3 - k), lw = W >> (3 - k);
    const size_t n = (size_t)up2x_bwd_stat_rows(B, lh, lw, 64 >> k) * (64 >> k) * 2;
    maxpart = n > maxpart ? n : maxpart;
  }
  w.partials = take(maxpart);
  w.partials_cap = maxpart;
  w.fwd_end = off;
  if (training) {
    w.coef = take(6 * 64);   // 3 x C float64 BN-backward coefficients of the layer being processed
    w.coef_f = take(4 * (size_t)nt.total_channels);
    w.bpart = take((size_t)dgrad_border_waves(B, H, W, 16) * 32);   // level 0 is the largest user
    for (int l = 0; l < SIFSR_NUM_BN_LAYERS; ++l) w.g[l] = take(nt.L[l].cout * N[nt.L[l].level]);
    w.dy_border = take(16 * N[0]);   // the largest layer (16 channels at level 0 = 32 at level 1 = 64 at level 2)
    w.dyB[0] = w.dyB[1] = w.dyB[2] = w.dy_border;
    for (int k = 0; k < 3; ++k) w.gP[k] = take(pc[k] * N[k + 1]);
    w.gU[0] = take(64 * N[2]); w.gU[1] = take(32 * N[1]); w.gU[2] = take(16 * N[0]);
    w.slabs = take(1024 * 288);   // edge-layer partials
    w.gram = take(conv_in_gram_scratch_floats());
    w.slab_l[0] = 0; w.slab_cap[0] = 0;
    for (int l = 1; l < SIFSR_NUM_BN_LAYERS; ++l) {
      const int lvh = H >> nt.L[l].level, lvw = W >> nt.L[l].level;
      const int ntiles = B * ((lvh + 7) / 8) * ((lvw + 15) / 16);
      // upper bound over chunkings (x-dim blocks * chunks <= blocks at one chunk)
      // (9 values per weight pair: the Winograd F(3x3,2x2) kernels apply their output transform before they write a slab)
      // ... and over the two kernel families: the tap-domain grid scales with SIFSR_DBG_WGRAD_GRID_PCT, the Winograd one does not
      const int nb_tap = wgrad_blocks(nt.L[l].cin, nt.L[l].cout, 1, ntiles, false), nb_wino = wgrad_blocks(nt.L[l].cin, nt.L[l].cout, 1, ntiles, true);
      w.slab_cap[l] = (size_t)(nb_tap > nb_wino ? nb_tap : nb_wino) * 9 * nt.L[l].cin * nt.L[l].cout;
      // ... and the fused input + weight gradient kernel of the 16 -> 16 layers writes one slab per workgroup
      if (nt.L[l].cin == 16 && nt.L[l].cout == 16 && conv3x3_bwd16_applies(B, lvh, lvw)) {
        const size_t need = (size_t)conv3x3_bwd16_grid(B, lvh, lvw) * 9 * 256;
        if (need > w.slab_cap[l]) w.slab_cap[l] = need;
      }
      w.slab_l[l] = take(w.slab_cap[l]);
    }
  }
  w.total = off;
  return SIFSR_OK;
}

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
namespace {

struct Ctx {
  const NetTable& nt;
  WsLayout lay;
  float* ws;
  const float* params;
  int B, H, W;
  hipStream_t s;
  WgradReduceJob* jobs = nullptr;   // backward: slab reductions deferred to one batched launch
  int* njobs = nullptr;
  bool wino_wgrads = false;         // backward: the Winograd-domain weight-gradient kernels (and the fused 16 -> 16 kernel) may run
  int bf16 = 0;                     // 1: bf16 MFMA operands (config 5)
  struct SideLane* side = nullptr;  // backward: the weight gradients' own stream (nullptr = everything on s)
  bool* forked = nullptr;           // set once anything was enqueued on the side stream (SideLaneGuard)
  int lvH(int lv) const { return H >> lv; }
  int lvW(int lv) const { return W >> lv; }
  float* f(size_t off) const { return ws + off; }
  const float* scale(int l) const { return ws + lay.scale + nt.L[l].ch_off; }
  const float* shift(int l) const { return ws + lay.shift + nt.L[l].ch_off; }
};

ConvSrc src_raw(const float* p, int C) { ConvSrc s; s.ptr = p; s.scale = nullptr; s.shift = nullptr; s.C = C; s.coff = 0; s.nq = C / 16; return s; }
ConvSrc src_act(const Ctx& c, int l) {
  ConvSrc s; s.ptr = c.f(c.lay.y[l]); s.scale = c.scale(l); s.shift = c.shift(l); s.C = c.nt.L[l].cout; s.coff = 0; s.nq = s.C / 16; return s;
}
ConvSrc src_none() { ConvSrc s; s.ptr = nullptr; s.scale = nullptr; s.shift = nullptr; s.C = 0; s.coff = 0; s.nq = 0; return s; }

#define SIFSR_TRY(expr) do { int rc__ = (expr); if (rc__ != SIFSR_OK) return rc__; } while (0)

// ---- the weight gradients' stream.  A layer's wgrad is a leaf of the backward dataflow: it reads dy_l and the saved
// forward activations (each in its own workspace region, never rewritten during the backward) and writes its own
// slab region, and nothing but the final slab reduction consumes it.  The chain that *is* serial (BatchNorm backward
// -> dgrad -> pool / upsample adjoints) alternates MFMA-bound and HBM-bound kernels, so the wgrads run on a second,
// lower-priority stream and fill the matrix cores while the chain's memory-bound kernels stream:
//   main:  ... bn_bwd(l) --record ev[l]--> dgrad(l) -> bn_bwd(l-1) ...            -> wait(join) -> slab reduce
//   side:                  wait ev[l] -> wgrad(l)            ... -> record join
// Fork / join by events only.  Measured (B = 64):
// 8.52 -> 8.40 ms per step, and -> 8.0 ms once the chain's kernels raise their wave priority (SIFSR_CHAIN_PRIO, common.h):
// without it the small latency-bound kernels of the chain (border fold, BatchNorm finalize) ran 5-10x slower beside a
// weight-gradient kernel than alone.  Issuing wgrad(l) behind dgrad(l) instead of beside it was worse (8.57 ms), and
// so was running the border fold beside the main dgrad kernel on a third stream (two more event hand-offs per layer).
// SIFSR_WGRAD_STREAM=0 / sifsr_set_wgrad_stream(0) disables the second stream.
struct SideLane {
  hipStream_t s = nullptr;
  hipEvent_t ev[SIFSR_NUM_BN_LAYERS] = {};
  hipEvent_t join = nullptr;
  hipEvent_t aux = nullptr;    // side -> main hand-off that is not the final join (the first layer's Gram matrix)
  bool ok = false;
  std::mutex in_use;   // held for a whole backward enqueue: two host threads of one device never interleave on ev[]
};

int g_side_override = -1;   // sifsr_engine_set_wgrad_stream: -1 = environment default

SideLane* side_lane(hipStream_t main_stream) {
  static const int env_default = getenv("SIFSR_WGRAD_STREAM") ? atoi(getenv("SIFSR_WGRAD_STREAM")) : 1;   // 0: single stream
  if (!(g_side_override >= 0 ? g_side_override : env_default)) return nullptr;
  static std::mutex mu;
  static SideLane lanes[16];   // per device; events are re-recorded per call, so calls on different caller streams are
                               // safe as long as their enqueues do not interleave (SideLane::in_use, SideLaneGuard)
  static bool tried[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  // Under stream capture the backward stays on the one stream: the fork/join would be legal, but hipGraph replay of the
  // resulting two-branch graph is slow on ROCm 7.2 (measured 11.6 ms per step against 8.5 ms for the linear graph and
  // 8.0 ms for eager two-stream execution at batch 64).
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(main_stream, &cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (cap != hipStreamCaptureStatusNone) return nullptr;
  if (main_stream != nullptr) {   // the caller's stream must live on the calling thread's device (the lane is per device)
    hipDevice_t sdev = -1;
    if (hipStreamGetDevice(main_stream, &sdev) != hipSuccess || (int)sdev != dev) { (void)hipGetLastError(); return nullptr; }
  }
  std::lock_guard<std::mutex> lk(mu);
  SideLane& L = lanes[dev];
  if (!tried[dev]) {
    tried[dev] = true;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = numerically largest = lowest priority
    bool ok = hipStreamCreateWithPriority(&L.s, hipStreamNonBlocking, lo) == hipSuccess;
    for (int i = 0; ok && i < SIFSR_NUM_BN_LAYERS; ++i) ok = hipEventCreateWithFlags(&L.ev[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&L.join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&L.aux, hipEventDisableTiming) == hipSuccess;
    L.ok = ok;
    (void)hipGetLastError();
  }
  return L.ok ? &L : nullptr;
}

// Holds the lane for one backward call and guarantees the join: once any weight gradient has been forked onto the
// second stream, EVERY way out of sifsr_engine_backward -- including an early error return -- records the join event
// behind the side stream's work and makes the caller's stream wait for it, so the caller may free or reuse the
// workspace / gradient buffers in stream order as with a single-stream call.
struct SideLaneGuard {
  SideLane* lane; hipStream_t main; bool forked = false, joined = false;
  SideLaneGuard(SideLane* l, hipStream_t m) : lane(l), main(m) { if (lane) lane->in_use.lock(); }
  int join() {
    if (!lane || joined) return SIFSR_OK;
    joined = true;
    if (!forked) return SIFSR_OK;
    if (hipEventRecord(lane->join, lane->s) != hipSuccess || hipStreamWaitEvent(main, lane->join, 0) != hipSuccess) return SIFSR_ERR_ARG;
    return SIFSR_OK;
  }
  ~SideLaneGuard() { if (lane) { (void)join(); lane->in_use.unlock(); } }
};

// ---- optional per-kernel timing (bench.py roofline): HIP events on the launch stream around ONE
// selected (layer, phase) launch inside the normal schedule.  phase 1 fwd conv, 2 dgrad, 3 wgrad.
// Up to PROF_SLOTS (layer, phase) selections are timed side by side.  The event pairs are created by
// sifsr_engine_profile_select / _add (i.e. before the caller's timed region), never inside a launch; a launch that finds
// its pool exhausted is simply not timed.  One mutex serialises selection, use and read.
constexpr int PROF_SLOTS = 24;
constexpr size_t PROF_POOL = 1024;   // launches of one selected kernel that can be timed before the next select
struct ProfSlot {
  int layer = -1, phase = 0;
  std::vector<hipEvent_t> start, stop;   // the pool
  size_t used = 0;
};
struct ProfState {
  ProfSlot slot[PROF_SLOTS];
  int active = 0;                        // number of slots in use (0: profiling off, the common case)
  std::mutex mu;
};
ProfState g_prof;

struct ProfScope {
  hipStream_t s; int sl = -1; long idx = -1;
  ProfScope(int layer, int phase, hipStream_t st) : s(st) {
    if (g_prof.active == 0) return;   // unlocked fast path
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int k = 0; k < g_prof.active; ++k) {
      ProfSlot& p = g_prof.slot[k];
      if (p.layer == layer && p.phase == phase && p.used < p.start.size()) {
        sl = k; idx = (long)p.used++;
        (void)hipEventRecord(p.start[idx], s);
        return;
      }
    }
  }
  ~ProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    ProfSlot& p = g_prof.slot[sl];
    if ((size_t)idx < p.stop.size()) (void)hipEventRecord(p.stop[idx], s);
  }
};

// forward of one MFMA Conv(-BN) unit
int conv_unit_fwd(const Ctx& c, int l, ConvSrc s0, ConvSrc s1, int training, float* running, float momentum, float eps) {
  const LayerInfo& L = c.nt.L[l];
  ConvArgs a;
  a.src[0] = s0; a.src[1] = s1;
  a.dst[0].ptr = c.f(c.lay.y[l]); a.dst[0].C = L.cout; a.dst[0].coff = 0;
  a.dst[1] = a.dst[0];
  a.bf16 = c.bf16;
  {
    const size_t n = (size_t)9 * L.cin * L.cout;   // bf16 packs live behind the layer's fp32 dgrad pack
    a.wpack = a.bf16 ? c.f(c.lay.wdg) + 4 * (size_t)L.wpack_off + n : c.f(c.lay.wfwd) + L.wpack_off;
  }
  a.wpack_wino = c.f(c.lay.wwf) + (size_t)L.wpack_off / 9 * 16;
  a.addend = nullptr; a.addC = 0;
  a.stat_partials = training ? c.f(c.lay.partials) : nullptr;
  a.dst_split = L.cout / 16;
  a.B = c.B; a.H = c.lvH(L.level); a.W = c.lvW(L.level);
  a.NQ = L.cin / 16;
  {
    ProfScope ps(l, 1, c.s);
    SIFSR_TRY(launch_conv3x3_mfma(a, L.cout, 0, c.s));
  }
  if (training) {
    const int nblk = conv3x3_grid_blocks(c.B, a.H, a.W, L.cout, conv3x3_wino_kind(a, L.cout, 0));
    SIFSR_TRY(launch_bn_finalize(c.f(c.lay.partials), nblk, L.cout, (double)c.B * a.H * a.W, c.params + L.gamma_off,
                                 c.params + L.beta_off, running + L.run_off, running + L.run_off + L.cout, momentum, eps,
                                 c.f(c.lay.mean) + L.ch_off, c.f(c.lay.invstd) + L.ch_off,
                                 c.f(c.lay.scale) + L.ch_off, c.f(c.lay.shift) + L.ch_off, c.s));
  }
  return SIFSR_OK;
}

// BatchNorm+ReLU backward of unit l, statistics half: g = dL/d relu(bn(y_l)) -> dgamma / dbeta into grads and the
// coefficients with which the CONSUMERS of dL/dy_l (this layer's input- and weight-gradient convolutions; the fused head
// kernel for inbloc.bloc.0) form it from (g, y_l) while staging -- the elementwise pass and dL/dy_l itself do not exist.
// gp != nullptr: g is first completed by the AvgPool adjoint of the half-resolution gradient gp, in place (bn.hip PoolAdj).
// fused_stats > 0: the sums were already produced by the dgrad (fused_stats = its workgroup count) + border kernel of the
// layer above (conv_unit_dgrad with bn_layer), so the reduce pass over (g, y) is skipped.
// border_rows: rows the border-fold kernel added to bpart (default: those of a 16-channel dgrad); 0 when the sums came
// from the upsample adjoint (resample.hip), which has no border part.
int bn_unit_bwd(const Ctx& c, int l, float* g, float* grads, const float* gp = nullptr, int fused_stats = 0, int border_rows = -1,
                bool writeback = true) {   // writeback = false (with gp): the consumer adds the pooling adjoint itself while staging
  const LayerInfo& L = c.nt.L[l];
  const int lh = c.lvH(L.level), lw = c.lvW(L.level);
  const size_t npix = c.lay.npix[L.level];
  float* coef_f = c.f(c.lay.coef_f) + 4 * (size_t)L.ch_off;
  if (fused_stats > 0) {
    if (gp != nullptr) return SIFSR_ERR_ARG;
    return launch_bn_bwd_finalize2(c.f(c.lay.partials), fused_stats, c.f(c.lay.bpart), border_rows >= 0 ? border_rows : dgrad_border_waves(c.B, lh, lw, 16),
                                   L.cout, (double)npix, c.scale(l), c.f(c.lay.mean) + L.ch_off,
                                   c.f(c.lay.invstd) + L.ch_off, grads + L.gamma_off, grads + L.beta_off,
                                   reinterpret_cast<double*>(c.f(c.lay.coef)), c.s, c.shift(l), c.params + L.beta_off, coef_f);
  }
  size_t nb = npix / 256;
  const int nblk = (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
  SIFSR_TRY(launch_bn_bwd_reduce(g, c.f(c.lay.y[l]), c.scale(l), c.shift(l), c.f(c.lay.mean) + L.ch_off, c.f(c.lay.invstd) + L.ch_off,
                                 L.cout, npix, c.f(c.lay.partials), nblk, c.s, gp, lh, lw, (gp && writeback) ? g : nullptr));
  return launch_bn_bwd_finalize(c.f(c.lay.partials), nblk, L.cout, (double)npix, c.scale(l), c.f(c.lay.mean) + L.ch_off,
                                c.f(c.lay.invstd) + L.ch_off, grads + L.gamma_off, grads + L.beta_off,
                                reinterpret_cast<double*>(c.f(c.lay.coef)), c.s, c.shift(l), c.params + L.beta_off, coef_f);
}

static bool wgrad_wino_policy(int cin, int cout) {
  // SIFSR_WGRAD_WINO: 0 = tap-domain kernels, 1 = Winograd up to 32 output channels, 2 = every layer (default; +5 % on the step)
  static const int mode = getenv("SIFSR_WGRAD_WINO") ? atoi(getenv("SIFSR_WGRAD_WINO")) : 2;
  (void)cin;
  return mode == 2 || (mode == 1 && cout <= 32);
}

// weight gradient of MFMA unit l from its forward inputs and dy
// dy_stored: `dy` is dL/dy_l itself (ub3.convbloc.bloc.3, written by the fused tail); otherwise it is g_l and dL/dy_l is
// formed while staging from (g_l, y_l, the coefficients bn_unit_bwd left)
int conv_unit_wgrad(const Ctx& c, int l, ConvSrc s0, ConvSrc s1, const float* dy, float* grads, bool dy_stored = false) {
  const LayerInfo& L = c.nt.L[l];
  WgradArgs a;
  a.src[0] = s0; a.src[1] = s1;
  a.dy = dy; a.slabs = c.f(c.lay.slab_l[l]);
  if (!dy_stored) { a.dy_y = c.f(c.lay.y[l]); a.dy_coef = c.f(c.lay.coef_f) + 4 * (size_t)L.ch_off; }
  a.B = c.B; a.H = c.lvH(L.level); a.W = c.lvW(L.level);
  a.NQ = L.cin / 16;
  a.ntiles = c.B * ((a.H + 7) / 8) * ((a.W + 15) / 16);
  a.bf16 = c.bf16;
  // Winograd F(3x3,2x2) where it is the faster form (measured per shape, tools/sweep_layers.sh)
  const bool wino = c.wino_wgrads && wgrad_wino_policy(L.cin, L.cout) && conv3x3_wgrad_use_wino(a, L.cin, L.cout);
  const int nbi = wino ? wgrad_wino_nbi_chunk(a, L.cin) : wgrad_nbi_chunk(a, L.cin);
  const int nblk = wgrad_blocks(L.cin, L.cout, L.cin / (16 * nbi), a.ntiles, wino);
  // every workgroup (x cin chunks) writes one slab of (16 | 9) * cin_chunk * cout floats into this layer's region
  if ((size_t)nblk * (L.cin / (16 * nbi)) * 9 * (16 * nbi) * L.cout > c.lay.slab_cap[l]) return SIFSR_ERR_WORKSPACE;
  hipStream_t ws = c.s;
  if (c.side != nullptr && c.jobs != nullptr) {   // dy_l is complete on the main stream at this point
    if (hipEventRecord(c.side->ev[l], c.s) != hipSuccess || hipStreamWaitEvent(c.side->s, c.side->ev[l], 0) != hipSuccess)
      return SIFSR_ERR_ARG;
    ws = c.side->s;
    if (c.forked) *c.forked = true;
  }
  {
    ProfScope ps(l, 3, ws);
    if (wino) SIFSR_TRY(launch_conv3x3_wgrad_wino(a, L.cin, L.cout, nblk, ws));
    else SIFSR_TRY(launch_conv3x3_wgrad(a, L.cin, L.cout, nblk, ws));
  }
  if (c.jobs != nullptr) {
    WgradReduceJob& j = c.jobs[(*c.njobs)++];   // (both kernel families leave tap-domain slabs)
    j.slab_off = c.lay.slab_l[l]; j.nblk = nblk; j.cin = L.cin; j.cout = L.cout; j.nbi_chunk = nbi; j.w_off = L.w_off;
  } else {
    SIFSR_TRY(launch_wgrad_reduce(a.slabs, nblk, L.cin, L.cout, nbi, grads + L.w_off, c.s));
  }
  return SIFSR_OK;
}

// input gradient of MFMA unit l: g_in = conv^T(dy) with the replicate-border fold.
// Output channels [0, split_ch) -> (g0, C0); the rest -> (g1, C1).  addend (C = cin) is added to g0.
// bn_layer >= 0: g0 is the complete gradient w.r.t. relu(bn(y_bn_layer)) (16 channels, no split, no addend): the kernel
// and the border kernel also emit that layer's BatchNorm-backward sums; returns the number of partial rows through
// *stat_rows (0 when the fusion does not apply).
// dy_stored: as for conv_unit_wgrad.
int conv_unit_dgrad(const Ctx& c, int l, const float* dy, float* g0, int C0, int split_ch, float* g1, int C1,
                    const float* addend, int bn_layer = -1, int* stat_rows = nullptr, bool dy_stored = false) {
  const LayerInfo& L = c.nt.L[l];
  const bool fuse = bn_layer >= 0 && L.cin == 16 && C0 == 16 && split_ch == 16 && g1 == nullptr && addend == nullptr &&
                    c.nt.L[bn_layer].cout == 16 && c.nt.L[bn_layer].level == L.level;
  if (stat_rows) *stat_rows = 0;
  ConvArgs a;
  a.src[0] = src_raw(dy, L.cout); a.src[1] = src_none();
  const float* dy_edge = dy;   // what the border-fold kernel reads (border pixels only)
  if (!dy_stored) {
    a.bw_y = c.f(c.lay.y[l]); a.bw_coef = c.f(c.lay.coef_f) + 4 * (size_t)L.ch_off; a.bw_border = c.f(c.lay.dy_border);
    dy_edge = a.bw_border;
  }
  a.dst[0].ptr = g0; a.dst[0].C = C0; a.dst[0].coff = 0;
  a.dst[1].ptr = g1 ? g1 : g0; a.dst[1].C = g1 ? C1 : C0; a.dst[1].coff = 0;
  a.bf16 = c.bf16;
  const float* wdg_f32 = c.f(c.lay.wdg) + 4 * (size_t)L.wpack_off;
  {
    const size_t n = (size_t)9 * L.cin * L.cout;
    a.wpack = a.bf16 ? wdg_f32 + n + n / 2 : wdg_f32;   // [fp32 dgrad | fwd bf16 (n/2 floats) | dgrad bf16 | unused]
  }
  a.wpack_wino = c.f(c.lay.wwd) + (size_t)L.wpack_off / 9 * 16;
  a.addend = addend; a.addC = L.cin;
  a.stat_partials = nullptr;
  a.dst_split = split_ch / 16;
  a.B = c.B; a.H = c.lvH(L.level); a.W = c.lvW(L.level);
  a.NQ = L.cout / 16;
  if (fuse) {
    a.stat_partials = c.f(c.lay.partials);
    a.bn_y = c.f(c.lay.y[bn_layer]); a.bn_scale = c.scale(bn_layer); a.bn_shift = c.shift(bn_layer);
    if (stat_rows) *stat_rows = conv3x3_grid_blocks(c.B, a.H, a.W, L.cin, conv3x3_wino_kind(a, L.cin, 1));
  }
  {
    ProfScope ps(l, 2, c.s);
    SIFSR_TRY(launch_conv3x3_mfma(a, L.cin, 1, c.s));
  }
  SIFSR_TRY(launch_dgrad_border_fix(dy_edge, L.cout, wdg_f32, L.cin, g0, C0, split_ch, g1 ? g1 : g0, g1 ? C1 : C0, c.B, a.H,
                                    a.W, c.s, c.bf16, fuse ? a.bn_y : nullptr, fuse ? a.bn_scale : nullptr,
                                    fuse ? a.bn_shift : nullptr, fuse ? c.f(c.lay.bpart) : nullptr));
  return SIFSR_OK;
}

// Input gradient AND weight gradient of a 16 -> 16 channel MFMA unit in ONE kernel (conv_bwd16.hip) where its shape allows:
// (g_l, y_l, the forward input) are read once for both passes.  Arguments as conv_unit_wgrad + conv_unit_dgrad of the same
// layer (s0 = the forward input, 16 channels; gin = the gradient w.r.t. it, `addend` added; bn_layer as conv_unit_dgrad).
// Returns SIFSR_OK with *applied = false when the separate kernels have to run instead (other shapes, bf16 mode, switched off).
// The kernel runs on the CALLER's stream (it is part of the serial chain); its weight-gradient slabs join the Winograd
// reduction list, and ev[l] of the second stream's lane marks their completion for a reduction issued there.
bool bwd16_usable(const Ctx& c, int l, ConvSrc s0) {
  const LayerInfo& L = c.nt.L[l];
  // bf16 mode: the fused kernel exists for bf16 STORAGE too (fp32 Winograd arithmetic on widened values), but there it LOSES:
  // 14,450 against 15,900 patches/s on the bf16 step -- with half the bytes its fp32 matrix + transform work is the bottleneck,
  // while the separate bf16 kernels contract with 16x cheaper MFMAs.  SIFSR_BF16_BWD16=1 selects it for A/B.
  static const int bf16_fused = getenv("SIFSR_BF16_BWD16") ? atoi(getenv("SIFSR_BF16_BWD16")) : 0;
  return L.cin == 16 && L.cout == 16 && (c.bf16 == 0 || bf16_fused) && c.wino_wgrads && wgrad_wino_policy(16, 16) && s0.C == 16 &&
         s0.coff == 0 && conv3x3_bwd16_applies(c.B, c.lvH(L.level), c.lvW(L.level));
}
// dy_mode 0: `dy` is dL/dy itself.  1: `dy` is g = dL/d relu(bn(y_l)), dL/dy formed while staging.  2 (l = ub3.convbloc.bloc.3): `dy` is
// d loss / d sr and g the input gradient of outlay, recomputed while staging (no tail_bwd_apply pass).
int conv_unit_bwd16(const Ctx& c, int l, ConvSrc s0, const float* dy, float* gin, const float* addend, int bn_layer, int* stat_rows,
                    int dy_mode, bool* applied, bool store_dz = false, const float* pool_gp = nullptr) {
  const LayerInfo& L = c.nt.L[l];
  *applied = false;
  if (stat_rows) *stat_rows = 0;
  const int lh = c.lvH(L.level), lw = c.lvW(L.level);
  if (!bwd16_usable(c, l, s0)) return dy_mode == 2 ? SIFSR_ERR_ARG : SIFSR_OK;
  const bool dy_stored = dy_mode == 0;
  const bool fuse = bn_layer >= 0 && addend == nullptr && c.nt.L[bn_layer].cout == 16 && c.nt.L[bn_layer].level == L.level;
  const int grid = conv3x3_bwd16_grid(c.B, lh, lw);
  if ((size_t)grid * 9 * 256 > c.lay.slab_cap[l] || (fuse && (size_t)grid * 32 > c.lay.partials_cap)) return SIFSR_ERR_WORKSPACE;
  Bwd16Args a;
  a.x = s0.ptr; a.x_scale = s0.scale; a.x_shift = s0.shift;
  if (dy_mode == 2) { a.tail_dsr = dy; a.tail_w = c.params + c.nt.out_w_off; }
  else a.g = dy;
  if (!dy_stored) { a.y = c.f(c.lay.y[l]); a.coef = c.f(c.lay.coef_f) + 4 * (size_t)L.ch_off; a.dy_border = c.f(c.lay.dy_border); }
  a.wpack_wino = c.f(c.lay.wwd) + (size_t)L.wpack_off / 9 * 16;
  a.gin = gin; a.addend = addend;
  if (fuse) {
    a.bn_y = c.f(c.lay.y[bn_layer]); a.bn_scale = c.scale(bn_layer); a.bn_shift = c.shift(bn_layer); a.stat_partials = c.f(c.lay.partials);
    if (stat_rows) *stat_rows = grid;
  }
  a.slabs = c.f(c.lay.slab_l[l]);
  a.B = c.B; a.H = lh; a.W = lw;
  a.half = c.bf16;
  if (store_dz && !fuse) return SIFSR_ERR_ARG;
  a.store_dz = store_dz ? 1 : 0;
  a.pool_gp = pool_gp;   // gin leaves the kernel (and the border fold) multiplied by the ReLU mask of the layer below
  {
    ProfScope ps(l, 2, c.s);     // one launch = both passes of the layer: timed as its input-gradient selection
    SIFSR_TRY(launch_conv3x3_bwd16(a, c.s));
  }
  const float* wdg_f32 = c.f(c.lay.wdg) + 4 * (size_t)L.wpack_off;
  SIFSR_TRY(launch_dgrad_border_fix(dy_stored ? dy : a.dy_border, 16, wdg_f32, 16, gin, 16, 16, gin, 16, c.B, lh, lw, c.s, c.bf16,
                                    fuse ? a.bn_y : nullptr, fuse ? a.bn_scale : nullptr, fuse ? a.bn_shift : nullptr,
                                    fuse ? c.f(c.lay.bpart) : nullptr, a.store_dz));
  WgradReduceJob& j = c.jobs[(*c.njobs)++];
  j.slab_off = c.lay.slab_l[l]; j.nblk = grid; j.cin = 16; j.cout = 16; j.nbi_chunk = 1; j.w_off = L.w_off;
  if (c.side != nullptr && hipEventRecord(c.side->ev[l], c.s) != hipSuccess) return SIFSR_ERR_ARG;   // slabs of l complete
  *applied = true;
  return SIFSR_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
int sifsr_engine_forward(const float* x, float* sr, const float* params, float* running, long long* nbt, float* ws,
                         size_t ws_floats, int B, int H, int W, int training, float momentum, float eps,
                         hipStream_t s, int bf16) {
  if (!x || !sr || !params || !running || !ws) return SIFSR_ERR_ARG;
  Ctx c{sifsr_net(), WsLayout(), ws, params, B, H, W, s};
  if (bf16 < 0 || bf16 > 1) return SIFSR_ERR_ARG;
  c.bf16 = bf16;
  const HalfStorageScope storage(bf16 == 1);   // bf16 mode: every activation-like tensor of the workspace is stored as bf16
  SIFSR_TRY(sifsr_layout(B, H, W, training, &c.lay));
  // a forward touches [0, fwd_end) only; the backward regions behind it are checked by sifsr_engine_backward.  So a
  // training-mode forward that will never be followed by a backward (torch.no_grad(): BatchNorm recalibration,
  // frozen-model evaluation in train mode) runs in a forward-sized workspace, as nn.Module allows.
  if (ws_floats < c.lay.fwd_end) return SIFSR_ERR_WORKSPACE;
  const NetTable& nt = c.nt;
  const WsLayout& w = c.lay;

  SIFSR_TRY(launch_pack_weights(params, c.f(w.wfwd), c.f(w.wdg), s, c.f(w.wwf), c.f(w.wwd), training ? nbt : nullptr, SIFSR_NUM_BN_LAYERS));
  if (!training) {
    SIFSR_TRY(launch_bn_eval_coeffs(params, running, eps, c.f(w.scale), c.f(w.shift), s));
  }

  // inbloc (DoubleConvolution, model.py:596)
  {
    const LayerInfo& L = nt.L[L_IN0];
    SIFSR_TRY(launch_conv_in_fwd(x, params + L.w_off, c.f(w.y[L_IN0]), training ? c.f(w.partials) : nullptr, B, H, W, s));
    if (training)
      SIFSR_TRY(launch_bn_finalize(c.f(w.partials), conv_in_fwd_blocks(B, H, W), 16, (double)B * H * W, params + L.gamma_off,
                                   params + L.beta_off, running + L.run_off, running + L.run_off + 16, momentum, eps,
                                   c.f(w.mean) + L.ch_off, c.f(w.invstd) + L.ch_off, c.f(w.scale) + L.ch_off,
                                   c.f(w.shift) + L.ch_off, s));
  }
  SIFSR_TRY(conv_unit_fwd(c, L_IN3, src_act(c, L_IN0), src_none(), training, running, momentum, eps));

  // encoder: DownBlock_pool x3 (model.py:597-599, :528-531)
  static const int enc_prev[3] = {L_IN3, L_D1C, L_D2C};
  static const int enc_a[3] = {L_D1A, L_D2A, L_D3A}, enc_b[3] = {L_D1B, L_D2B, L_D3B}, enc_c[3] = {L_D1C, L_D2C, L_D3C};
  static const int pc[3] = {16, 32, 64};
  for (int k = 0; k < 3; ++k) {
    const int lp = enc_prev[k];
    SIFSR_TRY(launch_bnrelu_pool2(c.f(w.y[lp]), c.scale(lp), c.shift(lp), c.f(w.P[k]), B, c.lvH(k), c.lvW(k), pc[k], s));
    SIFSR_TRY(conv_unit_fwd(c, enc_a[k], src_raw(c.f(w.P[k]), pc[k]), src_none(), training, running, momentum, eps));
    SIFSR_TRY(conv_unit_fwd(c, enc_b[k], src_act(c, enc_a[k]), src_none(), training, running, momentum, eps));
    SIFSR_TRY(launch_bnrelu_add(c.f(w.P[k]), c.f(w.y[enc_b[k]]), c.scale(enc_b[k]), c.shift(enc_b[k]), c.f(w.R[k]), pc[k],
                                w.npix[k + 1], s));
    SIFSR_TRY(conv_unit_fwd(c, enc_c[k], src_raw(c.f(w.R[k]), pc[k]), src_none(), training, running, momentum, eps));
  }

  // decoder: UpBlock x3 (model.py:601-603, :235-248); cat([up, skip], 1)
  static const int dec_low[3] = {L_D3C, L_U1B, L_U2B}, dec_skip[3] = {L_D2C, L_D1C, L_IN3};
  static const int dec_a[3] = {L_U1A, L_U2A, L_U3A}, dec_b[3] = {L_U1B, L_U2B, L_U3B};
  static const int uc[3] = {64, 32, 16};
  for (int k = 0; k < 3; ++k) {
    const int lv = 2 - k;   // output level of this UpBlock
    const int ll = dec_low[k];
    SIFSR_TRY(launch_bnrelu_up2x(c.f(w.y[ll]), c.scale(ll), c.shift(ll), c.f(w.U[k]), B, c.lvH(lv + 1), c.lvW(lv + 1), uc[k], s));
    SIFSR_TRY(conv_unit_fwd(c, dec_a[k], src_raw(c.f(w.U[k]), uc[k]), src_act(c, dec_skip[k]), training, running, momentum, eps));
    SIFSR_TRY(conv_unit_fwd(c, dec_b[k], src_act(c, dec_a[k]), src_none(), training, running, momentum, eps));
  }

  // outlay (model.py:605)
  SIFSR_TRY(launch_conv_out_fwd(c.f(w.y[L_U3B]), c.scale(L_U3B), c.shift(L_U3B), params + nt.out_w_off,
                                params + nt.out_b_off, sr, B, H, W, s));
  return SIFSR_OK;
}

// ---------------------------------------------------------------------------------------------
// backward (requires the workspace of the matching training-mode forward)
// ---------------------------------------------------------------------------------------------
int sifsr_engine_backward(const float* x, const float* dsr, const float* params, float* grads, float* ws,
                          size_t ws_floats, int B, int H, int W, hipStream_t s, int bf16) {
  if (!x || !dsr || !params || !grads || !ws) return SIFSR_ERR_ARG;
  Ctx c{sifsr_net(), WsLayout(), ws, params, B, H, W, s};
  if (bf16 < 0 || bf16 > 1) return SIFSR_ERR_ARG;
  c.bf16 = bf16;
  const HalfStorageScope storage(bf16 == 1);
  SIFSR_TRY(sifsr_layout(B, H, W, 1, &c.lay));
  if (ws_floats < c.lay.total) return SIFSR_ERR_WORKSPACE;
  const NetTable& nt = c.nt;
  const WsLayout& w = c.lay;
  WgradReduceJob jobs[16];
  int njobs = 0;
  c.jobs = jobs; c.njobs = &njobs;
  c.wino_wgrads = true;
  auto finish_wgrads = [&](hipStream_t st) -> int {
    if (njobs > 0) SIFSR_TRY(launch_wgrad_reduce_batched(ws, jobs, njobs, grads, st));
    return SIFSR_OK;
  };
  // tiny problems are launch-latency-bound: the 17 event hand-offs cost more than the overlap returns (batch 1 at 256x256:
  // 1.67 ms with the second stream, 1.56 without; batch 4: 1.72 against 1.81)
  c.side = (size_t)B * H * W >= 2u * 65536u || g_side_override == 1 ? side_lane(s) : nullptr;
  SideLaneGuard lane_guard(c.side, s);
  c.forked = &lane_guard.forked;

  // SIFSR_HEAD_LINEAR=1 (A/B; off by default): the first layer's weight gradient in its linear form (edge_conv.hip: dW = sd * D +
  // k1 * (W G) + k0 * X): the Gram matrix G and the sums X of the input patches depend on the network input alone -- second stream,
  // now; D needs dz of inbloc.bloc.0, which the fused kernel of inbloc.bloc.3 stores in place of g.  One 16-channel tensor less to
  // read (the D pass: 75 us against 136 us for the fused head kernel, stand-alone), but the Gram kernel costs 93 us of vector
  // issue beside the chain's first kernels: 10,390 against 10,420 patches/s on the step (same device, interleaved) -- not adopted.
  static const int head_linear_env = getenv("SIFSR_HEAD_LINEAR") ? atoi(getenv("SIFSR_HEAD_LINEAR")) : 0;
  const bool head_linear = head_linear_env != 0 && bwd16_usable(c, L_IN3, src_act(c, L_IN0));
  if (head_linear) {
    hipStream_t gs = s;
    if (c.side != nullptr) {
      if (hipEventRecord(c.side->ev[L_IN0], s) != hipSuccess || hipStreamWaitEvent(c.side->s, c.side->ev[L_IN0], 0) != hipSuccess) return SIFSR_ERR_ARG;
      lane_guard.forked = true;
      gs = c.side->s;
    }
    SIFSR_TRY(launch_conv_in_gram(x, c.f(w.gram), B, H, W, gs));
    if (c.side != nullptr && hipEventRecord(c.side->aux, gs) != hipSuccess) return SIFSR_ERR_ARG;
  }
  static const int tail_apply_forced = getenv("SIFSR_TAIL_APPLY") ? atoi(getenv("SIFSR_TAIL_APPLY")) : 0;   // 1: keep the separate second pass (A/B)
  bool tail_in_bwd16 = false;
  // outlay backward fused with the BatchNorm+ReLU backward of ub3.convbloc.bloc.3 (fused_edges.hip): the outlay
  // input gradient is recomputed from dsr in both passes instead of being stored; dy(L_U3B) -> g[L_U3B]
  {
    const LayerInfo& L = nt.L[L_U3B];
    int nblk = B * ((H + 15) / 16) * ((W + 15) / 16);
    if (nblk > 768) nblk = 768;   // persistent: three 168-register workgroups per CU (fused_edges.hip)
    const float* y = c.f(w.y[L_U3B]);
    SIFSR_TRY(launch_tail_bwd_reduce(y, c.scale(L_U3B), c.shift(L_U3B), c.f(w.mean) + L.ch_off, c.f(w.invstd) + L.ch_off, dsr,
                                     params + nt.out_w_off, c.f(w.slabs), c.f(w.partials), nblk, B, H, W, s));
    if (grads + nt.out_b_off != grads + nt.out_w_off + 144) return SIFSR_ERR_ARG;
    // with the fused 16 -> 16 backward kernel the second pass is part of that kernel's staging (conv_bwd16.hip, mode 2)
    tail_in_bwd16 = !tail_apply_forced && bwd16_usable(c, L_U3B, src_act(c, L_U3A));
    SIFSR_TRY(launch_bn_bwd_finalize(c.f(w.partials), nblk, 16, (double)w.npix[0], c.scale(L_U3B), c.f(w.mean) + L.ch_off,
                                     c.f(w.invstd) + L.ch_off, grads + L.gamma_off, grads + L.beta_off,
                                     reinterpret_cast<double*>(c.f(w.coef)), s, c.shift(L_U3B), params + L.beta_off,
                                     c.f(w.coef_f) + 4 * (size_t)L.ch_off, c.f(w.slabs), 145, grads + nt.out_w_off));   // (+ outlay dW / db)
    if (!tail_in_bwd16)
      SIFSR_TRY(launch_tail_bwd_apply(y, c.scale(L_U3B), c.shift(L_U3B), reinterpret_cast<const double*>(c.f(w.coef)), dsr,
                                      params + nt.out_w_off, c.f(w.g[L_U3B]), B, H, W, s));
  }

  // decoder, last to first
  static const int dec_low[3] = {L_D3C, L_U1B, L_U2B}, dec_skip[3] = {L_D2C, L_D1C, L_IN3};
  static const int dec_a[3] = {L_U1A, L_U2A, L_U3A}, dec_b[3] = {L_U1B, L_U2B, L_U3B};
  static const int uc[3] = {64, 32, 16};
  // the upsample adjoint that completes g of a low-resolution layer also leaves that layer's BatchNorm-backward sums (one
  // row per workgroup): up_rows > 0 tells the next bn_unit_bwd of that layer to skip its reduce pass
  int up_rows = 0;
  int last_fused = -1;   // the last layer whose slabs the fused 16 -> 16 kernel wrote on the caller's stream (its ev[] marks them complete)
  for (int k = 2; k >= 0; --k) {
    const int lv = 2 - k;
    const int la = dec_a[k], lb = dec_b[k], ls = dec_skip[k], ll = dec_low[k];
    // second conv of the DoubleConvolution (k == 2: dy already produced by the fused tail above)
    if (k != 2) SIFSR_TRY(bn_unit_bwd(c, lb, c.f(w.g[lb]), grads, nullptr, up_rows, 0));
    int rows_a = 0;
    bool fused_b = false;
    SIFSR_TRY(conv_unit_bwd16(c, lb, src_act(c, la), k == 2 && tail_in_bwd16 ? dsr : c.f(w.g[lb]), c.f(w.g[la]), nullptr, la, &rows_a,
                              k == 2 ? (tail_in_bwd16 ? 2 : 0) : 1, &fused_b));
    if (fused_b) last_fused = lb;
    if (!fused_b) {
      SIFSR_TRY(conv_unit_wgrad(c, lb, src_act(c, la), src_none(), c.f(w.g[lb]), grads, k == 2));
      SIFSR_TRY(conv_unit_dgrad(c, lb, c.f(w.g[lb]), c.f(w.g[la]), nt.L[la].cout, nt.L[lb].cin, nullptr, 0, nullptr, la, &rows_a, k == 2));
    }
    // first conv: input = cat([U_k, relu(bn(y_skip))])
    SIFSR_TRY(bn_unit_bwd(c, la, c.f(w.g[la]), grads, nullptr, rows_a));
    SIFSR_TRY(conv_unit_wgrad(c, la, src_raw(c.f(w.U[k]), uc[k]), src_act(c, ls), c.f(w.g[la]), grads));
    SIFSR_TRY(conv_unit_dgrad(c, la, c.f(w.g[la]), c.f(w.gU[k]), uc[k], uc[k], c.f(w.g[ls]), nt.L[ls].cout, nullptr));
    up_rows = up2x_bwd_stat_rows(B, c.lvH(lv + 1), c.lvW(lv + 1), uc[k]);
    if ((size_t)up_rows * uc[k] * 2 > w.partials_cap) up_rows = 0;
    SIFSR_TRY(launch_up2x_bwd(c.f(w.gU[k]), c.f(w.g[ll]), B, c.lvH(lv + 1), c.lvW(lv + 1), uc[k], s,
                              up_rows ? c.f(w.y[ll]) : nullptr, up_rows ? c.scale(ll) : nullptr, up_rows ? c.shift(ll) : nullptr,
                              up_rows ? c.f(w.partials) : nullptr));
  }

  // encoder, last to first
  static const int enc_prev[3] = {L_IN3, L_D1C, L_D2C};
  static const int enc_a[3] = {L_D1A, L_D2A, L_D3A}, enc_b[3] = {L_D1B, L_D2B, L_D3B}, enc_c[3] = {L_D1C, L_D2C, L_D3C};
  static const int pc[3] = {16, 32, 64};
  static const int early_reduce = getenv("SIFSR_DBG_EARLY_REDUCE") ? atoi(getenv("SIFSR_DBG_EARLY_REDUCE")) : 3;   // 0: one batch at the end; 1: + one before db1; 2: + one after the decoder; 3: one per encoder stage
  for (int k = 2; k >= 0; --k) {
    const int la = enc_a[k], lb = enc_b[k], lc = enc_c[k], lp = enc_prev[k];
    if (c.side != nullptr && (early_reduce == 1 ? k == 0 : early_reduce == 2 ? (k == 0 || k == 2) : early_reduce == 3 ? true : false)) {
      // the slabs of every layer so far are reduced NOW on the second stream, between its weight gradients (one batch per encoder
      // stage), instead of in one batch at the end: there the whole reduction -- 370 MB then, HBM-bound -- ran beside the first
      // layer's weight gradient, the last kernel of the chain and HBM-bound like it, and set the end of the step
      if (last_fused >= 0) {
        if (hipStreamWaitEvent(c.side->s, c.side->ev[last_fused], 0) != hipSuccess) return SIFSR_ERR_ARG;
        lane_guard.forked = true;
      }
      SIFSR_TRY(finish_wgrads(c.side->s));
      njobs = 0;
    }
    // lastconv: input R_k = P_k + relu(bn(y_b)); its gradient is both g(a_b) and part of g(P_k).
    // y_c also feeds the next pooling stage (k < 2): that AvgPool adjoint (of gP[k+1], computed in the previous
    // iteration) is folded into this BatchNorm backward instead of a separate accumulate pass over g[lc].
    // (k == 2: g of db3.lastconv came from the last upsample adjoint above, with its sums)
    SIFSR_TRY(bn_unit_bwd(c, lc, c.f(w.g[lc]), grads, k < 2 ? c.f(w.gP[k + 1]) : nullptr, k == 2 ? up_rows : 0, 0));
    SIFSR_TRY(conv_unit_wgrad(c, lc, src_raw(c.f(w.R[k]), pc[k]), src_none(), c.f(w.g[lc]), grads));
    int rows_b = 0, rows_a = 0;
    SIFSR_TRY(conv_unit_dgrad(c, lc, c.f(w.g[lc]), c.f(w.g[lb]), pc[k], pc[k], nullptr, 0, nullptr, lb, &rows_b));
    // residual DoubleConvolution (g[lb] survives untouched: it is also the skip gradient added to gP[k] below)
    SIFSR_TRY(bn_unit_bwd(c, lb, c.f(w.g[lb]), grads, nullptr, rows_b));
    bool fused_b = false;
    SIFSR_TRY(conv_unit_bwd16(c, lb, src_act(c, la), c.f(w.g[lb]), c.f(w.g[la]), nullptr, la, &rows_a, 1, &fused_b));
    if (!fused_b) {
      SIFSR_TRY(conv_unit_wgrad(c, lb, src_act(c, la), src_none(), c.f(w.g[lb]), grads));
      SIFSR_TRY(conv_unit_dgrad(c, lb, c.f(w.g[lb]), c.f(w.g[la]), pc[k], pc[k], nullptr, 0, nullptr, la, &rows_a));
    }
    SIFSR_TRY(bn_unit_bwd(c, la, c.f(w.g[la]), grads, nullptr, rows_a));
    bool fused_a = false;
    SIFSR_TRY(conv_unit_bwd16(c, la, src_raw(c.f(w.P[k]), pc[k]), c.f(w.g[la]), c.f(w.gP[k]), c.f(w.g[lb]), -1, nullptr, 1, &fused_a));
    if (!fused_a) {
      SIFSR_TRY(conv_unit_wgrad(c, la, src_raw(c.f(w.P[k]), pc[k]), src_none(), c.f(w.g[la]), grads));
      SIFSR_TRY(conv_unit_dgrad(c, la, c.f(w.g[la]), c.f(w.gP[k]), pc[k], pc[k], nullptr, 0, c.f(w.g[lb])));
    }
    // AvgPool adjoint of gP[k] onto the skip gradient g[lp]: folded into the BatchNorm backward of lp (above / below)
    (void)lp;
  }

  // inbloc
  // inbloc.bloc.3 also feeds the first pooling stage: its gradient is g (the decoder skip) + the AvgPool adjoint of gP[0].  Where the
  // fused kernel runs it adds the adjoint while staging (a 67 MB read) and the reduction does not write the sum back (268 MB)
  static const int pool_on_load_env = getenv("SIFSR_DBG_POOL_ON_LOAD") ? atoi(getenv("SIFSR_DBG_POOL_ON_LOAD")) : 1;
  const bool pool_on_load = pool_on_load_env != 0 && bwd16_usable(c, L_IN3, src_act(c, L_IN0));
  SIFSR_TRY(bn_unit_bwd(c, L_IN3, c.f(w.g[L_IN3]), grads, c.f(w.gP[0]), 0, -1, !pool_on_load));
  int rows_in0 = 0;
  bool fused_in3 = false;
  SIFSR_TRY(conv_unit_bwd16(c, L_IN3, src_act(c, L_IN0), c.f(w.g[L_IN3]), c.f(w.g[L_IN0]), nullptr, L_IN0, &rows_in0, 1, &fused_in3, head_linear,
                            pool_on_load ? c.f(w.gP[0]) : nullptr));
  if (pool_on_load && !fused_in3) return SIFSR_ERR_ARG;
  if (head_linear && !fused_in3) return SIFSR_ERR_ARG;
  if (!fused_in3) SIFSR_TRY(conv_unit_wgrad(c, L_IN3, src_act(c, L_IN0), src_none(), c.f(w.g[L_IN3]), grads));
  // that was the last MFMA layer: all 16 layers' weight-gradient slabs -> OIHW gradients, one launch.  With the second
  // stream it follows the last weight gradient there (it writes only the conv-weight regions of `grads`, which nothing
  // on the caller's stream touches) and overlaps the head of the chain instead of trailing it.  The fused 16 -> 16 kernels
  // wrote their slabs on the CALLER's stream: the last of them (this layer's) is what the second stream waits for.
  if (c.side != nullptr) {
    if (fused_in3) {
      if (hipStreamWaitEvent(c.side->s, c.side->ev[L_IN3], 0) != hipSuccess) return SIFSR_ERR_ARG;
      lane_guard.forked = true;
    }
    SIFSR_TRY(finish_wgrads(c.side->s));
  }
  if (!fused_in3) SIFSR_TRY(conv_unit_dgrad(c, L_IN3, c.f(w.g[L_IN3]), c.f(w.g[L_IN0]), 16, 16, nullptr, 0, nullptr, L_IN0, &rows_in0));
  // first layer: no input gradient, so dy(L_IN0) is consumed by the weight gradient alone and is formed on the
  // fly from (g, y) in its staging loop; its BatchNorm-backward sums came out of the dgrad above -> finalize only
  {
    const LayerInfo& L = nt.L[L_IN0];
    const size_t npix = w.npix[0];
    const float* y = c.f(w.y[L_IN0]);
    if (rows_in0 > 0) {
      SIFSR_TRY(bn_unit_bwd(c, L_IN0, c.f(w.g[L_IN0]), grads, nullptr, rows_in0));
    } else {
      const size_t nb = npix / 256;
      const int nblk_r = (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
      SIFSR_TRY(launch_bn_bwd_reduce(c.f(w.g[L_IN0]), y, c.scale(L_IN0), c.shift(L_IN0), c.f(w.mean) + L.ch_off,
                                     c.f(w.invstd) + L.ch_off, 16, npix, c.f(w.partials), nblk_r, s));
      SIFSR_TRY(launch_bn_bwd_finalize(c.f(w.partials), nblk_r, 16, (double)npix, c.scale(L_IN0), c.f(w.mean) + L.ch_off,
                                       c.f(w.invstd) + L.ch_off, grads + L.gamma_off, grads + L.beta_off,
                                       reinterpret_cast<double*>(c.f(w.coef)), s));
    }
    int nblk = B * ((H + 15) / 16) * ((W + 15) / 16);
    if (head_linear) {      // g[L_IN0] holds dz
      if (nblk > 768) nblk = 768;
      SIFSR_TRY(launch_conv_in_dz_wgrad(x, c.f(w.g[L_IN0]), c.f(w.slabs), nblk, B, H, W, s));
      if (c.side != nullptr && hipStreamWaitEvent(s, c.side->aux, 0) != hipSuccess) return SIFSR_ERR_ARG;
      SIFSR_TRY(launch_conv_in_dw_combine(c.f(w.slabs), nblk, conv_in_gram_result(c.f(w.gram)), params + L.w_off,
                                          reinterpret_cast<const double*>(c.f(w.coef)), grads + L.w_off, s));
    } else {
      if (nblk > 1024) nblk = 1024;
      SIFSR_TRY(launch_conv_in_wgrad_fused(x, c.f(w.g[L_IN0]), y, c.scale(L_IN0), c.shift(L_IN0),
                                           reinterpret_cast<const double*>(c.f(w.coef)), c.f(w.slabs), nblk,
                                           grads + L.w_off, B, H, W, s));
    }
  }
  if (c.side != nullptr) {   // hand the second stream's work back to the caller's stream
    SIFSR_TRY(lane_guard.join());
  } else {
    SIFSR_TRY(finish_wgrads(s));
  }
  return SIFSR_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling hooks
// ---------------------------------------------------------------------------------------------
int sifsr_engine_set_wgrad_stream(int on) {
  g_side_override = on < 0 ? -1 : (on ? 1 : 0);
  return SIFSR_OK;
}

static int prof_add_locked(int layer, int phase) {
  if (g_prof.active >= PROF_SLOTS) return -1;
  ProfSlot& p = g_prof.slot[g_prof.active];
  while (p.start.size() < PROF_POOL) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) break;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); break; }
    p.start.push_back(e0); p.stop.push_back(e1);
  }
  p.used = 0; p.layer = layer; p.phase = phase;
  return g_prof.active++;
}
int sifsr_engine_profile_select(int layer, int phase) {   // layer < 0: profiling off; else exactly this one selection (slot 0)
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.active = 0;
  if (layer >= 0 && prof_add_locked(layer, phase) < 0) return SIFSR_ERR_ARG;
  return SIFSR_OK;
}
int sifsr_engine_profile_add(int layer, int phase) {      // one more selection next to the existing ones; returns its slot or < 0
  if (layer < 0) return -1;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  return prof_add_locked(layer, phase);
}
int sifsr_engine_profile_read(int slot, float* total_ms, int* count) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  float tot = 0.f; int n = 0;
  if (slot >= 0 && slot < g_prof.active) {
    ProfSlot& p = g_prof.slot[slot];
    for (size_t i = 0; i < p.used; ++i) {
      float ms = 0.f;
      if (hipEventSynchronize(p.stop[i]) == hipSuccess && hipEventElapsedTime(&ms, p.start[i], p.stop[i]) == hipSuccess) { tot += ms; ++n; }
    }
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = n;
  return SIFSR_OK;
}
