// extern "C" surface of libsifsr_hip.so -- see include/sifsr_hip.h for the contract.
#include "../../include/sifsr_hip.h"

#include "engine.h"

#define S(stream) ((hipStream_t)(stream))

static ConvSrc mk_src(const float* p, int C, const float* sc, const float* sh) {
  ConvSrc s; s.ptr = p; s.scale = p ? sc : nullptr; s.shift = p ? sh : nullptr; s.C = p ? C : 0; s.coff = 0; s.nq = p ? C / 16 : 0;
  return s;
}

int sifsr_abi_version(void) { return 3; }   // 2: Winograd-domain entry points, fused BatchNorm-backward forms (round 2); 3: sifsr_conv3x3_bwd16(_tail), split-bf16 entry points removed (round 3)
int sifsr_num_params(void) { return sifsr_net().total_params; }
int sifsr_num_running(void) { return sifsr_net().total_running; }
int sifsr_layer_table(int* out, int capacity_rows) {
  const NetTable& nt = sifsr_net();
  for (int l = 0; l < SIFSR_NUM_BN_LAYERS && l < capacity_rows; ++l) {
    const LayerInfo& L = nt.L[l];
    int* r = out + 8 * l;
    r[0] = L.cin; r[1] = L.cout; r[2] = L.level; r[3] = L.w_off; r[4] = L.gamma_off; r[5] = L.beta_off; r[6] = L.run_off; r[7] = L.ch_off;
  }
  return SIFSR_NUM_BN_LAYERS;
}

size_t sifsr_model_workspace_bytes(int B, int H, int W, int training) {
  WsLayout l;
  if (sifsr_layout(B, H, W, training, &l) != SIFSR_OK) return 0;
  return l.total * sizeof(float);
}
int sifsr_model_workspace_regions(int B, int H, int W, size_t* out, int capacity) {
  WsLayout l;
  if (sifsr_layout(B, H, W, 1, &l) != SIFSR_OK) return 0;
  size_t v[56]; int n = 0;
  for (int i = 0; i < 17; ++i) v[n++] = l.y[i];
  for (int i = 0; i < 3; ++i) v[n++] = l.P[i];
  for (int i = 0; i < 3; ++i) v[n++] = l.R[i];
  for (int i = 0; i < 3; ++i) v[n++] = l.U[i];
  for (int i = 0; i < 17; ++i) v[n++] = l.g[i];
  for (int i = 0; i < 3; ++i) v[n++] = l.dyB[i];
  for (int i = 0; i < 3; ++i) v[n++] = l.gP[i];
  for (int i = 0; i < 3; ++i) v[n++] = l.gU[i];
  v[n++] = l.mean; v[n++] = l.invstd; v[n++] = l.scale; v[n++] = l.shift;
  for (int i = 0; i < n && i < capacity; ++i) out[i] = v[i];
  return n;
}
int sifsr_model_forward(const float* x, float* sr, const float* params, float* running, long long* nbt, void* workspace,
                        size_t workspace_bytes, int B, int H, int W, int training, float momentum, float eps, void* stream) {
  return sifsr_engine_forward(x, sr, params, running, nbt, (float*)workspace, workspace_bytes / sizeof(float), B, H, W,
                              training, momentum, eps, S(stream));
}
int sifsr_model_backward(const float* x, const float* dsr, const float* params, float* grads, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, void* stream) {
  return sifsr_engine_backward(x, dsr, params, grads, (float*)workspace, workspace_bytes / sizeof(float), B, H, W, S(stream));
}

int sifsr_model_forward_ex(const float* x, float* sr, const float* params, float* running, long long* nbt, void* workspace,
                           size_t workspace_bytes, int B, int H, int W, int training, float momentum, float eps, int compute,
                           void* stream) {
  if (compute < 0 || compute > 2) return SIFSR_ERR_ARG;
  return sifsr_engine_forward(x, sr, params, running, nbt, (float*)workspace, workspace_bytes / sizeof(float), B, H, W,
                              training, momentum, eps, S(stream), compute);
}
int sifsr_model_backward_ex(const float* x, const float* dsr, const float* params, float* grads, void* workspace,
                            size_t workspace_bytes, int B, int H, int W, int compute, void* stream) {
  if (compute < 0 || compute > 2) return SIFSR_ERR_ARG;
  return sifsr_engine_backward(x, dsr, params, grads, (float*)workspace, workspace_bytes / sizeof(float), B, H, W, S(stream),
                               compute);
}

int launch_pack_weights_one(const float* w, int cin, int cout, float* wfwd, float* wdg, hipStream_t s, float* wwf = nullptr, float* wwd = nullptr);
int sifsr_pack_conv_weights(const float* w_oihw, int cin, int cout, float* wfwd, float* wdgrad, void* stream) {
  if (cin % 16 || cout % 16) return SIFSR_ERR_SHAPE;
  return launch_pack_weights_one(w_oihw, cin, cout, wfwd, wdgrad, S(stream));
}

int sifsr_pack_conv_weights_wino(const float* w_oihw, int cin, int cout, float* wwf, float* wwd, void* stream) {
  if (cin % 16 || cout % 16 || !w_oihw || !wwf || !wwd) return SIFSR_ERR_SHAPE;
  return launch_pack_weights_one(w_oihw, cin, cout, nullptr, nullptr, S(stream), wwf, wwd);
}
int sifsr_conv3x3_stat_blocks(int B, int H, int W, int cout) { return conv3x3_grid_blocks(B, H, W, cout); }
int sifsr_conv3x3_stat_blocks_wino(int B, int H, int W, int cin, int cout) {
  ConvArgs a; a.B = B; a.H = H; a.W = W; a.wpack_wino = reinterpret_cast<const float*>(1);   // shape decision only
  a.NQ = cin / 16;
  return conv3x3_grid_blocks(B, H, W, cout, conv3x3_wino_kind(a, cout, 0));
}

int sifsr_conv3x3_fwd(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                      const float* scale1, const float* shift1, const float* wfwd, float* y, int cout,
                      float* stat_partials, int B, int H, int W, void* stream) {
  if (!src0 || C0 % 16 || (src1 && C1 % 16)) return SIFSR_ERR_SHAPE;
  ConvArgs a;
  a.src[0] = mk_src(src0, C0, scale0, shift0);
  a.src[1] = mk_src(src1, C1, scale1, shift1);
  a.dst[0].ptr = y; a.dst[0].C = cout; a.dst[0].coff = 0; a.dst[1] = a.dst[0];
  a.wpack = wfwd; a.addend = nullptr; a.addC = 0; a.stat_partials = stat_partials; a.dst_split = cout / 16;
  a.B = B; a.H = H; a.W = W; a.NQ = a.src[0].nq + a.src[1].nq;
  return launch_conv3x3_mfma(a, cout, 0, S(stream));
}

int sifsr_conv3x3_fwd_wino(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                           const float* scale1, const float* shift1, const float* wfwd, const float* wwf, float* y, int cout,
                           float* stat_partials, int B, int H, int W, void* stream) {
  if (!src0 || !wfwd || !wwf || C0 % 16 || (src1 && C1 % 16)) return SIFSR_ERR_SHAPE;
  ConvArgs a;
  a.src[0] = mk_src(src0, C0, scale0, shift0);
  a.src[1] = mk_src(src1, C1, scale1, shift1);
  a.dst[0].ptr = y; a.dst[0].C = cout; a.dst[0].coff = 0; a.dst[1] = a.dst[0];
  a.wpack = wfwd; a.wpack_wino = wwf; a.addend = nullptr; a.addC = 0; a.stat_partials = stat_partials; a.dst_split = cout / 16;
  a.B = B; a.H = H; a.W = W; a.NQ = a.src[0].nq + a.src[1].nq;
  return launch_conv3x3_mfma(a, cout, 0, S(stream));
}

int sifsr_conv3x3_dgrad_wino(const float* dy, int cout, const float* wdgrad, const float* wwd, int cin, float* g0, int C0,
                             float* g1, int C1, const float* addend, int B, int H, int W, void* stream) {
  if (cout % 16 || cin % 16 || C0 % 16 || (g1 && (C1 % 16 || C0 + C1 != cin)) || (!g1 && C0 != cin) || (addend && g1))
    return SIFSR_ERR_SHAPE;
  if (!wwd) return SIFSR_ERR_ARG;
  ConvArgs a;
  a.src[0] = mk_src(dy, cout, nullptr, nullptr); a.src[1] = mk_src(nullptr, 0, nullptr, nullptr);
  a.dst[0].ptr = g0; a.dst[0].C = C0; a.dst[0].coff = 0;
  a.dst[1].ptr = g1 ? g1 : g0; a.dst[1].C = g1 ? C1 : C0; a.dst[1].coff = 0;
  a.wpack = wdgrad; a.wpack_wino = wwd; a.addend = addend; a.addC = cin; a.stat_partials = nullptr; a.dst_split = C0 / 16;
  a.B = B; a.H = H; a.W = W; a.NQ = cout / 16;
  int rc = launch_conv3x3_mfma(a, cin, 1, S(stream));
  if (rc) return rc;
  return launch_dgrad_border_fix(dy, cout, wdgrad, cin, g0, C0, C0, g1 ? g1 : g0, g1 ? C1 : C0, B, H, W, S(stream));
}

int sifsr_conv3x3_dgrad(const float* dy, int cout, const float* wdgrad, const float* w_oihw, int cin, float* g0, int C0,
                        float* g1, int C1, const float* addend, int B, int H, int W, void* stream) {
  if (cout % 16 || cin % 16 || C0 % 16 || (g1 && (C1 % 16 || C0 + C1 != cin)) || (!g1 && C0 != cin) || (addend && g1))
    return SIFSR_ERR_SHAPE;
  ConvArgs a;
  a.src[0] = mk_src(dy, cout, nullptr, nullptr); a.src[1] = mk_src(nullptr, 0, nullptr, nullptr);
  a.dst[0].ptr = g0; a.dst[0].C = C0; a.dst[0].coff = 0;
  a.dst[1].ptr = g1 ? g1 : g0; a.dst[1].C = g1 ? C1 : C0; a.dst[1].coff = 0;
  a.wpack = wdgrad; a.addend = addend; a.addC = cin; a.stat_partials = nullptr; a.dst_split = C0 / 16;
  a.B = B; a.H = H; a.W = W; a.NQ = cout / 16;
  (void)w_oihw;   // kept in the signature for ABI stability; the border fold uses the tap-major pack
  int rc = launch_conv3x3_mfma(a, cin, 1, S(stream));
  if (rc) return rc;
  return launch_dgrad_border_fix(dy, cout, wdgrad, cin, g0, C0, C0, g1 ? g1 : g0, g1 ? C1 : C0, B, H, W, S(stream));
}

// The same input gradient with the layer's BatchNorm+ReLU backward applied while staging: g = dL/d relu(bn(y)), y, coef_f
// (sifsr_bn_relu_bwd_coef) -> dL/dy is never stored, except on the image border (`border`, cout-channel NHWC indexing).
int sifsr_conv3x3_dgrad_fused(const float* g, const float* y, const float* coef_f, int cout, const float* wdgrad,
                              const float* wwd, int cin, float* g0, int C0, float* g1, int C1, const float* addend, float* border,
                              int B, int H, int W, void* stream) {
  if (cout % 16 || cin % 16 || C0 % 16 || (g1 && (C1 % 16 || C0 + C1 != cin)) || (!g1 && C0 != cin) || (addend && g1))
    return SIFSR_ERR_SHAPE;
  if (!g || !y || !coef_f || !border) return SIFSR_ERR_ARG;
  ConvArgs a;
  a.src[0] = mk_src(g, cout, nullptr, nullptr); a.src[1] = mk_src(nullptr, 0, nullptr, nullptr);
  a.bw_y = y; a.bw_coef = coef_f; a.bw_border = border;
  a.dst[0].ptr = g0; a.dst[0].C = C0; a.dst[0].coff = 0;
  a.dst[1].ptr = g1 ? g1 : g0; a.dst[1].C = g1 ? C1 : C0; a.dst[1].coff = 0;
  a.wpack = wdgrad; a.wpack_wino = wwd; a.addend = addend; a.addC = cin; a.stat_partials = nullptr; a.dst_split = C0 / 16;
  a.B = B; a.H = H; a.W = W; a.NQ = cout / 16;
  int rc = launch_conv3x3_mfma(a, cin, 1, S(stream));
  if (rc) return rc;
  return launch_dgrad_border_fix(border, cout, wdgrad, cin, g0, C0, C0, g1 ? g1 : g0, g1 ? C1 : C0, B, H, W, S(stream));
}

// bf16-operand forms (config 5): both bf16 packs live in the second half of the `wdgrad` buffer written by
// sifsr_pack_conv_weights ([fp32 dgrad pack n | fwd bf16 n/2 | dgrad bf16 n/2 | unused 2n], n = 9*cin*cout floats)
static int conv3x3_fwd_lowp(int mode, const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                            const float* scale1, const float* shift1, const float* wdgrad, float* y, int cout,
                            float* stat_partials, int B, int H, int W, void* stream) {
  if (!src0 || !wdgrad || C0 % 16 || (src1 && C1 % 16)) return SIFSR_ERR_SHAPE;
  ConvArgs a;
  a.src[0] = mk_src(src0, C0, scale0, shift0);
  a.src[1] = mk_src(src1, C1, scale1, shift1);
  a.dst[0].ptr = y; a.dst[0].C = cout; a.dst[0].coff = 0; a.dst[1] = a.dst[0];
  a.NQ = a.src[0].nq + a.src[1].nq;
  const size_t n = (size_t)9 * (16 * a.NQ) * cout;
  a.wpack = wdgrad + n; a.bf16 = mode;
  a.addend = nullptr; a.addC = 0; a.stat_partials = stat_partials; a.dst_split = cout / 16;
  a.B = B; a.H = H; a.W = W;
  return launch_conv3x3_mfma(a, cout, 0, S(stream));
}
int sifsr_conv3x3_fwd_bf16(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                           const float* scale1, const float* shift1, const float* wdgrad, float* y, int cout,
                           float* stat_partials, int B, int H, int W, void* stream) {
  return conv3x3_fwd_lowp(1, src0, C0, scale0, shift0, src1, C1, scale1, shift1, wdgrad, y, cout, stat_partials, B, H, W, stream);
}
static int conv3x3_dgrad_lowp(int mode, const float* dy, int cout, const float* wdgrad, int cin, float* g0, int C0, float* g1, int C1,
                              const float* addend, int B, int H, int W, void* stream) {
  if (cout % 16 || cin % 16 || C0 % 16 || (g1 && (C1 % 16 || C0 + C1 != cin)) || (!g1 && C0 != cin) || (addend && g1))
    return SIFSR_ERR_SHAPE;
  ConvArgs a;
  a.src[0] = mk_src(dy, cout, nullptr, nullptr); a.src[1] = mk_src(nullptr, 0, nullptr, nullptr);
  a.dst[0].ptr = g0; a.dst[0].C = C0; a.dst[0].coff = 0;
  a.dst[1].ptr = g1 ? g1 : g0; a.dst[1].C = g1 ? C1 : C0; a.dst[1].coff = 0;
  const size_t n = (size_t)9 * cin * cout;
  a.wpack = wdgrad + n + n / 2; a.bf16 = mode;
  a.addend = addend; a.addC = cin; a.stat_partials = nullptr; a.dst_split = C0 / 16;
  a.B = B; a.H = H; a.W = W; a.NQ = cout / 16;
  int rc = launch_conv3x3_mfma(a, cin, 1, S(stream));
  if (rc) return rc;
  // border fold: rounds its operands like the main kernel
  return launch_dgrad_border_fix(dy, cout, wdgrad, cin, g0, C0, C0, g1 ? g1 : g0, g1 ? C1 : C0, B, H, W, S(stream), mode);
}
int sifsr_conv3x3_dgrad_bf16(const float* dy, int cout, const float* wdgrad, int cin, float* g0, int C0, float* g1, int C1,
                             const float* addend, int B, int H, int W, void* stream) {
  return conv3x3_dgrad_lowp(1, dy, cout, wdgrad, cin, g0, C0, g1, C1, addend, B, H, W, stream);
}
size_t sifsr_conv3x3_wgrad_scratch_floats(int cin, int cout, int nblk) { return (size_t)nblk * wgrad_slab_floats(cin, cout); }

int sifsr_conv3x3_wgrad(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                        const float* scale1, const float* shift1, const float* dy, int cout, float* scratch, int nblk,
                        float* dw, int B, int H, int W, void* stream) {
  if (!src0 || C0 % 16 || (src1 && C1 % 16)) return SIFSR_ERR_SHAPE;
  WgradArgs a;
  a.src[0] = mk_src(src0, C0, scale0, shift0);
  a.src[1] = mk_src(src1, C1, scale1, shift1);
  a.dy = dy; a.slabs = scratch; a.B = B; a.H = H; a.W = W;
  a.NQ = a.src[0].nq + a.src[1].nq;
  a.ntiles = B * ((H + 7) / 8) * ((W + 15) / 16);
  const int cin = 16 * a.NQ;
  if (nblk > a.ntiles) nblk = a.ntiles;
  int rc = launch_conv3x3_wgrad(a, cin, cout, nblk, S(stream));
  if (rc) return rc;
  return launch_wgrad_reduce(scratch, nblk, cin, cout, wgrad_nbi_chunk(a, cin), dw, S(stream));
}

int sifsr_conv3x3_wgrad_fused(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                              const float* scale1, const float* shift1, const float* g, const float* y, const float* coef_f,
                              int cout, float* scratch, int nblk, float* dw, int B, int H, int W, void* stream) {
  if (!src0 || C0 % 16 || (src1 && C1 % 16)) return SIFSR_ERR_SHAPE;
  if (!g || !y || !coef_f) return SIFSR_ERR_ARG;
  WgradArgs a;
  a.src[0] = mk_src(src0, C0, scale0, shift0);
  a.src[1] = mk_src(src1, C1, scale1, shift1);
  a.dy = g; a.dy_y = y; a.dy_coef = coef_f; a.slabs = scratch; a.B = B; a.H = H; a.W = W;
  a.NQ = a.src[0].nq + a.src[1].nq;
  a.ntiles = B * ((H + 7) / 8) * ((W + 15) / 16);
  const int cin = 16 * a.NQ;
  if (nblk > a.ntiles) nblk = a.ntiles;
  int rc = launch_conv3x3_wgrad(a, cin, cout, nblk, S(stream));
  if (rc) return rc;
  return launch_wgrad_reduce(scratch, nblk, cin, cout, wgrad_nbi_chunk(a, cin), dw, S(stream));
}

// Winograd F(3x3, 2x2) form.  scratch = nblk slabs of 9*cin*cout floats (the kernel applies the output transform before it writes them)
size_t sifsr_conv3x3_wgrad_wino_scratch_floats(int cin, int cout, int nblk) { return (size_t)nblk * 9 * cin * cout; }

int sifsr_conv3x3_wgrad_wino(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                             const float* scale1, const float* shift1, const float* g, const float* y, const float* coef_f,
                             int cout, float* scratch, int nblk, float* dw, int B, int H, int W, void* stream) {
  if (!src0 || C0 % 16 || (src1 && C1 % 16)) return SIFSR_ERR_SHAPE;
  if (!g || ((y != nullptr) != (coef_f != nullptr))) return SIFSR_ERR_ARG;
  WgradArgs a;
  a.src[0] = mk_src(src0, C0, scale0, shift0);
  a.src[1] = mk_src(src1, C1, scale1, shift1);
  a.dy = g; a.dy_y = y; a.dy_coef = coef_f; a.slabs = scratch; a.B = B; a.H = H; a.W = W;
  a.NQ = a.src[0].nq + a.src[1].nq;
  a.ntiles = B * ((H + 7) / 8) * ((W + 15) / 16);
  const int cin = 16 * a.NQ;
  if (nblk > a.ntiles) nblk = a.ntiles;
  if (!conv3x3_wgrad_use_wino(a, cin, cout)) return SIFSR_ERR_SHAPE;
  int rc = launch_conv3x3_wgrad_wino(a, cin, cout, nblk, S(stream));
  if (rc) return rc;
  return launch_wgrad_reduce(scratch, nblk, cin, cout, wgrad_wino_nbi_chunk(a, cin), dw, S(stream));
}

// Input gradient AND weight gradient of a 16 -> 16 channel layer from one read of its operands (conv_bwd16.hip).
int sifsr_conv3x3_bwd16_stat_rows(int B, int H, int W) {
  return conv3x3_bwd16_applies(B, H, W) ? conv3x3_bwd16_grid(B, H, W) + dgrad_border_waves(B, H, W, 16) : 0;
}
size_t sifsr_conv3x3_bwd16_scratch_floats(int B, int H, int W) {
  return conv3x3_bwd16_applies(B, H, W) ? (size_t)conv3x3_bwd16_grid(B, H, W) * 9 * 256 : 0;
}
static int bwd16_op(Bwd16Args a, const float* dy_edge, const float* wdgrad, float* bn_partials, float* scratch, float* dw, void* stream) {
  const int B = a.B, H = a.H, W = a.W;
  const int grid = conv3x3_bwd16_grid(B, H, W);
  a.slabs = scratch;
  a.half = sifsr_half_storage() ? 1 : 0;      // sifsr_set_op_storage_bf16(1): x, g, y, border, gin, addend, bn_y are bf16 tensors
  int rc = launch_conv3x3_bwd16(a, S(stream));
  if (rc) return rc;
  // (bf16 storage: the fold reads / updates bf16 tensors but keeps fp32 products, like the main kernel -- flag 2)
  rc = launch_dgrad_border_fix(dy_edge, 16, wdgrad, 16, a.gin, 16, 16, a.gin, 16, B, H, W, S(stream), a.half,
                               bn_partials ? a.bn_y : nullptr, bn_partials ? a.bn_scale : nullptr, bn_partials ? a.bn_shift : nullptr,
                               bn_partials ? bn_partials + (size_t)grid * 32 : nullptr);
  if (rc) return rc;
  return launch_wgrad_reduce(scratch, grid, 16, 16, 1, dw, S(stream));
}
int sifsr_conv3x3_bwd16(const float* x, const float* x_scale, const float* x_shift, const float* g, const float* y,
                        const float* coef_f, float* border, const float* wdgrad, const float* wwd, float* gin,
                        const float* addend, const float* bn_y, const float* bn_scale, const float* bn_shift,
                        float* bn_partials, float* scratch, float* dw, int B, int H, int W, void* stream) {
  if (!conv3x3_bwd16_applies(B, H, W)) return SIFSR_ERR_SHAPE;
  if (!x || !g || !wdgrad || !wwd || !gin || !scratch || !dw) return SIFSR_ERR_ARG;
  if ((y != nullptr) != (coef_f != nullptr) || (y != nullptr && !border)) return SIFSR_ERR_ARG;
  if (bn_partials != nullptr && (!bn_y || !bn_scale || !bn_shift || addend != nullptr)) return SIFSR_ERR_ARG;
  Bwd16Args a;
  a.x = x; a.x_scale = x_scale; a.x_shift = x_shift; a.g = g; a.y = y; a.coef = coef_f; a.dy_border = y ? border : nullptr;
  a.wpack_wino = wwd; a.gin = gin; a.addend = addend;
  if (bn_partials != nullptr) { a.bn_y = bn_y; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.stat_partials = bn_partials; }
  a.B = B; a.H = H; a.W = W;
  return bwd16_op(a, y ? border : g, wdgrad, bn_partials, scratch, dw, stream);
}
// ... of a layer that also feeds a pooling stage (inbloc.bloc.3): the upstream gradient is g + 0.25 * pool_gp[b][y/2][x/2] (the AvgPool2d(2,2)
// adjoint of the half-resolution gradient pool_gp, 16 channels), added while staging; otherwise sifsr_conv3x3_bwd16 with y != NULL, no addend
int sifsr_conv3x3_bwd16_pool(const float* x, const float* x_scale, const float* x_shift, const float* g, const float* pool_gp,
                             const float* y, const float* coef_f, float* border, const float* wdgrad, const float* wwd, float* gin,
                             const float* bn_y, const float* bn_scale, const float* bn_shift, float* bn_partials, float* scratch,
                             float* dw, int B, int H, int W, void* stream) {
  if (!conv3x3_bwd16_applies(B, H, W)) return SIFSR_ERR_SHAPE;
  if (!x || !g || !pool_gp || !y || !coef_f || !border || !wdgrad || !wwd || !gin || !scratch || !dw) return SIFSR_ERR_ARG;
  if (bn_partials != nullptr && (!bn_y || !bn_scale || !bn_shift)) return SIFSR_ERR_ARG;
  Bwd16Args a;
  a.x = x; a.x_scale = x_scale; a.x_shift = x_shift; a.g = g; a.pool_gp = pool_gp; a.y = y; a.coef = coef_f; a.dy_border = border;
  a.wpack_wino = wwd; a.gin = gin;
  if (bn_partials != nullptr) { a.bn_y = bn_y; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.stat_partials = bn_partials; }
  a.B = B; a.H = H; a.W = W;
  return bwd16_op(a, border, wdgrad, bn_partials, scratch, dw, stream);
}
// ... of the LAST 16 -> 16 layer (ub3.convbloc.bloc.3): the upstream gradient g is the input gradient of outlay (Conv2d 16 -> 1,
// replicate padding, model.py:605) and is recomputed while staging from dsr = d loss / d sr [B][H][W] (fp32) and w_out [1][16][3][3]
int sifsr_conv3x3_bwd16_tail(const float* x, const float* x_scale, const float* x_shift, const float* dsr, const float* w_out,
                             const float* y, const float* coef_f, float* border, const float* wdgrad, const float* wwd, float* gin,
                             const float* bn_y, const float* bn_scale, const float* bn_shift, float* bn_partials, float* scratch,
                             float* dw, int B, int H, int W, void* stream) {
  if (!conv3x3_bwd16_applies(B, H, W)) return SIFSR_ERR_SHAPE;
  if (!x || !dsr || !w_out || !y || !coef_f || !border || !wdgrad || !wwd || !gin || !scratch || !dw) return SIFSR_ERR_ARG;
  if (bn_partials != nullptr && (!bn_y || !bn_scale || !bn_shift)) return SIFSR_ERR_ARG;
  Bwd16Args a;
  a.x = x; a.x_scale = x_scale; a.x_shift = x_shift; a.tail_dsr = dsr; a.tail_w = w_out; a.y = y; a.coef = coef_f; a.dy_border = border;
  a.wpack_wino = wwd; a.gin = gin;
  if (bn_partials != nullptr) { a.bn_y = bn_y; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.stat_partials = bn_partials; }
  a.B = B; a.H = H; a.W = W;
  return bwd16_op(a, border, wdgrad, bn_partials, scratch, dw, stream);
}

// bf16-operand form (config 5): x and dy rounded to bf16 when read from LDS, fp32 accumulation
int sifsr_conv3x3_wgrad_bf16(const float* src0, int C0, const float* scale0, const float* shift0, const float* src1, int C1,
                        const float* scale1, const float* shift1, const float* dy, int cout, float* scratch, int nblk,
                        float* dw, int B, int H, int W, void* stream) {
  if (!src0 || C0 % 16 || (src1 && C1 % 16)) return SIFSR_ERR_SHAPE;
  WgradArgs a;
  a.src[0] = mk_src(src0, C0, scale0, shift0);
  a.src[1] = mk_src(src1, C1, scale1, shift1);
  a.dy = dy; a.slabs = scratch; a.B = B; a.H = H; a.W = W;
  a.NQ = a.src[0].nq + a.src[1].nq;
  a.ntiles = B * ((H + 7) / 8) * ((W + 15) / 16);
  const int cin = 16 * a.NQ;
  a.bf16 = 1;
  if (nblk > a.ntiles) nblk = a.ntiles;
  int rc = launch_conv3x3_wgrad(a, cin, cout, nblk, S(stream));
  if (rc) return rc;
  return launch_wgrad_reduce(scratch, nblk, cin, cout, wgrad_nbi_chunk(a, cin), dw, S(stream));
}

// Activation storage of the SINGLE-OPERATOR entry points that are not convolutions (BatchNorm reductions, pooling / upsampling
// and their adjoints, the thin convs, the fused tail): 1 = their activation tensors are bf16 (what the model's bf16 mode
// runs), 0 = fp32 (default).  Per calling thread; the model-level calls set it themselves from their compute mode.
static thread_local HalfStorageScope* t_op_storage = nullptr;
int sifsr_set_op_storage_bf16(int on) {
  if (t_op_storage != nullptr) { delete t_op_storage; t_op_storage = nullptr; }
  if (on) t_op_storage = new HalfStorageScope(true);
  return SIFSR_OK;
}
int sifsr_conv_in_stat_blocks(int B, int H, int W) { return conv_in_fwd_blocks(B, H, W); }
int sifsr_conv_in_fwd(const float* x, const float* w, float* y, float* stat_partials, int B, int H, int W, void* stream) {
  return launch_conv_in_fwd(x, w, y, stat_partials, B, H, W, S(stream));
}
int sifsr_conv_in_wgrad(const float* x, const float* dy, float* scratch, int nblk, float* dw, int B, int H, int W, void* stream) {
  return launch_conv_in_wgrad(x, dy, scratch, nblk, dw, B, H, W, S(stream));
}
int sifsr_conv_out_fwd(const float* y, const float* scale, const float* shift, const float* w, const float* bias, float* sr,
                       int B, int H, int W, void* stream) {
  return launch_conv_out_fwd(y, scale, shift, w, bias, sr, B, H, W, S(stream));
}
int sifsr_conv_out_dgrad(const float* dsr, const float* w, float* g, int B, int H, int W, void* stream) {
  return launch_conv_out_dgrad(dsr, w, g, B, H, W, S(stream));
}
int sifsr_conv_out_wgrad(const float* y, const float* scale, const float* shift, const float* dsr, float* scratch, int nblk,
                         float* dwb, int B, int H, int W, void* stream) {
  return launch_conv_out_wgrad(y, scale, shift, dsr, scratch, nblk, dwb, dwb + 144, B, H, W, S(stream));
}

int sifsr_conv_out_bn_relu_bwd(const float* y, const float* scale, const float* shift, const float* mean,
                               const float* invstd, const float* dsr, const float* w, float* scratch, int nblk,
                               float* dwb, float* dgamma, float* dbeta, double* coef, float* dy, int B, int H, int W,
                               void* stream) {
  if (!y || !scale || !shift || !mean || !invstd || !dsr || !w || !scratch || !dwb || !dgamma || !dbeta || !coef || !dy)
    return SIFSR_ERR_ARG;
  float* bnpart = scratch + (((size_t)nblk * 145 + 63) & ~(size_t)63);
  int rc = launch_tail_bwd_reduce(y, scale, shift, mean, invstd, dsr, w, scratch, bnpart, nblk, B, H, W, S(stream));
  if (rc) return rc;
  rc = launch_sum_partials(scratch, nblk, 145, dwb, S(stream));
  if (rc) return rc;
  rc = launch_bn_bwd_finalize(bnpart, nblk, 16, (double)B * H * W, scale, mean, invstd, dgamma, dbeta, coef, S(stream));
  if (rc) return rc;
  return launch_tail_bwd_apply(y, scale, shift, coef, dsr, w, dy, B, H, W, S(stream));
}
// the same weight gradient from dz = g * [y*scale+shift > 0] and the BatchNorm-backward coefficients alone (edge_conv.hip, "head"):
// dW = sd * D + k1 * (W G) + k0 * X with D = sum dz p^T, G = sum p p^T, X = sum p over the input patches p
size_t sifsr_conv_in_bwd_linear_scratch_floats(int nblk) { return conv_in_gram_scratch_floats() + (size_t)(nblk > 0 ? nblk : 0) * 288; }
int sifsr_conv_in_bwd_linear(const float* x, const float* dz, const float* w, const double* coef, float* scratch, int nblk,
                             float* dw, int B, int H, int W, void* stream) {
  if (!x || !dz || !w || !coef || !scratch || !dw || nblk < 1) return SIFSR_ERR_ARG;
  float* part = scratch + conv_in_gram_scratch_floats();
  int rc = launch_conv_in_gram(x, scratch, B, H, W, S(stream));
  if (rc) return rc;
  rc = launch_conv_in_dz_wgrad(x, dz, part, nblk, B, H, W, S(stream));
  if (rc) return rc;
  return launch_conv_in_dw_combine(part, nblk, conv_in_gram_result(scratch), w, coef, dw, S(stream));
}
int sifsr_conv_in_bn_relu_bwd(const float* x, const float* g, const float* y, const float* scale, const float* shift,
                              const float* mean, const float* invstd, float* scratch, int nblk, float* dw, float* dgamma,
                              float* dbeta, double* coef, int B, int H, int W, void* stream) {
  if (!x || !g || !y || !scale || !shift || !mean || !invstd || !scratch || !dw || !dgamma || !dbeta || !coef)
    return SIFSR_ERR_ARG;
  if (nblk < 1 || nblk > 1024) return SIFSR_ERR_ARG;
  const size_t npix = (size_t)B * H * W;
  int rc = launch_bn_bwd_reduce(g, y, scale, shift, mean, invstd, 16, npix, scratch, nblk, S(stream));
  if (rc) return rc;
  rc = launch_bn_bwd_finalize(scratch, nblk, 16, (double)npix, scale, mean, invstd, dgamma, dbeta, coef, S(stream));
  if (rc) return rc;
  return launch_conv_in_wgrad_fused(x, g, y, scale, shift, coef, scratch, nblk, dw, B, H, W, S(stream));
}

int sifsr_bn_finalize(const float* stat_partials, int nblk, int C, double count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                      float* scale, float* shift, void* stream) {
  return launch_bn_finalize(stat_partials, nblk, C, count, gamma, beta, running_mean, running_var, momentum, eps, mean,
                            invstd, scale, shift, S(stream));
}
int sifsr_bn_relu_bwd(const float* g, const float* y, const float* scale, const float* shift, const float* mean,
                      const float* invstd, int C, size_t npix, float* partials, int nblk, float* dgamma, float* dbeta,
                      double* coef, float* dy, const float* gpool, int H, int W, void* stream) {
  int rc = launch_bn_bwd_reduce(g, y, scale, shift, mean, invstd, C, npix, partials, nblk, S(stream), gpool, H, W);
  if (rc) return rc;
  rc = launch_bn_bwd_finalize(partials, nblk, C, (double)npix, scale, mean, invstd, dgamma, dbeta, coef, S(stream));
  if (rc) return rc;
  return launch_bn_bwd_apply(g, y, scale, shift, coef, C, npix, dy, S(stream), gpool, H, W);
}

int sifsr_bn_relu_bwd_coef(float* g, const float* y, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* beta, int C, size_t npix, float* partials, int nblk, float* dgamma,
                           float* dbeta, double* coef, float* coef_f, const float* gpool, int H, int W, void* stream) {
  if (!g || !y || !beta || !coef || !coef_f) return SIFSR_ERR_ARG;
  int rc = launch_bn_bwd_reduce(g, y, scale, shift, mean, invstd, C, npix, partials, nblk, S(stream), gpool, H, W, gpool ? g : nullptr);
  if (rc) return rc;
  return launch_bn_bwd_finalize(partials, nblk, C, (double)npix, scale, mean, invstd, dgamma, dbeta, coef, S(stream), shift, beta, coef_f);
}

int sifsr_bnrelu_pool2(const float* y, const float* scale, const float* shift, float* out, int B, int H, int W, int C, void* stream) {
  return launch_bnrelu_pool2(y, scale, shift, out, B, H, W, C, S(stream));
}
int sifsr_bnrelu_add(const float* p, const float* y, const float* scale, const float* shift, float* out, int C, size_t npix, void* stream) {
  return launch_bnrelu_add(p, y, scale, shift, out, C, npix, S(stream));
}
int sifsr_bnrelu_up2x(const float* y, const float* scale, const float* shift, float* out, int B, int Hin, int Win, int C, void* stream) {
  return launch_bnrelu_up2x(y, scale, shift, out, B, Hin, Win, C, S(stream));
}
int sifsr_pool2_bwd(const float* gp, float* g, int B, int H, int W, int C, int accumulate, void* stream) {
  return launch_pool2_bwd(gp, g, B, H, W, C, accumulate, S(stream));
}
int sifsr_up2x_bwd_stat_rows(int B, int Hin, int Win, int C) { return up2x_bwd_stat_rows(B, Hin, Win, C); }
int sifsr_up2x_bwd_bn_sums(const float* gu, float* g, int B, int Hin, int Win, int C, const float* y, const float* scale,
                           const float* shift, float* partials, void* stream) {
  if (!y || !scale || !shift || !partials) return SIFSR_ERR_ARG;
  return launch_up2x_bwd(gu, g, B, Hin, Win, C, S(stream), y, scale, shift, partials);
}
int sifsr_up2x_bwd(const float* gu, float* g, int B, int Hin, int Win, int C, void* stream) {
  return launch_up2x_bwd(gu, g, B, Hin, Win, C, S(stream));
}

int sifsr_gauss9_reflect_fwd(const float* x, const float* taps9, float* out, int B, int H, int W, void* stream) {
  return launch_blur_fwd(x, taps9, out, B, H, W, S(stream));
}
int sifsr_gauss9_reflect_bwd(const float* g, const float* taps9, float* gx, int B, int H, int W, void* stream) {
  return launch_blur_bwd(g, taps9, gx, B, H, W, S(stream));
}
int sifsr_gauss9_decimate4_fwd(const float* x, const float* taps9, float* out_lr, int B, int H, int W, void* stream) {
  return launch_blurdec_fwd(x, taps9, out_lr, B, H, W, S(stream));
}
int sifsr_gauss9_decimate4_bwd(const float* g_lr, const float* taps9, float* gx, int B, int H, int W, void* stream) {
  return launch_blurdec_bwd(g_lr, taps9, gx, B, H, W, S(stream));
}
int sifsr_sobel4_fwd(const float* x, float* out_b4hw, int B, int H, int W, void* stream) { return launch_sobel_fwd(x, out_b4hw, B, H, W, S(stream)); }
int sifsr_sobel4_bwd(const float* g_b4hw, float* gx, int B, int H, int W, void* stream) { return launch_sobel_bwd(g_b4hw, gx, B, H, W, S(stream)); }
int sifsr_huber_partial_blocks(size_t n) { return huber_partial_blocks(n); }
int sifsr_huber_fwd(const float* a, const float* b, float bscale, size_t n, float* partials, float* out, void* stream) {
  return launch_huber_fwd(a, b, bscale, n, partials, out, S(stream));
}
int sifsr_huber_bwd(const float* a, const float* b, float bscale, const float* gout, size_t n, float* ga, void* stream) {
  return launch_huber_bwd(a, b, bscale, gout, n, ga, S(stream));
}
size_t sifsr_sif_loss_workspace_bytes(int kind, int B, int H, int W) { return sif_loss_workspace_floats(kind, B, H, W) * sizeof(float); }
int sifsr_sif_loss(int kind, const float* sr, const float* lst, const float* ndvi, int B, int H, int W, float mean, float std,
                   float alpha, float gamma, const float* taps_ds9, const float* taps_ftm9, void* workspace,
                   size_t workspace_bytes, float* losses3, float* dsr, void* stream) {
  if (workspace_bytes < sifsr_sif_loss_workspace_bytes(kind, B, H, W)) return SIFSR_ERR_WORKSPACE;
  return launch_sif_loss(kind, sr, lst, ndvi, B, H, W, mean, std, alpha, gamma, taps_ds9, taps_ftm9, (float*)workspace,
                         losses3, dsr, S(stream));
}

int sifsr_adam_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  return launch_adam_flat(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, S(stream));
}

int sifsr_set_wgrad_stream(int on) { return sifsr_engine_set_wgrad_stream(on); }
int sifsr_profile_select(int layer, int phase) { return sifsr_engine_profile_select(layer, phase); }
int sifsr_profile_read(float* total_ms, int* count) { return sifsr_engine_profile_read(0, total_ms, count); }
int sifsr_profile_add(int layer, int phase) { return sifsr_engine_profile_add(layer, phase); }
int sifsr_profile_read_slot(int slot, float* total_ms, int* count) { return sifsr_engine_profile_read(slot, total_ms, count); }

// ---- input pipeline / metrics ----
int sifsr_tiles_prepare(const float* lst, const float* ndvi, float* x, int tiles_y, int tiles_x, int win, int lst_h, int lst_w,
                        int granule, float mean_lst, float std_lst, float mean_ndvi, float std_ndvi, int clip_ndvi,
                        void* stream) {
  if (!lst || !ndvi || !x || tiles_y < 1 || tiles_x < 1) return SIFSR_ERR_ARG;
  const int hr = 4 * win;
  if (granule) {
    // tiles cut out of one raster: LST (lst_h, lst_w) row-major, NDVI (4 lst_h, 4 lst_w)
    if (tiles_y * win > lst_h || tiles_x * win > lst_w) return SIFSR_ERR_SHAPE;
    return launch_tiles_prepare(lst, ndvi, x, tiles_y * tiles_x, tiles_x, win, (long long)win * lst_w, win, lst_w,
                                (long long)hr * 4 * lst_w, hr, 4 * lst_w, mean_lst, std_lst, mean_ndvi, std_ndvi, clip_ndvi,
                                S(stream));
  }
  // a batch of separate tiles: lst (T,1,win,win), ndvi (T,1,hr,hr)
  return launch_tiles_prepare(lst, ndvi, x, tiles_y * tiles_x, 1, win, (long long)win * win, 0, win, (long long)hr * hr, 0, hr,
                              mean_lst, std_lst, mean_ndvi, std_ndvi, clip_ndvi, S(stream));
}
int sifsr_tiles_paste(const float* sr, float* out, int tiles_y, int tiles_x, int win, int lst_w, float mean_lst, float std_lst,
                      void* stream) {
  if (!sr || !out || tiles_x * win > lst_w) return SIFSR_ERR_ARG;
  return launch_tiles_paste(sr, out, tiles_y * tiles_x, tiles_x, 4 * win, (long long)4 * lst_w, mean_lst, std_lst, S(stream));
}
size_t sifsr_psnr_ssim_scratch_bytes(int B, int H, int W) { return psnr_ssim_scratch_bytes(B, H, W); }
int sifsr_psnr_ssim(const float* pred, const float* targ, int B, int H, int W, void* scratch, size_t scratch_bytes, float* out2,
                    void* stream) {
  if (!pred || !targ || !scratch || !out2) return SIFSR_ERR_ARG;
  if (scratch_bytes < psnr_ssim_scratch_bytes(B, H, W)) return SIFSR_ERR_WORKSPACE;
  return launch_psnr_ssim(pred, targ, B, H, W, scratch, out2, S(stream));
}

// ---- Fourier-domain evaluation ----
size_t sifsr_fft2_attenuation_scratch_bytes(int B, int H, int W) { return fourier_scratch_bytes(B, H, W); }
int sifsr_fft2_attenuation(const float* img, int B, int H, int W, void* scratch, size_t scratch_bytes, float* mag,
                           float* spectrum, void* stream) {
  if (!img || !scratch || (!mag && !spectrum)) return SIFSR_ERR_ARG;
  if (scratch_bytes < fourier_scratch_bytes(B, H, W)) return SIFSR_ERR_WORKSPACE;
  return launch_fft2_attenuation(img, B, H, W, scratch, mag, spectrum, S(stream));
}

// ---- scale-invariance baseline: the 'norm-L4' decimation of us.downsampling ----
int sifsr_l4pool4(const float* x, float* out, int B, int H, int W, void* stream) {
  if (!x || !out) return SIFSR_ERR_ARG;
  return launch_l4pool4(x, out, B, H, W, S(stream));
}

// hipGraph-capturable Adam: the step counter (int64) and the two bias-correction coefficients live in device memory
int sifsr_adam_flat_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, long long* step_dev, float* coef2, float grad_scale,
                        void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq) return SIFSR_ERR_ARG;
  return launch_adam_flat_dev(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_dev, coef2,
                              grad_scale, S(stream));
}
