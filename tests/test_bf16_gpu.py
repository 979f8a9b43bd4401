"""BASELINE.json config 5 ("ModelB bf16 mixed-precision ... MFMA-bf16 conv tiles") as built since round 3: every activation-like
tensor inside the network is STORED as bf16 ("bf16 activations ..., bytes = half"), the operands of
the sixteen 3x3 MFMA convs are bf16, accumulation / BatchNorm statistics / master weights / optimizer are fp32.

The reference has no mixed-precision code, so there is nothing of its own to pin against; the yardsticks are
  * per kernel, EXACT arithmetic: the same bf16 inputs contracted (or pooled, interpolated, reduced) in fp32 on the CPU -- fp32
    outputs (weight gradients, statistics, the model output) to 1e-5, bf16 outputs to one rounding of the result
    (|err| <= 2^-8 |value|, checked as 4e-3 of the tensor's maximum AND a mean error of at most 2.5e-3 of the mean magnitude -- RNE to 8 bits averages ~1.4e-3);
  * end to end, the arithmetic's own noise: the oracle's emulation of this mode (oracle.BF16_CONVS + oracle.BF16_STORE) sits
    ~1e-2 from fp32; two correct implementations of it differ by individual rounding flips, so the network-level checks ask that
    the HIP result is as close to the emulation as the emulation is to fp32, and closer to fp32 than torch.autocast(bfloat16)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sif_oracle as O
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
MEAN, STD = 307.2378, 5.5698
BF = torch.bfloat16


@pytest.fixture(scope="module")
def sifsr():
    import sifsr as pkg
    assert torch.cuda.is_available()
    return pkg


@pytest.fixture()
def L(sifsr):
    """the C-ABI with the single-operator entry points switched to bf16 activation storage for the duration of one test"""
    from sifsr import _lib
    _lib.call("sifsr_set_op_storage_bf16", 1)
    yield _lib
    _lib.call("sifsr_set_op_storage_bf16", 0)


def make_model(sifsr, sd, bf16=True):
    m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1)
    m.load_state_dict(sd, strict=True)
    m.compute_dtype = "bf16" if bf16 else "fp32"
    return m.cuda()


def rb(t):
    return t.to(BF).float()


def nhwc_bf(t):
    """NCHW float tensor -> NHWC bf16 on the GPU"""
    return t.permute(0, 2, 3, 1).contiguous().to(BF).cuda()


def from_nhwc(t):
    return t.float().permute(0, 3, 1, 2).cpu()


def S():
    return torch.cuda.current_stream().cuda_stream


def assert_bf16_close(got, ref, what=""):
    """`got` (a bf16-stored result, as float) against the fp32-computed `ref`: one rounding of the result."""
    got, ref = got.double(), ref.double()
    scale = float(ref.abs().max())
    err = (got - ref).abs()
    assert float(err.max()) <= 4e-3 * scale, (what, float(err.max()), scale)
    assert float(err.mean()) <= 2.5e-3 * float(ref.abs().mean()), (what, float(err.mean()), float(ref.abs().mean()))


class _Emul:
    def __enter__(self):
        O.BF16_CONVS = True
        O.BF16_STORE = True

    def __exit__(self, *a):
        O.BF16_CONVS = False
        O.BF16_STORE = False


def test_bf16_eval_and_train_forward_backward(sifsr):
    sd = O.synthetic_state(41)
    lst, lst_up, ndvi = O.synthetic_batch(43, 2)
    x = torch.cat((lst_up, ndvi), 1)
    with _Emul():
        y_ref = O.modelb2_forward(copy.deepcopy(sd), x, training=False)
        sd_o = copy.deepcopy(sd)
        sr_o, (ds_o, pl_o, loss_o), g_o = O.forward_backward(sd_o, lst, lst_up, ndvi, MEAN, STD, 0.5, -0.25, "sr2")
    y_fp32 = O.modelb2_forward(copy.deepcopy(sd), x, training=False)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y_autocast = O.modelb2_forward(copy.deepcopy(sd), x, training=False).float()

    m = make_model(sifsr, sd).eval()
    with torch.inference_mode():
        y = m(x.cuda()).cpu()
    assert y.dtype == torch.float32                       # the model's input and output stay fp32
    e_emul, e_fp32, e_auto = rel_err(y, y_ref), rel_err(y, y_fp32), rel_err(y, y_autocast)
    e_ref = rel_err(y_ref, y_fp32)
    print(f"bf16 eval forward: vs bf16 oracle {e_emul:.2e} | vs fp32 oracle {e_fp32:.2e} | vs torch.autocast {e_auto:.2e}"
          f" | bf16 oracle vs fp32 {e_ref:.2e} | autocast vs fp32 {rel_err(y_autocast, y_fp32):.2e}")
    assert e_emul < 1.5 * e_ref + 1e-3              # within the arithmetic's own rounding-flip noise
    assert 1e-5 < e_fp32 < 2 * e_ref                # really is the bf16 path, and no worse than its emulation
    assert e_fp32 < 1.5 * rel_err(y_autocast, y_fp32)

    # training step: forward, losses, BN buffers, gradients
    m = make_model(sifsr, sd).train()
    xg = x.cuda()
    sr = m(xg)
    ds, pl, loss = sifsr.sif_loss("sr2", sr, lst.cuda(), ndvi.cuda(), MEAN, STD, 0.5, -0.25)
    loss.backward()
    assert rel_err(sr.detach().cpu(), sr_o) < 5e-2
    assert abs(float(loss.detach()) - float(loss_o)) < 1e-2 * abs(float(loss_o))
    msd = m.state_dict()
    for k in sd_o:
        if k.endswith(("running_mean", "running_var")):
            assert rel_err(msd[k].float().cpu(), sd_o[k].float()) < 1e-2, k
    # gradients: the yardstick is how far the bf16 ORACLE itself sits from the fp32 oracle (relative L2; rounding and ReLU
    # flips are sparse and large in max-norm)
    _, _, g_f = O.forward_backward(copy.deepcopy(sd), lst, lst_up, ndvi, MEAN, STD, 0.5, -0.25, "sr2")
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    worst, worst_ref = 0.0, 0.0
    for n, p in m.named_parameters():
        assert p.grad.dtype == torch.float32
        e, e_r = l2(p.grad.cpu(), g_o[n]), l2(g_o[n], g_f[n])
        worst, worst_ref = max(worst, e), max(worst_ref, e_r)
        assert e < 2.0 * e_r + 2e-2, (n, e, e_r)
    print(f"bf16 train step: worst gradient relative L2: HIP vs bf16 oracle {worst:.2e} | that oracle vs fp32 {worst_ref:.2e}")


@pytest.mark.parametrize("case", [(16, 16, 32, 48, 2), (64, 32, 16, 16, 1), (32, 64, 24, 40, 1), (128, 64, 16, 16, 1)])
def test_bf16_conv_kernels_exact_arithmetic(sifsr, case):
    """One conv -- forward (with the producing layer's folded BatchNorm + ReLU), input gradient, weight gradient -- on bf16
    tensors against the same values contracted in fp32 on the CPU."""
    from sifsr import _lib as L
    cin, cout, H, W, B = case
    rs = np.random.RandomState(sum(case))
    x = rb(torch.from_numpy(rs.standard_normal((B, cin, H, W)).astype(np.float32)))          # what the bf16 tensor holds
    w = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32))
    dy = rb(torch.from_numpy(rs.standard_normal((B, cout, H, W)).astype(np.float32)))
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, cin).astype(np.float32))
    sh = torch.from_numpy((0.3 * rs.standard_normal(cin)).astype(np.float32))
    conv = lambda a, b: F.conv2d(F.pad(a, (1, 1, 1, 1), mode="replicate"), b)
    a_in = rb(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))                          # operand after the fp32 transform, rounded
    y_ref = conv(a_in, rb(w))
    xa = x.clone().requires_grad_(True)
    (gx_ref,) = torch.autograd.grad(conv(xa, rb(w)), xa, dy)
    wa = w.clone().requires_grad_(True)
    (gw_ref,) = torch.autograd.grad(conv(x, wa), wa, dy)
    wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights", w.cuda(), cin, cout, wf, wd, S())
    y = torch.empty(B, H, W, cout, dtype=BF, device="cuda"); gx = torch.empty(B, H, W, cin, dtype=BF, device="cuda")
    nblk = L.call("sifsr_conv3x3_stat_blocks", B, H, W, cout)
    part = torch.empty(nblk, cout, 2, device="cuda")
    L.call("sifsr_conv3x3_fwd_bf16", nhwc_bf(x), cin, sc.cuda(), sh.cuda(), None, 0, None, None, wd, y, cout, part, B, H, W, S())
    L.call("sifsr_conv3x3_dgrad_bf16", nhwc_bf(dy), cout, wd, cin, gx, cin, None, 0, None, B, H, W, S())
    scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_scratch_floats", cin, cout, 4), device="cuda")
    gw = torch.empty_like(w, device="cuda")
    L.call("sifsr_conv3x3_wgrad_bf16", nhwc_bf(x), cin, None, None, None, 0, None, None, nhwc_bf(dy), cout, scratch, 4, gw, B, H, W, S())
    torch.cuda.synchronize()
    assert_bf16_close(from_nhwc(y), y_ref, "forward")
    assert_bf16_close(from_nhwc(gx), gx_ref, "input gradient")
    assert rel_err(gw.cpu(), gw_ref) < 1e-5                      # fp32 output of exact products
    # the statistics describe the STORED values
    ys = from_nhwc(y).double()
    got = part.double().sum(0).cpu()
    assert torch.allclose(got[:, 0], ys.sum((0, 2, 3)), rtol=1e-4, atol=1e-2) and torch.allclose(got[:, 1], (ys * ys).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


def test_bf16_storage_resampling_and_batchnorm_kernels(L):
    """The non-convolution kernels on bf16 tensors: AvgPool of relu(bn(y)), the residual sum, the bilinear x2 upsample and its
    adjoint (with the fused BatchNorm-backward sums), the BatchNorm-backward reduction with the pooling adjoint folded in.
    Arithmetic is fp32 on the widened values; bf16 outputs are one rounding of the fp32 result, fp32 outputs exact."""
    rs = np.random.RandomState(7)
    B, C, H, W = 2, 32, 16, 24
    y = rb(torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)))
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); sh = torch.from_numpy((0.3 * rs.standard_normal(C)).astype(np.float32))
    act = F.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    dsc, dsh = sc.cuda(), sh.cuda()
    out = torch.empty(B, H // 2, W // 2, C, dtype=BF, device="cuda")
    L.call("sifsr_bnrelu_pool2", nhwc_bf(y), dsc, dsh, out, B, H, W, C, S())
    assert_bf16_close(from_nhwc(out), F.avg_pool2d(act, 2, 2), "pool")
    p = rb(torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)))
    out = torch.empty(B, H, W, C, dtype=BF, device="cuda")
    L.call("sifsr_bnrelu_add", nhwc_bf(p), nhwc_bf(y), dsc, dsh, out, C, B * H * W, S())
    assert_bf16_close(from_nhwc(out), p + act, "residual sum")
    out = torch.empty(B, 2 * H, 2 * W, C, dtype=BF, device="cuda")
    L.call("sifsr_bnrelu_up2x", nhwc_bf(y), dsc, dsh, out, B, H, W, C, S())
    assert_bf16_close(from_nhwc(out), F.interpolate(act, scale_factor=2, mode="bilinear", align_corners=True), "upsample")
    # adjoint of the upsample + BatchNorm-backward sums of the low-resolution layer
    gu = rb(torch.from_numpy(rs.standard_normal((B, C, 2 * H, 2 * W)).astype(np.float32)))
    a = torch.zeros(B, C, H, W, requires_grad=True)
    (g_ref,) = torch.autograd.grad(F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True), a, gu)
    rows = L.call("sifsr_up2x_bwd_stat_rows", B, H, W, C)
    parts = torch.empty(rows, C, 2, device="cuda")
    g = torch.empty(B, H, W, C, dtype=BF, device="cuda")
    L.call("sifsr_up2x_bwd_bn_sums", nhwc_bf(gu), g, B, H, W, C, nhwc_bf(y), dsc, dsh, parts, S())
    torch.cuda.synchronize()
    assert_bf16_close(from_nhwc(g), g_ref, "upsample adjoint")
    gs = from_nhwc(g).double()                                        # the sums are those of the stored gradient
    dz = torch.where((y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0, gs, torch.zeros_like(gs))
    ref = torch.stack((dz.sum((0, 2, 3)), (dz * y.double()).sum((0, 2, 3))), 1)
    assert rel_err(parts.double().sum(0).cpu(), ref) < 1e-5
    # BatchNorm-backward reduction with the AvgPool adjoint folded in (completes g in place, as bf16)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); beta = torch.from_numpy((0.3 * rs.standard_normal(C)).astype(np.float32))
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    gg = rb(torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)))
    gp = rb(torch.from_numpy(rs.standard_normal((B, C, H // 2, W // 2)).astype(np.float32)))
    g_eff = rb(gg + 0.25 * gp.repeat_interleave(2, 2).repeat_interleave(2, 3))
    dgd = nhwc_bf(gg)
    npix = B * H * W
    partials = torch.empty(1024 * C * 2, device="cuda")
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    coef = torch.empty(3 * C, dtype=torch.float64, device="cuda"); coef_f = torch.empty(4 * C, device="cuda")
    L.call("sifsr_bn_relu_bwd_coef", dgd, nhwc_bf(y), scale.cuda(), shift.cuda(), mean.cuda(), invstd.cuda(), beta.cuda(), C, npix,
           partials, 3, dgam, dbet, coef, coef_f, nhwc_bf(gp), H, W, S())
    torch.cuda.synchronize()
    assert_bf16_close(from_nhwc(dgd), g_eff, "completed gradient")
    gst = from_nhwc(dgd).double()
    z = y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    dzz = torch.where(z > 0, gst, torch.zeros_like(gst))
    xhat = (y.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
    assert rel_err(dbet.cpu(), dzz.sum((0, 2, 3))) < 1e-4 and rel_err(dgam.cpu(), (dzz * xhat).sum((0, 2, 3))) < 1e-4


def test_bf16_storage_thin_convs_and_tail(L):
    """inbloc.bloc.0 forward (fp32 x -> bf16 y), outlay forward (bf16 y -> fp32 sr) and the fused tail of the backward
    (bf16 y -> fp32 dW / db / dgamma / dbeta, bf16 dy)."""
    rs = np.random.RandomState(9)
    B, H, W = 2, 32, 48
    conv = lambda a, b, bias=None: F.conv2d(F.pad(a, (1, 1, 1, 1), mode="replicate"), b, bias)
    x = torch.from_numpy(rs.standard_normal((B, 2, H, W)).astype(np.float32))
    w_in = torch.from_numpy((0.3 * rs.standard_normal((16, 2, 3, 3))).astype(np.float32))
    y = torch.empty(B, H, W, 16, dtype=BF, device="cuda")
    nblk = L.call("sifsr_conv_in_stat_blocks", B, H, W)
    part = torch.empty(nblk, 16, 2, device="cuda")
    L.call("sifsr_conv_in_fwd", x.cuda(), w_in.cuda(), y, part, B, H, W, S())
    torch.cuda.synchronize()
    assert_bf16_close(from_nhwc(y), conv(x, w_in), "conv_in")
    ys = from_nhwc(y).double()
    assert torch.allclose(part.double().sum(0).cpu()[:, 0], ys.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    # outlay forward
    yraw = rb(torch.from_numpy(rs.standard_normal((B, 16, H, W)).astype(np.float32)))
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 16).astype(np.float32)); beta = torch.from_numpy((0.3 * rs.standard_normal(16)).astype(np.float32))
    mean, var = yraw.mean((0, 2, 3)), yraw.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    w_out = torch.from_numpy((0.2 * rs.standard_normal((1, 16, 3, 3))).astype(np.float32)); b_out = torch.tensor([0.1])
    act = F.relu(yraw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    sr = torch.empty(B, 1, H, W, device="cuda")
    L.call("sifsr_conv_out_fwd", nhwc_bf(yraw), scale.cuda(), shift.cuda(), w_out.cuda(), b_out.cuda(), sr, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(sr.cpu(), conv(act, w_out, b_out)) < 1e-5
    # fused tail: float64 autograd of conv(relu(bn(y))) on the bf16-valued y
    dsr = torch.from_numpy(rs.standard_normal((B, 1, H, W)).astype(np.float32))
    y64 = yraw.double().requires_grad_(True)
    ga64, be64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    w64, b64 = w_out.double().requires_grad_(True), b_out.double().requires_grad_(True)
    out64 = conv(F.relu(F.batch_norm(y64, None, None, ga64, be64, training=True, eps=1e-5)), w64, b64)
    dy_ref, dga_ref, dbe_ref, dw_ref, db_ref = torch.autograd.grad((out64 * dsr.double()).sum(), [y64, ga64, be64, w64, b64])
    nb = 8
    scratch = torch.empty(2 * ((nb * 145 + 63) // 64 * 64) + nb * 64, device="cuda")
    dwb = torch.empty(145, device="cuda"); dgam, dbet = torch.empty(16, device="cuda"), torch.empty(16, device="cuda")
    coef = torch.empty(48, dtype=torch.float64, device="cuda")
    dy = torch.empty(B, H, W, 16, dtype=BF, device="cuda")
    L.call("sifsr_conv_out_bn_relu_bwd", nhwc_bf(yraw), scale.cuda(), shift.cuda(), mean.cuda(), invstd.cuda(), dsr.cuda(), w_out.cuda(),
           scratch, nb, dwb, dgam, dbet, coef, dy, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(dwb[:144].cpu().view(1, 16, 3, 3), dw_ref) < 1e-4 and abs(float(dwb[144]) - float(db_ref)) < 1e-4 * abs(float(db_ref))
    assert rel_err(dgam.cpu(), dga_ref) < 1e-4 and rel_err(dbet.cpu(), dbe_ref) < 1e-4
    assert_bf16_close(from_nhwc(dy), dy_ref, "tail dy")


def test_bf16_train_steps_run_and_decrease_loss(sifsr):
    torch.manual_seed(0)
    m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).cuda()
    m.compute_dtype = "bf16"
    opt = sifsr.FlatAdam(m.parameters(), lr=1e-3)
    stats = dict(sifsr.dataset.DEFAULT_STATS)
    lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(8, torch.device("cuda"), seed=3)
    losses = [float(sifsr.train.train_step(m, opt, lst, lst_up, ndvi, stats, 0.5, -0.25, "sr2")[2].detach()) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_bf16_other_sizes_and_determinism(sifsr):
    """Partial tiles at every level (48x80 patches) and run-to-run bitwise determinism of a bf16 training step."""
    torch.manual_seed(1)
    m = sifsr.ModelB_2(2).cuda()
    m.compute_dtype = "bf16"
    x = torch.randn(3, 2, 48, 80, device="cuda")
    lst = torch.randn(3, 1, 12, 20, device="cuda"); ndvi = x[:, 1:2].contiguous()
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        m.train()
        sr = m(x)
        _, _, loss = sifsr.sif_loss("sr2", sr, lst, ndvi, MEAN, STD, 0.5, -0.25)
        loss.backward()
        grads.append(m.flat_grad().clone())
    assert torch.isfinite(grads[0]).all() and torch.equal(grads[0], grads[1])
    # against the fp32 mode on the same weights: the mode's own noise level, nothing structural
    m.compute_dtype = "fp32"
    m.zero_grad(set_to_none=True)
    sr32 = m(x)
    _, _, l32 = sifsr.sif_loss("sr2", sr32, lst, ndvi, MEAN, STD, 0.5, -0.25)
    l32.backward()
    g32 = m.flat_grad()
    assert rel_err(sr.detach(), sr32.detach()) < 8e-2
    assert float((grads[0] - g32).norm() / g32.norm()) < 0.5      # (small random-init patches: ~0.25)


@pytest.mark.parametrize("case", [(32, 48, 2, True, True), (64, 64, 3, True, False), (128, 128, 4, False, True)])
def test_bf16_storage_fused_backward_of_16_channel_layers(L, case):
    """sifsr_conv3x3_bwd16 on bf16 tensors (available to the bf16 mode, SIFSR_BF16_BWD16=1; not its default -- the separate bf16-MFMA
    kernels are faster once the bytes are halved): input gradient (bf16 out,
    incl. the border fold) and weight gradient (fp32 out) of z = relu(bn(conv(a))), a = relu(x*xs + xsh), against float64
    autograd on the values the bf16 tensors hold; the BatchNorm sums of the layer below are those of the stored gradient."""
    H, W, B, dyf, with_stats = case
    C = 16
    rs = np.random.RandomState(sum(case[:3]))
    x = rb(torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)))
    xs = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); xsh = torch.from_numpy((0.3 * rs.standard_normal(C)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32))
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); beta = torch.from_numpy((0.5 * rs.standard_normal(C)).astype(np.float32))
    g = rb(torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)))
    conv = lambda a, b: F.conv2d(F.pad(a, (1, 1, 1, 1), mode="replicate"), b)
    a64 = F.relu(x.double() * xs.double().view(1, -1, 1, 1) + xsh.double().view(1, -1, 1, 1)).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    # the layer's own raw output as the bf16 tensor the forward would have stored, and its batch statistics
    y = rb(conv(a64.detach().float(), w))
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    if dyf:
        # dL/dy from (g, y) with the coefficients the statistics pass leaves (float64 reference of the same formula)
        z = y.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        dz = torch.where(z > 0, g.double(), torch.zeros_like(z))
        n = B * H * W
        xhat = (y.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
        dbeta, dgamma = dz.sum((0, 2, 3)), (dz * xhat).sum((0, 2, 3))
        dy64 = scale.double().view(1, -1, 1, 1) * (dz - dbeta.view(1, -1, 1, 1) / n - xhat * dgamma.view(1, -1, 1, 1) / n)
    else:
        dy64 = g.double()
    ga_ref, gw_ref = torch.autograd.grad(conv(a64, w64), [a64, w64], dy64)
    S_ = S()
    wf = torch.empty(9 * C * C, device="cuda"); wd = torch.empty(4 * 9 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights", w.cuda(), C, C, wf, wd, S_)
    wwf = torch.empty(16 * C * C, device="cuda"); wwd = torch.empty(16 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", w.cuda(), C, C, wwf, wwd, S_)
    dx, dg, dyy = nhwc_bf(x), nhwc_bf(g), nhwc_bf(y)
    coef_f = None
    if dyf:
        partials = torch.empty(1024 * C * 2, device="cuda")
        dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        coef = torch.empty(3 * C, dtype=torch.float64, device="cuda"); coef_f = torch.empty(4 * C, device="cuda")
        L.call("sifsr_bn_relu_bwd_coef", dg, dyy, scale.cuda(), shift.cuda(), mean.cuda(), invstd.cuda(), beta.cuda(), C, B * H * W,
               partials, 4, dgam, dbet, coef, coef_f, None, H, W, S_)
    rows = L.call("sifsr_conv3x3_bwd16_stat_rows", B, H, W)
    scratch = torch.empty(L.call("sifsr_conv3x3_bwd16_scratch_floats", B, H, W), device="cuda")
    gin = torch.empty(B, H, W, C, dtype=BF, device="cuda"); dw = torch.empty(C, C, 3, 3, device="cuda")
    border = torch.zeros(B, H, W, C, dtype=BF, device="cuda")
    bnp = torch.empty(rows, C, 2, device="cuda") if with_stats else None
    dxs, dxsh = xs.cuda(), xsh.cuda()                          # (the fused sums require bn_y / scale / shift to BE x / x_scale / x_shift)
    L.call("sifsr_conv3x3_bwd16", dx, dxs, dxsh, dg if dyf else nhwc_bf(dy64.float()), dyy if dyf else None, coef_f,
           border if dyf else None, wd, wwd, gin, None, dx if with_stats else None, dxs if with_stats else None,
           dxsh if with_stats else None, bnp, scratch, dw, B, H, W, S_)
    torch.cuda.synchronize()
    dy_used = dy64 if dyf else rb(dy64.float()).double()      # stored dL/dy is itself a bf16 tensor
    ga_ref2, gw_ref2 = torch.autograd.grad(conv(a64, w64), [a64, w64], dy_used) if not dyf else (ga_ref, gw_ref)
    got = from_nhwc(gin).double()
    # the border fold contracts bf16-rounded operands (as the bf16 input-gradient kernel): one more rounding on the image border
    assert float((got - ga_ref2).abs().max()) <= 1.2e-2 * float(ga_ref2.abs().max())
    assert float((got - ga_ref2).abs().mean()) <= 2.5e-3 * float(ga_ref2.abs().mean())
    assert rel_err(dw.cpu(), gw_ref2) < (2e-3 if dyf else 1e-5)     # dyf: dL/dy is formed from bf16 (g, y) with fp32 coefficients
    if with_stats:
        zpos = (x.double() * xs.double().view(1, -1, 1, 1) + xsh.double().view(1, -1, 1, 1)) > 0
        dzs = torch.where(zpos, got, torch.zeros_like(got))
        ref = torch.stack((dzs.sum((0, 2, 3)), (dzs * x.double()).sum((0, 2, 3))), 1)
        assert rel_err(bnp.double().sum(0).cpu(), ref) < 2e-3          # (border fold deltas are summed unrounded)
