"""GPU parity: every C-ABI operator (include/sifsr_hip.h) against plain PyTorch fp32 CPU math of the
same op (the oracle's building blocks).  Tolerances are max|a-b| / max|b|; the north-star bar is
1e-4 relative, fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star: within 1e-4 relative fp32


@pytest.fixture(scope="module")
def L():
    import sifsr
    from sifsr import _lib
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return _lib


def dev(t):
    return t.contiguous().cuda()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def S():
    return torch.cuda.current_stream().cuda_stream


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


def conv_rep(x, w, b=None):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), w, b)


# (C0, C1, cout, H, W, B, affine0, affine1)
CONV_CASES = [
    (16, 0, 16, 32, 48, 2, True, False),
    (16, 0, 32, 16, 16, 1, False, False),
    (32, 0, 32, 32, 32, 2, True, False),
    (32, 0, 64, 16, 32, 1, True, False),
    (64, 0, 64, 32, 32, 1, True, False),
    (64, 64, 64, 32, 32, 1, False, True),    # ub1.convbloc.bloc.0: cat([up, skip])
    (32, 32, 32, 32, 32, 1, False, True),    # ub2
    (16, 16, 16, 32, 32, 2, False, True),    # ub3
    (64, 0, 32, 16, 16, 2, True, False),
    (32, 0, 16, 32, 16, 1, True, False),
    # partial tiles: H / W not multiples of 16 (deeper levels of 64x64 ... 48x80 patches)
    (64, 0, 64, 8, 8, 3, True, False),
    (64, 64, 64, 12, 20, 2, False, True),
    (32, 0, 32, 24, 40, 2, True, False),
    (16, 0, 32, 20, 36, 1, True, False),
    (64, 0, 64, 4, 6, 2, True, False),
    (16, 16, 16, 12, 20, 2, True, True),     # a 32-channel Cin chunk across the two sources, partial tiles
]


def _mk_inputs(rs, C0, C1, B, H, W, aff0, aff1):
    x0 = rnd(rs, B, C0, H, W)
    x1 = rnd(rs, B, C1, H, W) if C1 else None
    sc0 = torch.from_numpy(rs.uniform(0.5, 1.5, C0).astype(np.float32)) if aff0 else None
    sh0 = rnd(rs, C0, scale=0.3) if aff0 else None
    sc1 = torch.from_numpy(rs.uniform(0.5, 1.5, C1).astype(np.float32)) if (C1 and aff1) else None
    sh1 = rnd(rs, C1, scale=0.3) if (C1 and aff1) else None

    def act(x, sc, sh):
        return F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]) if sc is not None else x
    a = act(x0, sc0, sh0)
    if C1:
        a = torch.cat([a, act(x1, sc1, sh1)], 1)
    return x0, x1, sc0, sh0, sc1, sh1, a


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_fwd_dgrad_wgrad(L, case):
    C0, C1, cout, H, W, B, aff0, aff1 = case
    cin = C0 + C1
    rs = np.random.RandomState(hash(case) % 2**31)
    x0, x1, sc0, sh0, sc1, sh1, a = _mk_inputs(rs, C0, C1, B, H, W, aff0, aff1)
    w = rnd(rs, cout, cin, 3, 3, scale=(2.0 / (9 * cin)) ** 0.5)
    a = a.requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = conv_rep(a, wr)
    dy = rnd(rs, B, cout, H, W)
    ga_ref, gw_ref = torch.autograd.grad((y_ref * dy).sum(), [a, wr])

    dw_, d0 = dev(w), dev(nhwc(x0))
    d1 = dev(nhwc(x1)) if C1 else None
    dsc0, dsh0 = (dev(sc0), dev(sh0)) if aff0 else (None, None)
    dsc1, dsh1 = (dev(sc1), dev(sh1)) if sc1 is not None else (None, None)
    wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights", dw_, cin, cout, wf, wd, S())

    # ---- forward + BN statistic partials ----
    y = torch.empty(B, H, W, cout, device="cuda")
    nblk = L.call("sifsr_conv3x3_stat_blocks", B, H, W, cout)
    part = torch.empty(nblk, cout, 2, device="cuda")
    L.call("sifsr_conv3x3_fwd", d0, C0, dsc0, dsh0, d1, C1, dsc1, dsh1, wf, y, cout, part, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(y.cpu()), y_ref) < TOL
    ps = part.cpu().double().sum(0)
    yr = y_ref.detach().double()
    assert torch.allclose(ps[:, 0], yr.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(ps[:, 1], (yr * yr).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)

    # ---- dgrad (incl. replicate border fold), split destinations for the concat case ----
    ddy = dev(nhwc(dy))
    g0 = torch.full((B, H, W, C0), float("nan"), device="cuda")
    g1 = torch.full((B, H, W, C1), float("nan"), device="cuda") if C1 else None
    L.call("sifsr_conv3x3_dgrad", ddy, cout, wd, dw_, cin, g0, C0, g1, C1, None, B, H, W, S())
    torch.cuda.synchronize()
    g = nchw(g0.cpu()) if not C1 else torch.cat([nchw(g0.cpu()), nchw(g1.cpu())], 1)
    assert rel_err(g, ga_ref) < TOL
    add = rnd(rs, B, cin, H, W)
    if not C1:   # residual addend variant
        g2 = torch.empty(B, H, W, cin, device="cuda")
        L.call("sifsr_conv3x3_dgrad", ddy, cout, wd, dw_, cin, g2, cin, None, 0, dev(nhwc(add)), B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(nchw(g2.cpu()), ga_ref + add) < TOL

    # ---- the same two passes in the Winograd F(2x2,3x3) domain (what the model runs; odd sizes fall back to the taps)
    wwf = torch.empty(16 * cin * cout, device="cuda"); wwd = torch.empty(16 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", dw_, cin, cout, wwf, wwd, S())
    yw = torch.full((B, H, W, cout), float("nan"), device="cuda")
    nblk_w = L.call("sifsr_conv3x3_stat_blocks_wino", B, H, W, cin, cout)
    part_w = torch.empty(nblk_w, cout, 2, device="cuda")
    L.call("sifsr_conv3x3_fwd_wino", d0, C0, dsc0, dsh0, d1, C1, dsc1, dsh1, wf, wwf, yw, cout, part_w, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(yw.cpu()), y_ref) < TOL
    ps = part_w.cpu().double().sum(0)
    assert torch.allclose(ps[:, 0], yr.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(ps[:, 1], (yr * yr).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    g0 = torch.full((B, H, W, C0), float("nan"), device="cuda")
    g1 = torch.full((B, H, W, C1), float("nan"), device="cuda") if C1 else None
    L.call("sifsr_conv3x3_dgrad_wino", ddy, cout, wd, wwd, cin, g0, C0, g1, C1, None, B, H, W, S())
    torch.cuda.synchronize()
    g = nchw(g0.cpu()) if not C1 else torch.cat([nchw(g0.cpu()), nchw(g1.cpu())], 1)
    assert rel_err(g, ga_ref) < TOL
    if not C1:
        g2 = torch.full((B, H, W, cin), float("nan"), device="cuda")
        L.call("sifsr_conv3x3_dgrad_wino", ddy, cout, wd, wwd, cin, g2, cin, None, 0, dev(nhwc(add)), B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(nchw(g2.cpu()), ga_ref + add) < TOL

    # ---- wgrad ----
    for nb in (1, 7, 64):
        scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_scratch_floats", cin, cout, nb), device="cuda")
        dwo = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
        L.call("sifsr_conv3x3_wgrad", d0, C0, dsc0, dsh0, d1, C1, dsc1, dsh1, ddy, cout, scratch, nb, dwo, B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(dwo.cpu(), gw_ref) < TOL, nb
        if H % 2 == 0 and W % 2 == 0:                           # Winograd F(3x3, 2x2) form
            scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_wino_scratch_floats", cin, cout, nb), device="cuda")
            dwx = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
            L.call("sifsr_conv3x3_wgrad_wino", d0, C0, dsc0, dsh0, d1, C1, dsc1, dsh1, ddy, None, None, cout, scratch, nb, dwx, B, H, W, S())
            torch.cuda.synchronize()
            assert rel_err(dwx.cpu(), gw_ref) < TOL, nb


@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 24, 40), (2, 40, 24)])   # full and partial 16x16 tiles
def test_conv_in(L, shape):
    rs = np.random.RandomState(1)
    B, H, W = shape
    x = rnd(rs, B, 2, H, W)
    w = rnd(rs, 16, 2, 3, 3, scale=0.3).requires_grad_(True)
    y_ref = conv_rep(x, w)
    dy = rnd(rs, B, 16, H, W)
    (gw_ref,) = torch.autograd.grad((y_ref * dy).sum(), [w])
    y = torch.empty(B, H, W, 16, device="cuda")
    nblk = L.call("sifsr_conv_in_stat_blocks", B, H, W)
    assert nblk == min(2048, B * ((H + 15) // 16) * ((W + 15) // 16))
    part = torch.empty(nblk, 16, 2, device="cuda")
    L.call("sifsr_conv_in_fwd", dev(x), dev(w.detach()), y, part, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(y.cpu()), y_ref) < TOL
    assert torch.allclose(part.cpu().double().sum(0)[:, 0], y_ref.detach().double().sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(part.cpu().double().sum(0)[:, 1], (y_ref.detach().double() ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    scratch = torch.empty(8 * 288, device="cuda")
    dw = torch.empty(16, 2, 3, 3, device="cuda")
    L.call("sifsr_conv_in_wgrad", dev(x), dev(nhwc(dy)), scratch, 5, dw, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), gw_ref) < TOL


@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 24, 40), (2, 40, 24)])
def test_conv_out(L, shape):
    rs = np.random.RandomState(2)
    B, H, W = shape
    yraw = rnd(rs, B, 16, H, W)
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, 16).astype(np.float32)); sh = rnd(rs, 16, scale=0.3)
    a = F.relu(yraw * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    w = rnd(rs, 1, 16, 3, 3, scale=0.2).requires_grad_(True)
    b = rnd(rs, 1).requires_grad_(True)
    out_ref = conv_rep(a, w, b)
    dsr = rnd(rs, B, 1, H, W)
    ga_ref, gw_ref, gb_ref = torch.autograd.grad((out_ref * dsr).sum(), [a, w, b])
    out = torch.empty(B, 1, H, W, device="cuda")
    dy_, dsc, dsh, dw_, db_ = dev(nhwc(yraw)), dev(sc), dev(sh), dev(w.detach()), dev(b.detach())
    L.call("sifsr_conv_out_fwd", dy_, dsc, dsh, dw_, db_, out, B, H, W, S())
    g = torch.empty(B, H, W, 16, device="cuda")
    L.call("sifsr_conv_out_dgrad", dev(dsr), dw_, g, B, H, W, S())
    scratch = torch.empty(8 * 145, device="cuda")
    dwb = torch.empty(145, device="cuda")
    L.call("sifsr_conv_out_wgrad", dy_, dsc, dsh, dev(dsr), scratch, 5, dwb, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), out_ref) < TOL
    assert rel_err(nchw(g.cpu()), ga_ref) < TOL
    assert rel_err(dwb.cpu()[:144].view(1, 16, 3, 3), gw_ref) < TOL
    assert rel_err(dwb.cpu()[144:], gb_ref) < TOL


@pytest.mark.parametrize("C", [16, 32, 64])
def test_batchnorm_fwd_bwd(L, C):
    rs = np.random.RandomState(3 + C)
    B, H, W = 2, 32, 32
    y = (rnd(rs, B, C, H, W) * 1.7 + 0.4).requires_grad_(True)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)).requires_grad_(True)
    beta = rnd(rs, C, scale=0.2).requires_grad_(True)
    rm, rv = rnd(rs, C, scale=0.1), torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32))
    rm_ref, rv_ref = rm.clone(), rv.clone()
    a_ref = F.relu(F.batch_norm(y, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    g = rnd(rs, B, C, H, W)
    gy_ref, gg_ref, gb_ref = torch.autograd.grad((a_ref * g).sum(), [y, gamma, beta])

    yd = dev(nhwc(y.detach()))
    # statistics partials as a conv epilogue would produce them (per 16x16 tile)
    t = y.detach().unfold(2, 16, 16).unfold(3, 16, 16)            # B,C,ty,tx,16,16
    p1 = t.sum((-1, -2)).permute(0, 2, 3, 1).reshape(-1, C)
    p2 = (t * t).sum((-1, -2)).permute(0, 2, 3, 1).reshape(-1, C)
    part = dev(torch.stack([p1, p2], -1))
    nblk = part.shape[0]
    drm, drv = dev(rm), dev(rv)
    mean, invstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
    L.call("sifsr_bn_finalize", part, nblk, C, float(B * H * W), dev(gamma.detach()), dev(beta.detach()), drm, drv,
           0.1, 1e-5, mean, invstd, scale, shift, S())
    torch.cuda.synchronize()
    assert rel_err(drm.cpu(), rm_ref) < 1e-5 and rel_err(drv.cpu(), rv_ref) < 1e-5
    a = F.relu(y.detach() * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None])
    assert rel_err(a, a_ref) < TOL

    npix = B * H * W
    nb = 8
    partials = torch.empty(nb * C * 2, device="cuda")
    dgam, dbet = (torch.empty(C, device="cuda") for _ in range(2))
    coef = torch.empty(3 * C, dtype=torch.float64, device="cuda")
    dy = torch.empty(B, H, W, C, device="cuda")
    L.call("sifsr_bn_relu_bwd", dev(nhwc(g)), yd, scale, shift, mean, invstd, C, npix, partials, nb, dgam, dbet, coef, dy,
           None, 0, 0, S())
    torch.cuda.synchronize()
    assert rel_err(dgam.cpu(), gg_ref) < TOL and rel_err(dbet.cpu(), gb_ref) < TOL
    assert rel_err(nchw(dy.cpu()), gy_ref) < TOL

    # the activation also feeds AvgPool2d(2,2): the pooled tensor's gradient is folded in on the fly
    a2 = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    gp = rnd(rs, B, C, H // 2, W // 2)
    gy2, gg2, gb2 = torch.autograd.grad((a2 * g).sum() + (F.avg_pool2d(a2, 2, 2) * gp).sum(), [y, gamma, beta])
    L.call("sifsr_bn_relu_bwd", dev(nhwc(g)), yd, scale, shift, mean, invstd, C, npix, partials, nb, dgam, dbet, coef, dy,
           dev(nhwc(gp)), H, W, S())
    torch.cuda.synchronize()
    assert rel_err(dgam.cpu(), gg2) < TOL and rel_err(dbet.cpu(), gb2) < TOL
    assert rel_err(nchw(dy.cpu()), gy2) < TOL


@pytest.mark.parametrize("C,hw", [(16, (16, 24)), (64, (16, 24)), (32, (64, 64)), (16, (36, 20)), (8, (16, 24))])
def test_resample(L, C, hw):
    rs = np.random.RandomState(4 + C)
    B, (H, W) = 2, hw
    yraw = rnd(rs, B, C, H, W)
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); sh = rnd(rs, C, scale=0.3)
    a = F.relu(yraw * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    yd, dsc, dsh = dev(nhwc(yraw)), dev(sc), dev(sh)

    # avg pool
    p_ref = F.avg_pool2d(a, 2, 2)
    gp = rnd(rs, *p_ref.shape)
    (ga_ref,) = torch.autograd.grad((p_ref * gp).sum(), [a])
    p = torch.empty(B, H // 2, W // 2, C, device="cuda")
    L.call("sifsr_bnrelu_pool2", yd, dsc, dsh, p, B, H, W, C, S())
    base = rnd(rs, B, C, H, W)
    gacc = dev(nhwc(base))
    L.call("sifsr_pool2_bwd", dev(nhwc(gp)), gacc, B, H, W, C, 1, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(p.cpu()), p_ref) < TOL
    assert rel_err(nchw(gacc.cpu()), base + ga_ref) < TOL

    # residual add
    pp = rnd(rs, B, C, H, W)
    r = torch.empty(B, H, W, C, device="cuda")
    L.call("sifsr_bnrelu_add", dev(nhwc(pp)), yd, dsc, dsh, r, C, B * H * W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(r.cpu()), pp + a.detach()) < TOL

    # bilinear x2, align_corners=True
    u_ref = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)
    gu = rnd(rs, *u_ref.shape)
    (ga2_ref,) = torch.autograd.grad((u_ref * gu).sum(), [a])
    u = torch.empty(B, 2 * H, 2 * W, C, device="cuda")
    L.call("sifsr_bnrelu_up2x", yd, dsc, dsh, u, B, H, W, C, S())
    g = torch.empty(B, H, W, C, device="cuda")
    L.call("sifsr_up2x_bwd", dev(nhwc(gu)), g, B, H, W, C, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(u.cpu()), u_ref) < TOL
    assert rel_err(nchw(g.cpu()), ga2_ref) < TOL
    # ... with the BatchNorm-backward sums of the low-resolution layer (sum dz, sum dz*y; dz = g*[y*scale+shift > 0])
    rows = L.call("sifsr_up2x_bwd_stat_rows", B, H, W, C)
    if rows:
        part = torch.full((rows, C, 2), float("nan"), device="cuda")
        g2 = torch.empty(B, H, W, C, device="cuda")
        L.call("sifsr_up2x_bwd_bn_sums", dev(nhwc(gu)), g2, B, H, W, C, yd, dsc, dsh, part, S())
        torch.cuda.synchronize()
        assert torch.equal(g2, g)
        dz = ga2_ref.double() * (yraw.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] > 0)
        ps = part.cpu().double().sum(0)
        assert torch.allclose(ps[:, 0], dz.sum((0, 2, 3)), rtol=1e-4, atol=1e-4)
        assert torch.allclose(ps[:, 1], (dz * yraw.double()).sum((0, 2, 3)), rtol=1e-4, atol=1e-4)
    else:
        assert C == 8


@pytest.mark.parametrize("hw", [(256, 256), (64, 128), (40, 56), (100, 36), (12, 16), (37, 50)])
def test_loss_operators(L, hw):
    import sifsr
    from oracle import sif_oracle as O
    H, W = hw
    rs = np.random.RandomState(5)
    x = rnd(rs, 2, 1, H, W)
    xk = x * 5.5698 + 307.2378
    ops = [("ftm", lambda t: O.get_output_ftm(t, mtf=0.25), lambda t: sifsr.get_output_ftm(t, mtf=0.25), x),
           ("sobel", O.sobel_bank, sifsr.sobel_bank, x)]
    if H % 4 == 0 and W % 4 == 0:          # any size >= 10 works; the decimating operator needs multiples of 4
        ops.insert(0, ("downscale", O.downscale_LST_SR_to_LR, sifsr.downscale_LST_SR_to_LR, xk))
    for name, fo, fh, inp in ops:
        a = inp.clone().requires_grad_(True)
        yo = fo(a)
        wgt = rnd(rs, *yo.shape)
        (go,) = torch.autograd.grad((yo * wgt).sum(), a)
        b = inp.clone().cuda().requires_grad_(True)
        yh = fh(b)
        (gh,) = torch.autograd.grad((yh * wgt.cuda()).sum(), b)
        assert rel_err(yh, yo) < TOL, name
        assert rel_err(gh, go) < TOL, name
    # huber
    a = (rnd(rs, 2, 4, 32, 32) * 1.5).requires_grad_(True)
    t = rnd(rs, 2, 4, 32, 32)
    lo = O.huber(a, -0.4 * t)
    (go,) = torch.autograd.grad(lo * 1.7, a)
    ad = a.detach().cuda().requires_grad_(True)
    lh = sifsr.huber_loss(ad, t.cuda(), -0.4)
    (gh,) = torch.autograd.grad(lh * 1.7, ad)
    assert abs(float(lh.detach()) - float(lo.detach())) < TOL * abs(float(lo.detach()))
    assert rel_err(gh, go) < TOL


@pytest.mark.parametrize("hw", [(256, 256), (48, 80), (32, 32), (100, 36)])
@pytest.mark.parametrize("kind,alpha,gamma", [("sr2", 0.5, -0.25), ("sr1", 0.99, -0.5), ("sr2", 0.1, -0.4)])
def test_fused_sif_loss(L, kind, alpha, gamma, hw):
    import sifsr
    from oracle import sif_oracle as O
    rs = np.random.RandomState(6)
    B, (H, W) = 2, hw
    sr = (rnd(rs, B, 1, H, W) * 1.3).requires_grad_(True)     # |e| > 1 on a fraction: both Huber branches
    lst = rnd(rs, B, 1, H // 4, W // 4)
    ndvi = rnd(rs, B, 1, H, W).clamp(-3, 3)
    mean, std = 307.2378, 5.5698
    ds_o, pl_o, loss_o = O.LOSSES[kind](sr, lst, ndvi, mean, std, alpha, gamma)
    (g_o,) = torch.autograd.grad(loss_o, sr)
    srd = sr.detach().cuda().requires_grad_(True)
    ds, pl, loss = sifsr.sif_loss(kind, srd, lst.cuda(), ndvi.cuda(), mean, std, alpha, gamma)
    (g,) = torch.autograd.grad(loss, srd)
    for got, ref in ((ds, ds_o), (pl, pl_o), (loss, loss_o)):
        assert abs(float(got.detach()) - float(ref.detach())) < TOL * abs(float(ref.detach())), (kind, float(got.detach()), float(ref.detach()))
    assert rel_err(g, g_o) < TOL


def test_adam_flat(L):
    from oracle import sif_oracle as O
    rs = np.random.RandomState(7)
    n = 10007
    p0 = rnd(rs, n)
    sd = {"p": p0.clone()}
    adam = O.AdamState(["p"], 1e-3)
    p = p0.clone().cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = rnd(rs, n, scale=0.01)
        adam.step(sd, {"p": g})
        L.call("sifsr_adam_flat", p, dev(g * 4), m, v, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 0.25, S())
    torch.cuda.synchronize()
    assert rel_err(p.cpu(), sd["p"]) < 1e-6
    assert rel_err(p.cpu() - p0, sd["p"] - p0) < 1e-4


@pytest.mark.parametrize("shape", [(16, 16, 256, 256, 2), (16, 16, 128, 128, 8), (32, 16, 256, 256, 1),
                                   (64, 32, 128, 128, 4), (64, 64, 32, 32, 16), (128, 64, 64, 64, 4), (32, 64, 64, 64, 16)])
def test_conv3x3_full_size_grids(L, shape):
    """Layer-sized problems: enough tiles that the persistent kernels run two workgroups per CU and several
    tiles per workgroup (the small cases above never do) -- forward and dgrad, repeated to expose races."""
    cin, cout, H, W, B = shape
    rs = np.random.RandomState(sum(shape))
    x = rnd(rs, B, cin, H, W)
    w = rnd(rs, cout, cin, 3, 3, scale=(2.0 / (9 * cin)) ** 0.5)
    a = x.clone().requires_grad_(True)
    y_ref = conv_rep(a, w)
    dy = rnd(rs, B, cout, H, W)
    (ga_ref,) = torch.autograd.grad((y_ref * dy).sum(), [a])
    wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda")
    dw_ = dev(w)
    L.call("sifsr_pack_conv_weights", dw_, cin, cout, wf, wd, S())
    dx, ddy = dev(nhwc(x)), dev(nhwc(dy))
    for _ in range(3):
        y = torch.full((B, H, W, cout), float("nan"), device="cuda")
        g = torch.full((B, H, W, cin), float("nan"), device="cuda")
        L.call("sifsr_conv3x3_fwd", dx, cin, None, None, None, 0, None, None, wf, y, cout, None, B, H, W, S())
        L.call("sifsr_conv3x3_dgrad", ddy, cout, wd, dw_, cin, g, cin, None, 0, None, B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(nchw(y.cpu()), y_ref) < TOL
        assert rel_err(nchw(g.cpu()), ga_ref) < TOL
    # the Winograd-domain kernels the model runs (16 out: two workgroups per CU with the weights in LDS; 32 / 64 out: all eight
    # waves staging and contracting; weight gradient with the engine's grid of one round of resident workgroups)
    wr = w.clone().requires_grad_(True)
    (gw_ref,) = torch.autograd.grad((conv_rep(x, wr) * dy).sum(), [wr])
    wwf = torch.empty(16 * cin * cout, device="cuda"); wwd = torch.empty(16 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", dw_, cin, cout, wwf, wwd, S())
    nstat = L.call("sifsr_conv3x3_stat_blocks_wino", B, H, W, cin, cout)
    for _ in range(3):
        y = torch.full((B, H, W, cout), float("nan"), device="cuda")
        g = torch.full((B, H, W, cin), float("nan"), device="cuda")
        part = torch.full((nstat, cout, 2), float("nan"), device="cuda")
        L.call("sifsr_conv3x3_fwd_wino", dx, cin, None, None, None, 0, None, None, wf, wwf, y, cout, part, B, H, W, S())
        L.call("sifsr_conv3x3_dgrad_wino", ddy, cout, wd, wwd, cin, g, cin, None, 0, None, B, H, W, S())
        for nb in (512 if cout < 64 else 256, 97):
            scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_wino_scratch_floats", cin, cout, nb), device="cuda")
            dwx = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
            L.call("sifsr_conv3x3_wgrad_wino", dx, cin, None, None, None, 0, None, None, ddy, None, None, cout, scratch, nb, dwx, B, H, W, S())
            torch.cuda.synchronize()
            assert rel_err(dwx.cpu(), gw_ref) < TOL, nb
        assert rel_err(nchw(y.cpu()), y_ref) < TOL
        assert rel_err(nchw(g.cpu()), ga_ref) < TOL
        ps = part.sum(0).cpu()
        assert torch.allclose(ps[:, 0], y_ref.detach().sum((0, 2, 3)), rtol=1e-4, atol=2e-2)
        assert torch.allclose(ps[:, 1], (y_ref.detach() ** 2).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


def _bn_setup(L, rs, y, gamma, beta):
    """batch statistics -> (mean, invstd, scale, shift) on the device through sifsr_bn_finalize"""
    B, C, H, W = y.shape
    if H % 16 == 0 and W % 16 == 0:
        t = y.unfold(2, 16, 16).unfold(3, 16, 16)
        p1 = t.sum((-1, -2)).permute(0, 2, 3, 1).reshape(-1, C)
        p2 = (t * t).sum((-1, -2)).permute(0, 2, 3, 1).reshape(-1, C)
    else:                                                  # (unfold would drop the partial tiles: one row per image)
        p1, p2 = y.double().sum((2, 3)).float(), (y.double() ** 2).sum((2, 3)).float()
    part = dev(torch.stack([p1, p2], -1))
    mean, invstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
    L.call("sifsr_bn_finalize", part, part.shape[0], C, float(B * H * W), dev(gamma), dev(beta), None, None, 0.1, 1e-5,
           mean, invstd, scale, shift, S())
    return mean, invstd, scale, shift


@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 16, 16), (3, 64, 32), (2, 24, 40), (1, 19, 37)])
def test_fused_tail_backward(L, shape):
    """sifsr_conv_out_bn_relu_bwd == autograd of conv_out(relu(bn(y))) w.r.t. (y, gamma, beta, w_out, b_out):
    the image borders (replicate-padding adjoint) are inside every case, (1,16,16) is one tile with all four; the last two
    shapes end in partial 16x16 tiles (the image border lies inside a tile)."""
    B, H, W = shape
    rs = np.random.RandomState(11 + H)
    y = (rnd(rs, B, 16, H, W) * 1.3 + 0.2).requires_grad_(True)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 16).astype(np.float32)).requires_grad_(True)
    beta = rnd(rs, 16, scale=0.2).requires_grad_(True)
    w = rnd(rs, 1, 16, 3, 3, scale=0.2).requires_grad_(True)
    b = rnd(rs, 1).requires_grad_(True)
    a = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    out = conv_rep(a, w, b)
    dsr = rnd(rs, B, 1, H, W)
    gy, gg, gb, gw, gbias = torch.autograd.grad((out * dsr).sum(), [y, gamma, beta, w, b])

    mean, invstd, scale, shift = _bn_setup(L, rs, y.detach(), gamma.detach(), beta.detach())
    for nblk in (3, B * ((H + 15) // 16) * ((W + 15) // 16)):
        scratch = torch.empty(64 + nblk * (145 + 32), device="cuda")
        dwb = torch.empty(145, device="cuda")
        dgam, dbet = (torch.empty(16, device="cuda") for _ in range(2))
        coef = torch.empty(48, dtype=torch.float64, device="cuda")
        dy = torch.full((B, H, W, 16), float("nan"), device="cuda")
        L.call("sifsr_conv_out_bn_relu_bwd", dev(nhwc(y.detach())), scale, shift, mean, invstd, dev(dsr), dev(w.detach()),
               scratch, nblk, dwb, dgam, dbet, coef, dy, B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(nchw(dy.cpu()), gy) < TOL
        assert rel_err(dgam.cpu(), gg) < TOL and rel_err(dbet.cpu(), gb) < TOL
        assert rel_err(dwb.cpu()[:144].view(1, 16, 3, 3), gw) < TOL
        assert rel_err(dwb.cpu()[144:], gbias) < TOL


@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 16, 16), (2, 24, 40), (8, 128, 128)])
def test_fused_head_backward(L, shape):
    """sifsr_conv_in_bn_relu_bwd == autograd of relu(bn(conv_in(x))) w.r.t. (w_in, gamma, beta)"""
    B, H, W = shape
    rs = np.random.RandomState(17 + H)
    x = rnd(rs, B, 2, H, W)
    w = rnd(rs, 16, 2, 3, 3, scale=0.3).requires_grad_(True)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 16).astype(np.float32)).requires_grad_(True)
    beta = rnd(rs, 16, scale=0.2).requires_grad_(True)
    y = conv_rep(x, w)
    a = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    g = rnd(rs, B, 16, H, W)
    gw, gg, gb = torch.autograd.grad((a * g).sum(), [w, gamma, beta])
    mean, invstd, scale, shift = _bn_setup(L, rs, y.detach(), gamma.detach(), beta.detach())
    nblk = 3
    scratch = torch.empty(nblk * 288, device="cuda")
    dw = torch.empty(16, 2, 3, 3, device="cuda")
    dgam, dbet = (torch.empty(16, device="cuda") for _ in range(2))
    coef = torch.empty(48, dtype=torch.float64, device="cuda")
    L.call("sifsr_conv_in_bn_relu_bwd", dev(x), dev(nhwc(g)), dev(nhwc(y.detach())), scale, shift, mean, invstd, scratch,
           nblk, dw, dgam, dbet, coef, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), gw) < TOL
    assert rel_err(dgam.cpu(), gg) < TOL and rel_err(dbet.cpu(), gb) < TOL
    # the linear form (round 3): the same dW from dz = g*[z > 0], the coefficients just returned, and the network input alone
    dz = dev(nhwc(g)) * ((dev(nhwc(y.detach())) * scale + shift) > 0)
    for nb in (3, B * ((H + 15) // 16) * ((W + 15) // 16)):
        sc2 = torch.empty(L.call("sifsr_conv_in_bwd_linear_scratch_floats", nb), device="cuda")
        dw2 = torch.full((16, 2, 3, 3), float("nan"), device="cuda")
        L.call("sifsr_conv_in_bwd_linear", dev(x), dz.contiguous(), dev(w.detach()), coef, sc2, nb, dw2, B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(dw2.cpu(), gw) < TOL, (nb, rel_err(dw2.cpu(), gw))


# ---- BatchNorm+ReLU backward fused into the input- and weight-gradient convolutions (round 2) --------------------
# (cin, cout, H, W, B, with_pool_adjoint)
FUSED_CASES = [
    (16, 16, 32, 48, 2, False),
    (16, 16, 32, 32, 2, True),      # inbloc.bloc.3: g completed in place by the AvgPool adjoint
    (16, 32, 16, 32, 1, True),      # db1.lastconv
    (32, 32, 32, 32, 2, False),
    (32, 64, 16, 16, 2, True),      # db2.lastconv
    (64, 64, 32, 32, 1, False),
    (64, 32, 16, 16, 2, False),     # ub1.convbloc.bloc.3
    (128, 64, 16, 16, 1, False),    # ub1.convbloc.bloc.0
    (32, 16, 24, 40, 2, False),     # partial tiles
    (64, 64, 12, 20, 2, False),
    (16, 16, 20, 36, 1, True),
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_bn_relu_backward_fused_into_dgrad_and_wgrad(L, case):
    """z = relu(bn_train(conv(a))) with upstream gradient g on z: the model's backward never stores dL/dy -- it is formed
    from (g, y) inside the staging of the input-gradient and weight-gradient convolutions (bn_bwd4).  Checked against
    float64 autograd of the same three PyTorch ops (F.conv2d replicate, F.batch_norm training, relu), and against the
    unfused chain of C-ABI calls (sifsr_bn_relu_bwd + sifsr_conv3x3_dgrad / _wgrad)."""
    cin, cout, H, W, B, pool = case
    rs = np.random.RandomState(hash(case) % 2**31)
    a = rnd(rs, B, cin, H, W)
    w = rnd(rs, cout, cin, 3, 3, scale=(2.0 / (9 * cin)) ** 0.5)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    beta = rnd(rs, cout, scale=0.5)
    g = rnd(rs, B, cout, H, W)
    gp = rnd(rs, B, cout, H // 2, W // 2) if pool else None
    # float64 reference
    a64, w64 = a.double().requires_grad_(True), w.double().requires_grad_(True)
    ga64, be64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y64 = conv_rep(a64, w64)
    z64 = F.relu(F.batch_norm(y64, None, None, ga64, be64, training=True, eps=1e-5))
    g_eff = g.double() + (0.25 * gp.double().repeat_interleave(2, 2).repeat_interleave(2, 3) if pool else 0.0)
    (dy64,) = torch.autograd.grad((z64 * g_eff).sum(), y64, retain_graph=True)
    ga_ref, gw_ref, dgam_ref, dbet_ref = torch.autograd.grad((z64 * g_eff).sum(), [a64, w64, ga64, be64])

    # device: forward conv + statistics, then the backward pieces
    wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights", dev(w), cin, cout, wf, wd, S())
    da = dev(nhwc(a))
    y = torch.empty(B, H, W, cout, device="cuda")
    nblk = L.call("sifsr_conv3x3_stat_blocks", B, H, W, cout)
    part = torch.empty(nblk, cout, 2, device="cuda")
    L.call("sifsr_conv3x3_fwd", da, cin, None, None, None, 0, None, None, wf, y, cout, part, B, H, W, S())
    mean, invstd, scale, shift = (torch.empty(cout, device="cuda") for _ in range(4))
    rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    L.call("sifsr_bn_finalize", part, nblk, cout, float(B * H * W), dev(gamma), dev(beta), rm, rv, 0.1, 1e-5, mean, invstd, scale, shift, S())
    npix = B * H * W
    nb = max(1, min(1024, npix // 256))
    partials = torch.empty(max(nb, 1024) * cout * 2, device="cuda")
    dgam, dbet = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
    coef = torch.empty(3 * cout, dtype=torch.float64, device="cuda")
    coef_f = torch.empty(4 * cout, device="cuda")
    dg = dev(nhwc(g))
    dgp = dev(nhwc(gp)) if pool else None
    # unfused chain first (it does not modify g)
    dy_u = torch.empty(B, H, W, cout, device="cuda")
    L.call("sifsr_bn_relu_bwd", dg, y, scale, shift, mean, invstd, cout, npix, partials, nb, dgam, dbet, coef, dy_u, dgp, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(nchw(dy_u.cpu()), dy64) < TOL
    gin_u = torch.empty(B, H, W, cin, device="cuda")
    L.call("sifsr_conv3x3_dgrad", dy_u, cout, wd, dev(w), cin, gin_u, cin, None, 0, None, B, H, W, S())
    # fused
    dgam.fill_(float("nan")); dbet.fill_(float("nan"))
    L.call("sifsr_bn_relu_bwd_coef", dg, y, scale, shift, mean, invstd, dev(beta), cout, npix, partials, nb, dgam, dbet, coef, coef_f, dgp, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(dgam.cpu(), dgam_ref) < TOL and rel_err(dbet.cpu(), dbet_ref) < TOL
    if pool:
        assert rel_err(nchw(dg.cpu()), g_eff) < 1e-6       # completed in place
    wwf = torch.empty(16 * cin * cout, device="cuda"); wwd = torch.empty(16 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", dev(w), cin, cout, wwf, wwd, S())
    edge = torch.zeros(H, W, dtype=torch.bool); edge[0] = edge[-1] = True; edge[:, 0] = edge[:, -1] = True
    for wino in (None, wwd):                               # tap-domain and Winograd-domain contraction
        border = torch.full((B, H, W, cout), float("nan"), device="cuda")
        gin = torch.full((B, H, W, cin), float("nan"), device="cuda")
        L.call("sifsr_conv3x3_dgrad_fused", dg, y, coef_f, cout, wd, wino, cin, gin, cin, None, 0, None, border, B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(nchw(gin.cpu()), ga_ref) < TOL
        assert rel_err(gin, gin_u) < 2e-5                      # the fp32 on-load form against the float64 stored form
        bc = nchw(border.cpu())
        assert torch.isnan(bc[:, :, ~edge]).all()              # only border pixels are written ...
        assert rel_err(bc[:, :, edge], dy64[:, :, edge]) < TOL  # ... with dL/dy
    for nbk in (1, 7, 64):
        scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_scratch_floats", cin, cout, nbk), device="cuda")
        dwo = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
        L.call("sifsr_conv3x3_wgrad_fused", da, cin, None, None, None, 0, None, None, dg, y, coef_f, cout, scratch, nbk, dwo, B, H, W, S())
        torch.cuda.synchronize()
        assert rel_err(dwo.cpu(), gw_ref) < TOL, nbk
        if H % 2 == 0 and W % 2 == 0:
            scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_wino_scratch_floats", cin, cout, nbk), device="cuda")
            dwx = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
            L.call("sifsr_conv3x3_wgrad_wino", da, cin, None, None, None, 0, None, None, dg, y, coef_f, cout, scratch, nbk, dwx, B, H, W, S())
            torch.cuda.synchronize()
            assert rel_err(dwx.cpu(), gw_ref) < TOL, nbk


# ---- input gradient AND weight gradient of a 16 -> 16 layer in one kernel (round 3, conv_bwd16.hip) ----------------------
# (H, W, B, input is a raw conv output with folded BatchNorm, dL/dy formed on load from (g, y), residual addend, fused BN sums)
BWD16_CASES = [
    (32, 48, 2, True, True, False, True),      # inbloc.bloc.3 / db1.res.3: everything on
    (32, 32, 1, True, False, False, True),     # ub3.convbloc.bloc.3: dL/dy stored (the fused tail wrote it)
    (48, 32, 2, False, True, True, False),     # db1.res.0: stored input (the pooled tensor), residual gradient added
    (128, 128, 8, True, True, False, True),    # 512 tiles on 256 workgroups: two tiles per workgroup, XCD-strided walk
    (64, 32, 3, False, False, False, False),
]


@pytest.mark.parametrize("case", BWD16_CASES)
def test_conv3x3_bwd16_fused_dgrad_wgrad(L, case):
    """z = relu(bn_train(conv(a))), a = relu(x*xs + xsh) (or x itself), upstream gradient g on z.  One call must return the
    gradient w.r.t. a (incl. the replicate-border fold, + addend), the weight gradient, and -- when asked -- the
    BatchNorm-backward sums of the layer below (sum dz, sum dz*x with dz = gin*[x*xs+xsh > 0]).  Reference: float64 autograd
    of the PyTorch ops; also compared with the separate Winograd kernels the model runs for other shapes."""
    H, W, B, x_bn, dyf, with_add, with_stats = case
    C = 16
    rs = np.random.RandomState(hash(case) % 2**31)
    x = rnd(rs, B, C, H, W)
    xs = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)) if x_bn else None
    xsh = rnd(rs, C, scale=0.3) if x_bn else None
    w = rnd(rs, C, C, 3, 3, scale=(2.0 / (9 * C)) ** 0.5)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32))
    beta = rnd(rs, C, scale=0.5)
    g = rnd(rs, B, C, H, W)
    addend = rnd(rs, B, C, H, W) if with_add else None
    # ---- float64 reference
    x64 = x.double()
    a64 = (F.relu(x64 * xs.double().view(1, C, 1, 1) + xsh.double().view(1, C, 1, 1)) if x_bn else x64).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = conv_rep(a64, w64)
    z64 = F.relu(F.batch_norm(y64, None, None, gamma.double(), beta.double(), training=True, eps=1e-5))
    (dy64,) = torch.autograd.grad((z64 * g.double()).sum(), y64, retain_graph=True)
    ga_ref, gw_ref = torch.autograd.grad((z64 * g.double()).sum(), [a64, w64])
    gin_ref = ga_ref + (addend.double() if with_add else 0.0)
    # ---- device: forward conv + statistics + coefficients (as the model's schedule), then the fused backward
    wf = torch.empty(9 * C * C, device="cuda"); wd = torch.empty(4 * 9 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights", dev(w), C, C, wf, wd, S())
    wwf = torch.empty(16 * C * C, device="cuda"); wwd = torch.empty(16 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", dev(w), C, C, wwf, wwd, S())
    dx = dev(nhwc(x))
    dxs, dxsh = (dev(xs), dev(xsh)) if x_bn else (None, None)
    y = torch.empty(B, H, W, C, device="cuda")
    nblk = L.call("sifsr_conv3x3_stat_blocks", B, H, W, C)
    part = torch.empty(nblk, C, 2, device="cuda")
    L.call("sifsr_conv3x3_fwd", dx, C, dxs, dxsh, None, 0, None, None, wf, y, C, part, B, H, W, S())
    mean, invstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    L.call("sifsr_bn_finalize", part, nblk, C, float(B * H * W), dev(gamma), dev(beta), rm, rv, 0.1, 1e-5, mean, invstd, scale, shift, S())
    npix = B * H * W
    nb = max(1, min(1024, npix // 256))
    partials = torch.empty(max(nb, 1024) * C * 2, device="cuda")
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    coef = torch.empty(3 * C, dtype=torch.float64, device="cuda")
    coef_f = torch.empty(4 * C, device="cuda")
    dg = dev(nhwc(g))
    L.call("sifsr_bn_relu_bwd_coef", dg, y, scale, shift, mean, invstd, dev(beta), C, npix, partials, nb, dgam, dbet, coef, coef_f, None, H, W, S())
    rows = L.call("sifsr_conv3x3_bwd16_stat_rows", B, H, W)
    assert rows > 0
    scratch = torch.empty(L.call("sifsr_conv3x3_bwd16_scratch_floats", B, H, W), device="cuda")
    gin = torch.full((B, H, W, C), float("nan"), device="cuda")
    dw = torch.full((C, C, 3, 3), float("nan"), device="cuda")
    border = torch.full((B, H, W, C), float("nan"), device="cuda")
    bnp = torch.full((rows, C, 2), float("nan"), device="cuda") if with_stats else None
    if dyf:
        g_arg, y_arg, c_arg, b_arg = dg, y, coef_f, border
    else:
        g_arg, y_arg, c_arg, b_arg = dev(nhwc(dy64.float())), None, None, None
    L.call("sifsr_conv3x3_bwd16", dx, dxs, dxsh, g_arg, y_arg, c_arg, b_arg, wd, wwd, gin, dev(nhwc(addend)) if with_add else None,
           dx if with_stats else None, dxs if with_stats else None, dxsh if with_stats else None, bnp, scratch, dw, B, H, W, S())
    torch.cuda.synchronize()
    e_gin, e_dw = rel_err(nchw(gin.cpu()), gin_ref), rel_err(dw.cpu(), gw_ref)
    print(f"bwd16 {case}: gin {e_gin:.2e}, dw {e_dw:.2e}")
    assert e_gin < TOL and e_dw < TOL
    if dyf:
        edge = torch.zeros(H, W, dtype=torch.bool); edge[0] = edge[-1] = True; edge[:, 0] = edge[:, -1] = True
        bc = nchw(border.cpu())
        assert torch.isnan(bc[:, :, ~edge]).all() and rel_err(bc[:, :, edge], dy64[:, :, edge]) < TOL
    if with_stats:
        assert x_bn
        zpos = (x64 * xs.double().view(1, C, 1, 1) + xsh.double().view(1, C, 1, 1)) > 0
        dz = torch.where(zpos, gin_ref, torch.zeros_like(gin_ref))
        s_ref = torch.stack((dz.sum(dim=(0, 2, 3)), (dz * x64).sum(dim=(0, 2, 3))), dim=1)       # [C][2]
        s_got = bnp.double().sum(dim=0).cpu()
        assert rel_err(s_got, s_ref) < TOL, (s_got, s_ref)
    # the separate kernels on the same operands: same results to fp32 rounding
    gin_s = torch.full((B, H, W, C), float("nan"), device="cuda")
    if dyf:
        L.call("sifsr_conv3x3_dgrad_fused", dg, y, coef_f, C, wd, wwd, C, gin_s, C, None, 0, dev(nhwc(addend)) if with_add else None,
               torch.empty_like(border), B, H, W, S())
    else:
        L.call("sifsr_conv3x3_dgrad_wino", g_arg, C, wd, wwd, C, gin_s, C, None, 0, None, B, H, W, S())
    sc2 = torch.empty(L.call("sifsr_conv3x3_wgrad_wino_scratch_floats", C, C, 64), device="cuda")
    dw_s = torch.empty(C, C, 3, 3, device="cuda")
    L.call("sifsr_conv3x3_wgrad_wino", dx, C, dxs, dxsh, None, 0, None, None, g_arg, y_arg, c_arg, C, sc2, 64, dw_s, B, H, W, S())
    torch.cuda.synchronize()
    assert rel_err(gin, gin_s) < 2e-6 and rel_err(dw, dw_s) < 2e-5


@pytest.mark.parametrize("shape", [(32, 48, 2), (128, 128, 8), (48, 32, 3)])
def test_conv3x3_bwd16_tail_recomputes_the_outlay_input_gradient(L, shape):
    """ub3.convbloc.bloc.3 -> outlay: sr = conv_rep(z, w_out), z = relu(bn_train(conv_rep(a, w))).  Given dsr = d loss / d sr the
    tail entry must return the same (gin, dw, border dL/dy, BatchNorm sums of the layer below) as float64 autograd -- the
    upstream gradient g = outlay^T(dsr), replicate-padding adjoint included, exists only inside the kernel's staging -- and the
    same as sifsr_conv3x3_bwd16 fed the float64 g."""
    H, W, B = shape
    C = 16
    rs = np.random.RandomState(4242 + H + W)
    x = rnd(rs, B, C, H, W)
    xs = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); xsh = rnd(rs, C, scale=0.3)
    w = rnd(rs, C, C, 3, 3, scale=(2.0 / (9 * C)) ** 0.5)
    w_out = rnd(rs, 1, C, 3, 3, scale=0.2)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); beta = rnd(rs, C, scale=0.5)
    dsr = rnd(rs, B, 1, H, W)
    x64 = x.double()
    a64 = F.relu(x64 * xs.double().view(1, C, 1, 1) + xsh.double().view(1, C, 1, 1)).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = conv_rep(a64, w64)
    z64 = F.relu(F.batch_norm(y64, None, None, gamma.double(), beta.double(), training=True, eps=1e-5))
    loss = (conv_rep(z64, w_out.double()) * dsr.double()).sum()
    g64, dy64 = torch.autograd.grad(loss, [z64, y64], retain_graph=True)
    ga_ref, gw_ref = torch.autograd.grad(loss, [a64, w64])
    wf = torch.empty(9 * C * C, device="cuda"); wd = torch.empty(4 * 9 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights", dev(w), C, C, wf, wd, S())
    wwf = torch.empty(16 * C * C, device="cuda"); wwd = torch.empty(16 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", dev(w), C, C, wwf, wwd, S())
    dx, dxs, dxsh = dev(nhwc(x)), dev(xs), dev(xsh)
    y = torch.empty(B, H, W, C, device="cuda")
    nblk = L.call("sifsr_conv3x3_stat_blocks", B, H, W, C)
    part = torch.empty(nblk, C, 2, device="cuda")
    L.call("sifsr_conv3x3_fwd", dx, C, dxs, dxsh, None, 0, None, None, wf, y, C, part, B, H, W, S())
    mean, invstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    L.call("sifsr_bn_finalize", part, nblk, C, float(B * H * W), dev(gamma), dev(beta), rm, rv, 0.1, 1e-5, mean, invstd, scale, shift, S())
    npix = B * H * W
    nb = max(1, min(1024, npix // 256))
    partials = torch.empty(max(nb, 1024) * C * 2, device="cuda")
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    coef = torch.empty(3 * C, dtype=torch.float64, device="cuda"); coef_f = torch.empty(4 * C, device="cuda")
    dg = dev(nhwc(g64.float()))
    L.call("sifsr_bn_relu_bwd_coef", dg, y, scale, shift, mean, invstd, dev(beta), C, npix, partials, nb, dgam, dbet, coef, coef_f, None, H, W, S())
    rows = L.call("sifsr_conv3x3_bwd16_stat_rows", B, H, W)
    out = {}
    for kind in ("tail", "stored_g"):
        scratch = torch.empty(L.call("sifsr_conv3x3_bwd16_scratch_floats", B, H, W), device="cuda")
        gin = torch.full((B, H, W, C), float("nan"), device="cuda")
        dw = torch.full((C, C, 3, 3), float("nan"), device="cuda")
        border = torch.full((B, H, W, C), float("nan"), device="cuda")
        bnp = torch.full((rows, C, 2), float("nan"), device="cuda")
        if kind == "tail":
            L.call("sifsr_conv3x3_bwd16_tail", dx, dxs, dxsh, dev(dsr.reshape(B, H, W)), dev(w_out), y, coef_f, border, wd, wwd, gin,
                   dx, dxs, dxsh, bnp, scratch, dw, B, H, W, S())
        else:
            L.call("sifsr_conv3x3_bwd16", dx, dxs, dxsh, dg, y, coef_f, border, wd, wwd, gin, None, dx, dxs, dxsh, bnp, scratch, dw,
                   B, H, W, S())
        torch.cuda.synchronize()
        out[kind] = (gin, dw, border, bnp.double().sum(dim=0).cpu())
    gin, dw, border, s_got = out["tail"]
    e_gin, e_dw = rel_err(nchw(gin.cpu()), ga_ref), rel_err(dw.cpu(), gw_ref)
    print(f"bwd16 tail {shape}: gin {e_gin:.2e}, dw {e_dw:.2e}")
    assert e_gin < TOL and e_dw < TOL
    edge = torch.zeros(H, W, dtype=torch.bool); edge[0] = edge[-1] = True; edge[:, 0] = edge[:, -1] = True
    bc = nchw(border.cpu())
    assert torch.isnan(bc[:, :, ~edge]).all() and rel_err(bc[:, :, edge], dy64[:, :, edge]) < TOL
    zpos = (x64 * xs.double().view(1, C, 1, 1) + xsh.double().view(1, C, 1, 1)) > 0
    dz = torch.where(zpos, ga_ref, torch.zeros_like(ga_ref))
    s_ref = torch.stack((dz.sum(dim=(0, 2, 3)), (dz * x64).sum(dim=(0, 2, 3))), dim=1)
    assert rel_err(s_got, s_ref) < TOL
    assert rel_err(gin, out["stored_g"][0]) < 2e-6 and rel_err(dw, out["stored_g"][1]) < 2e-5


@pytest.mark.parametrize("shape", [(32, 48, 2), (128, 128, 8)])
def test_conv3x3_bwd16_pool_adds_the_pooling_adjoint_while_staging(L, shape):
    """inbloc.bloc.3: its output feeds the decoder skip AND AvgPool2d(2,2); the upstream gradient is g + pool^T(gp).  The pooled entry
    (g and gp given separately) must equal sifsr_conv3x3_bwd16 fed the completed gradient -- gin, dW, border dL/dy, BatchNorm sums."""
    H, W, B = shape
    C = 16
    rs = np.random.RandomState(777 + H)
    x = rnd(rs, B, C, H, W)
    xs = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); xsh = rnd(rs, C, scale=0.3)
    w = rnd(rs, C, C, 3, 3, scale=(2.0 / (9 * C)) ** 0.5)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); beta = rnd(rs, C, scale=0.5)
    g = rnd(rs, B, C, H, W); gp = rnd(rs, B, C, H // 2, W // 2)
    wf = torch.empty(9 * C * C, device="cuda"); wd = torch.empty(4 * 9 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights", dev(w), C, C, wf, wd, S())
    wwf = torch.empty(16 * C * C, device="cuda"); wwd = torch.empty(16 * C * C, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", dev(w), C, C, wwf, wwd, S())
    dx, dxs, dxsh = dev(nhwc(x)), dev(xs), dev(xsh)
    y = torch.empty(B, H, W, C, device="cuda")
    nblk = L.call("sifsr_conv3x3_stat_blocks", B, H, W, C)
    part = torch.empty(nblk, C, 2, device="cuda")
    L.call("sifsr_conv3x3_fwd", dx, C, dxs, dxsh, None, 0, None, None, wf, y, C, part, B, H, W, S())
    mean, invstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    L.call("sifsr_bn_finalize", part, nblk, C, float(B * H * W), dev(gamma), dev(beta), rm, rv, 0.1, 1e-5, mean, invstd, scale, shift, S())
    npix = B * H * W
    nb = max(1, min(1024, npix // 256))
    partials = torch.empty(max(nb, 1024) * C * 2, device="cuda")
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    coef = torch.empty(3 * C, dtype=torch.float64, device="cuda"); coef_f = torch.empty(4 * C, device="cuda")
    dg, dgp = dev(nhwc(g)), dev(nhwc(gp))
    g_eff = dg.clone()                                     # completed in place by the reduction (the single-op entry writes it back)
    L.call("sifsr_bn_relu_bwd_coef", g_eff, y, scale, shift, mean, invstd, dev(beta), C, npix, partials, nb, dgam, dbet, coef, coef_f, dgp, H, W, S())
    torch.cuda.synchronize()
    ref_eff = g + 0.25 * gp.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    assert rel_err(nchw(g_eff.cpu()), ref_eff) < 1e-6
    rows = L.call("sifsr_conv3x3_bwd16_stat_rows", B, H, W)
    out = {}
    for kind in ("pool", "completed"):
        scratch = torch.empty(L.call("sifsr_conv3x3_bwd16_scratch_floats", B, H, W), device="cuda")
        gin = torch.full((B, H, W, C), float("nan"), device="cuda")
        dw = torch.full((C, C, 3, 3), float("nan"), device="cuda")
        border = torch.full((B, H, W, C), float("nan"), device="cuda")
        bnp = torch.full((rows, C, 2), float("nan"), device="cuda")
        if kind == "pool":
            L.call("sifsr_conv3x3_bwd16_pool", dx, dxs, dxsh, dg, dgp, y, coef_f, border, wd, wwd, gin, dx, dxs, dxsh, bnp, scratch, dw,
                   B, H, W, S())
        else:
            L.call("sifsr_conv3x3_bwd16", dx, dxs, dxsh, g_eff, y, coef_f, border, wd, wwd, gin, None, dx, dxs, dxsh, bnp, scratch, dw,
                   B, H, W, S())
        torch.cuda.synchronize()
        out[kind] = (gin.clone(), dw.clone(), border.clone(), bnp.double().sum(dim=0).cpu())
    a_, b_ = out["pool"], out["completed"]
    assert rel_err(a_[0], b_[0]) < 2e-6 and rel_err(a_[1], b_[1]) < 2e-5
    edge = torch.zeros(H, W, dtype=torch.bool); edge[0] = edge[-1] = True; edge[:, 0] = edge[:, -1] = True
    ba, bb = nchw(a_[2].cpu()), nchw(b_[2].cpu())
    assert torch.isnan(ba[:, :, ~edge]).all() and rel_err(ba[:, :, edge], bb[:, :, edge]) < 2e-6
    assert rel_err(a_[3], b_[3]) < 1e-5


def test_conv3x3_bwd16_rejects_other_shapes(L):
    assert L.call("sifsr_conv3x3_bwd16_stat_rows", 2, 24, 32) == 0 and L.call("sifsr_conv3x3_bwd16_scratch_floats", 2, 16, 16) == 0
    t = torch.zeros(2, 24, 32, 16, device="cuda")
    w = torch.zeros(4 * 9 * 256, device="cuda")
    with pytest.raises(Exception):
        L.call("sifsr_conv3x3_bwd16", t, None, None, t, None, None, None, w, w, t, None, None, None, None, None, t, w, 2, 24, 32, S())
