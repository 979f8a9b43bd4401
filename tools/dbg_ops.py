import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import sifsr
from sifsr import _lib as L
from tests.conftest import rel_err
from tests.test_ops_gpu import dev, nhwc, nchw, S, rnd, conv_rep
rs = np.random.RandomState(0)
def border_boost(t, f=30.0):
    t = t.clone(); t[..., 0, :] *= f; t[..., -1, :] *= f; t[..., :, 0] *= f; t[..., :, -1] *= f; return t
# up2x_bwd at scale
for (B, Hin, C) in ((2, 64, 32), (2, 32, 64), (2, 128, 16)):
    a = rnd(rs, B, C, Hin, Hin).requires_grad_(True)
    u = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)
    gu = border_boost(rnd(rs, *u.shape))
    (ga,) = torch.autograd.grad((u * gu).sum(), [a])
    g = torch.empty(B, Hin, Hin, C, device="cuda")
    L.call("sifsr_up2x_bwd", dev(nhwc(gu)), g, B, Hin, Hin, C, S()); torch.cuda.synchronize()
    print("up2x_bwd", B, Hin, C, rel_err(nchw(g.cpu()), ga))
# dgrad at scale: (C0, C1, cout, H)
for (C0, C1, cout, H, B) in ((32, 32, 32, 128, 2), (64, 64, 64, 64, 2), (16, 16, 16, 256, 2), (16, 0, 16, 256, 2), (32, 0, 64, 64, 2), (64,0,64,32,2), (16,0,32,128,2)):
    cin = C0 + C1
    a = rnd(rs, B, cin, H, H).requires_grad_(True)
    w = rnd(rs, cout, cin, 3, 3, scale=(2.0 / (9 * cin)) ** 0.5)
    y = conv_rep(a, w)
    dy = border_boost(rnd(rs, B, cout, H, H))
    (ga,) = torch.autograd.grad((y * dy).sum(), [a])
    wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda"); dw_ = dev(w)
    L.call("sifsr_pack_conv_weights", dw_, cin, cout, wf, wd, S())
    g0 = torch.empty(B, H, H, C0, device="cuda"); g1 = torch.empty(B, H, H, C1, device="cuda") if C1 else None
    L.call("sifsr_conv3x3_dgrad", dev(nhwc(dy)), cout, wd, dw_, cin, g0, C0, g1, C1, None, B, H, H, S()); torch.cuda.synchronize()
    g = nchw(g0.cpu()) if not C1 else torch.cat([nchw(g0.cpu()), nchw(g1.cpu())], 1)
    d = (g - ga).abs()
    inner = d[..., 1:-1, 1:-1].max().item(); edge = d.max().item()
    print("dgrad", C0, C1, cout, H, "rel", rel_err(g, ga), "inner abs", inner, "edge abs", edge, "max ref", ga.abs().max().item())
