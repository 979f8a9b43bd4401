"""Isolate per-op error on REALISTIC backward data (fp64 oracle tape -> cast to fp32 -> one HIP op -> compare with fp64)."""
import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F, numpy as np
import sifsr
from sifsr import _lib as L
from oracle import sif_oracle as O
from tests.test_ops_gpu import dev, nhwc, nchw, S


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu(); return float((a - b).norm() / b.norm())


MEAN, STD = 307.2378, 5.5698
kind, alpha, gamma, ws_, bs_ = "sr1", 0.99, -0.5, 32, 42
B = 2
sd = O.synthetic_state(ws_); lst, lst_up, ndvi = O.synthetic_batch(bs_, B)
dt = torch.float64
s = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
for n in O.param_names(): s[n] = s[n].clone().requires_grad_(True)
tape = {}
orig_conv, orig_interp = O._conv3x3_rep, F.interpolate


def rec(x, w, b=None):
    if x.requires_grad: x.retain_grad()
    y = orig_conv(x, w, b); y.retain_grad(); tape[id(w)] = (x, y); return y


ups = []


def rec_up(x, **kw):
    x.retain_grad(); u = orig_interp(x, **kw); u.retain_grad(); ups.append((x, u)); return u


O._conv3x3_rep = rec
O.F.interpolate = rec_up
sr = O.modelb2_forward(s, torch.cat((lst_up, ndvi), 1).to(dt), True)
O.F.interpolate = orig_interp
_, _, loss = O.LOSSES[kind](sr, lst.to(dt), ndvi.to(dt), MEAN, STD, alpha, gamma)
loss.backward()
O._conv3x3_rep = orig_conv
name = "ub2.convbloc.bloc.0"
w = s[name + ".weight"]; xin, y = tape[id(w)]
dy64, gin64 = y.grad, xin.grad            # (B,32,128,128), (B,64,128,128)
cout, cin, H = 32, 64, 128
wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda"); dw_ = dev(w.detach().float())
L.call("sifsr_pack_conv_weights", dw_, cin, cout, wf, wd, S())
g0 = torch.empty(B, H, H, 32, device="cuda"); g1 = torch.empty(B, H, H, 32, device="cuda")
L.call("sifsr_conv3x3_dgrad", dev(nhwc(dy64.float())), cout, wd, dw_, cin, g0, 32, g1, 32, None, B, H, H, S()); torch.cuda.synchronize()
g = torch.cat([nchw(g0.cpu()), nchw(g1.cpu())], 1)
xin32 = xin.detach().float().requires_grad_(True)
(gin32,) = torch.autograd.grad((orig_conv(xin32, w.detach().float()) * dy64.float()).sum(), xin32)
print("dgrad(ub2.0) on real data: hip vs64 %.2e   cpu32 vs64 %.2e ; first-half(gU) hip %.2e cpu %.2e" % (
    rel(g, gin64), rel(gin32, gin64), rel(g[:, :32], gin64[:, :32]), rel(gin32[:, :32], gin64[:, :32])))
# up2x adjoint: ups[1] is ub2's upsample (x = relu(bn(y_u1b)) 32ch@64 -> 128)
xu, u = ups[1]
gu64, gx64 = u.grad, xu.grad
print("   shapes", tuple(gu64.shape), tuple(gx64.shape), " gU vs dgrad-first-half consistency %.2e" % rel(gin64[:, :32], gu64))
gl = torch.empty(B, 64, 64, 32, device="cuda")
L.call("sifsr_up2x_bwd", dev(nhwc(gu64.float())), gl, B, 64, 64, 32, S()); torch.cuda.synchronize()
a32 = xu.detach().float().requires_grad_(True)
(gx32,) = torch.autograd.grad((orig_interp(a32, scale_factor=2, mode="bilinear", align_corners=True) * gu64.float()).sum(), a32)
print("up2x_bwd on real data: hip vs64 %.2e   cpu32 vs64 %.2e   hip vs cpu32 %.2e" % (
    rel(nchw(gl.cpu()), gx64), rel(gx32, gx64), rel(nchw(gl.cpu()), gx32)))
print("   gu stats: mean %.3e  std %.3e ; gx mean %.3e std %.3e" % (gu64.mean(), gu64.std(), gx64.mean(), gx64.std()))
