"""Layer-by-layer parity of the backward pass: dL/dy_l of every conv (HIP workspace) vs the oracle in fp32 and fp64."""
import sys, os, copy, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import sifsr
from sifsr import _lib as L
from oracle import sif_oracle as O
from tests.conftest import rel_err as rel_max
def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm())
MEAN, STD = 307.2378, 5.5698
kind, alpha, gamma, ws_, bs_ = sys.argv[1] if len(sys.argv) > 1 else "sr1", 0.99, -0.5, 32, 42
if kind == "sr2": alpha, gamma, ws_, bs_ = 0.5, -0.25, 31, 41
B = 2
sd = O.synthetic_state(ws_); lst, lst_up, ndvi = O.synthetic_batch(bs_, B)

def oracle_tape(dtype):
    s = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for n in O.param_names(): s[n] = s[n].clone().requires_grad_(True)
    tape = {}
    orig = O._conv3x3_rep
    def rec(x, w, b=None):
        y = orig(x, w, b); y.retain_grad(); tape[id(w)] = y; return y
    O._conv3x3_rep = rec
    try:
        sr = O.modelb2_forward(s, torch.cat((lst_up, ndvi), 1).to(dtype), True)
        sr.retain_grad()
        _, _, loss = O.LOSSES[kind](sr, lst.to(dtype), ndvi.to(dtype), MEAN, STD, alpha, gamma)
        loss.backward()
    finally:
        O._conv3x3_rep = orig
    out = {}
    for conv, bn, cin, cout in O.CONV_BN_LAYERS:
        y = tape[id(s[conv + ".weight"])]
        z = F.batch_norm(y.detach(), None, None, s[bn + ".weight"].detach(), s[bn + ".bias"].detach(), True, 0.0, 1e-5)
        out[conv] = (y.detach(), y.grad.detach(), (z > 0) if O.RELU_MASKS is None else O.RELU_MASKS[bn])
    out["sr"] = (sr.detach(), sr.grad.detach())
    return out, {n: s[n].grad for n in O.param_names()}

m = sifsr.ModelB_2(2); m.load_state_dict(sd); m = m.cuda().train()
x = torch.cat((lst_up, ndvi), 1).cuda()
# run through the C ABI directly so that we keep the workspace
fp, fr, fn = m._flat_state(x.device)
H = W = 256
wsb = L.call("sifsr_model_workspace_bytes", B, H, W, 1)
ws = torch.empty(wsb // 4, dtype=torch.float32, device="cuda")
sr = torch.empty(B, 1, H, W, device="cuda")
S = torch.cuda.current_stream().cuda_stream
L.call("sifsr_model_forward", x, sr, fp, fr, fn, ws, wsb, B, H, W, 1, 0.1, 1e-5, S)
srr = sr.clone().requires_grad_(True)
_, _, loss = sifsr.sif_loss(kind, srr, lst.cuda(), ndvi.cuda(), MEAN, STD, alpha, gamma)
(dsr,) = torch.autograd.grad(loss, srr)
grads = torch.empty_like(fp)
L.call("sifsr_model_backward", x, dsr.contiguous(), fp, grads, ws, wsb, B, H, W, S)
torch.cuda.synchronize()
reg = (ctypes.c_size_t * 56)()
assert L.call("sifsr_model_workspace_regions", B, H, W, reg, 56) == 56
reg = list(reg)
tab = (ctypes.c_int * (17 * 8))(); L.call("sifsr_layer_table", tab, 17)
masks = {}
for l, (conv, bn, cin, cout) in enumerate(O.CONV_BN_LAYERS):
    lv, choff = tab[l * 8 + 2], tab[l * 8 + 7]; h = H >> lv
    y = ws[reg[l]:reg[l] + B * h * h * cout].view(B, h, h, cout)
    masks[bn] = ((y.double() * ws[reg[54] + choff:reg[54] + choff + cout].double() + ws[reg[55] + choff:reg[55] + choff + cout].double()) > 0).permute(0, 3, 1, 2).cpu()
if "--masks" in sys.argv: O.RELU_MASKS = masks
t32, g32 = oracle_tape(torch.float32)
t64, g64 = oracle_tape(torch.float64)
O.RELU_MASKS = None
print("dsr: hip vs64 %.2e | cpu32 vs64 %.2e" % (rel_err(dsr, t64["sr"][1]), rel_err(t32["sr"][1], t64["sr"][1])))
resB = {3: 0, 6: 1, 9: 2}
for l, (conv, bn, cin, cout) in enumerate(O.CONV_BN_LAYERS):
    lv = tab[l * 8 + 2]; h = H >> lv
    n = B * h * h * cout
    yoff = reg[l]
    goff = reg[26 + l] if l not in resB else reg[43 + resB[l]]
    y = ws[yoff:yoff + n].view(B, h, h, cout).permute(0, 3, 1, 2).cpu()
    dy = ws[goff:goff + n].view(B, h, h, cout).permute(0, 3, 1, 2).cpu()
    choff = tab[l * 8 + 7]
    sc = ws[reg[54] + choff: reg[54] + choff + cout].cpu(); sh = ws[reg[55] + choff: reg[55] + choff + cout].cpu()
    mh = torch.addcmul(sh[None, :, None, None], y, sc[None, :, None, None]) > 0
    print("   mask flips vs fp64: hip %d  cpu32 %d  (of %d)" % ((mh != t64[conv][2]).sum(), (t32[conv][2] != t64[conv][2]).sum(), mh.numel()), end=" ")
    print("%-36s y: hip %.1e cpu %.1e | dy: hip vs64 %.2e  cpu32 vs64 %.2e  hip vs cpu32 %.2e" % (
        conv, rel_err(y, t64[conv][0]), rel_err(t32[conv][0], t64[conv][0]),
        rel_err(dy, t64[conv][1]), rel_err(t32[conv][1], t64[conv][1]), rel_err(dy, t32[conv][1])))

off = 0
for n, p in m.named_parameters():
    gh = grads[off:off + p.numel()].view(p.shape).cpu(); off += p.numel()
    print("%-45s grad: hip vs64 %.2e  cpu32 vs64 %.2e" % (n, rel_max(gh, g64[n]), rel_max(g32[n], g64[n])))
