"""Layer-by-layer forward parity of the bf16-operand mode (training forward): raw conv outputs y_l from the HIP
workspace vs the oracle's bf16-operand emulation and vs the fp32 oracle."""
import sys, os, copy, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sifsr
from sifsr import _lib as L
from oracle import sif_oracle as O
B, H, W = 2, 256, 256
sd = O.synthetic_state(41); lst, lst_up, ndvi = O.synthetic_batch(43, B)
x = torch.cat((lst_up, ndvi), 1)

def tape(bf16):
    t = {}
    orig = O._conv3x3_rep
    def rec(a, w, b=None):
        y = orig(a, w, b); t.setdefault(id(w), y.detach()); return y
    O._conv3x3_rep = rec; O.BF16_CONVS = bf16
    s = copy.deepcopy(sd)
    try:
        O.modelb2_forward(s, x, True)
    finally:
        O._conv3x3_rep = orig; O.BF16_CONVS = False
    return {conv: t[id(s[conv + ".weight"])] for conv, bn, ci, co in O.CONV_BN_LAYERS if id(s[conv + ".weight"]) in t}, s

m = sifsr.ModelB_2(2); m.load_state_dict(sd); m = m.cuda().train()
fp, fr, fn = m._flat_state(torch.device("cuda", 0))
wsb = L.call("sifsr_model_workspace_bytes", B, H, W, 1)
ws = torch.empty(wsb // 4, dtype=torch.float32, device="cuda")
sr = torch.empty(B, 1, H, W, device="cuda")
S = torch.cuda.current_stream().cuda_stream
L.call("sifsr_model_forward_ex", x.cuda(), sr, fp, fr, fn, ws, wsb, B, H, W, 1, 0.1, 1e-5, 1, S)
torch.cuda.synchronize()
reg = (ctypes.c_size_t * 56)(); L.call("sifsr_model_workspace_regions", B, H, W, reg, 56)
tab = (ctypes.c_int * (17 * 8))(); L.call("sifsr_layer_table", tab, 17)
e = lambda a, b: float((a - b).abs().max() / b.abs().max())
# the rounded conv in the emulation is an autograd Function: hook by output instead
import torch.nn.functional as F
def tape2(bf16):
    outs = []
    orig = O._bn_relu
    def rec(xx, s, bn, training):
        outs.append((bn, xx.detach())); return orig(xx, s, bn, training)
    O._bn_relu = rec; O.BF16_CONVS = bf16
    try:
        O.modelb2_forward(copy.deepcopy(sd), x, True)
    finally:
        O._bn_relu = orig; O.BF16_CONVS = False
    return dict(outs)
t16, t32 = tape2(True), tape2(False)
for l, (conv, bn, cin, cout) in enumerate(O.CONV_BN_LAYERS):
    lv = tab[l * 8 + 2]; h, w = H >> lv, W >> lv
    y = ws[reg[l]:reg[l] + B * h * w * cout].view(B, h, w, cout).permute(0, 3, 1, 2).cpu()
    print("%-36s hip vs emul %.2e | hip vs fp32 %.2e | emul vs fp32 %.2e" % (conv, e(y, t16[bn]), e(y, t32[bn]), e(t16[bn], t32[bn])))
