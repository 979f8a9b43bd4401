"""Micro-benchmark of single conv ops through the C ABI (for PMC / tuning).
usage: python tools/bench_conv.py [fwd|dgrad|wgrad|wgradx|bwd16] [cin] [cout] [H] [B] [iters]   (wgradx = Winograd F(3x3,2x2) weight gradient;
       bwd16 = input + weight gradient of a 16 -> 16 layer in one kernel, with dL/dy formed on load and the fused BatchNorm sums)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
from sifsr import _lib as L
op = sys.argv[1] if len(sys.argv) > 1 else "fwd"
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 16
H = int(sys.argv[4]) if len(sys.argv) > 4 else 256
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
mode = sys.argv[7] if len(sys.argv) > 7 else "wino"      # wino (fp32, what the model runs) | fp32 (tap-domain kernel) | bf16
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, H, H, cin, device=dev)
sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.3
w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
wf = torch.empty(9 * cin * cout, device=dev); wd = torch.empty(4 * 9 * cin * cout, device=dev)
S = torch.cuda.current_stream().cuda_stream
L.call("sifsr_pack_conv_weights", w, cin, cout, wf, wd, S)
wwf = torch.empty(16 * cin * cout, device=dev); wwd = torch.empty(16 * cin * cout, device=dev)
L.call("sifsr_pack_conv_weights_wino", w, cin, cout, wwf, wwd, S)
y = torch.empty(B, H, H, cout, device=dev)
part = torch.empty(B * (H // 16) * (H // 16) * cout * 2, device=dev)
dy = torch.randn(B, H, H, cout, device=dev)
g = torch.empty(B, H, H, cin, device=dev)
nblk = int(os.environ.get("NBLK", 2048 if 9 * cin * cout <= 4608 else 1024))
scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_scratch_floats", cin, cout, nblk), device=dev)
dw = torch.empty_like(w)
if op == "wgradx":
    scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_wino_scratch_floats", cin, cout, nblk), device=dev)
if op == "bwd16":
    assert cin == 16 and cout == 16
    ycur = torch.randn(B, H, H, 16, device=dev); coef = torch.randn(64, device=dev) * 0.1 + 0.5
    border = torch.empty(B, H, H, 16, device=dev)
    rows = L.call("sifsr_conv3x3_bwd16_stat_rows", B, H, H)
    bnp = torch.empty(rows * 32, device=dev)
    scratch = torch.empty(L.call("sifsr_conv3x3_bwd16_scratch_floats", B, H, H), device=dev)
def run():
    if op == "bwd16":
        L.call("sifsr_conv3x3_bwd16", x, sc, sh, dy, ycur, coef, border, wd, wwd, g, None, x, sc, sh, bnp, scratch, dw, B, H, H, S)
    elif op == "fwd" and mode == "wino":
        L.call("sifsr_conv3x3_fwd_wino", x, cin, sc, sh, None, 0, None, None, wf, wwf, y, cout, part, B, H, H, S)
    elif op == "dgrad" and mode == "wino":
        L.call("sifsr_conv3x3_dgrad_wino", dy, cout, wd, wwd, cin, g, cin, None, 0, None, B, H, H, S)
    elif op == "fwd" and mode == "fp32":
        L.call("sifsr_conv3x3_fwd", x, cin, sc, sh, None, 0, None, None, wf, y, cout, part, B, H, H, S)
    elif op == "fwd":
        L.call("sifsr_conv3x3_fwd_" + mode, x, cin, sc, sh, None, 0, None, None, wd, y, cout, part, B, H, H, S)
    elif op == "dgrad" and mode == "fp32":
        L.call("sifsr_conv3x3_dgrad", dy, cout, wd, w, cin, g, cin, None, 0, None, B, H, H, S)
    elif op == "dgrad":
        L.call("sifsr_conv3x3_dgrad_" + mode, dy, cout, wd, cin, g, cin, None, 0, None, B, H, H, S)
    elif op == "wgradx":
        L.call("sifsr_conv3x3_wgrad_wino", x, cin, sc, sh, None, 0, None, None, dy, None, None, cout, scratch, nblk, dw, B, H, H, S)
    else:
        L.call("sifsr_conv3x3_wgrad", x, cin, sc, sh, None, 0, None, None, dy, cout, scratch, nblk, dw, B, H, H, S)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2 * 9 * cin * cout * H * H * B * (2 if op == "bwd16" else 1)
print(f"[{mode}] {op} {cin}->{cout} @{H}^2 B={B}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.1f}% of fp32 MFMA peak)")
