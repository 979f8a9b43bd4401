"""Time the thin-conv / fused head+tail entry points at full size (B=64, 256x256) through the C-ABI.
Usage (GPU box): python tools/bench_edges.py [NBLK]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sifsr
from sifsr import _lib as L

B, H, W = 64, 256, 256
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 768
S = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn(B, H, W, 16, device="cuda", generator=g)
gg = torch.randn(B, H, W, 16, device="cuda", generator=g)
x = torch.randn(B, 2, H, W, device="cuda", generator=g)
dsr = torch.randn(B, 1, H, W, device="cuda", generator=g)
sr = torch.empty(B, 1, H, W, device="cuda")
w_out = torch.randn(144, device="cuda") * 0.1
b_out = torch.zeros(1, device="cuda")
w_in = torch.randn(288, device="cuda") * 0.3
scale = torch.rand(16, device="cuda") + 0.5
shift = torch.randn(16, device="cuda") * 0.2
mean = torch.randn(16, device="cuda") * 0.1
invstd = torch.rand(16, device="cuda") + 0.5
scratch = torch.empty(64 + 1024 * 16 * 64 * 32, device="cuda")
dwb = torch.empty(145, device="cuda"); dw = torch.empty(288, device="cuda")
dgam = torch.empty(16, device="cuda"); dbet = torch.empty(16, device="cuda")
coef = torch.zeros(48, dtype=torch.float64, device="cuda")
dy = torch.empty_like(y)
part = torch.empty(B * 256 * 32, device="cuda")


def timeit(name, fn, mb, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:34s} {us:8.1f} us   {mb / us * 1e-0 / 1e3:6.2f} TB/s (algorithmic {mb:.0f} MB)")


T = y.numel() * 4 / 1e6   # one 16-channel tensor, MB
timeit("conv_out_bn_relu_bwd (fused tail)", lambda: L.call("sifsr_conv_out_bn_relu_bwd", y, scale, shift, mean, invstd, dsr, w_out,
       scratch, nblk, dwb, dgam, dbet, coef, dy, B, H, W, S()), 3 * T)
timeit("conv_in_bn_relu_bwd (fused head)", lambda: L.call("sifsr_conv_in_bn_relu_bwd", x, gg, y, scale, shift, mean, invstd,
       scratch, min(nblk, 1024), dw, dgam, dbet, coef, B, H, W, S()), 4 * T)
timeit("conv_in_fwd", lambda: L.call("sifsr_conv_in_fwd", x, w_in, dy, part, B, H, W, S()), T + T / 8)
timeit("conv_out_fwd", lambda: L.call("sifsr_conv_out_fwd", y, scale, shift, w_out, b_out, sr, B, H, W, S()), T + T / 16)
timeit("bn_relu_bwd C=16", lambda: L.call("sifsr_bn_relu_bwd", gg, y, scale, shift, mean, invstd, 16, B * H * W, scratch, 1024,
       dgam, dbet, coef, dy, None, 0, 0, S()), 5 * T)
