"""Sweep the number of partial-sum workgroups of the BatchNorm-backward reduce at full size (B=64, 256x256x16)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
from sifsr import _lib as L
B, H, W, C = 64, 256, 256, 16
S = lambda: torch.cuda.current_stream().cuda_stream
y = torch.randn(B, H, W, C, device="cuda"); g = torch.randn(B, H, W, C, device="cuda"); dy = torch.empty_like(y)
scale = torch.rand(C, device="cuda") + 0.5; shift = torch.randn(C, device="cuda") * 0.2
mean = torch.randn(C, device="cuda") * 0.1; invstd = torch.rand(C, device="cuda") + 0.5
dgam = torch.empty(C, device="cuda"); dbet = torch.empty(C, device="cuda")
coef = torch.zeros(3 * C, dtype=torch.float64, device="cuda")
scratch = torch.empty(16384 * C * 2, device="cuda")
for nblk in (256, 512, 1024, 2048, 4096, 8192):
    fn = lambda: L.call("sifsr_bn_relu_bwd", g, y, scale, shift, mean, invstd, C, B * H * W, scratch, nblk, dgam, dbet, coef, dy, None, 0, 0, S())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"nblk {nblk:5d}: reduce+finalize+apply {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
