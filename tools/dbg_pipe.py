"""Debugging aid: where do the Winograd conv results (forward / input gradient) differ from torch?  python tools/dbg_pipe.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F, sifsr
from sifsr import _lib as L
torch.manual_seed(0)


def show(tag, got, ref):
    err = (got - ref).abs()
    bad = err > 1e-3
    print(f"{tag}: bad {int(bad.sum())} / {bad.numel()}  max {float(err.max()):.3g}")
    if bad.any():
        idx = bad.nonzero()
        print("  bad b:", sorted(set(idx[:, 0].tolist()))[:8], " c:", sorted(set(idx[:, 1].tolist()))[:20])
        print("  bad y:", sorted(set(idx[:, 2].tolist()))[:40])
        print("  bad x:", sorted(set(idx[:, 3].tolist()))[:40])


def run(cin, cout, H, W, B):
    x = torch.randn(B, cin, H, W); w = torch.randn(cout, cin, 3, 3) * 0.1
    S = torch.cuda.current_stream().cuda_stream
    wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights", w.cuda(), cin, cout, wf, wd, S)
    wwf = torch.empty(16 * cin * cout, device="cuda"); wwd = torch.empty(16 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights_wino", w.cuda(), cin, cout, wwf, wwd, S)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.full((B, H, W, cout), float("nan"), device="cuda")
    nblk = L.call("sifsr_conv3x3_stat_blocks_wino", B, H, W, cin, cout)
    part = torch.empty(max(nblk, 1) * cout * 2, device="cuda")
    L.call("sifsr_conv3x3_fwd_wino", xd, cin, None, None, None, 0, None, None, wf, wwf, y, cout, part, B, H, W, S)
    torch.cuda.synchronize()
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), w)
    show(f"fwd {cin}->{cout} {H}x{W} B={B}", y.cpu().permute(0, 3, 1, 2), ref)
    dy = torch.randn(B, cout, H, W)
    g = torch.full((B, H, W, cin), float("nan"), device="cuda")
    L.call("sifsr_conv3x3_dgrad_wino", dy.permute(0, 2, 3, 1).contiguous().cuda(), cout, wd, wwd, cin, g, cin, None, 0, None, B, H, W, S)
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    (F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="replicate"), w) * dy).sum().backward()
    show(f"dgrad {cin}<-{cout}", g.cpu().permute(0, 3, 1, 2), xr.grad)


for cfg in [(16, 16, 16, 16, 1), (16, 32, 16, 16, 1), (32, 32, 32, 32, 1), (64, 64, 32, 32, 2)]:
    run(*cfg)
