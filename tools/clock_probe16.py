"""Where a tile's time goes in the fused 16 -> 16 backward kernel (conv_bwd16.hip), per wave role: shader-clock cycles spent
staging (incl. the wait for the tile's loads), contracting (+ epilogue) and waiting at the per-tile barrier.
  bash tools/build_ab.sh clk16 -DSIFSR_DIAG_CLOCK
  SIFSR_LIB=$PWD/tools/ab/libsifsr_clk16.so python tools/clock_probe16.py [H]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
from sifsr import _lib as L
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, dev = 64, "cuda"
torch.manual_seed(0)
x = torch.randn(B, H, H, 16, device=dev); sc = torch.rand(16, device=dev) + 0.5; sh = torch.randn(16, device=dev) * 0.3
w = torch.randn(16, 16, 3, 3, device=dev) * (2.0 / 144) ** 0.5
S = torch.cuda.current_stream().cuda_stream
wf = torch.empty(9 * 256, device=dev); wd = torch.empty(36 * 256, device=dev); wwf = torch.empty(16 * 256, device=dev); wwd = torch.empty(16 * 256, device=dev)
L.call("sifsr_pack_conv_weights", w, 16, 16, wf, wd, S); L.call("sifsr_pack_conv_weights_wino", w, 16, 16, wwf, wwd, S)
dy = torch.randn(B, H, H, 16, device=dev); ycur = torch.randn(B, H, H, 16, device=dev); coef = torch.randn(64, device=dev) * 0.1 + 0.5
border = torch.empty(B, H, H, 16, device=dev); g = torch.empty(B, H, H, 16, device=dev); dw = torch.empty_like(w)
bnp = torch.empty(L.call("sifsr_conv3x3_bwd16_stat_rows", B, H, H) * 32, device=dev)
scratch = torch.empty(L.call("sifsr_conv3x3_bwd16_scratch_floats", B, H, H), device=dev)
run = lambda: L.call("sifsr_conv3x3_bwd16", x, sc, sh, dy, ycur, coef, border, wd, wwd, g, None, x, sc, sh, bnp, scratch, dw, B, H, H, S)
h = L.lib(); out = (ctypes.c_ulonglong * 10)()
for _ in range(5): run()
torch.cuda.synchronize(); h.sifsr_debug_timers16(out, 1)
for _ in range(20): run()
torch.cuda.synchronize(); h.sifsr_debug_timers16(out, 1)
for r, name in ((0, "input-gradient role "), (4, "weight-gradient role")):
    n = max(1, out[r + 3])
    print(f"{name}: of the staging time, waiting for the prefetched loads: {out[8 + r // 4] / n:8.0f}")
    print(f"{name}: per tile {out[r] / n:8.0f} cycles staging (+ load wait), {out[r + 1] / n:8.0f} contracting, {out[r + 2] / n:8.0f} at the barrier"
          f"   ({n // 20} tile iterations per launch; ticks of s_memtime -- use the ratios)")
