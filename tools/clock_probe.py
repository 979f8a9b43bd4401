"""In-kernel clock of the eight-wave Winograd kernel (debugging build with s_memtime / s_memrealtime stamps around the item loop):
shader cycles per 100 MHz tick, summed over workgroups.  MI355X_MICROARCH.md, 'DVFS give-back' item 6.
  bash tools/build_ab.sh clk -DSIFSR_DIAG_CLOCK
  SIFSR_LIB=$PWD/tools/ab/libsifsr_clk.so python tools/clock_probe.py fwd 64 32 128
Round 2: 2.34-2.38 GHz (forward 64->32 @128^2, 128->64 @64^2, input gradient 32<-64 @128^2) -- the chip does not hold its clock
down under these kernels; after >= 2 s of back-to-back launches on the same tensors a launch takes 204 us where the 10-launch
micro-benchmark (tools/bench_conv.py) reads 246 us."""
import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
from sifsr import _lib as L
op, cin, cout, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = 64
x = torch.randn(B, H, H, cin, device="cuda"); sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda") * 0.3
w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
S = torch.cuda.current_stream().cuda_stream
wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(36 * cin * cout, device="cuda")
L.call("sifsr_pack_conv_weights", w, cin, cout, wf, wd, S)
wwf = torch.empty(16 * cin * cout, device="cuda"); wwd = torch.empty(16 * cin * cout, device="cuda")
L.call("sifsr_pack_conv_weights_wino", w, cin, cout, wwf, wwd, S)
y = torch.empty(B, H, H, cout, device="cuda"); part = torch.empty(B * (H // 16) ** 2 * cout * 2, device="cuda")
dy = torch.randn(B, H, H, cout, device="cuda"); g = torch.empty(B, H, H, cin, device="cuda")
def run():
    if op == "fwd": L.call("sifsr_conv3x3_fwd_wino", x, cin, sc, sh, None, 0, None, None, wf, wwf, y, cout, part, B, H, H, S)
    else: L.call("sifsr_conv3x3_dgrad_wino", dy, cout, wd, wwd, cin, g, cin, None, 0, None, B, H, H, S)
h = L.lib(); out = (ctypes.c_ulonglong * 4)()
t_end = torch.cuda.Event(enable_timing=True); t0 = torch.cuda.Event(enable_timing=True)
for _ in range(20): run()
torch.cuda.synchronize()
import time
t = time.time()
while time.time() - t < 2.5: run()          # >= 2 s of back-to-back launches before the stamped ones
torch.cuda.synchronize(); h.sifsr_debug_timers(out, 1)
t0.record()
for _ in range(50): run()
t_end.record(); torch.cuda.synchronize(); h.sifsr_debug_timers(out, 1)
print(f"{op} {cin}->{cout} @{H}^2: {t0.elapsed_time(t_end) / 50 * 1e3:.1f} us per launch; in-kernel clock {out[0] / out[1] * 0.1:.3f} GHz")
