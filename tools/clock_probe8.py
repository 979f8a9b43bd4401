"""Where an item of the eight-wave Winograd kernel (conv_wino8.hip) goes: shader-clock ticks per phase and wave team, from the
-DSIFSR_DIAG_CLOCK build.  The stamps serialise the phases (s_memtime waits for the LDS reads before it), so read the ratios.
  bash tools/build_ab.sh clk -DSIFSR_DIAG_CLOCK
  SIFSR_LIB=$PWD/tools/ab/libsifsr_clk.so python tools/clock_probe8.py fwd 64 32 128"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
h = L.lib(); out = (ctypes.c_ulonglong * 12)()
for _ in range(10): run()
torch.cuda.synchronize(); h.sifsr_debug_timers8(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize(); h.sifsr_debug_timers8(out, 1)
print(f"{op} {cin}->{cout} @{H}^2: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (stamped build)")
for r, name in ((0, "contract-first team (waves 0-3)"), (6, "stage-first team    (waves 4-7)")):
    n = max(1, out[r + 5])
    v = [out[r + i] / n for i in range(5)]
    print(f"{name}: per item {v[0]:7.0f} staging | {v[1]:7.0f} window reads | {v[2]:7.0f} input transform | {v[3]:7.0f} MFMAs + output transform | "
          f"{v[4]:7.0f} barrier   = {sum(v):7.0f} ticks   ({n // 20} items per launch and reporting wave)")
