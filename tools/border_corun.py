"""Micro-experiment: duration of the dgrad border kernel alone vs beside a weight-gradient kernel on another stream.
GPU box: rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/border_corun.py ; then tools/trace_overlap.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
from sifsr import _lib as L
dev = "cuda"
B, H, cin, cout = 64, 256, 16, 16
torch.manual_seed(0)
x = torch.randn(B, H, H, cin, device=dev); dy = torch.randn(B, H, H, cout, device=dev)
sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.3
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
wf = torch.empty(9 * cin * cout, device=dev); wd = torch.empty(4 * 9 * cin * cout, device=dev)
g = torch.empty(B, H, H, cin, device=dev)
nblk = 1024
scratch = torch.empty(L.call("sifsr_conv3x3_wgrad_scratch_floats", cin, cout, nblk), device=dev)
dw = torch.empty_like(w)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=0)
S = lambda s: s.cuda_stream
L.call("sifsr_pack_conv_weights", w, cin, cout, wf, wd, S(torch.cuda.current_stream()))
torch.cuda.synchronize()
def dgrad(s): L.call("sifsr_conv3x3_dgrad", dy, cout, wd, w, cin, g, cin, None, 0, None, B, H, H, S(s))
def wgrad(s): L.call("sifsr_conv3x3_wgrad", x, cin, sc, sh, None, 0, None, None, dy, cout, scratch, nblk, dw, B, H, H, S(s))
for rep in range(3):       # phase A: alone
    dgrad(s1); torch.cuda.synchronize()
for rep in range(3):       # phase B: wgrad x2 on s2 first, then dgrad on s1
    [wgrad(s2) for _ in range(6)]; [dgrad(s1) for _ in range(3)]; torch.cuda.synchronize()
print("done")
