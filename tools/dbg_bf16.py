import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F, sifsr
from sifsr import _lib as L
rs = np.random.RandomState(0)
B, cin, cout, H, W = 2, 16, 16, 32, 32
x = torch.from_numpy(rs.standard_normal((B, cin, H, W)).astype(np.float32))
w = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32))
rb = lambda t: t.to(torch.bfloat16).float()
conv = lambda a, b: F.conv2d(F.pad(a, (1, 1, 1, 1), mode="replicate"), b)
y32, y16 = conv(x, w), conv(rb(x), rb(w))
S = torch.cuda.current_stream().cuda_stream
wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda")
L.call("sifsr_pack_conv_weights", w.cuda(), cin, cout, wf, wd, S)
xd = x.permute(0, 2, 3, 1).contiguous().cuda()
y = torch.empty(B, H, W, cout, device="cuda")
L.call("sifsr_conv3x3_fwd_bf16", xd, cin, None, None, None, 0, None, None, wd, y, cout, None, B, H, W, S)
torch.cuda.synchronize()
yy = y.permute(0, 3, 1, 2).cpu()
e = lambda a, b: float((a - b).abs().max() / b.abs().max())
print("hip bf16 vs emul", e(yy, y16), " vs fp32", e(yy, y32), " emul vs fp32", e(y16, y32))
# check the bf16 pack directly
n = 9 * cin * cout
h = wd[n:].view(torch.bfloat16)[:n].float().cpu()
print("bf16 fwd pack vs rounded fp32 pack:", float((h - rb(wf.cpu())).abs().max()))
