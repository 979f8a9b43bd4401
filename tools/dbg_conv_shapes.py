import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import sifsr
from sifsr import _lib as L
from tests.test_ops_gpu import dev, nhwc, nchw, S, rnd, conv_rep
rs = np.random.RandomState(0)
for (cin, cout, H, W, B) in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or ((16,16,256,256,2),):
    x = rnd(rs, B, cin, H, W); w = rnd(rs, cout, cin, 3, 3, scale=(2.0/(9*cin))**0.5)
    y_ref = conv_rep(x, w)
    wf = torch.empty(9*cin*cout, device="cuda"); wd = torch.empty(2*9*cin*cout, device="cuda")
    L.call("sifsr_pack_conv_weights", dev(w), cin, cout, wf, wd, S())
    y = torch.full((B,H,W,cout), float("nan"), device="cuda")
    L.call("sifsr_conv3x3_fwd", dev(nhwc(x)), cin, None, None, None, 0, None, None, wf, y, cout, None, B, H, W, S())
    torch.cuda.synchronize()
    d = (nchw(y.cpu()) - y_ref).abs()
    bad = (d > 1e-3) | torch.isnan(d)
    print(cin, cout, H, W, B, "maxerr", float(d.nan_to_num(9).max()), "bad px", int(bad.any(1).sum()))
    if bad.any():
        idx = bad.any(1).nonzero()
        print("   bad b,y,x range:", idx.min(0).values.tolist(), idx.max(0).values.tolist(), " cols:", sorted(set(idx[:,2].tolist()))[:20], " rows:", sorted(set(idx[:,1].tolist()))[:20])
