"""Eager vs hipGraph-replayed training step at small batch sizes.  GPU box: python tools/bench_graph_step.py"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
dev = torch.device("cuda", 0)
stats = dict(sifsr.dataset.DEFAULT_STATS)
for B in (1, 4, 16, 64):
    res = []
    for graphed in (False, True):
        torch.manual_seed(0)
        m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
        opt = sifsr.FlatAdam(m.parameters(), lr=1e-4, capturable=graphed)
        lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(B, dev, seed=1)
        st = sifsr.train.GraphedTrainStep(m, opt, B, stats, 0.5, -0.25) if graphed else None
        f = (lambda: st(lst, lst_up, ndvi)) if graphed else (lambda: sifsr.train.train_step(m, opt, lst, lst_up, ndvi, stats, 0.5, -0.25))
        for _ in range(6): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 40
        for _ in range(n): f()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / n * 1e3)
    print(f"batch {B:3d}: eager {res[0]:.3f} ms/step ({B / res[0] * 1e3:.0f} patches/s) | graph replay {res[1]:.3f} ms/step ({B / res[1] * 1e3:.0f} patches/s)")
