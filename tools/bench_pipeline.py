"""Timing of the rows next to the hot path: a whole 1200x1200 MODIS granule through predict_granule (324 tiles,
predict.py:84-103) and the per-batch PSNR/SSIM (utils.py:548-578) at batch 64.  GPU box: python tools/bench_pipeline.py"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
dev = torch.device("cuda", 0)
stats = dict(sifsr.dataset.DEFAULT_STATS)
torch.manual_seed(0)
model = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev).eval()
lst_g = torch.randn(1200, 1200, device=dev) * 5.5 + 307
ndvi_g = (torch.randn(4800, 4800, device=dev) * 0.3 + 0.6)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for batch in (162, 324):
    ms = timeit(lambda: sifsr.predict.predict_granule(model, lst_g, ndvi_g, stats, batch=batch))
    print(f"predict_granule 1200x1200 -> 4800x4800 (324 tiles, batch {batch}): {ms:.2f} ms  = {324 / ms * 1e3:.0f} tiles/s")
x, tiles = sifsr.pipeline.granule_to_tiles(lst_g, ndvi_g, stats)
ms = timeit(lambda: sifsr.pipeline.granule_to_tiles(lst_g, ndvi_g, stats), 20)
print(f"granule_to_tiles alone: {ms*1e3:.0f} us ({x.numel()*4/1e6:.0f} MB written)")
p = torch.randn(64, 1, 256, 256, device=dev); t = p + 0.1 * torch.randn_like(p)
ms = timeit(lambda: sifsr.metrics.psnr_ssim(p, t), 20)
print(f"psnr_ssim batch 64: {ms*1e3:.0f} us")
