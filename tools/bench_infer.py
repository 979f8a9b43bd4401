"""BASELINE.json config 4: eval forward, batch 256 full tiles 256x256, 1x MI355X, hipGraph-captured.
Prints tiles/s (eager and graph replay) and the forward FLOP fraction of the fp32 MFMA peak."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
it = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
m = sifsr.ModelB_2(2).cuda().eval()
stats = dict(sifsr.dataset.DEFAULT_STATS)
lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(B, "cuda")
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it
te = timeit(lambda: sifsr.predict.predict_tiles(m, lst_up, ndvi, stats, batch=B))
gp = sifsr.predict.GraphedPredictor(m, B, stats)
tg = timeit(lambda: gp(lst_up, ndvi))
fl = 3_605_004_288 * B
print(f"inference B={B}: eager {B/te:.0f} tiles/s ({te*1e3:.2f} ms), graph {B/tg:.0f} tiles/s ({tg*1e3:.2f} ms), "
      f"fwd {fl/tg/1e12:.1f} TFLOP/s = {fl/tg/1e12/157.3*100:.1f}% of fp32 MFMA peak")
for b in (1, 8):
    gp1 = sifsr.predict.GraphedPredictor(m, b, stats)
    t1e = timeit(lambda: sifsr.predict.predict_tiles(m, lst_up[:b], ndvi[:b], stats, batch=b))
    t1g = timeit(lambda: gp1(lst_up[:b], ndvi[:b]))
    print(f"inference B={b}: eager {t1e*1e3:.3f} ms/call, graph {t1g*1e3:.3f} ms/call")
