"""Which host-side ops issue the small device copies seen in the profile?  (GPU box)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sifsr
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
opt = sifsr.FlatAdam(model.parameters(), lr=1e-4)
stats = dict(sifsr.dataset.DEFAULT_STATS)
lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(8, dev, seed=1)
for _ in range(3):
    sifsr.train.train_step(model, opt, lst, lst_up, ndvi, stats, 0.5, -0.25, "sr2")
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    sifsr.train.train_step(model, opt, lst, lst_up, ndvi, stats, 0.5, -0.25, "sr2")
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
ev = [e for e in prof.events() if "copy" in e.name.lower() or "Memcpy" in e.name]
import collections
c = collections.Counter((e.name, str(e.input_shapes)[:80]) for e in ev)
for k, v in c.most_common(20):
    print(v, k)
