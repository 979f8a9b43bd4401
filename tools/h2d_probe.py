"""Host-to-device bandwidth of the box for the batch tensors of one training step (pinned memory, async copies)."""
import time, torch
ts = [torch.randn(64, 1, 64, 64).pin_memory(), torch.randn(64, 1, 256, 256).pin_memory(), torch.randn(64, 1, 256, 256).pin_memory()]
ds = [torch.empty_like(t, device="cuda") for t in ts]
nbytes = sum(t.numel() * 4 for t in ts)
for _ in range(3):
    for d, h in zip(ds, ts): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    for d, h in zip(ds, ts): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"{nbytes / 1e6:.1f} MB per step in {dt * 1e3:.2f} ms = {nbytes / dt / 1e9:.1f} GB/s (pinned, async, nothing else running)")
big = torch.empty(256 << 20, dtype=torch.uint8).pin_memory(); dbig = torch.empty_like(big, device="cuda")
dbig.copy_(big, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter(); dbig.copy_(big, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"256 MiB in one copy: {big.numel() / dt / 1e9:.1f} GB/s")
