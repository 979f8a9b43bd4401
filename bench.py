#!/usr/bin/env python3
"""Headline benchmark: LST+NDVI 256x256 training patches/s (BASELINE.json), SR2 step at batch 64/GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
cat(lst_up, ndvi) -> ModelB_2 forward -> SIF loss (SR2) -> backward -> [gradient all-reduce] -> Adam
(train_model_B_gradFTM.py:94-121), through the HIP path only.  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline     -- the dominant kernel (selected with --roofline-kernel, default the 16->16 256^2
                  forward conv) timed with HIP events on its launch stream INSIDE the timed steps
                  (sifsr_profile_*), algorithmic FLOPs / average duration vs the fp32 MFMA peak;
                  `step` carries the same ratio for the whole step (SURVEY.md §8 d FLOPs per patch).
  cpu_baseline -- the oracle (CPU restatement of the reference, kind "port") timed on this box's
                  host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / IPC on this driver needs dmabuf handles (before any HIP init)

import torch

TRAIN_FLOPS_PER_PATCH = 10_777_264_128      # SURVEY.md §8 d (conv MACs x2: fwd + dgrad + wgrad)
TRAIN_BYTES_PER_PATCH = 195_821_568
PEAK_FP32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32
PEAK_HBM_GBS = 8000.0

# name -> (layer index in the engine table, phase, algorithmic FLOPs per patch)
#   phase 1 = forward conv, 2 = dgrad, 3 = wgrad.  2*9*Cin*Cout*H*W per patch.
ROOFLINE_KERNELS = {
    "fwd_16x16_256": (1, 1, 2 * 9 * 16 * 16 * 256 * 256),     # inbloc.bloc.3 forward
    "dgrad_16x16_256": (16, 2, 2 * 9 * 16 * 16 * 256 * 256),  # ub3.convbloc.bloc.3 dgrad
    "wgrad_16x16_256": (16, 3, 2 * 9 * 16 * 16 * 256 * 256),  # ub3.convbloc.bloc.3 wgrad
    "fwd_32x16_256": (15, 1, 2 * 9 * 32 * 16 * 256 * 256),    # ub3.convbloc.bloc.0 forward
    "wgrad_32x16_256": (15, 3, 2 * 9 * 32 * 16 * 256 * 256),
    "dgrad_32x16_256": (15, 2, 2 * 9 * 32 * 16 * 256 * 256),
}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, written by
    tools/summarize_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    for f in reversed(files):
        try:
            t = json.load(open(f))["per_launch"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if t:
            return round(t["bytes"]), os.path.basename(f)
    return None, None


def cpu_baseline(kind, alpha, gamma, lr, mean, std, target_seconds=15.0):
    """Time the oracle's train step (fwd + loss + bwd + Adam) on the host cores; bounded sample."""
    from oracle import sif_oracle as O
    cores = int(os.environ.get("SIFSR_CPU_THREADS", min(16, os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    B = 8
    sd = O.synthetic_state(0)
    lst, lst_up, ndvi = O.synthetic_batch(1234, B)
    adam = O.AdamState(O.param_names(), lr)
    O.train_step(sd, adam, lst, lst_up, ndvi, mean, std, alpha, gamma, kind)       # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < target_seconds and n < 40):
        O.train_step(sd, adam, lst, lst_up, ndvi, mean, std, alpha, gamma, kind)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"{n} SR2 train steps (fwd+loss+bwd+Adam) at batch {B}, 256x256, fp32, torch CPU, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="patches per GPU per step")
    ap.add_argument("--kind", default="sr2", choices=["sr2", "sr1"])
    ap.add_argument("--roofline-kernel", default="fwd_16x16_256", choices=sorted(ROOFLINE_KERNELS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16x3"],
                    help="f32 = the headline fp32 path; bf16 = BASELINE.json config 5 (bf16 MFMA operands in the 3x3 conv "
                         "forward / input-gradient / weight-gradient passes, fp32 accumulation and storage); bf16x3 = fp32 on the bf16 "
                         "matrix cores (exact three-way bf16 split of every conv operand, six cross products; fp32-level results) "
                         "-- neither is the default")
    args = ap.parse_args()

    import sifsr
    from sifsr import _lib as L
    from sifsr import distributed as dp

    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    kind = args.kind
    alpha, gamma, lr = (0.5, -0.25, 1e-4) if kind == "sr2" else (0.99, -0.5, 1e-3)
    stats = dict(sifsr.dataset.DEFAULT_STATS)
    torch.manual_seed(0)
    model = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
    model.compute_dtype = {"f32": "fp32", "bf16": "bf16", "bf16x3": "bf16x3"}[args.dtype]
    opt = sifsr.FlatAdam(model.parameters(), lr=lr)
    lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(args.batch, dev, seed=1234 + rank)

    def step():
        return sifsr.train.train_step(model, opt, lst, lst_up, ndvi, stats, alpha, gamma, kind)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    layer, phase, kflops = ROOFLINE_KERNELS[args.roofline_kernel]
    L.call("sifsr_profile_select", layer, phase)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    fence()
    dt = time.perf_counter() - t0
    import ctypes
    kms, kcount = ctypes.c_float(0), ctypes.c_int(0)
    L.call("sifsr_profile_read", ctypes.byref(kms), ctypes.byref(kcount))
    L.call("sifsr_profile_select", -1, 0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert all(torch.isfinite(v) for v in losses), "non-finite loss"

    patches = args.batch * world * args.steps
    value = patches / dt
    per_gpu = value / world
    out = {
        "metric": "LST+NDVI 256x256 patches/sec (train fwd+bwd), whole job",
        "value": round(value, 2), "unit": "patches/s", "per_gpu": round(per_gpu, 2),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16 conv operands (fwd, dgrad, wgrad), f32 accumulate/storage",
                  "bf16x3": "f32 as 3-term bf16 splits (conv fwd, dgrad: 6 bf16 MFMA cross products), f32 accumulate/storage; wgrad f32 MFMA"}[args.dtype],
        "data": "synthetic",
        "config": {"workload": f"ModelB SIF-NN-{kind.upper()} ({'gradFTM' if kind == 'sr2' else 'predef_filters'} loss) "
                               f"batch {args.batch}/GPU, synthetic 256x256, {world}x MI355X, fwd+loss+bwd+Adam",
                   "batch_per_gpu": args.batch, "patch": "256x256 (LST 64x64 + NDVI 256x256)",
                   "parallelism": f"dp{world}", "loss": kind},
    }
    if rank == 0:
        kavg_ms = kms.value / max(1, kcount.value)
        kt = kflops * args.batch / (kavg_ms * 1e-3) / 1e12 if kavg_ms > 0 else 0.0
        step_tf = TRAIN_FLOPS_PER_PATCH * per_gpu / 1e12
        traffic, traffic_src = measured_traffic(args.roofline_kernel) if args.dtype == "f32" else (None, None)
        if args.dtype == "f32":
            out["roofline"] = {
                "bound": "mfma", "kernel": args.roofline_kernel, "achieved": round(kt, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(kt / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes/launch (PMC)", "traffic_source": traffic_src,
                "kernel_avg_ms": round(kavg_ms, 4), "kernel_launches_timed": kcount.value,
                "step": {"achieved": round(step_tf, 2), "frac": round(step_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                         "hbm_GBs_algorithmic": round(TRAIN_BYTES_PER_PATCH * per_gpu / 1e9, 1)},
            }
        else:
            # bf16 operands: the matrix-core peak rises 16x, the bytes do not change (fp32 storage) -> HBM-bound
            # (SURVEY.md §8 d).  Algorithmic bytes of the selected conv launch = its input + output activations.
            cin, cout = {"16x16": (16, 16), "32x16": (32, 16)}[args.roofline_kernel.split("_")[1]]
            kbytes = (cin + cout) * 256 * 256 * 4 * args.batch
            gbs = kbytes / (kavg_ms * 1e-3) / 1e9 if kavg_ms > 0 else 0.0
            out["roofline"] = {
                "bound": "hbm", "kernel": args.roofline_kernel, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "kernel_avg_ms": round(kavg_ms, 4),
                "kernel_launches_timed": kcount.value,
                "step": {"hbm_GBs_algorithmic": round(TRAIN_BYTES_PER_PATCH * per_gpu / 1e9, 1),
                         "frac": round(TRAIN_BYTES_PER_PATCH * per_gpu / 1e9 / PEAK_HBM_GBS, 4)},
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, alpha, gamma, lr, stats["mean_lst"], stats["std_lst"])
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
