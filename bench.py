#!/usr/bin/env python3
"""Headline benchmark: LST+NDVI 256x256 training patches/s (BASELINE.json), SR2 step at batch 64/GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
cat(lst_up, ndvi) -> ModelB_2 forward -> SIF loss (SR2) -> backward -> [gradient all-reduce] -> Adam
(train_model_B_gradFTM.py:94-121), through the HIP path only.  Rank 0 prints ONE JSON line.

`value` = patches of all ranks / wall time of the K timed steps (barrier + synchronize on both sides, max over ranks).
Next to it, from one HIP event per step boundary on the step's stream: `ms_per_step_median` and `ms_per_step_mean_events`.

Extra objects in that line:
  roofline     -- the dominant kernel class of the step (largest time share in the newest profiles/*_kernel_stats.csv; override
                  with --roofline-kernel), one representative launch of it timed with HIP events on its launch stream INSIDE
                  the timed steps (sifsr_profile_*): algorithmic FLOPs / average duration vs the fp32 MFMA peak.  `kernels`
                  carries the same for the forward / input-gradient / weight-gradient launches of the 16-channel 256^2 layers
                  side by side, `step` the whole-step ratio (SURVEY.md §8 d FLOPs per patch), `traffic` the PMC-measured HBM
                  bytes of that launch from the newest committed profiles/*_traffic.json (rocprofv3 --pmc passes of this same
                  command; PMC counters cannot be collected from inside the run).
                  The forward and input-gradient convolutions run in the Winograd F(2x2,3x3) domain (4/9 of the algorithmic
                  multiply-adds are executed on the matrix cores), so `frac` is a time-to-solution ratio against the fp32
                  MFMA roofline of the direct algorithm, not a pipe-utilisation figure (that one is in profiles/*_mfma_util.txt).
  cpu_baseline -- the oracle (CPU restatement of the reference, kind "port") timed on this box's host cores on a bounded
                  sample: batch 8 and batch 16, at all physical cores and at 8 threads (rank 0, N=1 only).

--mode infer : BASELINE.json config 4 -- eval forward of 256 full tiles per call, hipGraph replay (predict.GraphedPredictor);
               value = tiles/s, roofline.step.frac against the 43,634 tiles/s fp32 ceiling of SURVEY.md §8 d.
--dtype bf16 : BASELINE.json config 5 (bf16 MFMA operands); HBM-bound, the roofline object switches to GB/s.
"""
import argparse
import csv
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / IPC on this driver needs dmabuf handles (before any HIP init)

import torch

TRAIN_FLOPS_PER_PATCH = 10_777_264_128      # SURVEY.md §8 d (conv MACs x2: fwd + dgrad + wgrad)
FWD_FLOPS_PER_PATCH = 3_605_004_288
TRAIN_BYTES_PER_PATCH = 195_821_568
FWD_BYTES_PER_PATCH = 65_273_856
PEAK_FP32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32
PEAK_HBM_GBS = 8000.0

# name -> (layer index in the engine table, phase, algorithmic FLOPs per patch, kernel-name pattern in a rocprofv3 stats file)
#   phase 1 = forward conv, 2 = dgrad, 3 = wgrad.  2*9*Cin*Cout*H*W per patch.
F16 = 2 * 9 * 16 * 16 * 256 * 256
ROOFLINE_KERNELS = {
    "fwd_16x16_256": (1, 1, F16, "conv3x3_mfma_kernel<1, false"),      # inbloc.bloc.3 forward
    "dgrad_16x16_256": (1, 2, F16, "conv3x3_mfma_kernel<1, true"),     # inbloc.bloc.3 input gradient (+ fused BN sums of inbloc.bloc.0)
    "wgrad_16x16_256": (1, 3, F16, "conv3x3_wgrad_wino_kernel<1, 1, true"),   # inbloc.bloc.3 weight gradient
    "fwd_32x16_256": (15, 1, 2 * F16, "conv3x3_mfma_kernel<1, false"),  # ub3.convbloc.bloc.0 forward
    "wgrad_32x16_256": (15, 3, 2 * F16, "conv3x3_wgrad_wino_kernel<1, 2"),
    "dgrad_32x16_256": (15, 2, 2 * F16, "conv3x3_mfma_kernel<2, true"),
}
SIDE_BY_SIDE = ("fwd_16x16_256", "dgrad_16x16_256", "wgrad_16x16_256")
WINOGRAD = {"fwd_16x16_256", "dgrad_16x16_256", "fwd_32x16_256", "dgrad_32x16_256"}      # F(2x2,3x3)
WINOGRAD_W = {"wgrad_16x16_256", "wgrad_32x16_256"}                                        # F(3x3,2x2)


def dominant_kernel():
    """The roofline kernel whose kernel CLASS has the largest time share in the newest committed rocprofv3 kernel-stats file."""
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv")) if "single" not in os.path.basename(f)),
                   key=os.path.getmtime)   # profiles of THIS command (default two-stream backward), not the single-stream diagnostics
    for f in reversed(files):
        try:
            rows = list(csv.DictReader(open(f)))
        except (OSError, ValueError):
            continue
        best = None
        for r in rows:
            for name in SIDE_BY_SIDE:
                if ROOFLINE_KERNELS[name][3] in r.get("Name", ""):
                    share = float(r["TotalDurationNs"])
                    if best is None or share > best[0]:
                        best = (share, name)
        if best:
            return best[1], os.path.basename(f)
    return "wgrad_16x16_256", None


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, written by
    tools/summarize_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=os.path.getmtime)
    for f in reversed(files):
        try:
            t = json.load(open(f))["per_launch"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if t:
            return round(t["bytes"]), os.path.basename(f)
    return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(kind, alpha, gamma, lr, mean, std, budget_seconds=24.0):
    """Time the oracle's train step (fwd + loss + bwd + Adam) on the host cores: batch 8 and 16, at all physical cores and at
    8 threads (SURVEY.md §8 d), a bounded sample of the same workload.  `value` = the best of the measured configurations."""
    from oracle import sif_oracle as O
    phys = physical_cores()
    thread_sets = sorted({phys, min(8, phys)}, reverse=True)
    runs = []
    per_cfg = budget_seconds / (2 * len(thread_sets))
    for nt in thread_sets:
        torch.set_num_threads(nt)
        for B in (8, 16):
            sd = O.synthetic_state(0)
            lst, lst_up, ndvi = O.synthetic_batch(1234, B)
            adam = O.AdamState(O.param_names(), lr)
            O.train_step(sd, adam, lst, lst_up, ndvi, mean, std, alpha, gamma, kind)       # warm-up
            t0 = time.perf_counter()
            n = 0
            while n < 2 or (time.perf_counter() - t0 < per_cfg and n < 20):
                O.train_step(sd, adam, lst, lst_up, ndvi, mean, std, alpha, gamma, kind)
                n += 1
            dt = time.perf_counter() - t0
            runs.append({"threads": nt, "batch": B, "steps": n, "patches_per_s": round(B * n / dt, 3)})
    best = max(runs, key=lambda r: r["patches_per_s"])
    return {"value": best["patches_per_s"], "unit": "patches/s", "cores": best["threads"], "kind": "port",
            "sample": f"{kind.upper()} train steps (fwd+loss+bwd+Adam), 256x256, fp32, torch {torch.__version__} CPU on "
                      f"{cpu_model()} ({phys} physical cores); best of the runs listed",
            "runs": runs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="patches per GPU per step (default 64; 256 tiles for --mode infer)")
    ap.add_argument("--kind", default="sr2", choices=["sr2", "sr1"])
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--roofline-kernel", default="auto", choices=["auto"] + sorted(ROOFLINE_KERNELS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solo", action="store_true", help="skip the extra single-stream steps that time the selected kernels alone "
                                                            "(profiling runs: keeps the kernel trace to the steps of the benchmark itself)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16x3"],
                    help="f32 = the headline fp32 path; bf16 = BASELINE.json config 5 (bf16 MFMA operands in the 3x3 conv "
                         "forward / input-gradient / weight-gradient passes, fp32 accumulation and storage); bf16x3 = fp32 on the bf16 "
                         "matrix cores (exact three-way bf16 split of every conv operand, six cross products; fp32-level results) "
                         "-- neither is the default")
    ap.add_argument("--host-io", action="store_true",
                    help="report, next to the normal line, the rate with every step's inputs starting in (pinned) host memory and, for "
                         "--mode infer, the result copied back: the PCIe-inclusive figure (never `value`)")
    args = ap.parse_args()

    import ctypes

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

    infer = args.mode == "infer"
    batch = args.batch or (256 if infer else 64)
    kind = args.kind
    alpha, gamma, lr = (0.5, -0.25, 1e-4) if kind == "sr2" else (0.99, -0.5, 1e-3)
    stats = dict(sifsr.dataset.DEFAULT_STATS)
    torch.manual_seed(0)
    model = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
    model.compute_dtype = {"f32": "fp32", "bf16": "bf16", "bf16x3": "bf16x3"}[args.dtype]
    lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(batch, dev, seed=1234 + rank)

    if infer:
        predictor = sifsr.predict.GraphedPredictor(model, batch, stats)      # captured once; replayed per call

        def step():
            return (predictor(lst_up, ndvi),)
    else:
        opt = sifsr.FlatAdam(model.parameters(), lr=lr)
        dp.broadcast_parameters(model, opt, src=0)                           # replicas start identical whatever their seeds

        def step():
            return sifsr.train.train_step(model, opt, lst, lst_up, ndvi, stats, alpha, gamma, kind)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def host_io_rate():
        """Steps whose inputs start in pinned host memory (two buffer sets: the copy of step i+1 is enqueued on a second stream
        while step i computes) and, for inference, whose output ends in pinned host memory.  patches (tiles) per second."""
        n = 40
        hb = [[t.cpu().pin_memory() for t in (lst, lst_up, ndvi)] for _ in range(2)]
        db = [[torch.empty_like(t) for t in (lst, lst_up, ndvi)] for _ in range(2)]
        out_h = torch.empty(batch, 1, 256, 256).pin_memory() if infer else None
        copy_s = torch.cuda.Stream()
        ready = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]

        def upload(i):
            k = i & 1
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(freed[k])
                for d_, h_ in zip(db[k], hb[k]):
                    d_.copy_(h_, non_blocking=True)
                ready[k].record(copy_s)

        def run(count):
            upload(0)
            for i in range(count):
                k = i & 1
                if i + 1 < count:
                    upload(i + 1)
                torch.cuda.current_stream().wait_event(ready[k])
                a_, b_, c_ = db[k]
                if infer:
                    out_h.copy_(predictor(b_, c_), non_blocking=True)
                else:
                    sifsr.train.train_step(model, opt, a_, b_, c_, stats, alpha, gamma, kind)
                freed[k].record()

        for k in range(2):
            freed[k].record()
        run(4)                 # first touches of the pinned buffers
        fence()
        t0 = time.perf_counter()
        run(n)
        fence()
        return n * batch / (time.perf_counter() - t0)

    for _ in range(args.warmup):
        step()

    # kernels timed side by side inside the timed steps (event pools are created here, outside the timed region)
    timed = []
    dom, dom_src = (args.roofline_kernel, None) if args.roofline_kernel != "auto" else dominant_kernel()
    if not infer:
        for name in dict.fromkeys((dom,) + SIDE_BY_SIDE):
            layer, phase, _, _ = ROOFLINE_KERNELS[name]
            slot = L.call("sifsr_profile_select", layer, phase) if not timed else L.call("sifsr_profile_add", layer, phase)
            timed.append((name, 0 if not timed else slot))
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    per_step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    ktimes = {}
    for name, slot in timed:
        kms, kcount = ctypes.c_float(0), ctypes.c_int(0)
        L.call("sifsr_profile_read_slot", slot, ctypes.byref(kms), ctypes.byref(kcount))
        ktimes[name] = (kms.value / max(1, kcount.value), kcount.value)
    L.call("sifsr_profile_select", -1, 0)
    # The weight-gradient kernels run on the library's second stream BESIDE the rest of the backward pass, so their in-step
    # duration is that of a kernel sharing the machine.  A few extra, untimed steps with the single-stream schedule give the
    # same launches' stand-alone durations, reported next to the in-step ones (`solo_ms`).
    ksolo = {}
    if timed and not args.no_solo:
        L.call("sifsr_set_wgrad_stream", 0)
        try:
            step()
            for j, (name, _) in enumerate(timed):
                layer, phase, _, _ = ROOFLINE_KERNELS[name]
                if j == 0:
                    L.call("sifsr_profile_select", layer, phase)
                else:
                    L.call("sifsr_profile_add", layer, phase)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            for name, slot in timed:
                kms, kcount = ctypes.c_float(0), ctypes.c_int(0)
                L.call("sifsr_profile_read_slot", slot, ctypes.byref(kms), ctypes.byref(kcount))
                ksolo[name] = kms.value / max(1, kcount.value)
        finally:
            L.call("sifsr_profile_select", -1, 0)
            L.call("sifsr_set_wgrad_stream", -1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert all(bool(torch.isfinite(v).all()) for v in out), "non-finite result"

    units = batch * world * args.steps
    value = units / dt
    per_gpu = value / world
    what = "tiles" if infer else "patches"
    out_json = {
        "metric": ("LST+NDVI 256x256 tiles/sec (eval forward, hipGraph replay), whole job" if infer else
                   "LST+NDVI 256x256 patches/sec (train fwd+bwd), whole job"),
        "value": round(value, 2), "unit": f"{what}/s", "per_gpu": round(per_gpu, 2),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000 * dt / args.steps, 4),
        "ms_per_step_median": round(statistics.median(per_step_ms), 4),
        "ms_per_step_mean_events": round(sum(per_step_ms) / len(per_step_ms), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16 conv operands (fwd, dgrad, wgrad), f32 accumulate/storage",
                  "bf16x3": "f32 as 3-term bf16 splits (conv fwd, dgrad: 6 bf16 MFMA cross products), f32 accumulate/storage; wgrad f32 MFMA"}[args.dtype],
        "data": "synthetic",
        "config": ({"workload": f"ModelB inference-only (predict.py path), batch {batch} full tiles 256x256, {world}x MI355X, "
                                "hipGraph-captured eval forward + de-normalisation", "batch_per_gpu": batch,
                    "patch": "256x256 (LST 64x64 upsampled + NDVI 256x256)", "parallelism": f"replicas x{world}"} if infer else
                   {"workload": f"ModelB SIF-NN-{kind.upper()} ({'gradFTM' if kind == 'sr2' else 'predef_filters'} loss) "
                                f"batch {batch}/GPU, synthetic 256x256, {world}x MI355X, fwd+loss+bwd+Adam",
                    "batch_per_gpu": batch, "patch": "256x256 (LST 64x64 + NDVI 256x256)",
                    "parallelism": f"dp{world}", "loss": kind}),
    }
    if rank == 0:
        flops_unit = FWD_FLOPS_PER_PATCH if infer else TRAIN_FLOPS_PER_PATCH
        bytes_unit = FWD_BYTES_PER_PATCH if infer else TRAIN_BYTES_PER_PATCH
        step_tf = flops_unit * per_gpu / 1e12
        step_obj = {"achieved": round(step_tf, 2), "frac": round(step_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                    "hbm_GBs_algorithmic": round(bytes_unit * per_gpu / 1e9, 1),
                    "ceiling_units_per_s": round(PEAK_FP32_MFMA_TFLOPS * 1e12 / flops_unit, 0)}
        if infer:
            out_json["roofline"] = {"bound": "mfma", "kernel": "whole eval forward (graph replay)", "achieved": round(step_tf, 2),
                                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(step_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                                    "traffic": None, "step": step_obj}
        elif args.dtype == "f32":
            kern = {}
            for name, (avg_ms, n) in ktimes.items():
                tf = ROOFLINE_KERNELS[name][2] * batch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
                solo = ksolo.get(name, 0.0)
                tf_solo = ROOFLINE_KERNELS[name][2] * batch / (solo * 1e-3) / 1e12 if solo > 0 else 0.0
                kern[name] = {"avg_ms": round(avg_ms, 4), "launches_timed": n, "achieved": round(tf, 2),
                              "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                              "solo_ms": round(solo, 4), "solo_frac": round(tf_solo / PEAK_FP32_MFMA_TFLOPS, 4),
                              "concurrent": "runs on the second stream beside the input-gradient chain" if name.startswith("wgrad") else
                                            ("shares the machine with the previous layer's weight gradient" if name.startswith("dgrad") else "alone"),
                              "algorithm": "winograd F(2x2,3x3): 4/9 of the algorithmic MACs executed" if name in WINOGRAD else
                                           ("winograd F(3x3,2x2): 4/9 of the algorithmic MACs executed" if name in WINOGRAD_W else "direct")}
            traffic, traffic_src = measured_traffic(dom)
            k = kern[dom]
            out_json["roofline"] = {
                "bound": "mfma", "kernel": dom, "kernel_chosen_from": dom_src or "--roofline-kernel", "achieved": k["achieved"],
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": k["frac"], "traffic": traffic,
                "traffic_unit": "HBM bytes/launch (PMC)", "traffic_source": traffic_src,
                "kernel_avg_ms": k["avg_ms"], "kernel_launches_timed": k["launches_timed"], "algorithm": k["algorithm"],
                "solo_ms": k["solo_ms"], "solo_frac": k["solo_frac"], "concurrent": k["concurrent"],
                "kernels": kern, "step": step_obj,
            }
        else:
            # bf16 operands: the matrix-core peak rises 16x, the bytes do not change (fp32 storage) -> HBM-bound
            # (SURVEY.md §8 d).  Algorithmic bytes of the selected conv launch = its input + output activations.
            avg_ms, n = ktimes[dom]
            cin, cout = {"16x16": (16, 16), "32x16": (32, 16)}[dom.split("_")[1]]
            kbytes = (cin + cout) * 256 * 256 * 4 * batch
            gbs = kbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            out_json["roofline"] = {
                "bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "kernel_avg_ms": round(avg_ms, 4),
                "kernel_launches_timed": n,
                "step": {"hbm_GBs_algorithmic": round(bytes_unit * per_gpu / 1e9, 1),
                         "frac": round(bytes_unit * per_gpu / 1e9 / PEAK_HBM_GBS, 4)},
            }
        if args.host_io:
            out_json["host_io"] = {"value": round(host_io_rate(), 2), "unit": out_json["unit"],
                                   "what": "inputs start in pinned host memory every step (double-buffered async H2D on a second stream)"
                                           + ("; output copied back to pinned host memory" if infer else "")
                                           + " -- the PCIe-inclusive rate, not the headline value"}
        if world == 1 and not args.no_cpu_baseline and not infer:
            out_json["cpu_baseline"] = cpu_baseline(kind, alpha, gamma, lr, stats["mean_lst"], stats["std_lst"])
        print(json.dumps(out_json), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
