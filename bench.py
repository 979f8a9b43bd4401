#!/usr/bin/env python3
"""Headline benchmark: LST+NDVI 256x256 training patches/s (BASELINE.json), SR2 step at batch 64/GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
cat(lst_up, ndvi) -> ModelB_2 forward -> SIF loss (SR2) -> backward -> [gradient all-reduce] -> Adam
(train_model_B_gradFTM.py:94-121), through the HIP path only.  Rank 0 prints ONE JSON line.

`value` = patches of all ranks / wall time of the K timed steps (barrier + synchronize on both sides, max over ranks).
Next to it, from one HIP event per step boundary on the step's stream: `ms_per_step_median` and `ms_per_step_mean_events`.

N > 1 without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset): the process starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`
as a CHILD before importing torch or touching HIP, relays its output and exits with its return code (never an exec of a
process that has initialised the GPU).  The N > 1 line adds `allreduce_ms` (HIP events around the gradient exchange),
`dist_backend` / `rccl_ranks`, and the per-rank wall time per step (`rank_ms_per_step_min` / `_max`).

Extra objects in that line:
  roofline     -- the dominant kernel CLASS of the step: per-class sums of TotalDurationNs over the newest two-stream
                  profiles/*_kernel_stats.csv, the class's launches (engine layer table below) ALL timed with HIP events on
                  their launch stream INSIDE the timed steps (sifsr_profile_*): class algorithmic FLOPs per step / class
                  time per step vs the fp32 MFMA peak.  `kernels` carries the launches of inbloc.bloc.3 (16->16 @256^2: forward, and
                  the fused input + weight gradient) and ub3.convbloc.bloc.0 (32->16 @256^2) side by side, `step` the whole-step ratio (SURVEY.md §8 d FLOPs
                  per patch), `traffic` the PMC-measured HBM bytes per launch of that class from the newest committed
                  profiles/*_traffic.json (rocprofv3 --pmc passes of this same command; PMC counters cannot be collected
                  from inside the run).
                  All three conv passes run in the Winograd domain (4/9 of the algorithmic multiply-adds are executed on the
                  matrix cores), so `frac` is a time-to-solution ratio against the fp32 MFMA roofline of the direct
                  algorithm, not a pipe-utilisation figure (that one is in profiles/*_mfma_util.txt).
  also         -- measured in the same process after the timed region (N = 1, default mode): BASELINE.json config 4
                  (`infer_b256`: eval forward of 256 tiles, hipGraph replay) and config 5 (`bf16`), so that a driver record of
                  the default command carries them.  --no-also skips it (profiling runs).
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
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / IPC on this driver needs dmabuf handles (before any HIP init)

torch = None    # imported by main() AFTER the self-launch decision: the launching parent never loads torch or touches HIP

TRAIN_FLOPS_PER_PATCH = 10_777_264_128      # SURVEY.md §8 d (conv MACs x2: fwd + dgrad + wgrad)
FWD_FLOPS_PER_PATCH = 3_605_004_288
TRAIN_BYTES_PER_PATCH = 195_821_568
FWD_BYTES_PER_PATCH = 65_273_856
PEAK_FP32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32
PEAK_HBM_GBS = 8000.0

# ---- the engine's layer table (csrc/engine.hip build_net(): the 17 Conv-BN-ReLU units in parameters() order) ----
LAYER_NAMES = ("inbloc.bloc.0", "inbloc.bloc.3", "db1.res.0", "db1.res.3", "db1.lastconv", "db2.res.0", "db2.res.3", "db2.lastconv",
               "db3.res.0", "db3.res.3", "db3.lastconv", "ub1.bloc.0", "ub1.bloc.3", "ub2.bloc.0", "ub2.bloc.3", "ub3.bloc.0", "ub3.bloc.3")
LAYER_CIN = (2, 16, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 64, 64, 32, 32, 16)
LAYER_COUT = (16, 16, 16, 16, 32, 32, 32, 64, 64, 64, 64, 64, 32, 32, 16, 16, 16)
LAYER_LEVEL = (0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 2, 2, 1, 1, 0, 0)
PHASE_NAMES = {1: "fwd", 2: "dgrad", 3: "wgrad"}


def layer_flops(layer, hw=256):
    """Algorithmic FLOPs per patch of ONE pass (forward, input gradient or weight gradient) of MFMA unit `layer`."""
    side = hw >> LAYER_LEVEL[layer]
    return 2 * 9 * LAYER_CIN[layer] * LAYER_COUT[layer] * side * side


FUSED_BWD16 = (1, 2, 3, 16)     # 16 -> 16 layers at 256^2 / 128^2: input AND weight gradient in one launch (conv_bwd16.hip)


def member_flops(layer, phase, hw=256):
    """Algorithmic FLOPs per patch of the launch that (layer, phase) names: both backward passes for a fused 16 -> 16 layer."""
    return layer_flops(layer, hw) * (2 if layer in FUSED_BWD16 and phase == 2 else 1)


def kernel_class(layer, phase):
    """rocprofv3 kernel-name prefix of the launch the fp32 engine issues for (layer, phase) at 256x256 patches (engine.hip
    conv_unit_fwd / _dgrad / _wgrad / _bwd16; conv_mfma.hip / conv_wino8.hip / conv_wgrad_wino.hip / conv_bwd16.hip dispatch).
    None: no launch of its own (the first thin conv; the weight gradient of a fused 16 -> 16 layer, which is part of the
    layer's (layer, 2) launch)."""
    cin, cout = LAYER_CIN[layer], LAYER_COUT[layer]
    if layer == 0:
        return None                                                  # the thin first conv is not an MFMA unit of these classes
    if layer in FUSED_BWD16 and phase != 1:
        return "conv3x3_bwd16_kernel" if phase == 2 else None
    if phase == 1:
        return "conv3x3_mfma_kernel<1, false" if cout == 16 else f"conv3x3_wino8_kernel<{cout // 16}, false"
    if phase == 2:
        nb = min(cin, 64) // 16                                      # 128 input channels: two 64-channel launches
        if nb == 1:
            return "conv3x3_mfma_kernel<1, true, 0, " + ("false" if layer == 16 else "true")
        return f"conv3x3_wino8_kernel<{nb}, true"
    nbo = min(cout, 32) // 16                                        # 64 output channels: two 32-channel halves
    nbi = 1 if cin < 32 else 2
    return f"conv3x3_wgrad_wino_kernel<{nbo}, {nbi}, " + ("false" if layer == 16 else "true")


def class_table():
    """{class name pattern: [(layer, phase), ...]} for every MFMA launch of a training step."""
    t = {}
    for layer in range(1, 17):
        for phase in (1, 2, 3):
            cls = kernel_class(layer, phase)
            if cls is not None:
                t.setdefault(cls, []).append((layer, phase))
    return t


# inbloc.bloc.3 (16 -> 16 @256^2: forward, and the fused input + weight gradient) and ub3.convbloc.bloc.0 (32 -> 16 @256^2)
SIDE_BY_SIDE = {"fwd_16x16_256": (1, 1), "bwd16_16x16_256": (1, 2), "fwd_32x16_256": (15, 1), "dgrad_32x16_256": (15, 2),
                "wgrad_32x16_256": (15, 3)}


def dominant_class(stats_file=None):
    """The kernel CLASS with the largest summed duration in the newest committed two-stream rocprofv3 kernel-stats file
    (every CSV row whose name carries the class prefix is added up).  Returns (class pattern, source file)."""
    files = [stats_file] if stats_file else sorted(
        (f for f in glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv"))
         if "single" not in os.path.basename(f) and "bf16" not in os.path.basename(f)), key=os.path.getmtime)
    table = class_table()
    for f in reversed(files):
        try:
            rows = list(csv.DictReader(open(f)))
        except (OSError, ValueError):
            continue
        tot = {}
        for r in rows:
            name = r.get("Name", "")
            for cls in sorted(table, key=len, reverse=True):          # longest prefix first: "<1, true, 0, true" before "<1, true"
                if cls in name:
                    tot[cls] = tot.get(cls, 0.0) + float(r["TotalDurationNs"])
                    break
        if tot:
            return max(tot, key=tot.get), os.path.basename(f)
    return "conv3x3_wgrad_wino_kernel<2, 2, true", None


def measured_traffic(key):
    """HBM bytes per launch of `key` from the newest committed PMC summary (profiles/*_traffic.json, written by
    tools/summarize_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=os.path.getmtime)
    for f in reversed(files):
        try:
            per = json.load(open(f))["per_launch"]
        except (OSError, ValueError, KeyError):
            continue
        t = per.get(key)
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="patches per GPU per step (default 64; 256 tiles for --mode infer)")
    ap.add_argument("--kind", default="sr2", choices=["sr2", "sr1"])
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--roofline-class", default="auto",
                    help="kernel class (name prefix as in kernel_class()) the roofline object reports; auto = the class with the "
                         "largest time share in the newest profiles/*_kernel_stats.csv")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solo", action="store_true", help="skip the extra single-stream steps that time the selected kernels alone "
                                                            "(profiling runs: keeps the kernel trace to the steps of the benchmark itself)")
    ap.add_argument("--no-also", action="store_true", help="skip the config-4 / config-5 measurements appended to the default line")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = the headline fp32 path; bf16 = BASELINE.json config 5 (bf16 MFMA operands in the 3x3 conv "
                         "forward / input-gradient / weight-gradient passes, fp32 accumulation) -- not the default")
    ap.add_argument("--host-io", action="store_true",
                    help="report, next to the normal line, the rate with every step's inputs starting in (pinned) host memory and, for "
                         "--mode infer, the result copied back: the PCIe-inclusive figure (never `value`)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="N > 1 plumbing check without a GPU: every rank joins the process group (gloo), all-reduces its rank and "
                         "rank 0 prints {\"dry_run\": true, \"world\": N, \"rank_sum\": ...}")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as a child
    `torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1) and return its exit code.  Called before torch is
    imported: this parent never initialises HIP, and nothing is exec'ed."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")          # torchrun would set 1 and warn; the ranks' host work is launch enqueueing only
    print(f"[bench] --gpus {args.gpus}: starting {args.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def dry_run(args):
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "world": world, "n_gpus": args.gpus, "rank_sum": float(t.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if world == args.gpus else 1


def main():
    global torch
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch as _torch
    torch = _torch
    if args.dry_run_launch:
        sys.exit(dry_run(args))

    import ctypes

    import sifsr
    from sifsr import _lib as L
    from sifsr import distributed as dp

    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    model.compute_dtype = {"f32": "fp32", "bf16": "bf16"}[args.dtype]
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

    # ---- kernels timed inside the timed steps (event pools are created here, outside the timed region): every launch of the
    # dominant class + the three passes of inbloc.bloc.3 side by side
    table = class_table()
    dom_cls, dom_src = (args.roofline_class, "--roofline-class") if args.roofline_class != "auto" else dominant_class()
    if dom_cls not in table:
        raise SystemExit(f"--roofline-class {dom_cls!r}: not one of {sorted(table)}")
    selections = []                                   # [(layer, phase)] in slot order; on EVERY rank: the extra single-stream
    if not infer:                                     # steps below contain the gradient all-reduce, a collective
        for lp in (list(table[dom_cls]) if args.dtype == "f32" else []) + list(SIDE_BY_SIDE.values()):
            if lp not in selections:
                selections.append(lp)

    def select_all():
        for j, (layer, phase) in enumerate(selections):
            rc = L.call("sifsr_profile_select", layer, phase) if j == 0 else L.call("sifsr_profile_add", layer, phase)
            if j > 0 and rc != j:
                raise SystemExit(f"sifsr_profile_add({layer}, {phase}) -> {rc}: out of profiling slots")

    def read_all():
        res = {}
        for j, lp in enumerate(selections):
            kms, kcount = ctypes.c_float(0), ctypes.c_int(0)
            L.call("sifsr_profile_read_slot", j, ctypes.byref(kms), ctypes.byref(kcount))
            res[lp] = (kms.value / max(1, kcount.value), kcount.value)
        return res

    ar_timer = None
    if world > 1 and not infer:
        ar_timer = dp.AllreduceTimer(args.steps + 8)
    select_all()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    if ar_timer is not None:
        dp.set_allreduce_timer(ar_timer)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    fence()
    dt_local = dt = time.perf_counter() - t0
    dp.set_allreduce_timer(None)
    per_step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    ktimes = read_all()
    L.call("sifsr_profile_select", -1, 0)
    # The weight-gradient kernels run on the library's second stream BESIDE the rest of the backward pass, so their in-step
    # duration is that of a kernel sharing the machine.  A few extra, untimed steps with the single-stream schedule give the
    # same launches' stand-alone durations, reported next to the in-step ones (`solo_ms`).
    ksolo = {}
    if selections and not args.no_solo:
        L.call("sifsr_set_wgrad_stream", 0)
        try:
            step()
            select_all()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            ksolo = {lp: v[0] for lp, v in read_all().items()}
        finally:
            L.call("sifsr_profile_select", -1, 0)
            L.call("sifsr_set_wgrad_stream", -1)
    dist_info = {}
    if world > 1:
        on_dev = torch.distributed.get_backend() == "nccl"
        t = torch.tensor([dt_local], dtype=torch.float64, device=dev if on_dev else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        dt = max(per_rank)
        ar = torch.tensor([ar_timer.mean_ms() if ar_timer is not None else 0.0], dtype=torch.float64, device=dev if on_dev else "cpu")
        torch.distributed.all_reduce(ar, op=torch.distributed.ReduceOp.MAX)
        dist_info = {"dist_backend": torch.distributed.get_backend(), "rccl_ranks": world if on_dev else 0,
                     "allreduce_ms": round(float(ar.item()), 4),
                     "allreduce_what": "one sum all-reduce of the flat 282,705-float gradient buffer per step, HIP events on the step's "
                                       "stream around it (max over ranks of the per-rank mean; includes waiting for the slowest rank)",
                     "rank_ms_per_step_min": round(1000 * min(per_rank) / args.steps, 4),
                     "rank_ms_per_step_max": round(1000 * max(per_rank) / args.steps, 4)}
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
        "dtype": {"f32": "f32", "bf16": "bf16 conv operands (fwd, dgrad, wgrad), f32 accumulate/storage"}[args.dtype],
        "data": "synthetic",
        "config": ({"workload": f"ModelB inference-only (predict.py path), batch {batch} full tiles 256x256, {world}x MI355X, "
                                "hipGraph-captured eval forward + de-normalisation", "batch_per_gpu": batch,
                    "patch": "256x256 (LST 64x64 upsampled + NDVI 256x256)", "parallelism": f"replicas x{world}"} if infer else
                   {"workload": f"ModelB SIF-NN-{kind.upper()} ({'gradFTM' if kind == 'sr2' else 'predef_filters'} loss) "
                                f"batch {batch}/GPU, synthetic 256x256, {world}x MI355X, fwd+loss+bwd+Adam",
                    "batch_per_gpu": batch, "patch": "256x256 (LST 64x64 + NDVI 256x256)",
                    "parallelism": f"dp{world}", "loss": kind}),
    }
    out_json.update(dist_info)
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
            def entry(lp):
                avg_ms, n = ktimes[lp]
                fl = member_flops(*lp) * batch
                tf = fl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
                solo = ksolo.get(lp, 0.0)
                tf_solo = fl / (solo * 1e-3) / 1e12 if solo > 0 else 0.0
                return {"layer": LAYER_NAMES[lp[0]], "pass": "dgrad+wgrad (one launch)" if lp[0] in FUSED_BWD16 and lp[1] == 2 else PHASE_NAMES[lp[1]],
                        "gflop": round(fl / 1e9, 2),
                        "avg_ms": round(avg_ms, 4), "launches_timed": n, "achieved": round(tf, 2),
                        "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                        "solo_ms": round(solo, 4), "solo_frac": round(tf_solo / PEAK_FP32_MFMA_TFLOPS, 4)}
            members = [entry(lp) for lp in table[dom_cls]]
            cls_gflop = sum(m["gflop"] for m in members)                      # per step
            cls_ms = sum(m["avg_ms"] for m in members)
            cls_solo = sum(m["solo_ms"] for m in members)
            cls_tf = cls_gflop / cls_ms if cls_ms > 0 else 0.0                # GFLOP / ms = TFLOP/s
            cls_tf_solo = cls_gflop / cls_solo if cls_solo > 0 else 0.0
            traffic, traffic_src = measured_traffic(dom_cls)
            side = {name: entry(lp) for name, lp in SIDE_BY_SIDE.items()}
            out_json["roofline"] = {
                "bound": "mfma", "kernel": dom_cls + ">", "kernel_chosen_from": dom_src,
                "what": "every launch of the step's dominant kernel class, timed with HIP events on its launch stream inside the "
                        "timed steps; achieved = class algorithmic FLOPs per step / class time per step",
                "launches_per_step": len(members), "gflop_per_step": round(cls_gflop, 2), "class_ms_per_step": round(cls_ms, 4),
                "achieved": round(cls_tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(cls_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                "solo_ms_per_step": round(cls_solo, 4), "solo_frac": round(cls_tf_solo / PEAK_FP32_MFMA_TFLOPS, 4),
                "concurrent": ("runs on the second stream beside the input-gradient chain (solo_* = the same launches on one stream)"
                               if "wgrad" in dom_cls else "on the caller's stream; the previous layer's weight gradient may run beside it"),
                "algorithm": ("winograd F(3x3,2x2)" if "wgrad" in dom_cls else "winograd F(2x2,3x3) + F(3x3,2x2)" if "bwd16" in dom_cls
                              else "winograd F(2x2,3x3)") + ": 4/9 of the algorithmic MACs executed",
                "traffic": traffic, "traffic_unit": "HBM bytes per launch, class average (PMC)", "traffic_source": traffic_src,
                "members": members, "kernels": side, "step": step_obj,
            }
        else:
            # bf16 operands: the matrix-core peak rises 16x, the bytes do not change (fp32 storage) -> HBM-bound
            # (SURVEY.md §8 d).  Algorithmic bytes of inbloc.bloc.3's forward launch = its input + output activations.
            avg_ms, n = ktimes[(1, 1)]
            kbytes = (16 + 16) * 256 * 256 * 4 * batch
            gbs = kbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            out_json["roofline"] = {
                "bound": "hbm", "kernel": "inbloc.bloc.3 forward (16->16 @256^2)", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "kernel_avg_ms": round(avg_ms, 4),
                "kernel_launches_timed": n,
                "step": {"hbm_GBs_algorithmic": round(bytes_unit * per_gpu / 1e9, 1),
                         "frac": round(bytes_unit * per_gpu / 1e9 / PEAK_HBM_GBS, 4),
                         "note": "against the fp32-storage bytes this mode moves; SURVEY.md §8 d's bf16 ceiling (81,707 patches/s) "
                                 "assumes bf16 storage", "frac_of_bf16_storage_ceiling": round(per_gpu / 81707.0, 4)},
            }
        if args.host_io:
            out_json["host_io"] = {"value": round(host_io_rate(), 2), "unit": out_json["unit"],
                                   "what": "inputs start in pinned host memory every step (double-buffered async H2D on a second stream)"
                                           + ("; output copied back to pinned host memory" if infer else "")
                                           + " -- the PCIe-inclusive rate, not the headline value"}
        if world == 1 and not infer and args.dtype == "f32" and not args.no_also:
            out_json["also"] = also_configs(sifsr, dev, stats, lst, lst_up, ndvi, alpha, gamma, lr, kind)
        if world == 1 and not args.no_cpu_baseline and not infer:
            out_json["cpu_baseline"] = cpu_baseline(kind, alpha, gamma, lr, stats["mean_lst"], stats["std_lst"])
        print(json.dumps(out_json), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def also_configs(sifsr, dev, stats, lst, lst_up, ndvi, alpha, gamma, lr, kind):
    """BASELINE.json configs 4 and 5 measured in this process after the headline's timed region (a few seconds each), so a
    record of the default command carries them: `infer_b256` = eval forward of 256 tiles by hipGraph replay (tiles/s against
    the 43,634 tiles/s fp32 ceiling), `bf16` = the SR step with bf16 MFMA operands at batch 64 (patches/s; frac against the
    bytes the mode moves and against SURVEY.md §8 d's bf16-storage ceiling)."""
    res = {}

    def timed(fn, warm, n):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    try:
        torch.manual_seed(0)
        m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
        b = 256
        _, lu, nd = sifsr.dataset.synthetic_device_batch(b, dev, seed=99)
        pred = sifsr.predict.GraphedPredictor(m, b, stats)
        sec = timed(lambda: pred(lu, nd), 3, 20)
        rate = b / sec
        res["infer_b256"] = {"config": "BASELINE.json config 4: eval forward, batch 256 tiles, hipGraph replay, fp32", "value": round(rate, 1),
                             "unit": "tiles/s", "ms_per_batch": round(1000 * sec, 4),
                             "step_frac": round(FWD_FLOPS_PER_PATCH * rate / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "bound": "mfma (fp32)"}
        del pred, m, lu, nd
    except Exception as e:                                   # never lose the headline line to an appendix
        res["infer_b256"] = {"error": repr(e)}
    try:
        torch.manual_seed(0)
        m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
        m.compute_dtype = "bf16"
        o = sifsr.FlatAdam(m.parameters(), lr=lr)
        sec = timed(lambda: sifsr.train.train_step(m, o, lst, lst_up, ndvi, stats, alpha, gamma, kind), 5, 20)
        rate = lst.shape[0] / sec
        res["bf16"] = {"config": "BASELINE.json config 5 (one GPU's share): SR step, batch 64, bf16 MFMA operands, fp32 accumulation",
                       "value": round(rate, 1), "unit": "patches/s", "ms_per_step": round(1000 * sec, 4), "bound": "hbm",
                       "step_frac_fp32_storage_bytes": round(TRAIN_BYTES_PER_PATCH * rate / 1e9 / PEAK_HBM_GBS, 4),
                       "step_frac_bf16_storage_ceiling": round(rate / 81707.0, 4)}
    except Exception as e:
        res["bf16"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
