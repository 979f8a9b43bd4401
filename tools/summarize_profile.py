#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh output directory into profiles/<tag>_summary.md + copies of the CSV stats.

usage: python tools/summarize_profile.py gpurun_out/r01 r01 [steps_profiled=7]
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; per MI355X_MICROARCH.md §HBM, FETCH_SIZE on gfx950
reports exactly half of the bytes of a wide coalesced streaming read, so the read side is doubled.
"""
import csv, glob, os, shutil, sys, collections

src, tag = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
os.makedirs("profiles", exist_ok=True)
stats = max(glob.glob(f"{src}/stats/*/*kernel_stats.csv"), key=os.path.getmtime)   # (gpurun merges runs into the same directory: newest)
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def pmc(kind):
    f = glob.glob(f"{src}/pmc_{kind}/*/*counter_collection.csv")
    if not f:
        return {}
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


fetch, write = pmc("fetch"), pmc("write")
mode = "--dtype bf16: bf16 stored activations, bf16 MFMA operands" if "bf16" in tag else "fp32"
if "single" in tag:
    mode += "; SIFSR_WGRAD_STREAM=0, everything on one stream"
lines = [f"# rocprofv3 summary `{tag}` — bench.py --steps 5 --warmup 2 (N=1, batch 64, SR2, {mode})", "",
         f"Total kernel time {tot/1e6:.1f} ms over {steps} steps = **{tot/1e6/steps:.2f} ms/step**.", "",
         "| kernel | calls | avg µs | ms/step | % | HBM read MB/launch (2×FETCH_SIZE) | HBM write MB/launch |", "|---|---|---|---|---|---|---|"]
for r in rows[:30]:
    k = short(r["Name"])
    fr = fetch.get(k); wr = write.get(k)
    lines.append("| `%s` | %s | %.1f | %.3f | %.1f | %s | %s |" % (
        k[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps,
        100 * float(r["TotalDurationNs"]) / tot,
        "%.1f" % (2 * fr * 1024 / 1e6) if fr is not None else "", "%.1f" % (wr * 1024 / 1e6) if wr is not None else ""))
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# ---- per-class HBM traffic (what bench.py reports as roofline.traffic): the average bytes per launch of every kernel class of
# bench.class_table(), from the two PMC passes, and the whole step ----
import json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402  (class table only; bench.py imports torch lazily)

def pmc_rows(kind):
    f = glob.glob(f"{src}/pmc_{kind}/*/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []

frows, wrows = pmc_rows("fetch"), pmc_rows("write")
classes = sorted(bench.class_table(), key=len, reverse=True)      # longest prefix first
def class_of(name):
    for c in classes:
        if c in name:
            return c
    return None
traffic = {}
for c in classes:
    fr = [float(r["Counter_Value"]) for r in frows if class_of(r["Kernel_Name"]) == c]
    wr = [float(r["Counter_Value"]) for r in wrows if class_of(r["Kernel_Name"]) == c]
    if fr and wr:
        rb, wb = 2 * 1024 * sum(fr) / len(fr), 1024 * sum(wr) / len(wr)
        traffic[c] = {"read_bytes": rb, "write_bytes": wb, "bytes": rb + wb, "launches_profiled": len(fr)}
tot_r = 2 * 1024 * sum(float(r["Counter_Value"]) for r in frows) / steps
tot_w = 1024 * sum(float(r["Counter_Value"]) for r in wrows) / steps
trace = glob.glob(f"{src}/stats/*/*kernel_trace.csv")
if trace:
    trows = list(csv.DictReader(open(trace[0])))
    for c in traffic:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trows if class_of(r["Kernel_Name"]) == c]
        if d:
            traffic[c]["avg_us_rocprofv3"] = sum(d) / len(d)
with open(f"profiles/{tag}_summary.md", "a") as fh:
    fh.write("\n| kernel class | launches/step | avg µs (kernel trace) | HBM read MB / launch | HBM write MB / launch |\n|---|---|---|---|---|\n")
    for c, t in traffic.items():
        fh.write("| `%s>` | %d | %.1f | %.1f | %.1f |\n" % (c, len(bench.class_table()[c]), t.get("avg_us_rocprofv3", float("nan")), t["read_bytes"] / 1e6, t["write_bytes"] / 1e6))
    if frows and wrows:
        alg = bench.TRAIN_BYTES_PER_PATCH * 64
        fh.write("\nWhole step (all kernels, PMC): %.2f GB read + %.2f GB written = **%.2f GB = %.2fx the algorithmic %.2f GB** (SURVEY.md section 8 d).\n"
                 % (tot_r / 1e9, tot_w / 1e9, (tot_r + tot_w) / 1e9, (tot_r + tot_w) / alg, alg / 1e9))
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tag {tag}; FETCH_SIZE doubled (gfx950)",
           "step": {"read_bytes": tot_r, "write_bytes": tot_w, "bytes": tot_r + tot_w, "algorithmic_bytes": bench.TRAIN_BYTES_PER_PATCH * 64},
           "per_launch": traffic}, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))

# ---- category roll-up ----
cats = collections.OrderedDict([
    ("conv fwd (MFMA)", lambda k: any(t in k and k.split(t)[1].split(",")[1].strip().startswith("false") for t in ("conv3x3_mfma_kernel<", "conv3x3_wino8_kernel<"))),
    ("conv dgrad + wgrad, one kernel (16->16 layers)", lambda k: "conv3x3_bwd16" in k),
    ("conv dgrad (MFMA)", lambda k: any(t in k and k.split(t)[1].split(",")[1].strip().startswith("true") for t in ("conv3x3_mfma_kernel<", "conv3x3_wino8_kernel<"))),
    ("dgrad border", lambda k: "dgrad_border" in k),
    ("conv wgrad (MFMA)", lambda k: "conv3x3_wgrad" in k),
    ("wgrad slab reduce", lambda k: "wgrad_reduce" in k or "wgrad_wino_reduce" in k or "wgrad_wino_finish" in k),
    ("fused tail (outlay bwd + BN bwd)", lambda k: "tail_bwd" in k),
    ("BatchNorm backward", lambda k: "bn_bwd" in k),
    ("BatchNorm finalize / eval", lambda k: "bn_finalize" in k or "bn_eval" in k or "nbt_" in k),
    ("thin convs (in/out)", lambda k: "conv_in" in k or "conv_out" in k or "sum_partials" in k),
    ("pool / upsample / residual", lambda k: "pool2" in k or "up2x" in k or "bnrelu_add" in k),
    ("SIF loss", lambda k: "sif_loss" in k or "loss_finalize" in k or "blur" in k or "huber" in k or "sobel" in k),
    ("Adam + weight pack", lambda k: "adam" in k or "pack_weights" in k or "pack_wino" in k or "pack_all" in k),
])
agg = collections.OrderedDict((c, 0.0) for c in cats); agg["other (torch cat/copies)"] = 0.0
for r in rows:
    k = short(r["Name"]) + ("<" + r["Name"].split("<", 1)[1] if "<" in r["Name"] else "")
    for c, f in cats.items():
        if f(r["Name"]):
            agg[c] += float(r["TotalDurationNs"]); break
    else:
        agg["other (torch cat/copies)"] += float(r["TotalDurationNs"])
extra = ["", "| category | ms/step | % |", "|---|---|---|"]
for c, v in agg.items():
    extra.append("| %s | %.3f | %.1f |" % (c, v / 1e6 / steps, 100 * v / tot))
open(f"profiles/{tag}_summary.md", "a").write("\n".join(extra) + "\n")
print("\n".join(extra))
