"""Summarise tools/pmc_mfma.sh: per kernel, SQ_VALU_MFMA_BUSY_CYCLES relative to the registers-only calibration loop."""
import csv, glob, os, sys, collections

def load(d):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name = {}
    for f in cc:
        for r in csv.DictReader(open(f)):
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    return per, name, dur

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]

def main():
    out = sys.argv[1]
    per, name, dur = load(os.path.join(out, "cal"))
    cal = None
    print("calibration (tools/mfma_peak.hip, MFMA from registers only):")
    for d in sorted(per, key=int):
        c = per[d]
        if c.get("SQ_INSTS_MFMA", 0) <= 0: continue
        r = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"]
        print(f"  {short(name[d]):24s} {dur.get(d, 0):10.1f} us  MFMA_BUSY/GUI_ACTIVE = {r:.3f}   clock = GUI_ACTIVE/8/t = {c['GRBM_GUI_ACTIVE'] / 8 / dur[d] / 1e3:.3f} GHz"
              f"   busy cycles / MFMA inst = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_MFMA']:.2f}")
        cal = r
    per, name, dur = load(os.path.join(out, "step"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for d, c in per.items():
        k = short(name[d])
        cnt[k] += 1
        for kk, v in c.items(): agg[k][kk] += v
        agg[k]["us"] += dur.get(d, 0.0)
    print(f"\nbench step kernels (MFMA utilisation = MFMA_BUSY/GUI_ACTIVE normalised by the calibration ratio {cal:.3f}):")
    print(f"{'kernel':46s} {'n':>4s} {'avg us':>8s} {'MFMA util':>9s} {'clock GHz':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s}")
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        if c.get("SQ_INSTS_MFMA", 0) <= 0 or c["us"] < 50: continue
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"] / cal
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        print(f"{k:46s} {cnt[k]:4d} {c['us'] / cnt[k]:8.1f} {util:9.3f} {c['GRBM_GUI_ACTIVE'] / 8 / c['us'] / 1e3:9.3f} {c['SQ_WAIT_ANY'] / wc:8.3f} {c['SQ_WAIT_INST_ANY'] / wc:9.3f} {c['SQ_ACTIVE_INST_ANY'] / wc:7.3f}")

main()
