"""Concurrency of a rocprofv3 kernel trace: wall time covered by >=1 / >=2 kernels, and a timeline of the last step.
usage: python tools/trace_overlap.py <dir with *_kernel_trace.csv> [--timeline]"""
import csv, glob, os, sys

def main():
    d = sys.argv[1]
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)   # newest run in the directory
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    # last step = from the last weight-pack kernel (the head of a forward) to the end
    idx = [i for i, r in enumerate(rows) if "pack_all_kernel" in r[2] or "pack_weights_kernel" in r[2]]
    lo = idx[-1]
    step = rows[lo:]
    t0 = step[0][0]
    ev = []
    for s, e, n, q in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    cover = [0, 0, 0, 0]
    cur = 0; last = ev[0][0]
    for t, dlt in ev:
        cover[min(cur, 3)] += t - last
        last = t; cur += dlt
    tot = step[-1][1] - t0
    print(f"{f}\nlast step: {len(step)} kernels, wall {tot/1e6:.3f} ms, sum of durations {sum(e-s for s,e,_,_ in step)/1e6:.3f} ms")
    print("idle %.3f ms, exactly one kernel %.3f ms, two %.3f ms, three+ %.3f ms" % tuple(c / 1e6 for c in cover))
    if "--timeline" in sys.argv:
        for s, e, n, q in step:
            print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} q{q} {n[:90]}")

main()
