#!/bin/bash
# GPU box: tail_bwd_reduce_kernel's time in the ablation builds (tools/build_ab.sh tablN -DTAIL_ABL=N), rocprofv3 kernel stats of tools/bench_edges.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in cur "$@"; do
  if [ $n = cur ]; then export SIFSR_LIB=""; else export SIFSR_LIB=$ROOT/tools/ab/libsifsr_$n.so; fi
  rm -rf /tmp/tr_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$n -- python3 $ROOT/tools/bench_edges.py > /tmp/tr_$n.log 2>&1 || { tail -5 /tmp/tr_$n.log; exit 1; }
  python3 - $n <<'PY'
import csv, glob, sys
n = sys.argv[1]
f = glob.glob(f"/tmp/tr_{n}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "tail_bwd" in r["Name"]:
        print(n, r["Name"].split("(")[1][-30:] if False else r["Name"][28:60], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
