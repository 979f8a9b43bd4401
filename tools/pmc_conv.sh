#!/bin/bash
# GPU box: SQ counters of ONE conv layer (tools/bench_conv.py args), two passes.  bash tools/pmc_conv.sh TAG fwd 16 16 256 64 10 wino
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/bench_conv.py "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/bench_conv.py "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k)
        g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        for n, v in sorted(c.items()):
            m = sum(v) / len(v)
            print(f"   {n:34s} {m:16.0f}   / GUI_ACTIVE = {m / g:9.3f}")
PY
