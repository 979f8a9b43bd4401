#!/bin/bash
# GPU box: MFMA-pipe utilisation from SQ counters (north_star: "rocprof showing ... MFMA utilisation against peak").
#   1. calibration: tools/mfma_peak.hip (registers-only MFMA loop) under the same counters
#   2. the bench step
# bash tools/pmc_mfma.sh <tag>
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_mfma
mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 $ROOT/tools/mfma_peak.hip -o $OUT/mfma_peak || exit 1
$OUT/mfma_peak > $OUT/mfma_peak.txt 2>&1 || exit 1
cat $OUT/mfma_peak.txt
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/cal -- $OUT/mfma_peak > $OUT/cal.log 2>&1 || { tail -5 $OUT/cal.log; exit 1; }
echo "calibration pass done"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/step -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-solo --no-also > $OUT/step.log 2>&1 || { tail -5 $OUT/step.log; exit 1; }
echo "step pass done"
cd $ROOT && python3 tools/pmc_mfma_summary.py $OUT > $OUT/summary.txt && cat $OUT/summary.txt
rm -f $OUT/mfma_peak
