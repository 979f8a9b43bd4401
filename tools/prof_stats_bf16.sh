#!/bin/bash
# Run on the GPU box (via gpurun):  bash tools/prof_stats_bf16.sh TAG [single]   -- kernel-trace/stats of the bf16 step (config 5)
set -o pipefail
TAG=${1:-r03_bf16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$2" = "single" ]; then export SIFSR_WGRAD_STREAM=0; fi
CMD="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo --no-also --dtype bf16"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || exit 1
tail -1 $OUT/stats.log | cut -c1-200
