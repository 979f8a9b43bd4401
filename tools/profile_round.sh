#!/bin/bash
# Run on the GPU box (via gpurun):  bash tools/profile_round.sh r01
# kernel-trace/stats pass + two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-solo --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || exit 1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
echo "write pass done"
tail -1 $OUT/stats.log | cut -c1-300
