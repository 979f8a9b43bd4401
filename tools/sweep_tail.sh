#!/bin/bash
# GPU box: ablate parts of tail_bwd_reduce_kernel and read its time from rocprofv3
PKG=land-surface-temperature-super-resolution-with-a-scale-invariance-free-neural-approach_amd
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for abl in ${ABLS:-0}; do
  sed -i "s/^#define TAIL_ABL .*/#define TAIL_ABL $abl/" $ROOT/$PKG/csrc/fused_edges.hip
  (cd $ROOT && python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1) || exit 1
  rm -rf /tmp/pt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $ROOT/tools/bench_edges.py 1024 > /tmp/pt.log 2>&1 || exit 1
  f=$(find /tmp/pt -name '*kernel_stats.csv' | head -1)
  mkdir -p $ROOT/gpurun_out/abl && cp $f $ROOT/gpurun_out/abl/abl$abl.csv; echo "ABL=$abl done"
done
