#!/bin/bash
# GPU box: A/B one environment switch of the library on the same device, interleaved:
#   bash tools/ab_env.sh SIFSR_WGRAD_WINO "0 1 2" [bench args]
VAR=$1; VALS=$2; shift 2
for i in 1 2 3; do
  for v in $VALS; do
    echo -n "$VAR=$v: "; env $VAR=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo "$@" 2>/dev/null | tail -1 | cut -c1-130
  done
done
