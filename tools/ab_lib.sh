#!/bin/bash
# GPU box: A/B two builds of libsifsr_hip.so on the same device: bash tools/ab_lib.sh tools/ab/libsifsr_prev.so [bench args]
PREV=$1; shift
for i in 1 2 3; do
  echo -n "prev: "; SIFSR_LIB=$PWD/$PREV python bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | cut -c75-110
  echo -n "new : "; python bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | cut -c75-110
done
