#!/bin/bash
# GPU box: per-layer conv timings (fwd / dgrad) of the in-tree library vs variants under tools/ab/
for cfg in "fwd 16 16 256" "fwd 32 16 256" "fwd 64 32 128" "dgrad 16 16 256" "dgrad 32 32 64"; do
  timeout -k 10 120 python tools/bench_conv.py $cfg 64 20 2>/dev/null || exit 1
  for v in ${VARIANTS:-fake}; do
    echo -n "   $v: "; SIFSR_DBG_CONV_GRID=${GRID:-0} SIFSR_LIB=$PWD/tools/ab/libsifsr_$v.so timeout -k 10 120 python tools/bench_conv.py $cfg 64 20 2>/dev/null || exit 1
  done
done
