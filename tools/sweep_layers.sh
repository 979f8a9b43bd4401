#!/bin/bash
# per-layer conv timing at B=64 (all MFMA layer shapes), one line each
for op in ${OPS:-fwd dgrad wgrad}; do
for cfg in "16 16 256" "32 16 256" "16 16 128" "16 32 128" "64 32 128" "32 16 128" "32 32 64" "32 64 64" "128 64 64" "64 32 64" "64 64 32"; do
  set -- $cfg
  NBLK=${NBLK:-512} timeout -k 10 120 python tools/bench_conv.py $op $1 $2 $3 64 10 || exit 1
done; done
