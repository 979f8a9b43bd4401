#!/bin/bash
# GPU box: A/B conv_wgrad.hip variants (tools/variants/conv_wgrad_*.hip.txt) on the SAME device
PKG=land-surface-temperature-super-resolution-with-a-scale-invariance-free-neural-approach_amd
for v in "$@"; do
  cp tools/variants/conv_wgrad_$v.hip.txt $PKG/csrc/conv_wgrad.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  echo "== variant $v"
  for cfg in "16 16 256" "32 16 256" "16 16 128" "16 32 128" "64 32 128" "32 32 64" "128 64 64" "64 64 32"; do
    NBLK=${NBLK:-512} timeout -k 10 120 python tools/bench_conv.py wgrad $cfg 64 20 || exit 1
  done
done
