#!/bin/bash
# GPU box: A/B conv_mfma.hip variants (tools/variants/*.hip.txt) on the SAME device: rebuild, per-layer timings, step time
# CFGS="fwd 16 16 256;dgrad 16 16 256" overrides the layer list
PKG=land-surface-temperature-super-resolution-with-a-scale-invariance-free-neural-approach_amd
CFGS=${CFGS:-"fwd 16 16 256;dgrad 16 16 256;fwd 32 16 256;fwd 64 32 128;fwd 128 64 64;dgrad 128 64 64;dgrad 32 32 64;fwd 64 64 32"}
for v in "$@"; do
  cp tools/variants/conv_mfma_$v.hip.txt $PKG/csrc/conv_mfma.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  echo "== variant $v"
  IFS=';' read -ra LIST <<< "$CFGS"
  for cfg in "${LIST[@]}"; do
    timeout -k 10 120 python tools/bench_conv.py $cfg 64 20 2>/dev/null || exit 1
  done
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c75-140
done
