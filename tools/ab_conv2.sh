#!/bin/bash
PKG=land-surface-temperature-super-resolution-with-a-scale-invariance-free-neural-approach_amd
for v in "$@"; do
  cp tools/variants/conv_mfma_$v.hip.txt $PKG/csrc/conv_mfma.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  echo "== variant $v"
  for cfg in "fwd 16 16 256" "dgrad 16 16 256" "fwd 32 32 64" "fwd 128 64 64"; do
    timeout -k 10 120 python tools/bench_conv.py $cfg 64 20 || exit 1
  done
done
