#!/bin/bash
# GPU box: stand-alone timings of the wide conv layers for several library builds: bash tools/ab_layers.sh "w8p1 w8p2"
for n in cur $1; do
  if [ $n = cur ]; then export SIFSR_LIB=""; else export SIFSR_LIB=$PWD/tools/ab/libsifsr_$n.so; fi
  for cfg in "fwd 64 32 128" "fwd 128 64 64" "fwd 16 32 128" "fwd 32 64 64" "fwd 64 64 32" "dgrad 32 64 128" "dgrad 64 128 64"; do
    set -- $cfg
    echo -n "$n: "; NBLK=512 timeout -k 10 120 python tools/bench_conv.py $1 $2 $3 $4 64 100 2>/dev/null | tail -1
  done
done
