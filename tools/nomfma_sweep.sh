#!/bin/bash
# GPU box: every conv layer shape with and without its matrix work (tools/ab/libsifsr_nomfma.so = all three conv sources built with
# -DSIFSR_DIAG_NOMFMA: bash tools/build_ab.sh nomfma -DSIFSR_DIAG_NOMFMA): the second column is what the kernel's data movement alone costs.
for op in fwd dgrad wgradx; do
for cfg in "16 16 256" "32 16 256" "16 16 128" "16 32 128" "64 32 128" "32 16 128" "32 32 64" "32 64 64" "128 64 64" "64 32 64" "64 64 32"; do
  set -- $cfg
  a=$(NBLK=512 timeout -k 10 120 python tools/bench_conv.py $op $1 $2 $3 64 100 2>/dev/null | grep -o "[0-9.]* us")
  b=$(SIFSR_LIB=$PWD/tools/ab/libsifsr_nomfma.so NBLK=512 timeout -k 10 120 python tools/bench_conv.py $op $1 $2 $3 64 100 2>/dev/null | grep -o "[0-9.]* us")
  echo "$op $1->$2 @$3: $a | data movement only $b"
done; done
