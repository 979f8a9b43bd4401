#!/bin/bash
# GPU box: A/B several builds of the library on the same device, interleaved: bash tools/ab_libs.sh "pk0 pk2" [bench args]
# ("cur" = the in-tree library; NAME -> tools/ab/libsifsr_NAME.so)
NAMES=$1; shift
for i in 1 2 3; do
  for n in cur $NAMES; do
    if [ $n = cur ]; then L=""; else L=$PWD/tools/ab/libsifsr_$n.so; fi
    echo -n "$n: "; SIFSR_LIB=$L python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo --no-also "$@" 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['ms_per_step_median'])"
  done
done
