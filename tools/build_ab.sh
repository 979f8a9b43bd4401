#!/bin/bash
# Build a VARIANT of the whole library with extra compiler flags (same-device A/B through SIFSR_LIB):
#   bash tools/build_ab.sh NAME -DSIFSR_PK_MODE=0   ->  tools/ab/libsifsr_NAME.so   (objects under /tmp/sifsr_ab_NAME)
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/tools/ab
cd $ROOT && python - "$NAME" "$@" <<'PY'
import importlib, sys
name, extra = sys.argv[1], sys.argv[2:]
b = importlib.import_module("land-surface-temperature-super-resolution-with-a-scale-invariance-free-neural-approach_amd.build")
print(b.build(force=False, extra=extra, lib=f"tools/ab/libsifsr_{name}.so", obj_dir=f"/tmp/sifsr_ab_{name}"))
PY
