#!/bin/bash
# Build the library AS OF a git revision for a same-device A/B against the working tree:
#   bash tools/build_ref.sh HEAD prev   ->  tools/ab/libsifsr_prev.so   (sources exported to /tmp/sifsr_ref_prev)
set -e
REF=$1; NAME=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
PKG=$(basename $(ls -d $ROOT/land-surface*_amd))
rm -rf /tmp/sifsr_ref_$NAME && mkdir -p /tmp/sifsr_ref_$NAME $ROOT/tools/ab
cd $ROOT && git archive $REF $PKG/csrc include | tar -x -C /tmp/sifsr_ref_$NAME
python - "$NAME" "$PKG" <<'PY'
import importlib, sys
name, pkg = sys.argv[1], sys.argv[2]
b = importlib.import_module(pkg + ".build")
print(b.build(force=True, lib=f"tools/ab/libsifsr_{name}.so", obj_dir=f"/tmp/sifsr_ref_{name}/obj", csrc=f"/tmp/sifsr_ref_{name}/{pkg}/csrc"))
PY
