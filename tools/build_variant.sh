#!/bin/bash
# Build a variant of the library with extra -D flags on ONE translation unit (same-device A/B through SIFSR_LIB):
#   bash tools/build_variant.sh NAME conv_mfma.hip -DSIFSR_DBG_X   ->  tools/ab/libsifsr_NAME.so
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
PKG=$(ls -d $ROOT/land-surface*_amd)
mkdir -p $ROOT/tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden "$@" -c $PKG/csrc/$SRC -o /tmp/variant_$NAME.o
OBJS=$(ls $PKG/_obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/libsifsr_$NAME.so $OBJS /tmp/variant_$NAME.o
echo $ROOT/tools/ab/libsifsr_$NAME.so
