#!/bin/bash
# dev: build _ab/<name>.so with extra flags for the two translation units of sdf_fused.hip (the other objects are reused)
#   [MISO_SRC=<dir>] tools/build_variant.sh <name> "<flags for sdf_fused.o>" "<flags for sdf_train.o>"
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"; OUT=$REPO/_ab
cd "${MISO_SRC:-$REPO/miso_amd/csrc}"      # (MISO_SRC: a scratch copy of the sources with an experiment patched in)
name=$1; ff=$2; tf=$3
mkdir -p $OUT /tmp/ab_$name
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-variable -Wno-unused-but-set-variable"
/opt/rocm/bin/hipcc $BASE $ff -c sdf_fused.hip -o /tmp/ab_$name/sdf_fused.o &
/opt/rocm/bin/hipcc $BASE -DMISO_SDF_TRAIN_TU $tf -c sdf_fused.hip -o /tmp/ab_$name/sdf_train.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$name.so $(ls *.o | grep -v -e sdf_train.o -e sdf_fused.o) /tmp/ab_$name/sdf_fused.o /tmp/ab_$name/sdf_train.o
echo built _ab/$name.so
