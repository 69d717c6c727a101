#!/bin/bash
# dev: build _ab/<name>.so with extra flags for the two translation units of sdf_fused.hip (the other objects are reused)
#   tools/build_variant.sh <name> "<flags for sdf_fused.o>" "<flags for sdf_train.o>"
set -e
cd "$(dirname "$0")/../miso_amd/csrc"
name=$1; ff=$2; tf=$3
mkdir -p ../../_ab /tmp/ab_$name
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-variable -Wno-unused-but-set-variable"
/opt/rocm/bin/hipcc $BASE $ff -c sdf_fused.hip -o /tmp/ab_$name/sdf_fused.o &
/opt/rocm/bin/hipcc $BASE -DMISO_SDF_TRAIN_TU $tf -c sdf_fused.hip -o /tmp/ab_$name/sdf_train.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../_ab/$name.so $(ls *.o | grep -v -e sdf_train.o -e sdf_fused.o) /tmp/ab_$name/sdf_fused.o /tmp/ab_$name/sdf_train.o
echo built _ab/$name.so
