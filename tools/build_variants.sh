#!/bin/bash
# Cross-compiles variants of libgm_hip.so that differ in the flags k_normals.hip (or another file: FILE=...) is built
# with, into build/variants/libgm_hip_<tag>.so (git-ignored, travels with gpurun); select one with GM_LIB_PATH.
# usage: tools/build_variants.sh <tag> "<flags>" [<tag> "<flags>" ...]      (FILE=all: every file is built with the flags)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/geometric_mapping_amd/csrc
FILE=${FILE:-k_normals}
OUT=$ROOT/build/variants
mkdir -p $OUT
make -C $SRC -j8 > /dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fno-fast-math -Wno-unused-parameter -Wno-unused-result -Wno-unused-value"
pids=()
while [ $# -ge 2 ]; do
  tag=$1; extra=$2; shift 2
  (
    objs=""
    for f in gm_api k_crop k_sort k_normals k_frame k_voxel k_ransac k_nearest gm_ext gm_group; do
      if [ $f = $FILE ] || [ $FILE = all ]; then
        /opt/rocm/bin/hipcc $FLAGS $extra -c $SRC/$f.hip -o $OUT/${f}_$tag.o
        objs="$objs $OUT/${f}_$tag.o"
      else objs="$objs $SRC/$f.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o $OUT/libgm_hip_$tag.so
    echo "built $tag: $extra"
  ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
