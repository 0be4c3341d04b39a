#!/bin/bash
# Run ON THE GPU BOX: crop compaction time of library variants (tools/build_variants.sh, FILE=k_crop) on the 10 M frame.
# usage: tools/crop_diag.sh <tag> ...     (results: gpurun_out/crop_diag.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/crop_diag.txt
: > $out
for tag in "$@"; do
  export GM_LIB_PATH=$ROOT/build/variants/libgm_hip_$tag.so
  for pts in 10000000 1000000; do
    echo "== $tag $pts" | tee -a $out
    REPS=6 tools/kernel_times.sh cd_${tag}_$pts --points $pts --flags 0 2>&1 | grep "CropPred\|failed" | tee -a $out
  done
done
