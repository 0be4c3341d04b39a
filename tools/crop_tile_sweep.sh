#!/bin/bash
# Run ON THE GPU BOX: the crop compaction's time per tile shape (GM_CROP_TILE=<threads>x<items>[x<min waves per SIMD>]), rocprofv3 averages.
# usage: SHAPES="512x8 512x16 ..." tools/crop_tile_sweep.sh [<points> ...]        (results: gpurun_out/crop_tile_sweep.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/crop_tile_sweep.txt
: > $out
[ $# -eq 0 ] && set -- 10000000 1000000
for pts in "$@"; do
  for shape in ${SHAPES:-512x8 512x16 1024x8}; do
    export GM_CROP_TILE=$shape
    echo "== $shape $pts" | tee -a $out
    REPS=6 tools/kernel_times.sh ct_${shape}_$pts --points $pts --flags 0 2>&1 | grep "CropPred\|failed" | tee -a $out
  done
done
