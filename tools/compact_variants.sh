#!/bin/bash
# Run ON THE GPU BOX: per-kernel times (rocprofv3, one frame at a time) of library variants that differ in the
# compaction tile (tools/build_variants.sh with FILE=all), on the 10 M- and the 1 M-point frame.
# usage: tools/compact_variants.sh <tag> [<tag> ...]      (results: gpurun_out/compact_variants.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/compact_variants.txt
: > $out
for tag in "$@"; do
  export GM_LIB_PATH=$ROOT/build/variants/libgm_hip_$tag.so
  for pts in 10000000 1000000; do
    echo "== $tag $pts" | tee -a $out
    REPS=6 tools/kernel_times.sh cv_${tag}_$pts --points $pts --flags 12 2>&1 | grep "k_compact\|k_label\|stage_ms\|failed" | tee -a $out
  done
done
