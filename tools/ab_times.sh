#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel averages of two library variants, interleaved (A B A B), one frame at a time.
# usage: tools/ab_times.sh <tagA> <tagB> <grep-pattern> [stage_times args...]
A=$1; B=$2; PAT=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for t in $A $B; do
    echo "== $t (round $round)"
    GM_LIB_PATH=$ROOT/build/variants/libgm_hip_$t.so REPS=${REPS:-10} bash $ROOT/tools/kernel_times.sh ab_$t "$@" | grep -E "$PAT"
  done
done
