#!/bin/bash
# Builds libgm_hip.so of a git ref into build/variants/libgm_hip_<tag>.so (a throw-away worktree under /tmp), for A/B timing of
# two states of the tree on ONE GPU box (boxes differ by several per cent): GM_LIB_PATH selects the variant.
# usage: tools/ab_refs.sh <tag> <git-ref>      (the working tree itself: tools/ab_refs.sh <tag> WORKTREE)
set -e
TAG=$1; REF=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/variants
if [ "$REF" = WORKTREE ]; then
  make -C $ROOT/geometric_mapping_amd/csrc -j8 > /dev/null
  cp $ROOT/geometric_mapping_amd/libgm_hip.so $ROOT/build/variants/libgm_hip_$TAG.so
else
  WT=/tmp/gm_wt_$TAG
  rm -rf $WT; git -C $ROOT worktree prune; git -C $ROOT worktree add -f --detach $WT $REF > /dev/null 2>&1
  make -C $WT/geometric_mapping_amd/csrc -j8 > /dev/null
  cp $WT/geometric_mapping_amd/libgm_hip.so $ROOT/build/variants/libgm_hip_$TAG.so
  git -C $ROOT worktree remove --force $WT
fi
echo "built $TAG from $REF"
