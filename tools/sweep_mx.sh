#!/bin/bash
# usage (on the GPU box): tools/sweep_mx.sh "<flags>" "<flags>" ...  -- rebuild k_normals.hip with each flag set and time
# the neighbourhood kernels (frames one at a time) for GM_NORMALS_IMPL = auto and mfma; the first entry also runs valu
cd $GRAFT_REPO_ROOT
first=1
for F in "$@"; do
  touch geometric_mapping_amd/csrc/k_normals.hip
  make -C geometric_mapping_amd/csrc EXTRA="$F" > gpurun_out/sweep_build.log 2>&1 || { echo build failed; tail gpurun_out/sweep_build.log; exit 1; }
  echo "FLAGS [$F]"
  if [ $first = 1 ]; then impls=valu,auto0,auto; first=0; else impls=auto0,auto; fi
  timeout -k 10 300 python tools/normals_ab.py --impls $impls 2>&1 | tail -1 || exit 1
done
