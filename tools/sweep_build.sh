#!/bin/bash
# usage (on the GPU box): tools/sweep_build.sh "-DA=1" "-DA=2" ...   -- rebuild k_normals with each flag set, time the
# stage alone and the pipelined bench step
cd $GRAFT_REPO_ROOT
for F in "$@"; do
  touch geometric_mapping_amd/csrc/k_normals.hip
  make -C geometric_mapping_amd/csrc EXTRA="$F" > gpurun_out/sweep_build.log 2>&1 || { echo build failed; tail gpurun_out/sweep_build.log; exit 1; }
  echo "FLAGS $F"
  timeout -k 10 120 python tools/stage_times.py --flags 0 --reps 20 2>/dev/null | tail -1 | grep -o '"normals": [0-9.]*' || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done
