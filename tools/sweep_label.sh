#!/bin/bash
# points per block of the label pass (partial rows for the finalizer) vs single-frame and pipelined time; GPU box
cd $GRAFT_REPO_ROOT
for v in 4096 2048 1024; do
  touch geometric_mapping_amd/csrc/k_ransac.hip
  make -C geometric_mapping_amd/csrc EXTRA="-DGM_LABEL_PTS=$v" > gpurun_out/sweep_build.log 2>&1 || exit 1
  echo -n "label pts/block $v: "
  python3 tools/stage_times.py --points 1000000 --flags 8 --reps 20 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ransac_ms', d['stage_ms']['ransac'], 'total', d['stage_ms']['total'])"
done
touch geometric_mapping_amd/csrc/k_ransac.hip; make -C geometric_mapping_amd/csrc > gpurun_out/sweep_build.log 2>&1
