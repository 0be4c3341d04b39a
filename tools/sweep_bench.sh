#!/bin/bash
# usage (on the GPU box): tools/sweep_bench.sh "<flags>" ...  -- rebuild k_normals.hip with each flag set, report the pipelined
# bench step (3 frames in flight) and the kernel's exclusive time
cd $GRAFT_REPO_ROOT
for F in "$@"; do
  touch geometric_mapping_amd/csrc/*.hip
  make -C geometric_mapping_amd/csrc EXTRA="$F" > gpurun_out/sweep_build.log 2>&1 || { echo build failed; tail gpurun_out/sweep_build.log; exit 1; }
  echo -n "FLAGS [$F] "
  timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G pts/s %.3f  ms/step %.4f  k_excl %.4f  k_pipe %.4f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['avg_launch_ms_in_pipeline']))" || exit 1
done
touch geometric_mapping_amd/csrc/*.hip; make -C geometric_mapping_amd/csrc > gpurun_out/sweep_build.log 2>&1
