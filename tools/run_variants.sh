#!/bin/bash
# Run ON THE GPU BOX: times the neighbourhood kernel of each library variant built by tools/build_variants.sh.
# usage: tools/run_variants.sh [-t "<pytest -k expr>"] <tag> [<tag> ...]     (results: gpurun_out/variants_<tag>.txt)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
KEXPR=""
if [ "$1" = "-t" ]; then KEXPR=$2; shift 2; fi
for tag in "$@"; do
  export GM_LIB_PATH=$PWD/build/variants/libgm_hip_$tag.so
  out=gpurun_out/variants_$tag.txt
  echo "== $tag" | tee $out
  if [[ $tag == ph_* ]]; then
    timeout -k 10 120 python tools/normals_phases.py 2>&1 | tail -1 | tee -a $out
    timeout -k 10 120 python tools/normals_phases.py --radius 0.5 2>&1 | tail -1 | tee -a $out
    continue
  fi
  if [ -n "$KEXPR" ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "$KEXPR" 2>&1 | tail -3 | tee -a $out
  fi
  timeout -k 10 120 python tools/stage_times.py --flags 0 --reps 20 2>/dev/null | tail -1 | grep -o '"normals": [0-9.]*' | tee -a $out
  timeout -k 10 120 python tools/stage_times.py --flags 0 --reps 10 --radius 0.5 2>/dev/null | tail -1 | grep -o '"normals": [0-9.]*' | sed 's/normals/normals_r0.5/' | tee -a $out
  for rep in 1 2 3; do timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | grep -o "\"ms_per_step\": [0-9.]*" | head -1 | tee -a $out; done
done
