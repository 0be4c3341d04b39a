#!/bin/bash
# Run ON THE GPU BOX: the pipelined bench step of two library variants, interleaved (A B A B A B), 200 steps each.
# usage: tools/ab_bench.sh <tagA> <tagB>
A=$1; B=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for t in $A $B; do
    v=$(GM_LIB_PATH=$ROOT/build/variants/libgm_hip_$t.so python $ROOT/bench.py --steps 200 --warmup 10 --fixed-slots --no-secondary --no-cpu-baseline --group-points 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3), round(d['roofline']['kernel_avg_us'],1) if 'kernel_avg_us' in d['roofline'] else '')")
    echo "$t round $round: ms_per_step, Gpts/s: $v"
  done
done
