#!/bin/bash
# Run ON THE GPU BOX: the pipelined bench step of several library variants, interleaved, 200 steps each.
# usage: tools/ab_multi.sh <rounds> <tag> [<tag> ...]
N=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in $(seq 1 $N); do
  for t in "$@"; do
    v=$(GM_LIB_PATH=$ROOT/build/variants/libgm_hip_$t.so python $ROOT/bench.py --steps 200 --warmup 10 --fixed-slots --no-secondary --no-cpu-baseline --group-points 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3))")
    echo "$t round $round: ms_per_step, Gpts/s: $v"
  done
done
