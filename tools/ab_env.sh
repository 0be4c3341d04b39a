#!/bin/bash
# Run ON THE GPU BOX: the pipelined bench step under two settings of an environment variable, interleaved, 200 steps each.
# usage: tools/ab_env.sh VAR valueA valueB [rounds]
V=$1; A=$2; B=$3; N=${4:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in $(seq 1 $N); do
  for t in $A $B; do
    v=$(env $V=$t python $ROOT/bench.py --steps 200 --warmup 10 --fixed-slots --no-secondary --no-cpu-baseline --group-points 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3))")
    echo "$V=$t round $round: ms_per_step, Gpts/s: $v"
  done
done
