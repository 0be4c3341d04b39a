#!/bin/bash
# Run ON THE GPU BOX: the pipelined step by frames in flight and sort block shape (200 steps each, one box).
for st in 512 1024; do
  for sl in 3 4 5 6; do
    v=$(GM_SORT_THREADS=$st python bench.py --steps 200 --warmup 10 --slots $sl --fixed-slots --no-secondary --no-cpu-baseline --group-points 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3))")
    echo "sort_threads=$st slots=$sl ms_per_step,Gpts: $v"
  done
done
