#!/bin/bash
# k_normals time over GM_NORMALS_ROWS (y/z rows per radius) at a given radius; run on the GPU box
R=${1:-0.5}
for d in 1 2 3 4; do
  GM_NORMALS_ROWS=$d python3 tools/stage_times.py --points 1000000 --radius $R --reps 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows', $d, 'radius', d['radius'], 'normals_ms', d['stage_ms']['normals'], 'grid_ms', d['stage_ms']['grid'], 'n_valid', d['n_valid'])"
done
