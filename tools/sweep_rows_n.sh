#!/bin/bash
# rows per radius vs k_normals stage time over frame sizes at r = 0.5 (validates the host heuristic); GPU box
for n in 100000 300000 500000; do
  for d in 1 2 4; do
    GM_NORMALS_ROWS=$d python3 tools/stage_times.py --points $n --radius 0.5 --reps 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n', $n, 'rows', $d, 'normals_ms', d['stage_ms']['normals'])"
  done
  python3 tools/stage_times.py --points $n --radius 0.5 --reps 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n', $n, 'rows auto', 'normals_ms', d['stage_ms']['normals'])"
done
