#!/bin/bash
# GM_NORMALS_XCD (blocks per XCD chunk of k_normals' tile mapping) vs pipelined step; run on the GPU box
for x in 0 8 16 32 64; do
  echo -n "xcd_chunk $x: "
  GM_NORMALS_XCD=$x timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G pts/s %.3f  ms/step %.4f  k_excl %.4f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
