#!/bin/bash
# frames in flight vs step time (run on the GPU box)
for s in 2 3 4 6; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --slots $s > gpurun_out/slots_$s.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.loads(open('gpurun_out/slots_$s.json').read().strip().splitlines()[-1]); print($s, d['value'], d['ms_per_step'])"
done
