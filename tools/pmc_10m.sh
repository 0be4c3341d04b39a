#!/bin/bash
# Run ON THE GPU BOX: per-kernel PMC values of the 10 M-point frame, one counter per pass (never with a trace domain).
# usage: tools/pmc_10m.sh <out-tag> <counter> [<counter> ...]      (results: gpurun_out/pmc10m_<tag>.txt)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc10m_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $ROOT/tools/stage_times.py --points ${POINTS:-10000000} --flags ${FLAGS:-12} --reps 3 > $OUT/$c.log 2>&1 || echo "$c pass failed"
done
cd $ROOT
python3 - $OUT "$@" > gpurun_out/pmc10m_$TAG.txt <<'PY'
import collections, csv, glob, os, sys
out, counters = sys.argv[1], sys.argv[2:]
tab = collections.defaultdict(dict)
for c in counters:
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 3:]
        tab[k][c] = sum(v) / len(v)
print("kernel".ljust(56), *[c.rjust(16) for c in counters])
for k, d in sorted(tab.items()):
    print(k[:56].ljust(56), *[("%.4g" % d.get(c, float("nan"))).rjust(16) for c in counters])
PY
cat gpurun_out/pmc10m_$TAG.txt
