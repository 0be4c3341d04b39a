#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of one frame at a time (tools/stage_times.py), printed as a table.
# usage: tools/kernel_times.sh <out-tag> [stage_times args...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/stage_times.py --reps ${REPS:-20} "$@" > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; exit 1; }
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    per_frame = float(r["TotalDurationNs"]) / (3.0 + float(os.environ.get("REPS", "20"))) / 1e3     # 3 warm-ups + 20 frames
    tot += per_frame
    print("%-60s calls %5s  avg %8.1f us  per frame %8.1f us" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, per_frame))
print("sum per frame %.1f us" % tot)
PY
tail -1 $OUT/run.log
