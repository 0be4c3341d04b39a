#!/bin/bash
# Run ON THE GPU BOX: HBM bytes per launch of every kernel of a frame, one frame at a time (FETCH_SIZE / WRITE_SIZE in
# separate passes, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2 on gfx950, KiB units).
# usage: tools/pmc_hbm_frame.sh <out-tag> [stage_times args...]   -> gpurun_out/pmc_hbm_<tag>.json
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_hbm_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $ROOT/tools/stage_times.py --reps 5 "$@" > $OUT/$c.log 2>&1 || echo "$c pass failed (see $OUT/$c.log)"
done
cd $ROOT
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
out_dir, tag = sys.argv[1], sys.argv[2]
acc = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
for c in acc:
    for f in glob.glob(os.path.join(out_dir, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[c][r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
res = {}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f, w = acc["FETCH_SIZE"].get(k, []), acc["WRITE_SIZE"].get(k, [])
    fm = sum(f) / len(f) if f else 0.0
    wm = sum(w) / len(w) if w else 0.0
    res[k] = {"launches": max(len(f), len(w)), "fetched_MB": round(2.0 * fm * 1024 / 1e6, 2), "written_MB": round(wm * 1024 / 1e6, 2),
              "hbm_MB_per_launch": round((2.0 * fm + wm) * 1024 / 1e6, 2)}
json.dump(res, open(os.path.join(os.path.dirname(out_dir), "pmc_hbm_%s.json" % tag), "w"), indent=1)
tot = 0.0
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_MB_per_launch"]):
    if k.startswith("void gm::") or k.startswith("gm::"):
        print("%-60s %8.2f MB  (fetch %7.2f  write %7.2f)" % (k[:60], v["hbm_MB_per_launch"], v["fetched_MB"], v["written_MB"]))
PY
