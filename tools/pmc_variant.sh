#!/bin/bash
# Run ON THE GPU BOX: SQ counters of k_normals for one library variant (build/variants/libgm_hip_<tag>.so; "default" = the
# in-tree library).  usage: tools/pmc_variant.sh <tag> [stage_times args...]   -> gpurun_out/pmc_<tag>.json
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$TAG" != default ]; then export GM_LIB_PATH=$ROOT/build/variants/libgm_hip_$TAG.so; fi
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/stage_times.py --reps 3 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
cd $ROOT
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
out_dir, tag = sys.argv[1], sys.argv[2]
c = {}
dur = []
for f in glob.glob(os.path.join(out_dir, "p*", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gm::k_normals<" in r["Kernel_Name"] or r["Kernel_Name"].startswith("gm::k_normals("):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        c[k] = sum(v) / len(v)
for f in glob.glob(os.path.join(out_dir, "p*", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gm::k_normals<" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
c["kernel_us_under_pmc"] = sum(dur) / max(1, len(dur))
json.dump(c, open(os.path.join(os.path.dirname(out_dir), "pmc_%s.json" % tag), "w"), indent=1)
print(tag, json.dumps({k: round(v, 1) for k, v in sorted(c.items())}))
PY
