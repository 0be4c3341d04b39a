#!/bin/bash
# Run ON THE GPU BOX after tools/collect_profiles.sh: VALU / LDS / wait counters of k_normals (two passes of four
# counters; --pmc only with --kernel-trace) -> gpurun_out/profiles_<tag>/summary/<tag>_normals_valu_pmc.json
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_valu_a -- python3 $ROOT/tools/stage_times.py --reps 3 > $OUT/pmc_valu_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_valu_b -- python3 $ROOT/tools/stage_times.py --reps 3 > $OUT/pmc_valu_b.log 2>&1 || exit 2
# matrix-core counters (own pass; names differ between ROCm releases, so a failure here is not fatal)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_valu_c -- python3 $ROOT/tools/stage_times.py --reps 3 > $OUT/pmc_valu_c.log 2>&1 || echo "mfma counter pass failed (see pmc_valu_c.log)"
cd $ROOT
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
out_dir, tag = sys.argv[1], sys.argv[2]
c = {}
for d in ("pmc_valu_a", "pmc_valu_b", "pmc_valu_c"):
    for f in glob.glob(os.path.join(out_dir, d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gm::k_normals<" in r["Kernel_Name"] or r["Kernel_Name"].startswith("gm::k_normals("):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            c[k] = sum(v) / len(v)
simd = 1024 * c["GRBM_GUI_ACTIVE"] / 8
res = {"kernel": "gm::k_normals", "workload": "1 M-point frame, r = 0.1118 (tools/stage_times.py), mean per dispatch",
       "source": "rocprofv3 --pmc (two passes of four counters) --kernel-trace, " + tag, "counters": c,
       "derived": {"simd_cycles_available": simd, "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / simd,
                   "wave_cycles_waiting_any_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                   "lds_busy_frac": c["SQ_ACTIVE_INST_LDS"] * 4 / simd,
                   "mfma_busy_frac": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd) if "SQ_VALU_MFMA_BUSY_CYCLES" in c else None,
                   "note": "SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs"}}
json.dump(res, open(os.path.join(out_dir, "summary", tag + "_normals_valu_pmc.json"), "w"), indent=1)
print(json.dumps(res["derived"]), c["SQ_INSTS_VALU"])
PY
