#!/usr/bin/env python3
"""Per-kernel HBM roofline rows of the STREAMING kernels from a rocprofv3 kernel-stats CSV of one
frame size (tools/stage_times.py under rocprofv3 --kernel-trace --stats).

usage: streaming_roofline.py <kernel_stats.csv> <stage_times.log> <out.json> [<pmc_fetch_dir> <pmc_write_dir>]

With the two rocprofv3 --pmc output directories (FETCH_SIZE and WRITE_SIZE passes of the same command) every row also
carries the bytes the counters saw: hbm_bytes_pmc = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 counts a
wide coalesced read at half its bytes, MI355X_MICROARCH.md par. HBM).

Algorithmic bytes follow SURVEY.md par. 8(d) (each array touched once; float4 rows counted at the
12 B of x,y,z they carry, normals at 16 B); `bytes_moved` is what the kernel really loads/stores
(16-byte rows).  Peak: 8.0 TB/s spec, 6.29 TB/s measured copy peak (MI355X_MICROARCH.md)."""
import csv, json, sys

import collections, glob, os
stats, log, out = sys.argv[1:4]
pmc = {}
if len(sys.argv) >= 6:
    for d, name in ((sys.argv[4], "FETCH_SIZE"), (sys.argv[5], "WRITE_SIZE")):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == name:
                    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 3:]   # (drop the warm-up launches: smaller buffers are not touched yet)
            pmc.setdefault(k, {})[name] = sum(v) / len(v)
meta = {}
for line in open(log):
    if line.startswith("{"):
        meta = json.loads(line)
n_in, n_c, n_v = meta["points"], meta["n_cropped"], meta["n_valid"]
dur = {}
for r in csv.DictReader(open(stats)):
    dur[r["Name"].split("(")[0].replace("void ", "")] = (float(r["AverageNs"]) * 1e-9, int(r["Calls"]))

# kernel -> (reference step, algorithmic bytes, bytes really moved)
rows = {
    # round 2, second half: the compactions are single launches (chained scan); the NaN-normal removal also sums the
    # scatter-matrix terms of the survivors (getLocalFrame needs no pass of its own: its 16 B per point are not read again)
    "gm::k_compact<gm::CropPred, gm::CropEmit>": ("fromROSMsg + CropBox (order-preserving), one launch", 12 * n_in + 12 * n_c, 16 * n_in + 16 * n_c + 4 * n_c),
    "gm::k_compact<gm::ValidPred, gm::ValidEmit>": ("removeNaNNormals + ExtractIndices + getLocalFrame partial sums, one launch", 28 * n_c + 28 * n_v, 16 * n_c + 16 * n_v + 32 * n_v),
    "gm::k_compact_count<gm::CropPred>": ("CropBox predicate pass", 12 * n_in, 12 * n_in),
    "gm::k_compact_scatter<gm::CropPred, gm::CropEmit>": ("CropBox copy (order-preserving)", 12 * n_in + 12 * n_c, 12 * n_in + 16 * n_c + 4 * n_c),
    # since round 2 the predicate is a 1-byte flag per point written by k_normals (the 12 B of the normal it stands for are
    # counted where they are produced): the count pass streams 1 B per point
    "gm::k_compact_count<gm::ValidPred>": ("removeNaNNormals predicate pass (1-byte flags)", n_c, n_c),
    "gm::k_compact_scatter<gm::ValidPred, gm::ValidEmit>": ("removeNaNNormals + ExtractIndices", 28 * n_c + 28 * n_v, n_c + 32 * n_v + 32 * n_v),
    "gm::k_scatter_partials": ("getLocalFrame sum of (w n)(w n)^T", 16 * n_v, 16 * n_v),
    "gm::k_label<0>": ("plane inlier labelling", 12 * n_v + n_v, 16 * n_v + n_v),
    "gm::k_label<1>": ("cylinder inlier labelling", 12 * n_v + n_v, 16 * n_v + 2 * n_v),
    # frame pipeline since round 2: the label passes also sum their segment's moments (one pass less over the cloud);
    # algorithmic = labelling (12 + 1 B per point) + the per-segment covariance (12 B per point, SURVEY par. 8d)
    # (every array counted once: the moments reuse the points the labelling reads; the cylinder's normal moments read the
    #  normals of its inliers -- counted for every point, an upper bound)
    "gm::k_label<0, 1>": ("plane inlier labelling + plane-segment moments", 12 * n_v + n_v, 16 * n_v + n_v),
    "gm::k_label<1, 1>": ("cylinder inlier labelling + cylinder-segment moments (normals of the inliers)", 12 * n_v + n_v + 12 * n_v, 16 * n_v + 2 * n_v + 16 * n_v),
    # RANSAC scoring (SURVEY par. 8d: 12 N'' per pass, all hypotheses of the stage in one pass).  The last stage (8
    # hypotheses on every point) streams; the earlier stages see every 64th / 16th point: 12 B of each point they score
    "gm::k_score_stream<0, false>": ("plane RANSAC, last stage: 8 hypotheses on every point (streaming scorer)", 12 * n_v, 16 * n_v),
    "gm::k_score_stream<1, false>": ("cylinder RANSAC, last stage: 8 hypotheses on every point (streaming scorer)", 12 * n_v, 16 * n_v),
    "gm::k_score<0>": ("plane RANSAC, stage 1: 1024 hypotheses on every 64th point", 12 * (n_v // 64), 16 * (n_v // 64)),
    "gm::k_score<1>": ("cylinder RANSAC, stage 1: 1024 hypotheses on every 64th point", 12 * (n_v // 64), 16 * (n_v // 64)),
    "gm::k_score_sel<0, 256>": ("plane RANSAC, stage 2: 128 hypotheses on every 16th point", 12 * (n_v // 16), 16 * (n_v // 16)),
    "gm::k_score_sel<1, 256>": ("cylinder RANSAC, stage 2: 128 hypotheses on every 16th point", 12 * (n_v // 16), 16 * (n_v // 16)),
    "gm::k_segment_moments": ("per-segment covariance (points + normals of one label)", 24 * n_v + n_v, 32 * n_v + n_v),
    "gm::k_frame_moments": ("per-segment covariance, both segments in one pass (labels + points; normals of cylinder inliers)", 12 * n_v + n_v, 32 * n_v + n_v),
}
res = {"points": n_in, "n_cropped": n_c, "n_valid": n_v, "radius": meta.get("radius"),
       "peak_spec_GBs": 8000.0, "peak_measured_copy_GBs": 6290.0, "kernels": {}}
def find(table, key):
    """exact name, or the instantiation of it that carries further template arguments (the compactions' tile shape)"""
    if key in table:
        return table[key]
    if key.endswith(">") and "k_compact" in key:
        hits = [k for k in table if k.startswith(key[:-1] + ",")]
        if hits:
            return table[max(hits, key=lambda k: table[k][1] if isinstance(table[k], tuple) else 0)]
    return None
for k, (what, alg, moved) in rows.items():
    if find(dur, k) is None:
        continue
    t, calls = find(dur, k)
    res["kernels"][k] = {"reference_step": what, "avg_us": round(t * 1e6, 2), "calls": calls,
                         "algorithmic_bytes": alg, "algorithmic_GBs": round(alg / t / 1e9, 1),
                         "frac_of_spec": round(alg / t / 8.0e12, 3), "frac_of_measured_peak": round(alg / t / 6.29e12, 3),
                         "bytes_moved": moved, "moved_GBs": round(moved / t / 1e9, 1)}
    pk = find(pmc, k)
    if pk and "FETCH_SIZE" in pk and "WRITE_SIZE" in pk:
        pmc[k] = pk
        b = (2.0 * pmc[k]["FETCH_SIZE"] + pmc[k]["WRITE_SIZE"]) * 1024.0
        res["kernels"][k].update({"hbm_bytes_pmc": round(b), "pmc_GBs": round(b / t / 1e9, 1), "pmc_frac_of_spec": round(b / t / 8.0e12, 3),
                                  "FETCH_SIZE_KiB": round(pmc[k]["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(pmc[k]["WRITE_SIZE"], 1)})
json.dump(res, open(out, "w"), indent=1)
for k, v in res["kernels"].items():
    print(k[:52].ljust(52), "%8.1f us %8.1f GB/s alg (%.0f %% of spec)  %8.1f GB/s moved" %
          (v["avg_us"], v["algorithmic_GBs"], 100 * v["frac_of_spec"], v["moved_GBs"]))
