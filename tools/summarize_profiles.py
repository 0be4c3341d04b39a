#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/collect_profiles.sh into small files under
gpurun_out/profiles_<tag>/summary/ (copy those into profiles/ to commit them)."""
import collections, csv, glob, json, os, sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)

def one(pattern):
    f = glob.glob(os.path.join(out, pattern), recursive=True)
    return f[0] if f else None

# 1. kernel stats (per-kernel calls / average ns / share)
ks = one("trace/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(ks))) if ks else []
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
    for r in rows:
        w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

# 2. PMC: FETCH_SIZE / WRITE_SIZE per kernel per dispatch (KiB units -> bytes; gfx950: FETCH_SIZE
#    reads half of a wide coalesced stream, so it is doubled -- MI355X_MICROARCH.md par. HBM)
def pmc(pattern, counter):
    f = one(pattern)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc

fetch = pmc("pmc_fetch/**/*counter_collection.csv", "FETCH_SIZE")
write = pmc("pmc_write/**/*counter_collection.csv", "WRITE_SIZE")
bench = {}
for name in ("bench_trace.log",):
    p = os.path.join(out, name)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                bench = json.loads(line)
summary = {"tag": tag, "bench_line_under_kernel_trace": bench, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fv, wv = fetch.get(k, []), write.get(k, [])
    fm = sum(fv) / len(fv) if fv else 0.0
    wm = sum(wv) / len(wv) if wv else 0.0
    summary["kernels"][k] = {"dispatches": max(len(fv), len(wv)), "FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm,
                             "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0,
                             "correction": "FETCH_SIZE x2 (gfx950 wide-load half count), WRITE_SIZE x1"}
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_hbm.json"), "w"), indent=1)
# (the production kernel is a template since round 2: "void gm::k_normals<false>")
kn = next((v for k, v in summary["kernels"].items() if "gm::k_normals<" in k or k == "gm::k_normals"), {})
cfg = bench.get("config", {})
json.dump({"points": int(cfg.get("workload", "0").split("-")[0]) if cfg else None, "radius": cfg.get("neighborRadius"),
           "kernel": "gm::k_normals", "hbm_bytes_per_launch": kn.get("hbm_bytes_per_launch"),
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {tag}"},
          open(os.path.join(dst, f"{tag}_normals_pmc.json"), "w"), indent=1)
for r in rows[:8]:
    print(r["Name"].split("(")[0][:40].ljust(40), r["Calls"].rjust(5), "%10.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
print(json.dumps(kn))
print(json.dumps({k: bench.get(k) for k in ("value", "ms_per_step")}), json.dumps(bench.get("roofline")))
