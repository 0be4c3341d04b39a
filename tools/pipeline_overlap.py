#!/usr/bin/env python3
"""How the kernels of frames in flight share the chip: from a rocprofv3 --kernel-trace of the pipelined bench
(kernel_trace.csv), over the steady middle of the run: time with no kernel running, with exactly one, average number
running, and per kernel its summed duration per frame (in-pipeline) -- against the step.
usage: pipeline_overlap.py <dir with *kernel_trace.csv> [frames to skip at both ends = 40]"""
import csv, glob, json, os, sys
d = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
crops = [r for r in rows if "CropPred" in r[0]]
t0, t1 = crops[skip][1], crops[-skip][1]
frames = len(crops) - 2 * skip
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
ev = []
for _, s, e in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, t0, {}
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    last = t; cur += dlt
span = t1 - t0
per = {}
for nm, s, e in sel:
    per[nm] = per.get(nm, 0) + (e - s)
out = {"frames": frames, "step_us": round(span / frames / 1e3, 2), "idle_frac": round(hist.get(0, 0) / span, 4),
       "one_kernel_frac": round(hist.get(1, 0) / span, 4), "avg_running": round(sum(k * v for k, v in hist.items()) / span, 3),
       "in_pipeline_us_per_frame": {k[:48]: round(v / frames / 1e3, 1) for k, v in sorted(per.items(), key=lambda kv: -kv[1])}}
print(json.dumps(out, indent=1))
