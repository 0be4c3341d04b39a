#!/usr/bin/env python3
"""Blocking frames from device-resident rows, enqueued or replayed from the captured hipGraph (--graph 1); run under
rocprofv3 --kernel-trace and feed the kernel_trace.csv to --analyse to see where a frame's wall time goes:
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/graph_trace.py --graph 1
  python3 tools/graph_trace.py --analyse OUT"""
import argparse, csv, glob, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--ransac", type=int, default=1)
ap.add_argument("--analyse", default=None)
a = ap.parse_args()

if a.analyse:
    f = glob.glob(os.path.join(a.analyse, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # frames: split at the crop kernel
    frames, cur = [], []
    for r in rows:
        name = r["Kernel_Name"]
        if "k_zero_fill" in name and cur:
            frames.append(cur); cur = []
        cur.append((name.split("(")[0].replace("void ", "")[:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    frames.append(cur)
    frames = [fr for fr in frames if len(fr) > 5][-a.frames:]
    span = [(fr[-1][2] - fr[0][1]) / 1e3 for fr in frames]
    busy = [sum(e - s for _, s, e in fr) / 1e3 for fr in frames]
    print(json.dumps({"frames": len(frames), "kernels_per_frame": len(frames[-1]), "span_us_median": sorted(span)[len(span) // 2],
                      "kernel_sum_us_median": sorted(busy)[len(busy) // 2]}))
    fr = frames[-1]
    for i, (nm, s, e) in enumerate(fr):
        gap = (s - fr[i - 1][2]) / 1e3 if i else 0.0
        print("%-42s start %8.1f  dur %7.1f  gap before %6.1f" % (nm, (s - fr[0][1]) / 1e3, (e - s) / 1e3, gap))
    sys.exit(0)

import numpy as np, torch
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
n = a.points
r = synth.fixed_k_radius(n)
xyz = synth.tunnel_frame(n, seed=0)
rows = np.zeros((n, 4), dtype=np.float32); rows[:, :3] = xyz
dev = torch.from_numpy(rows).cuda()
flags = _lib.GM_CFG_DEFAULT | (_lib.GM_CFG_GRAPH if a.graph else 0) | (_lib.GM_CFG_RANSAC_CYLINDER if a.ransac else 0)
with g.GeometricMapping(neighborRadius=r, flags=flags, max_points=n, ransac_threshold=0.03) as c:
    cloud = c.cloud_from_device(dev.data_ptr(), n, 16)
    for _ in range(5):
        c.process_frame(cloud)
    ts = []
    for _ in range(a.frames):
        t0 = time.perf_counter()
        c.process_frame(cloud)
        ts.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"graph": a.graph, "median_ms": round(float(np.median(ts)), 4)}))
