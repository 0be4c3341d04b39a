#!/usr/bin/env python3
"""Blocking per-frame latency (wall clock of gm_process_frame from host rows) over frame sizes; run on the GPU box."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=50)
ap.add_argument("--flags", type=int, default=0, help="extra GM_CFG_* bits (8 = cylinder RANSAC, 64 = graph replay)")
a = ap.parse_args()
for n, r in ((50_000, 0.5), (100_000, None), (300_000, None), (1_000_000, None)):
    r = r or synth.fixed_k_radius(n)
    xyz = synth.tunnel_frame(n, seed=0)
    with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | a.flags, max_points=n) as c:
        for _ in range(5):
            c.process_frame(xyz)
        ts = []
        for _ in range(a.frames):
            t0 = time.perf_counter()
            res = c.process_frame(xyz)
            ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.sort(ts)
    print(json.dumps({"points": n, "radius": round(r, 4), "n_valid": res["n_valid"], "median_ms": round(float(np.median(ts)), 4),
                      "p10_ms": round(float(ts[len(ts) // 10]), 4), "p90_ms": round(float(ts[len(ts) * 9 // 10]), 4)}))
