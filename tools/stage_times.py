#!/usr/bin/env python3
"""Per-stage device time of one frame (hipEvent brackets inside libgm_hip.so)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--radius", type=float, default=None)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--flags", type=int, default=0)
a = ap.parse_args()
r = a.radius or synth.fixed_k_radius(a.points)
xyz = synth.tunnel_frame(a.points, seed=0)
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_STAGE_TIMING | a.flags, max_points=a.points) as c:
    for _ in range(3):
        c.process_frame(xyz)
    acc = {}
    for _ in range(a.reps):
        res = c.process_frame(xyz)
        for k, v in res["stage_ms"].items():
            acc[k] = acc.get(k, 0.0) + v / a.reps
print(json.dumps({"points": a.points, "radius": r, "n_cropped": res["n_cropped"], "n_valid": res["n_valid"],
                  "n_voxels": res["n_voxels"], "stage_ms": {k: round(v, 4) for k, v in acc.items()}}))
