#!/usr/bin/env python3
"""A lidar-shaped frame (synth.velodyne_tunnel) through the frame path with the launch file's values, for every y/z row
subdivision D of the search grid (GM_NORMALS_ROWS, read per context): does the host's choice (rows_per_radius,
csrc/gm_api.hip) still pick the best one when density varies 100x along a row?"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--rings", type=int, default=64)
ap.add_argument("--az", type=int, default=1800)
ap.add_argument("--radius", type=float, default=0.5)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
msg = synth.velodyne_tunnel(rings=a.rings, az=a.az, seed=7, point_step=32, row_pad=0)
rows = msg["data"]
n = a.rings * a.az
out = {"points": n, "radius": a.radius}
for d in ("auto", "1", "2", "3", "4"):
    if d == "auto":
        os.environ.pop("GM_NORMALS_ROWS", None)
    else:
        os.environ["GM_NORMALS_ROWS"] = d
    with g.GeometricMapping(neighborRadius=a.radius, voxelGridLeafSize=0.5, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_STAGE_TIMING,
                            max_points=n) as c:
        cl = c.cloud_from_rows(rows, n, 32, (0, 4, 8))
        for _ in range(3):
            r = c.process_frame(cl)
        km, tm = [], []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            r = c.process_frame(cl)
            tm.append((time.perf_counter() - t0) * 1e3)
            km.append(r["normals_kernel_ms"])
        out["D=" + d] = {"normals_kernel_ms": round(float(np.median(km)), 4), "frame_ms": round(float(np.median(tm)), 4),
                         "grid_ms": round(r["stage_ms"]["grid"], 4), "n_valid": r["n_valid"]}
print(json.dumps(out))
