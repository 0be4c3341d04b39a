#!/usr/bin/env python3
"""Tile timeline of k_normals (diagnostic build: make EXTRA=-DGM_NORMALS_TIMELINE): start / end tick of every tile with
>= 2 queries (wall_clock64 ticks; the tool scales them with the kernel duration it measures itself), from which the duration distribution, the number of tiles in flight over time and
the length of the drain at the end of the kernel follow."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
a = ap.parse_args()
r = synth.fixed_k_radius(a.points)
xyz = synth.tunnel_frame(a.points, seed=0)
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS, max_points=a.points) as c:
    c.process_frame(xyz)
    res = c.process_frame(xyz)
    counts = c.neighbor_counts().astype(np.int64)
kernel_us = res["normals_kernel_ms"] * 1e3
neg = -counts[counts < 0]
ends = neg[neg >= (1 << 29)] & 0x1FFFFFFF          # lane 0: end tick (bit 29 marks it)
durs = (neg[neg < (1 << 29)] - 1).astype(np.float64)  # lane 1: duration, ticks
# tick length: the first tile ends ~one tile after the kernel starts, the last one when it ends
us_per_tick = kernel_us / float(ends.max() - ends.min() + np.median(durs))
durs *= us_per_tick
rel = (ends - ends.max()) * us_per_tick             # time before the last tile ends, microseconds (<= 0)
print(json.dumps({"tiles_reporting": int(len(ends)), "kernel_us": round(kernel_us, 1), "us_per_tick": us_per_tick,
                  "wave_slot_occupancy": round(float(durs.sum() / (kernel_us * 4096.0)), 3),
                  "tile_duration_us": {str(q): round(float(np.quantile(durs, q)), 1) for q in (0.01, 0.1, 0.5, 0.9, 0.99, 1.0)},
                  "tiles_still_running_before_kernel_end": {"-%d us" % t: int((rel > -t).sum()) for t in (5, 10, 20, 40, 80, 160)}}))
hist, edges = np.histogram(-rel, bins=16)
for h, e0, e1 in zip(hist, edges[:-1], edges[1:]):
    print("ends %6.1f - %6.1f us before the kernel ends: %6d tiles" % (e0, e1, h))
