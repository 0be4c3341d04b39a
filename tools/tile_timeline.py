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
pro = (neg[(neg >= (1 << 28)) & (neg < (1 << 29))] & 0xFFFFF).astype(np.float64)   # lane 2: kernel entry -> tile start, ticks
durs = (neg[neg < (1 << 28)] - 1).astype(np.float64)  # lane 1: duration, ticks
# tick length: the first tile ends ~one tile after the kernel starts, the last one when it ends
us_per_tick = kernel_us / float(ends.max() - ends.min() + np.median(durs))
durs *= us_per_tick
rel = (ends - ends.max()) * us_per_tick             # time before the last tile ends, microseconds (<= 0)
if len(pro):
    print(json.dumps({"wave_entry_to_tile_start_us": {str(q): round(float(np.quantile(pro * us_per_tick, q)), 2) for q in (0.1, 0.5, 0.9, 0.99)}}))
print(json.dumps({"tiles_reporting": int(len(ends)), "kernel_us": round(kernel_us, 1), "us_per_tick": us_per_tick,
                  "wave_slot_occupancy": round(float(durs.sum() / (kernel_us * 4096.0)), 3),
                  "tile_duration_us": {str(q): round(float(np.quantile(durs, q)), 1) for q in (0.01, 0.1, 0.5, 0.9, 0.99, 1.0)},
                  "tiles_still_running_before_kernel_end": {"-%d us" % t: int((rel > -t).sum()) for t in (5, 10, 20, 40, 80, 160)}}))
hist, edges = np.histogram(-rel, bins=16)
for h, e0, e1 in zip(hist, edges[:-1], edges[1:]):
    print("ends %6.1f - %6.1f us before the kernel ends: %6d tiles" % (e0, e1, h))

# ---- how fast the chip fills: tiles started / in flight over the first microseconds
_m = min(len(ends), len(durs))
_e = (ends[:_m] - ends.min()) * us_per_tick
_s = _e - durs[:_m]
_t0 = _s.min()
print(json.dumps({"ramp": {"%d us" % t: {"started": int((_s - _t0 <= t).sum()), "in_flight": int(((_s - _t0 <= t) & (_e - _t0 > t)).sum())} for t in (5, 10, 15, 20, 30, 40, 60, 80, 100, 120)}}))

# ---- what would another dispatch order buy?  Greedy list scheduling of the measured durations on 4096 wave slots
# (ignores that a tile runs faster on an emptier SIMD: an upper bound on the gain of reordering)
import heapq
def makespan(d, slots=4096):
    h = [0.0] * slots
    heapq.heapify(h)
    for x in d:
        t = heapq.heappop(h)
        heapq.heappush(h, t + x)
    return max(h)
# lanes 0 and 1 of a tile report in the same order: pair them by position
m = min(len(ends), len(durs))
start = (ends[:m] - ends.min()) * us_per_tick - durs[:m]
order = np.argsort(start)
sim = {"as_dispatched": makespan(durs[:m][order]), "longest_first": makespan(np.sort(durs[:m])[::-1]),
       "two_classes_long_first": makespan(np.concatenate([durs[:m][order][durs[:m][order] >= np.median(durs[:m])],
                                                          durs[:m][order][durs[:m][order] < np.median(durs[:m])]])),
       "ideal_sum_over_slots": float(durs[:m].sum() / 4096.0)}
print(json.dumps({"list_scheduling_us": {k: round(float(v), 1) for k, v in sim.items()}}))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(os.path.join("gpurun_out", "tile_timeline.npz"), ends=ends, durs=durs, us_per_tick=us_per_tick)
