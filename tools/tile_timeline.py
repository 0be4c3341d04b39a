#!/usr/bin/env python3
"""Wave timeline of k_normals (diagnostic build: tools/build_variants.sh tl "-DGM_NORMALS_TIMELINE", GM_LIB_PATH=...).

Every wave of the kernel's straight-line copy leaves three 100 MHz wall-clock ticks: its first instruction, the moment
it holds its tile (kernel arguments, the tile list's counters, the tile), and its end.  From them: how fast the chip fills,
how many waves are resident over time, how long a slot stays empty between two waves, how the launch drains -- and what
another order of the same tiles would buy (greedy list scheduling of the measured durations on the slots)."""
import argparse, ctypes, heapq, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--radius", type=float, default=None)
a = ap.parse_args()
r = a.radius if a.radius else synth.fixed_k_radius(a.points)
xyz = synth.tunnel_frame(a.points, seed=0)
lib = _lib.load()
if not hasattr(lib, "gm_debug_timeline"):
    sys.exit("this libgm_hip.so is not a -DGM_NORMALS_TIMELINE build")
lib.gm_debug_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32]
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT, max_points=a.points) as c:
    c.process_frame(xyz)
    res = c.process_frame(xyz)
    buf = (ctypes.c_uint64 * (3 * 65536))()
    rc = lib.gm_debug_timeline(buf, 65536)
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
t = t[t[:, 2] > 0]
# the buffer is not cleared between launches: keep the waves of the LAST launch (their ticks are the largest)
last = t[:, 0] > t[:, 2].max() - 100_000      # within 1 ms of the last end
t = t[last]
us = 0.01                                     # 100 MHz
e0 = t[:, 0].min()
entry, have, end = (t[:, 0] - e0) * us, (t[:, 1] - e0) * us, (t[:, 2] - e0) * us
span = end.max()
kernel_us = res["normals_kernel_ms"] * 1e3
q = lambda x, qs=(0.01, 0.1, 0.5, 0.9, 0.99): {str(p): round(float(np.quantile(x, p)), 2) for p in qs}
print(json.dumps({"waves": int(len(t)), "kernel_us_hipEvents": round(kernel_us, 1), "first_entry_to_last_end_us": round(float(span), 1),
                  "entry_to_tile_known_us": q(have - entry), "tile_us": q(end - have),
                  "slot_time_held_by_tiles": round(float((end - have).sum() / (span * 4096.0)), 3),
                  "slot_time_held_by_waves": round(float((end - entry).sum() / (span * 4096.0)), 3)}))
marks = (1, 2, 5, 10, 20, 30, 40, 60, 80, 100, 120, 140)
print(json.dumps({"waves_entered_after_us": {str(m): int((entry <= m).sum()) for m in marks},
                  "waves_resident_at_us": {str(m): int(((entry <= m) & (end > m)).sum()) for m in marks}}))
print(json.dumps({"still_running_before_the_end": {"-%d us" % m: int((end > span - m).sum()) for m in (5, 10, 20, 30, 40, 60)}}))
# slot turnover: the k-th wave to enter after the first 4096 takes the slot of (roughly) the k-th wave to end
order_in = np.sort(entry)
order_out = np.sort(end)
n_first = int((entry <= 5.0).sum())
if len(order_in) > n_first + 100:
    gap = order_in[n_first:] - order_out[:len(order_in) - n_first]
    print(json.dumps({"first_fill_waves": n_first, "slot_empty_between_waves_us": q(gap)}))


def makespan(d, slots=4096, gap_us=0.0):
    h = [0.0] * slots
    heapq.heapify(h)
    for x in d:
        s = heapq.heappop(h)
        heapq.heappush(h, s + x + gap_us)
    return max(h)


d = end - entry
by_entry = d[np.argsort(entry)]
print(json.dumps({"list_scheduling_us": {"as_dispatched": round(makespan(by_entry), 1), "longest_first": round(makespan(np.sort(d)[::-1]), 1),
                                         "ideal_sum_over_slots": round(float(d.sum() / 4096.0), 1)}}))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(os.path.join("gpurun_out", "tile_timeline.npz"), ticks=t)
