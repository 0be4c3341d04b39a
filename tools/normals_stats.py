#!/usr/bin/env python3
"""Candidate-stream accounting of k_normals (diagnostic build: make EXTRA=-DGM_NORMALS_STATS).
Prints how many candidates the waves streamed against how many were true neighbours."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--radius", type=float, default=None)
a = ap.parse_args()
r = a.radius or synth.fixed_k_radius(a.points)
xyz = synth.tunnel_frame(a.points, seed=0)
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS, max_points=a.points) as c:
    res = c.process_frame(xyz)
    counts = c.neighbor_counts()
    out = (ctypes.c_uint32 * 32)()
    lib = _lib.load()
    lib.gm_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    rc = lib.gm_debug_counters(c._ctx, 0, out)
    d = list(out)
n = res["n_cropped"]
# DevCounters: n_cropped, n_valid, n_tiles, reserved0, n_voxels, vox_n, mm[6], scratch_total, pad[5]
tiles = d[2]
streamed, groupsum, staged, chunks = d[13], d[14], d[15], d[16]
hits = int(counts.astype(np.int64).sum())
print(json.dumps({"rc": rc, "impl": os.environ.get("GM_NORMALS_IMPL", "auto"), "n_cropped": n, "tiles": tiles, "tiles_on_valu": d[17], "lane_fill": n / (64.0 * tiles), "chunks_per_tile": chunks / tiles,
                  "mean_neighbours": hits / n, "wave_candidates_per_tile": streamed / tiles,
                  "mean_group_window_per_tile": groupsum / 4 / tiles, "staged_per_tile": staged / tiles,
                  "hit_rate_active_lanes": hits / (streamed * 64.0 * n / (64.0 * tiles)),
                  "hit_rate_all_lanes": hits / (streamed * 64.0)}))
