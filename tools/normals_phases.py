#!/usr/bin/env python3
"""Where a wave of k_normals spends a tile (diagnostic build: tools/build_variants.sh ph "-DGM_NORMALS_PHASES").
Shader-clock ticks (s_memtime) summed over all tiles, per phase; the stamps themselves cost ~10 %.
Built with -DGM_PH_ASSEMBLE as well, the slot "chunk_load_wait" (empty otherwise: the rows are prefetched) holds the
assembly of the next chunk and "chunk_features" the feature image alone."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--radius", type=float, default=None)
a = ap.parse_args()
r = a.radius or synth.fixed_k_radius(a.points)
xyz = synth.tunnel_frame(a.points, seed=0)
lib = _lib.load()
lib.gm_debug_phases.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
out = (ctypes.c_uint64 * 16)()
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_STAGE_TIMING, max_points=a.points) as c:
    for _ in range(3):
        c.process_frame(xyz)
    lib.gm_debug_phases(out)          # clear
    res = c.process_frame(xyz)
    rc = lib.gm_debug_phases(out)
d = list(out)
names = ["head_search", "operands", "chunk_load_wait", "chunk_features", "pair_loops", "epilogue", "tile_total", "tiles"]
tiles = max(1, d[7])
print(json.dumps({"rc": rc, "points": a.points, "radius": r, "normals_kernel_ms": res["normals_kernel_ms"], "tiles": d[7],
                  "ticks_per_tile": {n: round(d[i] / tiles, 1) for i, n in enumerate(names[:7])},
                  "share": {n: round(d[i] / max(1, d[6]), 4) for i, n in enumerate(names[:6])}}))
