#!/usr/bin/env python3
"""Diagnostic (build with EXTRA="-DGM_MD_DEBUG -DGM_NORMALS_STATS"): error of the distance MFMA over every tested pair."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
r = synth.fixed_k_radius(n)
xyz = synth.tunnel_frame(n, seed=0)
with g.GeometricMapping(neighborRadius=r, max_points=n) as c:
    c.process_frame(xyz)
    out = (ctypes.c_uint32 * 32)()
    lib = _lib.load()
    lib.gm_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    lib.gm_debug_counters(c._ctx, 0, out)
d = list(out)
print(json.dumps({"pairs_beyond_band": d[13], "max_err_over_r2": float(np.array([d[14]], np.uint32).view(np.float32)[0]),
                  "pairs_err_gt_1e-5": d[15], "max_v_over_r_x1000_of_bad": d[16], "max_u_over_r_x1000_of_bad": d[17]}))
