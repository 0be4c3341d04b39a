#!/usr/bin/env python3
"""D2H time of the bulky accessors after a 1 M-point frame (what the ROS node pays for /cloudOutput and /normalsOutput)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import synth
n = 1_000_000
xyz = synth.tunnel_frame(n, seed=0)
with g.GeometricMapping(neighborRadius=synth.fixed_k_radius(n), max_points=n) as c:
    res = c.process_frame(xyz)
    m = res["n_valid"]
    buf = np.empty((m, 4), dtype=np.float32)
    cnt = C.c_uint32(0)
    fp = buf.ctypes.data_as(C.POINTER(C.c_float))
    for name, fn in (("gm_get_cropped_xyz", c._L.gm_get_cropped_xyz), ("gm_get_normals", c._L.gm_get_normals)):
        for _ in range(3):
            fn(c._ctx, 0, fp, m, C.byref(cnt))
        t0 = time.perf_counter()
        for _ in range(10):
            fn(c._ctx, 0, fp, m, C.byref(cnt))
        dt = (time.perf_counter() - t0) / 10
        print("%s: %d rows, %.3f ms, %.1f GB/s" % (name, cnt.value, dt * 1e3, m * 16 / dt / 1e9))
