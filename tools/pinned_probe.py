#!/usr/bin/env python3
"""Upload-stage time of one frame: pageable rows (staging copy + H2D) against GM_CLOUD_PINNED rows (H2D only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
n = 1_000_000
xyz = synth.tunnel_frame(n, seed=0)
with g.GeometricMapping(neighborRadius=synth.fixed_k_radius(n), flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_STAGE_TIMING, max_points=n) as c:
    buf, as_cloud = c.pinned_rows(n, 12)
    buf[:] = xyz.reshape(-1).view(np.uint8)
    pc = as_cloud()
    for name, cl in (("pageable", xyz), ("pinned", pc), ("pageable", xyz), ("pinned", pc)):
        for _ in range(3):
            c.process_frame(cl)
        t0 = time.perf_counter()
        ups = []
        for _ in range(10):
            ups.append(c.process_frame(cl)["stage_ms"]["upload"])
        dt = (time.perf_counter() - t0) / 10
        print(name, "upload stage %.3f ms, wall per frame %.3f ms" % (float(np.mean(ups)), dt * 1e3))
