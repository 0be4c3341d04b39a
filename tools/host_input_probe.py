#!/usr/bin/env python3
"""Pipelined (3 frames in flight) step time with host input: pageable rows against GM_CLOUD_PINNED rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
n, slots, steps = 1_000_000, int(os.environ.get("SLOTS", "3")), 40
step = int(os.environ.get("STEP", "12"))
flags = _lib.GM_CFG_DEFAULT | (_lib.GM_CFG_RANSAC_CYLINDER if os.environ.get("RANSAC") else 0)
if os.environ.get("TIMING"):
    flags |= _lib.GM_CFG_STAGE_TIMING
if os.environ.get("TORCH"):
    import torch
    keep_t = [torch.zeros(4_000_000, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
frames = [synth.tunnel_frame(n, seed=s) for s in range(4)]
if step == 16:
    frames = [np.ascontiguousarray(np.hstack([f, np.zeros((n, 1), np.float32)])) for f in frames]
with g.GeometricMapping(neighborRadius=synth.fixed_k_radius(n), n_slots=slots, max_points=n, flags=flags,
                        ransac_hypotheses=1024, ransac_threshold=0.03) as c:
    pinned = []
    for f in frames:
        buf, as_cloud = c.pinned_rows(n, step)
        buf[:] = f.reshape(-1).view(np.uint8)
        pinned.append(as_cloud())
    pageable = [c._cloud_from_xyz(f) for f in frames]
    def run(inputs):
        for s in range(slots):
            c.submit_frame(s, inputs[s % 4])
        t0 = time.perf_counter()
        for i in range(steps):
            s = i % slots
            c.wait_frame(s)
            c.submit_frame(s, inputs[(i + slots) % 4])
        for s in range(slots):
            c.wait_frame((steps + s) % slots)
        return (time.perf_counter() - t0) / steps * 1e3
    for name, inp in (("pageable", pageable), ("pinned", pinned), ("pageable", pageable), ("pinned", pinned)):
        run(inp)
        print(name, "step", step, "ransac", bool(os.environ.get("RANSAC")), "timing", bool(os.environ.get("TIMING")), "torch", bool(os.environ.get("TORCH")), "slots", slots, "%.3f ms/step" % run(inp))
