#!/usr/bin/env python3
"""Soak: thousands of frames of varying size through one context (3 slots); device memory must stay flat once the
largest frame has been seen, results of a repeated frame must stay bitwise identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
rng = np.random.default_rng(0)
big = synth.tunnel_frame(1_200_000, seed=1, floor_z=-1.2, outlier_frac=0.01)
flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_PLANE | _lib.GM_CFG_RANSAC_CYLINDER
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
with g.GeometricMapping(neighborRadius=0.12, n_slots=3, flags=flags, ransac_hypotheses=1024, ransac_threshold=0.03) as c:
    ref = c.process_frame(big[:500_000])
    for s in range(3):
        c.process_frame(big)          # grow every slot to the largest frame
        c.submit_frame(s, big); c.wait_frame(s)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.time()
    inflight = []
    for i in range(frames):
        n = int(rng.choice([0, 1, 2, 63, 64, 65, 1000, rng.integers(0, 1_200_000)]))
        s = i % 3
        if len(inflight) == 3:
            c.wait_frame(inflight.pop(0))
        c.submit_frame(s, big[:n])
        inflight.append(s)
        if i % 500 == 499:
            print("frame", i + 1, "free MiB", torch.cuda.mem_get_info()[0] >> 20, "elapsed %.1f s" % (time.time() - t0), flush=True)
    while inflight:
        c.wait_frame(inflight.pop(0))
    again = c.process_frame(big[:500_000])
    free1 = torch.cuda.mem_get_info()[0]
    assert np.array_equal(ref["scatter"], again["scatter"]) and np.array_equal(ref["cylinder"], again["cylinder"], equal_nan=True)
    assert abs(free0 - free1) < (64 << 20), (free0, free1)
print("soak ok: %d frames, device memory delta %d KiB" % (frames, (free0 - free1) >> 10))
