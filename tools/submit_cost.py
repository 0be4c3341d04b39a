#!/usr/bin/env python3
"""Host time of gm_submit_frame (the calling thread enqueues the frame's launches) and of gm_wait_frame on a finished
frame, against the pipelined step: is the submitting thread the bottleneck with frames in flight?  Run on the GPU box."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
n = 1_000_000
r = synth.fixed_k_radius(n)
rows = np.zeros((n, 4), np.float32); rows[:, :3] = synth.tunnel_frame(n, seed=0)
dev = torch.from_numpy(rows).cuda()
flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_CYLINDER
with g.GeometricMapping(neighborRadius=r, flags=flags, n_slots=4, max_points=n) as c:
    cloud = c.cloud_from_device(dev.data_ptr(), n, 16)
    for s in range(4):
        c.submit_frame(s, cloud); c.wait_frame(s)
    ts, tw = [], []
    t_all = time.perf_counter()
    steps = 200
    for i in range(steps):
        s = i % 4
        if i >= 4:
            t0 = time.perf_counter(); c.wait_frame(s); tw.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); c.submit_frame(s, cloud); ts.append(time.perf_counter() - t0)
    for s in range(4):
        c.wait_frame((steps + s) % 4)
    t_all = time.perf_counter() - t_all
print(json.dumps({"ms_per_step": round(t_all / steps * 1e3, 4), "submit_us_median": round(float(np.median(ts)) * 1e6, 1),
                  "submit_us_p90": round(float(np.quantile(ts, 0.9)) * 1e6, 1), "wait_us_median": round(float(np.median(tw)) * 1e6, 1),
                  "host_busy_frac": round(float(np.sum(ts)) / t_all, 3)}))
