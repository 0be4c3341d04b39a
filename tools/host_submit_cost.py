import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import geometric_mapping_amd as g
from geometric_mapping_amd import synth, _lib
n=1_000_000; r=synth.fixed_k_radius(n)
xyz=synth.tunnel_frame(n,seed=0); rows=np.zeros((n,4),np.float32); rows[:,:3]=xyz
t=torch.from_numpy(rows).cuda(); torch.cuda.synchronize()
for flags in (_lib.GM_CFG_DEFAULT, _lib.GM_CFG_DEFAULT|_lib.GM_CFG_RANSAC_CYLINDER):
    with g.GeometricMapping(neighborRadius=r, n_slots=3, max_points=n, flags=flags) as c:
        cl=c.cloud_from_device(t.data_ptr(), n, 16)
        for _ in range(5): c.process_frame(cl)
        ts=[]
        for i in range(30):
            t0=time.perf_counter(); c.submit_frame(i%3, cl); ts.append(time.perf_counter()-t0)
            if i>=2: c.wait_frame((i-2)%3)
        print("flags",flags,"submit host us: median",np.median(ts)*1e6,"min",np.min(ts)*1e6)
