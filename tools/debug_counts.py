#!/usr/bin/env python3
"""Debug helper: per-point neighbour counts under two GM_NORMALS_IMPL settings; for mismatching points, how close the
decisive pair is to the radius (relative to r^2) and where the query sits in its tile."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.spatial import cKDTree
from geometric_mapping_amd import synth
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
n = int(sys.argv[1]); r = synth.fixed_k_radius(n)
xyz = synth.tunnel_frame(n, seed=0)
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS, max_points=n) as c:
    c.process_frame(xyz)
    np.save(sys.argv[2], c.neighbor_counts())
''' % ROOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out = {}
with tempfile.TemporaryDirectory() as d:
    for impl in ("valu", "auto"):
        f = os.path.join(d, impl + ".npy")
        subprocess.run([sys.executable, "-c", CHILD, str(n), f], env=dict(os.environ, GM_NORMALS_IMPL=impl), check=True)
        out[impl] = np.load(f)
r = synth.fixed_k_radius(n)
xyz = synth.tunnel_frame(n, seed=0)
xyz = xyz[np.all(np.abs(xyz) <= 5, axis=1)]
bad = np.nonzero(out["valu"] != out["auto"])[0]
print("mismatching points:", len(bad))
tree = cKDTree(xyz.astype(np.float64))
r2 = np.float32(r * r)
for i in bad[:12]:
    nb = np.array(tree.query_ball_point(xyz[i].astype(np.float64), r * 1.001))
    d = xyz[i][None, :] - xyz[nb]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    rel = (r2 - d2) / r2
    k = np.argsort(np.abs(rel))[:3]
    print(int(i), "valu", int(out["valu"][i]), "auto", int(out["auto"][i]), "closest pairs (r2-d2)/r2:", [float(rel[j]) for j in k], "xyz", xyz[i].tolist())
