#!/usr/bin/env python3
"""Regenerates tests/golden/markers_*.json from the committed tests/golden/*.npz.

Expected marker fields (every one the reference fills, src/tunnel_processing.cpp:161-205) for the eigen-basis arrows
and the surface-normal arrows of each golden cloud, computed by the oracle-side restatement oracle/markers_np.py from the
oracle's own frame outputs (f32-faithful eigen results; voxel centroids, 1-NN and normals).  "Parity unpinned": the
reference ships no marker fixtures and cannot run here.

  python tests/golden/make_markers.py
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import markers_np as mk  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402


def main():
    oc.build()
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        g = np.load(path)
        name = os.path.basename(path)[:-4]
        xyz = g["xyz"][g["crop_rows"]][g["valid_rows"]]                # the compacted cloud rvizNormals sees
        nrm = g["normals_f64"]                                         # frame output: already compacted, same order
        assert len(nrm) == len(xyz)
        cen = g["voxel_centroids"]
        idx = oc.nearest(xyz, cen)
        normals = mk.rviz_normals(cen, idx, nrm)
        eig = mk.rviz_eigens(g["evals_f32"], g["evecs_f32"])
        out = {"case": name, "eigenBasis": eig, "eigen_inputs": {"vals": [float(v) for v in g["evals_f32"]],
                                                                  "vecs_rowmajor": [[float(v) for v in r] for r in g["evecs_f32"]]},
               "n_normals": len(normals), "nearest_idx_head": [int(i) for i in idx[:12]],
               "normals_head": normals[:12], "normals_tail": normals[-2:]}
        with open(os.path.join(HERE, "markers_" + name + ".json"), "w") as f:
            json.dump(out, f, indent=1)
        print(name, len(normals), "normal markers")


if __name__ == "__main__":
    main()
