#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference cannot run here (ROS/PCL/FLANN/Eigen absent) and ships no fixtures, so
these vectors come from the build's OWN CPU restatement (oracle/gm_oracle.c), cross-
checked against its numpy/scipy twin at generation time -- "parity unpinned".  They
lock the oracle against drift and give the GPU tests inputs+expected outputs that
need neither the oracle's .so nor scipy.  Inputs are stored explicitly (float32), not
just as seeds, so a change of numpy's generator cannot silently move them.

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from geometric_mapping_amd import synth  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402

CASES = {
    # name: (generator kwargs, bound, radius, leaf, wf)
    "cylinder_x_2k": (dict(n=2000, seed=0), 5.0, 0.6, 0.5, 0.2),
    "cylinder_tilted_3k": (dict(n=3000, seed=1, axis=(1, 0.2, -0.1)), 5.0, 0.55, 0.5, 0.2),
    "tunnel_floor_outliers_4k": (dict(n=4000, seed=2, floor_z=-1.2, outlier_frac=0.02), 5.0, 0.5, 0.4, 0.25),
    "small_box_sparse_1k": (dict(n=1000, seed=3, outlier_frac=0.05), 2.5, 0.45, 0.3, 0.1),
    # a spinning 16-beam lidar inside the tunnel (synth.velodyne_tunnel: organised, NaN returns, density falling off with
    # range), the launch file's own values (launch/mapping.launch:7-10)
    "velodyne16_tunnel_7k": (dict(lidar=True, rings=16, az=450, seed=4), 5.0, 0.5, 0.5, 0.2),
}


def main():
    oc.build()
    for name, (gen, b, r, leaf, wf) in CASES.items():
        gen = dict(gen)
        xyz = synth.velodyne_tunnel(**gen)["xyz"] if gen.pop("lidar", False) else synth.tunnel_frame(**gen)
        keep = oc.crop_box(xyz, b)
        c1 = xyz[keep]
        n64, cnt = oc.normals(c1, r, oc.F64)
        valid = oc.finite_normals(n64)
        f64 = oc.process_frame(xyz, b, r, leaf, wf, oc.F64)
        f32 = oc.process_frame(xyz, b, r, leaf, wf, oc.F32_FAITHFUL)
        tw = onp.process_frame(xyz, b, r, leaf, wf)
        # generation-time cross-check against the independent twin
        assert np.array_equal(f64["xyz"], tw["xyz"]) and f64["n_voxels"] == tw["n_voxels"]
        assert np.abs(f64["normals"] - tw["normals"]).max() < 2e-6
        assert np.abs(f64["M"] - tw["M"]).max() / np.abs(tw["M"]).max() < 1e-6
        cen, key, vcnt, _ = oc.voxel_grid(f64["xyz"], leaf, oc.F64)
        out = os.path.join(HERE, name + ".npz")
        np.savez_compressed(
            out, xyz=xyz, params=np.array([b, r, leaf, wf], np.float64),
            crop_rows=keep.astype(np.int32), neighbor_counts=cnt.astype(np.int32), valid_rows=valid.astype(np.int32),
            normals_f64=f64["normals"], normals_f32=f32["normals"],
            voxel_centroids=cen, voxel_keys=key, voxel_counts=vcnt,
            M_f64=f64["M"], evals_f64=f64["evals"], evecs_f64=f64["evecs"],
            M_f32=f32["M"], evals_f32=f32["evals"], evecs_f32=f32["evecs"])
        print(name, os.path.getsize(out) // 1024, "KiB", "n'", len(keep), "n''", len(valid), "V", len(cen))


if __name__ == "__main__":
    main()
