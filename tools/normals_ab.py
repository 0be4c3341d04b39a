#!/usr/bin/env python3
"""A/B of the two formulations of the neighbourhood kernel (GM_NORMALS_IMPL=valu | mfma, read once per process, hence
child processes): kernel time alone, and how far the per-point outputs are apart (counts must be identical)."""
import argparse, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth
n, r, out, floor = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3], int(sys.argv[4])
r = r or synth.fixed_k_radius(n)
xyz = synth.tunnel_frame(n, seed=0, floor_z=-1.2 if floor else None, outlier_frac=0.01 if floor else 0.0)
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS | _lib.GM_CFG_STAGE_TIMING, max_points=n) as c:
    for _ in range(3): c.process_frame(xyz)
    ms = []
    for _ in range(10):
        res = c.process_frame(xyz); ms.append(res["normals_kernel_ms"])
    np.savez(out, nrm=c.normals(), counts=c.neighbor_counts(), scatter=res["scatter6"], ev=res["eigenvalues"], axis=res["center_axis"])
print(json.dumps({"normals_kernel_ms_min": min(ms), "normals_kernel_ms_median": float(np.median(ms)), "n_valid": res["n_valid"]}))
''' % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--radius", type=float, default=0.0)
ap.add_argument("--floor", type=int, default=0)
ap.add_argument("--impls", default="valu,auto0,auto", help="first one is the reference of the comparison")
a = ap.parse_args()
res = {}
with tempfile.TemporaryDirectory() as d:
    impls = a.impls.split(",")
    for impl in impls:
        env = dict(os.environ, GM_NORMALS_IMPL=impl)
        f = os.path.join(d, impl + ".npz")
        p = subprocess.run([sys.executable, "-c", CHILD, str(a.points), str(a.radius), f, str(a.floor)], env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            print(impl, "FAILED", p.stderr[-2000:]); sys.exit(1)
        res[impl] = (json.loads(p.stdout.strip().splitlines()[-1]), dict(np.load(f)))
    ta, A = res[impls[0]]
    out = {"points": a.points, impls[0]: ta}
    for impl in impls[1:]:
        tb, B = res[impl]
        na, nb = A["nrm"].astype(np.float64), B["nrm"].astype(np.float64)
        same_nan = np.array_equal(np.isnan(na), np.isnan(nb))
        ok = np.isfinite(na[:, 0]) & np.isfinite(nb[:, 0])
        ang = np.arcsin(np.clip(np.linalg.norm(np.cross(na[ok, :3], nb[ok, :3]), axis=1), 0, 1))
        curv = np.abs(na[ok, 3] - nb[ok, 3]) / np.maximum(np.abs(na[ok, 3]), 1e-30)
        out[impl] = dict(tb, speedup=ta["normals_kernel_ms_median"] / tb["normals_kernel_ms_median"],
                         counts_identical=bool(np.array_equal(A["counts"], B["counts"])), nan_pattern_identical=bool(same_nan),
                         counts_diff={"points": int((A["counts"] != B["counts"]).sum()), "sum_abs": int(np.abs(A["counts"].astype(np.int64) - B["counts"]).sum()),
                                      "max_abs": int(np.abs(A["counts"].astype(np.int64) - B["counts"]).max()), "signed_sum": int((B["counts"].astype(np.int64) - A["counts"]).sum())},
                         normal_angle_rad={"max": float(ang.max()), "p999": float(np.quantile(ang, 0.999))},
                         curvature_rel={"max": float(curv.max()), "p999": float(np.quantile(curv, 0.999))},
                         scatter_rel=float(np.abs(A["scatter"] - B["scatter"]).max() / np.abs(A["scatter"]).max()),
                         axis_angle=float(np.arcsin(min(1.0, np.linalg.norm(np.cross(A["axis"], B["axis"]))))))
    print(json.dumps(out))
