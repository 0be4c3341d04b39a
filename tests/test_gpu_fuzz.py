"""Seeded randomised GPU-vs-oracle sweep over the parameter space (bound, radius, leaf, weighting
factor, cloud shape/size, PointCloud2 layout), including the corners a fixed test forgets: radius
larger than the box, radius tiny against the box (grid clamps), leaf larger than the box, clouds
with duplicates / collinear runs / a single point, slabs of NaN rows.

Integer outputs must match exactly; float outputs to the tolerances of tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

from geometric_mapping_amd import synth

pytestmark = pytest.mark.gpu


def ang(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    s = np.linalg.norm(np.cross(a, b), axis=-1) / np.maximum(np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1), 1e-300)
    return np.arcsin(np.clip(s, 0, 1))


def make_cloud(rng, kind, n):
    if kind == "tunnel":
        return synth.tunnel_frame(n, seed=int(rng.integers(1 << 30)), radius=float(rng.uniform(0.5, 3.0)),
                                  length=float(rng.uniform(2, 14)), sigma=float(rng.uniform(0, 0.03)),
                                  axis=tuple(rng.normal(size=3)), floor_z=(-1.0 if rng.random() < 0.3 else None),
                                  outlier_frac=float(rng.choice([0, 0.01, 0.1])))
    if kind == "blob":
        return rng.normal(0, rng.uniform(0.2, 3.0), (n, 3)).astype(np.float32)
    if kind == "uniform":
        return rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    if kind == "plane":
        return synth.plane_patch(n, seed=int(rng.integers(1 << 30)), normal=tuple(rng.normal(size=3)), offset=float(rng.uniform(-2, 2)), half=4.0,
                                 sigma=float(rng.choice([0, 0.01])))
    if kind == "dupes":       # heavy duplication + a collinear run
        base = rng.uniform(-2, 2, (max(n // 10, 1), 3)).astype(np.float32)
        pts = base[rng.integers(len(base), size=n)]
        t = np.linspace(-1, 1, 50, dtype=np.float32)
        return np.vstack([pts, np.stack([t, t * 0.5, t * 0 + 1], axis=1)]).astype(np.float32)
    raise ValueError(kind)


import os

# (GM_FUZZ_CASES=<n>: a longer one-off sweep)
CASES = list(range(int(os.environ.get("GM_FUZZ_CASES", "28"))))
# Every case runs on the default path; six of them again on the fine-row instantiation of the neighbourhood kernel (y/z
# rows r/2 and r/4 wide: GM_NORMALS_ROWS, read per context) and two replayed from a captured graph (GM_CFG_GRAPH): the
# same oracle, the same tolerances.
VARIANTS = [(c, "default") for c in CASES] + [(0, "rows2"), (5, "rows4"), (7, "rows2"), (10, "rows4"), (15, "rows2"), (21, "rows4"),
                                              (3, "graph"), (16, "graph")]
OBSERVED = {}


@pytest.mark.parametrize("case,variant", VARIANTS)
def test_random_frame_matches_oracle(gm, oc, case, variant, monkeypatch):
    from geometric_mapping_amd import _lib
    if variant.startswith("rows"):
        monkeypatch.setenv("GM_NORMALS_ROWS", variant[4:])
    else:
        monkeypatch.delenv("GM_NORMALS_ROWS", raising=False)
    extra_flags = _lib.GM_CFG_GRAPH if variant == "graph" else int(os.environ.get("GM_FUZZ_FLAGS", "0"))
    rng = np.random.default_rng(1000 + case)
    kind = ["tunnel", "blob", "uniform", "plane", "dupes"][case % 5]
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 1000, 2047, 2048, 2049, 5000, 20000]))
    xyz = make_cloud(rng, kind, n)
    if rng.random() < 0.3 and len(xyz) > 10:   # slab of invalid rows in the middle
        k = len(xyz) // 3
        xyz[k:k + 5] = [np.nan, 0, 0]
        xyz[k + 5:k + 8] = [np.inf, 1, -np.inf]
    bound = float(rng.choice([0.5, 2.0, 5.0, 20.0]))
    radius = float(rng.choice([0.001, 0.05, 0.3, 0.5, 1.5, 12.0]))
    leaf = float(rng.choice([0.02, 0.1, 0.5, 3.0, 50.0]))
    wf = float(rng.choice([0.1, 0.2, 0.3, -0.2]))
    # bound the oracle's cost: huge radius on many points is O(n^2)
    if radius >= 1.5 and len(xyz) > 5000:
        xyz = xyz[:5000]
    step, offs = [(12, (0, 4, 8)), (16, (0, 4, 8)), (32, (8, 12, 16)), (22, (2, 6, 10))][case % 4]
    rows = synth.to_pointcloud2(xyz, point_step=step, offsets=offs, fill=0x5A)
    with gm.GeometricMapping(boxFilterBound=bound, voxelGridLeafSize=leaf, neighborRadius=radius, weightingFactor=wf,
                             flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS | extra_flags) as c:
        res = c.process_frame(c.cloud_from_rows(rows, len(xyz), step, offs))
        if variant == "graph":      # the second frame of a context is the replayed one
            res = c.process_frame(c.cloud_from_rows(rows, len(xyz), step, offs))
        cloud, crows = c.cropped_cloud()
        nrm = c.normals()
        cnt = c.neighbor_counts()
        cen, vcnt = c.voxel_centroids()
    keep = oc.crop_box(xyz, bound)
    o_n, o_cnt = oc.normals(xyz[keep], radius, oc.F64)
    assert res["n_cropped"] == len(keep)
    assert np.array_equal(cnt, o_cnt), (kind, n, bound, radius)          # neighbour sets, every point
    # NaN-normal removal: identical except where the covariance is rank deficient (collinear / coincident
    # neighbours): there PCL itself divides a rounding-noise cross product, so only well-posed points are pinned
    valid = oc.finite_normals(o_n)
    o = oc.process_frame(xyz, bound, radius, leaf, wf, oc.F64)
    if kind != "dupes":
        assert np.array_equal(crows, keep[valid]) and np.array_equal(cloud, xyz[keep][valid])
        assert res["n_valid"] == o["n_valid"] and res["n_voxels"] == o["n_voxels"]
        if len(cen):
            assert np.array_equal(vcnt.sum(), res["n_valid"])
            assert np.abs(cen - o["voxels"]).max() < 2e-6 * max(1.0, bound)
        if len(nrm):
            a = ang(nrm[:, :3], o["normals"][:, :3])
            well = o_cnt[valid] >= 8
            if well.any():
                # (ill-conditioned = near-isotropic covariance: a uniform cloud seen through a radius larger than the box;
                # the reference's own fp32 arithmetic = the oracle's f32_faithful mode)
                o32 = oc.normals(xyz[keep], radius, oc.F32_FAITHFUL)[0][valid]
                fin = well & np.isfinite(o32[:, 0])
                ref_dev = np.quantile(ang(o32[fin, :3], o["normals"][fin, :3]), 0.98) if fin.any() else 0.0
                q98 = float(np.quantile(a[well], 0.98))
                OBSERVED[f"{case}/{variant}"] = {"kind": kind, "n": int(len(xyz)), "radius": radius, "angle_q98": q98, "ref_dev_q98": float(ref_dev)}
                # north_star's 1e-5 (observed over the 28 cases: <= 3e-6 on well-conditioned neighbourhoods), or -- where
                # the neighbourhoods are ill-conditioned -- no further from the f64 value than the reference's own fp32
                # arithmetic is.  (A 3 000-case sweep, GM_FUZZ_CASES=3000: the worst ratio is 0.87, cases 2450 and 2520 -- a
                # round tunnel of 2 048 points seen through r = 12, every covariance that of the whole cloud with two nearly
                # equal small eigenvalues; the bound used to be half the reference's deviation, which those two exceed.)
                assert q98 < max(1e-5, ref_dev), (kind, n, radius, ref_dev)
        M = o["M"]
        if np.abs(M).max() > 0:
            rel = float(np.abs(res["scatter"] - M).max() / np.abs(M).max())
            OBSERVED.setdefault(f"{case}/{variant}", {})["scatter_rel"] = rel
            if rel >= 1e-5:
                # north_star's 1e-5 (observed: <= 1.4e-6 on well-conditioned frames) -- or, where the normals themselves are
                # ill-conditioned (case 865 of a 2 000-case sweep: a round tunnel seen through a radius larger than the
                # cloud, every point's covariance that of the whole tunnel with two nearly equal small eigenvalues: 1.4e-5),
                # what the reference's own fp32 arithmetic (the oracle's f32_faithful mode) deviates from the f64 value
                M32 = oc.process_frame(xyz, bound, radius, leaf, wf, oc.F32_FAITHFUL)["M"]
                rel32 = float(np.abs(M32 - M).max() / np.abs(M).max())
                OBSERVED[f"{case}/{variant}"]["scatter_rel_ref_f32"] = rel32
                assert rel < rel32, (kind, n, radius, rel, rel32)
    else:
        assert set(crows.tolist()) <= set(keep.tolist()) and np.all(np.diff(crows) > 0)
    assert np.isfinite(res["eigenvalues"]).all() and np.isfinite(res["eigenvectors"]).all()
    assert np.all(np.diff(res["eigenvalues"]) >= -1e-3 * max(1.0, abs(res["eigenvalues"][2])))


def test_fuzz_observations_are_written(gm):
    """(runs last in this file) what the cases above observed, for tightening the bounds: gpurun_out/fuzz_observed.json"""
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "fuzz_observed.json"), "w") as f:
        json.dump(OBSERVED, f, indent=1, sort_keys=True)
    assert len(OBSERVED) > 0


def test_validity_on_duplicate_heavy_clouds_is_pinned_where_it_is_defined(gm, oc):
    """Which points lose their normal (removeNaNNormalsFromPointCloud, src/tunnel_processing.cpp:74-85) on a cloud of
    duplicated sites?  A neighbourhood whose covariance is exactly singular has no smallest eigenvector: PCL divides a
    rounding-noise cross product there, and the restatements disagree among themselves (4 000 points on 400 sites:
    every point kept by PCL <= 1.9's float sums, 3 620 by an exact evaluation).  The product's contract:
      * fewer than 3 neighbours (the point included): removed -- exact, every implementation agrees;
      * a neighbourhood on which BOTH the f64 evaluation and the f32-faithful restatement yield a finite normal (a
        well-posed plane fit): kept, and the normal is the oracle's;
      * anything else (all neighbours coincident or collinear up to rounding): unspecified -- kept or removed."""
    from geometric_mapping_amd import _lib
    rng = np.random.default_rng(77)
    base = rng.uniform(-2, 2, (400, 3)).astype(np.float32)
    xyz = base[rng.integers(len(base), size=4000)]
    xyz = np.vstack([xyz, rng.uniform(-2, 2, (300, 3)).astype(np.float32)])      # some ordinary points among them
    radius = 0.3
    with gm.GeometricMapping(boxFilterBound=5.0, neighborRadius=radius, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS) as c:
        res = c.process_frame(xyz)
        cloud, crows = c.cropped_cloud()
        nrm = c.normals()
        cnt = c.neighbor_counts()
    keep = oc.crop_box(xyz, 5.0)
    n64, c64 = oc.normals(xyz[keep], radius, oc.F64)
    n32, _ = oc.normals(xyz[keep], radius, oc.F32_FAITHFUL)
    assert np.array_equal(cnt, c64)
    kept = np.zeros(len(keep), bool)
    kept[np.searchsorted(keep, crows)] = True
    must_drop = c64 < 3
    fin64 = np.zeros(len(keep), bool); fin64[oc.finite_normals(n64)] = True      # (finite_normals returns the kept indices)
    fin32 = np.zeros(len(keep), bool); fin32[oc.finite_normals(n32)] = True
    must_keep = fin64 & fin32
    assert not kept[must_drop].any()
    assert kept[must_keep].all()
    assert res["n_valid"] == int(kept.sum()) and must_keep.sum() > 300
    # where the two restatements agree on the direction (a well-conditioned fit, not rounding noise that happens to be
    # finite in both), the product's normal is theirs
    agree = must_keep.copy()
    agree[must_keep] = ang(n64[must_keep, :3], n32[must_keep, :3]) < 1e-3
    if agree.any():
        pos = np.searchsorted(keep[kept], keep[agree])
        assert np.quantile(ang(nrm[pos, :3], n64[agree, :3]), 0.98) < 1e-3
