"""GPU tests of the build-defined extensions (RANSAC scoring, labels, moments,
refits, 1-NN, compressed map).  The reference has no counterpart
(getCylinder is a stub, /root/reference src/tunnel_processing.cpp:149-154):
parity here is vs the build's own CPU restatement + analytic truth of the
synthetic generator -- never "vs reference".

Tolerances: inlier counts and labels bit-exact (integer work, same fp32 fma
chain on both sides); hypotheses 1e-6 (double arithmetic, fp contraction may
differ); moments 1e-12 relative (fp64 sums, different order).
"""
import numpy as np
import pytest

from geometric_mapping_amd import synth

pytestmark = pytest.mark.gpu
B, R, LEAF, WF, TAU = 5.0, 0.5, 0.5, 0.2, 0.03


def ang(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    s = np.linalg.norm(np.cross(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))
    return float(np.arcsin(min(1.0, s)))


@pytest.fixture(scope="module")
def ctx(gm):
    c = gm.GeometricMapping()
    yield c
    c.close()


@pytest.fixture(scope="module")
def scene(oc):
    xyz = synth.tunnel_frame(40000, seed=8, floor_z=-1.2, outlier_frac=0.01)
    xyz = xyz[oc.crop_box(xyz, B)]
    nrm, _ = oc.normals(xyz, R, oc.F64)
    keep = oc.finite_normals(nrm)
    return xyz[keep], nrm[keep]


def test_hypotheses_match_oracle(ctx, oc, scene):
    xyz, nrm = scene
    labels = (np.arange(len(xyz)) % 3 == 0).astype(np.uint8)
    for lab, want in ((None, 0), (labels, 0), (labels, 1)):
        hp = ctx.plane_hypotheses(xyz, 42, 512, lab, want)
        op = oc.plane_hypotheses(xyz, 42, 512, lab, want)
        assert np.array_equal(np.isnan(hp), np.isnan(op))
        ok = np.isfinite(op[:, 0])
        assert ok.mean() > 0.95 and np.abs(hp[ok] - op[ok]).max() < 1e-5
        hc = ctx.cylinder_hypotheses(xyz, nrm, 43, 512, lab, want)
        ocy = oc.cylinder_hypotheses(xyz, nrm, 43, 512, lab, want)
        assert np.array_equal(np.isnan(hc), np.isnan(ocy))
        ok = np.isfinite(ocy[:, 0]) & (np.abs(ocy[:, :3]).max(axis=1) < 1e3)
        rel = np.abs(hc[ok] - ocy[ok]) / np.maximum(np.abs(ocy[ok]), 1.0)
        assert np.quantile(rel, 0.99) < 1e-4      # ill-conditioned samples amplify the last double bits


def test_scoring_counts_bit_exact(ctx, oc, scene):
    xyz, nrm = scene
    hp = oc.plane_hypotheses(xyz, 1, 1024)
    hc = oc.cylinder_hypotheses(xyz, nrm, 2, 1024)
    labels = (np.arange(len(xyz)) % 4 == 1).astype(np.uint8)
    for lab, want in ((None, 0), (labels, 1), (labels, 0)):
        assert np.array_equal(ctx.score_planes(xyz, hp, TAU, lab, want), oc.score_planes(xyz, hp, TAU, lab, want))
        assert np.array_equal(ctx.score_cylinders(xyz, hc, TAU, lab, want), oc.score_cylinders(xyz, hc, TAU, lab, want))
    # ragged sizes around the per-block tile, tiny H, empty cloud
    for n in (0, 1, 63, 64, 2047, 2048, 2049):
        a = ctx.score_planes(xyz[:n], hp[:7], TAU)
        assert np.array_equal(a, oc.score_planes(xyz[:n], hp[:7], TAU))
    nanhyp = np.full((3, 4), np.nan, np.float32)
    assert ctx.score_planes(xyz, nanhyp, TAU).tolist() == [0, 0, 0]


def test_scoring_known_answer_perfect_models(ctx):
    pl = synth.plane_patch(5000, seed=3, normal=(0, 0, 1), offset=-1.2, half=3.0)
    assert ctx.score_planes(pl, np.array([[0, 0, 1, 1.2], [0, 0, 1, 1.3]], np.float32), TAU).tolist() == [5000, 0]
    cy = synth.cylinder_frame(5000, seed=5, sigma=0.0)
    h = np.array([[0, 0, 0, 1, 0, 0, 2.0], [0, 0, 0, 1, 0, 0, 2.1], [0, 0, 0.5, 1, 0, 0, 2.0]], np.float32)
    c = ctx.score_cylinders(cy, h, TAU)
    assert c[0] == 5000 and c[1] == 0 and 0 < c[2] < 5000


def test_segment_moments(ctx, oc, scene):
    xyz, nrm = scene
    labels = (np.arange(len(xyz)) % 3).astype(np.uint8)
    for lab, k in ((labels, 1), (labels, 2), (None, 0)):
        a = ctx.segment_moments(xyz, nrm, lab, k)
        b = oc.segment_moments(xyz, nrm, lab, k)
        assert a[0] == b[0]
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-12


def test_nearest(ctx, oc):
    xyz = synth.cylinder_frame(30000, seed=11)
    cen, _, _, _ = oc.voxel_grid(xyz, LEAF)
    a = ctx.nearest(xyz, cen)
    b = oc.nearest(xyz, cen)
    assert np.array_equal(a, b)                      # same fp32 distance, lowest index on ties
    assert ctx.nearest(xyz[:1], cen).tolist() == [0] * len(cen)


def preemptive_best(score_fn, cloud, hyp, labels, want=0):
    """The in-frame RANSAC's staged scoring (csrc/k_ransac.hip launch_score_preemptive), restated with the oracle's
    exhaustive scorer: all hypotheses on every 64th point -> 128 best (count desc, index asc) -> those on every
    16th point -> 8 best -> those on every point -> best full count (lowest index on ties).  H <= 128 starts at the
    second stage, H <= 8 is exhaustive."""
    ids = np.arange(len(hyp))
    for stride, keep in ((64, 128), (16, 8)):
        if len(ids) > keep:
            c = score_fn(cloud[::stride], hyp[ids], TAU, labels[::stride], want)
            ids = ids[np.lexsort((ids, -c))[:keep]]
    c = score_fn(cloud, hyp[ids], TAU, labels, want)
    k = np.lexsort((ids, -c))[0]
    return int(ids[k]), int(c[k])


def test_frame_ransac_plane_then_cylinder(gm, oc):
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(60000, seed=2, floor_z=-1.2, outlier_frac=0.01)
    H, seed = 512, 7
    flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_PLANE | _lib.GM_CFG_RANSAC_CYLINDER | _lib.GM_CFG_NEAREST
    with gm.GeometricMapping(flags=flags, ransac_hypotheses=H, ransac_threshold=TAU, ransac_seed=seed) as c:
        res = c.process_frame(xyz)
        cloud, _ = c.cropped_cloud()
        nrm = c.normals()
        lab = c.labels()
        nn = c.voxel_nearest()
        cen, _ = c.voxel_centroids()
        blob = c.compressed_map()
    # the same sequence on the CPU restatement, fed the GPU's own cloud+normals
    labels = np.zeros(len(cloud), np.uint8)
    hp = oc.plane_hypotheses(cloud, seed, H, labels, 0)
    bp, nbp = preemptive_best(oc.score_planes, cloud, hp, labels)
    assert res["plane_inliers"] == nbp
    assert np.abs(res["plane"] - hp[bp]).max() < 1e-5
    assert nbp >= 0.98 * oc.score_planes(cloud, hp, TAU, labels, 0).max()   # pre-selection loses (almost) nothing
    assert oc.label_plane(cloud, labels, 0, 1, res["plane"], TAU) == res["plane_inliers"]
    hc = oc.cylinder_hypotheses(cloud, nrm, seed + 1, H, labels, 0)
    bc, nbc = preemptive_best(oc.score_cylinders, cloud, hc, labels)
    assert abs(int(res["cylinder_inliers"]) - nbc) <= 2             # hypotheses agree to 1e-6, not bit for bit
    assert oc.label_cylinder(cloud, labels, 0, 2, res["cylinder"], TAU) == res["cylinder_inliers"]
    assert np.array_equal(lab, labels)
    # refits vs the restatement on the same labels
    assert np.abs(res["plane_refit"] - oc.refit_plane(oc.segment_moments(cloud, nrm, labels, 1))).max() < 1e-9 or \
        np.abs(res["plane_refit"] + oc.refit_plane(oc.segment_moments(cloud, nrm, labels, 1))).max() < 1e-9
    assert ang(res["cylinder_axis_refit"], oc.refit_axis(oc.segment_moments(cloud, nrm, labels, 2))) < 1e-9
    # analytic truth of the generator: floor z=-1.2, tunnel R=2 along x
    assert abs(abs(res["plane_refit"][2]) - 1) < 1e-4 and abs(abs(res["plane_refit"][3]) - 1.2) < 3e-3
    assert abs(res["cylinder"][6] - 2.0) < 0.05 and ang(res["cylinder"][3:6], [1, 0, 0]) < 0.05
    assert ang(res["cylinder_axis_refit"], [1, 0, 0]) < 5e-3
    assert res["plane_inliers"] > 5000 and res["cylinder_inliers"] > 20000
    # 1-NN of the voxel centroids
    assert np.array_equal(nn, oc.nearest(cloud, cen))
    # compressed map round trip
    m = gm.decode_compressed_map(blob)
    assert m["n_points"] == res["n_valid"] and len(m["voxels"]) == res["n_voxels"]
    assert np.array_equal(m["voxels"][:, :3], cen)
    assert [p["type"] for p in m["primitives"]] == [1, 2]
    assert m["primitives"][0]["inliers"] == res["plane_inliers"]
    assert np.allclose(m["primitives"][1]["params"], res["cylinder"])
    assert np.allclose(m["center_axis"], res["center_axis"])


def test_frame_ransac_cylinder_only_is_deterministic(gm):
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(50000, seed=4)
    flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_CYLINDER
    with gm.GeometricMapping(flags=flags, ransac_hypotheses=256) as c:
        a = c.process_frame(xyz)
        b = c.process_frame(xyz)
    assert a["cylinder_inliers"] == b["cylinder_inliers"] and np.array_equal(a["cylinder"], b["cylinder"])
    assert a["plane_inliers"] == 0 and np.isnan(a["plane"]).all()
    assert abs(a["cylinder"][6] - 2.0) < 0.05


def test_full_size_properties_10m_plane_and_cylinder(gm, oc):
    """BASELINE configs[2]: 10 M-point frame, plane + cylinder models, one GPU.  Too large for the oracle's normals in
    seconds, so the check is through size-independent properties: exact crop count and order, unit normals, label
    counts that add up, the labelling re-derived by the oracle's O(n) label pass from the reported models, analytic
    truth of the generator, and linearity of the segment moments."""
    from geometric_mapping_amd import _lib
    n = 10_000_000
    xyz = synth.tunnel_frame(n, seed=3, floor_z=-1.2, outlier_frac=0.01)
    flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_PLANE | _lib.GM_CFG_RANSAC_CYLINDER
    with gm.GeometricMapping(neighborRadius=synth.fixed_k_radius(n), flags=flags, ransac_hypotheses=1024,
                             ransac_threshold=TAU, ransac_seed=5, max_points=n) as c:
        res = c.process_frame(xyz)
        cloud, rows = c.cropped_cloud()
        nrm = c.normals()
        lab = c.labels()
        res2 = c.process_frame(xyz)
        lab2 = c.labels()
    inside = np.all(np.abs(xyz) <= 5.0, axis=1)
    assert res["n_cropped"] == int(inside.sum())
    assert np.all(np.diff(rows.astype(np.int64)) > 0) and np.array_equal(xyz[rows], cloud)
    assert res["n_valid"] == len(cloud) == len(nrm) == len(lab)
    assert np.abs(np.linalg.norm(nrm[:, :3].astype(np.float64), axis=1) - 1).max() < 1e-6
    # labels: counts add up, and the oracle's label pass over the same models reproduces them bit for bit
    assert int((lab == 1).sum()) == res["plane_inliers"] and int((lab == 2).sum()) == res["cylinder_inliers"]
    labels = np.zeros(len(cloud), np.uint8)
    assert oc.label_plane(cloud, labels, 0, 1, res["plane"], TAU) == res["plane_inliers"]
    assert oc.label_cylinder(cloud, labels, 0, 2, res["cylinder"], TAU) == res["cylinder_inliers"]
    assert np.array_equal(lab, labels)
    # analytic truth: floor z = -1.2, tunnel R = 2 along x
    assert abs(abs(res["plane_refit"][2]) - 1) < 1e-4 and abs(abs(res["plane_refit"][3]) - 1.2) < 2e-3
    assert abs(res["cylinder"][6] - 2.0) < 0.05 and ang(res["cylinder"][3:6], [1, 0, 0]) < 0.05
    assert ang(res["cylinder_axis_refit"], [1, 0, 0]) < 2e-3 and ang(res["center_axis"], [1, 0, 0]) < 5e-3
    # (floor = 29.5 % of the ring; the cylinder model is the best of 1024 two-point hypotheses whose normals come from
    # 3.5 cm neighbourhoods of sigma = 1 cm points, so it holds about half of the wall, not all of it)
    assert res["plane_inliers"] > 0.25 * len(cloud) and res["cylinder_inliers"] > 0.3 * len(cloud)
    # refits equal the oracle's on the device's own labels (fp64 sums over 8 M points: order differs)
    pr = oc.refit_plane(oc.segment_moments(cloud, nrm, lab, 1))
    assert min(np.abs(res["plane_refit"] - pr).max(), np.abs(res["plane_refit"] + pr).max()) < 1e-8
    assert ang(res["cylinder_axis_refit"], oc.refit_axis(oc.segment_moments(cloud, nrm, lab, 2))) < 1e-8
    # determinism at full size
    assert np.array_equal(lab, lab2) and res2["plane_inliers"] == res["plane_inliers"]
    assert np.array_equal(res2["scatter"], res["scatter"]) and np.array_equal(res2["cylinder"], res["cylinder"])


@pytest.mark.parametrize("H", [7, 8, 9, 127, 128, 129])
def test_frame_ransac_stage_boundaries(gm, oc, H):
    """The staged scoring switches shape at H = 8 (at or below: exhaustive, every hypothesis on every point) and at
    H = 128 (at or below: two stages, every 16th point then every point; above: three, starting on every 64th point;
    include/gm_hip.h, csrc/k_ransac.hip launch_score_preemptive): the plane winner must equal the restated staging
    (preemptive_best above) on the oracle's scorer at, just below and just above each boundary."""
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(50000, seed=4, floor_z=-1.2, outlier_frac=0.01)
    with gm.GeometricMapping(flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_PLANE, ransac_hypotheses=H,
                             ransac_threshold=TAU, ransac_seed=11) as c:
        res = c.process_frame(xyz)
        cloud, _ = c.cropped_cloud()
        lab = c.labels()
    labels = np.zeros(len(cloud), np.uint8)
    hp = oc.plane_hypotheses(cloud, 11, H, labels, 0)
    bp, nbp = preemptive_best(oc.score_planes, cloud, hp, labels)
    assert res["plane_inliers"] == nbp and np.abs(res["plane"] - hp[bp]).max() < 1e-5
    assert oc.label_plane(cloud, labels, 0, 1, res["plane"], TAU) == nbp and np.array_equal(lab, labels)
