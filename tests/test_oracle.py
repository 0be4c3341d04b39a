"""CPU tests of the oracle itself (no GPU): known-answer cases with analytic
truth, and the C restatement against its independent numpy/scipy twin.

The reference ships no tests or fixtures -> parity unpinned;
these are the only pins the oracle has.
"""
import numpy as np
import pytest

from geometric_mapping_amd import synth
from oracle import oracle_np as onp

B, R, LEAF, WF = 5.0, 0.5, 0.5, 0.2


def axis_angle(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    # via the cross product: arccos loses everything below ~3e-4 rad in float32
    s = np.linalg.norm(np.cross(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))
    return float(np.arcsin(min(1.0, s)))


# ---------------------------------------------------------------- crop

def test_crop_boundary_and_nan(oc):
    b = 5.0
    up = np.nextafter(np.float32(b), np.float32(np.inf))
    pts = np.array([[b, -b, b], [up, 0, 0], [0, -up, 0], [0, 0, 0], [np.nan, 0, 0], [0, np.inf, 0],
                    [4.9999995, 4.9999995, -4.9999995], [0, 0, -up]], dtype=np.float32)
    keep = oc.crop_box(pts, b)
    assert keep.tolist() == [0, 3, 6]
    assert np.array_equal(keep, onp.crop_box(pts, b))


def test_crop_order_idempotent_vs_twin(oc):
    xyz = synth.tunnel_frame(20000, seed=3, outlier_frac=0.02)
    keep = oc.crop_box(xyz, B)
    assert np.all(np.diff(keep) > 0)
    assert np.array_equal(keep, onp.crop_box(xyz, B))
    again = oc.crop_box(xyz[keep], B)
    assert np.array_equal(again, np.arange(len(keep)))
    assert 0.80 < len(keep) / len(xyz) < 0.86     # N'/N ~ 0.83


def test_crop_bound_is_cast_to_float(oc):
    # bound 0.1 (double) -> float 0.100000001490116; a point at float(0.1) is kept
    pts = np.array([[np.float32(0.1), 0, 0], [np.nextafter(np.float32(0.1), np.float32(1)), 0, 0]], dtype=np.float32)
    assert oc.crop_box(pts, 0.1).tolist() == [0]


def test_crop_empty(oc):
    assert len(oc.crop_box(np.zeros((0, 3), np.float32), B)) == 0


# ---------------------------------------------------------------- normals

def test_normals_plane_known_answer(oc):
    nvec = np.array([1.0, 2.0, 2.0]) / 3.0
    xyz = synth.plane_patch(4000, seed=1, normal=nvec, offset=1.5, half=1.5)
    for mode, tol_ang, tol_c in ((oc.F64, 2e-6, 1e-9), (oc.F32_FAITHFUL, 2e-3, 5e-4)):
        nrm, cnt = oc.normals(xyz, 0.4, mode)
        assert np.isfinite(nrm).all()
        ang = np.arcsin(np.clip(np.linalg.norm(np.cross(nrm[:, :3].astype(np.float64), nvec), axis=1), 0, 1))
        assert ang.max() < tol_ang
        assert nrm[:, 3].max() < tol_c
        # flipped towards the origin: (0 - p).n >= 0
        assert ((-xyz.astype(np.float64) * nrm[:, :3]).sum(axis=1) >= 0).all()
        assert cnt.min() >= 3


def test_normals_isolated_points_are_nan(oc):
    xyz = synth.cylinder_frame(3000, seed=2)
    lonely = np.array([[0, 0, 0], [0.0, 0.0, 0.3], [4.0, 4.0, 4.0], [4.0, 4.0, 4.2]], dtype=np.float32)
    cloud = np.vstack([xyz, lonely]).astype(np.float32)
    for mode in (oc.F64, oc.F32_FAITHFUL):
        nrm, cnt = oc.normals(cloud, 0.35, mode)
        assert cnt[-4:].tolist() == [2, 2, 2, 2]           # self + one neighbour
        assert np.isnan(nrm[-4:]).all()
        keep = oc.finite_normals(nrm)
        assert np.array_equal(keep, np.nonzero(np.isfinite(nrm[:, :3]).all(axis=1))[0])


def test_normals_radius_is_strict_and_fp32(oc):
    # d2 == fp32(r*r) exactly -> NOT a neighbour (FLANN RadiusResultSet: dist < radius)
    pts = np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]], dtype=np.float32)
    _, cnt = oc.normals(pts, 0.5, oc.F64)
    assert cnt.tolist() == [1, 1, 1, 1]
    _, cnt = oc.normals(pts, 0.5000001, oc.F64)
    assert cnt.tolist() == [4, 2, 2, 2]


@pytest.mark.parametrize("seed,axis", [(0, (1, 0, 0)), (5, (1, 0.2, -0.1))])
def test_normals_c_vs_numpy_twin(oc, seed, axis):
    xyz = synth.tunnel_frame(6000, seed=seed, axis=axis, outlier_frac=0.01)
    xyz = xyz[oc.crop_box(xyz, B)]
    n_c, cnt_c = oc.normals(xyz, R, oc.F64)
    n_p, cnt_p = onp.normals(xyz, R)
    assert np.array_equal(cnt_c, cnt_p)                    # identical neighbour sets
    assert np.array_equal(np.isnan(n_c), np.isnan(n_p))
    ok = np.isfinite(n_p[:, 0])
    assert np.abs(n_c[ok] - n_p[ok]).max() < 2e-6


def test_normals_threads_agree(oc):
    xyz = synth.cylinder_frame(4000, seed=9)
    a, _ = oc.normals(xyz, R, oc.F64, nthreads=1)
    b, _ = oc.normals(xyz, R, oc.F64, nthreads=4)
    assert np.array_equal(a, b, equal_nan=True)


def test_f32_faithful_is_close_to_f64(oc):
    # the reference's own fp32 noise floor: per-point curvature ~1e-3
    # relative, yet the final axis agrees to ~1e-5 rad
    xyz = synth.cylinder_frame(20000, seed=0)
    a = oc.process_frame(xyz, B, R, LEAF, WF, oc.F64)
    b = oc.process_frame(xyz, B, R, LEAF, WF, oc.F32_FAITHFUL)
    assert a["n_valid"] == b["n_valid"] and a["n_voxels"] == b["n_voxels"]
    assert axis_angle(a["evecs"][:, 0], b["evecs"][:, 0]) < 2e-5
    assert abs(a["evals"][1] - b["evals"][1]) / a["evals"][1] < 1e-5
    assert abs(a["evals"][2] - b["evals"][2]) / a["evals"][2] < 1e-5
    rel = np.abs(a["normals"][:, 3] - b["normals"][:, 3]) / a["normals"][:, 3]
    assert np.median(rel) < 2e-2


# ---------------------------------------------------------------- voxel grid

def test_voxel_single_cell_known_answer(oc):
    pts = np.array([[0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [0.3, 0.3, 0.3]], dtype=np.float32)
    cen, key, cnt, pt = oc.voxel_grid(pts, 0.5)
    assert not pt and len(cen) == 1 and cnt.tolist() == [3] and key.tolist() == [0]
    assert np.allclose(cen[0], 0.2, atol=1e-7)


def test_voxel_negative_coordinates_floor(oc):
    # floor, not truncation: -0.1 and +0.1 are different voxels for leaf 0.5
    pts = np.array([[-0.1, 0, 0], [0.1, 0, 0], [-0.4, 0, 0]], dtype=np.float32)
    cen, key, cnt, _ = oc.voxel_grid(pts, 0.5)
    assert key.tolist() == [0, 1] and cnt.tolist() == [2, 1]
    assert np.allclose(cen[0], [-0.25, 0, 0], atol=1e-7)


def test_voxel_vs_twin_and_properties(oc):
    xyz = synth.tunnel_frame(30000, seed=4)
    xyz = xyz[oc.crop_box(xyz, B)]
    for mode, tol in ((oc.F64, 0.0), (oc.F32_FAITHFUL, 5e-6)):
        cen, key, cnt, pt = oc.voxel_grid(xyz, LEAF, mode)
        c2, k2, n2 = onp.voxel_grid(xyz, LEAF)
        assert not pt
        assert np.array_equal(key, k2) and np.array_equal(cnt, n2)
        assert np.abs(cen - c2).max() <= tol
        assert np.all(np.diff(key) > 0)                     # ascending key order
        assert cnt.sum() == len(xyz)
        # every centroid lies inside its voxel
        inv = np.float32(1) / np.float32(LEAF)
        mn = np.floor(xyz.min(axis=0) * inv)
        div = np.floor(xyz.max(axis=0) * inv) - mn + 1
        ijk = np.floor(cen * inv) - mn
        assert np.array_equal((ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]).astype(np.int32), key)


def test_voxel_leaf_too_small_passthrough(oc):
    pts = np.array([[-5, -5, -5], [5, 5, 5], [0, 0, 0]], dtype=np.float32)
    cen, key, cnt, pt = oc.voxel_grid(pts, 0.005)          # (2001)^3 > INT_MAX
    assert pt and np.array_equal(cen, pts)


# ---------------------------------------------------------------- local frame

def _cyl_normals(n, axis, seed=0):
    rng = np.random.default_rng(seed)
    a, u, v = synth._basis(axis)
    th = rng.uniform(0, 2 * np.pi, n)
    nr = np.cos(th)[:, None] * u + np.sin(th)[:, None] * v
    return a, np.concatenate([nr, np.zeros((n, 1))], axis=1).astype(np.float32)


@pytest.mark.parametrize("axis", [(1, 0, 0), (1, 0.2, -0.1), (0, 0, 1)])
def test_local_frame_perfect_cylinder(oc, axis):
    a, nrm = _cyl_normals(50000, axis)
    w2 = np.exp((0.001 / WF) ** 2) ** 2
    for mode, tol in ((oc.F64, 1e-6), (oc.F32_FAITHFUL, 3e-4)):
        ev, V, M = oc.local_frame(nrm, WF, mode)
        assert axis_angle(V[:, 0], a) < tol
        assert abs(ev[0]) < 1e-3 * ev[2]
        assert abs((ev[1] + ev[2]) - len(nrm) * w2) / (len(nrm) * w2) < 1e-4
        assert np.allclose(V.T @ V, np.eye(3), atol=5e-6)
        assert ev[0] <= ev[1] <= ev[2]


def test_local_frame_plane_known_answer(oc):
    n = 1000
    nrm = np.tile(np.array([[0, 0, 1, 0]], dtype=np.float32), (n, 1))
    ev, V, M = oc.local_frame(nrm, WF, oc.F64)
    w2 = np.exp((0.001 / WF) ** 2) ** 2
    assert abs(ev[2] - n * w2) < 1e-6 * n and abs(ev[0]) < 1e-9 and abs(ev[1]) < 1e-9
    assert abs(abs(V[2, 2]) - 1) < 1e-7


def test_local_frame_weight_formula(oc):
    # precedence: (c + (.001/wf))^2, not ((c + .001)/wf)^2  (tunnel_processing.cpp:106)
    nrm = np.array([[1, 0, 0, 0.3]], dtype=np.float32)
    ev, V, M = oc.local_frame(nrm, 0.2, oc.F64)
    w = np.exp((np.float64(np.float32(0.3)) + 0.001 / 0.2) ** 2)
    assert abs(M[0, 0] - w * w) < 1e-12
    ev, V, M = oc.local_frame(nrm, 0.2, oc.F32_FAITHFUL)
    wf32 = np.float32(w)
    assert abs(M[0, 0] - float(np.float32(wf32 * np.float32(1)) ** 2)) < 1e-6


def test_local_frame_vs_twin(oc):
    xyz = synth.cylinder_frame(8000, seed=7, axis=(1, 0.2, -0.1))
    nrm, _ = oc.normals(xyz, R, oc.F64)
    ev, V, M = oc.local_frame(nrm, WF, oc.F64)
    e2, V2, M2 = onp.local_frame(nrm, WF)
    assert np.abs(M - M2).max() / np.abs(M2).max() < 1e-13
    assert np.abs(ev[1:] - e2[1:]).max() / e2[2] < 1e-6
    assert axis_angle(V[:, 0], V2[:, 0]) < 1e-6


def test_local_frame_empty_and_tiny(oc):
    ev, V, M = oc.local_frame(np.zeros((0, 4), np.float32), WF, oc.F64)
    assert np.all(ev == 0) and np.all(M == 0)


def test_eig3_against_numpy(oc):
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.normal(size=(3, 3)); A = A + A.T
        w, V = oc.eig3(A)
        w2, _ = np.linalg.eigh(A)
        assert np.allclose(w, w2, atol=1e-12)
        assert np.allclose(A @ V, V * w, atol=1e-11)
    w, V = oc.eig3(np.diag([3.0, 1.0, 2.0]))
    assert np.allclose(w, [1, 2, 3])


# ---------------------------------------------------------------- 1-NN + whole frame

def test_nearest_vs_twin(oc):
    xyz = synth.cylinder_frame(5000, seed=11)
    cen, _, _, _ = oc.voxel_grid(xyz, LEAF)
    a = oc.nearest(xyz, cen)
    b = onp.nearest(xyz, cen)
    da = np.linalg.norm(xyz[a] - cen, axis=1)
    db = np.linalg.norm(xyz[b] - cen, axis=1)
    assert np.allclose(da, db, rtol=1e-6, atol=1e-7)
    assert (a == b).mean() > 0.98      # fp32-vs-double near-ties pick different points


def test_whole_frame_c_vs_twin(oc):
    xyz = synth.tunnel_frame(8000, seed=1, outlier_frac=0.01)
    a = oc.process_frame(xyz, B, R, LEAF, WF, oc.F64)
    b = onp.process_frame(xyz, B, R, LEAF, WF)
    for k in ("n_in", "n_cropped", "n_valid", "n_voxels"):
        assert a[k] == b[k]
    assert a["n_valid"] < a["n_cropped"]                    # outliers produce NaN normals
    assert np.array_equal(a["xyz"], b["xyz"])
    assert np.abs(a["normals"] - b["normals"]).max() < 2e-6
    assert np.abs(a["voxels"] - b["voxels"]).max() == 0.0
    assert np.abs(a["M"] - b["M"]).max() / np.abs(b["M"]).max() < 1e-6
    assert axis_angle(a["evecs"][:, 0], b["evecs"][:, 0]) < 1e-6
    # known answer: tunnel axis is x
    assert axis_angle(a["evecs"][:, 0], [1, 0, 0]) < 5e-3


# ---------------------------------------------------------------- extensions (build-defined)

def test_plane_hypotheses_and_scores_on_perfect_plane(oc):
    nvec = np.array([0.0, 0.0, 1.0])
    xyz = synth.plane_patch(5000, seed=3, normal=nvec, offset=-1.2, half=3.0)
    hyp = oc.plane_hypotheses(xyz, seed=42, H=64)
    ok = np.isfinite(hyp[:, 0])
    assert ok.mean() > 0.9
    assert np.allclose(np.abs(hyp[ok, :3] @ nvec), 1.0, atol=1e-4)
    assert np.allclose(np.abs(hyp[ok, 3]), 1.2, atol=1e-4)
    cnt = oc.score_planes(xyz, hyp, 0.03)
    assert (cnt[ok] == len(xyz)).all() and (cnt[~ok] == 0).all()
    # determinism of the seeded sampler
    assert np.array_equal(hyp, oc.plane_hypotheses(xyz, seed=42, H=64), equal_nan=True)
    assert not np.array_equal(hyp, oc.plane_hypotheses(xyz, seed=43, H=64), equal_nan=True)


def test_cylinder_hypotheses_on_perfect_cylinder(oc):
    axis = (1, 0.2, -0.1)
    xyz = synth.cylinder_frame(4000, seed=5, sigma=0.0, axis=axis).astype(np.float32)
    a, u, v = synth._basis(axis)
    p = xyz.astype(np.float64)
    radial = p - (p @ a)[:, None] * a
    nrm = np.concatenate([-radial / np.linalg.norm(radial, axis=1, keepdims=True), np.zeros((len(p), 1))], axis=1).astype(np.float32)
    hyp = oc.cylinder_hypotheses(xyz, nrm, seed=7, H=128)
    ok = np.isfinite(hyp[:, 0])
    good = ok & (np.abs(hyp[:, 6] - 2.0) < 1e-2)
    assert good.sum() > 100
    assert np.abs(np.abs(hyp[good, 3:6] @ a) - 1).max() < 1e-3
    cnt = oc.score_cylinders(xyz, hyp, 0.03)
    assert (cnt[good] == len(xyz)).all()


def test_sequential_labels_and_refit(oc):
    xyz = synth.tunnel_frame(20000, seed=8, floor_z=-1.2, outlier_frac=0.01)
    xyz = xyz[oc.crop_box(xyz, B)]
    nrm, _ = oc.normals(xyz, R, oc.F64)
    keep = oc.finite_normals(nrm)
    xyz, nrm = xyz[keep], nrm[keep]
    labels = np.zeros(len(xyz), np.uint8)
    hp = oc.plane_hypotheses(xyz, 1, 256)
    cp = oc.score_planes(xyz, hp, 0.03, labels, 0)
    best = hp[np.argmax(cp)]
    n_pl = oc.label_plane(xyz, labels, 0, 1, best, 0.03)
    assert n_pl == cp.max() and (labels == 1).sum() == n_pl
    assert abs(abs(best[2]) - 1) < 1e-2 and abs(abs(best[3]) - 1.2) < 5e-2
    hc = oc.cylinder_hypotheses(xyz, nrm, 2, 256, labels, 0)   # samples only from unassigned points
    cc = oc.score_cylinders(xyz, hc, 0.03, labels, 0)
    bc = hc[np.argmax(cc)]
    n_cy = oc.label_cylinder(xyz, labels, 0, 2, bc, 0.03)
    assert n_cy == cc.max()
    assert abs(bc[6] - 2.0) < 0.05 and axis_angle(bc[3:6], [1, 0, 0]) < 0.05
    mom = oc.segment_moments(xyz, nrm, labels, 1)
    pl = oc.refit_plane(mom)
    assert mom[0] == n_pl and abs(abs(pl[2]) - 1) < 1e-4 and abs(abs(pl[3]) - 1.2) < 2e-3
    mom = oc.segment_moments(xyz, nrm, labels, 2)
    ax = oc.refit_axis(mom)
    assert axis_angle(ax, [1, 0, 0]) < 5e-3


# ------------------------------------------------------------------ fp32-faithful modes against their numpy twin

@pytest.mark.parametrize("shifted", [False, True])
def test_f32_faithful_normals_match_the_numpy_float32_twin(oc, shifted):
    """GMO_F32_FAITHFUL (PCL <= 1.9 un-shifted covariance) and GMO_F32_SHIFTED (PCL >= 1.10) against the separately
    written numpy float32-scalar twin: neighbour counts exact, every component to 2e-6 where the neighbourhood is not
    rank deficient (numpy's float32 trigonometry may differ from glibc's by an ulp), curvature to 1e-5 absolute."""
    from oracle import oracle_np as onp
    xyz = synth.tunnel_frame(900, seed=7, outlier_frac=0.02)
    xyz = xyz[np.all(np.abs(xyz) <= 5, axis=1)]
    r = 0.9
    mode = oc.F32_SHIFTED if shifted else oc.F32_FAITHFUL
    cn, cc = oc.normals(xyz, r, mode)
    tn, tc, _ = onp.normals_f32(xyz, r, shifted=shifted)
    assert np.array_equal(cc, tc)
    assert np.array_equal(np.isnan(cn[:, 0]), np.isnan(tn[:, 0]))
    ok = np.isfinite(cn[:, 0]) & (cc >= 6)
    assert ok.sum() > 500
    assert np.abs(cn[ok, :3] - tn[ok, :3]).max() < 2e-6
    assert np.abs(cn[ok, 3] - tn[ok, 3]).max() < 1e-5


def test_shifted_and_unshifted_covariance_agree_with_f64_to_their_own_precision(oc):
    """PCL-version ambiguity (package.xml pins none): the >= 1.10 shifted sums are ~100x closer to the f64 value than the
    <= 1.9 un-shifted ones on a cloud a few metres from the origin; both leave the frame-level fit within 1e-5."""
    xyz = synth.tunnel_frame(20000, seed=8)
    n64, _ = oc.normals(xyz, 0.5, oc.F64, nthreads=8)
    n19, _ = oc.normals(xyz, 0.5, oc.F32_FAITHFUL, nthreads=8)
    n110, _ = oc.normals(xyz, 0.5, oc.F32_SHIFTED, nthreads=8)
    ok = np.isfinite(n64[:, 0]) & np.isfinite(n19[:, 0]) & np.isfinite(n110[:, 0])
    e19 = np.abs(n19[ok, 3] - n64[ok, 3]) / n64[ok, 3]
    e110 = np.abs(n110[ok, 3] - n64[ok, 3]) / n64[ok, 3]
    assert np.median(e110) < 0.1 * np.median(e19)
    f64 = oc.process_frame(xyz, 5.0, 0.5, 0.5, 0.2, oc.F64, nthreads=8)
    for mode in (oc.F32_FAITHFUL, oc.F32_SHIFTED):
        f = oc.process_frame(xyz, 5.0, 0.5, 0.5, 0.2, mode, nthreads=8)
        a, b = f["evecs"][:, 0].astype(np.float64), f64["evecs"][:, 0].astype(np.float64)
        assert np.linalg.norm(np.cross(a, b)) < 1e-5
        assert abs(f["evals"][2] - f64["evals"][2]) / f64["evals"][2] < 1e-5


def test_f32_faithful_local_frame_sums_match_the_numpy_twin_bit_for_bit(oc):
    from oracle import oracle_np as onp
    xyz = synth.cylinder_frame(3000, seed=9)
    nrm, _ = oc.normals(xyz, 0.6, oc.F32_FAITHFUL)
    nrm = nrm[oc.finite_normals(nrm)]
    ev, V, M_c = oc.local_frame(nrm, 0.2, oc.F32_FAITHFUL)
    M_t = onp.local_frame_f32(nrm, 0.2)
    assert np.array_equal(M_c.astype(np.float32), M_t)          # sequential fp32 sums: same order, same roundings
    w = np.linalg.eigvalsh(M_t.astype(np.float64))
    assert np.allclose(ev, w, rtol=2e-6)
