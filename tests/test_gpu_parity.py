"""GPU parity tests: the HIP path (through the C ABI, ctypes) against the CPU
oracle on the same seeded inputs.  Run on the MI355X box with `-m gpu`.

Tolerances (stated once, used below):
  * crop, NaN removal, neighbour counts, voxel keys/counts: bit-exact (integer / index work);
  * voxel centroids: 1 float ulp-ish (2e-6 absolute on |x|<=5) vs the f64 oracle;
  * per-point normal direction: 1e-5 rad; per-point curvature: 1e-4 relative
    (fp32 neighbour accumulation; the reference's own fp32 formula is ~3e-3 off the
    f64 value, SURVEY.md par. 7);
  * frame outputs (north_star's 1e-5): centre axis 1e-5 rad, lambda1/lambda2 1e-5 relative,
    lambda0 1e-5 of lambda2 (it is a near-null eigenvalue: absolute, not relative).
"""
import numpy as np
import pytest

from geometric_mapping_amd import synth

pytestmark = pytest.mark.gpu

B, R, LEAF, WF = 5.0, 0.5, 0.5, 0.2


def ang(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    s = np.linalg.norm(np.cross(a, b), axis=-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))
    return np.arcsin(np.clip(s, 0, 1))


@pytest.fixture(scope="module")
def ctx(gm):
    from geometric_mapping_amd import _lib
    c = gm.GeometricMapping(boxFilterBound=B, voxelGridLeafSize=LEAF, neighborRadius=R, weightingFactor=WF,
                            flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS | _lib.GM_CFG_STAGE_TIMING)
    yield c
    c.close()


# ------------------------------------------------------------------ stages

def test_chop_cloud_bit_exact(ctx, oc):
    xyz = synth.tunnel_frame(50000, seed=3, outlier_frac=0.02)
    xyz[17] = [np.nan, 0, 0]
    xyz[18] = [0, np.inf, 0]
    xyz[19] = [5.0, -5.0, 5.0]
    xyz[20] = [np.nextafter(np.float32(5), np.float32(9)), 0, 0]
    out, rows = ctx.chopCloud(B, xyz)
    keep = oc.crop_box(xyz, B)
    assert np.array_equal(rows, keep)
    assert np.array_equal(out, xyz[keep])
    assert 19 in rows and 20 not in rows and 17 not in rows and 18 not in rows


@pytest.mark.parametrize("n", [1, 4095, 4096, 4097, 8192, 4096 * 513 + 5, 4096 * 1300 + 77])
def test_chop_cloud_order_at_compaction_tile_boundaries(ctx, n):
    """The crop is one launch of a chained scan over 4096-point tiles (csrc/gm_compact.hpp): sizes at and around tile
    boundaries and more tiles than one look-back trip spans (512) must all give CropBox's order-preserving result --
    checked against numpy, every row."""
    rng = np.random.default_rng(n)
    xyz = rng.uniform(-6.0, 6.0, size=(n, 3)).astype(np.float32)   # ~58 % inside the +-5 box
    if n > 100:
        xyz[rng.integers(0, n, 50)] = np.nan
        xyz[: min(n, 9000)] = 9.0          # a run of whole tiles without a survivor
    keep = np.flatnonzero(np.all(np.isfinite(xyz), axis=1) & np.all(np.abs(xyz) <= B, axis=1))
    out, rows = ctx.chopCloud(B, xyz)
    assert np.array_equal(rows, keep)
    assert np.array_equal(out, xyz[keep])


def test_chop_cloud_empty_and_all_outside(ctx):
    out, rows = ctx.chopCloud(B, np.zeros((0, 3), np.float32))
    assert out.shape == (0, 3)
    out, rows = ctx.chopCloud(B, np.full((1000, 3), 9.0, np.float32))
    assert out.shape == (0, 3)


@pytest.mark.parametrize("step,offs", [(12, (0, 4, 8)), (16, (0, 4, 8)), (32, (4, 8, 12)), (22, (0, 4, 8)), (22, (1, 9, 17))])
def test_chop_cloud_pointcloud2_layouts(ctx, oc, step, offs):
    xyz = synth.tunnel_frame(20000, seed=5)
    rows = synth.to_pointcloud2(xyz, point_step=step, offsets=offs, fill=0xAB)
    out, kept = ctx.chopCloud(B, ctx.cloud_from_rows(rows, len(xyz), step, offs))
    keep = oc.crop_box(xyz, B)
    assert np.array_equal(kept, keep) and np.array_equal(out, xyz[keep])


def test_chop_cloud_bigendian(ctx, oc):
    xyz = synth.tunnel_frame(5000, seed=6)
    be = xyz.astype(">f4").view(np.uint8).reshape(-1)
    out, kept = ctx.chopCloud(B, ctx.cloud_from_rows(be, len(xyz), 12, (0, 4, 8), bigendian=True))
    keep = oc.crop_box(xyz, B)
    assert np.array_equal(kept, keep) and np.array_equal(out, xyz[keep])


@pytest.mark.parametrize("n,seed,axis,radius", [(20000, 0, (1, 0, 0), 0.5), (30000, 5, (1, 0.2, -0.1), 0.4),
                                                (8000, 2, (0, 0, 1), 0.7)])
def test_get_normals_vs_oracle(ctx, oc, n, seed, axis, radius):
    xyz = synth.tunnel_frame(n, seed=seed, axis=axis, outlier_frac=0.01)
    xyz = xyz[oc.crop_box(xyz, B)]
    nrm, cloud, rows = ctx.getNormals(radius, xyz)
    o_n, o_cnt = oc.normals(xyz, radius, oc.F64)
    keep = oc.finite_normals(o_n)
    assert np.array_equal(rows, keep)                       # same NaN-normal removal
    assert np.array_equal(cloud, xyz[keep])
    cnt = ctx.neighbor_counts()                              # kept by GM_CFG_KEEP_COUNTS: pre-compaction order
    assert np.array_equal(cnt, o_cnt)                        # identical neighbour SETS sizes, every point
    o = o_n[keep]
    a = ang(nrm[:, :3], o[:, :3])
    # eigen-gap weighted: the direction is only defined to ~eps/gap; on this data the gap is healthy
    assert np.quantile(a, 0.999) < 1e-5 and a.max() < 1e-4
    assert (np.sign((nrm[:, :3] * o[:, :3]).sum(axis=1)) > 0).mean() > 0.999   # same flip
    rel = np.abs(nrm[:, 3] - o[:, 3]) / np.maximum(o[:, 3], 1e-12)
    assert np.quantile(rel, 0.999) < 1e-4


def test_get_normals_isolated_and_degenerate(ctx, oc):
    xyz = synth.cylinder_frame(3000, seed=2)
    lonely = np.array([[0, 0, 0], [0.0, 0.0, 0.3], [4.0, 4.0, 4.0], [4.0, 4.0, 4.2]], dtype=np.float32)
    same = np.tile(np.array([[3.0, -3.0, 3.0]], np.float32), (5, 1))      # zero covariance -> NaN (PCL divides 0/0)
    cloud = np.vstack([xyz, lonely, same]).astype(np.float32)
    nrm, out, rows = ctx.getNormals(0.35, cloud)
    o_n, _ = oc.normals(cloud, 0.35, oc.F64)
    keep = oc.finite_normals(o_n)
    assert np.array_equal(rows, keep)
    assert rows.max() < len(xyz)


def test_get_normals_radius_strict(ctx):
    pts = np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]], dtype=np.float32)
    ctx.getNormals(0.5, pts)
    assert ctx.neighbor_counts().tolist() == [1, 1, 1, 1]
    ctx.getNormals(0.5000001, pts)
    assert ctx.neighbor_counts().tolist() == [4, 2, 2, 2]


def test_get_normals_radius_extremes_vs_oracle(ctx, oc, gm):
    """The neighbour predicate d2 < (float)(r*r) must stay exact at every radius scale (the kernel evaluates it as a
    clamped fma with a radius-dependent power-of-two scale): tiny radii with coincident / nearly coincident points,
    huge radii (every point a neighbour), and points sitting exactly ON the radius."""
    rng = np.random.default_rng(11)
    base = rng.uniform(-1.0, 1.0, size=(300, 3)).astype(np.float32)
    dup = np.repeat(base[:40], 3, axis=0)                                  # coincident triples
    near = (base[:60] + np.float32(1e-6) * rng.standard_normal((60, 3))).astype(np.float32)
    cloud = np.vstack([base, dup, near]).astype(np.float32)
    for radius in (1e-12, 1e-7, 3e-6, 0.05, 0.7, 50.0, 1e6, 1e12):
        ctx.getNormals(radius, cloud)
        _, o_cnt = oc.normals(cloud, radius, oc.F64)
        assert np.array_equal(ctx.neighbor_counts(), o_cnt), radius
    # exact ties: lattice points at distance exactly r (d2 == r2 in fp32) are NOT neighbours, at several scales
    for scale in (2.0 ** -20, 2.0 ** -3, 1.0, 2.0 ** 10):
        pts = (np.array([[0, 0, 0], [3, 4, 0], [0, 3, 4], [4, 0, 3], [1, 1, 1]], dtype=np.float64) * scale).astype(np.float32)
        ctx.getNormals(5.0 * scale, pts)
        _, o_cnt = oc.normals(pts, 5.0 * scale, oc.F64)
        assert np.array_equal(ctx.neighbor_counts(), o_cnt), scale
        assert ctx.neighbor_counts()[0] == 2                               # itself and (1,1,1)*scale only
    with pytest.raises(gm.GmError):
        ctx.getNormals(1e-20, cloud)                                       # outside the admitted radius range


def test_get_normals_plane_known_answer(ctx):
    nvec = np.array([1.0, 2.0, 2.0]) / 3.0
    xyz = synth.plane_patch(4000, seed=1, normal=nvec, offset=1.5, half=1.5)
    nrm, cloud, rows = ctx.getNormals(0.4, xyz)
    assert len(nrm) == len(xyz)
    assert ang(nrm[:, :3], nvec).max() < 2e-6
    # exact value 0; the kernel adds <= 64-term fp32 partial sums of offsets into fp64 totals, so lambda0/trace is
    # good to about two fp32 epsilons (1.19e-7 each), whatever order the sort leaves the candidates in
    assert nrm[:, 3].max() < 2.5e-7
    assert ((-xyz.astype(np.float64) * nrm[:, :3]).sum(axis=1) >= 0).all()


def test_voxel_grid_vs_oracle(ctx, oc):
    xyz = synth.tunnel_frame(40000, seed=4)
    xyz = xyz[oc.crop_box(xyz, B)]
    for leaf in (0.5, 0.1, 0.27):
        cen, cnt, pt = ctx.voxelGrid(leaf, xyz)
        o_c, o_k, o_n, o_pt = oc.voxel_grid(xyz, leaf, oc.F64)
        assert pt == o_pt and len(cen) == len(o_c)
        assert np.array_equal(cnt, o_n)                      # same voxels, same order, same membership
        assert np.abs(cen - o_c).max() < 2e-6


def test_voxel_grid_known_answers(ctx):
    pts = np.array([[0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [0.3, 0.3, 0.3]], dtype=np.float32)
    cen, cnt, pt = ctx.voxelGrid(0.5, pts)
    assert cnt.tolist() == [3] and np.allclose(cen[0], 0.2, atol=1e-7)
    pts = np.array([[-0.1, 0, 0], [0.1, 0, 0], [-0.4, 0, 0]], dtype=np.float32)
    cen, cnt, pt = ctx.voxelGrid(0.5, pts)
    assert cnt.tolist() == [2, 1] and np.allclose(cen[0], [-0.25, 0, 0], atol=1e-7)
    cen, cnt, pt = ctx.voxelGrid(0.5, np.zeros((0, 3), np.float32))
    assert len(cen) == 0


def test_voxel_grid_leaf_too_small_passthrough(ctx, oc):
    pts = np.array([[-5, -5, -5], [5, 5, 5], [0, 0, 0], [1, 2, 3]], dtype=np.float32)
    cen, cnt, pt = ctx.voxelGrid(0.005, pts)
    assert pt and np.array_equal(cen, pts)


def test_local_frame_vs_oracle(ctx, oc):
    xyz = synth.cylinder_frame(30000, seed=7, axis=(1, 0.2, -0.1))
    nrm, _ = oc.normals(xyz, R, oc.F64)
    nrm = nrm[oc.finite_normals(nrm)]
    ev, V, M = ctx.getLocalFrame(len(nrm), WF, nrm)
    o_ev, o_V, o_M = oc.local_frame(nrm, WF, oc.F64)
    assert np.abs(M - o_M).max() / np.abs(o_M).max() < 5e-7   # fp32 store of w (reference semantics) vs double w
    assert ang(V[:, 0], o_V[:, 0]) < 1e-5
    assert abs(ev[1] - o_ev[1]) / o_ev[1] < 1e-5 and abs(ev[2] - o_ev[2]) / o_ev[2] < 1e-5
    assert abs(ev[0] - o_ev[0]) < 1e-5 * o_ev[2]
    assert np.allclose(V.T @ V, np.eye(3), atol=1e-6)
    # f32-faithful restatement of the reference's own arithmetic lands in the same place
    f_ev, f_V, _ = oc.local_frame(nrm, WF, oc.F32_FAITHFUL)
    assert ang(V[:, 0], f_V[:, 0]) < 2e-5


def test_local_frame_known_answers(ctx):
    n = 1000
    nrm = np.tile(np.array([[0, 0, 1, 0]], dtype=np.float32), (n, 1))
    ev, V, M = ctx.getLocalFrame(n, WF, nrm)
    w = np.float32(np.exp((0.001 / WF) ** 2))
    assert abs(ev[2] - n * float(w) ** 2) < 1e-4 and abs(ev[0]) < 1e-6 and abs(ev[1]) < 1e-6
    ev, V, M = ctx.getLocalFrame(0, WF, np.zeros((0, 4), np.float32))
    assert np.all(ev == 0) and np.all(M == 0)
    # cloudSize smaller than the cloud: only the first cloudSize rows count (tunnel_processing.cpp:104)
    nrm2 = np.vstack([nrm, np.tile(np.array([[1, 0, 0, 0]], np.float32), (10, 1))])
    ev2, _, M2 = ctx.getLocalFrame(n, WF, nrm2)
    assert abs(M2[0, 0]) < 1e-12


# ------------------------------------------------------------------ whole frame

@pytest.mark.parametrize("n,seed,axis,kw", [
    (50000, 0, (1, 0, 0), {}),                                           # BASELINE configs[0] shape
    (50000, 1, (1, 0.2, -0.1), dict(outlier_frac=0.01)),
    (60000, 2, (1, 0, 0), dict(floor_z=-1.2, outlier_frac=0.01)),        # plane + cylinder + outliers
])
def test_process_frame_vs_oracle(ctx, oc, n, seed, axis, kw):
    xyz = synth.tunnel_frame(n, seed=seed, axis=axis, **kw)
    res = ctx.process_frame(xyz)
    o = oc.process_frame(xyz, B, R, LEAF, WF, oc.F64)
    for k in ("n_in", "n_cropped", "n_valid", "n_voxels"):
        assert res[k] == o[k], k
    cloud, rows = ctx.cropped_cloud()
    assert np.array_equal(cloud, o["xyz"])                   # /choppedCloud: same points, same order
    assert np.array_equal(xyz[rows], cloud)
    nrm = ctx.normals()
    a = ang(nrm[:, :3], o["normals"][:, :3])
    assert np.quantile(a, 0.999) < 1e-5
    rel = np.abs(nrm[:, 3] - o["normals"][:, 3]) / np.maximum(o["normals"][:, 3], 1e-12)
    assert np.quantile(rel, 0.999) < 1e-4
    cen, cnt = ctx.voxel_centroids()
    assert np.abs(cen - o["voxels"]).max() < 2e-6
    # north_star bar: 1e-5
    assert ang(res["eigenvectors"][:, 0], o["evecs"][:, 0]) < 1e-5
    assert ang(res["center_axis"], o["evecs"][:, 0]) < 1e-5
    l, lo = res["eigenvalues"].astype(np.float64), o["evals"].astype(np.float64)
    assert abs(l[1] - lo[1]) / lo[1] < 1e-5 and abs(l[2] - lo[2]) / lo[2] < 1e-5
    assert abs(l[0] - lo[0]) < 1e-5 * lo[2]
    assert np.abs(res["scatter"] - o["M"]).max() / np.abs(o["M"]).max() < 1e-5
    # The f32-faithful restatements of the reference (PCL <= 1.9 un-shifted and PCL >= 1.10 shifted covariance) are the
    # noisy side here: their own distance from the f64 value bounds how close anything can be to them.  The GPU must
    # be at least as close to each of them as that mode is to f64, plus the 1e-5 bar (observed deltas: DESIGN.md par. 5,
    # tests/test_gpu_parity.py::test_deltas_against_the_f32_faithful_modes_are_reported).
    for mode in (oc.F32_FAITHFUL, oc.F32_SHIFTED):
        f = oc.process_frame(xyz, B, R, LEAF, WF, mode, want_outputs=False)
        own = ang(f["evecs"][:, 0], o["evecs"][:, 0])
        assert ang(res["center_axis"], f["evecs"][:, 0]) < own + 1e-5
        for k in (1, 2):
            own_l = abs(float(f["evals"][k]) - lo[k]) / lo[k]
            assert abs(l[k] - f["evals"][k]) / lo[k] < own_l + 1e-5


def test_deltas_against_the_f32_faithful_modes_are_reported(gm, oc, tmp_path_factory):
    """Parity unpinned, quantified: how far the GPU frame is from the oracle's two fp32-faithful modes and from its f64
    mode, per kind of cloud -- including the kinds where PCL's float rounding decides the outcome (duplicates, collinear
    runs: the kernel's fp64 solve returns NaN for an exactly singular covariance where PCL's float cross products
    usually give a finite normal, so n_valid may differ there).  Written to gpurun_out/parity_deltas.json for
    DESIGN.md par. 5; the asserts only bound what was observed when the table was made."""
    import json, os
    rng = np.random.default_rng(5)
    base = rng.uniform(-2, 2, (400, 3)).astype(np.float32)
    t = np.linspace(-1, 1, 300, dtype=np.float32)
    clouds = {
        "tunnel_50k (BASELINE configs[0] shape)": synth.tunnel_frame(50000, seed=0),
        "tunnel_floor_outliers_60k": synth.tunnel_frame(60000, seed=2, floor_z=-1.2, outlier_frac=0.01),
        "duplicates_4k (400 sites x10)": base[rng.integers(len(base), size=4000)],
        "collinear_run_300 + tunnel_5k": np.vstack([np.stack([t, t * 0.5, t * 0 + 1], axis=1), synth.tunnel_frame(5000, seed=3)]).astype(np.float32),
    }
    report = {}
    with gm.GeometricMapping(boxFilterBound=B, voxelGridLeafSize=LEAF, neighborRadius=R, weightingFactor=WF) as c:
        for name, xyz in clouds.items():
            res = c.process_frame(xyz)
            nrm = c.normals()
            _, rows = c.cropped_cloud()
            row = {"n_valid_gpu": int(res["n_valid"])}
            for tag, mode in (("f64", oc.F64), ("f32_pcl19", oc.F32_FAITHFUL), ("f32_pcl110", oc.F32_SHIFTED)):
                o = oc.process_frame(xyz, B, R, LEAF, WF, mode)
                both, ig, io = np.intersect1d(rows, oc.crop_box(xyz, B)[oc.finite_normals(oc.normals(xyz[oc.crop_box(xyz, B)], R, mode)[0])], return_indices=True)
                a = ang(nrm[ig, :3], o["normals"][io, :3]) if len(both) else np.zeros(1)
                row[tag] = {"n_valid": int(o["n_valid"]), "n_valid_diff": int(res["n_valid"]) - int(o["n_valid"]),
                            "normal_angle_p50": float(np.quantile(a, 0.5)), "normal_angle_p999": float(np.quantile(a, 0.999)),
                            "axis_angle": float(ang(res["center_axis"], o["evecs"][:, 0])),
                            "lambda2_rel": float(abs(float(res["eigenvalues"][2]) - float(o["evals"][2])) / max(float(o["evals"][2]), 1e-30))}
            report[name] = row
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_deltas.json"), "w") as f:
        json.dump(report, f, indent=1)
    for name in list(clouds)[:2]:                       # well-posed clouds: identical point sets, tight frame outputs
        r = report[name]
        assert r["f64"]["n_valid_diff"] == 0 and r["f32_pcl19"]["n_valid_diff"] == 0
        assert r["f64"]["axis_angle"] < 1e-5 and r["f64"]["normal_angle_p999"] < 1e-5
        assert r["f32_pcl110"]["axis_angle"] < 1e-5


def test_process_frame_is_deterministic(ctx):
    xyz = synth.tunnel_frame(40000, seed=9, outlier_frac=0.01)
    a = ctx.process_frame(xyz)
    na = ctx.normals()
    b = ctx.process_frame(xyz)
    nb = ctx.normals()
    assert np.array_equal(na, nb)                             # bitwise: stable sort + fixed-order sums
    assert np.array_equal(a["scatter6"], b["scatter6"])
    assert np.array_equal(a["eigenvectors"], b["eigenvectors"])


def test_process_frame_edge_cases(ctx):
    r = ctx.process_frame(np.zeros((0, 3), np.float32))
    assert r["n_in"] == 0 and r["n_cropped"] == 0 and r["n_valid"] == 0 and r["n_voxels"] == 0
    assert np.all(r["eigenvalues"] == 0)
    r = ctx.process_frame(np.full((500, 3), 7.0, np.float32))          # everything outside the box
    assert r["n_cropped"] == 0 and r["n_valid"] == 0
    r = ctx.process_frame(np.array([[0, 0, 1], [0, 1, 0]], np.float32))  # fewer than 3 points: all NaN normals
    assert r["n_cropped"] == 2 and r["n_valid"] == 0
    # ragged size right after a big frame (grow-only buffers, stale data must not leak)
    ctx.process_frame(synth.tunnel_frame(30000, seed=1))
    r = ctx.process_frame(synth.tunnel_frame(777, seed=2))
    assert r["n_in"] == 777 and 0 < r["n_cropped"] <= 777


@pytest.mark.parametrize("ransac", [False, True])
def test_empty_cloud_as_the_very_first_frame(gm, ransac):
    """A sensor that publishes an empty cloud first: a FRESH context (no buffers yet, max_points = 0) must come
    through every stage with null-free scratch (regression: k_compact_count<DensePred> wrote through sl.blk == NULL)."""
    from geometric_mapping_amd import _lib
    flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_NEAREST
    if ransac:
        flags |= _lib.GM_CFG_RANSAC_PLANE | _lib.GM_CFG_RANSAC_CYLINDER
    with gm.GeometricMapping(flags=flags) as c:
        r = c.process_frame(np.zeros((0, 3), np.float32))
        assert r["n_in"] == 0 and r["n_cropped"] == 0 and r["n_valid"] == 0 and r["n_voxels"] == 0
        assert np.all(r["eigenvalues"] == 0)
        if ransac:
            assert r["plane_inliers"] == 0 and r["cylinder_inliers"] == 0
        r = c.process_frame(synth.tunnel_frame(5000, seed=4))      # and the context is still usable
        assert r["n_valid"] > 0


def test_voxel_passthrough_flag_does_not_leak_into_frames(gm):
    """gm_voxel_grid (sort path) may raise GM_RES_VOXEL_PASSTHROUGH; a later frame on the dense-table path must not
    report the stale flag."""
    from geometric_mapping_amd import _lib
    with gm.GeometricMapping() as c:
        pts = np.array([[0, 0, 0], [1000, 1000, 1000]], np.float32)
        _, _, passed = c.voxelGrid(1e-3, pts)
        assert passed
        r = c.process_frame(synth.tunnel_frame(5000, seed=4))
        assert not (r["status_flags"] & _lib.GM_RES_VOXEL_PASSTHROUGH)


def test_process_frame_device_resident_rows(ctx, oc):
    torch = pytest.importorskip("torch")
    xyz = synth.tunnel_frame(30000, seed=3)
    rows = np.zeros((len(xyz), 4), np.float32)
    rows[:, :3] = xyz
    t = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    res = ctx.process_frame(ctx.cloud_from_device(t.data_ptr(), len(xyz), 16))
    ref = ctx.process_frame(xyz)
    assert res["n_valid"] == ref["n_valid"]
    assert np.array_equal(res["scatter6"], ref["scatter6"])


def test_graph_replays_device_resident_rows_at_any_address(gm):
    """GM_CFG_GRAPH with rows resident in HBM: the address of the rows travels through a device word like the point count,
    so frames handed over at different addresses replay ONE captured chain (it used to be re-captured per address) and
    every one of them equals the frame enqueued launch by launch; rows at a 4-byte-aligned address read with another row
    mode and take a second capture."""
    torch = pytest.importorskip("torch")
    from geometric_mapping_amd import _lib
    lib = _lib.load()
    frames = [synth.tunnel_frame(60000 - 37 * k, seed=70 + k, outlier_frac=0.01) for k in range(6)]
    pool = torch.zeros(6 * 60000 * 4 + 64, dtype=torch.float32, device="cuda")
    keys = ("n_cropped", "n_valid", "n_voxels", "eigenvalues", "eigenvectors", "scatter6")
    fl = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_CYLINDER
    with gm.GeometricMapping(neighborRadius=0.4, flags=fl) as a, gm.GeometricMapping(neighborRadius=0.4, flags=fl | _lib.GM_CFG_GRAPH) as b:
        def run(k, first_word):
            xyz = frames[k]
            rows = np.zeros((len(xyz), 4), np.float32)
            rows[:, :3] = xyz
            view = pool[first_word:first_word + rows.size]
            view.copy_(torch.from_numpy(rows.reshape(-1)))
            torch.cuda.synchronize()
            rb = b.process_frame(b.cloud_from_device(view.data_ptr(), len(xyz), 16))
            ra = a.process_frame(xyz)
            for key in keys:
                assert np.array_equal(np.asarray(ra[key]), np.asarray(rb[key]), equal_nan=True), key
            assert np.array_equal(np.asarray(ra["cylinder"]), np.asarray(rb["cylinder"]), equal_nan=True)
            assert np.array_equal(a.normals(), b.normals(), equal_nan=True)
            assert np.array_equal(a.cropped_cloud()[0], b.cropped_cloud()[0])
        for k in range(6):
            run(k, k * 60000 * 4)                      # six addresses, all 16-byte aligned
        assert lib.gm_debug_graph_captures(b._ctx, 0) == 1
        run(0, 1)                                      # 4-byte aligned rows: another row mode, its own capture
        run(1, 60000 * 4 + 1)
        assert lib.gm_debug_graph_captures(b._ctx, 0) == 2
        run(2, 0)                                      # ... and the first capture is still there
        assert lib.gm_debug_graph_captures(b._ctx, 0) == 2


@pytest.mark.parametrize("graph", [False, True])
def test_streaming_slots(gm, oc, graph):
    """(graph: every slot replays its own captured launch chain, GM_CFG_GRAPH)"""
    from geometric_mapping_amd import _lib
    frames = [synth.tunnel_frame(20000 - 37 * s, seed=s) for s in range(6)]
    with gm.GeometricMapping(n_slots=2, flags=_lib.GM_CFG_DEFAULT | (_lib.GM_CFG_GRAPH if graph else 0)) as c:
        single = [c.process_frame(f)["scatter6"] for f in frames]
        out = [None] * len(frames)
        c.submit_frame(0, frames[0])
        for i in range(1, len(frames)):
            c.submit_frame(i % 2, frames[i])
            out[i - 1] = c.wait_frame((i - 1) % 2)["scatter6"]
        out[-1] = c.wait_frame((len(frames) - 1) % 2)["scatter6"]
    for a, b in zip(single, out):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ full-size, size-independent properties

def test_full_size_properties_1m(gm):
    n = 1_000_000
    r = synth.fixed_k_radius(n)
    xyz = synth.tunnel_frame(n, seed=0)
    with gm.GeometricMapping(neighborRadius=r) as c:
        res = c.process_frame(xyz)
        cloud, rows = c.cropped_cloud()
        nrm = c.normals()
        cen, cnt = c.voxel_centroids()
    inside = np.all(np.abs(xyz) <= 5.0, axis=1)
    assert res["n_cropped"] == int(inside.sum())
    assert np.all(np.diff(rows) > 0) and np.array_equal(xyz[rows], cloud)     # order-preserving
    assert np.isfinite(nrm).all()
    assert np.abs(np.linalg.norm(nrm[:, :3].astype(np.float64), axis=1) - 1).max() < 1e-6
    assert ((-cloud.astype(np.float64) * nrm[:, :3]).sum(axis=1) >= 0).all()  # flipped to the sensor
    assert cnt.sum() == res["n_valid"] and res["n_voxels"] == len(cen)
    # scatter matrix: symmetric PSD, trace = sum w^2 >= n
    M = res["scatter"]
    assert np.trace(M) >= res["n_valid"] and np.trace(M) < 1.3 * res["n_valid"]
    assert res["eigenvalues"][0] >= -1e-3 and np.all(np.diff(res["eigenvalues"]) >= 0)
    assert ang(res["center_axis"], [1, 0, 0]) < 2e-3                           # analytic truth: tunnel along x
    # linearity: M(frame) == M(first half of normals) + M(second half)
    h = len(nrm) // 2
    with gm.GeometricMapping() as c2:
        _, _, Ma = c2.getLocalFrame(h, WF, nrm[:h])
        _, _, Mb = c2.getLocalFrame(len(nrm) - h, WF, nrm[h:])
    assert np.abs((Ma + Mb) - M).max() / np.abs(M).max() < 1e-12


def test_normals_grid_stride_path_is_bitwise_identical(gm):
    """Frames beyond 262 144 tiles (~16 M points) make each wave of k_normals walk several tiles (grid-stride).  No test
    frame is that large, so the path is forced on a small frame by capping the grid (GM_NORMALS_BLOCKS, read once per
    process -> a child process): which wave computes a tile must not change a single bit."""
    import subprocess, sys, tempfile, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import geometric_mapping_amd as g\n"
        "from geometric_mapping_amd import synth\n"
        "xyz = synth.tunnel_frame(120000, seed=9, floor_z=-1.2, outlier_frac=0.01)\n"
        "with g.GeometricMapping(neighborRadius=0.25) as c:\n"
        "    res = c.process_frame(xyz)\n"
        "    np.savez(sys.argv[1], nrm=c.normals(), cloud=c.cropped_cloud()[0], scatter=res['scatter'], cen=c.voxel_centroids()[0])\n"
    ) % root
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for tag, blocks in (("all", None), ("capped", "7")):
            env = dict(os.environ)
            env.pop("GM_NORMALS_BLOCKS", None)
            if blocks:
                env["GM_NORMALS_BLOCKS"] = blocks
            f = os.path.join(d, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            outs.append(dict(np.load(f)))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k


def test_tile_lists_hold_every_tile_of_a_sparse_lattice(gm, oc):
    """k_normals walks its tiles by cost class (x extent).  A class other than the last holds tiles of at least two points,
    so no list can hold more than capacity / 2 tiles and none can overflow (k_rows_and_tiles, csrc/k_normals.hip).  A
    lattice cloud whose rows hold one point every 1.2 r is cut into tiles of two or three points with two extents only --
    as many tiles per class as a frame can have.  Every point must get its neighbours (counts bit-exact) and its normal."""
    from geometric_mapping_amd import _lib
    r = 0.25
    ax = np.arange(-4.8, 4.8, 1.2 * r, dtype=np.float32)
    ay = np.arange(-1.0, 1.0, 0.9 * r, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ay, ay, indexing="ij")
    rng = np.random.default_rng(5)
    xyz = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)
    xyz += rng.normal(0, 0.01, xyz.shape).astype(np.float32)          # (no exactly coplanar neighbourhoods)
    xyz = xyz[rng.permutation(len(xyz))]
    with gm.GeometricMapping(boxFilterBound=5.0, neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS,
                             max_points=len(xyz)) as c:
        res = c.process_frame(xyz)
        cnt = c.neighbor_counts()
        nrm = c.normals()
    keep = oc.crop_box(xyz, 5.0)
    o_n, o_cnt = oc.normals(xyz[keep], r, oc.F64)
    assert res["n_cropped"] == len(keep) == len(xyz)
    assert np.array_equal(cnt, o_cnt)
    valid = oc.finite_normals(o_n)
    assert res["n_valid"] == len(valid)
    s = np.linalg.norm(np.cross(nrm[:, :3].astype(np.float64), o_n[valid][:, :3].astype(np.float64)), axis=1)
    assert np.quantile(np.arcsin(np.clip(s, 0, 1)), 0.999) < 1e-5


def test_chained_scan_epochs_wrap_cleanly(gm):
    """The chained scans tag their records with an epoch that wraps (host counter: every 2^29 - 2 launches of a slot;
    replayed graphs: every 2^24 frames).  At the wrap the record array is cleared so that no record of the previous
    period can read as ready.  The counters are started just below the wrap (GM_TEST_*, read at context creation -> child
    processes): frames of changing size on either side of it must be the frames of an ordinary context, bit for bit."""
    import subprocess, sys, tempfile, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import geometric_mapping_amd as g\n"
        "from geometric_mapping_amd import synth, _lib\n"
        "flags = _lib.GM_CFG_DEFAULT | int(sys.argv[2])\n"
        "out = {}\n"
        "with g.GeometricMapping(neighborRadius=0.25, flags=flags) as c:\n"
        "    for i, n in enumerate([90000, 5000, 90000, 20000, 90000, 300, 90000, 40000]):\n"
        "        xyz = synth.tunnel_frame(n, seed=30 + i, floor_z=-1.2, outlier_frac=0.01)\n"
        "        res = c.process_frame(xyz)\n"
        "        out['cloud%%d' %% i] = c.cropped_cloud()[0]; out['rows%%d' %% i] = c.cropped_cloud()[1]\n"
        "        out['nrm%%d' %% i] = c.normals(); out['scatter%%d' %% i] = res['scatter']; out['cen%%d' %% i] = c.voxel_centroids()[0]\n"
        "np.savez(sys.argv[1], **out)\n"
    ) % root
    from geometric_mapping_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        runs = {}
        for tag, flags, extra in (("plain", 0, {}), ("host_wrap", 0, {"GM_TEST_SCAN_EPOCH": str(0x1FFFFFFE - 7)}),
                                  ("graph", _lib.GM_CFG_GRAPH, {}),
                                  ("graph_wrap", _lib.GM_CFG_GRAPH, {"GM_TEST_FRAME_COUNTER": str(0x800000 - 3)}),
                                  ("graph_wrap2", _lib.GM_CFG_GRAPH, {"GM_TEST_FRAME_COUNTER": str(0x1000000 - 3)})):
            env = dict(os.environ)
            env.pop("GM_TEST_SCAN_EPOCH", None); env.pop("GM_TEST_FRAME_COUNTER", None)
            env.update(extra)
            f = os.path.join(d, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f, str(flags)], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            runs[tag] = dict(np.load(f))
    for tag in ("host_wrap", "graph", "graph_wrap", "graph_wrap2"):
        for k in runs["plain"]:
            assert np.array_equal(runs["plain"][k], runs[tag][k], equal_nan=True), (tag, k)


def test_pinned_host_rows_equal_pageable_rows(gm, oc):
    """GM_CLOUD_PINNED (rows in gm_host_alloc memory, no staging copy) is the same frame as pageable input, also when
    the frames are submitted asynchronously from two pinned buffers."""
    xyz = [synth.tunnel_frame(30000, seed=s, outlier_frac=0.01) for s in (21, 22)]
    with gm.GeometricMapping(n_slots=2) as c:
        ref = []
        for x in xyz:
            r = c.process_frame(x)
            ref.append((r["scatter"].copy(), c.normals().copy(), c.cropped_cloud()[0].copy()))
        clouds = []
        for x in xyz:
            buf, as_cloud = c.pinned_rows(len(x), 12)
            buf[:] = np.ascontiguousarray(x).reshape(-1).view(np.uint8)
            clouds.append(as_cloud())
        for s, cl in enumerate(clouds):
            c.submit_frame(s, cl)
        for s in range(2):
            r = c.wait_frame(s)
            assert np.array_equal(r["scatter"], ref[s][0])
            assert np.array_equal(c.normals(s), ref[s][1], equal_nan=True)
            assert np.array_equal(c.cropped_cloud(s)[0], ref[s][2])


def test_frame_smaller_than_capacity_with_another_sort_layout(gm):
    """The radix sort picks its tile from the element count, and the block count is NOT monotonic in it: a context
    with a 2.1 M-point capacity lays out 129 blocks (16 keys per thread), a 1.9 M-point frame 232 (8 keys per thread).
    The sort scratch must be sized for every count up to the capacity (regression: that frame used to run past the
    histogram scratch and failed with hipErrorInvalidValue)."""
    xyz = synth.tunnel_frame(1_900_000, seed=5, outlier_frac=0.01)
    r = synth.fixed_k_radius(1_900_000)
    with gm.GeometricMapping(neighborRadius=r, max_points=1_900_000) as c:
        ref = c.process_frame(xyz)
        ref_n = c.normals().copy()
    with gm.GeometricMapping(neighborRadius=r, max_points=2_100_000) as c:     # capacity = exactly 2.1 M
        c.process_frame(xyz[:1_000_000])
        res = c.process_frame(xyz)
        nrm = c.normals()
    assert np.array_equal(res["scatter"], ref["scatter"]) and res["n_valid"] == ref["n_valid"]
    assert np.array_equal(nrm, ref_n, equal_nan=True)


@pytest.mark.parametrize("impl", ["valu", "mfma", "auto0", "mfma0", "rows2", "rows4", "sortstaged"])
def test_every_formulation_of_the_neighbourhood_kernel_finds_the_same_neighbours(impl):
    """GM_NORMALS_IMPL (read once per process -> child processes): the all-VALU kernel, the production kernel (distances
    and moments on the matrix cores, exact re-evaluation inside a band) forced onto every tile, and the variant that
    keeps the neighbour predicate on the VALU must report the neighbour counts and NaN pattern of the default build bit for bit, and normals
    within 1e-5 rad -- on a dense frame and on a sparse one (thin neighbourhoods, long tiles).  rows2 / rows4: the same for
    the fine-row mode of the production kernel (GM_NORMALS_ROWS: y/z cells r/2 and r/4 wide, 25 / 81 rows per tile in
    passes, per-row x-reach), forced onto frames it would not be chosen for."""
    import subprocess, sys, tempfile, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import geometric_mapping_amd as g\n"
        "from geometric_mapping_amd import _lib, synth\n"
        "out = {}\n"
        "for name, xyz, r in (('dense', synth.tunnel_frame(150000, seed=4, floor_z=-1.2, outlier_frac=0.01), 0.3),\n"
        "                     ('sparse', synth.tunnel_frame(3000, seed=5, outlier_frac=0.05), 0.45)):\n"
        "    with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS) as c:\n"
        "        res = c.process_frame(xyz)\n"
        "        out[name + '_nrm'] = c.normals(); out[name + '_cnt'] = c.neighbor_counts(); out[name + '_sc'] = res['scatter6']\n"
        "np.savez(sys.argv[1], **out)\n"
    ) % root
    got = []
    with tempfile.TemporaryDirectory() as d:
        for tag in (None, impl):
            env = dict(os.environ)
            env.pop("GM_NORMALS_IMPL", None)
            env.pop("GM_NORMALS_ROWS", None)
            env.pop("GM_SORT_STAGED", None)
            if tag == "sortstaged":     # the LDS-staged radix scatter (chosen for frames beyond ~1 M points) on small frames
                env["GM_SORT_STAGED"] = "1"
            elif tag and tag.startswith("rows"):
                env["GM_NORMALS_ROWS"] = tag[4:]
            elif tag:
                env["GM_NORMALS_IMPL"] = tag
            f = os.path.join(d, (tag or "default") + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            got.append(dict(np.load(f)))
    a, b = got
    for name in ("dense", "sparse"):
        assert np.array_equal(a[name + "_cnt"], b[name + "_cnt"]), name
        na, nb = a[name + "_nrm"], b[name + "_nrm"]
        assert np.array_equal(np.isnan(na), np.isnan(nb))
        ok = np.isfinite(na[:, 0])
        assert np.quantile(ang(na[ok, :3], nb[ok, :3]), 0.999) < 1e-5
        assert np.abs(a[name + "_sc"] - b[name + "_sc"]).max() / np.abs(a[name + "_sc"]).max() < 1e-6


@pytest.mark.parametrize("ransac", [False, True, "nearest"])
def test_graph_replay_is_bitwise_the_enqueued_frame(gm, ransac):
    """GM_CFG_GRAPH: the frame's launch chain is captured once per bucket of frame sizes and replayed; frames of
    different sizes inside one bucket (the count travels through device memory), of another bucket (re-capture), an
    empty frame in between (enqueued directly) -- every result and every bulk output must equal the frame enqueued
    launch by launch, bit for bit."""
    from geometric_mapping_amd import _lib
    nearest = ransac == "nearest"   # (the node's configuration with displayNormals: 1-NN + gathered normals in the frame)
    ransac = ransac is True
    fl = _lib.GM_CFG_DEFAULT | (_lib.GM_CFG_RANSAC_CYLINDER if ransac else 0) | (_lib.GM_CFG_NEAREST if nearest else 0)
    sizes = [60000, 60000, 58111, 61440, 0, 200000, 57345, 60000]
    frames = [synth.tunnel_frame(n, seed=11 + i) if n else np.zeros((0, 3), np.float32) for i, n in enumerate(sizes)]
    keys = ("n_cropped", "n_valid", "n_voxels", "eigenvalues", "eigenvectors", "center_axis", "scatter6")
    with gm.GeometricMapping(neighborRadius=0.4, flags=fl) as a, gm.GeometricMapping(neighborRadius=0.4, flags=fl | _lib.GM_CFG_GRAPH) as b:
        for xyz in frames:
            ra, rb = a.process_frame(xyz), b.process_frame(xyz)
            for k in keys:
                assert np.array_equal(np.asarray(ra[k]), np.asarray(rb[k]), equal_nan=True), k
            if ransac:
                assert np.array_equal(np.asarray(ra["cylinder"]), np.asarray(rb["cylinder"]), equal_nan=True)
                assert ra["cylinder_inliers"] == rb["cylinder_inliers"]
            if len(xyz):
                assert np.array_equal(a.normals(), b.normals(), equal_nan=True)
                (ca, ia), (cb, ib) = a.cropped_cloud(), b.cropped_cloud()
                assert np.array_equal(ca, cb) and np.array_equal(ia, ib)
                (va, na), (vb, nb) = a.voxel_centroids(), b.voxel_centroids()
                assert np.array_equal(va, vb) and np.array_equal(na, nb)
                if nearest:
                    assert np.array_equal(a.voxel_normals(), b.voxel_normals(), equal_nan=True)


def test_poll_frame_and_cloud_output(gm):
    """gm_poll_frame never blocks and turns true once a submitted frame has finished; gm_set_cloud_output delivers
    /choppedCloud into caller-owned page-locked rows during the frame -- bytes identical to gm_get_cropped_xyz -- for
    blocking calls, submitted frames, and frames replayed from a graph."""
    from geometric_mapping_amd import _lib
    xyz = [synth.tunnel_frame(50000 + 3000 * s, seed=60 + s, outlier_frac=0.01) for s in range(3)]
    for flags in (_lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_CYLINDER, _lib.GM_CFG_DEFAULT | _lib.GM_CFG_GRAPH):
        with gm.GeometricMapping(n_slots=2, neighborRadius=0.3, flags=flags) as c:
            assert c.poll_frame(0) is False and c.poll_frame(1) is False      # no frame submitted
            bufs = [c.cloud_output(s, 70000) for s in range(2)]
            for k, x in enumerate(xyz):
                slot = k % 2
                c.submit_frame(slot, x)
                spins = 0
                while not c.poll_frame(slot):
                    spins += 1
                    assert spins < 50_000_000
                r = c.wait_frame(slot)
                assert c.poll_frame(slot) is True                              # stays true until the slot is submitted to again
                cloud, rows = c.cropped_cloud(slot)
                n = r["n_valid"]
                assert n == len(cloud) > 0
                assert np.array_equal(bufs[slot][:n, :3], cloud)
                assert np.array_equal(bufs[slot][:n, 3].copy().view(np.int32), rows)
            r = c.process_frame(xyz[0])                                        # blocking calls use slot 0
            assert np.array_equal(bufs[0][:r["n_valid"], :3], c.cropped_cloud(0)[0])
            with pytest.raises(gm.GmError):                                    # more points than the buffer has rows
                c.process_frame(synth.tunnel_frame(80000, seed=1))


@pytest.mark.parametrize("rings,az,step,pad", [(16, 900, 32, 24), (32, 600, 22, 10)])
def test_lidar_shaped_organised_cloud_matches_oracle(gm, oc, rings, az, step, pad):
    """What /velodyne_points carries (launch/mapping.launch:22): an ORGANISED height x width cloud of a spinning multi-beam
    lidar inside the tunnel -- rows of the message padded (row_step), 32-byte XYZIR or unaligned 22-byte XYZIRT points, NaN
    returns, the sensor at the origin, density falling off with range (neighbour counts from a dozen to thousands at the
    launch file's neighborRadius = 0.5) -- through the frame path with the launch file's values.  Crop, neighbour sets,
    NaN-normal removal and voxels exact; normals, scatter matrix and axis to north_star's 1e-5 where well-posed."""
    from geometric_mapping_amd import _lib
    msg = synth.velodyne_tunnel(rings=rings, az=az, seed=9, point_step=step, row_pad=pad)
    xyz = msg["xyz"]
    rows = synth.drop_row_padding(msg)          # (what the node does with padded rows)
    assert msg["row_step"] == az * step + pad and np.isnan(xyz[:, 0]).sum() > 0
    bound, radius, leaf, wf = 5.0, 0.5, 0.5, 0.2
    with gm.GeometricMapping(boxFilterBound=bound, voxelGridLeafSize=leaf, neighborRadius=radius, weightingFactor=wf,
                             flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_KEEP_COUNTS) as c:
        res = c.process_frame(c.cloud_from_rows(rows, len(xyz), step, msg["offsets"]))
        cloud, crows = c.cropped_cloud()
        nrm = c.normals()
        cnt = c.neighbor_counts()
        cen, vcnt = c.voxel_centroids()
    keep = oc.crop_box(xyz, bound)
    o_n, o_cnt = oc.normals(xyz[keep], radius, oc.F64)
    assert res["n_in"] == rings * az and res["n_cropped"] == len(keep)
    assert np.array_equal(cnt, o_cnt) and o_cnt.max() > 20 * max(int(o_cnt.min()), 1)     # density really varies
    valid = oc.finite_normals(o_n)
    o = oc.process_frame(xyz, bound, radius, leaf, wf, oc.F64)
    assert np.array_equal(crows, keep[valid]) and np.array_equal(cloud, xyz[keep][valid])
    assert res["n_valid"] == o["n_valid"] and res["n_voxels"] == o["n_voxels"]
    assert np.abs(cen - o["voxels"]).max() < 2e-6 * bound
    # normals: a neighbourhood that is one ring's arc is a line, its plane is rounding noise in any precision -- the bound
    # is north_star's 1e-5 or half of what the reference's own fp32 arithmetic deviates from the f64 value, as in the fuzz
    s = np.linalg.norm(np.cross(nrm[:, :3].astype(np.float64), o["normals"][:, :3].astype(np.float64)), axis=1)
    a = np.arcsin(np.clip(s, 0, 1))
    o32 = oc.normals(xyz[keep], radius, oc.F32_FAITHFUL)[0][valid]
    fin = np.isfinite(o32[:, 0])
    s32 = np.linalg.norm(np.cross(o32[fin, :3].astype(np.float64), o["normals"][fin, :3].astype(np.float64)), axis=1)
    ref_dev = float(np.quantile(np.arcsin(np.clip(s32, 0, 1)), 0.98))
    assert np.quantile(a, 0.98) < max(1e-5, 0.5 * ref_dev), (np.quantile(a, 0.98), ref_dev)
    M = o["M"]
    assert np.abs(res["scatter"] - M).max() / np.abs(M).max() < 1e-5
    ev = o["evals"].astype(np.float64)
    assert abs(res["eigenvalues"][2] - ev[2]) / ev[2] < 1e-5 and abs(res["eigenvalues"][1] - ev[1]) / ev[1] < 1e-5
    ax, oax = res["center_axis"].astype(np.float64), o["evecs"][:, 0].astype(np.float64)
    assert np.linalg.norm(np.cross(ax, oax)) / (np.linalg.norm(ax) * np.linalg.norm(oax)) < 1e-5


def test_stage_call_does_not_clobber_a_submitted_frame(gm):
    """The stage calls work in slot 0.  While a frame submitted to slot 0 has not been waited for they are refused
    (GM_ERR_NOT_READY) instead of overwriting the slot's buffers; the frame's result stays intact."""
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(40000, seed=5, outlier_frac=0.01)
    with gm.GeometricMapping(neighborRadius=0.3) as c:
        ref = c.process_frame(xyz)
        c.submit_frame(0, xyz)
        with pytest.raises(gm.GmError) as e:
            c.chopCloud(5.0, xyz[:100])
        assert e.value.status == _lib.GM_ERR_NOT_READY
        r = c.wait_frame(0)
        assert r["n_valid"] == ref["n_valid"] and np.array_equal(r["scatter6"], ref["scatter6"])
        out, rows = c.chopCloud(5.0, xyz[:100])            # after the wait the slot is free again
        assert len(out) == len(rows) <= 100
