"""gm_group: multi-device inside the C ABI (include/gm_hip.h "multi-device group"; SURVEY.md par. 8b/8e).

One GPU is all the test box has, so: (a) a 1-rank group exercises the real RCCL path (ncclCommInitAll with one device,
ncclAllGather on the frame's stream) and must reproduce gm_process_frame bit for bit; (b) several ranks on device 0
(GM_GROUP_LOOPBACK: records travel by device copies, everything else is the same code) must reproduce the unsharded
frame -- the BASELINE configs[3] shape (one frame over 4 ranks) at test size."""
import ctypes as C

import numpy as np
import pytest

from geometric_mapping_amd import synth


def ang(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.arcsin(min(1.0, np.linalg.norm(np.cross(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))))


def test_group_create_rejects_bad_arguments_without_a_gpu():
    from geometric_mapping_amd import _lib
    L = _lib.load()
    cfg = _lib.Config()
    L.gm_default_config(C.byref(cfg))
    grp = C.c_void_p()
    dev = (C.c_int32 * 2)(0, 0)
    assert L.gm_group_create(C.byref(cfg), dev, 0, 0, C.byref(grp)) == _lib.GM_ERR_INVALID_ARG
    assert L.gm_group_create(C.byref(cfg), dev, 2, 0, C.byref(grp)) == _lib.GM_ERR_INVALID_ARG   # device listed twice
    assert b"LOOPBACK" in L.gm_group_last_error(None)
    assert L.gm_group_size(None) == 0 and not L.gm_group_ctx(None, 0)
    assert L.gm_status_string(_lib.GM_ERR_COMM) == b"communication (RCCL) error"


@pytest.mark.gpu
def test_one_rank_group_over_rccl_equals_process_frame(gm):
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(60000, seed=11, floor_z=-1.2, outlier_frac=0.01)
    kw = dict(neighborRadius=0.4, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_CYLINDER, ransac_seed=5)
    with gm.GeometricMapping(**kw) as c:
        ref = c.process_frame(xyz)
        rcloud, rrows = c.cropped_cloud()
    with gm.GeometricMappingGroup([0], **kw) as g:          # distinct devices -> the RCCL communicator is built
        assert len(g) == 1
        res = g.process_frame(xyz)
        cloud, rows = g.cropped_cloud()
    for k in ("n_in", "n_cropped", "n_valid", "n_voxels", "cylinder_inliers"):
        assert res[k] == ref[k], k
    assert np.array_equal(res["scatter6"], ref["scatter6"])                      # one slab = the whole frame: bitwise
    assert np.array_equal(res["eigenvalues"], ref["eigenvalues"]) and np.array_equal(res["eigenvectors"], ref["eigenvectors"])
    assert np.array_equal(res["cylinder"], ref["cylinder"])
    assert np.array_equal(rows, rrows) and np.array_equal(cloud, rcloud)


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 4])
def test_loopback_ranks_reproduce_the_unsharded_frame(gm, n_ranks):
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(120000, seed=12, floor_z=-1.2, outlier_frac=0.01)
    kw = dict(neighborRadius=0.3, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_PLANE | _lib.GM_CFG_RANSAC_CYLINDER)
    with gm.GeometricMapping(**kw) as c:
        ref = c.process_frame(xyz)
        rcloud, rrows = c.cropped_cloud()
        with gm.GeometricMappingGroup([0] * n_ranks, loopback=True, **kw) as g:
            res = g.process_frame(xyz)
            cloud, rows = g.cropped_cloud()
        for k in ("n_in", "n_cropped", "n_valid", "n_voxels"):
            assert res[k] == ref[k], k
        assert np.array_equal(rows, rrows) and np.array_equal(cloud, rcloud)     # ownership: every point exactly once, input order
        M, Mr = res["scatter6"], ref["scatter6"]
        assert np.abs(M - Mr).max() / np.abs(Mr).max() < 1e-6
        assert ang(res["center_axis"], ref["center_axis"]) < 1e-5
        for k in (1, 2):
            assert abs(res["eigenvalues"][k] - ref["eigenvalues"][k]) / ref["eigenvalues"][k] < 1e-5
        # the vote: the winner's global inlier count is the count of that primitive on the unsharded frame's cloud
        for model, key, inl in ((0, "plane", "plane_inliers"), (1, "cylinder", "cylinder_inliers")):
            assert np.isfinite(res[key]).all() and res[inl] > 0
            want = c.score_frame(model, res[key][None, :], 0.03)[0]
            assert res[inl] == want, (key, res[inl], want)
        assert abs(res["cylinder"][6] - 2.0) < 0.05                              # analytic truth: R = 2 tunnel


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 4])
def test_loopback_voxel_outputs_are_the_unsharded_frames(gm, n_ranks):
    """SURVEY par. 8(e): slab edges lie on planes of the VoxelGrid lattice, so no voxel straddles two ranks and the merged
    centroid list IS the unsharded one, bit for bit; the nearest point of a centroid is searched on every rank (it may
    lie across an edge), so /surfaceNormals' inputs (src/tunnel_processing.cpp:214-252) are the unsharded frame's too."""
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(150000, seed=13, floor_z=-1.2, outlier_frac=0.01)
    kw = dict(neighborRadius=0.3, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_NEAREST)
    with gm.GeometricMapping(**kw) as c:
        ref = c.process_frame(xyz)
        rcen, rcnt = c.voxel_centroids()
        rnrm, rnn = c.voxel_normals(), c.voxel_nearest()
        with gm.GeometricMappingGroup([0] * n_ranks, loopback=True, **kw) as g:
            res = g.process_frame(xyz)
            cen, cnt = g.voxel_centroids()
            nrm, nn = g.voxel_normals(), g.voxel_nearest()
            edges, on_lattice = g.edges()
            t = g.timing()
    assert on_lattice and len(edges) == n_ranks + 1 and np.all(np.diff(edges[1:-1]) >= 0)
    leaf = 0.5
    assert np.allclose(edges[1:-1] / leaf, np.round(edges[1:-1] / leaf), atol=1e-6)      # planes k * leaf
    assert res["n_voxels"] == ref["n_voxels"] == len(cen)
    assert np.array_equal(cen, rcen) and np.array_equal(cnt, rcnt)
    # the same nearest points; their normals agree to rounding (a rank's tiles are cut from its own slab, so the tile
    # origins -- and with them the last bits of a point's moments -- differ from the unsharded frame's)
    assert np.array_equal(nn, rnn) and np.array_equal(np.isnan(nrm), np.isnan(rnrm))
    assert np.nanmax(np.abs(nrm - rnrm)) < 1e-5
    assert t["total_ms"] > 0 and abs(t["cut_ms"] + t["submit_ms"] + t["device_ms"] + t["merge_ms"] - t["total_ms"]) < 1e-6


@pytest.mark.gpu
def test_loopback_voxels_on_a_lattice_too_coarse_to_cut(gm):
    """A leaf larger than the box leaves nothing to cut along: the edges fall back to count quantiles, a voxel holds points
    of several ranks, and the ranks' exact fixed-point sums are added before the one division -- still the unsharded
    centroids bit for bit."""
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(80000, seed=14, outlier_frac=0.01)
    for leaf in (50.0, 4.0):
        kw = dict(neighborRadius=0.3, voxelGridLeafSize=leaf, flags=_lib.GM_CFG_DEFAULT)
        with gm.GeometricMapping(**kw) as c:
            ref = c.process_frame(xyz)
            rcen, rcnt = c.voxel_centroids()
            with gm.GeometricMappingGroup([0] * 4, loopback=True, **kw) as g:
                res = g.process_frame(xyz)
                cen, cnt = g.voxel_centroids()
                _, on_lattice = g.edges()
        assert not on_lattice
        assert res["n_voxels"] == ref["n_voxels"] and np.array_equal(cnt, rcnt) and np.array_equal(cen, rcen), leaf


@pytest.mark.gpu
def test_group_streams_frames_round_robin(gm):
    """BASELINE configs[4] inside the C ABI: frames handed round-robin to the group's devices (two ranks on this box's
    one GPU, two slots each), results returned in submission order, each bit-equal to the blocking call."""
    from geometric_mapping_amd import _lib
    kw = dict(neighborRadius=0.3, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_CYLINDER, ransac_seed=3)
    frames = [synth.tunnel_frame(20000 + 500 * (i % 7), seed=100 + i, outlier_frac=0.01) for i in range(100)]
    with gm.GeometricMapping(**kw) as c:
        ref = []
        for x in frames:
            r = c.process_frame(x)
            ref.append((r["n_valid"], r["scatter6"].copy(), r["eigenvectors"].copy(), r["cylinder"].copy(), c.cropped_cloud()[0][:50].copy()))
    got = []
    with gm.GeometricMappingGroup([0, 0], loopback=True, n_slots=2, **kw) as g:
        # (both ranks draw the same RANSAC hypotheses as the reference context only with the same seed: rank r uses seed + r)
        ranks = []
        i = 0
        while i < len(frames) or g.in_flight():
            while i < len(frames) and g.in_flight() < 4:
                g.submit_frame(frames[i]); i += 1
            r, rank, slot = g.wait_frame()
            ranks.append(rank)
            got.append((r, g.rank_fetch(rank, slot, "cropped_xyz")[:50, :3].copy(), rank))
        with pytest.raises(gm.GmError):
            g.wait_frame()
    assert len(got) == len(frames) and set(ranks) == {0, 1}
    for k, ((r, head, rank), x) in enumerate(zip(got, ref)):
        assert r["n_valid"] == x[0] and np.array_equal(r["scatter6"], x[1]) and np.array_equal(r["eigenvectors"], x[2]), k
        assert np.array_equal(head, x[4]), k
        if rank == 0:
            assert np.array_equal(r["cylinder"], x[3]), k     # rank 0 has the reference context's seed


@pytest.mark.gpu
def test_group_failure_leaves_no_stale_frame(gm):
    """A frame that fails on a rank (here: a row layout whose offsets do not fit) must leave nothing in flight and nothing
    for the accessors to pair with an older frame's rows."""
    from geometric_mapping_amd import _lib
    xyz = synth.tunnel_frame(40000, seed=15)
    with gm.GeometricMappingGroup([0, 0], loopback=True, neighborRadius=0.3) as g:
        ok = g.process_frame(xyz)
        assert ok["n_valid"] > 0 and len(g.cropped_cloud()[0]) == ok["n_valid"]
        bad = (_lib.Cloud(xyz.ctypes.data, len(xyz), 12, 0, 4, 10, 0), xyz)      # z at bytes 10..13 of a 12-byte row
        with pytest.raises(gm.GmError):
            g.process_frame(bad)
        with pytest.raises(gm.GmError):
            g.cropped_cloud()
        with pytest.raises(gm.GmError):
            g.voxel_centroids()
        again = g.process_frame(xyz)
        assert again["n_valid"] == ok["n_valid"] and np.array_equal(again["scatter6"], ok["scatter6"])


@pytest.mark.gpu
def test_group_poll_driven_drain_returns_every_frame_in_order(gm):
    """The node's publish loop (ros/geometric_mapping_node.cpp): frames leave the pipeline through the completion query
    gm_group_poll_frame -- before each submit and then until nothing is in flight -- never by waiting for the pipeline to
    fill.  All 100 frames come back, in submission order, each the blocking frame bit for bit; the query never blocks and
    answers "not ready" on an empty pipeline."""
    from geometric_mapping_amd import _lib
    kw = dict(neighborRadius=0.3, flags=_lib.GM_CFG_DEFAULT)
    frames = [synth.tunnel_frame(20000 + 700 * (i % 5), seed=300 + i, outlier_frac=0.01) for i in range(100)]
    with gm.GeometricMapping(**kw) as c:
        ref = []
        for x in frames:
            r = c.process_frame(x)
            ref.append((r["n_in"], r["n_valid"], r["scatter6"].copy(), c.cropped_cloud()[0][:40].copy()))
    got = []
    with gm.GeometricMappingGroup([0, 0], loopback=True, n_slots=2, **kw) as g:
        assert g.poll_frame() is False                       # nothing in flight
        i = polls = 0
        while len(got) < len(frames):
            while g.poll_frame():                            # publish what has finished
                r, rank, slot = g.wait_frame()
                got.append((r, g.rank_fetch(rank, slot, "cropped_xyz")[:40, :3].copy()))
            if i < len(frames) and g.in_flight() < 4:
                g.submit_frame(frames[i]); i += 1
            polls += 1
            assert polls < 50_000_000
        assert g.in_flight() == 0 and g.poll_frame() is False
    for k, ((r, head), x) in enumerate(zip(got, ref)):
        assert r["n_in"] == x[0] and r["n_valid"] == x[1] and np.array_equal(r["scatter6"], x[2]) and np.array_equal(head, x[3]), k


@pytest.mark.gpu
def test_streaming_after_a_sharded_frame_sees_whole_frames_again(gm):
    """A sharded frame leaves every rank with a slab range -- rank 0 with (-inf, edge[1]).  Frames streamed afterwards are
    whole frames: every rank must own all of x again, whichever rank the first streamed frame lands on (an odd number of
    streamed frames before the sharded one makes it rank 1)."""
    from geometric_mapping_amd import _lib
    kw = dict(neighborRadius=0.3, flags=_lib.GM_CFG_DEFAULT)
    frames = [synth.tunnel_frame(30000, seed=400 + i, outlier_frac=0.01) for i in range(6)]
    with gm.GeometricMapping(**kw) as c:
        ref = [c.process_frame(x) for x in frames]
    with gm.GeometricMappingGroup([0, 0], loopback=True, n_slots=1, **kw) as g:
        g.submit_frame(frames[0])                            # lands on rank 0; the next streamed frame goes to rank 1
        r0, rank0, _ = g.wait_frame()
        assert rank0 == 0 and r0["n_valid"] == ref[0]["n_valid"]
        sh = g.process_frame(frames[1])                      # sharded: ranks now own slabs
        assert sh["n_valid"] == ref[1]["n_valid"]
        seen = []
        for k in range(2, 6):
            g.submit_frame(frames[k])
            r, rank, _ = g.wait_frame()
            seen.append(rank)
            assert r["n_valid"] == ref[k]["n_valid"] and r["n_cropped"] == ref[k]["n_cropped"], (k, rank)
            assert np.array_equal(r["scatter6"], ref[k]["scatter6"]), (k, rank)
        assert set(seen) == {0, 1}
