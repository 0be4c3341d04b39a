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
