"""Every BASELINE.json config that one GPU can hold, at its full size, through size-independent properties
(the oracle finishes 1 M-point frames in ~1 s on 16 cores, so configs[1] is also checked against it directly):

  configs[1]  1 M-point frame, one RANSAC model (cylinder)          -> test_config1_*
  configs[2]  10 M points, plane + cylinder                         -> tests/test_gpu_ext.py::test_full_size_properties_10m_plane_and_cylinder
  configs[3]  10 M-point frame over 4 ranks + all-gather            -> test_config3_* (4 ranks on this one GPU: gm_group, GM_GROUP_LOOPBACK)
  configs[4]  100 x 1 M-point frames streamed on HIP streams         -> test_config4_* (3 slots on this one GPU)
What still needs more than one GPU: the RCCL transport between distinct devices and the 8-way frame round-robin."""
import numpy as np
import pytest

from geometric_mapping_amd import synth

pytestmark = pytest.mark.gpu
TAU = 0.03


def ang(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.arcsin(min(1.0, np.linalg.norm(np.cross(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))))


def test_config1_1m_frame_with_one_ransac_model(gm, oc):
    from geometric_mapping_amd import _lib
    n = 1_000_000
    r = synth.fixed_k_radius(n)
    xyz = synth.tunnel_frame(n, seed=0)
    flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_CYLINDER | _lib.GM_CFG_KEEP_COUNTS
    with gm.GeometricMapping(neighborRadius=r, flags=flags, ransac_hypotheses=1024, ransac_threshold=TAU, ransac_seed=1,
                             max_points=n) as c:
        res = c.process_frame(xyz)
        cloud, rows = c.cropped_cloud()
        nrm = c.normals()
        lab = c.labels()
        cnt = c.neighbor_counts()
    inside = np.all(np.abs(xyz) <= 5.0, axis=1)
    assert res["n_cropped"] == int(inside.sum()) and res["n_valid"] == len(cloud) == len(lab)
    assert np.all(np.diff(rows.astype(np.int64)) > 0) and np.array_equal(xyz[rows], cloud)
    # the one model: labels re-derived by the oracle's O(n) pass over the same primitive, bit for bit
    labels = np.zeros(len(cloud), np.uint8)
    assert oc.label_cylinder(cloud, labels, 0, 2, res["cylinder"], TAU) == res["cylinder_inliers"]
    assert np.array_equal(lab, labels)
    assert abs(res["cylinder"][6] - 2.0) < 0.05 and ang(res["cylinder"][3:6], [1, 0, 0]) < 0.05   # analytic truth
    assert ang(res["cylinder_axis_refit"], oc.refit_axis(oc.segment_moments(cloud, nrm, lab, 2))) < 1e-8
    assert res["cylinder_inliers"] > 0.5 * len(cloud)
    # the reference-faithful half of the frame against the oracle itself at full size (16 threads: ~1 s)
    o = oc.process_frame(xyz, 5.0, r, 0.5, 0.2, oc.F64, nthreads=16)
    assert res["n_valid"] == o["n_valid"] and res["n_voxels"] == o["n_voxels"]
    _, o_cnt = oc.normals(xyz[inside], r, oc.F64, nthreads=16)
    assert np.array_equal(cnt, o_cnt)                                                             # 833 k neighbour sets
    a = np.arcsin(np.clip(np.linalg.norm(np.cross(nrm[:, :3].astype(np.float64), o["normals"][:, :3].astype(np.float64)), axis=1), 0, 1))
    assert np.quantile(a, 0.999) < 1e-5 and a.max() < 1e-4
    c_rel = np.abs(nrm[:, 3].astype(np.float64) - o["normals"][:, 3]) / o["normals"][:, 3]
    assert np.quantile(c_rel, 0.999) < 1e-4
    M = o["M"]
    assert np.abs(res["scatter"] - M).max() / np.abs(M).max() < 1e-5
    assert ang(res["center_axis"], o["evecs"][:, 0]) < 1e-5
    for k in (1, 2):
        assert abs(res["eigenvalues"][k] - o["evals"][k]) / o["evals"][k] < 1e-5


def test_config4_hundred_1m_frames_streamed_over_three_slots(gm):
    """100 distinct 1 M-point frames through gm_submit_frame / gm_wait_frame, three in flight: every result is bit-equal
    to the blocking call on the same frame (stable sort + fixed-order reductions: no dependence on what else runs)."""
    n, n_frames, slots = 1_000_000, 100, 3
    r = synth.fixed_k_radius(n)
    frames = [synth.tunnel_frame(n, seed=100 + i) for i in range(8)]     # 8 distinct clouds ...
    def frame(i):                                                        # ... x a per-frame rigid shift: 100 distinct frames
        f = frames[i % 8].copy()
        f[:, 0] += np.float32(0.01 * (i // 8))
        return f
    with gm.GeometricMapping(neighborRadius=r, n_slots=slots, max_points=n) as c:
        ref = []
        for i in range(n_frames):
            x = c.process_frame(frame(i))
            ref.append((x["n_cropped"], x["n_valid"], x["n_voxels"], x["scatter6"].copy(), x["eigenvectors"].copy()))
        got = [None] * n_frames
        inflight = []
        keep = {}
        for i in range(n_frames):
            s = i % slots
            if len(inflight) == slots:
                j = inflight.pop(0)
                got[j] = c.wait_frame(j % slots)
            keep[s] = frame(i)                                           # the host rows may be reused once submit returns
            c.submit_frame(s, keep[s])
            inflight.append(i)
        for j in inflight:
            got[j] = c.wait_frame(j % slots)
    assert len({tuple(x[3]) for x in ref}) == n_frames                   # the frames really are distinct
    for i in range(n_frames):
        g, x = got[i], ref[i]
        assert (g["n_cropped"], g["n_valid"], g["n_voxels"]) == x[:3], i
        assert np.array_equal(g["scatter6"], x[3]) and np.array_equal(g["eigenvectors"], x[4]), i


def test_config3_10m_frame_over_four_ranks(gm):
    """One 10 M-point frame cut into 4 x-slabs + halo, every rank the unchanged pipeline with gm_set_owned_range, the
    records all-gathered and merged (gm_group; the four ranks share this box's one GPU, so the gather runs over device
    copies -- tests/test_group.py covers the RCCL transport with one rank)."""
    from geometric_mapping_amd import _lib
    n = 10_000_000
    r = synth.fixed_k_radius(n)
    xyz = synth.tunnel_frame(n, seed=3, floor_z=-1.2, outlier_frac=0.01)
    kw = dict(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT | _lib.GM_CFG_RANSAC_PLANE | _lib.GM_CFG_RANSAC_CYLINDER,
              ransac_hypotheses=1024, ransac_threshold=TAU, ransac_seed=5)
    with gm.GeometricMapping(max_points=n, **kw) as c:
        ref = c.process_frame(xyz)
        with gm.GeometricMappingGroup([0, 0, 0, 0], loopback=True, **kw) as g:
            res = g.process_frame(xyz)
            cloud, rows = g.cropped_cloud()
        for k in ("n_in", "n_cropped", "n_valid", "n_voxels"):
            assert res[k] == ref[k], k
        assert np.all(np.diff(rows.astype(np.int64)) > 0) and np.array_equal(xyz[rows], cloud)
        assert np.abs(res["scatter6"] - ref["scatter6"]).max() / np.abs(ref["scatter6"]).max() < 1e-6
        assert ang(res["center_axis"], ref["center_axis"]) < 1e-5
        for k in (1, 2):
            assert abs(res["eigenvalues"][k] - ref["eigenvalues"][k]) / ref["eigenvalues"][k] < 1e-5
        # merged vote: the winners' global counts are their counts on the unsharded frame's valid cloud
        assert res["plane_inliers"] == c.score_frame(0, res["plane"][None, :], TAU)[0]
        assert res["cylinder_inliers"] == c.score_frame(1, res["cylinder"][None, :], TAU)[0]
    assert abs(abs(res["plane"][2]) - 1) < 1e-2 and abs(abs(res["plane"][3]) - 1.2) < 2e-2          # floor z = -1.2
    assert abs(res["cylinder"][6] - 2.0) < 0.05 and ang(res["cylinder"][3:6], [1, 0, 0]) < 0.05


def test_bench_line_keeps_its_contract():
    """`python bench.py` as the driver runs it at N = 1 (a child process; fewer steps and a smaller frame so that the test
    takes seconds): ONE JSON line on stdout whose keys, types and cross-checks are the round contract's -- BASELINE.json's
    metric and unit, steps / warmup echoed, whole-job value = points x steps / time, `roofline` with achieved / peak / frac
    consistent, `cpu_baseline` from the oracle, no model keys in `config`."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, steps, warmup = 200_000, 6, 2
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", str(steps), "--warmup", str(warmup), "--points", str(n),
                          "--no-secondary", "--group-points", "0"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                   # ONE line
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "points/s"   # (BASELINE's string goes on ", 1/2/4/8 GPU")
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"].startswith("synthetic")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]      # whole-job points per second
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 8000.0
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == d["unit"] and cb["sample"]
    assert d["value"] > cb["value"]                                  # (a GPU that loses to the CPU oracle is a broken build)
