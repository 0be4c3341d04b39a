"""Marker formatting (SURVEY.md par. 8 f-1): the reference's rvizArrow / rvizNormals / rvizEigens
(/root/reference src/tunnel_processing.cpp:161-205, 225-256, 260-300) field for field.

Three statements of the same contract are held against each other:
  tests/golden/markers_*.json   committed fixtures (made by tests/golden/make_markers.py from the oracle's outputs)
  oracle/markers_np.py          numpy restatement (test infrastructure)
  host/gm_tunnel_processing.cpp the C++ host mirror the ROS node uses (the product), through host/gm_marker_dump
"Parity unpinned": the reference ships no marker fixtures and cannot run here."""
import glob
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "markers_*.json")))
DUMP = os.path.join(ROOT, "host", "gm_marker_dump")
IDS = [os.path.basename(p)[8:-5] for p in GOLD]


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True, capture_output=True)


def _hex(f):
    return "%08x" % struct.unpack("<I", struct.pack("<f", float(f)))[0]


def test_fixtures_exist():
    assert len(GOLD) == 4


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_oracle_restatement_reproduces_marker_fixtures(path, oc):
    from oracle import markers_np as mk
    fx = json.load(open(path))
    g = np.load(os.path.join(ROOT, "tests", "golden", fx["case"] + ".npz"))
    assert mk.rviz_eigens(g["evals_f32"], g["evecs_f32"]) == fx["eigenBasis"]
    xyz = g["xyz"][g["crop_rows"]][g["valid_rows"]]
    idx = oc.nearest(xyz, g["voxel_centroids"])
    assert [int(i) for i in idx[:12]] == fx["nearest_idx_head"]
    nm = mk.rviz_normals(g["voxel_centroids"], idx, g["normals_f64"])
    assert len(nm) == fx["n_normals"] and nm[:12] == fx["normals_head"] and nm[-2:] == fx["normals_tail"]


def test_marker_contract_known_answers():
    """The fields the reference hard-codes, independent of any data."""
    from oracle import markers_np as mk
    m = mk.rviz_arrow([1, 2, 3], [4, 5, 6], [0.1, 0.2, 0.3], [0.25, 0.5, 0.75, 1.0], "x")
    assert m["frame_id"] == "/velodyne" and m["id"] == 0 and m["type"] == 0 and m["action"] == 0 and m["seq"] == 0
    assert m["color"] == {"a": 0.25, "r": 0.5, "g": 0.75, "b": 1.0}          # Vector4f read as A,R,G,B (:199-202)
    e = mk.rviz_eigens([0.0, 3.0, 4.0], np.eye(3))
    assert [x["ns"] for x in e] == ["eigenBasis"] * 3 and [x["id"] for x in e] == [0, 1, 2]
    # scale = 0.1 - 0.05 * |lambda_i| / ||lambda||  (and 0.3 - 0.15.., 0.25 - 0.125..), :265, :274-278
    assert e[0]["scale"] == [float(np.float32(0.1)), float(np.float32(0.3)), float(np.float32(0.25))]
    assert abs(e[2]["scale"][0] - (0.1 - 0.05 * 0.8)) < 1e-7 and abs(e[1]["scale"][1] - (0.3 - 0.15 * 0.6)) < 1e-7
    assert [e[i]["color"] for i in range(3)] == [{"a": 1.0, "r": 1.0, "g": 0.0, "b": 0.0}, {"a": 1.0, "r": 0.0, "g": 1.0, "b": 0.0},
                                                 {"a": 1.0, "r": 0.0, "g": 0.0, "b": 1.0}]
    n = mk.rviz_normals(np.array([[1, 2, 3]], np.float32), [0], np.array([[0, 0, 1, 0.1]], np.float32))
    assert n[0]["points"] == [[1.0, 2.0, 3.0], [0.0, 0.0, 1.0]]              # the arrow END is the normal itself (:247-249)
    assert n[0]["scale"] == [float(np.float32(0.025)), float(np.float32(0.075)), float(np.float32(0.0625))]
    assert n[0]["color"] == {"a": 1.0, "r": 0.0, "g": 0.0, "b": 1.0} and n[0]["ns"] == "normals"


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_host_rviz_eigens_equals_fixture_field_for_field(path):
    """The C++ host mirror (no GPU needed: rvizEigens / rvizArrow are pure host formatting)."""
    _build()
    fx = json.load(open(path))
    vals = fx["eigen_inputs"]["vals"]
    V = np.array(fx["eigen_inputs"]["vecs_rowmajor"], np.float32)            # [row, col]; the host takes column-major
    args = [_hex(v) for v in vals] + [_hex(V[r, c]) for c in range(3) for r in range(3)]
    r = subprocess.run([DUMP, "eigens"] + args, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)["eigenBasis"]
    assert got == fx["eigenBasis"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_host_frame_markers_on_gpu_match_fixtures(path, tmp_path):
    """The whole callback through the C++ host on the GPU (GM_CFG_NEAREST: centroids + nearest normals come out of the
    frame itself): ids / ns / frame / scale / colour / order exact; start points = voxel centroids to 2e-6; end points =
    normals to 1e-5 rad (a voxel whose two nearest points are equidistant to fp32 rounding may pick the other one)."""
    _build()
    fx = json.load(open(path))
    g = np.load(os.path.join(ROOT, "tests", "golden", fx["case"] + ".npz"))
    b, rad, leaf, wf = g["params"]
    f = tmp_path / "xyz.f32"
    g["xyz"].astype(np.float32).tofile(f)
    r = subprocess.run([DUMP, "frame", str(f), str(len(g["xyz"])), repr(float(b)), repr(float(leaf)), repr(float(rad)), repr(float(wf))],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(r.stdout)
    assert got["n_voxels"] == fx["n_normals"] == len(got["normals"])
    want = fx["normals_head"] + fx["normals_tail"]
    have = got["normals"][:12] + got["normals"][-2:]
    bad_dir = 0
    for w, h in zip(want, have):
        for k in ("frame_id", "seq", "ns", "id", "type", "action", "scale", "color"):
            assert w[k] == h[k], k
        assert np.abs(np.array(w["points"][0]) - np.array(h["points"][0])).max() < 2e-6
        a, c = np.array(w["points"][1]), np.array(h["points"][1])
        if np.linalg.norm(np.cross(a, c)) > 1e-5:
            bad_dir += 1
    assert bad_dir <= 1
    # eigen basis: fixtures hold the f32-faithful oracle's eigenpairs; the GPU's agree to the frame tolerances
    for w, h in zip(fx["eigenBasis"], got["eigenBasis"]):
        for k in ("frame_id", "seq", "ns", "id", "type", "action", "color"):
            assert w[k] == h[k], k
        assert np.abs(np.array(w["scale"]) - np.array(h["scale"])).max() < 1e-5
        # arrow direction INCLUDING its sign (the library takes the column signs of Eigen's float solve); the two
        # cross-section eigenvectors of a round tunnel are only defined up to the lambda1/lambda2 gap
        gap = min(abs(fx["eigen_inputs"]["vals"][1] - fx["eigen_inputs"]["vals"][0]), abs(fx["eigen_inputs"]["vals"][2] - fx["eigen_inputs"]["vals"][1])) / fx["eigen_inputs"]["vals"][2]
        assert np.abs(np.array(w["points"][1]) - np.array(h["points"][1])).max() < 2e-5 / max(gap, 1e-3)
