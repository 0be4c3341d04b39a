"""The C++ host mirror (host/, plain g++ -- the reference's own language) over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "gm_host_test")


def test_host_mirror_builds_with_plain_gxx():
    subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True, capture_output=True)
    assert os.path.exists(EXE)


def test_host_mirror_keeps_reference_function_names():
    hdr = open(os.path.join(ROOT, "host", "gm_tunnel_processing.hpp")).read()
    for name in ("chopCloud", "getNormals", "getLocalFrame", "rvizArrow", "rvizNormals", "rvizEigens"):
        assert name in hdr, name
    node = open(os.path.join(ROOT, "ros", "geometric_mapping_node.cpp")).read()
    for s in ('"input", 1', '"cloudOutput", 10', '"normalsOutput", 10', '"eigenBasisOutput", 10', '"centerAxisOutput", 10',
              "displayCylinder", "geometric_mapping_node",
              "boxFilterBound", "voxelGridLeafSize", "neighborRadius", "weightingFactor", "displayCloud", "displayNormals",
              "displayCenterAxis", "usePCLViz"):
        assert s in node, s


@pytest.mark.gpu
def test_host_harness_runs_cloud_cb_sequence_on_gpu():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True, capture_output=True)
    r = subprocess.run([EXE, "50000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gm_host_test ok" in r.stdout
