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


C_EXE = os.path.join(ROOT, "examples", "gm_minimal")


def _build_c_example():
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "gm_minimal.c"), "-L", os.path.join(ROOT, "geometric_mapping_amd"),
                    "-lgm_hip", "-Wl,-rpath," + os.path.join(ROOT, "geometric_mapping_amd"), "-Wl,-rpath,/opt/rocm/lib",
                    "-lm", "-o", C_EXE], check=True, capture_output=True)


def test_plain_c_example_links_against_the_abi():
    """The boundary is usable from C99 with nothing but include/gm_hip.h and the shared library."""
    _build_c_example()
    assert os.path.exists(C_EXE)


@pytest.mark.gpu
def test_plain_c_example_runs_on_gpu():
    _build_c_example()
    r = subprocess.run([C_EXE, "60000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gm_minimal ok" in r.stdout
