"""SYNTAX / TYPE CHECK ONLY: ros/geometric_mapping_node.cpp against minimal stand-in ROS headers (tests/stubs/).

ROS does not exist in this image, so the catkin node cannot be built or run here.  This test only proves the file is
valid C++11 against the message fields and roscpp calls it uses (`g++ -fsyntax-only`); it says nothing about run-time
behaviour.  tests/stubs/README.md describes what the stand-ins are (and are not)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ros_node_type_checks_against_stand_in_headers():
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=return-type",
                        "-I", os.path.join(ROOT, "tests", "stubs"), os.path.join(ROOT, "ros", "geometric_mapping_node.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ros_node_uses_the_frame_outputs_for_surface_normals():
    """displayNormals must not re-run voxel grid + 1-NN through stage calls (round-1 review): the context carries
    GM_CFG_NEAREST and the markers come from the frame's own centroids / nearest normals; the cloud is fetched once."""
    src = open(os.path.join(ROOT, "ros", "geometric_mapping_node.cpp")).read()
    assert "GM_CFG_NEAREST" in src and "rvizNormalsFromFrame" in src
    assert src.count("choppedCloud()") == 1
    assert "row_step" in src
