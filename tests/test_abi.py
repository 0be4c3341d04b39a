"""C-ABI checks that need no GPU: the library loads, exports every symbol
include/gm_hip.h declares, its struct layouts match the ctypes mirror, and it fails
loudly (never falls back) when no device is visible."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from geometric_mapping_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 29
    for n in names:
        assert hasattr(L, n), n
    assert set(names) == set(L._gm_proto), "ctypes prototypes and header drifted apart"
    assert L.gm_abi_version() == 3


def test_header_is_plain_c_and_struct_layouts_match_ctypes():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "gm_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(gm_config), sizeof(gm_cloud), sizeof(gm_frame_result), sizeof(gm_map_header), sizeof(gm_map_primitive));
  printf("%zu %zu %zu %zu\n", offsetof(gm_config, boxFilterBound), offsetof(gm_config, ransac_seed), offsetof(gm_frame_result, scatter), offsetof(gm_frame_result, normals_kernel_ms));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    sizes = list(map(int, out))
    assert sizes[0] == C.sizeof(_lib.Config) and sizes[1] == C.sizeof(_lib.Cloud) and sizes[2] == C.sizeof(_lib.FrameResult)
    assert sizes[3] == 56 and sizes[4] == 40
    assert sizes[5] == _lib.Config.boxFilterBound.offset and sizes[6] == _lib.Config.ransac_seed.offset
    assert sizes[7] == _lib.FrameResult.scatter.offset and sizes[8] == _lib.FrameResult.normals_kernel_ms.offset


def test_default_config_is_the_launch_file():
    L = _lib.load()
    cfg = _lib.Config()
    L.gm_default_config(C.byref(cfg))
    # /root/reference launch/mapping.launch:7-10
    assert (cfg.boxFilterBound, cfg.voxelGridLeafSize, cfg.neighborRadius, cfg.weightingFactor) == (5.0, 0.5, 0.5, 0.2)
    assert cfg.struct_size == C.sizeof(_lib.Config) and cfg.flags == _lib.GM_CFG_VOXEL_GRID
    assert L.gm_status_string(0) == b"ok" and L.gm_status_string(3) == b"device error"


def test_create_rejects_bad_arguments_and_never_falls_back():
    L = _lib.load()
    ctx = C.c_void_p()
    cfg = _lib.Config()
    L.gm_default_config(C.byref(cfg))
    cfg.struct_size = 4
    assert L.gm_create(C.byref(cfg), C.byref(ctx)) == _lib.GM_ERR_INVALID_ARG
    L.gm_default_config(C.byref(cfg))
    cfg.voxelGridLeafSize = 0.0
    assert L.gm_create(C.byref(cfg), C.byref(ctx)) == _lib.GM_ERR_INVALID_ARG
    assert b"numeric" in L.gm_last_error(None)
    for h, tau in ((0, 0.03), (8193, 0.03), (64, 0.0), (64, float("nan"))):   # RANSAC parameters are checked at creation
        L.gm_default_config(C.byref(cfg))
        cfg.flags |= _lib.GM_CFG_RANSAC_CYLINDER
        cfg.ransac_hypotheses, cfg.ransac_threshold = h, tau
        assert L.gm_create(C.byref(cfg), C.byref(ctx)) == _lib.GM_ERR_INVALID_ARG
        assert b"ransac" in L.gm_last_error(None)
    L.gm_default_config(C.byref(cfg))
    st = L.gm_create(C.byref(cfg), C.byref(ctx))
    if st == _lib.GM_OK:          # running on a GPU box
        L.gm_destroy(ctx)
    else:                          # no device: a loud error, no CPU path behind the ABI
        assert st == _lib.GM_ERR_DEVICE and not ctx.value
        assert b"no CPU fallback" in L.gm_last_error(None) or b"gfx950" in L.gm_last_error(None)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "geometric_mapping_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "gm_oracle.h" not in text, f
                assert "libgm_oracle" not in text, f


def test_solve_local_frame_host_entry(oc):
    import geometric_mapping_amd as g
    rng = np.random.default_rng(0)
    A = rng.normal(size=(50, 3))
    M = A.T @ A
    sc = np.array([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]])
    ev, V = g.solve_local_frame(sc)
    w, W = oc.eig3(M)
    assert np.allclose(ev, w, rtol=1e-6) and np.all(np.diff(ev) >= 0)
    assert np.allclose(np.abs((V * W).sum(axis=0)), 1.0, atol=1e-5)


def test_solve_local_frame_has_the_eigenvector_signs_of_the_f32_faithful_solve():
    """Eigenvector signs are arbitrary mathematically but visible on /eigenBasisOutput (arrow directions).  The library
    returns fp64-Jacobi eigenpairs whose column signs are those of Eigen's float tridiagonal-QR sequence
    (SelfAdjointEigenSolver<MatrixXf>, /root/reference src/tunnel_processing.cpp:129) as the oracle restates it."""
    import glob
    import geometric_mapping_amd as g
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        d = np.load(p)
        M = d["M_f32"]
        ev, V = g.solve_local_frame(np.array([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]]))
        dots = (V.astype(np.float64) * d["evecs_f32"].astype(np.float64)).sum(axis=0)
        assert np.all(dots > 0.9999), (p, dots)
