"""ctypes binding of libgm_hip.so -- the C ABI declared in include/gm_hip.h.

There is deliberately NO fallback here: if the library is missing or does not
load, importing callers get an exception.  Nothing in this package imports
oracle/ (the CPU restatement is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgm_hip.so")
if os.environ.get("GM_LIB_PATH"):   # experiments only (tools/build_variants.sh): another build of the same library
    LIB_PATH = os.environ["GM_LIB_PATH"]
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gm_hip.h")

GM_OK = 0
GM_ERR_INVALID_ARG = 1
GM_ERR_TOO_FEW_POINTS = 2
GM_ERR_DEVICE = 3
GM_ERR_OOM = 4
GM_ERR_CAPACITY = 5
GM_ERR_NOT_READY = 6
GM_ERR_UNSUPPORTED = 7
GM_ERR_COMM = 8
GM_GROUP_LOOPBACK = 1
GM_GROUP_N_TIMINGS = 5

GM_CFG_VOXEL_GRID = 1 << 0
GM_CFG_NEAREST = 1 << 1
GM_CFG_RANSAC_PLANE = 1 << 2
GM_CFG_RANSAC_CYLINDER = 1 << 3
GM_CFG_STAGE_TIMING = 1 << 4
GM_CFG_KEEP_COUNTS = 1 << 5
GM_CFG_GRAPH = 1 << 6
GM_CFG_DEFAULT = GM_CFG_VOXEL_GRID

GM_CLOUD_DEVICE = 1 << 0
GM_CLOUD_BIGENDIAN = 1 << 1
GM_CLOUD_PINNED = 1 << 2

GM_RES_VOXEL_PASSTHROUGH = 1 << 0

GM_N_STAGES = 9
STAGE_NAMES = ("upload", "crop", "grid", "normals", "compact", "frame", "voxel", "ransac", "total")


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32),
                ("boxFilterBound", C.c_double), ("voxelGridLeafSize", C.c_double),
                ("neighborRadius", C.c_double), ("weightingFactor", C.c_double),
                ("device", C.c_int32), ("n_slots", C.c_uint32), ("max_points", C.c_uint32),
                ("ransac_hypotheses", C.c_uint32), ("ransac_threshold", C.c_double),
                ("ransac_seed", C.c_uint64)]


class Cloud(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n_points", C.c_uint32), ("point_step", C.c_uint32),
                ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32), ("flags", C.c_uint32)]


class FrameResult(C.Structure):
    _fields_ = [("n_in", C.c_uint32), ("n_cropped", C.c_uint32), ("n_valid", C.c_uint32), ("n_voxels", C.c_uint32),
                ("eigenvalues", C.c_float * 3), ("eigenvectors", C.c_float * 9), ("center_axis", C.c_float * 3),
                ("status_flags", C.c_uint32), ("scatter", C.c_double * 6),
                ("plane_inliers", C.c_uint32), ("cylinder_inliers", C.c_uint32),
                ("plane", C.c_float * 4), ("cylinder", C.c_float * 7),
                ("plane_refit", C.c_double * 4), ("cylinder_axis_refit", C.c_double * 3),
                ("stage_ms", C.c_float * GM_N_STAGES), ("normals_kernel_ms", C.c_float)]


class GmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libgm_hip: status {status}: {message}")
        self.status = status


_lib = None


def declared_symbols():
    """Every function name include/gm_hip.h declares (used by the ABI test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", text)))


def _preload_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own
    libamdhip64.so (same SONAME as /opt/rocm's).  If libgm_hip.so pulled in the
    system copy first, a later `import torch` would load a SECOND runtime and see
    no GPU.  So when torch is installed, its copy is loaded first and libgm_hip
    binds to it by SONAME; a C++ host without torch simply uses /opt/rocm's."""
    if os.environ.get("GM_HIP_SYSTEM_RUNTIME") == "1":
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return None
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    # the same for RCCL (gm_group_*): torch's bundled librccl.so shares no SONAME with /opt/rocm's, so a process could
    # map both; libgm_hip.so dlopens the one named here (csrc/gm_group.hip)
    rccl = os.path.join(libdir, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("GM_RCCL_PATH", rccl)
    path = os.path.join(libdir, "libamdhip64.so")
    if not os.path.exists(path):
        return None
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def load():
    """dlopen libgm_hip.so and attach prototypes.  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C geometric_mapping_amd/csrc).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32, u32p, fp, dp, i32p, u8p = (C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                         C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8))
    cfgp, cloudp, resp = C.POINTER(Config), C.POINTER(Cloud), C.POINTER(FrameResult)
    proto = {
        "gm_create": (C.c_int, [cfgp, C.POINTER(vp)]),
        "gm_destroy": (None, [vp]),
        "gm_default_config": (None, [cfgp]),
        "gm_host_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "gm_host_free": (C.c_int, [vp, vp]),
        "gm_host_register": (C.c_int, [vp, vp, C.c_size_t]),
        "gm_host_unregister": (C.c_int, [vp, vp]),
        "gm_abi_version": (u32, []),
        "gm_status_string": (C.c_char_p, [C.c_int]),
        "gm_last_error": (C.c_char_p, [vp]),
        "gm_process_frame": (C.c_int, [vp, cloudp, resp]),
        "gm_submit_frame": (C.c_int, [vp, u32, cloudp]),
        "gm_wait_frame": (C.c_int, [vp, u32, resp]),
        "gm_poll_frame": (C.c_int, [vp, u32]),
        "gm_set_cloud_output": (C.c_int, [vp, u32, fp, u32]),
        "gm_get_cropped_xyz": (C.c_int, [vp, u32, fp, u32, u32p]),
        "gm_get_normals": (C.c_int, [vp, u32, fp, u32, u32p]),
        "gm_get_voxel_centroids": (C.c_int, [vp, u32, fp, u32, u32p]),
        "gm_get_voxel_nearest": (C.c_int, [vp, u32, i32p, u32, u32p]),
        "gm_get_voxel_normals": (C.c_int, [vp, u32, fp, u32, u32p]),
        "gm_get_neighbor_counts": (C.c_int, [vp, u32, i32p, u32, u32p]),
        "gm_get_labels": (C.c_int, [vp, u32, u8p, u32, u32p]),
        "gm_chop_cloud": (C.c_int, [vp, cloudp, C.c_double, fp, u32, u32p]),
        "gm_get_normals_stage": (C.c_int, [vp, fp, u32, C.c_double, fp, fp, u32, u32p]),
        "gm_get_local_frame": (C.c_int, [vp, fp, u32, C.c_double, fp, fp, dp]),
        "gm_voxel_grid": (C.c_int, [vp, fp, u32, C.c_double, fp, u32, u32p, u32p]),
        "gm_nearest": (C.c_int, [vp, fp, u32, fp, u32, i32p]),
        "gm_solve_local_frame": (C.c_int, [dp, fp, fp]),
        "gm_set_owned_range": (C.c_int, [vp, C.c_double, C.c_double]),
        "gm_ext_available": (C.c_int, []),
        "gm_score_frame": (C.c_int, [vp, u32, C.c_int, fp, u32, C.c_double, u32, i32p]),
        "gm_score_planes": (C.c_int, [vp, fp, u32, u8p, u32, fp, u32, C.c_double, i32p]),
        "gm_score_cylinders": (C.c_int, [vp, fp, u32, u8p, u32, fp, u32, C.c_double, i32p]),
        "gm_plane_hypotheses": (C.c_int, [vp, fp, u32, u8p, u32, C.c_uint64, u32, fp]),
        "gm_cylinder_hypotheses": (C.c_int, [vp, fp, fp, u32, u8p, u32, C.c_uint64, u32, fp]),
        "gm_segment_moments": (C.c_int, [vp, fp, fp, u8p, u32, u32, dp]),
        "gm_get_compressed_map": (C.c_int, [vp, u32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "gm_group_create": (C.c_int, [cfgp, i32p, u32, u32, C.POINTER(vp)]),
        "gm_group_destroy": (None, [vp]),
        "gm_group_size": (u32, [vp]),
        "gm_group_ctx": (vp, [vp, u32]),
        "gm_group_last_error": (C.c_char_p, [vp]),
        "gm_group_process_frame": (C.c_int, [vp, cloudp, resp]),
        "gm_group_get_cropped_xyz": (C.c_int, [vp, fp, u32, u32p]),
        "gm_group_get_voxel_centroids": (C.c_int, [vp, fp, u32, u32p]),
        "gm_group_get_voxel_normals": (C.c_int, [vp, fp, u32, u32p]),
        "gm_group_get_voxel_nearest": (C.c_int, [vp, i32p, u32, u32p]),
        "gm_group_get_timing": (C.c_int, [vp, dp, u32]),
        "gm_group_get_edges": (C.c_int, [vp, dp, u32, u32p]),
        "gm_group_submit_frame": (C.c_int, [vp, cloudp]),
        "gm_group_wait_frame": (C.c_int, [vp, resp, u32p, u32p]),
        "gm_group_poll_frame": (C.c_int, [vp]),
        "gm_group_in_flight": (u32, [vp]),
    }
    for name, (res, args) in proto.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    L._gm_proto = proto
    _lib = L
    return L
