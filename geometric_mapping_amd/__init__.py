"""geometric_mapping_amd -- MI355X (gfx950) implementation of the per-frame
point-cloud path of the geometric_mapping ROS node, behind a C ABI
(include/gm_hip.h, geometric_mapping_amd/libgm_hip.so).

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI),
_lib.py (ctypes binding), api.py (host mirror of the reference's functions),
synth.py (synthetic frames), sharding.py (multi-GPU slabs).
"""
from . import synth  # noqa: F401


def load_library():
    """dlopen libgm_hip.so (raises if it is missing -- there is no fallback)."""
    from . import _lib
    return _lib.load()


def __getattr__(name):
    if name in ("GeometricMapping", "GeometricMappingGroup", "GmError", "solve_local_frame", "decode_compressed_map"):
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
