"""ctypes binding of oracle/libgm_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by geometric_mapping_amd (the product).
PARITY UNPINNED (see gm_oracle.h): no reference fixture exists for this path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

F64 = 0
F32_FAITHFUL = 1
F32_SHIFTED = 2   # PCL >= 1.10 covariance (offsets from the neighbourhood's first point), otherwise as F32_FAITHFUL

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgm_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("gm_oracle.c", "gm_oracle_ext.c", "gm_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "libgm_oracle.so"], check=True, capture_output=True)
    return _SO


class FrameResult(C.Structure):
    _fields_ = [("n_in", C.c_int), ("n_cropped", C.c_int), ("n_valid", C.c_int), ("n_voxels", C.c_int),
                ("evals", C.c_float * 3), ("evecs", C.c_float * 9), ("M", C.c_double * 9)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        # GM_ORACLE_SO: load another build of the same sources (the ASan/UBSan one: `make -C oracle
        # libgm_oracle_asan.so`, run the CPU tests with LD_PRELOAD=$(gcc -print-file-name=libasan.so))
        L = C.CDLL(os.environ.get("GM_ORACLE_SO", _SO))
        fp, ip, dp, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_uint8))
        L.gmo_crop_box.argtypes = [fp, C.c_int, C.c_double, ip]; L.gmo_crop_box.restype = C.c_int
        L.gmo_normals.argtypes = [fp, C.c_int, C.c_double, C.c_int, C.c_int, fp, ip]; L.gmo_normals.restype = C.c_int
        L.gmo_finite_normals.argtypes = [fp, C.c_int, ip]; L.gmo_finite_normals.restype = C.c_int
        L.gmo_voxel_grid.argtypes = [fp, C.c_int, C.c_double, C.c_int, fp, ip, ip, ip]; L.gmo_voxel_grid.restype = C.c_int
        L.gmo_local_frame.argtypes = [fp, C.c_int, C.c_double, C.c_int, dp, fp, fp]; L.gmo_local_frame.restype = None
        L.gmo_nearest.argtypes = [fp, C.c_int, fp, C.c_int, ip]; L.gmo_nearest.restype = None
        L.gmo_process_frame.argtypes = [fp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                        fp, fp, fp, C.POINTER(FrameResult)]
        L.gmo_process_frame.restype = C.c_int
        L.gmo_mix64.argtypes = [C.c_uint64]; L.gmo_mix64.restype = C.c_uint64
        L.gmo_plane_hypotheses.argtypes = [fp, C.c_int, u8p, C.c_int, C.c_uint64, C.c_int, fp]; L.gmo_plane_hypotheses.restype = None
        L.gmo_cylinder_hypotheses.argtypes = [fp, fp, C.c_int, u8p, C.c_int, C.c_uint64, C.c_int, fp]; L.gmo_cylinder_hypotheses.restype = None
        L.gmo_score_planes.argtypes = [fp, C.c_int, u8p, C.c_int, fp, C.c_int, C.c_double, ip]; L.gmo_score_planes.restype = None
        L.gmo_score_cylinders.argtypes = [fp, C.c_int, u8p, C.c_int, fp, C.c_int, C.c_double, ip]; L.gmo_score_cylinders.restype = None
        L.gmo_label_plane.argtypes = [fp, C.c_int, u8p, C.c_int, C.c_int, fp, C.c_double]; L.gmo_label_plane.restype = C.c_int
        L.gmo_label_cylinder.argtypes = [fp, C.c_int, u8p, C.c_int, C.c_int, fp, C.c_double]; L.gmo_label_cylinder.restype = C.c_int
        L.gmo_segment_moments.argtypes = [fp, fp, C.c_int, u8p, C.c_int, dp]; L.gmo_segment_moments.restype = None
        L.gmo_refit_plane.argtypes = [dp, dp]; L.gmo_refit_plane.restype = None
        L.gmo_refit_axis.argtypes = [dp, dp]; L.gmo_refit_axis.restype = None
        L.gmo_eig3.argtypes = [dp, dp, dp]; L.gmo_eig3.restype = None
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def _xyz(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def crop_box(xyz, bound):
    xyz = _xyz(xyz)
    idx = np.empty(max(len(xyz), 1), dtype=np.int32)
    m = lib().gmo_crop_box(_f(xyz), len(xyz), float(bound), _i(idx))
    return idx[:m].copy()


def normals(xyz, radius, mode=F64, nthreads=1):
    xyz = _xyz(xyz)
    out = np.empty((max(len(xyz), 1), 4), dtype=np.float32)
    cnt = np.zeros(max(len(xyz), 1), dtype=np.int32)
    r = lib().gmo_normals(_f(xyz), len(xyz), float(radius), mode, nthreads, _f(out), _i(cnt))
    if r < 0:
        raise MemoryError("gmo_normals failed")
    return out[:len(xyz)], cnt[:len(xyz)]


def finite_normals(nrm):
    nrm = np.ascontiguousarray(nrm, dtype=np.float32)
    idx = np.empty(max(len(nrm), 1), dtype=np.int32)
    m = lib().gmo_finite_normals(_f(nrm), len(nrm), _i(idx))
    return idx[:m].copy()


def voxel_grid(xyz, leaf, mode=F64):
    xyz = _xyz(xyz)
    n = len(xyz)
    out = np.empty((max(n, 1), 3), dtype=np.float32)
    key = np.empty(max(n, 1), dtype=np.int32)
    cnt = np.empty(max(n, 1), dtype=np.int32)
    pt = C.c_int(0)
    V = lib().gmo_voxel_grid(_f(xyz), n, float(leaf), mode, _f(out), _i(key), _i(cnt), C.byref(pt))
    if V < 0:
        raise MemoryError("gmo_voxel_grid failed")
    return out[:V].copy(), key[:V].copy(), cnt[:V].copy(), bool(pt.value)


def local_frame(nrm, wf, mode=F64):
    nrm = np.ascontiguousarray(nrm, dtype=np.float32)
    M = np.zeros(9, dtype=np.float64)
    ev = np.zeros(3, dtype=np.float32)
    V = np.zeros(9, dtype=np.float32)
    lib().gmo_local_frame(_f(nrm), len(nrm), float(wf), mode, _d(M), _f(ev), _f(V))
    # column-major (Eigen::Matrix3f) -> numpy [row, col]
    return ev, V.reshape(3, 3).T.copy(), M.reshape(3, 3)


def nearest(xyz, queries):
    xyz = _xyz(xyz)
    q = _xyz(queries)
    idx = np.empty(max(len(q), 1), dtype=np.int32)
    lib().gmo_nearest(_f(xyz), len(xyz), _f(q), len(q), _i(idx))
    return idx[:len(q)].copy()


def process_frame(xyz, bound, radius, leaf, wf, mode=F64, nthreads=1, want_outputs=True):
    xyz = _xyz(xyz)
    n = len(xyz)
    res = FrameResult()
    if want_outputs:
        oc = np.empty((max(n, 1), 3), dtype=np.float32)
        on = np.empty((max(n, 1), 4), dtype=np.float32)
        ov = np.empty((max(n, 1), 3), dtype=np.float32)
        rc = lib().gmo_process_frame(_f(xyz), n, bound, radius, leaf, wf, mode, nthreads, _f(oc), _f(on), _f(ov), C.byref(res))
    else:
        oc = on = ov = None
        rc = lib().gmo_process_frame(_f(xyz), n, bound, radius, leaf, wf, mode, nthreads, None, None, None, C.byref(res))
    if rc != 0:
        raise MemoryError("gmo_process_frame failed")
    out = dict(n_in=res.n_in, n_cropped=res.n_cropped, n_valid=res.n_valid, n_voxels=res.n_voxels,
               evals=np.array(res.evals[:], dtype=np.float32),
               evecs=np.array(res.evecs[:], dtype=np.float32).reshape(3, 3).T.copy(),
               M=np.array(res.M[:], dtype=np.float64).reshape(3, 3))
    if want_outputs:
        out["xyz"] = oc[:res.n_valid].copy()
        out["normals"] = on[:res.n_valid].copy()
        out["voxels"] = ov[:res.n_voxels].copy()
    return out


# ---- extensions ----

def plane_hypotheses(xyz, seed, H, labels=None, want=0):
    xyz = _xyz(xyz)
    out = np.empty((H, 4), dtype=np.float32)
    if labels is not None:
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
    lib().gmo_plane_hypotheses(_f(xyz), len(xyz), _u8(labels), want, C.c_uint64(seed), H, _f(out))
    return out


def cylinder_hypotheses(xyz, nrm, seed, H, labels=None, want=0):
    xyz = _xyz(xyz)
    nrm = np.ascontiguousarray(nrm, dtype=np.float32)
    out = np.empty((H, 7), dtype=np.float32)
    if labels is not None:
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
    lib().gmo_cylinder_hypotheses(_f(xyz), _f(nrm), len(xyz), _u8(labels), want, C.c_uint64(seed), H, _f(out))
    return out


def score_planes(xyz, hyp, tau, mask=None, want=0):
    xyz = _xyz(xyz)
    hyp = np.ascontiguousarray(hyp, dtype=np.float32)
    cnt = np.zeros(len(hyp), dtype=np.int32)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
    lib().gmo_score_planes(_f(xyz), len(xyz), _u8(mask), want, _f(hyp), len(hyp), float(tau), _i(cnt))
    return cnt


def score_cylinders(xyz, hyp, tau, mask=None, want=0):
    xyz = _xyz(xyz)
    hyp = np.ascontiguousarray(hyp, dtype=np.float32)
    cnt = np.zeros(len(hyp), dtype=np.int32)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
    lib().gmo_score_cylinders(_f(xyz), len(xyz), _u8(mask), want, _f(hyp), len(hyp), float(tau), _i(cnt))
    return cnt


def label_plane(xyz, labels, want, label, hyp4, tau):
    xyz = _xyz(xyz)
    hyp4 = np.ascontiguousarray(hyp4, dtype=np.float32)
    return lib().gmo_label_plane(_f(xyz), len(xyz), _u8(labels), want, label, _f(hyp4), float(tau))


def label_cylinder(xyz, labels, want, label, hyp7, tau):
    xyz = _xyz(xyz)
    hyp7 = np.ascontiguousarray(hyp7, dtype=np.float32)
    return lib().gmo_label_cylinder(_f(xyz), len(xyz), _u8(labels), want, label, _f(hyp7), float(tau))


def segment_moments(xyz, nrm, labels, label):
    xyz = _xyz(xyz)
    m = np.zeros(16, dtype=np.float64)
    nrm_c = np.ascontiguousarray(nrm, dtype=np.float32) if nrm is not None else None
    if labels is not None:
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
    lib().gmo_segment_moments(_f(xyz), _f(nrm_c) if nrm_c is not None else None, len(xyz), _u8(labels), label, _d(m))
    return m


def refit_plane(mom):
    out = np.zeros(4)
    lib().gmo_refit_plane(_d(np.ascontiguousarray(mom, dtype=np.float64)), _d(out))
    return out


def refit_axis(mom):
    out = np.zeros(3)
    lib().gmo_refit_axis(_d(np.ascontiguousarray(mom, dtype=np.float64)), _d(out))
    return out


def eig3(A):
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(9)
    w = np.zeros(3)
    V = np.zeros(9)
    lib().gmo_eig3(_d(A), _d(w), _d(V))
    return w, V.reshape(3, 3).T.copy()
