"""Host-side mirror of the reference's processing interface, over the C ABI.

The reference's path is four C++ free functions
(/root/reference include/geometric_mapping/tunnel_processing.hpp:38-54,77-82)
driven by cloud_cb (/root/reference src/geometric_mapping.cpp:48-125).  This
module keeps their names, argument order and meaning so the parity tests read
like the reference's call sites; every method is a thin ctypes call into
libgm_hip.so (include/gm_hip.h) -- no arithmetic happens in Python.

Clouds are numpy float32 arrays [n,3]; normals are [n,4] = (nx,ny,nz,curvature).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (GM_CFG_DEFAULT, GM_CFG_KEEP_COUNTS, GM_CFG_STAGE_TIMING, GM_CFG_VOXEL_GRID, GM_CLOUD_BIGENDIAN,
                   GM_CLOUD_DEVICE, GM_CLOUD_PINNED, GM_ERR_CAPACITY, GM_ERR_NOT_READY, GM_OK, Cloud, Config, FrameResult, GmError, STAGE_NAMES)

__all__ = ["GeometricMapping", "GeometricMappingGroup", "GmError", "solve_local_frame", "decode_compressed_map"]


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class GeometricMapping:
    """One gm_ctx.  Parameter names and defaults are the node's
    (/root/reference launch/mapping.launch:7-10, paramHandler.hpp:26-29)."""

    def __init__(self, boxFilterBound=5.0, voxelGridLeafSize=0.5, neighborRadius=0.5, weightingFactor=0.2,
                 device=0, flags=GM_CFG_DEFAULT, n_slots=1, max_points=0,
                 ransac_hypotheses=1024, ransac_threshold=0.03, ransac_seed=1):
        self._L = _lib.load()
        cfg = Config()
        self._L.gm_default_config(C.byref(cfg))
        cfg.flags = flags
        cfg.boxFilterBound = boxFilterBound
        cfg.voxelGridLeafSize = voxelGridLeafSize
        cfg.neighborRadius = neighborRadius
        cfg.weightingFactor = weightingFactor
        cfg.device = device
        cfg.n_slots = n_slots
        cfg.max_points = max_points
        cfg.ransac_hypotheses = ransac_hypotheses
        cfg.ransac_threshold = ransac_threshold
        cfg.ransac_seed = ransac_seed
        self.cfg = cfg
        self._ctx = C.c_void_p()
        self._pinned = []
        st = self._L.gm_create(C.byref(cfg), C.byref(self._ctx))
        if st != GM_OK:
            msg = self._L.gm_last_error(None).decode()
            self._ctx = None
            raise GmError(st, msg)
        self._keep = {}  # slot -> arrays that must outlive an async submit

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_ctx", None):
            for slot in getattr(self, "_cloud_slots", []):   # (waits for a frame that still writes the buffer)
                self._L.gm_set_cloud_output(self._ctx, slot, None, 0)
            self._cloud_slots = []
            for a in getattr(self, "_registered", []):
                self._L.gm_host_unregister(self._ctx, a.ctypes.data)
            self._registered = []
            for p in getattr(self, "_pinned", []):
                self._L.gm_host_free(self._ctx, p)
            self._pinned = []
            self._L.gm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != GM_OK:
            raise GmError(st, self._L.gm_last_error(self._ctx).decode())

    # ---- cloud descriptors ----
    @staticmethod
    def _cloud_from_xyz(xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if xyz.ndim != 2 or xyz.shape[1] not in (3, 4):
            raise ValueError("cloud must be [n,3] or [n,4] float32")
        step = 4 * xyz.shape[1]
        c = Cloud(xyz.ctypes.data, xyz.shape[0], step, 0, 4, 8, 0)
        return c, xyz

    @staticmethod
    def cloud_from_rows(data, n_points, point_step, offsets=(0, 4, 8), bigendian=False):
        """PointCloud2-style rows held in a numpy uint8 buffer."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if data.size < n_points * point_step:
            raise ValueError("row buffer smaller than n_points*point_step")
        c = Cloud(data.ctypes.data, n_points, point_step, offsets[0], offsets[1], offsets[2],
                  GM_CLOUD_BIGENDIAN if bigendian else 0)
        return c, data

    @staticmethod
    def cloud_from_device(ptr, n_points, point_step=16, offsets=(0, 4, 8)):
        """Rows already resident in this device's HBM (e.g. a torch tensor's data_ptr())."""
        return Cloud(int(ptr), n_points, point_step, offsets[0], offsets[1], offsets[2], GM_CLOUD_DEVICE), None

    def pinned_rows(self, n_points, point_step=12):
        """A page-locked row buffer (numpy uint8 view of gm_host_alloc memory) and a function that wraps it as a
        GM_CLOUD_PINNED cloud: frames submitted from it skip the staging copy.  Freed with the context."""
        ptr = C.c_void_p()
        nbytes = int(n_points) * int(point_step)
        self._check(self._L.gm_host_alloc(self._ctx, nbytes, C.byref(ptr)))
        self._pinned.append(ptr)
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))[:nbytes]

        def as_cloud(n=n_points, offsets=(0, 4, 8), bigendian=False):
            return Cloud(ptr.value, int(n), int(point_step), offsets[0], offsets[1], offsets[2],
                         GM_CLOUD_PINNED | (GM_CLOUD_BIGENDIAN if bigendian else 0)), buf
        return buf, as_cloud

    def _as_cloud(self, cloud):
        if isinstance(cloud, tuple) and isinstance(cloud[0], Cloud):
            return cloud
        return self._cloud_from_xyz(cloud)

    @staticmethod
    def _result(res):
        ev = np.array(res.eigenvalues[:], dtype=np.float32)
        V = np.array(res.eigenvectors[:], dtype=np.float32).reshape(3, 3).T.copy()  # column-major -> [row, col]
        sc = np.array(res.scatter[:], dtype=np.float64)
        M = np.array([[sc[0], sc[1], sc[2]], [sc[1], sc[3], sc[4]], [sc[2], sc[4], sc[5]]])
        return dict(n_in=res.n_in, n_cropped=res.n_cropped, n_valid=res.n_valid, n_voxels=res.n_voxels,
                    eigenvalues=ev, eigenvectors=V, center_axis=np.array(res.center_axis[:], dtype=np.float32),
                    scatter=M, scatter6=sc, status_flags=res.status_flags,
                    stage_ms={k: float(res.stage_ms[i]) for i, k in enumerate(STAGE_NAMES)},
                    normals_kernel_ms=float(res.normals_kernel_ms),
                    plane=np.array(res.plane[:], dtype=np.float32), cylinder=np.array(res.cylinder[:], dtype=np.float32),
                    plane_inliers=res.plane_inliers, cylinder_inliers=res.cylinder_inliers,
                    plane_refit=np.array(res.plane_refit[:]), cylinder_axis_refit=np.array(res.cylinder_axis_refit[:]))

    # ---- the callback: src/geometric_mapping.cpp:55-92 ----
    def process_frame(self, cloud):
        c, keep = self._as_cloud(cloud)
        res = FrameResult()
        self._check(self._L.gm_process_frame(self._ctx, C.byref(c), C.byref(res)))
        return self._result(res)

    def submit_frame(self, slot, cloud):
        c, keep = self._as_cloud(cloud)
        self._check(self._L.gm_submit_frame(self._ctx, slot, C.byref(c)))

    def wait_frame(self, slot):
        res = FrameResult()
        self._check(self._L.gm_wait_frame(self._ctx, slot, C.byref(res)))
        return self._result(res)

    def poll_frame(self, slot):
        """True when the slot's submitted frame has finished (wait_frame returns at once); never blocks."""
        st = self._L.gm_poll_frame(self._ctx, slot)
        if st == GM_OK:
            return True
        if st == GM_ERR_NOT_READY:
            return False
        self._check(st)

    def cloud_output_into(self, slot, array):
        """The same into memory the caller owns: a C-contiguous float32 [capacity, 4] numpy array (e.g. the buffer a message
        is published from) is page-locked (gm_host_register) and registered as the slot's /choppedCloud output.  The array
        must outlive the context (or a cloud_output_into(slot, None))."""
        if array is None:
            self._check(self._L.gm_set_cloud_output(self._ctx, slot, None, 0))
            return None
        assert array.dtype == np.float32 and array.ndim == 2 and array.shape[1] == 4 and array.flags["C_CONTIGUOUS"]
        self._check(self._L.gm_host_register(self._ctx, array.ctypes.data, array.nbytes))
        self._registered = getattr(self, "_registered", []) + [array]
        self._check(self._L.gm_set_cloud_output(self._ctx, slot, array.ctypes.data_as(C.POINTER(C.c_float)), array.shape[0]))
        self._cloud_slots = getattr(self, "_cloud_slots", []) + [slot]
        return array

    def cloud_output(self, slot, capacity):
        """Registers a page-locked /choppedCloud buffer for the slot (gm_set_cloud_output): every later frame of the slot
        copies its valid cloud there while the rest of the frame runs.  Returns the float32 [capacity, 4] view
        (x, y, z, bits(input row)); rows [0, n_valid) are the frame's once wait_frame has returned."""
        ptr = C.c_void_p()
        self._check(self._L.gm_host_alloc(self._ctx, int(capacity) * 16, C.byref(ptr)))
        self._pinned.append(ptr)
        self._check(self._L.gm_set_cloud_output(self._ctx, slot, C.cast(ptr, C.POINTER(C.c_float)), int(capacity)))
        self._cloud_slots = getattr(self, "_cloud_slots", []) + [slot]
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(max(int(capacity), 1), 4))[:int(capacity)]

    def _fetch(self, fn, slot, width, dtype=np.float32):
        n = C.c_uint32(0)
        st = fn(self._ctx, slot, None, 0, C.byref(n))
        if st not in (GM_OK, GM_ERR_CAPACITY):
            self._check(st)
        out = np.empty((n.value, width) if width > 1 else (n.value,), dtype=dtype)
        if n.value:
            ptr = out.ctypes.data_as(fn.argtypes[2])
            self._check(fn(self._ctx, slot, ptr, n.value, C.byref(n)))
        return out

    def cropped_cloud(self, slot=0):
        """/choppedCloud: (xyz [n,3], input row index [n])."""
        a = self._fetch(self._L.gm_get_cropped_xyz, slot, 4)
        return a[:, :3].copy(), a[:, 3].copy().view(np.int32)

    def normals(self, slot=0):
        return self._fetch(self._L.gm_get_normals, slot, 4)

    def voxel_centroids(self, slot=0):
        """(centroids [V,3], points per voxel [V]) in ascending voxel-key order."""
        a = self._fetch(self._L.gm_get_voxel_centroids, slot, 4)
        return a[:, :3].copy(), a[:, 3].astype(np.int32)

    def neighbor_counts(self, slot=0):
        return self._fetch(self._L.gm_get_neighbor_counts, slot, 1, np.int32)

    def voxel_nearest(self, slot=0):
        """Index (into cropped_cloud()) of the nearest point of every voxel centroid (GM_CFG_NEAREST)."""
        return self._fetch(self._L.gm_get_voxel_nearest, slot, 1, np.int32)

    def voxel_normals(self, slot=0):
        """normals->at(kIndices[0]) per voxel centroid (tunnel_processing.cpp:247-249): [V,4], needs GM_CFG_NEAREST."""
        return self._fetch(self._L.gm_get_voxel_normals, slot, 4)

    def labels(self, slot=0):
        """Extension: segment label per valid point (0 none, 1 plane, 2 cylinder)."""
        return self._fetch(self._L.gm_get_labels, slot, 1, np.uint8)

    def compressed_map(self, slot=0):
        """Extension: raw bytes of the build-defined map record (see decode_compressed_map)."""
        n = C.c_size_t(0)
        st = self._L.gm_get_compressed_map(self._ctx, slot, None, 0, C.byref(n))
        if st not in (GM_OK, GM_ERR_CAPACITY):
            self._check(st)
        buf = np.zeros(n.value, dtype=np.uint8)
        self._check(self._L.gm_get_compressed_map(self._ctx, slot, buf.ctypes.data, n.value, C.byref(n)))
        return buf

    def set_owned_range(self, lo, hi):
        self._check(self._L.gm_set_owned_range(self._ctx, float(lo), float(hi)))

    # ---- the reference's stage functions (tunnel_processing.hpp) ----
    def chopCloud(self, bound, cloud):
        """tunnel_processing.hpp:38.  Returns (cloudChopped [n',3], kept input rows [n'])."""
        c, keep = self._as_cloud(cloud)
        cap = max(c.n_points, 1)
        out = np.empty((cap, 4), dtype=np.float32)
        n = C.c_uint32(0)
        self._check(self._L.gm_chop_cloud(self._ctx, C.byref(c), float(bound), _f32(out), cap, C.byref(n)))
        out = out[:n.value]
        return out[:, :3].copy(), out[:, 3].copy().view(np.int32)

    def getNormals(self, neighborRadius, cloud):
        """tunnel_processing.hpp:41-45.  The reference compacts `cloud` in place and
        returns the normals; here both come back: (normals [n'',4], cloud [n'',3], kept rows [n''])."""
        xyz = np.ascontiguousarray(cloud, dtype=np.float32)
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError("cloud must be [n,3]")
        n0 = xyz.shape[0]
        cap = max(n0, 1)
        oc = np.empty((cap, 4), dtype=np.float32)
        on = np.empty((cap, 4), dtype=np.float32)
        n = C.c_uint32(0)
        self._check(self._L.gm_get_normals_stage(self._ctx, _f32(xyz), n0, float(neighborRadius), _f32(oc), _f32(on),
                                                 cap, C.byref(n)))
        oc, on = oc[:n.value], on[:n.value]
        return on.copy(), oc[:, :3].copy(), oc[:, 3].copy().view(np.int32)

    def getLocalFrame(self, cloudSize, weightingFactor, cloud_normals):
        """tunnel_processing.hpp:48-54.  Returns (eigenVals [3] ascending, eigenVecs [3,3] columns, M [3,3] fp64)."""
        nrm = np.ascontiguousarray(cloud_normals, dtype=np.float32)
        if nrm.ndim != 2 or nrm.shape[1] != 4:
            raise ValueError("normals must be [n,4] (nx,ny,nz,curvature)")
        if cloudSize > nrm.shape[0]:
            raise ValueError("cloudSize exceeds the normals cloud (the reference's .at() would throw)")
        ev = np.zeros(3, dtype=np.float32)
        V = np.zeros(9, dtype=np.float32)
        sc = np.zeros(6, dtype=np.float64)
        self._check(self._L.gm_get_local_frame(self._ctx, _f32(nrm), int(cloudSize), float(weightingFactor), _f32(ev),
                                               _f32(V), sc.ctypes.data_as(C.POINTER(C.c_double))))
        M = np.array([[sc[0], sc[1], sc[2]], [sc[1], sc[3], sc[4]], [sc[2], sc[4], sc[5]]])
        return ev, V.reshape(3, 3).T.copy(), M

    def voxelGrid(self, leafSize, cloud):
        """The pcl::VoxelGrid half of rvizNormals (tunnel_processing.hpp:77-82).
        Returns (centroids [V,3], counts [V], passthrough flag)."""
        xyz = np.ascontiguousarray(cloud, dtype=np.float32)
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError("cloud must be [n,3]")
        cap = max(xyz.shape[0], 1)
        out = np.empty((cap, 4), dtype=np.float32)
        n = C.c_uint32(0)
        fl = C.c_uint32(0)
        self._check(self._L.gm_voxel_grid(self._ctx, _f32(xyz), xyz.shape[0], float(leafSize), _f32(out), cap,
                                          C.byref(n), C.byref(fl)))
        out = out[:n.value]
        return out[:, :3].copy(), out[:, 3].astype(np.int32), bool(fl.value & _lib.GM_RES_VOXEL_PASSTHROUGH)


    # ---- extensions (no reference counterpart; SURVEY.md par. 8a-ext) ----
    @staticmethod
    def _u8(a):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.uint8)
        return a, a.ctypes.data_as(C.POINTER(C.c_uint8))

    def nearest(self, cloud, queries):
        xyz = np.ascontiguousarray(cloud, dtype=np.float32)
        q = np.ascontiguousarray(queries, dtype=np.float32)
        idx = np.empty(max(len(q), 1), dtype=np.int32)
        self._check(self._L.gm_nearest(self._ctx, _f32(xyz), len(xyz), _f32(q), len(q),
                                       idx.ctypes.data_as(C.POINTER(C.c_int32))))
        return idx[:len(q)].copy()

    def plane_hypotheses(self, cloud, seed, H, labels=None, want=0):
        xyz = np.ascontiguousarray(cloud, dtype=np.float32)
        lab, lp = self._u8(labels)
        out = np.empty((H, 4), dtype=np.float32)
        self._check(self._L.gm_plane_hypotheses(self._ctx, _f32(xyz), len(xyz), lp, want, seed, H, _f32(out)))
        return out

    def cylinder_hypotheses(self, cloud, normals, seed, H, labels=None, want=0):
        xyz = np.ascontiguousarray(cloud, dtype=np.float32)
        nrm = np.ascontiguousarray(normals, dtype=np.float32)
        lab, lp = self._u8(labels)
        out = np.empty((H, 7), dtype=np.float32)
        self._check(self._L.gm_cylinder_hypotheses(self._ctx, _f32(xyz), _f32(nrm), len(xyz), lp, want, seed, H, _f32(out)))
        return out

    def _score(self, fn, cloud, hyp, tau, labels, want):
        xyz = np.ascontiguousarray(cloud, dtype=np.float32)
        hyp = np.ascontiguousarray(hyp, dtype=np.float32)
        lab, lp = self._u8(labels)
        cnt = np.zeros(len(hyp), dtype=np.int32)
        self._check(fn(self._ctx, _f32(xyz), len(xyz), lp, want, _f32(hyp), len(hyp), float(tau),
                       cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        return cnt

    def score_planes(self, cloud, hyp4, tau, labels=None, want=0):
        return self._score(self._L.gm_score_planes, cloud, hyp4, tau, labels, want)

    def score_cylinders(self, cloud, hyp7, tau, labels=None, want=0):
        return self._score(self._L.gm_score_cylinders, cloud, hyp7, tau, labels, want)

    def score_frame(self, model, hyp, tau, slot=0, unlabelled_only=False):
        """Inlier counts of caller-supplied hypotheses ([H,4] plane rows if model == 0, [H,7] cylinder rows if 1) on the
        valid cloud the last frame left in `slot` (owned points only when sharded): gm_score_frame."""
        w = 4 if model == 0 else 7
        hyp = np.ascontiguousarray(np.asarray(hyp, dtype=np.float32).reshape(-1, w))
        counts = np.zeros(max(len(hyp), 1), dtype=np.int32)
        if len(hyp):
            self._check(self._L.gm_score_frame(self._ctx, slot, int(model), _f32(hyp), len(hyp), float(tau),
                                               1 if unlabelled_only else 0, counts.ctypes.data_as(C.POINTER(C.c_int32))))
        return counts[:len(hyp)]

    def segment_moments(self, cloud, normals, labels, label):
        xyz = np.ascontiguousarray(cloud, dtype=np.float32)
        nrm = np.ascontiguousarray(normals, dtype=np.float32) if normals is not None else None
        lab, lp = self._u8(labels)
        mom = np.zeros(16, dtype=np.float64)
        self._check(self._L.gm_segment_moments(self._ctx, _f32(xyz), _f32(nrm) if nrm is not None else None, lp,
                                               len(xyz), label, mom.ctypes.data_as(C.POINTER(C.c_double))))
        return mom


def decode_compressed_map(buf):
    """Parse gm_get_compressed_map bytes (gm_map_header / gm_map_primitive in include/gm_hip.h)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    if bytes(buf[:4]) != b"GMAP":
        raise ValueError("not a GMAP record")
    version, nprim, nvox = np.frombuffer(buf, np.uint32, 3, 4)
    leaf, bound = np.frombuffer(buf, np.float32, 2, 16)
    n_points = int(np.frombuffer(buf, np.uint32, 1, 24)[0])
    ev = np.frombuffer(buf, np.float32, 3, 32).copy()
    axis = np.frombuffer(buf, np.float32, 3, 44).copy()
    off = 56
    prims = []
    for _ in range(int(nprim)):
        typ, inl = np.frombuffer(buf, np.uint32, 2, off)
        params = np.frombuffer(buf, np.float32, 7, off + 8).copy()
        prims.append(dict(type=int(typ), inliers=int(inl), params=params))
        off += 40
    vox = np.frombuffer(buf, np.float32, 4 * int(nvox), off).reshape(-1, 4).copy()
    return dict(version=int(version), leaf=float(leaf), bound=float(bound), n_points=n_points, eigenvalues=ev,
                center_axis=axis, primitives=prims, voxels=vox)


class GeometricMappingGroup:
    """gm_group: one host thread driving every listed GPU; ONE frame sharded spatially across them with an in-library
    RCCL all-gather of the per-rank records (include/gm_hip.h, "multi-device group").  devices=[0, 0, ...] with
    loopback=True runs several ranks on one GPU (tests on a 1-GPU box): same code, records travel by device copies."""

    def __init__(self, devices, loopback=False, **cfg_kw):
        if not loopback:
            # the group loads RCCL (the copy PyTorch bundles when PyTorch is installed: _lib.py).  Observed on ROCm 7.2 /
            # torch 2.10: a process that initialises RCCL first and imports torch afterwards aborts at interpreter exit
            # ("double free or corruption"); the other order is clean.  So torch, if present, goes first.
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        self._L = _lib.load()
        cfg = Config()
        self._L.gm_default_config(C.byref(cfg))
        for k, v in cfg_kw.items():
            setattr(cfg, k, v)
        dev = (C.c_int32 * len(devices))(*devices)
        self._grp = C.c_void_p()
        st = self._L.gm_group_create(C.byref(cfg), dev, len(devices), _lib.GM_GROUP_LOOPBACK if loopback else 0, C.byref(self._grp))
        if st != GM_OK:
            msg = self._L.gm_group_last_error(None).decode()
            self._grp = None
            raise GmError(st, msg)

    def close(self):
        if getattr(self, "_grp", None):
            self._L.gm_group_destroy(self._grp)
            self._grp = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._L.gm_group_size(self._grp))

    def _check(self, st):
        if st != GM_OK:
            raise GmError(st, self._L.gm_group_last_error(self._grp).decode())

    def process_frame(self, cloud):
        c, keep = cloud if (isinstance(cloud, tuple) and isinstance(cloud[0], Cloud)) else GeometricMapping._cloud_from_xyz(cloud)
        res = FrameResult()
        self._check(self._L.gm_group_process_frame(self._grp, C.byref(c), C.byref(res)))
        return GeometricMapping._result(res)

    def _fetch(self, fn, width, dtype=np.float32):
        n = C.c_uint32(0)
        st = fn(self._grp, None, 0, C.byref(n))
        if st not in (GM_OK, GM_ERR_CAPACITY):
            self._check(st)
        out = np.empty((n.value, width) if width > 1 else (n.value,), dtype=dtype)
        if n.value:
            self._check(fn(self._grp, out.ctypes.data_as(fn.argtypes[1]), n.value, C.byref(n)))
        return out

    def cropped_cloud(self):
        """/choppedCloud of the sharded frame in the single-GPU order: (xyz [n,3], input row index [n])."""
        out = self._fetch(self._L.gm_group_get_cropped_xyz, 4)
        return out[:, :3].copy(), out[:, 3].copy().view(np.int32)

    def voxel_centroids(self):
        """(centroids [V,3], points per voxel [V]) of the sharded frame, ascending voxel-key order."""
        a = self._fetch(self._L.gm_group_get_voxel_centroids, 4)
        return a[:, :3].copy(), a[:, 3].astype(np.int32)

    def voxel_normals(self):
        """normals->at(kIndices[0]) per voxel centroid of the sharded frame: [V,4] (GM_CFG_NEAREST)."""
        return self._fetch(self._L.gm_group_get_voxel_normals, 4)

    def voxel_nearest(self):
        """Index (into cropped_cloud()) of the nearest valid point of every voxel centroid (GM_CFG_NEAREST)."""
        return self._fetch(self._L.gm_group_get_voxel_nearest, 1, np.int32)

    def timing(self):
        """Wall-clock split of the last process_frame call, milliseconds."""
        t = (C.c_double * _lib.GM_GROUP_N_TIMINGS)()
        self._check(self._L.gm_group_get_timing(self._grp, t, _lib.GM_GROUP_N_TIMINGS))
        return dict(zip(("cut_ms", "submit_ms", "device_ms", "merge_ms", "total_ms"), (float(x) for x in t)))

    def edges(self):
        """(slab edges [n_ranks + 1], whether they lie on planes of the VoxelGrid lattice)."""
        e = (C.c_double * (len(self) + 1))()
        on = C.c_uint32(0)
        self._check(self._L.gm_group_get_edges(self._grp, e, len(self) + 1, C.byref(on)))
        return np.array(e[:]), bool(on.value)

    # ---- streaming: whole frames round-robin over the devices
    def submit_frame(self, cloud):
        c, keep = cloud if (isinstance(cloud, tuple) and isinstance(cloud[0], Cloud)) else GeometricMapping._cloud_from_xyz(cloud)
        self._check(self._L.gm_group_submit_frame(self._grp, C.byref(c)))

    def wait_frame(self):
        """The oldest frame in flight: (result, rank, slot)."""
        res = FrameResult()
        rank, slot = C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.gm_group_wait_frame(self._grp, C.byref(res), C.byref(rank), C.byref(slot)))
        return GeometricMapping._result(res), int(rank.value), int(slot.value)

    def in_flight(self):
        return int(self._L.gm_group_in_flight(self._grp))

    def poll_frame(self):
        """True when the oldest frame in flight has finished (wait_frame returns at once); never blocks."""
        st = self._L.gm_group_poll_frame(self._grp)
        if st == GM_OK:
            return True
        if st == GM_ERR_NOT_READY:
            return False
        self._check(st)

    def rank_fetch(self, rank, slot, what):
        """Bulky output of a streamed frame: what in {"cropped_xyz", "normals", "voxel_centroids"} -> float32 [n,4]."""
        fn = getattr(self._L, "gm_get_" + what)
        ctx = self._L.gm_group_ctx(self._grp, rank)
        n = C.c_uint32(0)
        st = fn(ctx, slot, None, 0, C.byref(n))
        if st not in (GM_OK, GM_ERR_CAPACITY):
            raise GmError(st, self._L.gm_last_error(ctx).decode())
        out = np.empty((n.value, 4), dtype=np.float32)
        if n.value:
            st = fn(ctx, slot, _f32(out), n.value, C.byref(n))
            if st != GM_OK:
                raise GmError(st, self._L.gm_last_error(ctx).decode())
        return out


def solve_local_frame(scatter6):
    """Eigen-solve a merged scatter matrix (multi-GPU merge unit)."""
    L = _lib.load()
    sc = np.ascontiguousarray(scatter6, dtype=np.float64)
    ev = np.zeros(3, dtype=np.float32)
    V = np.zeros(9, dtype=np.float32)
    st = L.gm_solve_local_frame(sc.ctypes.data_as(C.POINTER(C.c_double)), _f32(ev), _f32(V))
    if st != GM_OK:
        raise GmError(st, "gm_solve_local_frame")
    return ev, V.reshape(3, 3).T.copy()
