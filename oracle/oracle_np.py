"""numpy/scipy twin of the C restatement -- an INDEPENDENT second statement of
the same semantics (cKDTree radius search, numpy.linalg.eigh), used only to
cross-check oracle/gm_oracle.c in tests.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see gm_oracle.h).
Citations are to /root/reference.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is present in the dev container and on the GPU box image
    from scipy.spatial import cKDTree
except Exception:  # pragma: no cover
    cKDTree = None


def crop_box(xyz, bound):
    """src/tunnel_processing.cpp:39-49 (CropBox, bounds cast to float, closed box)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    lo, hi = np.float32(-bound), np.float32(bound)
    fin = np.isfinite(xyz).all(axis=1)
    with np.errstate(invalid="ignore"):
        outside = (xyz < lo).any(axis=1) | (xyz > hi).any(axis=1)
    return np.nonzero(fin & ~outside)[0].astype(np.int32)


def _neighbour_lists(xyz, radius):
    """Exact fp32 neighbour sets: candidates from a slightly inflated double
    search, then the FLANN predicate ((dx*dx+dy*dy)+dz*dz in fp32) < fp32(r*r)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    tree = cKDTree(xyz.astype(np.float64))
    cand = tree.query_ball_point(xyz.astype(np.float64), r=float(radius) * (1 + 1e-4) + 1e-6)
    r2 = np.float32(float(radius) * float(radius))
    out = []
    for i, c in enumerate(cand):
        c = np.asarray(sorted(c), dtype=np.int64)
        d = xyz[i][None, :] - xyz[c]                      # float32
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]   # float32, each op rounded
        out.append(c[d2 < r2])
    return out


def normals(xyz, radius):
    """src/tunnel_processing.cpp:58-70; double arithmetic (mathematical value)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    n = len(xyz)
    out = np.full((n, 4), np.nan, dtype=np.float32)
    cnt = np.zeros(n, dtype=np.int32)
    for i, nb in enumerate(_neighbour_lists(xyz, radius)):
        cnt[i] = len(nb)
        if len(nb) < 3:
            continue
        p = xyz[nb].astype(np.float64)
        mu = p.mean(axis=0)
        q = p - mu
        Cm = q.T @ q / len(nb)                            # biased, /m like PCL
        w, V = np.linalg.eigh(Cm)
        nv = V[:, 0]
        tr = np.trace(Cm)
        curv = abs(w[0] / tr) if tr != 0 else 0.0
        if np.dot(-xyz[i].astype(np.float64), nv) < 0:     # flip towards viewpoint (0,0,0)
            nv = -nv
        out[i, :3] = nv
        out[i, 3] = curv
    return out, cnt


def voxel_grid(xyz, leaf):
    """src/tunnel_processing.cpp:214-220 (pcl::VoxelGrid), centroids in double."""
    xyz = np.asarray(xyz, dtype=np.float32)
    if len(xyz) == 0:
        return np.zeros((0, 3), np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = xyz.min(axis=0), xyz.max(axis=0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(xyz * inv) - min_b.astype(np.float32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    uk, first, counts = np.unique(ks, return_index=True, return_counts=True)
    sums = np.add.reduceat(xyz[order].astype(np.float64), first, axis=0)
    cen = (sums / counts[:, None]).astype(np.float32)
    return cen, uk.astype(np.int32), counts.astype(np.int32)


def local_frame(nrm, wf):
    """src/tunnel_processing.cpp:92-148: w=exp((c+.001/wf)^2), M=sum w^2 n n^T, eigh."""
    nrm = np.asarray(nrm, dtype=np.float32).astype(np.float64)
    w = np.exp((nrm[:, 3] + .001 / wf) ** 2)
    wn = nrm[:, :3] * w[:, None]
    M = wn.T @ wn
    ev, V = np.linalg.eigh(M)
    return ev, V, M


def nearest(xyz, queries):
    tree = cKDTree(np.asarray(xyz, dtype=np.float64))
    _, idx = tree.query(np.asarray(queries, dtype=np.float64), k=1)
    return idx.astype(np.int32)


def process_frame(xyz, bound, radius, leaf, wf):
    """src/geometric_mapping.cpp:48-125, processing half."""
    xyz = np.asarray(xyz, dtype=np.float32)
    keep = crop_box(xyz, bound)
    c1 = xyz[keep]
    nrm, _ = normals(c1, radius)
    ok = np.isfinite(nrm[:, :3]).all(axis=1)
    c2, n2 = c1[ok], nrm[ok]
    cen, key, cnt = voxel_grid(c2, leaf)
    ev, V, M = local_frame(n2, wf)
    return dict(n_in=len(xyz), n_cropped=len(c1), n_valid=len(c2), n_voxels=len(cen),
                xyz=c2, normals=n2, voxels=cen, evals=ev, evecs=V, M=M)
