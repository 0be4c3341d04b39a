"""Multi-GPU sharding of one frame into x-slabs with a radius halo, and the merge
of per-slab results (SURVEY.md par. 8e).

The path shards because every per-point stage depends only on points within
`neighborRadius` and the final fit is a SUM (M = sum w^2 n n^T) followed by a
3x3 solve.  Slab g owns points with x in [edges[g], edges[g+1]) and also receives
halo points within `halo` of its edges: neighbours only, never query results
(gm_set_owned_range).  The cut is made on the host before H2D, so no device
P2P halo exchange exists; the only data-path collective is the all-gather of
one small record per rank (6 fp64 scatter entries + counts).

Pure numpy host logic: no arithmetic of the path itself happens here except the
count-weighted merge of voxel centroids split by a slab edge.
"""
from __future__ import annotations

import numpy as np

__all__ = ["slab_edges", "cut_slabs", "merge_scatter", "merge_clouds", "merge_voxels", "RECORD_LEN",
           "pack_record", "unpack_records", "PRIMITIVE_LEN", "pack_primitives", "unpack_primitives", "vote_primitives"]


def slab_edges(xyz, n_slabs, bound):
    """Balanced edges along x: equal counts of in-box points per slab.
    Returns n_slabs+1 float64 edges, first = -inf, last = +inf."""
    xyz = np.asarray(xyz, dtype=np.float32)
    lo, hi = np.float32(-bound), np.float32(bound)
    with np.errstate(invalid="ignore"):
        inside = np.isfinite(xyz).all(axis=1) & ~((xyz < lo).any(axis=1) | (xyz > hi).any(axis=1))
    x = np.sort(xyz[inside, 0].astype(np.float64))
    edges = [-np.inf]
    for g in range(1, n_slabs):
        if len(x) == 0:
            edges.append(0.0)
        else:
            q = x[min(len(x) - 1, (len(x) * g) // n_slabs)]
            # edges must be float32-representable: ownership is tested in fp32 on the device
            edges.append(float(np.float32(q)))
    edges.append(np.inf)
    return np.asarray(edges, dtype=np.float64)


def cut_slabs(xyz, edges, halo):
    """For each slab: (rows [m] int64 = input rows sent to that rank, in input order)."""
    x = np.asarray(xyz, dtype=np.float32)[:, 0].astype(np.float64)
    h = float(halo)
    out = []
    for g in range(len(edges) - 1):
        with np.errstate(invalid="ignore"):
            sel = (x >= edges[g] - h) & (x < edges[g + 1] + h)
        out.append(np.nonzero(sel)[0])
    return out


RECORD_LEN = 10  # scatter xx,xy,xz,yy,yz,zz, n_in, n_cropped(own+halo), n_valid, n_voxels


def pack_record(res):
    r = np.zeros(RECORD_LEN, dtype=np.float64)
    r[:6] = res["scatter6"]
    r[6:] = [res["n_in"], res["n_cropped"], res["n_valid"], res["n_voxels"]]
    return r


def unpack_records(recs):
    recs = np.asarray(recs, dtype=np.float64).reshape(-1, RECORD_LEN)
    return recs[:, :6], recs[:, 6:].astype(np.int64)


# ---- fitted primitives of a sharded frame (north_star: "all-gather of fitted primitives") -------------------------
# Every rank fits its own plane / cylinder on its slab (the in-frame RANSAC).  Those fits are CANDIDATES for the
# whole frame: round 1 all-gathers them (PRIMITIVE_LEN doubles per rank), then every rank counts the inliers of every
# candidate on its OWN slab (gm_score_frame: owned points only, so counts add up exactly), round 2 sums the counts
# over the ranks, and the candidate with the largest global count wins (lowest rank on ties).  Two collectives of a
# few hundred bytes; no point ever crosses a link.
PRIMITIVE_LEN = 13  # plane a,b,c,d | cylinder px,py,pz,dx,dy,dz,r | plane_inliers, cylinder_inliers (local)


def pack_primitives(res):
    r = np.full(PRIMITIVE_LEN, np.nan, dtype=np.float64)
    r[0:4] = res["plane"]
    r[4:11] = res["cylinder"]
    r[11:13] = [res["plane_inliers"], res["cylinder_inliers"]]
    return r


def unpack_primitives(recs):
    """-> (planes [world,4] float32, cylinders [world,7] float32, local inlier counts [world,2])."""
    recs = np.asarray(recs, dtype=np.float64).reshape(-1, PRIMITIVE_LEN)
    return recs[:, 0:4].astype(np.float32), recs[:, 4:11].astype(np.float32), recs[:, 11:13].astype(np.int64)


def vote_primitives(candidates, global_counts):
    """candidates [world, k] rows, global_counts [world] = inliers of each candidate summed over all slabs.
    Returns (winning row or None, its count, index): largest count, lowest rank on ties; candidates holding a NaN
    (a rank whose slab produced no model) never win."""
    cand = np.asarray(candidates)
    cnt = np.asarray(global_counts, dtype=np.int64).copy()
    ok = np.isfinite(cand).all(axis=1)
    if not ok.any():
        return None, 0, -1
    cnt[~ok] = -1
    k = int(np.argmax(cnt))  # argmax returns the first maximum: lowest rank on ties
    return cand[k].copy(), int(cnt[k]), k


def merge_scatter(scatter6_rows):
    """Sum of per-slab fp64 scatter partials, in rank order (deterministic)."""
    m = np.zeros(6, dtype=np.float64)
    for row in np.asarray(scatter6_rows, dtype=np.float64).reshape(-1, 6):
        m += row
    return m


def merge_clouds(parts):
    """parts: list of (global_rows [k], payload [k, c]) per slab -> payload in
    single-GPU order (ascending input row), rows."""
    rows = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.int64)
    pay = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, 3), np.float32)
    order = np.argsort(rows, kind="stable")
    return pay[order], rows[order]


def merge_voxels(parts, leaf):
    """parts: list of (centroids [v,3] float32, counts [v]) per slab.  Voxel cells sit
    at multiples of `leaf` in absolute coordinates (floor(x * (1/leaf)) in fp32, as
    pcl::VoxelGrid computes them), so per-slab centroids of the same cell are
    pieces of one voxel: count-weighted mean.  Returned in the single-GPU order
    (ascending key: x fastest, then y, then z)."""
    cen = np.concatenate([p[0] for p in parts]).astype(np.float32)
    cnt = np.concatenate([p[1] for p in parts]).astype(np.int64)
    if len(cen) == 0:
        return cen, cnt.astype(np.int32)
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(cen * inv).astype(np.int64)
    ijk -= ijk.min(axis=0)
    div = ijk.max(axis=0) + 1
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    key, cen, cnt = key[order], cen[order], cnt[order]
    uk, first = np.unique(key, return_index=True)
    wsum = np.add.reduceat(cen.astype(np.float64) * cnt[:, None], first, axis=0)
    csum = np.add.reduceat(cnt, first)
    return (wsum / csum[:, None]).astype(np.float32), csum.astype(np.int32)
