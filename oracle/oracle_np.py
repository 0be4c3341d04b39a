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


# ---------------------------------------------------------------------------------------------------------------
# fp32-faithful twin: a second, separately written statement of the oracle's GMO_F32_FAITHFUL / GMO_F32_SHIFTED
# modes (PCL's fp32 accumulators in FLANN's sorted-by-distance order, pcl::eigen33's trigonometric roots, getLocalFrame's
# sequential fp32 scatter, Eigen's float tridiagonal QR) in numpy float32 SCALAR arithmetic -- pure-Python loops, small
# clouds only.  numpy's float32 sqrt / atan2 / sin / cos may differ from glibc's by an ulp, so the twin pins the C code
# to ~1e-6, its fp32 running sums bit for bit.
# ---------------------------------------------------------------------------------------------------------------
F = np.float32


def _sorted_neighbours(xyz, radius):
    """FLANN radiusSearch with sorted = true (pcl::search::KdTree's default): by fp32 distance, then by index."""
    xyz = np.asarray(xyz, dtype=np.float32)
    out = []
    for i, nb in enumerate(_neighbour_lists(xyz, radius)):
        d = xyz[i][None, :] - xyz[nb]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        out.append(nb[np.lexsort((nb, d2))])
    return out


def _roots2(b, c):
    d = F(np.float64(b) * np.float64(b) - 4.0 * np.float64(c))     # Scalar(b*b - 4.0*c): double expression -> float
    if d < 0:
        d = F(0)
    sd = F(np.sqrt(d))
    return [F(0), F(0.5) * (b - sd), F(0.5) * (b + sd)]


def _compute_roots(m):
    """pcl::computeRoots (pcl/common/impl/eigen.hpp), float."""
    c0 = (m[0][0] * m[1][1] * m[2][2] + F(2) * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2]
          - m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1])
    c1 = (m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2])
    c2 = m[0][0] + m[1][1] + m[2][2]
    if abs(c0) < np.finfo(np.float32).eps:
        return _roots2(c2, c1)
    inv3, sqrt3 = F(1.0 / 3.0), F(np.sqrt(F(3)))
    c2_3 = c2 * inv3
    a_3 = (c1 - c2 * c2_3) * inv3
    if a_3 > 0:
        a_3 = F(0)
    half_b = F(0.5) * (c0 + c2_3 * (F(2) * c2_3 * c2_3 - c1))
    q = half_b * half_b + a_3 * a_3 * a_3
    if q > 0:
        q = F(0)
    rho = F(np.sqrt(-a_3))
    theta = F(np.arctan2(F(np.sqrt(-q)), half_b)) * inv3
    ct, st = F(np.cos(theta)), F(np.sin(theta))
    r = [c2_3 + F(2) * rho * ct, c2_3 - rho * (ct + sqrt3 * st), c2_3 - rho * (ct - sqrt3 * st)]
    if r[0] >= r[1]:
        r[0], r[1] = r[1], r[0]
    if r[1] >= r[2]:
        r[1], r[2] = r[2], r[1]
        if r[0] >= r[1]:
            r[0], r[1] = r[1], r[0]
    if r[0] <= 0:
        return _roots2(c2, c1)
    return r


def _eigen33_smallest(C):
    """pcl::eigen33(mat, eigenvalue, eigenvector): scale, smallest root, largest row cross product."""
    scale = F(np.max(np.abs(C)))
    if scale <= np.finfo(np.float32).tiny:
        scale = F(1)
    s = [[F(C[i][j]) / scale for j in range(3)] for i in range(3)]
    roots = _compute_roots(s)
    ev = roots[0] * scale
    for k in range(3):
        s[k][k] = s[k][k] - roots[0]

    def cross(a, b):
        return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
    v = [cross(s[0], s[1]), cross(s[0], s[2]), cross(s[1], s[2])]
    ln = [x[0] * x[0] + x[1] * x[1] + x[2] * x[2] for x in v]
    if ln[0] >= ln[1] and ln[0] >= ln[2]:
        k = 0
    elif ln[1] >= ln[0] and ln[1] >= ln[2]:
        k = 1
    else:
        k = 2
    with np.errstate(invalid="ignore", divide="ignore"):
        nrm = F(np.sqrt(ln[k]))
        return ev, [v[k][0] / nrm, v[k][1] / nrm, v[k][2] / nrm]


def normals_f32(xyz, radius, shifted=False):
    """NormalEstimation::computeFeature in PCL's float arithmetic (src/tunnel_processing.cpp:58-70):
    shifted=False: PCL <= 1.9 computeMeanAndCovarianceMatrix (un-shifted), True: PCL >= 1.10 (offsets from the first
    neighbour).  Returns (normals [n,4], counts [n], the nine fp32 running sums per point [n,9])."""
    xyz = np.asarray(xyz, dtype=np.float32)
    n = len(xyz)
    out = np.full((n, 4), np.nan, dtype=np.float32)
    cnt = np.zeros(n, dtype=np.int32)
    sums = np.zeros((n, 9), dtype=np.float32)
    for i, nb in enumerate(_sorted_neighbours(xyz, radius)):
        cnt[i] = len(nb)
        if len(nb) < 3:
            continue
        K = xyz[nb[0]] if shifted else np.zeros(3, np.float32)
        a = [F(0)] * 9
        for j in nb:
            p = xyz[j] - K
            a[0] += p[0] * p[0]; a[1] += p[0] * p[1]; a[2] += p[0] * p[2]
            a[3] += p[1] * p[1]; a[4] += p[1] * p[2]; a[5] += p[2] * p[2]
            a[6] += p[0]; a[7] += p[1]; a[8] += p[2]
        sums[i] = a
        m = F(len(nb))
        a = [x / m for x in a]
        C = [[a[0] - a[6] * a[6], a[1] - a[6] * a[7], a[2] - a[6] * a[8]],
             [F(0), a[3] - a[7] * a[7], a[4] - a[7] * a[8]], [F(0), F(0), a[5] - a[8] * a[8]]]
        C[1][0], C[2][0], C[2][1] = C[0][1], C[0][2], C[1][2]
        ev, v = _eigen33_smallest(np.array(C, dtype=np.float32))
        tr = C[0][0] + C[1][1] + C[2][2]
        with np.errstate(invalid="ignore", divide="ignore"):
            curv = abs(ev / tr) if tr != 0 else F(0)
        q = xyz[i]
        ct = (F(0) - q[0]) * v[0] + (F(0) - q[1]) * v[1] + (F(0) - q[2]) * v[2]
        if ct < 0:
            v = [-v[0], -v[1], -v[2]]
        out[i] = [v[0], v[1], v[2], curv]
    return out, cnt, sums


def local_frame_f32(normals4, wf):
    """getLocalFrame as the reference's float arithmetic runs it (src/tunnel_processing.cpp:100-124): weight evaluated
    in double and stored to float (:106), fp32 products, sequential fp32 sums.  Returns M [3,3] float32."""
    import math
    M = [[F(0)] * 3 for _ in range(3)]
    for nr in np.asarray(normals4, dtype=np.float32):
        w = F(math.exp(math.pow(float(nr[3]) + .001 / wf, 2)))
        a, b, c = w * nr[0], w * nr[1], w * nr[2]
        M[0][0] += a * a; M[0][1] += a * b; M[0][2] += a * c
        M[1][1] += b * b; M[1][2] += b * c; M[2][2] += c * c
    M[1][0], M[2][0], M[2][1] = M[0][1], M[0][2], M[1][2]
    return np.array(M, dtype=np.float32)
