#!/usr/bin/env python3
"""CPU design study for k_normals' search grid (no GPU needed).

Counts, for a synthetic frame, how many candidates a wave would stream under a
given grid design: y/z rows of edge r/D, per-row x-reach sqrt(r^2 - gap^2),
tiles of <= Q consecutive points of one row, G lane groups with their own
x-window per row.  Reports the fraction of tested pair slots that are true
neighbours -- the number tools/normals_stats.py measures on the GPU for the
design that is actually built.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree

from geometric_mapping_amd import synth


def simulate(xyz, r, D, Q=64, G=4, rnd=8, bound=5.0, mfma_block=0):
    h = 1.001 * r / D
    lo = -bound
    ny = int(np.floor(2 * bound / h)) + 1
    iy = np.clip(np.floor((xyz[:, 1] - lo) / h).astype(np.int64), 0, ny - 1)
    iz = np.clip(np.floor((xyz[:, 2] - lo) / h).astype(np.int64), 0, ny - 1)
    row = iz * ny + iy
    order = np.lexsort((xyz[:, 0], row))
    row = row[order]
    x = xyz[order, 0].astype(np.float64)
    n = len(x)
    key = row.astype(np.float64) * 64.0 + (x + 16.0)  # row-major, x inside a row
    # tiles: chunks of Q from each row start
    starts = np.flatnonzero(np.r_[True, row[1:] != row[:-1]])
    ends = np.r_[starts[1:], n]
    t_s, t_e = [], []
    for s, e in zip(starts, ends):
        a = np.arange(s, e, Q)
        t_s.append(a)
        t_e.append(np.minimum(a + Q, e))
    t_s = np.concatenate(t_s)
    t_e = np.concatenate(t_e)
    nt = len(t_s)
    gl = Q // G
    # group x intervals (inactive lanes repeat the tile's last query)
    gmin = np.empty((nt, G))
    gmax = np.empty((nt, G))
    for g in range(G):
        a = np.minimum(t_s + g * gl, t_e - 1)
        b = np.minimum(t_s + g * gl + gl - 1, t_e - 1)
        gmin[:, g] = x[a]
        gmax[:, g] = x[b]
    trow = row[t_s]
    ty, tz = trow % ny, trow // ny
    streamed = np.zeros(nt)
    groupsum = np.zeros(nt)
    mfma = np.zeros(nt)
    nonempty_rows = np.zeros(nt)
    for a in range(-D, D + 1):
        for b in range(-D, D + 1):
            gy = max(0, abs(a) - 1) * h * 0.999
            gz = max(0, abs(b) - 1) * h * 0.999
            reach2 = r * r - gy * gy - gz * gz
            if reach2 <= 0:
                continue
            reach = np.sqrt(reach2) + r / 64.0  # fine-cell rounding of the window ends
            yy, zz = ty + a, tz + b
            ok = (yy >= 0) & (yy < ny) & (zz >= 0) & (zz < ny)
            nrow = (zz * ny + yy).astype(np.float64)
            wl = np.zeros((nt, G))
            for g in range(G):
                kb = nrow * 64.0 + (gmin[:, g] - reach + 16.0)
                ke = nrow * 64.0 + (gmax[:, g] + reach + 16.0)
                sb = np.searchsorted(key, kb, side="left")
                se = np.searchsorted(key, ke, side="right")
                wl[:, g] = np.where(ok, se - sb, 0)
            mx = wl.max(axis=1)
            streamed += np.ceil((mx + 1.5) / rnd) * rnd * (mx > 0)  # +1.5: window start aligned down to 4
            groupsum += wl.sum(axis=1)
            if mfma_block:
                mfma += (np.ceil((wl + 3.5) / mfma_block) * mfma_block * (wl > 0)).sum(axis=1)
            nonempty_rows += mx > 0
    return dict(n=n, tiles=nt, lane_fill=n / (Q * nt), streamed_per_tile=streamed.mean(),
                group_window_per_tile=groupsum.mean() / G, rows_per_tile=nonempty_rows.mean(),
                pair_slots=float((streamed * Q).sum()), group_slots=float(groupsum.sum() * gl),
                mfma_slots=float(mfma.sum() * gl))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--floor", action="store_true")
    a = ap.parse_args()
    r = a.radius or synth.fixed_k_radius(a.points)
    xyz = synth.tunnel_frame(a.points, seed=0, floor_z=-1.2 if a.floor else None, outlier_frac=0.01 if a.floor else 0.0)
    keep = np.all(np.abs(xyz) <= 5.0, axis=1)
    xyz = xyz[keep]
    rng = np.random.default_rng(1)
    sample = rng.choice(len(xyz), 20000, replace=False)
    tree = cKDTree(xyz)
    k = np.mean([len(v) for v in tree.query_ball_point(xyz[sample], r * 0.99999)])
    hits = k * len(xyz)
    print(json.dumps({"n_cropped": len(xyz), "r": r, "mean_k": k}))
    for D, Q, G in [(1, 64, 4), (2, 64, 4), (2, 64, 8), (3, 64, 4), (2, 32, 2), (2, 128, 8), (4, 64, 4)]:
        s = simulate(xyz, r, D, Q, G, mfma_block=32)
        s.update(D=D, Q=Q, G=G, hit_all_lanes=hits / s["pair_slots"], hit_group_ideal=hits / s["group_slots"],
                 hit_mfma32=hits / s["mfma_slots"])
        print(json.dumps({k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in s.items()}))


if __name__ == "__main__":
    main()
