#!/usr/bin/env python3
"""Where the blocks of the cell sort's passes spend their time (diagnostic build: FILE=k_sort tools/build_variants.sh
tl "-DGM_SORT_TIMELINE"; GM_LIB_PATH=build/variants/libgm_hip_tl.so).  100 MHz ticks per phase, per pass: median and
max over the tiles, and the span of the pass (first entry to last end)."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometric_mapping_amd as g
from geometric_mapping_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--radius", type=float, default=None)
a = ap.parse_args()
r = a.radius or synth.fixed_k_radius(a.points)
xyz = synth.tunnel_frame(a.points, seed=0)
lib = _lib.load()
lib.gm_debug_sort_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
buf = (ctypes.c_uint64 * (4 * 256 * 8))()
with g.GeometricMapping(neighborRadius=r, flags=_lib.GM_CFG_DEFAULT, max_points=a.points) as c:
    for _ in range(3):
        c.process_frame(xyz)
    res = c.process_frame(xyz)
    rc = lib.gm_debug_sort_timeline(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4, 256, 8).astype(np.int64)
names = ["ticket", "load+count", "publish", "rank", "stage", "lookback", "write"]
out = {"rc": rc, "n_cropped": res["n_cropped"]}
for p in range(4):
    tp = t[p]
    ok = (tp[:, 0] > 0) & (tp[:, 7] >= tp[:, 0])
    if not ok.any():
        continue
    tp = tp[: int(ok.sum())] if ok[: int(ok.sum())].all() else tp[ok]
    d = np.diff(tp, axis=1) / 100.0   # us
    out["pass%d" % p] = {"tiles": int(ok.sum()), "span_us": float((tp[:, 7].max() - tp[:, 0].min()) / 100.0),
                         "entry_spread_us": float((tp[:, 0].max() - tp[:, 0].min()) / 100.0),
                         "median_us": {n: round(float(np.median(d[:, i])), 2) for i, n in enumerate(names)},
                         "max_us": {n: round(float(d[:, i].max()), 2) for i, n in enumerate(names)},
                         "block_total_median_us": round(float(np.median(tp[:, 7] - tp[:, 0]) / 100.0), 2)}
    # absolute view: when is a tile's record out (since the first block's entry), and how long does a look-back take once
    # every record it needs is out (max over the tiles before it of their publish time)
    t0 = tp[:, 0].min()
    pub = (tp[:, 3] - t0) / 100.0
    lb_end = (tp[:, 6] - t0) / 100.0
    lb_start = (tp[:, 5] - t0) / 100.0
    need = np.maximum.accumulate(pub)          # (tiles are in ticket order: index = tile)
    out["pass%d" % p]["publish_at_us"] = {"min": round(float(pub.min()), 2), "median": round(float(np.median(pub)), 2), "max": round(float(pub.max()), 2)}
    out["pass%d" % p]["lookback_after_inputs_us"] = {"median": round(float(np.median(lb_end - np.maximum(need, lb_start))), 2), "max": round(float((lb_end - np.maximum(need, lb_start)).max()), 2)}
    out["pass%d" % p]["entry_at_us"] = {"median": round(float(np.median(tp[:, 0] - t0) / 100.0), 2), "max": round(float((tp[:, 0] - t0).max() / 100.0), 2)}
    out["pass%d" % p]["end_at_us"] = {"median": round(float(np.median(tp[:, 7] - t0) / 100.0), 2), "max": round(float((tp[:, 7] - t0).max() / 100.0), 2)}
# the tile cutter (k_rows_and_tiles) of the same frame
lib.gm_debug_cutter_timeline.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
cbuf = (ctypes.c_uint64 * (512 * 8))()
if lib.gm_debug_cutter_timeline(cbuf) == 0:
    ct = np.frombuffer(cbuf, dtype=np.uint64).reshape(512, 8).astype(np.int64)
    ok = (ct[:, 0] > 0) & (ct[:, 7] >= ct[:, 0])
    ct = ct[ok]
    if len(ct):
        cn = ["ticket", "keys+rows", "row_walk", "cut+classify", "rank", "lookback", "write"]
        d = np.diff(ct, axis=1) / 100.0
        out["cutter"] = {"blocks": int(len(ct)), "span_us": float((ct[:, 7].max() - ct[:, 0].min()) / 100.0),
                         "median_us": {n: round(float(np.median(d[:, i])), 2) for i, n in enumerate(cn)},
                         "max_us": {n: round(float(d[:, i].max()), 2) for i, n in enumerate(cn)}}
        t0 = ct[:, 0].min()
        pub = (ct[:, 5] - t0) / 100.0
        need = np.maximum.accumulate(pub)
        lbe = (ct[:, 6] - t0) / 100.0
        out["cutter"]["publish_at_us"] = {"min": round(float(pub.min()), 2), "median": round(float(np.median(pub)), 2), "p90": round(float(np.quantile(pub, 0.9)), 2), "max": round(float(pub.max()), 2)}
        out["cutter"]["lookback_after_inputs_us"] = {"median": round(float(np.median(lbe - need)), 2), "max": round(float((lbe - need).max()), 2)}
        late = np.argsort(-pub)[:6]
        out["cutter"]["latest_publishers"] = [{"block": int(b), "publish_at": round(float(pub[b]), 2), **{n: round(float(d[b, i]), 2) for i, n in enumerate(cn[:5])}} for b in late]
        early = np.argsort(pub)[:3]
        out["cutter"]["earliest_publishers"] = [{"block": int(b), "publish_at": round(float(pub[b]), 2), **{n: round(float(d[b, i]), 2) for i, n in enumerate(cn[:5])}} for b in early]
        out["cutter"]["entry_at_us"] = {"median": round(float(np.median(ct[:, 0] - t0) / 100.0), 2), "max": round(float((ct[:, 0] - t0).max() / 100.0), 2)}
print(json.dumps(out, indent=1))
