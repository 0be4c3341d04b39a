#!/usr/bin/env python3
"""bench.py -- points/sec of the full per-frame path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (ingest+crop -> cell sort -> radius normals ->
NaN compaction -> voxel grid -> scatter matrix + 3x3 eigen [-> RANSAC ext]) over one
synthetic 1M-point tunnel frame whose PointCloud2 rows are already resident in HBM.
Workload = BASELINE.json configs[1] ("1M-pt synthetic tunnel frame, single MI355X"),
fixed-k neighbour radius r = 0.5*sqrt(50000/N) (SURVEY.md par. 8d), launch-file values
for everything else.

N > 1 (one process per GPU, RCCL via torch.distributed):
  --mode frames (default): every rank runs its own stream of 1M-pt frames (frames are
      independent: BASELINE config 5); the only collective is one all-gather of the
      fitted records at the end of the timed region.  scaling = weak.
  --mode slab: ONE frame cut into x-slabs with a radius halo; per step each rank
      processes its slab, all-gathers a 10-double record (scatter partials + counts)
      over RCCL and solves the merged 3x3.  scaling = strong.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy peak 6290
HBM_MEASURED_GBS = 6290.0
ALGO_BYTES_PER_POINT = 40.0    # SURVEY.md par. 8d: 12 B/pt in + 28 B per cropped point out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--radius", type=float, default=None, help="default: fixed-k 0.5*sqrt(50000/points)")
    ap.add_argument("--mode", choices=("frames", "slab"), default="frames")
    ap.add_argument("--slots", type=int, default=3, help="frames in flight per GPU (HIP streams)")
    ap.add_argument("--ransac", type=int, default=1,
                    help="1: the step includes ONE RANSAC model (cylinder, H=1024: BASELINE configs[1]); 0: reference-faithful path only")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra reference-faithful / host-input legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL) is the real thing; gloo only rehearses the N>1 control flow on a 1-GPU box")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this device")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box shows all
    256 hardware threads but grants a 16-CPU quota per GPU; more threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(points, radius, threads):
    """The oracle (f32-faithful restatement of the reference's PCL/Eigen path), timed on
    this box's host cores on ONE frame of the same workload.  Baseline only."""
    from geometric_mapping_amd import synth
    from oracle import oracle_c as oc
    oc.build()
    xyz = synth.tunnel_frame(points, seed=0)
    ncores = threads or usable_cpus()
    t0 = time.perf_counter()
    oc.process_frame(xyz, 5.0, radius, 0.5, 0.2, oc.F32_FAITHFUL, nthreads=ncores, want_outputs=False)
    dt_all = time.perf_counter() - t0
    # single thread = how the reference actually executes (ros::spin, non-OMP NormalEstimation).
    # Bounded sample: one fifth of the frame; density drops 5x, so r*sqrt(5) keeps k (~256) and
    # with it the work per point.
    sub = xyz[: points // 5]
    t0 = time.perf_counter()
    oc.process_frame(sub, 5.0, radius * np.sqrt(5.0), 0.5, 0.2, oc.F32_FAITHFUL, nthreads=1, want_outputs=False)
    dt_one = time.perf_counter() - t0
    return {"value": points / dt_all, "unit": "points/s", "cores": int(ncores), "kind": "port",
            "sample": f"1 frame of the same workload ({points} pts, r={radius:.4f}), oracle f32_faithful, OpenMP over query points",
            "single_thread_value": (points // 5) / dt_one,
            "single_thread_sample": f"{points // 5} pts at matched k (r*sqrt(5)), 1 thread"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import torch  # before libgm_hip: one HIP runtime per process (geometric_mapping_amd/_lib.py)
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback to time)")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import geometric_mapping_amd as g
    from geometric_mapping_amd import _lib, sharding, synth

    n = args.points
    radius = args.radius if args.radius else synth.fixed_k_radius(n)
    bound, leaf, wf = 5.0, 0.5, 0.2
    flags = _lib.GM_CFG_DEFAULT | _lib.GM_CFG_STAGE_TIMING
    ransac_on = False
    if args.ransac:
        probe = g.load_library()
        # the extension is part of the step only once its kernels exist in the library
        ransac_on = bool(getattr(probe, "gm_ext_available", None) and probe.gm_ext_available())
        if ransac_on:
            flags |= _lib.GM_CFG_RANSAC_CYLINDER

    def rows16(xyz):
        a = np.zeros((len(xyz), 4), dtype=np.float32)
        a[:, :3] = xyz
        return a

    n_slots = max(1, args.slots)
    mode = args.mode if world > 1 else "frames"
    if mode == "frames":
        n_frames = 4
        frames_host = [synth.tunnel_frame(n, seed=1000 * rank + s) for s in range(n_frames)]
        dev = [torch.from_numpy(rows16(f)).cuda() for f in frames_host]
        pts_per_step_rank = n
        own = None
    else:
        full = synth.tunnel_frame(n, seed=0)
        edges = sharding.slab_edges(full, world, bound)
        rows = sharding.cut_slabs(full, edges, halo=radius * 1.01)[rank]
        dev = [torch.from_numpy(rows16(full[rows])).cuda()]
        pts_per_step_rank = None  # the frame is shared: n points per step for the whole job
        own = (edges[rank], edges[rank + 1])
        n_slots = 1
    torch.cuda.synchronize()

    tau_r = 0.03  # RANSAC inlier threshold (3 sigma of the generator), H = 1024 hypotheses per model

    def make_ctx(fl, slots):
        c = g.GeometricMapping(boxFilterBound=bound, voxelGridLeafSize=leaf, neighborRadius=radius, weightingFactor=wf,
                               device=local_rank, flags=fl, n_slots=slots, max_points=max(len(d) for d in dev),
                               ransac_hypotheses=1024, ransac_threshold=tau_r, ransac_seed=1 + rank)
        if own is not None:
            c.set_owned_range(*own)
        return c

    ctx = make_ctx(flags, n_slots)
    clouds = [ctx.cloud_from_device(d.data_ptr(), d.shape[0], 16) for d in dev]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rec_dev = torch.zeros(sharding.RECORD_LEN, dtype=torch.float64, device=coll_dev)
    gathered = torch.zeros(world * sharding.RECORD_LEN, dtype=torch.float64, device=coll_dev)

    prim_dev = torch.zeros(sharding.PRIMITIVE_LEN, dtype=torch.float64, device=coll_dev)
    prim_all = torch.zeros(world * sharding.PRIMITIVE_LEN, dtype=torch.float64, device=coll_dev)
    votes_dev = torch.zeros(2 * world, dtype=torch.int64, device=coll_dev)

    def slab_step(c):
        res = c.process_frame(clouds[0])
        rec_dev.copy_(torch.from_numpy(sharding.pack_record(res)))
        dist.all_gather_into_tensor(gathered, rec_dev)
        sc, _ = sharding.unpack_records(gathered.cpu().numpy())
        g.solve_local_frame(sharding.merge_scatter(sc))
        if ransac_on:
            # primitive vote (sharding.py): all-gather of every rank's fitted primitives, each rank counts every
            # candidate on its own resident slab, all-reduce of the counts, largest global count wins
            prim_dev.copy_(torch.from_numpy(sharding.pack_primitives(res)))
            dist.all_gather_into_tensor(prim_all, prim_dev)
            planes, cyls, _ = sharding.unpack_primitives(prim_all.cpu().numpy())
            v = np.zeros(2 * world, dtype=np.int64)
            ok = np.isfinite(cyls).all(axis=1)
            if ok.any():
                v[world:][ok] = c.score_frame(1, cyls[ok], tau_r)
            votes_dev.copy_(torch.from_numpy(v))
            dist.all_reduce(votes_dev, op=dist.ReduceOp.SUM)
            res["cylinder_voted"] = sharding.vote_primitives(cyls, votes_dev.cpu().numpy()[world:])
        return res

    def run(c, inputs, steps, slots, results):
        if mode == "slab":
            for i in range(steps):
                r = slab_step(c)
                if results is not None:
                    results.append(r)
            return
        # frames: keep `slots` frames in flight on their own HIP streams
        inflight = []
        for i in range(steps):
            slot = i % slots
            if len(inflight) == slots:
                r = c.wait_frame(inflight.pop(0))
                if results is not None:
                    results.append(r)
            c.submit_frame(slot, inputs[i % len(inputs)])
            inflight.append(slot)
        while inflight:
            r = c.wait_frame(inflight.pop(0))
            if results is not None:
                results.append(r)

    def timed(c, inputs, slots):
        res = []
        run(c, inputs, args.warmup, slots, None)
        barrier()
        t0 = time.perf_counter()
        run(c, inputs, args.steps, slots, res)
        if world > 1 and mode == "frames":
            # fitted records of every rank's last frame to every rank (the node publishes them all)
            rec_dev.copy_(torch.from_numpy(sharding.pack_record(res[-1])))
            dist.all_gather_into_tensor(gathered, rec_dev)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, res

    dt, results = timed(ctx, clouds, n_slots)
    total_points = (n * args.steps * world) if mode == "frames" else (n * args.steps)
    value = total_points / dt

    secondary = {}
    if not args.no_secondary and world == 1:
        if ransac_on:  # the reference-faithful path alone (no extension work in the step)
            with make_ctx(_lib.GM_CFG_DEFAULT, n_slots) as c2:
                dt2, _ = timed(c2, clouds, n_slots)
            secondary["reference_faithful_path_only"] = {"value": n * args.steps / dt2, "ms_per_step": dt2 / args.steps * 1e3}
        # rows handed over as HOST buffers (what the ROS node does): pinned staging + H2D inside the step.
        # Never the headline value (DESIGN.md par. 7).
        host_inputs = [ctx._cloud_from_xyz(rows16(f)) for f in frames_host] if mode == "frames" else None
        if host_inputs:
            dt3, _ = timed(ctx, host_inputs, n_slots)
            secondary["host_input_pcie_inclusive"] = {"value": n * args.steps / dt3, "ms_per_step": dt3 / args.steps * 1e3,
                                                      "input": "pageable host rows: staging copy on the calling thread + H2D"}
            # the same with rows the caller already holds in page-locked memory (gm_host_alloc, GM_CLOUD_PINNED)
            pinned_inputs = []
            for f in frames_host:
                r16 = rows16(f)
                buf, as_cloud = ctx.pinned_rows(r16.shape[0], 16)
                buf[:] = r16.reshape(-1).view(np.uint8)
                pinned_inputs.append(as_cloud())
            for pc in pinned_inputs:   # a page-locked buffer's first DMA pays a one-time mapping cost: not steady state
                ctx.process_frame(pc)
            dt4, _ = timed(ctx, pinned_inputs, n_slots)
            secondary["host_input_pinned_pcie_inclusive"] = {"value": n * args.steps / dt4, "ms_per_step": dt4 / args.steps * 1e3,
                                                             "input": "page-locked host rows (GM_CLOUD_PINNED): H2D only"}

    # the same kernel with the chip to itself: frames one at a time (this is also what the kernel sees under
    # rocprofv3, whose host-side overhead keeps the frames from overlapping: profiles/README.md)
    k_ms_excl = None
    if mode == "frames":
        ex = [ctx.process_frame(clouds[i % len(clouds)])["normals_kernel_ms"] for i in range(min(args.steps, 10))]
        k_ms_excl = float(np.mean(ex))

    # dominant kernel: the neighbourhood-normals kernel, HIP-event bracketed on the stream
    # it runs on, inside the timed region (gm_frame_result.normals_kernel_ms)
    k_ms = float(np.mean([r["normals_kernel_ms"] for r in results])) if results else 0.0
    n_crop = float(np.mean([r["n_cropped"] for r in results])) if results else 0.0
    algo_bytes = 28.0 * n_crop                      # SURVEY par. 8d: normals = 12 N' read + 16 N' written
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic = None
    prof = os.path.join(ROOT, "profiles", "r01_normals_pmc.json")
    if os.path.exists(prof) and mode == "frames":
        try:
            with open(prof) as f:
                pj = json.load(f)
            if pj.get("points") == n and abs(pj.get("radius", 0) - radius) < 1e-9:
                traffic = pj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # the kernel's real bound: fp32 VALU issue.  Busy fraction from committed PMC counters (same workload); the
    # live part is the lane-operation rate implied by the instruction count of that profile and THIS run's duration.
    valu = None
    vprof = os.path.join(ROOT, "profiles", "r01_normals_valu_pmc.json")
    if os.path.exists(vprof) and traffic is not None:
        try:
            with open(vprof) as f:
                vj = json.load(f)
            insts = float(vj["counters"]["SQ_INSTS_VALU"])
            valu = {"valu_busy_frac_pmc": vj["derived"]["valu_busy_frac"], "SQ_INSTS_VALU_per_launch": insts,
                    "wave_instr_per_s": insts / (k_ms_excl * 1e-3) if k_ms_excl else None,
                    "peak_wave_instr_per_s_packed": 1024 * 2.4e9 / 4.0,
                    "source": "profiles/r01_normals_valu_pmc.json, profiles/r01_valu_rate_probe.txt"}
        except Exception:
            valu = None

    if rank == 0:
        out = {
            "metric": "points/sec full segment+fit, 1M-pt synthetic tunnel frame",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if mode == "frames" else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-pt synthetic tunnel frame (R=2 m, L=12 m, sigma=0.01), rows resident in HBM",
                       "neighborRadius": radius, "k_regime": "fixed-k (~256 neighbours)", "boxFilterBound": bound,
                       "voxelGridLeafSize": leaf, "weightingFactor": wf, "mode": mode, "frames_in_flight": n_slots,
                       "ransac_model": "cylinder, H=1024, tau=0.03 (extension)" if ransac_on else "none (reference-faithful path)",
                       "parallelism": f"{mode}x{world}", "collective_backend": args.dist_backend if world > 1 else None},
            "roofline": {"kernel": "k_normals", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": k_ms, "avg_launch_ms_exclusive": k_ms_excl,
                         "algorithmic_bytes_per_launch": algo_bytes, "valu": valu,
                         "note": "VALU-bound neighbour loop (k~256): HBM fraction is reported as the contract asks, see "
                                 "DESIGN.md par. 4 for its VALU roofline. avg_launch_ms is hipEvent-bracketed inside the timed "
                                 "region, where the kernel shares the chip with the other frames in flight; "
                                 "avg_launch_ms_exclusive (frames one at a time) is the figure rocprofv3's kernel stats show"},
            "whole_path_hbm": {"algorithmic_bytes_per_point": ALGO_BYTES_PER_POINT,
                               "achieved_GBs": value * ALGO_BYTES_PER_POINT / 1e9,
                               "frac_of_spec": value * ALGO_BYTES_PER_POINT / 1e9 / HBM_PEAK_GBS},
            "stage_ms_last_frame": {k: round(v, 4) for k, v in results[-1]["stage_ms"].items()} if results else {},
        }
        out.update(secondary)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, radius, args.cpu_threads)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
