#!/usr/bin/env python3
"""bench.py -- frames/sec of the grid-vision per-frame hot path on MI355X.

A "step" is one frame: points pass (transform + bin + ray ends + bbox test),
Bresenham free-space ray-march, and the grid pass (decay, rectangles, hit/miss,
clamp, sigmoid, int8 pack) over one synthetic cloud that is already resident in
HBM when the timed region starts.

N=1   BASELINE.json configs[2]: 1M-point cloud, 2000x2000 @ 0.1 m grid, 50
      bboxes + 50 poses.
N>1   configs[3]: one independent 1M-point frame stream per GPU (no data-path
      collective; torch.distributed only for the barrier and the max-over-ranks).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=0, help="0 = 3 at N=1 / 4 at N>1")
    ap.add_argument("--cloud", choices=["uniform", "lidar"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hit-counts", action="store_true", help="skip the second measurement with int32 hit counts (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def cpu_baseline(config, cloud_fn, tfs, bboxes, poses, budget_s):
    """The CPU oracle (a port: the reference itself cannot be built here) timed on
    this box's host cores, single thread like the reference node, on a bounded
    sample: whole frames of the same workload until ~budget_s have elapsed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from gvamd import synth
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    x, y, z, _ = cloud_fn(config)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)

    def frame():
        hits, _ = og.bin_points(m_base, x, y, z)
        miss, _ = og.raymarch(m_base, x, y, z, dedupe=False)   # per-point march, like a LineIterator user
        cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
        ol.extract_cloud_per_bbox(K, cx, cy, cz, bboxes, synth.IMG_W, synth.IMG_H)
        og.frame_update(poses, hits, miss)
        og.to_occupancy_grid()

    frame()   # warm-up (page-in)
    t0 = time.perf_counter()
    n = 0
    while True:
        frame()
        n += 1
        if time.perf_counter() - t0 >= budget_s or n >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} whole frames of the same workload (per-point Bresenham, no dedupe), {dt:.1f} s"}


def cpu_baseline_all_cores(config, cloud_fn, tfs, bboxes, poses, budget_s):
    """the same oracle on every host core.  The points are split into one chunk per
    thread (ctypes releases the GIL inside the C calls); each thread bins, ray-marches and bbox-tests
    its chunk into private grids, which are then summed / OR-ed and finalised once."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from gvamd import synth
    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32))
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    workers = [ol.OGrid(g.grid_x, g.grid_y, g.resolution) for _ in range(cores)]   # geometry holders
    x, y, z, _ = cloud_fn(config)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    cuts = [len(x) * k // cores for k in range(cores + 1)]

    def chunk(k):
        sl = slice(cuts[k], cuts[k + 1])
        w = workers[k]
        hits, _ = w.bin_points(m_base, x[sl], y[sl], z[sl])
        miss, _ = w.raymarch(m_base, x[sl], y[sl], z[sl], dedupe=False)
        cx, cy, cz = ol.transform_cloud(m_cam, x[sl], y[sl], z[sl])
        ol.extract_cloud_per_bbox(K, cx, cy, cz, bboxes, synth.IMG_W, synth.IMG_H)
        return hits, miss

    pool = ThreadPoolExecutor(cores)

    def frame():
        parts = list(pool.map(chunk, range(cores)))
        hits, miss = parts[0]
        for h2, m2 in parts[1:]:
            hits = hits + h2
            miss = np.maximum(miss, m2)
        og.frame_update(poses, hits, miss)
        og.to_occupancy_grid()

    frame()
    t0 = time.perf_counter()
    n = 0
    while True:
        frame()
        n += 1
        if time.perf_counter() - t0 >= budget_s or n >= 100:
            break
    dt = time.perf_counter() - t0
    pool.shutdown()
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} whole frames, points split over {cores} threads with private grids, reduced and finalised once, {dt:.1f} s"}


def measured_copy_rate(torch):
    """device-to-device copy rate of this box (read + write bytes / time): the second peak of SURVEY 8(d)"""
    n = 256 << 20
    a_ = torch.empty(n, dtype=torch.uint8, device="cuda")
    b_ = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        b_.copy_(a_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b_.copy_(a_)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    del a_, b_
    return 2.0 * n / (ms * 1e-3) / 1e9


def main():
    a = parse()
    # RCCL (and some HIP runtime paths) print banners on stdout; the contract is ONE JSON line
    # there, so everything incidental goes to stderr and the result line to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = max(a.gpus, 1)
    import torch
    dist = None
    ndev = torch.cuda.device_count()
    on_gpu_collectives = ndev >= world
    if world > 1:
        import torch.distributed as dist
        if on_gpu_collectives:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL
        else:
            # rehearsal on a box with fewer GPUs than ranks: ranks share devices, gloo for the barrier
            local_rank = local_rank % max(ndev, 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(0)
        local_rank = 0

    import gvamd
    from gvamd import synth
    config = a.config or (3 if n_gpus == 1 else 4)
    cfg = synth.CONFIGS[config]
    g = cfg["grid"]
    cloud_fn = synth.cloud_uniform if a.cloud == "uniform" else synth.cloud_lidar_like
    tfs = synth.transforms(perturbed=True)
    bboxes = synth.detections(config)
    poses = synth.lshape_poses(config)
    sharded = (config == 5)
    if sharded:
        # configs[4]: ONE 10M-point frame, points partitioned N/world per rank, RCCL reduce inside the library
        xa, ya, za, _ = synth.cloud_lidar_like(config, cfg["n"] // 2)
        xb, yb, zb, _ = synth.cloud_uniform(config, cfg["n"] - cfg["n"] // 2)
        xf, yf, zf = np.concatenate([xa, xb]), np.concatenate([ya, yb]), np.concatenate([za, zb])
        lo_i, hi_i = len(xf) * rank // world, len(xf) * (rank + 1) // world
        x, y, z = xf[lo_i:hi_i], yf[lo_i:hi_i], zf[lo_i:hi_i]
        N_total = len(xf)
    else:
        x, y, z, _ = cloud_fn(config, seed_extra=rank)
        N_total = len(x)
    N, G = len(x), g.nx * g.ny

    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution, device=local_rank)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)   # resident in HBM before the timed region
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    h.set_detections(flags, bboxes=bboxes, poses=poses)
    if sharded:
        uid = [gvamd.GridVisionHIP.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        h.comm_init(uid[0], rank, world)

    def barrier():
        h.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if sharded:
            h.process_frame_sharded(flags, bboxes=bboxes, poses=poses)   # synchronous, collective
        else:
            h.enqueue_frame()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu_collectives else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = a.steps if sharded else a.steps * world   # sharded: all ranks work on the same frame
    fps = frames / dt

    out = None
    if rank == 0:
        # per-kernel device time, HIP events on the handle's own stream (gv_time_frame_stages)
        stages = h.time_frame_stages(max(10, min(a.steps, 50)))
        n_rays, n_visits = h.ray_stats()
        frame_ms = sum(stages.values())
        bytes_frame = 12.0 * N + 13.0 * G
        # Algorithmic bytes per launch (DESIGN.md "Kernels"): SURVEY 8(d) counts 12 B/point
        # (x,y,z read) and 13 B/cell (log-odds r+w, occupancy w, int8 w); the count grids the
        # ray stage works on are implementation traffic there.  For the ray kernels the
        # minimum any implementation of that stage moves is used instead: end flags in
        # (2 bits/cell in both orientations = G/2 B) and one miss byte per cell out.
        alg = {"points": 12.0 * N, "ray_ends": 5.0 * G, "ray_march": 1.5 * G, "finalize": 13.0 * G,
               "detections": 120.0 * (len(bboxes) + len(poses))}
        kern = {"points": "k_bin_partition", "ray_ends": "k_bin_tiles", "ray_march": "k_ray_sectors",
                "finalize": "k_finalize_tiles", "detections": "k_rects_from_poses"}
        dom = max(stages, key=stages.get)
        kernels = [{"stage": k, "kernel": kern[k], "ms": stages[k], "algorithmic_bytes": alg[k],
                    "GBps": alg[k] / (stages[k] * 1e-3) / 1e9 if stages[k] > 0 else None,
                    "frac_hbm_peak": alg[k] / (stages[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS if stages[k] > 0 else None}
                   for k in stages]
        dom_s = stages[dom] * 1e-3
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(kern[dom])
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": kern[dom], "achieved": alg[dom] / dom_s / 1e9, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": alg[dom] / dom_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes": alg[dom], "kernel_ms": stages[dom],
                "note": "the dominant kernel is the sector ray-march: LDS/latency bound, not HBM bound; "
                        "see kernels[] for the HBM-bound passes (points 12N, finalize 13G)"}
        out = {
            "metric": f"frames/sec into grid ({N_total // 1000000}M-pt cloud / {g.nx}x{g.ny} @ {g.resolution} m grid)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "i32 hit counts / f64 cell index / f32 log-odds", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{config - 1}]: {N}-point {a.cloud} cloud per GPU, "
                                   f"{g.nx}x{g.ny} @ {g.resolution} m grid, {len(bboxes)} bboxes + {len(poses)} poses, "
                                   "bin (int32 hit counts) + ray-march + bbox test + grid pass",
                       "points": N, "cells": G, "parallelism": (f"points-sharded x{world} + RCCL reduce-scatter" if sharded else f"frame-per-gpu x{world}"),
                       "devices_visible": ndev},
            "mpoints_per_s": N_total * fps / 1e6,
            "frame_roofline": {"algorithmic_bytes": bytes_frame, "achieved_GBps": bytes_frame * fps / world / 1e9,
                               "frac_of_hbm_peak": bytes_frame * fps / world / 1e9 / HBM_PEAK_GBPS},
            "stage_ms": stages, "stage_ms_sum": frame_ms, "kernels": kernels,
            "ray_march": {"rays": n_rays, "equivalent_cell_visits": n_visits,
                          "mcell_visits_per_s": (n_visits / (stages["ray_march"] * 1e-3) / 1e6) if stages["ray_march"] > 0 else None},
            "roofline": roof,
        }
        try:
            copy_gbps = measured_copy_rate(torch)
            out["frame_roofline"]["measured_copy_GBps"] = copy_gbps
            out["frame_roofline"]["frac_of_measured_copy"] = bytes_frame * fps / world / 1e9 / copy_gbps
            roof["peak_measured_copy"] = copy_gbps
        except Exception as e:   # never fail the bench line over the auxiliary peak
            out["frame_roofline"]["measured_copy_GBps"] = None
            print("copy-rate measurement failed:", e, file=sys.stderr)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(config, cloud_fn, tfs, bboxes, poses, a.cpu_seconds)
            out["gpu_over_cpu"] = fps / out["cpu_baseline"]["value"]
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(config, cloud_fn, tfs, bboxes, poses, a.cpu_seconds)
            out["gpu_over_cpu_all_cores"] = fps / out["cpu_baseline_all_cores"]["value"]
        else:
            out["cpu_baseline"] = None
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
