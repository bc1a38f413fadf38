"""Multi-rank semantics of the sharded frame on CPU: world_size 2, gloo.

Each rank bins + ray-marches ITS slice of the points with the oracle, the count grids are
reduced (int32 sum / uint8 max), each rank finalises its row band, bands are gathered.
The result must be bit-identical to the single-rank oracle frame.  This checks the
algorithm the RCCL path in libgridvision_hip.so implements (points partition + integer
reduce + band finalise + band gather); the RCCL calls themselves only run with world = 1
on the one-GPU box (tests/test_gpu_parity.py::test_sharded_frame_world1_matches_plain).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "grid-vision_amd"))
    import torch
    import torch.distributed as dist
    import oracle_lib as ol
    from gvamd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    config = 1
    g = synth.CONFIGS[config]["grid"]
    tfs = synth.transforms(True)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    poses = synth.lshape_poses(config, 12)
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    ny, nx = og.ny, og.nx
    for frame in range(3):
        x, y, z, _ = synth.cloud_uniform(config, 20_000, seed_extra=frame)
        n = len(x)
        lo, hi = n * rank // world, n * (rank + 1) // world        # contiguous N/world slice
        hits, _ = og.bin_points(m_base, x[lo:hi], y[lo:hi], z[lo:hi])
        miss, _ = og.raymarch(m_base, x[lo:hi], y[lo:hi], z[lo:hi])
        th = torch.from_numpy(hits)
        tm = torch.from_numpy(miss.astype(np.int32))
        dist.all_reduce(th, op=dist.ReduceOp.SUM)                   # reduce(-scatter) of the count grids
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        og.frame_update(poses, th.numpy(), tm.numpy().astype(np.uint8))
        data, _ = og.to_occupancy_grid()
        # band r = rows [ny*r/world, ny*(r+1)/world); packed band sits reversed in OccupancyGrid.data
        y0, y1 = ny * rank // world, ny * (rank + 1) // world
        G = nx * ny
        band = torch.from_numpy(data[G - y1 * nx: G - y0 * nx].copy())
        parts = [torch.zeros(((ny * (r + 1) // world) - (ny * r // world)) * nx, dtype=torch.int8) for r in range(world)]
        dist.all_gather(parts, band) if len({p.numel() for p in parts}) == 1 else None
        if rank == 0:
            np.save(os.path.join(tmpdir, f"gathered_{frame}.npy"),
                    np.concatenate([p.numpy() for p in reversed(parts)]))
            np.save(os.path.join(tmpdir, f"lo_{frame}.npy"), og.log_odds.copy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_frame_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    import oracle_lib as ol
    from gvamd import synth

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    config = 1
    g = synth.CONFIGS[config]["grid"]
    tfs = synth.transforms(True)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    poses = synth.lshape_poses(config, 12)
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    for frame in range(3):
        x, y, z, _ = synth.cloud_uniform(config, 20_000, seed_extra=frame)
        hits, _ = og.bin_points(m_base, x, y, z)
        miss, _ = og.raymarch(m_base, x, y, z)
        og.frame_update(poses, hits, miss)
        data, _ = og.to_occupancy_grid()
        assert np.array_equal(np.load(tmp_path / f"gathered_{frame}.npy"), data)
        assert np.array_equal(np.load(tmp_path / f"lo_{frame}.npy"), og.log_odds)
