"""Multi-rank semantics of the sharded frame on CPU: world_size 2 and 3, gloo.

This replays the algorithm the product runs (gv_api.hip: enqueue_frame_sharded, gv_shard.hip), step by step, with
the oracle's pieces and torch.distributed in place of the HIP kernels and RCCL:
  1. every rank turns ITS slice of the points into ray-END BITMAPS (hit ends, clipped ends) and partial hit counts;
  2. exchange 1: the bitmaps are cut into `world` equal slices (the product's gv_shard_slice_words), slice q goes to
     rank q (send / recv pairs), is OR-ed there and all-gathered -- every rank holds the complete end bitmaps;
  3. every rank marches every world-th ray of the dispatch order into a partial FREE-CELL bitmap;
  4. exchange 2: the free-cell bitmaps are packed by row band -- whole 64-row blocks, the product's own
     gv_shard_band_rows, called through the C ABI -- band q goes to rank q and is OR-ed there;
  5. rank q runs the grid pass on band q; the packed int8 bands are broadcast from their owners;
  6. GV_FRAME_KEEP_COUNTS: the hit counts are reduced band by band onto the band's owner.
OR and integer sums commute, so the union of the bands must equal the single-rank oracle frame bit for bit, for a
grid whose rows do not divide into equal bands as well.  RCCL itself only runs with world = 1 on the one-GPU box
(tests/test_gpu_parity.py::test_sharded_frame_world1_matches_plain); every (rank, world) of the PRODUCT code runs on
one device in test_sharded_frame_every_rank_emulated."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _pack(bits):
    """bool[G] -> uint32 words (padded with zero bits)"""
    return np.packbits(np.concatenate([bits, np.zeros((-len(bits)) % 32, bool)]), bitorder="little").view(np.uint32).copy()


def _unpack(words, n):
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)


def _exchange_slices(dist, torch, rank, world, send_slices):
    """slice q of every rank ends up at rank q (the product's grouped ncclSend / ncclRecv)"""
    recv = [torch.zeros_like(send_slices[0]) for _ in range(world)]
    reqs = []
    for q in range(world):
        if q == rank:
            recv[q].copy_(send_slices[q])
            continue
        reqs.append(dist.isend(send_slices[q], dst=q))
        reqs.append(dist.irecv(recv[q], src=q))
    for r in reqs:
        r.wait()
    return recv


def _worker(rank, world, port, tmpdir, grid, n_pts):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "grid-vision_amd"))
    import torch
    import torch.distributed as dist
    import oracle_lib as ol
    import gvamd
    from gvamd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tfs = synth.transforms(True)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    poses = synth.lshape_poses(1, 12)
    og = ol.OGrid(*grid)
    ny, nx, G = og.ny, og.nx, og.G
    bands = [gvamd.shard_band_rows(q, world, ny) for q in range(world)]   # the product's band function
    y0, y1 = bands[rank]
    for frame in range(3):
        x, y, z, _ = synth.cloud_uniform(1, n_pts, seed_extra=frame)
        n = len(x)
        lo, hi = n * rank // world, n * (rank + 1) // world        # contiguous N/world slice
        # 1. private end bitmaps + partial counts of this rank's points
        hits, _ = og.bin_points(m_base, x[lo:hi], y[lo:hi], z[lo:hi])
        kind, ex, ey = og.ray_ends(m_base, x[lo:hi], y[lo:hi], z[lo:hi])
        cell = ey.astype(np.int64) * nx + ex
        hit_b, clip_b = np.zeros(G, bool), np.zeros(G, bool)
        hit_b[cell[kind == 1]] = True
        clip_b[cell[kind == 2]] = True
        ends = np.concatenate([_pack(hit_b), _pack(clip_b)])
        # 2. exchange 1: equal slices, OR, all-gather
        slice_w = gvamd.shard_slice_words(len(ends), world)
        padded = np.zeros(slice_w * world, np.uint32)
        padded[:len(ends)] = ends
        t = torch.from_numpy(padded.view(np.int32))
        got = _exchange_slices(dist, torch, rank, world, [t[q * slice_w:(q + 1) * slice_w].clone() for q in range(world)])
        mine = np.bitwise_or.reduce(np.stack([g_.numpy().view(np.uint32) for g_ in got]), axis=0)
        parts = [torch.zeros(slice_w, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine.view(np.int32)))
        full = np.concatenate([p.numpy().view(np.uint32) for p in parts])[:len(ends)]
        w1 = len(_pack(hit_b))
        all_hit, all_clip = _unpack(full[:w1], G), _unpack(full[w1:], G)
        # 3. this rank's share of the ray stage: every world-th ray of the (sorted) dispatch order
        e_hit, e_clip = np.nonzero(all_hit)[0], np.nonzero(all_clip)[0]
        cells = np.concatenate([e_hit, e_clip])
        kinds = np.concatenate([np.full(len(e_hit), 1, np.uint8), np.full(len(e_clip), 2, np.uint8)])
        share = slice(rank, None, world)
        free = og.march_ends(m_base, (cells[share] % nx).astype(np.int32), (cells[share] // nx).astype(np.int32), kinds[share]).astype(bool)
        # 4. exchange 2: free cells packed by band (whole 64-row blocks), band q to rank q, OR
        chunk = max((b1 - b0) for b0, b1 in bands) * nx
        chunk_w = (chunk + 31) // 32
        send = []
        for b0, b1 in bands:
            w = np.zeros(chunk_w, np.uint32)
            pk = _pack(free[b0 * nx:b1 * nx])
            w[:len(pk)] = pk
            send.append(torch.from_numpy(w.view(np.int32)))
        got = _exchange_slices(dist, torch, rank, world, send)
        band_free = _unpack(np.bitwise_or.reduce(np.stack([g_.numpy().view(np.uint32) for g_ in got]), axis=0), (y1 - y0) * nx)
        # 6. counts: reduced band by band onto the owner (grouped ncclReduce / ncclReduceScatter in the product)
        for q, (b0, b1) in enumerate(bands):
            if b1 > b0:
                tb = torch.from_numpy(hits[b0 * nx:b1 * nx].copy())
                dist.reduce(tb, dst=q, op=dist.ReduceOp.SUM)
                if q == rank:
                    band_hits = tb.numpy()
        # 5. grid pass on my band (the hit rule reads the OR-ed hit bitmap), int8 bands broadcast from their owners
        h_full = np.zeros(G, np.int32)
        m_full = np.zeros(G, np.uint8)
        h_full[y0 * nx:y1 * nx] = all_hit[y0 * nx:y1 * nx]
        m_full[y0 * nx:y1 * nx] = band_free
        og.frame_update(poses, h_full, m_full)
        data, _ = og.to_occupancy_grid()
        gathered = np.zeros(G, np.int8)
        for q, (b0, b1) in enumerate(bands):
            if b1 <= b0:
                continue
            tb = torch.from_numpy(data[G - b1 * nx: G - b0 * nx].copy())   # band q sits reversed in OccupancyGrid.data
            dist.broadcast(tb, src=q)
            gathered[G - b1 * nx: G - b0 * nx] = tb.numpy()
        np.save(os.path.join(tmpdir, f"lo_{frame}_{rank}.npy"), og.log_odds[y0 * nx:y1 * nx].copy())
        if y1 > y0:
            np.save(os.path.join(tmpdir, f"hits_{frame}_{rank}.npy"), band_hits)
        if rank == 0:
            np.save(os.path.join(tmpdir, f"gathered_{frame}.npy"), gathered)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,grid,n_pts", [(2, (100, 100, 0.5), 6000), (3, (120, 200, 0.25), 5000)])
def test_sharded_frame_gloo(tmp_path, world, grid, n_pts):
    import torch.multiprocessing as mp
    import oracle_lib as ol
    import gvamd
    from gvamd import synth

    port = 29500 + (os.getpid() * 7 + world) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), grid, n_pts), nprocs=world, join=True)
    tfs = synth.transforms(True)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    poses = synth.lshape_poses(1, 12)
    og = ol.OGrid(*grid)
    nx, ny = og.nx, og.ny
    bands = [gvamd.shard_band_rows(q, world, ny) for q in range(world)]
    assert bands[0][0] == 0 and bands[-1][1] == ny and all(bands[q][1] == bands[q + 1][0] for q in range(world - 1))
    assert all(b0 % 64 == 0 for b0, _ in bands)
    if world == 3:
        assert len({b1 - b0 for b0, b1 in bands}) > 1   # bands of different length: the grouped-reduce form
    for frame in range(3):
        x, y, z, _ = synth.cloud_uniform(1, n_pts, seed_extra=frame)
        hits, _ = og.bin_points(m_base, x, y, z)
        miss, _ = og.raymarch(m_base, x, y, z)
        og.frame_update(poses, hits, miss)
        data, _ = og.to_occupancy_grid()
        assert np.array_equal(np.load(tmp_path / f"gathered_{frame}.npy"), data)
        for q, (b0, b1) in enumerate(bands):
            assert np.array_equal(np.load(tmp_path / f"lo_{frame}_{q}.npy"), og.log_odds[b0 * nx:b1 * nx])
            if b1 > b0:
                assert np.array_equal(np.load(tmp_path / f"hits_{frame}_{q}.npy"), hits[b0 * nx:b1 * nx])
