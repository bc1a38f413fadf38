"""Streaming frames (fresh pinned cloud + detections every frame) for a kernel-trace: run under
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/stream_run.py [frames] [pub]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth
config = 3
frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
pins, dets = [], []
for f in range(3):
    x, y, z, _ = synth.cloud_uniform(config, seed_extra=100 + f)
    n0 = len(x)
    if "split" in sys.argv:
        p3 = tuple(gvamd.PinnedF32(n0) for _ in range(3))
        p3[0].array[:], p3[1].array[:], p3[2].array[:] = x, y, z
        pins.append(tuple(p.array for p in p3) + (p3,))
    else:
        blk = gvamd.PinnedF32(3 * n0)   # one block, one DMA
        blk.array[:n0], blk.array[n0:2 * n0], blk.array[2 * n0:] = x, y, z
        pins.append((blk.array[:n0], blk.array[n0:2 * n0], blk.array[2 * n0:], blk))
    dets.append((synth.detections(config, seed_extra=f), synth.lshape_poses(config, seed_extra=f)))
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
pub = [gvamd.PinnedF32((h.G + 3) // 4) for _ in range(2)] if "pub" in sys.argv else None
per = []
done = 0
while done < frames:
    nchunk = min(100, frames - done)
    t0 = time.perf_counter()
    for f in range(done, done + nchunk):
        px, py, pz, _keep = pins[f % 3]
        h.upload_xyz_async(px, py, pz)
        h.set_detections_async(flags, bboxes=dets[f % 3][0], poses=dets[f % 3][1])
        h.enqueue_frame()
        if pub is not None:   # "pub": the packed grid goes back to pinned host memory for every frame:
            if "sched" in sys.argv:   # by a kernel on the public stream (gv_publish_grid_async)
                h.publish_grid_async(pub[f % 2].array.view("int8")[:h.G])
            else:                     # on the public stream right behind the grid pass (round 3)
                h.to_occupancy_grid_async(pub[f % 2].array.view("int8")[:h.G])
    if "nosync" not in sys.argv or done + nchunk >= frames:
        h.synchronize()
    per.append(round((time.perf_counter() - t0) / nchunk * 1e6, 1))
    done += nchunk
print("us/frame per chunk of 100:", per)
h.close()
