"""Sweep of the sector kernel's tail policy (GV_MARCH_LIMIT, GV_FLAT_K; read at gv_create): the kernel alone
(stage timing) and the pipelined frame rate, config 3.  python3 tools/sector_sweep.py [lidar]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = (synth.cloud_lidar_like if "lidar" in sys.argv else synth.cloud_uniform)(config)
bb, pp = synth.detections(config), synth.lshape_poses(config)
flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
ref = None
for ml, fk in ((32768, 8), (65536, 8), (150000, 8), (400000, 8), (32768, 4), (150000, 4), (150000, 16), (400000, 16)):
    os.environ["GV_MARCH_LIMIT"] = str(ml)
    os.environ["GV_FLAT_K"] = str(fk)
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    h.set_detections(flags, bboxes=bb, poses=pp)
    for _ in range(300):
        h.enqueue_frame()
    h.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(400):
            h.enqueue_frame()
        h.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
    st = h.time_frame_stages(30)
    m = h.miss()
    if ref is None:
        ref = m
    print(f"march_limit {ml:7d} flat_k {fk:2d}: sectors alone {st['ray_march']*1e3:6.1f} us, pipelined frame {best:6.1f} us, miss equal {np.array_equal(m, ref)}")
    h.close()
