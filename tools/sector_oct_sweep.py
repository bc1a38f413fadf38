"""Sweep of the sectors per octant (GV_LOG2S_OCT, read at gv_create) with the helper workgroups in place:
the sector kernel alone (stage timing) and the pipelined frame, config 3.  python3 tools/sector_oct_sweep.py"""
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
for spec in ("", "6,6,6,6,6,6,5,5", "6,6,6,6,7,7,6,6", "6,6,6,6,6,6,6,6", "7,7,7,7,7,7,5,5", "6,6,6,6,7,7,4,4", "6,6,6,6,6,6,4,4"):
    if spec:
        os.environ["GV_LOG2S_OCT"] = spec
    else:
        os.environ.pop("GV_LOG2S_OCT", None)
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
    print(f"log2s_oct {spec or 'default':18s}: sectors alone {st['ray_march']*1e3:6.1f} us, pipelined frame {best:6.1f} us, miss equal {np.array_equal(m, ref)}", flush=True)
    h.close()
