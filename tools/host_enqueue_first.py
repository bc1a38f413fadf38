"""Host time of the first gv_frame_enqueue calls behind a synchronisation (nothing to wait for): what a launch of a frame
costs the calling thread.  python3 tools/host_enqueue_first.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = synth.cloud_uniform(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=synth.detections(config), poses=synth.lshape_poses(config))
for _ in range(100):
    h.enqueue_frame()
h.synchronize()
rows = []
for rep in range(200):
    ts = []
    for f in range(6):
        t0 = time.perf_counter()
        h.enqueue_frame()
        ts.append((time.perf_counter() - t0) * 1e6)
    h.synchronize()
    rows.append(ts)
a = np.array(rows)
print("host us per gv_frame_enqueue, calls 1..6 behind a synchronisation (median of 200):", np.round(np.median(a, axis=0), 1).tolist())
h.close()
