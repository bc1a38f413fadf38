"""The PCA / kNN path at config-3 size for a kernel trace:
   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/pca_run.py [uniform|lidar] [reps]
Prints host-observed ms per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

config = 3
which = sys.argv[1] if len(sys.argv) > 1 else "uniform"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = (synth.cloud_uniform if which == "uniform" else synth.cloud_lidar_like)(config)
bboxes = synth.detections(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
calls = {"depth_k10": lambda: h.compute_depth_for_bboxes(bboxes, 10),
         "bbox_pose": lambda: h.compute_bbox_pose(bboxes),
         "bbox_pose_ground_removed": lambda: h.compute_bbox_pose_ground_removed(bboxes)}
for name, call in calls.items():
    call()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = call()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(which, name, "ms median", round(float(np.median(ts)), 3), "min", round(min(ts), 3))
ids = h.bbox_id()
cnt = np.bincount(ids[ids >= 0], minlength=len(bboxes))
print("points per bbox: total", int(cnt.sum()), "max", int(cnt.max()), "median", int(np.median(cnt)))
_, valid, _ = h.compute_bbox_pose_ground_removed(bboxes)
print("valid poses", int(valid.sum()))
h.close()
