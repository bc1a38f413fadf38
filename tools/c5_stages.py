"""config-5 workload (10 M points, half lidar-like + half uniform, 4000 x 4000 grid) on one GPU: per-kernel times alone and
the pipelined frame; GV_LIB_AB selects another build.  python3 tools/c5_stages.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth
c5 = synth.CONFIGS[5]
g = c5["grid"]
tfs = synth.transforms(True)
n5 = c5["n"]
xa, ya, za, _ = synth.cloud_lidar_like(5, n5 // 2)
xb, yb, zb, _ = synth.cloud_uniform(5, n5 - n5 // 2)
x, y, z = np.concatenate([xa, xb]), np.concatenate([ya, yb]), np.concatenate([za, zb])
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH)
for _ in range(30):
    h.enqueue_frame()
h.synchronize()
reps = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(40):
        h.enqueue_frame()
    h.synchronize()
    reps.append((time.perf_counter() - t0) / 40 * 1e6)
st = [h.time_frame_stages(10) for _ in range(3)]
best = {k: min(s[k] for s in st) * 1e3 for k in st[0]}
print(f"{os.path.basename(os.environ.get('GV_LIB_AB', 'shipped')):20s} c5: points {best['points']:6.1f} ends {best['ray_ends']:6.1f} sectors {best['ray_march']:6.1f} grid {best['finalize']:6.1f} us | "
      f"pipelined frame min {min(reps):6.1f} median {sorted(reps)[2]:6.1f} us")
h.close()
