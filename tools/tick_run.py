"""gv_tick at config-3 size on the scene with objects, for a kernel trace:
   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/tick_run.py [pca|vision] [ticks]
Prints host-observed ms per tick."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

branch = sys.argv[1] if len(sys.argv) > 1 else "pca"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = synth.CONFIGS[3]["grid"]
tfs = synth.transforms(True)
x, y, z, b = synth.scene_with_objects(tfs)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
st, dy = gvamd.filter_bboxes(b)
net = synth.network_outputs(len(dy))
pin = gvamd.PinnedI8(h.G)
kw = dict(k_near=4, vision=branch == "vision", net=net if branch == "vision" else None, grid_out=pin.array)
for _ in range(3):
    r = h.tick(b, **kw)
ts = []
for _ in range(ticks):
    t0 = time.perf_counter()
    r = h.tick(b, **kw)
    ts.append((time.perf_counter() - t0) * 1e3)
print(branch, "tick ms median", round(float(np.median(ts)), 4), "min", round(min(ts), 4), "poses", len(r["poses"]), "depths", len(r["depths"]))
ids = h.bbox_id() if branch == "pca" else None
if ids is not None:
    cnt = np.bincount(ids[ids >= 0], minlength=len(b))
    print("points per bbox: total", int(cnt.sum()), "max", int(cnt.max()), "median", int(np.median(cnt)))
pin.close()
h.close()
