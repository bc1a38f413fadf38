"""gv_tick: host time of the enqueue and of the wait, separately (is the tick bound by the host's launches?)
python3 tools/tick_split.py [pca|vision] [ticks] [nogrid]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

branch = sys.argv[1] if len(sys.argv) > 1 else "pca"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nogrid = len(sys.argv) > 3
g = synth.CONFIGS[3]["grid"]
tfs = synth.transforms(True)
x, y, z, b = synth.scene_with_objects(tfs)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
st, dy = gvamd.filter_bboxes(b)
net = synth.network_outputs(len(dy))
pin = gvamd.PinnedI8(h.G)
kw = dict(k_near=4, vision=branch == "vision", net=net if branch == "vision" else None, grid_out=None if nogrid else pin.array)
for _ in range(5):
    h.tick(b, **kw)
te, tw = [], []
for _ in range(ticks):
    t0 = time.perf_counter()
    h.tick_enqueue(b, **kw)
    t1 = time.perf_counter()
    h.tick_wait()
    t2 = time.perf_counter()
    te.append((t1 - t0) * 1e6); tw.append((t2 - t1) * 1e6)
print(f"{branch}{' (no grid copy)' if nogrid else ''}: enqueue median {np.median(te):.1f} us (min {min(te):.1f}), wait median {np.median(tw):.1f} us, "
      f"total median {np.median(np.array(te) + np.array(tw)):.1f} us")
pin.close()
h.close()
