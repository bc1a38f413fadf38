"""Native device timeline (diagnostic build, GV_TIMELINE=1) of ONE 20-frame region that starts and ends with an empty
pipeline, as the driver's bench command times it: where the fill and the drain go.  python3 tools/native_timeline_k20.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GV_TIMELINE"] = "1"
os.environ.setdefault("GV_LIB_AB", os.path.join(ROOT, "tools", "_diag", "libgv_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

config, K = 3, 20
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = synth.cloud_uniform(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=synth.detections(config), poses=synth.lshape_poses(config))
lib = h._lib
lib.gv_debug_frame_no.restype = C.c_uint64
for rep in range(30):      # warm: regions as the bench runs them
    for _ in range(K):
        h.enqueue_frame()
    h.synchronize()
assert lib.gv_debug_timeline(h._h, None, C.c_size_t(0)) == 0
f0 = int(lib.gv_debug_frame_no(h._h))
t0 = time.perf_counter()
for _ in range(K):
    h.enqueue_frame()
t1 = time.perf_counter()
h.synchronize()
t2 = time.perf_counter()
buf = np.zeros((4096, 4, 2), np.uint64)
assert lib.gv_debug_timeline(h._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(4096)) == 0
idx = [(f0 + i) % 4096 for i in range(K)]
t = buf[idx].astype(np.float64) * 0.01
t -= t[0, 0, 0]
names = ["partition", "tiles", "sectors", "grid pass"]
print(f"host: enqueue of {K} frames {1e6 * (t1 - t0):.0f} us, region {1e6 * (t2 - t0):.0f} us = {1e6 * (t2 - t0) / K:.1f} us per frame")
print(f"device: first partition workgroup -> last grid-pass workgroup {t[K - 1, 3, 1]:.0f} us; grid passes end at", " ".join(f"{v:.0f}" for v in t[:, 3, 1]))
rows = []
for f in list(range(0, 4)) + list(range(K - 3, K)):
    for k in range(4):
        rows.append((t[f, k, 0], t[f, k, 1], f, k))
for a, b, f, k in sorted(rows):
    print(f"    {a:8.1f} .. {b:8.1f}  ({b - a:5.1f})  frame {f:2d} lane {f % 3}  {names[k]}")
h.close()
