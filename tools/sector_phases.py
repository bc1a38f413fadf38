"""Diagnostic: per-phase shader-clock cycles of the sector kernel (GV_SECTOR_DBG=1)."""
import ctypes as C, os, sys
os.environ["GV_SECTOR_DBG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
cloud = synth.cloud_lidar_like if "lidar" in sys.argv else synth.cloud_uniform
x, y, z, _ = cloud(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH)
for _ in range(3):
    h.enqueue_frame()
h.synchronize()
log2s = int(os.environ.get("GV_LOG2S", "8"))
nwg = 8 << log2s
buf = np.zeros((nwg, 16), np.uint64)
rc = h._lib.gv_debug_sector_stamps(h._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(nwg))
assert rc == 0, rc
names = ["init", "scan", "stage", "cnt+pfx", "place", "rmq", "thresh", "compact", "march", "edge", "flush"]  # stamps 0..11
t = buf.astype(np.int64)
NS = len(names)
full = t[:, NS] > 0
print("workgroups", nwg, "with ends", int(full.sum()))
d = np.diff(t[full][:, :NS + 1], axis=1)
for k, nme in enumerate(names):
    print(f"{nme:8s} mean {d[:, k].mean():9.0f}  p50 {np.median(d[:, k]):9.0f}  p99 {np.percentile(d[:, k], 99):9.0f}  max {d[:, k].max():9.0f} cycles")
tot = t[full][:, NS] - t[full][:, 0]
print(f"total    mean {tot.mean():9.0f}  max {tot.max():9.0f} cycles ; kernel span {(t[full][:, NS].max() - t[full][:, 0].min())} ticks")
print("stage", h.time_frame_stages(20))
