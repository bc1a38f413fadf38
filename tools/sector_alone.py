"""The sector kernel alone (serial frame, stage timing) on the uniform and the lidar-like cloud: python3 tools/sector_alone.py [diag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "diag" in sys.argv:
    os.environ.setdefault("GV_LIB_AB", os.path.join(ROOT, "tools", "_diag", "libgv_diag.so"))
os.environ["GV_PIPELINE"] = "0"
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import gvamd
from gvamd import synth
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
for name, cloud in (("uniform", synth.cloud_uniform), ("lidar_like", synth.cloud_lidar_like)):
    x, y, z, _ = cloud(config)
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH)
    for _ in range(50):
        h.enqueue_frame()
    h.synchronize()
    st = sorted(h.time_frame_stages(30)["ray_march"] for _ in range(5))
    print(f"{name:10s} sectors alone: min {st[0]*1e3:6.2f} us  median {st[2]*1e3:6.2f} us", flush=True)
    h.close()
