"""Diagnostic build only: the sector kernel alone with parts of it switched off (GV_ABLATE bits; results are wrong
by construction, only the time is read).  No stamps (GV_SECTOR_DBG unset).  python3 tools/sector_ablate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GV_LIB_AB", os.path.join(ROOT, "tools", "_diag", "libgv_diag.so"))
os.environ["GV_PIPELINE"] = "0"
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import gvamd
from gvamd import synth
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = (synth.cloud_lidar_like if "lidar" in sys.argv else synth.cloud_uniform)(config)
names = {0: "nothing off", 4096: "no overflow-list scans", 8: "launch + init only", 16: "+ scan", 32: "+ nothing else (return before groups)", 2: "no gather loop", 1024: "no boundary walks",
         2048: "no exact cells (todo)", 3072: "no walks, no exact cells", 4: "no flush", 256: "no marched tail (flat instead)",
         2 | 512 | 256: "no gather, no tails", 2 | 512 | 256 | 4: "no gather, no tails, no flush"}
for abl, nm in names.items():
    os.environ["GV_ABLATE"] = str(abl)
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH)
    for _ in range(50):
        h.enqueue_frame()
    h.synchronize()
    st = min(h.time_frame_stages(30)["ray_march"] for _ in range(3))
    print(f"ablate {abl:5d} ({nm:40s}): sectors alone {st*1e3:6.1f} us", flush=True)
    h.close()
