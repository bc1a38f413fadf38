"""Device timeline of the pipelined frames (timing events around every kernel; diagnostic)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GV_LIB_AB", os.path.join(ROOT, "tools", "_diag", "libgv_diag.so"))   # -DGV_DIAG build: tools/build_diag.sh
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
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=synth.detections(config), poses=synth.lshape_poses(config))
for _ in range(30):
    h.enqueue_frame()
h.synchronize()
F = 40
out = np.zeros(F * 10, np.float32)
rc = h._lib.gv_debug_pipeline_trace(h._h, C.c_int32(F), out.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = out.reshape(F, 5, 2)
names = ["rects", "binning", "-", "sectors", "gridpass"]
print("frame period (sectors start to start):", np.diff(t[10:, 3, 0]).mean().round(1), "us")
print("mean durations:", {n: round(float((t[10:, k, 1] - t[10:, k, 0]).mean()), 1) for k, n in enumerate(names)})
b = t[20, 0, 0]
for f in range(20, 24):
    print(f"frame {f}: " + "  ".join(f"{n} {t[f,k,0]-b:7.1f}-{t[f,k,1]-b:7.1f}" for k, n in enumerate(names)))
print("gap bitmaps(f) end -> sectors(f) start:", (t[10:, 3, 0] - t[10:, 2, 1]).mean().round(1), " sectors(f) end -> gridpass(f) start:", (t[10:, 4, 0] - t[10:, 3, 1]).mean().round(1),
      " sectors(f-1) end -> sectors(f) start:", (t[11:, 3, 0] - t[10:-1, 3, 1]).mean().round(1), " gridpass(f-3) end -> rects(f) start:", (t[13:, 0, 0] - t[10:-3, 4, 1]).mean().round(1))
h.close()
