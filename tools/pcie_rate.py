"""PCIe-inclusive frame rate: the caller hands over a host SoA cloud every frame
(gv_cloud_upload_xyz = pageable host memory -> HBM) before the frame is enqueued.
Reported in DESIGN.md; never the bench `value`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import gvamd
from gvamd import synth

config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = synth.cloud_uniform(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=synth.detections(config),
                 poses=synth.lshape_poses(config))
for _ in range(5):
    h.upload_xyz(x, y, z); h.enqueue_frame()
h.synchronize()
K = 100
t0 = time.perf_counter()
for _ in range(K):
    h.upload_xyz(x, y, z)
    h.enqueue_frame()
h.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
for _ in range(K):
    h.upload_xyz(x, y, z)
h.synchronize()
du = time.perf_counter() - t1
print(json.dumps({"pcie_inclusive_frames_per_s": K / dt, "ms_per_frame": dt / K * 1e3,
                  "upload_only_ms": du / K * 1e3, "upload_GBps": 12e6 * K / du / 1e9,
                  "note": "12 MB SoA cloud from pageable host memory per frame, synchronous upload then async frame"}))
