"""Host-side cost of gv_frame_enqueue (is the pipelined bench host-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import gvamd
from gvamd import synth
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
for cloud in (synth.cloud_uniform, synth.cloud_lidar_like):
    x, y, z, _ = cloud(config)
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=synth.detections(config), poses=synth.lshape_poses(config))
    for _ in range(50):
        h.enqueue_frame()
    h.synchronize()
    K = 400
    t0 = time.perf_counter()
    for _ in range(K):
        h.enqueue_frame()
    t1 = time.perf_counter()
    h.synchronize()
    t2 = time.perf_counter()
    print(cloud.__name__, f"host enqueue {1e6*(t1-t0)/K:.1f} us/frame, total {1e6*(t2-t0)/K:.1f} us/frame")
    h.close()
