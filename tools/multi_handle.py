"""How much would frame-parallel streams buy?  K independent handles on ONE GPU, frames enqueued
round-robin from one host thread; aggregate frames/s against one handle.  GV_PIPELINE=0 makes every
handle a single in-order stream (partition, tiles, sectors, grid pass back to back).
   python3 tools/multi_handle.py K [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import gvamd
from gvamd import synth
config = 3
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 600
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = synth.cloud_uniform(config)
flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
hs = []
for k in range(K):
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    h.set_detections(flags, bboxes=synth.detections(config), poses=synth.lshape_poses(config))
    hs.append(h)
for rep in range(3):
    for f in range(60):
        hs[f % K].enqueue_frame()
    for h in hs:
        h.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        hs[f % K].enqueue_frame()
    th = time.perf_counter() - t0
    for h in hs:
        h.synchronize()
    dt = time.perf_counter() - t0
    print(f"handles {K} pipeline {os.environ.get('GV_PIPELINE', '1')}: {frames / dt:9.0f} frames/s  {dt / frames * 1e6:6.1f} us/frame  host {th / frames * 1e6:5.1f} us/frame", flush=True)
for h in hs:
    h.close()
