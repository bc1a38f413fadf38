"""Solo time of the partition pass with its parts switched on and off (frame flags):
   python3 tools/part_ablate.py"""
import os, sys
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
B, R, X = gvamd.FRAME_BIN, gvamd.FRAME_RAYMARCH, gvamd.FRAME_BBOX_TEST
for name, fl in (("bin", B), ("bin+ray", B | R), ("bin+bbox", B | X), ("bin+ray+bbox", B | R | X)):
    h.set_detections(fl, bboxes=synth.detections(config), poses=synth.lshape_poses(config))
    h.time_frame_stages(5)
    st = h.time_frame_stages(30)
    print(f"{name:14s} points {st['points'] * 1e3:6.1f} us   ray_ends {st['ray_ends'] * 1e3:6.1f} us", flush=True)
h.close()
