"""Does the phase between the two lanes matter?  After a synchronize the first frame is enqueued alone, the host
spins for X microseconds, then 400 frames follow back to back; the lanes free-run from there (no event ties them).
Pipelined frame time per X, config 3.  python3 tools/lane_offset.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import gvamd
from gvamd import synth
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = (synth.cloud_lidar_like if "lidar" in sys.argv else synth.cloud_uniform)(config)
bb, pp = synth.detections(config), synth.lshape_poses(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=bb, poses=pp)
for _ in range(600):
    h.enqueue_frame()
h.synchronize()
for rnd in range(2):
    for X in (0, 10, 25, 40, 60, 90):
        res = []
        for rep in range(5):
            t0 = time.perf_counter()
            h.enqueue_frame()
            t1 = time.perf_counter()
            while (time.perf_counter() - t1) * 1e6 < X:
                pass
            for _ in range(399):
                h.enqueue_frame()
            h.synchronize()
            res.append((time.perf_counter() - t0) / 400 * 1e6)
        print(f"offset {X:3d} us: frame {min(res):5.1f} min {sorted(res)[2]:5.1f} median us", flush=True)
h.close()
