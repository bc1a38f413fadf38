"""A/B of the sector helpers (GV_SECTOR_HELPERS=0/1): kernel alone and pipelined frame, config 3 + config 5."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth
for config, fn in ((3, synth.cloud_uniform), (3, synth.cloud_lidar_like), (5, synth.cloud_uniform)):
    g = synth.CONFIGS[config]["grid"]
    tfs = synth.transforms(True)
    x, y, z, _ = fn(config)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH
    ref = None
    for hv in ("0", "1"):
        os.environ["GV_SECTOR_HELPERS"] = hv
        h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
        h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
        h.upload_xyz(x, y, z)
        h.set_detections(flags)
        for _ in range(200):
            h.enqueue_frame()
        h.synchronize()
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            for _ in range(200):
                h.enqueue_frame()
            h.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        st = h.time_frame_stages(20)
        m = h.miss()
        if ref is None:
            ref = m
        print(f"config {config} {fn.__name__:16s} helpers {hv}: sectors alone {st['ray_march']*1e3:6.1f} us, pipelined frame {best:6.1f} us, miss equal {np.array_equal(m, ref)}")
        h.close()
