"""Stage times of the points pass under different frame flags (GPU experiment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import gvamd
from gvamd import synth

config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
bboxes, poses = synth.detections(config), synth.lshape_poses(config)
for cloud_name, fn in (("uniform", synth.cloud_uniform), ("lidar", synth.cloud_lidar_like)):
    x, y, z, _ = fn(config)
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    for name, flags, nb in [("bin", gvamd.FRAME_BIN, 50), ("bin+ray", gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH, 50),
                            ("bbox50", gvamd.FRAME_BBOX_TEST, 50), ("bbox5", gvamd.FRAME_BBOX_TEST, 5),
                            ("bbox0", gvamd.FRAME_BBOX_TEST, 0),
                            ("all", gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, 50),
                            ("all+cell", gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_CELL_IDX, 50)]:
        h.set_detections(flags, bboxes=bboxes[:nb], poses=poses)
        for _ in range(5):
            h.enqueue_frame()
        h.synchronize()
        st = h.time_frame_stages(30)
        print(f"{cloud_name:8s} {name:9s} points {st['points']*1e3:7.1f} us   (frame {sum(st.values())*1e3:7.1f} us)")
    h.close()
