"""Soak of the synchronous calls' result block (CallDone: the last kernel stores its results into pinned host memory
and publishes a sequence word the host spins on): thousands of kNN / ground-plane / computeBBoxPose calls on a
lidar-like cloud with frames in flight on the lanes beside them; every call's results must equal the first call's of
its kind, bit for bit -- a payload read before it was complete would differ.  python3 tools/result_block_soak.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = synth.cloud_lidar_like(config, 200_000)
bboxes = synth.detections(config)
poses = synth.lshape_poses(config)
flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(flags, bboxes=bboxes, poses=poses)


def snap(kind, k):
    if kind == 0:
        d, d2 = h.compute_depth_for_bboxes(bboxes, k)
        return d.tobytes() + d2.tobytes()
    if kind == 1:
        m, mask, coeff = h.segment_ground_plane()
        return np.int64(m).tobytes() + np.asarray(coeff).tobytes() + mask.tobytes()
    if kind == 2:
        p, v = h.compute_bbox_pose(bboxes)
        return p.tobytes() + v.tobytes()
    p, v, n = h.compute_bbox_pose_ground_removed(bboxes)
    return p.tobytes() + v.tobytes() + np.int64(n).tobytes()


ref = {}
bad = 0
t0 = time.time()
for r in range(rounds):
    for _ in range(3):
        h.enqueue_frame()   # frames in flight on the lanes while the calls run on the public stream
    for kind, k in ((0, 10), (0, 32), (1, 0), (2, 0), (3, 0)):
        s = snap(kind, k)
        key = (kind, k)
        if key not in ref:
            ref[key] = s
        elif s != ref[key]:
            bad += 1
            print("MISMATCH round", r, "call", key)
    if (r + 1) % 500 == 0:
        print(f"round {r + 1}: {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
h.synchronize()
h.close()
print("calls", rounds * 5, "mismatches", bad)
sys.exit(1 if bad else 0)
