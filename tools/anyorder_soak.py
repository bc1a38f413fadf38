"""Soak of the overlapped partition pass: the same long frame sequence (config 3; detections swapped every 37 frames,
the cloud every 211) with GV_ANYORDER=1 and =0, digests of the layers and of the last frame's outputs every 1000
frames.  Every digest must agree between the two runs.  python3 tools/anyorder_soak.py [frames]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(frames):
    sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
    import numpy as np
    import gvamd
    from gvamd import synth
    config = 3
    g = synth.CONFIGS[config]["grid"]
    tfs = synth.transforms(True)
    clouds = [synth.cloud_uniform(config)[:3], synth.cloud_lidar_like(config)[:3]]
    dets = [(synth.detections(config, seed_extra=d), synth.lshape_poses(config, seed_extra=d)) for d in range(3)]
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(*clouds[0])
    h.set_detections(flags, bboxes=dets[0][0], poses=dets[0][1])
    out = []
    for f in range(frames):
        if f and f % 211 == 0:
            h.upload_xyz(*clouds[(f // 211) % 2])
        if f and f % 37 == 0:
            d = dets[(f // 37) % 3]
            h.set_detections_async(flags, bboxes=d[0], poses=d[1])
        h.enqueue_frame()
        if (f + 1) % 1000 == 0:
            h.synchronize()
            dig = hashlib.sha1()
            for arr in (h.log_odds(), h.hits(), h.miss(), h.bbox_id()):
                dig.update(np.ascontiguousarray(arr).tobytes())
            out.append(dig.hexdigest()[:16])
    h.synchronize()
    h.close()
    print(" ".join(out))


if __name__ == "__main__":
    if os.environ.get("SOAK_CHILD"):
        child(int(os.environ["SOAK_CHILD"]))
    else:
        frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
        res = {}
        for mode in ("1", "0"):
            env = dict(os.environ, SOAK_CHILD=str(frames), GV_ANYORDER=mode)
            res[mode] = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True).stdout.strip()
            print(f"GV_ANYORDER={mode}: {res[mode][:120]} ...")
        same = res["1"] == res["0"] and len(res["1"].split()) == frames // 1000
        print("digests agree:", same, f"({len(res['1'].split())} checkpoints)")
        sys.exit(0 if same else 1)
