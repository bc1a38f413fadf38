"""A/B of library builds (kernel experiments): for every .so given, config 3 with the headline flags -- stage times
(each kernel alone), the pipelined frame, and a digest of the outputs (hits, bbox ids, miss, log-odds) so that a
faster build that computes something else shows.  Variants are built into tools/_ab/ by hand, e.g.
  GV_HIPCC_EXTRA="-DGV_PART_BATCH=2" python3 -c "..."; every variant runs in a process of its own (GV_LIB_AB).
python3 tools/lib_ab.py [lidar] lib1.so lib2.so ..."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import time
    sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
    import numpy as np
    import gvamd
    from gvamd import synth
    config = 3
    g = synth.CONFIGS[config]["grid"]
    tfs = synth.transforms(True)
    x, y, z, _ = (synth.cloud_lidar_like if os.environ.get("AB_CLOUD") == "lidar" else synth.cloud_uniform)(config)
    bb, pp = synth.detections(config), synth.lshape_poses(config)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    h.set_detections(flags, bboxes=bb, poses=pp)
    h.enqueue_frame()
    h.synchronize()
    dig = hashlib.sha1()
    for arr in (h.hits(), h.bbox_id(), h.miss(), h.log_odds()):
        dig.update(np.ascontiguousarray(arr).tobytes())
    for _ in range(400):
        h.enqueue_frame()
    h.synchronize()
    reps = []
    for rep in range(7):
        t0 = time.perf_counter()
        for _ in range(400):
            h.enqueue_frame()
        h.synchronize()
        reps.append((time.perf_counter() - t0) / 400 * 1e6)
    st = [h.time_frame_stages(30) for _ in range(3)]
    best = {k: min(s[k] for s in st) * 1e3 for k in st[0]}
    print(f"{os.path.basename(os.environ.get('GV_LIB_AB', 'shipped')):28s} points {best['points']:5.1f} ends {best['ray_ends']:5.1f} sectors {best['ray_march']:5.1f} "
          f"grid {best['finalize']:5.1f} us | pipelined frame min {min(reps):5.1f} median {sorted(reps)[3]:5.1f} us | outputs {dig.hexdigest()[:12]}", flush=True)
    h.close()


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child()
    else:
        args = [a for a in sys.argv[1:]]
        cloud = "lidar" if "lidar" in args else "uniform"
        libs = [a for a in args if a != "lidar"]
        for lib in libs:
            env = dict(os.environ, AB_CHILD="1", AB_CLOUD=cloud)
            if lib != "shipped":
                env["GV_LIB_AB"] = os.path.abspath(lib)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
