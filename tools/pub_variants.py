"""Where should the per-frame download of the packed grid run?  Streamed frames (fresh cloud every frame) with the
4 MB OccupancyGrid.data going back to pinned host memory (a) on the public stream (gv_to_occupancy_grid_async),
(b) on a stream of its own behind an event on the public stream.  us per frame per chunk of 100."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth
hip = C.CDLL("libamdhip64.so")
config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
pins, dets = [], []
for f in range(3):
    x, y, z, _ = synth.cloud_uniform(config, seed_extra=100 + f)
    n0 = len(x)
    blk = gvamd.PinnedF32(3 * n0)
    blk.array[:n0], blk.array[n0:2 * n0], blk.array[2 * n0:] = x, y, z
    pins.append((blk.array[:n0], blk.array[n0:2 * n0], blk.array[2 * n0:], blk))
    dets.append((synth.detections(config, seed_extra=f), synth.lshape_poses(config, seed_extra=f)))
flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
for variant in ("public", "own_stream", "public", "own_stream"):
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    pub = [gvamd.PinnedF32((h.G + 3) // 4) for _ in range(2)]
    occ, _, _ = h.device_layers()
    s2, evs = C.c_void_p(), [C.c_void_p() for _ in range(8)]
    assert hip.hipStreamCreateWithFlags(C.byref(s2), 1) == 0
    for e in evs:
        assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0
    public = C.c_void_p(h.stream())
    per = []
    fno = 0
    for chunk in range(8):
        t0 = time.perf_counter()
        for _ in range(100):
            px, py, pz, _k = pins[fno % 3]
            h.upload_xyz_async(px, py, pz)
            h.set_detections_async(flags, bboxes=dets[fno % 3][0], poses=dets[fno % 3][1])
            h.enqueue_frame()
            dst = pub[fno % 2].array.view("int8")[:h.G]
            if variant == "public":
                h.to_occupancy_grid_async(dst)
            else:
                e = evs[fno % 8]
                assert hip.hipEventRecord(e, public) == 0
                assert hip.hipStreamWaitEvent(s2, e, 0) == 0
                assert hip.hipMemcpyAsync(C.c_void_p(dst.ctypes.data), C.c_void_p(occ), C.c_size_t(h.G), 2, s2) == 0
            fno += 1
        h.synchronize()
        hip.hipStreamSynchronize(s2)
        per.append(round((time.perf_counter() - t0) / 100 * 1e6, 1))
    print(variant, per)
    hip.hipStreamDestroy(s2)
    h.close()
