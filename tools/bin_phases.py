"""Diagnostic (needs the -DGV_DIAG build: tools/build_diag.sh, GV_LIB_AB=tools/_diag/libgv_diag.so):
per-phase shader-clock cycles of the binning kernels (GV_BIN_DBG=1)."""
import ctypes as C, os, sys
os.environ["GV_BIN_DBG"] = "1"
os.environ["GV_PIPELINE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GV_LIB_AB", os.path.join(ROOT, "tools", "_diag", "libgv_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

config = 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
cloud = synth.cloud_lidar_like if "lidar" in sys.argv else synth.cloud_uniform
x, y, z, _ = cloud(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=synth.detections(config),
                 poses=synth.lshape_poses(config))
for _ in range(5):
    h.enqueue_frame()
h.synchronize()
names = {0: ["points", "clip", "scan+row", "scatter", "write"],
         1: ["role", "zero LDS", "gather+hist", "hits out", "bitmaps"]}
for which, label in ((0, "k_bin_partition"), (1, "k_bin_tiles")):
    nwg = 8192
    buf = np.zeros((nwg, 16), np.uint64)
    rc = h._lib.gv_debug_bin_stamps(h._h, C.c_int(which), buf.ctypes.data_as(C.c_void_p), C.c_size_t(nwg))
    assert rc == 0, rc
    t = buf.astype(np.int64)
    live = t[:, 5] > 0
    t = t[live]
    print(label, "workgroups", int(live.sum()), "kernel span", int(t[:, 5].max() - t[:, 0].min()), "ticks; first start spread",
          int(t[:, 0].max() - t[:, 0].min()))
    d = np.diff(t[:, :6], axis=1)
    for k, nme in enumerate(names[which]):
        print(f"  {nme:10s} mean {d[:, k].mean():9.0f}  p50 {np.median(d[:, k]):9.0f}  max {d[:, k].max():9.0f}")
    tot = t[:, 5] - t[:, 0]
    print(f"  total      mean {tot.mean():9.0f}  max {tot.max():9.0f}")
    if "detail" in sys.argv:
        idx = np.nonzero(live)[0]
        order = np.argsort(-d[:, 2])[:14]
        print("  slowest in phase 2 (workgroup index: phase cycles):")
        for o in order:
            print(f"    wg {idx[o]:5d}: {d[o].tolist()}  start {int(t[o, 0] - t[:, 0].min())}  in-lane {int(t[o, 6] - t[o, 2])} list(w0) {int(t[o, 7] - t[o, 6])} wait {int(t[o, 3] - t[o, 7])}")
print("stages", h.time_frame_stages(20))
