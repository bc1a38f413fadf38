"""Native device timeline of the pipelined frame (diagnostic build, GV_TIMELINE=1): every workgroup of every launch
reports the constant-rate clock (s_memrealtime, 100 MHz) when it starts and ends; min / max per launch = the kernel's
residence on the device, with no marker packet in any queue and nothing intercepting the launches -- the partition
pass keeps its barrier-free launch.  python3 tools/native_timeline.py [lidar] [frames]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GV_TIMELINE"] = "1"
os.environ.setdefault("GV_LIB_AB", os.path.join(ROOT, "tools", "_diag", "libgv_diag.so"))   # tools/build_diag.sh
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

config = 3
frames = next((int(a) for a in sys.argv[1:] if a.isdigit()), 600)
lanes = 2 if os.environ.get("GV_LANES") == "2" else 3
g = synth.CONFIGS[config]["grid"]
tfs = synth.transforms(True)
x, y, z, _ = (synth.cloud_lidar_like if "lidar" in sys.argv else synth.cloud_uniform)(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
h.upload_xyz(x, y, z)
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=synth.detections(config), poses=synth.lshape_poses(config))
lib = h._lib
lib.gv_debug_frame_no.restype = C.c_uint64
for _ in range(600):
    h.enqueue_frame()
h.synchronize()
assert lib.gv_debug_timeline(h._h, None, C.c_size_t(0)) == 0
f0 = int(lib.gv_debug_frame_no(h._h))
for _ in range(frames):
    h.enqueue_frame()
h.synchronize()
buf = np.zeros((4096, 4, 2), np.uint64)
assert lib.gv_debug_timeline(h._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(4096)) == 0
idx = [(f0 + i) % 4096 for i in range(frames)]
t = buf[idx].astype(np.float64) * 0.01   # us (100 MHz)
t -= t[0, 0, 0]
names = ["partition", "tiles", "sectors", "grid pass"]
inner = slice(20, frames - 20)
print(f"{frames} frames, {lanes} lanes, native (diagnostic build: a few % slower than the shipped one); period per frame {((t[frames - 21, 3, 1] - t[20, 3, 1]) / (frames - 41)):.1f} us")
for k, nme in enumerate(names):
    d = t[inner, k, 1] - t[inner, k, 0]
    print(f"  {nme:10s} on the device: mean {d.mean():5.1f}  p10 {np.percentile(d, 10):5.1f}  p90 {np.percentile(d, 90):5.1f} us")
gap_pt = t[inner, 1, 0] - t[inner, 0, 1]
gap_ts = t[inner, 2, 0] - t[inner, 1, 1]
gap_sf = t[inner, 3, 0] - t[inner, 2, 1]
ov = t[20 + lanes:frames - 20, 0, 0] - t[20:frames - 20 - lanes, 2, 1]    # partition(f + lanes) start minus sectors(f) end: negative = overlap
print(f"  gaps: partition -> tiles {gap_pt.mean():.1f}, tiles -> sectors {gap_ts.mean():.1f}, sectors -> grid pass {gap_sf.mean():.1f} us; "
      f"partition(f+{lanes}) starts {(-ov).mean():.1f} us before sectors(f) ends (same lane; end = last of every eighth workgroup)")
lat = t[inner, 3, 1] - t[inner, 0, 0]
print(f"  frame latency first partition workgroup -> last grid-pass workgroup: mean {lat.mean():.1f} us")
# kernels resident over time
ev = []
for f in range(20, frames - 20):
    for k in range(4):
        ev.append((t[f, k, 0], 1)); ev.append((t[f, k, 1], -1))
ev.sort()
res, last, cur = {}, ev[0][0], 0
for tt, dlt in ev:
    res[cur] = res.get(cur, 0.0) + (tt - last)
    last, cur = tt, cur + dlt
tot = sum(res.values())
print("  kernels resident: " + ", ".join(f"{k}: {100 * v / tot:.0f} %" for k, v in sorted(res.items())))
print(f"  excerpt (us; lane = frame % {lanes}):")
rows = []
for f in range(100, 108):
    for k in range(4):
        rows.append((t[f, k, 0], t[f, k, 1], f, k))
for a, b, f, k in sorted(rows):
    print(f"    {a - t[100, 0, 0]:8.1f} .. {b - t[100, 0, 0]:8.1f}  frame {f} lane {f % lanes}  {names[k]}")
h.close()
