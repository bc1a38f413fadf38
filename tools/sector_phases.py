"""Diagnostic: per-phase shader-clock cycles of the sector kernel (GV_SECTOR_DBG=1)."""
import ctypes as C, os, sys
os.environ["GV_SECTOR_DBG"] = "1"
os.environ["GV_PIPELINE"] = "0"   # one launch at a time writes the stamp buffer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GV_LIB_AB", os.path.join(ROOT, "tools", "_diag", "libgv_diag.so"))   # -DGV_DIAG build: tools/build_diag.sh
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
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH)
for _ in range(3):
    h.enqueue_frame()
h.synchronize()
nwg = 8 << 12   # the whole buffer: coarse stamps in the first half, fine ones (diag) from slot 16384 on
buf = np.zeros((nwg, 16), np.uint64)
rc = h._lib.gv_debug_sector_stamps(h._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(nwg))
assert rc == 0, rc
names = ["init", "scan", "append", "dense", "merge", "rmq", "thresh", "compact", "march", "edge", "flush"]  # stamps 0..11
t = buf.astype(np.int64)
NS = len(names)
full = t[:, NS] > 0
full[16384:] = False   # (fine stamps)
print("workgroups", nwg, "with ends", int(full.sum()))
d = np.diff(t[full][:, :NS + 1], axis=1)
for k, nme in enumerate(names):
    print(f"{nme:8s} mean {d[:, k].mean():9.0f}  p50 {np.median(d[:, k]):9.0f}  p99 {np.percentile(d[:, k], 99):9.0f}  max {d[:, k].max():9.0f} cycles")
tot = t[full][:, NS] - t[full][:, 0]
print(f"total    mean {tot.mean():9.0f}  max {tot.max():9.0f} cycles ; kernel span {(t[full][:, NS].max() - t[full][:, 0].min())} ticks")
print("stage", h.time_frame_stages(20))
if "detail" in sys.argv:
    order = np.argsort(-tot)[:12]
    t0 = t[full][:, 0].min()
    print("heaviest workgroups: wg octant sector start end | phases")
    for i in order:
        v12, v13 = int(buf[i, 12]), int(buf[i, 13])
        print(i, int(buf[i, 14]) >> 32 & 0xFF, int(buf[i, 14]) & 0xFFFFFFFF, "S", 1 << (int(buf[i, 14]) >> 40), "T", v12 >> 48, "maxreach", (v12 >> 32) & 0xFFFF, "imax", (v12 >> 16) & 0xFFFF, "n", v12 & 0xFFFF,
              "nlong", v13 & 0xFFFFFFFF, "tail_steps", v13 >> 32, "nover", int(buf[i, 15]) & 0xFFFFFFFF, "sorted", (int(buf[i, 15]) >> 32) & 1, "maxcnt", (int(buf[i, 15]) >> 40) & 0xFFF, "nbig", int(buf[i, 15]) >> 52, d[i].tolist())
    print("total percentiles", [int(np.percentile(tot, q)) for q in (10, 50, 90, 99, 100)])
    st = t[full][:, 0] - t0
    print("start-time percentiles", [int(np.percentile(st, q)) for q in (10, 50, 60, 90, 99, 100)], "last end", int((t[full][:, NS] - t0).max()))
    # per-octant mean total
    octs = ((buf[:, 14] >> np.uint64(32)) & np.uint64(0xFF)).astype(np.int64)
    for o in range(8):
        sel = full & (octs == o)
        if sel.any():
            print("octant", o, "workgroups", int(sel.sum()), "mean", int((t[sel][:, NS] - t[sel][:, 0]).mean()), "max", int((t[sel][:, NS] - t[sel][:, 0]).max()))

    nov = (buf[:, 15] & np.uint64(0xFFFFFFFF)).astype(np.int64)[full]
    print("overflow list: mean", nov.mean(), "p50", np.median(nov), "p90", np.percentile(nov, 90), "max", nov.max(), "sorted-mode wgs", int((((buf[:, 15] >> np.uint64(32)) & np.uint64(1)) > 0).sum()))
    print("corr(edge cycles, nover)", np.corrcoef(d[:, 9], nov)[0, 1])
    for o in range(8):
        sel = (octs == o)[full]
        if sel.any():
            print("octant", o, "phase means", [int(v) for v in d[sel].mean(axis=0)])
    idx_full = np.nonzero(full)[0]
    for w in (169, 209, 38, 166, 384, 320):
        j = np.nonzero(idx_full == w)[0]
        if len(j):
            print("wg", w, "phases", d[j[0]].tolist())
        f = t[16384 + w]
        print("   fine: slots", int(f[1] - f[0]), "append loop", int(f[2] - f[1]), "append barrier", int(t[w, 3] - f[2]), "| gather", int(f[5] - f[4]), "flat", int(f[6] - f[5]),
              "end barrier", int(t[w, 10] - f[6]), "flat tasks", int(buf[16384 + w, 11]) >> 32, "lw", (int(buf[16384 + w, 11]) >> 16) & 0xFFFF, "ncol", int(buf[16384 + w, 11]) & 0xFFFF)
    b12 = buf[:, 12].astype(np.uint64)
    Tv = (b12 >> np.uint64(48)).astype(np.int64); mr = ((b12 >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    im = ((b12 >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64); nn = (b12 & np.uint64(0xFFFF)).astype(np.int64)
    nl = (buf[:, 13] & np.uint64(0xFFFFFFFF)).astype(np.int64); ts = (buf[:, 13] >> np.uint64(32)).astype(np.int64)
    print("T>=imax:", int((Tv >= im).sum()), " marched tails:", int(((Tv < im) & (nl <= 512) & (ts <= 32768)).sum()),
          " full-loop tails:", int(((Tv < im) & ~((nl <= 512) & (ts <= 32768))).sum()))
    fl = ((Tv < im) & ~((nl <= 512) & (ts <= 32768)))[full]
    Tv, mr, im, nn, nl, ts = Tv[full], mr[full], im[full], nn[full], nl[full], ts[full]
    print("full-loop WGs: mean edge", d[fl][:, 9].mean() if fl.any() else 0, "others mean edge", d[~fl][:, 9].mean())
    print("full-loop: columns beyond T mean", (im - Tv)[fl].mean() if fl.any() else 0, " of which beyond maxreach", np.maximum(im - np.maximum(mr, Tv), 0)[fl].mean() if fl.any() else 0)
    print("n ends: mean", nn.mean(), "max", nn.max())
