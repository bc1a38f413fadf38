"""Deterministic synthetic inputs for the grid-vision hot path.

All randomness comes from a counter-based splitmix64 coded here with integer
numpy ops only, so every platform produces the same bits.  Nothing in this file
touches the oracle or the HIP library: it only makes inputs.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
BASE_SEED = 0x9E3779B97F4A7C15


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


class Stream:
    """splitmix64 in counter mode: draw k = mix64(seed + (k+1)*GOLDEN)."""

    def __init__(self, seed: int, stream: int = 0):
        with np.errstate(over="ignore"):
            s = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
        self.seed = _mix64(np.array([s], dtype=np.uint64))[0]
        self.pos = 0

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(self.pos + 1, self.pos + n + 1, dtype=np.uint64)
            out = _mix64(self.seed + idx * GOLDEN)
        self.pos += n
        return out

    def uniform01(self, n: int) -> np.ndarray:
        """fp32 in [0,1) with 24 random bits (exact in fp32)."""
        return ((self.u64(n) >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)

    def uniform(self, n: int, lo: float, hi: float) -> np.ndarray:
        return (np.float32(lo) + self.uniform01(n) * np.float32(hi - lo)).astype(np.float32)

    def integers(self, n: int, lo: int, hi: int) -> np.ndarray:
        """integers in [lo, hi)"""
        return (lo + (self.u64(n) % np.uint64(hi - lo)).astype(np.int64)).astype(np.int64)


@dataclasses.dataclass
class GridCfg:
    grid_x: int
    grid_y: int
    resolution: float

    @property
    def nx(self) -> int:
        return int(round(self.grid_x / self.resolution))

    @property
    def ny(self) -> int:
        return int(round(self.grid_y / self.resolution))

    @property
    def pos_x(self) -> float:
        return float(self.grid_x // 3)


# BASELINE.json configs
CONFIGS = {
    1: dict(n=10_000, grid=GridCfg(100, 100, 0.5), dets=0),
    2: dict(n=100_000, grid=GridCfg(200, 200, 0.2), dets=0),
    3: dict(n=1_000_000, grid=GridCfg(200, 200, 0.1), dets=50),
    4: dict(n=1_000_000, grid=GridCfg(200, 200, 0.1), dets=50),
    5: dict(n=10_000_000, grid=GridCfg(200, 200, 0.05), dets=0),
}

# camera intrinsics: config/grid_vision_cfg.yaml:16-19, 10-11
FX, FY, CX, CY = 320.0, 320.0, 320.0, 240.0
IMG_W, IMG_H = 640, 480


def quat_from_matrix(m: np.ndarray) -> tuple:
    """Shepperd's method, fp64; only used to MAKE test transforms."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        return ((m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s)
    i = int(np.argmax([m[0, 0], m[1, 1], m[2, 2]]))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0) * 2
    q = [0.0, 0.0, 0.0, 0.0]
    q[i] = 0.25 * s
    q[3] = (m[k, j] - m[j, k]) / s
    q[j] = (m[j, i] + m[i, j]) / s
    q[k] = (m[k, i] + m[i, k]) / s
    return tuple(q)


def _rot(axis: str, deg: float) -> np.ndarray:
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    if axis == "z":
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1.0, 0], [-s, 0, c]])
    return np.array([[1.0, 0, 0], [0, c, -s], [0, s, c]])


def transforms(perturbed: bool = False) -> dict:
    """tf (qx,qy,qz,qw,tx,ty,tz) for base<-lidar, camera<-lidar, base<-camera."""
    r_bl = np.eye(3)
    t_bl = np.array([0.0, 0.0, 1.8])
    # x_cam = -y_lidar, y_cam = -z_lidar, z_cam = x_lidar
    r_cl = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
    t_cl = np.array([0.0, 0.4, -0.3])
    if perturbed:  # 3 degree yaw + pitch, exercises fp32 rounding
        r_cl = r_cl @ _rot("z", 3.0) @ _rot("y", 3.0)
        r_bl = _rot("z", 3.0) @ _rot("y", 3.0)
    # base<-camera = base<-lidar o (camera<-lidar)^-1
    r_bc = r_bl @ r_cl.T
    t_bc = t_bl - r_bc @ t_cl
    def pack(r, t):
        return np.array([*quat_from_matrix(r), *t], dtype=np.float64)
    return dict(base_lidar=pack(r_bl, t_bl), cam_lidar=pack(r_cl, t_cl), base_cam=pack(r_bc, t_bc))


def _map_extent(g: GridCfg):
    lx, ly = g.nx * g.resolution, g.ny * g.resolution
    hix, hiy = g.pos_x + 0.5 * lx, 0.5 * ly
    return hix - lx, hix, hiy - ly, hiy, lx, ly


def cloud_uniform(config: int, n: int | None = None, seed_extra: int = 0):
    """SoA lidar-frame cloud, uniform over the map extent widened by 5% per side."""
    cfg = CONFIGS[config]
    n = cfg["n"] if n is None else n
    lox, hix, loy, hiy, lx, ly = _map_extent(cfg["grid"])
    st = Stream(BASE_SEED ^ config, stream=1 + 16 * seed_extra)
    x = st.uniform(n, lox - 0.05 * lx, hix + 0.05 * lx)
    y = st.uniform(n, loy - 0.05 * ly, hiy + 0.05 * ly)
    z = st.uniform(n, -2.0, 4.0)
    inten = st.uniform01(n)
    return x, y, z, inten


def cloud_lidar_like(config: int, n: int | None = None, seed_extra: int = 0):
    """range ~ Exp(mean 15 m) capped at 120 m, uniform azimuth, 64 elevation rings."""
    cfg = CONFIGS[config]
    n = cfg["n"] if n is None else n
    st = Stream(BASE_SEED ^ config, stream=2 + 16 * seed_extra)
    u = st.uniform01(n).astype(np.float64)
    r = np.minimum(-15.0 * np.log1p(-u), 120.0)
    az = st.uniform01(n).astype(np.float64) * (2.0 * math.pi)
    ring = st.integers(n, 0, 64).astype(np.float64)
    el = np.radians(-25.0 + ring * (28.0 / 63.0))
    x = (r * np.cos(el) * np.cos(az)).astype(np.float32)
    y = (r * np.cos(el) * np.sin(az)).astype(np.float32)
    z = (r * np.sin(el)).astype(np.float32)
    inten = st.uniform01(n)
    return x, y, z, inten


def detections(config: int, b: int | None = None, seed_extra: int = 0):
    """B integer-valued pixel bboxes inside 640x480 (widths 20..200 px), labels
    uniform over 0..9, confidences sorted descending; returns a structured array
    with the BoundingBox layout (4 f64, f32, i32)."""
    cfg = CONFIGS[config]
    b = cfg["dets"] if b is None else b
    st = Stream(BASE_SEED ^ config, stream=3 + 16 * seed_extra)
    w = st.integers(b, 20, 201)
    h = st.integers(b, 20, 201)
    x0 = (st.u64(b) % (IMG_W - w).astype(np.uint64)).astype(np.int64)
    y0 = (st.u64(b) % (IMG_H - h).astype(np.uint64)).astype(np.int64)
    label = st.integers(b, 0, 10)
    conf = np.sort(st.uniform(b, 0.6, 1.0))[::-1]
    out = np.zeros(b, dtype=BBOX_DTYPE)
    out["x_min"], out["y_min"] = x0, y0
    out["x_max"], out["y_max"] = x0 + w, y0 + h
    out["confidence"] = conf
    out["label"] = label
    return out


def lshape_poses(config: int, b: int | None = None, seed_extra: int = 0):
    """B LShapePose in the BASE frame: position inside the map, length 0.5..5,
    width 0.5..2.5; a few are pushed across the border on purpose
    so the 'any corner outside -> skip' rule (occupancy_grid.cpp:152-156) runs."""
    cfg = CONFIGS[config]
    b = cfg["dets"] if b is None else b
    lox, hix, loy, hiy, lx, ly = _map_extent(cfg["grid"])
    st = Stream(BASE_SEED ^ config, stream=4 + 16 * seed_extra)
    out = np.zeros(b, dtype=LSHAPE_DTYPE)
    out["px"] = st.uniform(b, lox, hix).astype(np.float64)
    out["py"] = st.uniform(b, loy, hiy).astype(np.float64)
    out["pz"] = 0.0
    out["qw"] = 1.0
    out["length"] = st.uniform(b, 0.5, 5.0).astype(np.float64)
    out["width"] = st.uniform(b, 0.5, 2.5).astype(np.float64)
    out["height"] = st.uniform(b, 1.0, 2.0).astype(np.float64)
    return out


def _quat_to_matrix(q):
    x, y, z, w = (float(v) for v in q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def scene_with_objects(tfs: dict, n_total: int = 1_000_000, seed: int = 17, n_obj: int = 40, per: int = 6000):
    """A config-3 sized scene WITH STRUCTURE, the workload of the tick legs (a uniform cloud loses every point to the
    radius filter: no poses): a ground plane, n_obj dense objects in front of the camera -- each inside a pixel box of
    its own, labels cycling vehicle / person / bike / motorbike / traffic light, so four of five boxes are dynamic --,
    a lidar-like near field and uniform clutter, shuffled as a sensor delivers it.  Returns x, y, z (lidar frame,
    float32) and the boxes (BBOX_DTYPE, confidence descending).  numpy's own generator: inputs only, nothing here is
    compared bit for bit across platforms."""
    rng = np.random.default_rng(seed)
    xs, ys, zs = [], [], []
    ocx = rng.uniform(6.0, 60.0, n_obj)
    ocy = rng.uniform(-0.6, 0.6, n_obj) * ocx
    ocz = rng.uniform(-1.0, 0.4, n_obj)
    for k in range(n_obj):
        ang = rng.uniform(0, np.pi)
        a, c = rng.uniform(-2.2, 2.2, per), rng.uniform(-0.8, 0.8, per)
        xs.append(ocx[k] + a * np.cos(ang) - c * np.sin(ang))
        ys.append(ocy[k] + a * np.sin(ang) + c * np.cos(ang))
        zs.append(ocz[k] + rng.uniform(-0.5, 0.5, per))
    ng = int(0.45 * n_total)
    gx_ = rng.uniform(1.0, 90.0, ng)
    gy_ = rng.uniform(-60.0, 60.0, ng)
    xs.append(gx_); ys.append(gy_); zs.append(-1.75 + 0.004 * gx_ + rng.normal(0, 0.012, ng))
    nl = int(0.15 * n_total)
    lx, ly, lz, _ = cloud_lidar_like(3, nl, seed_extra=seed)
    xs.append(lx); ys.append(ly); zs.append(lz)
    nu = n_total - n_obj * per - ng - nl
    assert nu >= 0, "n_total too small for the objects"
    xs.append(rng.uniform(-44, 176, nu)); ys.append(rng.uniform(-110, 110, nu)); zs.append(rng.uniform(-2, 4, nu))
    x = np.concatenate(xs).astype(np.float32)
    y = np.concatenate(ys).astype(np.float32)
    z = np.concatenate(zs).astype(np.float32)
    # pixel boxes around the objects (fp64 projection: the boxes are inputs, not results)
    tf = tfs["cam_lidar"]
    r, t = _quat_to_matrix(tf[:4]), np.asarray(tf[4:7], dtype=np.float64)
    boxes = []
    for k in range(n_obj):
        sl = slice(k * per, (k + 1) * per)
        pc = np.stack([x[sl], y[sl], z[sl]], axis=1).astype(np.float64) @ r.T + t
        ok = pc[:, 2] > 0.1
        if ok.sum() < 100:
            continue
        u = FX * pc[ok, 0] / pc[ok, 2] + CX
        v = FY * pc[ok, 1] / pc[ok, 2] + CY
        x0, x1 = max(0.0, np.percentile(u, 2)), min(IMG_W - 1.0, np.percentile(u, 98))
        y0, y1 = max(0.0, np.percentile(v, 2)), min(IMG_H - 1.0, np.percentile(v, 98))
        if x1 - x0 > 3 and y1 - y0 > 3:
            boxes.append((float(np.float32(x0 + 0.25)), float(np.float32(y0 + 0.5)), float(np.float32(x1 + 0.75)), float(np.float32(y1))))
    b = np.zeros(len(boxes), dtype=BBOX_DTYPE)
    for i, (x0, y0, x1, y1) in enumerate(boxes):
        b[i] = (x0, y0, x1, y1, 0.99 - 0.01 * i, [9, 2, 0, 1, 5][i % 5])
    perm = rng.permutation(len(x))
    return x[perm], y[perm], z[perm], b


def network_outputs(b: int, seed: int = 7):
    """Synthetic vision-orientation network outputs: orient[b,2,2] (cos,sin per
    bin), conf[b,2], dims[b,3] residuals (vision_orientation.cpp:461-463)."""
    st = Stream(BASE_SEED ^ seed, stream=5)
    ang = st.uniform(2 * b, -math.pi, math.pi).astype(np.float64)
    orient = np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32).reshape(b, 4)
    conf = st.uniform(2 * b, 0.0, 1.0).reshape(b, 2)
    dims = st.uniform(3 * b, -0.3, 0.3).reshape(b, 3)
    return orient, conf, dims


# POD layouts shared by the ctypes bindings (C-ABI and oracle use the same ones)
BBOX_DTYPE = np.dtype([("x_min", "<f8"), ("y_min", "<f8"), ("x_max", "<f8"), ("y_max", "<f8"),
                       ("confidence", "<f4"), ("label", "<i4")], align=True)
LSHAPE_DTYPE = np.dtype([("px", "<f8"), ("py", "<f8"), ("pz", "<f8"),
                         ("qx", "<f8"), ("qy", "<f8"), ("qz", "<f8"), ("qw", "<f8"),
                         ("length", "<f8"), ("width", "<f8"), ("height", "<f8")], align=True)
assert BBOX_DTYPE.itemsize == 40 and LSHAPE_DTYPE.itemsize == 80
