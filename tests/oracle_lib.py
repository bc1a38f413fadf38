"""ctypes binding of the CPU oracle (oracle/libgv_oracle.so).

TEST INFRASTRUCTURE ONLY.  The product path (grid-vision_amd/) never imports
this module; see oracle/gv_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))

from gvamd.synth import BBOX_DTYPE, LSHAPE_DTYPE  # noqa: E402

_LIB = None


class Grid(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("res", C.c_double),
                ("len_x", C.c_double), ("len_y", C.c_double),
                ("pos_x", C.c_double), ("pos_y", C.c_double),
                ("log_odds", C.POINTER(C.c_float)), ("occupancy", C.POINTER(C.c_float))]


class Tf(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("qx", "qy", "qz", "qw", "tx", "ty", "tz")]


class Cam(C.Structure):
    _fields_ = [("network_h", C.c_int32), ("network_w", C.c_int32), ("orig_h", C.c_int32),
                ("orig_w", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float)]


def build():
    src_newer = False
    so = os.path.join(ORACLE_DIR, "libgv_oracle.so")
    if os.path.exists(so):
        t = os.path.getmtime(so)
        for f in os.listdir(ORACLE_DIR):
            if f.endswith((".c", ".h")) and os.path.getmtime(os.path.join(ORACLE_DIR, f)) > t:
                src_newer = True
    if not os.path.exists(so) or src_newer:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.gvo_estimated_depth.restype = C.c_float
        _LIB.gvo_compute_alpha.restype = C.c_float
        _LIB.gvo_compute_theta_ray.restype = C.c_float
        _LIB.gvo_project_points.restype = C.c_size_t
        _LIB.gvo_segment_ground_plane.restype = C.c_size_t
    return _LIB


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_tf(v) -> Tf:
    return Tf(*[float(t) for t in v])


def make_cam(fx=320.0, fy=320.0, cx=320.0, cy=240.0, w=640, h=480, nw=224, nh=224) -> Cam:
    return Cam(nh, nw, h, w, fx, fy, cx, cy)


class OGrid:
    """Owns one oracle grid (log_odds + occupancy)."""

    def __init__(self, grid_x: int, grid_y: int, res: float):
        self.g = Grid()
        rc = lib().gvo_grid_init(C.byref(self.g), C.c_uint8(grid_x), C.c_uint8(grid_y), C.c_double(res))
        if rc != 0:
            raise ValueError(f"gvo_grid_init failed: {rc}")
        self.nx, self.ny = self.g.nx, self.g.ny
        self.G = self.nx * self.ny

    def __del__(self):
        try:
            lib().gvo_grid_free(C.byref(self.g))
        except Exception:
            pass

    @property
    def log_odds(self):
        return np.ctypeslib.as_array(self.g.log_odds, shape=(self.G,))

    @property
    def occupancy(self):
        return np.ctypeslib.as_array(self.g.occupancy, shape=(self.G,))

    def get_index(self, x, y):
        ix, iy = C.c_int32(), C.c_int32()
        ok = lib().gvo_get_index(C.byref(self.g), C.c_double(x), C.c_double(y), C.byref(ix), C.byref(iy))
        return bool(ok), ix.value, iy.value

    def update_map(self):
        lib().gvo_update_map(C.byref(self.g))

    def update_map_poses(self, poses):
        poses = np.ascontiguousarray(poses, dtype=LSHAPE_DTYPE)
        lib().gvo_update_map_poses(C.byref(self.g), poses.ctypes.data_as(C.c_void_p), C.c_int32(len(poses)))

    def update_map_points(self, pts, bboxes):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        bboxes = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        lib().gvo_update_map_points(C.byref(self.g), _p(pts, C.c_double),
                                    bboxes.ctypes.data_as(C.c_void_p), C.c_int32(len(bboxes)))

    def update_cells_fast(self, corners):
        c = np.ascontiguousarray(corners, dtype=np.float64).reshape(8)
        return int(lib().gvo_update_grid_cells_fast(C.byref(self.g), _p(c, C.c_double)))

    def frame_update(self, poses, hits, miss):
        poses = np.ascontiguousarray(poses if poses is not None else np.zeros(0, LSHAPE_DTYPE), dtype=LSHAPE_DTYPE)
        hp = _p(hits, C.c_int32) if hits is not None else None
        mp = _p(miss, C.c_uint8) if miss is not None else None
        lib().gvo_frame_update(C.byref(self.g), poses.ctypes.data_as(C.c_void_p), C.c_int32(len(poses)), hp, mp)

    def to_occupancy_grid(self):
        data = np.zeros(self.G, dtype=np.int8)
        info = np.zeros(5, dtype=np.float64)
        lib().gvo_to_occupancy_grid(C.byref(self.g), _p(data, C.c_int8), _p(info, C.c_double))
        return data, info

    def bin_points(self, m_base, x, y, z):
        x, y, z = f32(x), f32(y), f32(z)
        hits = np.zeros(self.G, dtype=np.int32)
        cell = np.zeros(len(x), dtype=np.int32)
        m = f32(m_base).reshape(16)
        lib().gvo_bin_points(C.byref(self.g), _p(m, C.c_float), _p(x, C.c_float), _p(y, C.c_float),
                             _p(z, C.c_float), C.c_size_t(len(x)), _p(hits, C.c_int32), _p(cell, C.c_int32))
        return hits, cell

    def raymarch(self, m_base, x, y, z, dedupe=True):
        x, y, z = f32(x), f32(y), f32(z)
        miss = np.zeros(self.G, dtype=np.uint8)
        m = f32(m_base).reshape(16)
        visits = C.c_uint64(0)
        lib().gvo_raymarch(C.byref(self.g), _p(m, C.c_float), _p(x, C.c_float), _p(y, C.c_float),
                           _p(z, C.c_float), C.c_size_t(len(x)), _p(miss, C.c_uint8),
                           C.c_int(1 if dedupe else 0), C.byref(visits))
        return miss, visits.value

    def ray_ends(self, m_base, x, y, z):
        """(kind, ex, ey) of every point: kind 0 none, 1 hit end, 2 clipped end"""
        bx, by, bz = transform_cloud(m_base, x, y, z)
        ox, oy = float(f32(m_base).reshape(16)[3]), float(f32(m_base).reshape(16)[7])
        n = len(bx)
        kind = np.zeros(n, np.uint8)
        ex, ey = np.zeros(n, np.int32), np.zeros(n, np.int32)
        if not self.get_index(ox, oy)[0]:
            return kind, ex, ey
        cx, cy = C.c_int32(), C.c_int32()
        L = lib()
        for i in range(n):
            kind[i] = L.gvo_ray_end(C.byref(self.g), C.c_double(ox), C.c_double(oy), C.c_float(bx[i]), C.c_float(by[i]),
                                    C.c_float(bz[i]), C.byref(cx), C.byref(cy))
            ex[i], ey[i] = cx.value, cy.value
        return kind, ex, ey

    def march_ends(self, m_base, ex, ey, kind):
        m = f32(m_base).reshape(16)
        miss = np.zeros(self.G, dtype=np.uint8)
        ex, ey = np.ascontiguousarray(ex, np.int32), np.ascontiguousarray(ey, np.int32)
        kind = np.ascontiguousarray(kind, np.uint8)
        lib().gvo_march_ends(C.byref(self.g), C.c_double(float(m[3])), C.c_double(float(m[7])), _p(ex, C.c_int32),
                             _p(ey, C.c_int32), _p(kind, C.c_uint8), C.c_size_t(len(ex)), _p(miss, C.c_uint8), None)
        return miss

    def ray_end(self, ox, oy, px, py, pz=0.0):
        ex, ey = C.c_int32(), C.c_int32()
        k = lib().gvo_ray_end(C.byref(self.g), C.c_double(ox), C.c_double(oy), C.c_float(px),
                              C.c_float(py), C.c_float(pz), C.byref(ex), C.byref(ey))
        return int(k), ex.value, ey.value


def tf_to_matrix4f(tf):
    m = np.zeros(16, dtype=np.float32)
    t = make_tf(tf)
    lib().gvo_tf_to_matrix4f(C.byref(t), _p(m, C.c_float))
    return m


def transform_cloud(m, x, y, z):
    x, y, z = f32(x), f32(y), f32(z)
    ox, oy, oz = np.empty_like(x), np.empty_like(y), np.empty_like(z)
    m = f32(m).reshape(16)
    lib().gvo_transform_cloud(_p(m, C.c_float), _p(x, C.c_float), _p(y, C.c_float), _p(z, C.c_float),
                              _p(ox, C.c_float), _p(oy, C.c_float), _p(oz, C.c_float), C.c_size_t(len(x)))
    return ox, oy, oz


def tf_point(tf, p):
    t = make_tf(tf)
    i = np.ascontiguousarray(p, dtype=np.float64)
    o = np.zeros(3, dtype=np.float64)
    lib().gvo_tf_point(C.byref(t), _p(i, C.c_double), _p(o, C.c_double))
    return o


def tf_pose(tf, pose7):
    t = make_tf(tf)
    i = np.ascontiguousarray(pose7, dtype=np.float64)
    o = np.zeros(7, dtype=np.float64)
    lib().gvo_tf_pose(C.byref(t), _p(i, C.c_double), _p(o, C.c_double))
    return o


def set_rpy(r, p, y):
    q = np.zeros(4, dtype=np.float64)
    lib().gvo_set_rpy(C.c_double(r), C.c_double(p), C.c_double(y), _p(q, C.c_double))
    return q


def set_intrinsic(fx, fy, cx, cy):
    k = np.zeros(9, dtype=np.float64)
    lib().gvo_set_intrinsic(C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), _p(k, C.c_double))
    return k


def k_inverse(k):
    k = np.ascontiguousarray(k, dtype=np.float64)
    o = np.zeros(9, dtype=np.float64)
    lib().gvo_k_inverse(_p(k, C.c_double), _p(o, C.c_double))
    return o


def extract_bboxes(boxes, scores, conf_thr, iou_thr, orig_w, orig_h, resize):
    boxes, scores = f32(boxes), f32(scores)
    n, c = scores.shape
    out = np.zeros(max(n, 1), dtype=BBOX_DTYPE)
    k = lib().gvo_extract_bboxes(_p(boxes, C.c_float), _p(scores, C.c_float), C.c_int32(n), C.c_int32(c),
                                 C.c_double(conf_thr), C.c_double(iou_thr), C.c_int32(orig_w),
                                 C.c_int32(orig_h), C.c_int32(resize), out.ctypes.data_as(C.c_void_p))
    return out[:k].copy()


def nms(bboxes, iou_thr):
    b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE).copy()
    out = np.zeros(max(len(b), 1), dtype=BBOX_DTYPE)
    k = lib().gvo_nms(b.ctypes.data_as(C.c_void_p), C.c_int32(len(b)), C.c_float(iou_thr),
                      out.ctypes.data_as(C.c_void_p))
    return out[:k].copy()


def denormalize(bboxes, orig_w, orig_h, resize):
    b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE).copy()
    lib().gvo_denormalize(b.ctypes.data_as(C.c_void_p), C.c_int32(len(b)), C.c_int32(orig_w),
                          C.c_int32(orig_h), C.c_int32(resize))
    return b


def filter_bboxes(bboxes):
    b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
    st = np.zeros(max(len(b), 1), dtype=BBOX_DTYPE)
    dy = np.zeros(max(len(b), 1), dtype=BBOX_DTYPE)
    nd = C.c_int32(0)
    ns = lib().gvo_filter_bboxes(b.ctypes.data_as(C.c_void_p), C.c_int32(len(b)),
                                 st.ctypes.data_as(C.c_void_p), dy.ctypes.data_as(C.c_void_p), C.byref(nd))
    return st[:ns].copy(), dy[:nd.value].copy()


def project_points(K, x, y, z):
    x, y, z = f32(x), f32(y), f32(z)
    K = np.ascontiguousarray(K, dtype=np.float64)
    u, v, d = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    m = lib().gvo_project_points(_p(K, C.c_double), _p(x, C.c_float), _p(y, C.c_float), _p(z, C.c_float),
                                 C.c_size_t(len(x)), _p(u, C.c_float), _p(v, C.c_float), _p(d, C.c_float))
    return u[:m].copy(), v[:m].copy(), d[:m].copy()


def depth_for_bboxes(u, v, d, bboxes, k):
    u, v, d = f32(u), f32(v), f32(d)
    b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
    depths = np.zeros(len(b), dtype=np.float32)
    d2 = np.zeros((len(b), k), dtype=np.float32)
    lib().gvo_depth_for_bboxes(_p(u, C.c_float), _p(v, C.c_float), _p(d, C.c_float), C.c_size_t(len(u)),
                               b.ctypes.data_as(C.c_void_p), C.c_int32(len(b)), C.c_int32(k),
                               _p(depths, C.c_float), _p(d2, C.c_float))
    return depths, d2


def pixel_to_3d(px, py, depth, kinv):
    kinv = np.ascontiguousarray(kinv, dtype=np.float64)
    o = np.zeros(3, dtype=np.float64)
    lib().gvo_pixel_to_3d(C.c_float(px), C.c_float(py), C.c_float(depth), _p(kinv, C.c_double), _p(o, C.c_double))
    return o


def extract_cloud_per_bbox(K, x, y, z, bboxes, w, h):
    x, y, z = f32(x), f32(y), f32(z)
    K = np.ascontiguousarray(K, dtype=np.float64)
    b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
    out = np.zeros(len(x), dtype=np.int32)
    lib().gvo_extract_cloud_per_bbox(_p(K, C.c_double), _p(x, C.c_float), _p(y, C.c_float), _p(z, C.c_float),
                                     C.c_size_t(len(x)), b.ctypes.data_as(C.c_void_p), C.c_int32(len(b)),
                                     C.c_int32(w), C.c_int32(h), _p(out, C.c_int32))
    return out


def radius_outlier(x, y, z, radius=0.4, min_pts=10):
    x, y, z = f32(x), f32(y), f32(z)
    keep = np.zeros(len(x), dtype=np.uint8)
    lib().gvo_radius_outlier(_p(x, C.c_float), _p(y, C.c_float), _p(z, C.c_float), C.c_size_t(len(x)),
                             C.c_double(radius), C.c_int32(min_pts), _p(keep, C.c_uint8))
    return keep


def radius_outlier_grid(x, y, z, radius=0.4, min_pts=10):
    x, y, z = f32(x), f32(y), f32(z)
    keep = np.zeros(len(x), dtype=np.uint8)
    lib().gvo_radius_outlier_grid(_p(x, C.c_float), _p(y, C.c_float), _p(z, C.c_float), C.c_size_t(len(x)),
                                  C.c_double(radius), C.c_int32(min_pts), _p(keep, C.c_uint8))
    return keep


def pca_bbox(x, y, z):
    x, y, z = f32(x), f32(y), f32(z)
    out = np.zeros(1, dtype=LSHAPE_DTYPE)
    ok = lib().gvo_pca_bbox(_p(x, C.c_float), _p(y, C.c_float), _p(z, C.c_float), C.c_size_t(len(x)),
                            out.ctypes.data_as(C.c_void_p))
    return bool(ok), out[0]


def pca_angle_deg(major_y, major_x):
    """cloud_detections.cpp:227 alone"""
    f = lib().gvo_pca_angle_deg
    f.restype = C.c_float
    return float(f(C.c_float(major_y), C.c_float(major_x)))


def generate_bins(n):
    o = np.zeros(n, dtype=np.float32)
    lib().gvo_generate_bins(C.c_int32(n), _p(o, C.c_float))
    return o


def compute_alpha(orient4, argmax, bins):
    o, b = f32(orient4), f32(bins)
    return float(lib().gvo_compute_alpha(_p(o, C.c_float), C.c_int32(argmax), _p(b, C.c_float)))


def compute_theta_ray(cam: Cam, bbox):
    b = np.ascontiguousarray(bbox, dtype=BBOX_DTYPE).reshape(1)
    return float(lib().gvo_compute_theta_ray(C.byref(cam), b.ctypes.data_as(C.c_void_p)))


def calc_location(cam: Cam, dims3, bbox, alpha, theta_ray):
    b = np.ascontiguousarray(bbox, dtype=BBOX_DTYPE).reshape(1)
    d = np.ascontiguousarray(dims3, dtype=np.float64)
    pose = np.zeros(7, dtype=np.float64)
    err = C.c_float(0)
    lib().gvo_calc_location(C.byref(cam), _p(d, C.c_double), b.ctypes.data_as(C.c_void_p),
                            C.c_float(alpha), C.c_float(theta_ray), _p(pose, C.c_double), C.byref(err))
    return pose, err.value


def calc_location_all(cam: Cam, dims3, bbox, alpha, theta_ray):
    """(64, 3) solutions and (64,) residuals of every constraint set of calcLocation"""
    b = np.ascontiguousarray(bbox, dtype=BBOX_DTYPE).reshape(1)
    d = np.ascontiguousarray(dims3, dtype=np.float64)
    loc = np.zeros((64, 3), dtype=np.float32)
    err = np.zeros(64, dtype=np.float32)
    lib().gvo_calc_location_all(C.byref(cam), _p(d, C.c_double), b.ctypes.data_as(C.c_void_p), C.c_float(alpha),
                                C.c_float(theta_ray), _p(loc, C.c_float), _p(err, C.c_float))
    return loc, err


def post_process(cam: Cam, orient, conf, dims, bboxes):
    orient, conf, dims = f32(orient), f32(conf), f32(dims)
    b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
    out = np.zeros(max(len(b), 1), dtype=LSHAPE_DTYPE)
    m = lib().gvo_post_process(C.byref(cam), _p(orient, C.c_float), _p(conf, C.c_float), _p(dims, C.c_float),
                               b.ctypes.data_as(C.c_void_p), C.c_int32(len(b)), out.ctypes.data_as(C.c_void_p))
    return out[:m].copy()


def segment_ground_plane(x, y, z, thr=0.04, iters=50, seed=12345):
    x, y, z = f32(x), f32(y), f32(z)
    inl = np.zeros(len(x), dtype=np.uint8)
    coeff = np.zeros(4, dtype=np.float32)
    m = lib().gvo_segment_ground_plane(_p(x, C.c_float), _p(y, C.c_float), _p(z, C.c_float), C.c_size_t(len(x)),
                                       C.c_double(thr), C.c_int32(iters), C.c_uint64(seed), _p(inl, C.c_uint8),
                                       _p(coeff, C.c_float))
    return int(m), inl, coeff
