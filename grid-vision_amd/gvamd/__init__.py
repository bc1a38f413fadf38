"""Python (ctypes) binding of the C ABI in include/gridvision_hip.h.

This is host-side plumbing for tests and bench.py; all compute happens in
libgridvision_hip.so (hand-written gfx950 kernels).  There is NO CPU fallback:
if the library is missing or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from .synth import BBOX_DTYPE, LSHAPE_DTYPE

GV_OK = 0
STATUS = {0: "GV_OK", 1: "GV_ERR_BAD_ARG", 2: "GV_ERR_HIP", 3: "GV_ERR_RCCL", 4: "GV_ERR_NO_DEVICE",
          5: "GV_ERR_STATE", 6: "GV_ERR_TF"}

FRAME_BIN = 1 << 0
FRAME_RAYMARCH = 1 << 1
FRAME_BBOX_TEST = 1 << 2
FRAME_KEEP_CELL_IDX = 1 << 3
FRAME_KEEP_COUNTS = 1 << 4
FRAME_VISION_ORIENT = 1 << 5

TICK_VISION_ORIENT = 1 << 0
TICK_LIDAR_BIN = 1 << 1
TICK_LIDAR_RAYMARCH = 1 << 2

STAGES = ("detections", "points", "ray_ends", "ray_march", "finalize")

# every symbol include/gridvision_hip.h declares
ABI_SYMBOLS = [
    "gv_create", "gv_destroy", "gv_last_error", "gv_abi_version", "gv_grid_geometry", "gv_reset",
    "gv_set_transforms", "gv_cloud_upload_xyz", "gv_cloud_upload_pointcloud2",
    "gv_transform_lidar_to_camera", "gv_extract_cloud_per_bbox", "gv_compute_depth_for_bboxes",
    "gv_convert_pixels_to_3d", "gv_compute_bbox_pose", "gv_segment_ground_plane",
    "gv_compute_bbox_pose_ground_removed", "gv_vision_post_process",
    "gv_transform_lshape_objects", "gv_extract_bboxes", "gv_filter_bboxes", "gv_get_intrinsics",
    "gv_update_map", "gv_update_map_poses", "gv_update_map_points", "gv_to_occupancy_grid",
    "gv_get_log_odds", "gv_get_occupancy", "gv_set_log_odds", "gv_frame_set_detections",
    "gv_frame_enqueue", "gv_synchronize", "gv_process_frame", "gv_get_hits", "gv_get_miss",
    "gv_get_cell_idx", "gv_get_bbox_id", "gv_get_ray_stats", "gv_stream", "gv_time_frames",
    "gv_time_frame_stages", "gv_comm_unique_id", "gv_comm_init", "gv_comm_destroy",
    "gv_process_frame_sharded", "gv_comm_band",
    "gv_cloud_upload_xyz_async", "gv_cloud_upload_pointcloud2_async", "gv_cloud_upload_wait", "gv_host_alloc",
    "gv_host_free", "gv_frame_set_detections_async", "gv_frame_fence",
    "gv_to_occupancy_grid_async", "gv_frame_enqueue_sharded", "gv_time_frame_sharded_stages", "gv_shard_band_rows",
    "gv_shard_slice_words", "gv_device_layers", "gv_tick_enqueue", "gv_tick_wait", "gv_tick",
    "gv_comm_info", "gv_publish_grid_async",
]


class CamParams(C.Structure):
    _fields_ = [("network_h", C.c_int32), ("network_w", C.c_int32), ("orig_h", C.c_int32),
                ("orig_w", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float)]


class Transform(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("qx", "qy", "qz", "qw", "tx", "ty", "tz")]


class GridInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("resolution", C.c_double),
                ("origin_x", C.c_double), ("origin_y", C.c_double)]


class FrameDesc(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("bboxes", C.c_void_p), ("n_bboxes", C.c_int32),
                ("poses", C.c_void_p), ("n_poses", C.c_int32), ("orient", C.c_void_p),
                ("conf", C.c_void_p), ("dims", C.c_void_p)]


class TickDesc(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("bboxes", C.c_void_p), ("n_bboxes", C.c_int32), ("orient", C.c_void_p),
                ("conf", C.c_void_p), ("dims", C.c_void_p), ("n_net", C.c_int32), ("k_near", C.c_int32),
                ("grid_out", C.c_void_p)]


class TickResult(C.Structure):
    _fields_ = [("n_static", C.c_int32), ("n_dynamic", C.c_int32), ("static_bboxes", C.c_void_p), ("depths", C.c_void_p),
                ("base_points_xyz", C.c_void_p), ("poses", C.c_void_p), ("n_poses", C.c_int32), ("pca_empty", C.c_int32)]


class PinnedI8:
    """int8 array in page-locked host memory: the packed grid's landing place (gv_tick grid_out, gv_to_occupancy_grid_async)"""

    def __init__(self, n):
        self._lib = load()
        self._p = C.c_void_p()
        rc = self._lib.gv_host_alloc(C.byref(self._p), C.c_size_t(max(int(n), 1)))
        if rc:
            raise GVError(rc, "gv_host_alloc")
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_int8)), shape=(int(n),))

    def close(self):
        if self._p:
            self.array = None
            self._lib.gv_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GVError(RuntimeError):
    def __init__(self, code, where, detail=""):
        super().__init__(f"{where}: {STATUS.get(code, code)} {detail}".strip())
        self.code = code


_LIB = None


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True):
    """Load libgridvision_hip.so (builds it with hipcc when stale/missing)."""
    global _LIB
    if _LIB is None:
        path = _build.build() if build_if_missing else _build.LIB
        alt = os.environ.get("GV_LIB_AB")   # A/B measurements only: another build of the same library
        if alt:
            path = alt
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with hipcc (python -m gvamd.build)")
        _LIB = C.CDLL(path)
        _LIB.gv_last_error.restype = C.c_char_p
        _LIB.gv_stream.restype = C.c_void_p
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_tf(v):
    return Transform(*[float(t) for t in v]) if v is not None else None


def extract_bboxes(boxes, scores, conf_thr, iou_thr, orig_w, orig_h, resize):
    """object_detection::extract_bboxes on precomputed detector outputs (host side)."""
    boxes, scores = _f32(boxes), _f32(scores)
    n, c = scores.shape
    out = np.zeros(max(n, 1), dtype=BBOX_DTYPE)
    m = C.c_int32(0)
    rc = load().gv_extract_bboxes(_ptr(boxes), _ptr(scores), C.c_int32(n), C.c_int32(c), C.c_double(conf_thr),
                                  C.c_double(iou_thr), C.c_int32(orig_w), C.c_int32(orig_h), C.c_int32(resize),
                                  _ptr(out), C.byref(m))
    if rc:
        raise GVError(rc, "gv_extract_bboxes")
    return out[:m.value].copy()


def filter_bboxes(bboxes):
    b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
    st = np.zeros(max(len(b), 1), dtype=BBOX_DTYPE)
    dy = np.zeros(max(len(b), 1), dtype=BBOX_DTYPE)
    ns, nd = C.c_int32(0), C.c_int32(0)
    rc = load().gv_filter_bboxes(_ptr(b), C.c_int32(len(b)), _ptr(st), C.byref(ns), _ptr(dy), C.byref(nd))
    if rc:
        raise GVError(rc, "gv_filter_bboxes")
    return st[:ns.value].copy(), dy[:nd.value].copy()


def shard_band_rows(rank, world, ny):
    """rows [y0, y1) rank `rank` of `world` finalises (the product's own band function; needs no GPU)"""
    y0, y1 = C.c_int32(), C.c_int32()
    rc = load().gv_shard_band_rows(C.c_int32(rank), C.c_int32(world), C.c_int32(ny), C.byref(y0), C.byref(y1))
    if rc:
        raise GVError(rc, "gv_shard_band_rows")
    return y0.value, y1.value


def shard_slice_words(words, world):
    lib = load()
    lib.gv_shard_slice_words.restype = C.c_int64
    return int(lib.gv_shard_slice_words(C.c_int64(words), C.c_int32(world)))


class PinnedF32:
    """float32 array in page-locked host memory (gv_host_alloc) for the asynchronous uploads."""

    def __init__(self, n):
        self._lib = load()
        self._p = C.c_void_p()
        rc = self._lib.gv_host_alloc(C.byref(self._p), C.c_size_t(max(int(n), 1) * 4))
        if rc:
            raise GVError(rc, "gv_host_alloc")
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_float)), shape=(int(n),))

    def close(self):
        if self._p:
            self.array = None
            self._lib.gv_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GridVisionHIP:
    """One handle = one MI355X + one stream + one resident occupancy grid."""

    def __init__(self, grid_x, grid_y, resolution, fx=320.0, fy=320.0, cx=320.0, cy=240.0,
                 image_w=640, image_h=480, network_w=224, network_h=224, device=-1):
        self._lib = load()
        self._h = C.c_void_p()
        cam = CamParams(network_h, network_w, image_h, image_w, fx, fy, cx, cy)
        rc = self._lib.gv_create(C.byref(self._h), C.c_uint8(grid_x), C.c_uint8(grid_y), C.c_double(resolution),
                                 C.byref(cam), C.c_int(device))
        if rc:
            self._h = C.c_void_p()
            raise GVError(rc, "gv_create", "(no MI355X visible: this library has no CPU path)" if rc == 4 else "")
        nx, ny = C.c_int32(), C.c_int32()
        px, py = C.c_double(), C.c_double()
        self._ck(self._lib.gv_grid_geometry(self._h, C.byref(nx), C.byref(ny), C.byref(px), C.byref(py)), "geometry")
        self.nx, self.ny, self.pos_x, self.pos_y = nx.value, ny.value, px.value, py.value
        self.G = self.nx * self.ny
        self.n = 0

    # ---- plumbing
    def _ck(self, rc, where):
        if rc:
            raise GVError(rc, where, (self._lib.gv_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.gv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- setup
    def reset(self):
        self._ck(self._lib.gv_reset(self._h), "gv_reset")

    def set_transforms(self, cam_lidar=None, base_cam=None, base_lidar=None):
        a, b, c = make_tf(cam_lidar), make_tf(base_cam), make_tf(base_lidar)
        self._ck(self._lib.gv_set_transforms(self._h, C.byref(a) if a else None, C.byref(b) if b else None,
                                             C.byref(c) if c else None), "gv_set_transforms")

    def upload_xyz(self, x, y, z):
        x, y, z = _f32(x), _f32(y), _f32(z)
        self._ck(self._lib.gv_cloud_upload_xyz(self._h, _ptr(x), _ptr(y), _ptr(z), C.c_size_t(len(x))), "upload_xyz")
        self.n = len(x)

    def upload_xyz_async(self, x, y, z):
        """x, y, z: float32 arrays that stay alive and unchanged until upload_wait() (pinned: see PinnedF32)."""
        for a in (x, y, z):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        self._ck(self._lib.gv_cloud_upload_xyz_async(self._h, _ptr(x), _ptr(y), _ptr(z), C.c_size_t(len(x))),
                 "upload_xyz_async")
        self.n = len(x)

    def upload_wait(self):
        self._ck(self._lib.gv_cloud_upload_wait(self._h), "upload_wait")

    def upload_pointcloud2(self, data: np.ndarray, n, point_step, off_x, off_y, off_z):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        self._ck(self._lib.gv_cloud_upload_pointcloud2(self._h, _ptr(data), C.c_size_t(n), C.c_uint32(point_step),
                                                       C.c_uint32(off_x), C.c_uint32(off_y), C.c_uint32(off_z)),
                 "upload_pointcloud2")
        self.n = n

    def upload_pointcloud2_async(self, data: np.ndarray, n, point_step, off_x, off_y, off_z):
        """data: uint8 bytes (pinned for a truly asynchronous copy) that stay unchanged until upload_wait()"""
        assert data.dtype == np.uint8 and data.flags["C_CONTIGUOUS"] and data.size >= n * point_step
        self._ck(self._lib.gv_cloud_upload_pointcloud2_async(self._h, _ptr(data), C.c_size_t(n), C.c_uint32(point_step),
                                                             C.c_uint32(off_x), C.c_uint32(off_y), C.c_uint32(off_z)),
                 "upload_pointcloud2_async")
        self.n = n

    # ---- reference-surface calls
    def transform_lidar_to_camera(self):
        x, y, z = (np.empty(self.n, np.float32) for _ in range(3))
        self._ck(self._lib.gv_transform_lidar_to_camera(self._h, _ptr(x), _ptr(y), _ptr(z)), "transform")
        return x, y, z

    def extract_cloud_per_bbox(self, bboxes):
        b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        ids = np.empty(self.n, np.int32)
        counts = np.zeros(max(len(b), 1), np.int32)
        self._ck(self._lib.gv_extract_cloud_per_bbox(self._h, _ptr(b), C.c_int32(len(b)), _ptr(ids), _ptr(counts)),
                 "extract_cloud_per_bbox")
        return ids, counts[:len(b)]

    def compute_depth_for_bboxes(self, bboxes, k):
        b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        depths = np.zeros(len(b), np.float32)
        d2 = np.zeros((len(b), k), np.float32)
        self._ck(self._lib.gv_compute_depth_for_bboxes(self._h, _ptr(b), C.c_int32(len(b)), C.c_int32(k),
                                                       _ptr(depths), _ptr(d2)), "compute_depth_for_bboxes")
        return depths, d2

    def convert_pixels_to_3d(self, bboxes, depths):
        b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        d = _f32(depths)
        out = np.zeros((len(b), 3), np.float64)
        self._ck(self._lib.gv_convert_pixels_to_3d(self._h, _ptr(b), _ptr(d), C.c_int32(len(b)), _ptr(out)),
                 "convert_pixels_to_3d")
        return out

    def compute_bbox_pose(self, bboxes):
        b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        poses = np.zeros(max(len(b), 1), dtype=LSHAPE_DTYPE)
        valid = np.zeros(max(len(b), 1), np.uint8)
        self._ck(self._lib.gv_compute_bbox_pose(self._h, _ptr(b), C.c_int32(len(b)), _ptr(poses), _ptr(valid)),
                 "compute_bbox_pose")
        return poses[:len(b)], valid[:len(b)]

    def segment_ground_plane(self, threshold=0.04, iterations=50, seed=12345):
        mask = np.zeros(max(self.n, 1), np.uint8)
        coeff = np.zeros(4, np.float32)
        m = C.c_int64(0)
        self._ck(self._lib.gv_segment_ground_plane(self._h, C.c_double(threshold), C.c_int32(iterations),
                                                   C.c_uint64(seed), _ptr(mask), _ptr(coeff), C.byref(m)),
                 "segment_ground_plane")
        return m.value, mask[:self.n], coeff

    def compute_bbox_pose_ground_removed(self, bboxes):
        b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        poses = np.zeros(max(len(b), 1), dtype=LSHAPE_DTYPE)
        valid = np.zeros(max(len(b), 1), np.uint8)
        npz = C.c_int32(0)
        self._ck(self._lib.gv_compute_bbox_pose_ground_removed(self._h, _ptr(b), C.c_int32(len(b)), _ptr(poses),
                                                               _ptr(valid), C.byref(npz)), "compute_bbox_pose_gr")
        return poses[:len(b)], valid[:len(b)], npz.value

    def vision_post_process(self, orient, conf, dims, bboxes):
        o, c, d = _f32(orient), _f32(conf), _f32(dims)
        b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        poses = np.zeros(max(len(b), 1), dtype=LSHAPE_DTYPE)
        m = C.c_int32(0)
        self._ck(self._lib.gv_vision_post_process(self._h, _ptr(o), _ptr(c), _ptr(d), _ptr(b), C.c_int32(len(b)),
                                                  _ptr(poses), C.byref(m)), "vision_post_process")
        return poses[:m.value].copy()

    def transform_lshape_objects(self, poses):
        p = np.ascontiguousarray(poses, dtype=LSHAPE_DTYPE).copy()
        self._ck(self._lib.gv_transform_lshape_objects(self._h, _ptr(p), C.c_int32(len(p))), "transform_lshape")
        return p

    def intrinsics(self):
        k, ki = np.zeros(9), np.zeros(9)
        self._ck(self._lib.gv_get_intrinsics(self._h, _ptr(k), _ptr(ki)), "intrinsics")
        return k, ki

    def update_map(self):
        self._ck(self._lib.gv_update_map(self._h), "gv_update_map")

    def update_map_poses(self, poses):
        p = np.ascontiguousarray(poses, dtype=LSHAPE_DTYPE)
        self._ck(self._lib.gv_update_map_poses(self._h, _ptr(p), C.c_int32(len(p))), "gv_update_map_poses")

    def update_map_points(self, pts, bboxes):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
        self._ck(self._lib.gv_update_map_points(self._h, _ptr(pts), _ptr(b), C.c_int32(len(b))), "update_map_points")

    def to_occupancy_grid(self):
        data = np.empty(self.G, np.int8)
        info = GridInfo()
        self._ck(self._lib.gv_to_occupancy_grid(self._h, _ptr(data), C.byref(info)), "to_occupancy_grid")
        return data, info

    def to_occupancy_grid_async(self, pinned_i8):
        """pinned_i8: int8 view of G bytes of pinned memory; complete once gv_stream has passed this point"""
        assert pinned_i8.dtype == np.int8 and pinned_i8.size == self.G
        self._ck(self._lib.gv_to_occupancy_grid_async(self._h, _ptr(pinned_i8)), "to_occupancy_grid_async")

    def publish_grid_async(self, pinned_i8):
        """the packed grid to PINNED host memory by a kernel on the public stream (the copy engines stay with the uploads)"""
        assert pinned_i8.dtype == np.int8 and pinned_i8.size == self.G
        self._ck(self._lib.gv_publish_grid_async(self._h, _ptr(pinned_i8)), "gv_publish_grid_async")

    def log_odds(self):
        out = np.empty(self.G, np.float32)
        self._ck(self._lib.gv_get_log_odds(self._h, _ptr(out)), "get_log_odds")
        return out

    def occupancy(self):
        out = np.empty(self.G, np.float32)
        self._ck(self._lib.gv_get_occupancy(self._h, _ptr(out)), "get_occupancy")
        return out

    def set_log_odds(self, a):
        a = _f32(a)
        assert a.size == self.G
        self._ck(self._lib.gv_set_log_odds(self._h, _ptr(a)), "set_log_odds")

    # ---- fused frame
    def _desc(self, flags, bboxes=None, poses=None, net=None):
        self._keep = []
        d = FrameDesc()
        d.flags = flags
        if bboxes is not None and len(bboxes):
            b = np.ascontiguousarray(bboxes, dtype=BBOX_DTYPE)
            self._keep.append(b)
            d.bboxes, d.n_bboxes = b.ctypes.data, len(b)
        if poses is not None and len(poses):
            p = np.ascontiguousarray(poses, dtype=LSHAPE_DTYPE)
            self._keep.append(p)
            d.poses, d.n_poses = p.ctypes.data, len(p)
        if net is not None:
            o, c, dm = (_f32(t) for t in net)
            self._keep += [o, c, dm]
            d.orient, d.conf, d.dims = o.ctypes.data, c.ctypes.data, dm.ctypes.data
        return d

    def set_detections(self, flags, bboxes=None, poses=None, net=None):
        d = self._desc(flags, bboxes, poses, net)
        self._ck(self._lib.gv_frame_set_detections(self._h, C.byref(d)), "frame_set_detections")

    def set_detections_async(self, flags, bboxes=None, poses=None, net=None):
        d = self._desc(flags, bboxes, poses, net)
        self._ck(self._lib.gv_frame_set_detections_async(self._h, C.byref(d)), "frame_set_detections_async")

    def frame_fence(self):
        self._ck(self._lib.gv_frame_fence(self._h), "frame_fence")

    def stream(self):
        return self._lib.gv_stream(self._h)

    def device_layers(self):
        """device addresses (ints) of occ_i8, log_odds, occupancy"""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self._lib.gv_device_layers(self._h, C.byref(a), C.byref(b), C.byref(c)), "device_layers")
        return a.value, b.value, c.value

    def enqueue_frame(self):
        self._ck(self._lib.gv_frame_enqueue(self._h), "frame_enqueue")

    def synchronize(self):
        self._ck(self._lib.gv_synchronize(self._h), "synchronize")

    def process_frame(self, flags, bboxes=None, poses=None, net=None):
        d = self._desc(flags, bboxes, poses, net)
        self._ck(self._lib.gv_process_frame(self._h, C.byref(d)), "process_frame")

    def hits(self):
        out = np.empty(self.G, np.int32)
        self._ck(self._lib.gv_get_hits(self._h, _ptr(out)), "get_hits")
        return out

    def miss(self):
        out = np.empty(self.G, np.int32)
        self._ck(self._lib.gv_get_miss(self._h, _ptr(out)), "get_miss")
        return out

    def cell_idx(self):
        out = np.empty(self.n, np.int32)
        self._ck(self._lib.gv_get_cell_idx(self._h, _ptr(out)), "get_cell_idx")
        return out

    def bbox_id(self):
        out = np.empty(self.n, np.int32)
        self._ck(self._lib.gv_get_bbox_id(self._h, _ptr(out)), "get_bbox_id")
        return out

    def ray_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._lib.gv_get_ray_stats(self._h, C.byref(a), C.byref(b)), "ray_stats")
        return a.value, b.value

    # ---- the node's tick (timerCallback from filterBBoxes on, one host wait)
    def tick_enqueue(self, bboxes, k_near=4, net=None, vision=False, lidar_bin=False, lidar_raymarch=False, grid_out=None):
        """net: (orient, conf, dims) of the DYNAMIC boxes in filter_bboxes order (vision=True); grid_out: PinnedI8 array"""
        b = np.ascontiguousarray(bboxes if bboxes is not None else np.zeros(0, BBOX_DTYPE), dtype=BBOX_DTYPE)
        d = TickDesc()
        d.flags = (TICK_VISION_ORIENT if vision else 0) | (TICK_LIDAR_BIN if lidar_bin else 0) | (TICK_LIDAR_RAYMARCH if lidar_raymarch else 0)
        d.bboxes, d.n_bboxes = (b.ctypes.data if len(b) else None), len(b)
        keep = [b]
        if net is not None:
            o, c, dm = (_f32(t) for t in net)
            keep += [o, c, dm]
            d.orient, d.conf, d.dims, d.n_net = o.ctypes.data, c.ctypes.data, dm.ctypes.data, len(o.reshape(-1, 4))
        d.k_near = k_near
        if grid_out is not None:
            assert grid_out.dtype == np.int8 and grid_out.size == self.G
            d.grid_out = grid_out.ctypes.data
        self._tick_keep = keep
        self._tick_nb = len(b)
        self._ck(self._lib.gv_tick_enqueue(self._h, C.byref(d)), "gv_tick_enqueue")

    def tick_wait(self):
        nb = max(self._tick_nb, 1)
        st = np.zeros(nb, dtype=BBOX_DTYPE)
        depths = np.zeros(nb, np.float32)
        pts = np.zeros((nb, 3), np.float64)
        poses = np.zeros(nb, dtype=LSHAPE_DTYPE)
        r = TickResult()
        r.static_bboxes, r.depths, r.base_points_xyz, r.poses = st.ctypes.data, depths.ctypes.data, pts.ctypes.data, poses.ctypes.data
        self._ck(self._lib.gv_tick_wait(self._h, C.byref(r)), "gv_tick_wait")
        return {"n_static": r.n_static, "n_dynamic": r.n_dynamic, "static_bboxes": st[:r.n_static], "depths": depths[:r.n_static],
                "base_points": pts[:r.n_static], "poses": poses[:r.n_poses], "pca_empty": bool(r.pca_empty)}

    def tick(self, bboxes, **kw):
        self.tick_enqueue(bboxes, **kw)
        return self.tick_wait()

    # ---- multi-GPU (RCCL inside the library)
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = load().gv_comm_unique_id(buf)
        if rc:
            raise GVError(rc, "gv_comm_unique_id")
        return bytes(buf)

    def comm_init(self, uid: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._ck(self._lib.gv_comm_init(self._h, buf, C.c_int32(rank), C.c_int32(world)), "gv_comm_init")

    def comm_info(self):
        """(ranks in the communicator, this rank, its device) as RCCL reports them"""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.gv_comm_info(self._h, C.byref(a), C.byref(b), C.byref(c)), "gv_comm_info")
        return a.value, b.value, c.value

    def comm_destroy(self):
        self._ck(self._lib.gv_comm_destroy(self._h), "gv_comm_destroy")

    def comm_band(self):
        b, e = C.c_int64(), C.c_int64()
        self._ck(self._lib.gv_comm_band(self._h, C.byref(b), C.byref(e)), "gv_comm_band")
        return b.value, e.value

    def enqueue_frame_sharded(self):
        self._ck(self._lib.gv_frame_enqueue_sharded(self._h), "frame_enqueue_sharded")

    def time_frame_sharded_stages(self, frames):
        ms = (C.c_float * 6)()
        self._ck(self._lib.gv_time_frame_sharded_stages(self._h, C.c_int32(frames), ms), "time_frame_sharded_stages")
        return dict(zip(("bin", "exchange_ends", "sectors", "exchange_free", "grid_pass", "gather"), [float(v) for v in ms]))

    def process_frame_sharded(self, flags, bboxes=None, poses=None, net=None):
        d = self._desc(flags, bboxes, poses, net)
        self._ck(self._lib.gv_process_frame_sharded(self._h, C.byref(d)), "process_frame_sharded")

    def frame_sharded_emulated(self, world, flags, bboxes=None, poses=None, net=None):
        d = self._desc(flags, bboxes, poses, net)
        self._ck(self._lib.gv_test_frame_sharded_emulated(self._h, C.byref(d), C.c_int32(world)), "frame_sharded_emulated")

    def time_frames(self, frames):
        ms = C.c_float(0)
        self._ck(self._lib.gv_time_frames(self._h, C.c_int32(frames), C.byref(ms)), "time_frames")
        return ms.value

    def time_frame_stages(self, frames):
        ms = (C.c_float * len(STAGES))()
        self._ck(self._lib.gv_time_frame_stages(self._h, C.c_int32(frames), ms), "time_frame_stages")
        return dict(zip(STAGES, [float(v) for v in ms]))
