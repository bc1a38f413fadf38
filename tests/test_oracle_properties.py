"""CPU tests of the oracle and the host-side logic (no GPU): properties the domain offers,
the frozen fixture, and the host-only entry points of the C ABI."""
import ctypes as C
import os
import re

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle_lib as ol
from gvamd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ---------------------------------------------------------------- getIndex ----
@settings(max_examples=300, deadline=None)
@given(st.floats(-60, 60, allow_nan=False), st.floats(-30, 30, allow_nan=False))
def test_index_in_range_iff_inside_map(x, y):
    g = ol.OGrid(50, 20, 0.1)
    ok, ix, iy = g.get_index(x, y)
    inside = (-9.0 < x <= 41.0) and (-10.0 < y <= 10.0)
    if abs(x - 41.0) > 1e-9 and abs(x + 9.0) > 1e-9 and abs(abs(y) - 10.0) > 1e-9:
        assert ok == inside
    if ok:
        assert 0 <= ix < 500 and 0 <= iy < 200
        # the cell's extent contains the point (cell (0,0) is the +x,+y corner)
        assert 41.0 - (ix + 1) * 0.1 - 1e-9 <= x <= 41.0 - ix * 0.1 + 1e-9
        assert 10.0 - (iy + 1) * 0.1 - 1e-9 <= y <= 10.0 - iy * 0.1 + 1e-9


def test_count_conservation_and_cell_consistency():
    cfg = synth.CONFIGS[1]["grid"]
    g = ol.OGrid(cfg.grid_x, cfg.grid_y, cfg.resolution)
    tfs = synth.transforms(True)
    m = ol.tf_to_matrix4f(tfs["base_lidar"])
    x, y, z, _ = synth.cloud_uniform(1)
    hits, cell = g.bin_points(m, x, y, z)
    inmap = cell >= 0
    assert hits.sum() == inmap.sum() > 0
    assert 0.1 < 1 - inmap.mean() < 0.25, "the fixture keeps ~17% of the points outside the map"
    assert np.array_equal(np.bincount(cell[inmap], minlength=g.G), hits)
    # cell index agrees with getIndex on the transformed point
    bx, by, bz = ol.transform_cloud(m, x, y, z)
    for i in range(0, len(x), 997):
        ok, ix, iy = g.get_index(float(bx[i]), float(by[i]))
        assert (cell[i] >= 0) == ok
        if ok:
            assert cell[i] == iy * g.nx + ix


def test_raymarch_dedupe_equals_per_point_and_is_monotone():
    cfg = synth.CONFIGS[1]["grid"]
    g = ol.OGrid(cfg.grid_x, cfg.grid_y, cfg.resolution)
    m = ol.tf_to_matrix4f(synth.transforms(False)["base_lidar"])
    x, y, z, _ = synth.cloud_lidar_like(1, 3000)
    a, va = g.raymarch(m, x, y, z, dedupe=True)
    b, vb = g.raymarch(m, x, y, z, dedupe=False)
    assert np.array_equal(a, b) and va <= vb
    # more points can only free more cells
    c, _ = g.raymarch(m, x[:1500], y[:1500], z[:1500])
    assert np.all(a >= c)
    # the sensor's own cell is traversed by every ray that leaves it
    ok, ox, oy = g.get_index(0.0, 0.0)
    assert ok and a[oy * g.nx + ox] == 1


def test_bresenham_line_matches_line_iterator_semantics():
    """one ray: cells visited = LineIterator(O, E) without E (hit) / with E (clipped)."""
    g = ol.OGrid(100, 100, 0.5)
    m = ol.tf_to_matrix4f(np.array([0, 0, 0, 1, 0.0, 0.0, 0.0]))
    ok, ox, oy = g.get_index(0.0, 0.0)
    for px, py in [(10.3, 4.2), (-7.7, 12.1), (3.0, -20.2), (-5.1, -5.1), (40.0, 0.2), (0.1, 30.0)]:
        miss, visits = g.raymarch(m, [px], [py], [0.0])
        ok2, ex, ey = g.get_index(px, py)
        assert ok2
        dx, dy = abs(ex - ox), abs(ey - oy)
        sx, sy = (1 if ex >= ox else -1), (1 if ey >= oy else -1)
        cells = []
        cx, cy = ox, oy
        if dx >= dy:
            den, num, add, n = dx, dx // 2, dy, dx + 1
        else:
            den, num, add, n = dy, dy // 2, dx, dy + 1
        for _ in range(n):
            cells.append((cx, cy))
            num += add
            if num >= den:
                num -= den
                if dx >= dy:
                    cy += sy
                else:
                    cx += sx
            if dx >= dy:
                cx += sx
            else:
                cy += sy
        assert cells[-1] == (ex, ey)
        exp = np.zeros(g.G, np.uint8)
        for (cx, cy) in cells[:-1]:
            exp[cy * g.nx + cx] = 1
        assert np.array_equal(miss, exp) and visits == n - 1
    # an out-of-map point: the ray is clipped to the border and its end cell counts as free
    miss, visits = g.raymarch(m, [500.0], [3.0], [0.0])
    k, ex, ey = g.ray_end(0.0, 0.0, 500.0, 3.0)
    assert k == 2 and ex == 0 and miss[ey * g.nx + ex] == 1


# --------------------------------------------------------------- transforms ----
def test_rigid_transform_against_fp64():
    tfs = synth.transforms(True)
    for key in ("base_lidar", "cam_lidar"):
        m = ol.tf_to_matrix4f(tfs[key]).reshape(4, 4)
        q = tfs[key][:4]
        # rotation part is orthonormal to fp32 accuracy and matches the quaternion
        r = m[:3, :3].astype(np.float64)
        assert np.allclose(r @ r.T, np.eye(3), atol=3e-7)
        x, y, z, w = q
        ref = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(r, ref, atol=3e-7)
    x, y, z, _ = synth.cloud_uniform(1, 500)
    m = ol.tf_to_matrix4f(tfs["cam_lidar"])
    ox, oy, oz = ol.transform_cloud(m, x, y, z)
    p = np.stack([x, y, z, np.ones_like(x)]).astype(np.float64)
    ref = m.reshape(4, 4).astype(np.float64) @ p
    assert np.allclose(np.stack([ox, oy, oz]), ref[:3], rtol=2e-6, atol=2e-5)
    # pose transform: position through tf_point, composition is a rotation
    pose = np.array([1.0, 2.0, 3.0, 0.0, 0.0, 0.38268343236508978, 0.92387953251128674])
    out = ol.tf_pose(tfs["base_cam"], pose)
    assert out[:3].tolist() == ol.tf_point(tfs["base_cam"], pose[:3]).tolist()
    assert abs(np.linalg.norm(out[3:]) - 1.0) < 1e-12


# -------------------------------------------------------------- detections ----
def test_bbox_first_match_rule():
    K = ol.set_intrinsic(320.0, 320.0, 320.0, 240.0)
    b = np.array([(100, 100, 300, 300, 0.9, 9), (200, 200, 400, 400, 0.8, 2), (0, 0, 639, 479, 0.7, 0)],
                 dtype=synth.BBOX_DTYPE)
    # camera-frame points projecting to chosen pixels at z = 2: u = 160x + 320, v = 160y + 240
    def cam(u, v):
        return (u - 320.0) / 160.0, (v - 240.0) / 160.0, 2.0
    pts = [cam(250, 250), cam(350, 350), cam(50, 50), cam(300, 300), cam(300.5, 300)]
    x, y, z = (np.array(t, np.float32) for t in zip(*pts))
    ids = ol.extract_cloud_per_bbox(K, x, y, z, b, 640, 480)
    assert ids.tolist() == [0, 1, 2, 0, 1]          # overlap -> first box; inclusive max edge
    # behind the camera, z <= 0.001, non-finite, outside the image
    x = np.array([0, 0, np.nan, 5.0], np.float32); y = np.zeros(4, np.float32); z = np.array([-1, 0.001, 1, 1], np.float32)
    assert ol.extract_cloud_per_bbox(K, x, y, z, b, 640, 480).tolist() == [-1, -1, -1, -1]


def test_nms_and_filter_properties():
    s = synth.Stream(4321, 2)
    n, c = 200, 10
    ctr = s.uniform(2 * n, 0.2, 0.8).reshape(n, 2)
    wh = s.uniform(2 * n, 0.02, 0.3).reshape(n, 2)
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
    scores = s.uniform(n * c, 0.0, 1.0).reshape(n, c).astype(np.float32)
    out = ol.extract_bboxes(boxes, scores, 0.6, 0.5, 640, 480, 416)
    assert 0 < len(out) < n
    conf = out["confidence"]
    assert np.all(conf[:-1] >= conf[1:]) and conf.min() >= np.float32(0.6)
    for f in ("x_min", "y_min", "x_max", "y_max"):
        assert np.array_equal(out[f], np.trunc(out[f]))      # int truncation (object_detection.cpp:234-237)
    stat, dyn = ol.filter_bboxes(out)
    assert len(stat) + len(dyn) == len(out)
    assert set(dyn["label"]) <= {0, 1, 2, 9} and not (set(stat["label"]) & {0, 1, 2, 9})


def test_vision_orientation_geometry():
    cam = ol.make_cam()
    bins = ol.generate_bins(2)
    # alpha = atan2(sin, cos) + bin centre - pi
    a = ol.compute_alpha([1.0, 0.0, 0.0, 1.0], 0, bins)
    assert a == pytest.approx(float(np.float32(np.pi / 2) - np.float32(np.pi)), abs=1e-6)
    # a box straight ahead of a 4 m car: located on the optical axis, in front of the camera
    b = np.array([(270, 200, 370, 280, 0.9, 9)], dtype=synth.BBOX_DTYPE)
    th = ol.compute_theta_ray(cam, b)
    assert th == 0.0
    pose, err = ol.calc_location(cam, [3.9, 1.6, 1.5], b, float(np.float32(-np.pi / 2)), th)
    assert pose[2] > 2.0 and abs(pose[0]) < 1.0 and np.isfinite(err)
    poses = ol.post_process(cam, [[1, 0, 0, 1]], [[0.9, 0.1]], [[0, 0, 0]], b)
    assert len(poses) == 1 and poses[0]["length"] == pytest.approx(3.884, abs=1e-6)
    b["label"] = 5   # traffic light: skipped (vision_orientation.cpp:496-499)
    assert len(ol.post_process(cam, [[1, 0, 0, 1]], [[0.9, 0.1]], [[0, 0, 0]], b)) == 0


def test_knn_depth_upper_median_and_radius_filter():
    u = np.array([10, 11, 12, 13, 300], np.float32); v = np.array([10, 10, 10, 10, 300], np.float32)
    d = np.array([5, 1, 9, 3, 0.5], np.float32)
    b = np.array([(0, 0, 22, 20, 0.9, 3)], dtype=synth.BBOX_DTYPE)     # centre (11, 10)
    depths, d2 = ol.depth_for_bboxes(u, v, d, b, 4)
    assert depths[0] == 5.0          # depths {1,3,5,9} -> element at size/2 = upper median
    assert np.all(np.diff(d2[0]) >= 0)
    x = np.concatenate([np.zeros(11), [5.0]]).astype(np.float32) + np.linspace(0, 0.1, 12).astype(np.float32) * 0
    keep = ol.radius_outlier(x, np.zeros(12, np.float32), np.zeros(12, np.float32))
    assert keep.tolist() == [1] * 11 + [0]       # 11 coincident points keep each other; the loner goes


# ------------------------------------------------------------ frozen fixture ----
def test_oracle_matches_frozen_fixture():
    sys_path = os.path.join(HERE, "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_frame_fixture", os.path.join(sys_path, "make_frame_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.compute()
    gold = np.load(os.path.join(sys_path, "frame_small.npz"))
    assert set(now) == set(gold.files)
    for k in gold.files:
        assert np.array_equal(now[k], gold[k]), k


# ------------------------------------------------------------------ C ABI ----
def _header_functions():
    txt = open(os.path.join(ROOT, "include", "gridvision_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gv_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import gvamd
    lib = gvamd.load()
    declared = _header_functions()
    assert len(declared) >= 40
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(gvamd.ABI_SYMBOLS) == declared, "gvamd.ABI_SYMBOLS must list exactly the header's functions"
    assert lib.gv_abi_version() == 4


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_gpu():
    import gvamd
    with pytest.raises(gvamd.GVError) as e:
        gvamd.GridVisionHIP(50, 20, 0.1)
    assert e.value.code == 4      # GV_ERR_NO_DEVICE: the product path fails loudly, it never falls back


def test_host_side_entry_points_match_oracle():
    import gvamd
    s = synth.Stream(99, 4)
    n, c = 400, 10
    ctr = s.uniform(2 * n, 0.1, 0.9).reshape(n, 2)
    wh = s.uniform(2 * n, 0.02, 0.4).reshape(n, 2)
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
    scores = s.uniform(n * c, 0.0, 1.0).reshape(n, c).astype(np.float32)
    for thr, iou in [(0.5, 0.4), (0.6, 0.6), (0.9, 0.3), (1.1, 0.5)]:
        got = gvamd.extract_bboxes(boxes, scores, thr, iou, 640, 480, 416)
        exp = ol.extract_bboxes(boxes, scores, thr, iou, 640, 480, 416)
        assert got.tobytes() == exp.tobytes()
        gs, gd = gvamd.filter_bboxes(got)
        es, ed = ol.filter_bboxes(exp)
        assert gs.tobytes() == es.tobytes() and gd.tobytes() == ed.tobytes()
    lib = gvamd.load()
    assert lib.gv_extract_bboxes(None, None, C.c_int32(5), C.c_int32(10), C.c_double(0.5), C.c_double(0.5),
                                 C.c_int32(640), C.c_int32(480), C.c_int32(416), None, None) == 1   # GV_ERR_BAD_ARG
    assert lib.gv_destroy(None) == 1 and lib.gv_update_map(None) == 1


def test_bench_refuses_to_shrink_a_multi_gpu_job():
    """bench.py --gpus N without a torchrun environment launches the ranks itself; with fewer GPUs
    visible than asked for it must fail loudly instead of reporting n_gpus: 1 (no GPU here: exit 2)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, timeout=120)
    assert p.returncode != 0
    assert b'"n_gpus"' not in p.stdout
    assert b"refusing" in p.stderr


def test_shard_band_and_slice_helpers_partition_the_grid():
    """gv_shard_band_rows / gv_shard_slice_words (pure host functions of the C ABI, the ones the sharded frame and
    the multi-rank CPU test use): bands are whole 64-row blocks, contiguous, cover [0, ny) exactly once for every
    world size; slices are multiples of 4 words and `world` of them cover the bitmap."""
    import gvamd
    for ny in (1, 63, 64, 65, 200, 800, 1000, 2000, 4000, 8000):
        for world in (1, 2, 3, 5, 8, 16):
            bands = [gvamd.shard_band_rows(r, world, ny) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == ny
            for r in range(world):
                y0, y1 = bands[r]
                assert 0 <= y0 <= y1 <= ny and (y0 % 64 == 0 or y0 == ny) and (y1 % 64 == 0 or y1 == ny)
                if r + 1 < world:
                    assert bands[r + 1][0] == y1
    for words in (0, 1, 5, 1000, 2_097_152 + 3):
        for world in (1, 2, 3, 8):
            sl = gvamd.shard_slice_words(words, world)
            assert sl % 4 == 0 and sl * world >= words and (sl - 4) * world < max(words, 1) + 4 * world


def test_oracle_reproduces_pca_fixture():
    """tests/golden/pca_small.npz (made by tests/golden/make_pca_fixture.py from the oracle): kNN depths, the RANSAC
    ground plane and the PCA poses of a small seeded scene.  Freezes the oracle's arithmetic on the reference's other
    per-frame loops against accidental change."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_pca_fixture", os.path.join(HERE, "golden", "make_pca_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = np.load(os.path.join(HERE, "golden", "pca_small.npz"))
    now = mk.compute()
    assert sorted(gold.files) == sorted(now.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], now[k]), k
    assert int(gold["ground_n"][0]) > 8000 and gold["pose_valid"].sum() >= 6


def test_radius_outlier_grid_equals_all_pairs():
    """oracle/cloud_detections.c: the cell-grid form of RadiusOutlierRemoval (what bench.py times as the CPU baseline of
    the PCA tick, and what the 1 M-point tests may use) keeps exactly the points the all-pairs statement keeps:
    clustered clouds, points exactly on cell borders and at the radius, NaN / far-away coordinates, tiny clouds."""
    rng = np.random.default_rng(21)
    clouds = []
    for trial in range(6):
        cen = rng.uniform(-20, 20, (8, 3))
        pts = np.concatenate([cen[i] + rng.normal(0, rng.uniform(0.05, 0.6), (int(rng.integers(5, 400)), 3)) for i in range(8)]
                             + [rng.uniform(-25, 25, (500, 3))])
        clouds.append(pts)
    lattice = np.stack(np.meshgrid(np.arange(-4, 5), np.arange(-4, 5), np.arange(-2, 3)), -1).reshape(-1, 3) * 0.41
    clouds.append(lattice)                         # points on the borders of the 0.41 m cells
    clouds.append(lattice * (0.4 / 0.41))          # neighbours exactly at the radius
    wild = rng.normal(0, 0.2, (300, 3))
    wild[0] = [np.nan, 0, 0]; wild[1] = [3e7, 0, 0]; wild[2] = [-3e7, 1, 1]; wild[3] = [np.inf, 0, 0]
    clouds.append(wild)
    clouds.append(rng.normal(0, 0.1, (7, 3)))      # fewer points than min_pts + 1
    clouds.append(np.zeros((0, 3)))
    for pts in clouds:
        pts = pts.astype(np.float32)
        x, y, z = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
        a = ol.radius_outlier(x, y, z, 0.4, 10)
        b = ol.radius_outlier_grid(x, y, z, 0.4, 10)
        assert np.array_equal(a, b)
    assert ol.radius_outlier_grid(*[c.astype(np.float32) for c in clouds[0].T], 0.4, 10).mean() > 0.2
