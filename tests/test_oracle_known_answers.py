"""Pins the CPU oracle to the hand-derivable known answers of SURVEY.md 8(c).

The reference has no tests or fixtures of its own ("parity unpinned", see
oracle/gv_oracle.h); these are the only vectors that exist for this path.
"""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as ol
from gvamd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
    KA = json.load(f)


def test_grid_geometry():
    for c in KA["grid_sizes"]:
        g = ol.OGrid(c["grid_x"], c["grid_y"], c["res"])
        assert (g.nx, g.ny) == (c["nx"], c["ny"])
        assert g.g.pos_x == c["pos_x"] and g.g.pos_y == 0.0
        assert np.all(g.log_odds == 0.0) and np.all(g.occupancy == 0.5)


def test_get_index_known_answers():
    d = KA["default_yaml_grid"]
    g = ol.OGrid(d["grid_x"], d["grid_y"], d["resolution"])
    for c in KA["get_index"]:
        ok, ix, iy = g.get_index(*c["p"])
        assert ok == c["valid"], c
        if ok:
            assert [ix, iy] == c["idx"], c


def test_get_index_rejects_non_finite():
    g = ol.OGrid(50, 20, 0.1)
    for p in [(math.nan, 0.0), (0.0, math.inf), (-math.inf, 0.0)]:
        assert g.get_index(*p)[0] is False


def test_empty_frame_sequence():
    g = ol.OGrid(50, 20, 0.1)
    seq = []
    for k in range(12):
        g.update_map()
        seq.append(float(g.log_odds[0]))
        assert np.all(g.log_odds == g.log_odds[0])
        if k == 0:
            assert g.occupancy[0] == pytest.approx(KA["empty_frame_first"]["occupancy"], abs=1e-7)
            assert g.to_occupancy_grid()[0][0] == KA["empty_frame_first"]["int8"]
    assert np.array(seq, dtype=np.float32).tolist() == np.array(KA["empty_frame_log_odds"], dtype=np.float32).tolist()
    assert g.occupancy[0] == pytest.approx(KA["empty_frame_saturated"]["occupancy"], abs=1e-7)
    data, info = g.to_occupancy_grid()
    assert np.all(data == KA["empty_frame_saturated"]["int8"])
    assert info.tolist() == [500.0, 200.0, 0.1, 16.0 - 25.0, -10.0]


def test_object_every_frame_sequence():
    g = ol.OGrid(50, 20, 0.1)
    pose = np.zeros(1, dtype=synth.LSHAPE_DTYPE)
    pose["px"], pose["py"], pose["length"], pose["width"], pose["qw"] = 16.0, 0.0, 2.0, 1.0, 1.0
    ok, ix, iy = g.get_index(16.0, 0.0)
    k = iy * g.nx + ix
    seq = []
    for _ in range(7):
        g.update_map_poses(pose)
        seq.append(float(g.log_odds[k]))
    assert np.array(seq, dtype=np.float32).tolist() == np.array(KA["object_every_frame_log_odds"], dtype=np.float32).tolist()
    assert g.occupancy[k] == pytest.approx(KA["object_saturated"]["occupancy"], abs=1e-6)
    assert g.to_occupancy_grid()[0][g.G - 1 - k] == KA["object_saturated"]["int8"]


def test_rect_block_add_6x4():
    c = KA["rect_block_add_6x4"]
    g = ol.OGrid(c["grid_x"], c["grid_y"], c["res"])
    assert (g.nx, g.ny) == (6, 4)
    pose = np.zeros(1, dtype=synth.LSHAPE_DTYPE)
    for k in ("px", "py", "length", "width"):
        pose[k] = c["pose"][k]
    g.update_map_poses(pose)
    lo = g.log_odds.reshape(g.ny, g.nx)  # [iy, ix]
    exp = np.full((4, 6), np.float32(-0.2), dtype=np.float32)
    x0, x1 = c["ix_range"]
    y0, y1 = c["iy_range"]
    exp[y0:y1 + 1, x0:x1 + 1] = np.float32(-0.2) + np.float32(0.85)
    assert np.array_equal(lo, exp)


def test_rect_any_corner_outside_skips_box():
    g = ol.OGrid(50, 20, 0.1)
    # corner order {left_back, left_front, right_front, right_back}; 3 inside, 1 outside
    assert g.update_cells_fast([0, 0, 1, 0, 1, 1, 0, 10.5]) == 0
    assert np.all(g.log_odds == 0.0)
    assert g.update_cells_fast([0, 0, 1, 0, 1, 1, 0, 1]) == 1
    assert np.count_nonzero(g.log_odds) == 11 * 11


def test_sigmoid_values():
    g = ol.OGrid(6, 4, 1.0)
    for l, p in [(0.0, 0.5), (-2.0, 0.119202934), (3.6, 0.973403)]:
        g.log_odds[:] = np.float32(l) - np.float32(-0.2) if False else np.float32(l)
        ol.lib().gvo_clamp_and_sigmoid(__import__("ctypes").byref(g.g))
        assert g.occupancy[0] == pytest.approx(p, abs=1e-6)


def test_theta_ray_and_bins():
    t = KA["theta_ray"]
    cam = ol.make_cam(fx=t["fx"], w=t["W"])
    for c in t["cases"]:
        b = np.zeros(1, dtype=synth.BBOX_DTYPE)
        b["x_min"], b["x_max"] = c["x_min"], c["x_max"]
        assert ol.compute_theta_ray(cam, b) == pytest.approx(c["theta"], abs=1e-7)
    assert ol.generate_bins(2).tolist() == np.array(KA["bins2"], dtype=np.float32).tolist()


def test_denormalize_and_projection():
    d = KA["denormalize"]
    b = np.zeros(1, dtype=synth.BBOX_DTYPE)
    b["x_min"] = b["x_max"] = d["norm"]
    out = ol.denormalize(b, d["orig_w"], 480, d["resize"])
    assert out["x_min"][0] == d["expect"] and out["x_max"][0] == d["expect"]
    p = KA["projection"]
    K = ol.set_intrinsic(*p["K"])
    u, v, z = ol.project_points(K, [p["p_cam"][0]], [p["p_cam"][1]], [p["p_cam"][2]])
    assert (u[0], v[0], z[0]) == (p["u"], p["v"], p["p_cam"][2])
    ids = ol.extract_cloud_per_bbox(K, [1.0], [0.5], [2.0],
                                    np.array([(470, 300, 490, 330, 0.9, 9)], dtype=synth.BBOX_DTYPE), 640, 480)
    assert ids.tolist() == [0]


def test_k_inverse_closed_form():
    K = ol.set_intrinsic(320.0, 320.0, 320.0, 240.0)
    Ki = ol.k_inverse(K).reshape(3, 3)
    exp = np.array([[1 / 320.0, 0, -1.0], [0, 1 / 320.0, -0.75], [0, 0, 1.0]])
    assert np.allclose(Ki, exp, rtol=0, atol=1e-15)
    assert np.allclose(Ki @ K.reshape(3, 3), np.eye(3), atol=1e-12)


def test_pca_angle_promotion_canaries():
    """cloud_detections.cpp:227 `std::atan2(major.y, major.x) * 180.0f / CV_PI`: the float product is divided by the
    DOUBLE CV_PI and narrowed once.  At (2, 1) and (5, 12) that differs from an all-float division by one ulp
    (63.43495 vs 63.434948, 22.619865 vs 22.619864): the round-3 restatement used the float form."""
    f = np.float32
    for my, mx, want, float_form in ((2.0, 1.0, 63.43495, 63.434948), (5.0, 12.0, 22.619865, 22.619864)):
        got = f(ol.pca_angle_deg(my, mx))
        assert got == f(want) and got != f(float_form)
        prod = f(f(np.arctan2(f(my), f(mx))) * f(180.0))
        assert got == f(np.float64(prod) / 3.1415926535897932384626433832795)
    assert ol.pca_angle_deg(0.0, 1.0) == 0.0 and f(ol.pca_angle_deg(1.0, 0.0)) == f(90.0)
