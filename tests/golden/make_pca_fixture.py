"""Generates tests/golden/pca_small.npz from the CPU oracle: the outputs of the reference's other per-frame loops
on a small seeded scene (ground plane + clusters + clutter) -- kNN depths and distances (cloud_detections.cpp:8-87),
the RANSAC ground plane by outcome (:105-138; oracle/ransac.c is its definition), and computeBBoxPose with and
without ground removal (:140-321).  Inputs are re-derived from the seeded generator; the file holds expected outputs
only.  Run from the repo root:  python tests/golden/make_pca_fixture.py
There is no reference binary to generate vectors from; this freezes the oracle (the round-3 change of the plane
refinement to one-pass moments would have shown up here) and gives the GPU tests a second comparison point."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "grid-vision_amd"))
import oracle_lib as ol  # noqa: E402
from gvamd import synth  # noqa: E402


def scene():
    """20 k points: a tilted ground plane, 8 dense clusters in front of the camera, uniform clutter; 8 bboxes"""
    rng = np.random.default_rng(2024)
    tfs = synth.transforms(True)
    n_c, per = 8, 700
    ox = rng.uniform(6.0, 40.0, n_c); oy = rng.uniform(-0.5, 0.5, n_c) * ox; oz = rng.uniform(-1.0, 0.3, n_c)
    xs, ys, zs = [], [], []
    for k in range(n_c):
        a = rng.uniform(0, np.pi)
        u, v = rng.uniform(-1.8, 1.8, per), rng.uniform(-0.6, 0.6, per)
        xs.append(ox[k] + u * np.cos(a) - v * np.sin(a)); ys.append(oy[k] + u * np.sin(a) + v * np.cos(a))
        zs.append(oz[k] + rng.uniform(-0.4, 0.4, per))
    ng = 9000
    gx = rng.uniform(1.0, 60.0, ng); gy = rng.uniform(-35.0, 35.0, ng)
    xs.append(gx); ys.append(gy); zs.append(-1.7 + 0.006 * gx + rng.normal(0, 0.012, ng))
    nu = 20000 - n_c * per - ng
    xs.append(rng.uniform(-20, 70, nu)); ys.append(rng.uniform(-40, 40, nu)); zs.append(rng.uniform(-2, 3, nu))
    x = np.concatenate(xs).astype(np.float32); y = np.concatenate(ys).astype(np.float32); z = np.concatenate(zs).astype(np.float32)
    perm = rng.permutation(len(x))
    x, y, z = x[perm], y[perm], z[perm]
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
    inv = np.argsort(perm)
    boxes = []
    for k in range(n_c):
        idx = inv[k * per:(k + 1) * per]
        ok = cz[idx] > 0.1
        u = synth.FX * cx[idx][ok] / cz[idx][ok] + synth.CX
        v = synth.FY * cy[idx][ok] / cz[idx][ok] + synth.CY
        boxes.append((float(np.floor(max(0, u.min() - 1))) + 0.25, float(np.floor(max(0, v.min() - 1))) + 0.5,
                      float(np.ceil(min(639, u.max() + 1))) + 0.75, float(np.ceil(min(479, v.max() + 1)))))
    b = np.zeros(len(boxes), dtype=synth.BBOX_DTYPE)
    for i, (x0, y0, x1, y1) in enumerate(boxes):
        b[i] = (x0, y0, x1, y1, 0.95 - 0.01 * i, [9, 2, 0, 1, 5][i % 5])
    return tfs, x, y, z, (cx, cy, cz), b


POSE_FIELDS = ("px", "py", "pz", "qx", "qy", "qz", "qw", "length", "width", "height")


def poses_of(cx, cy, cz, K, b):
    ids = ol.extract_cloud_per_bbox(K, cx, cy, cz, b, synth.IMG_W, synth.IMG_H)
    valid = np.zeros(len(b), np.uint8)
    out = np.zeros((len(b), len(POSE_FIELDS)), np.float64)
    kept = np.zeros(len(b), np.int32)
    for i in range(len(b)):
        sel = ids == i
        kp = ol.radius_outlier(cx[sel], cy[sel], cz[sel], 0.4, 10).astype(bool)
        ok, e = ol.pca_bbox(cx[sel][kp], cy[sel][kp], cz[sel][kp])
        valid[i] = ok
        kept[i] = int(kp.sum())
        if ok:
            out[i] = [e[f] for f in POSE_FIELDS]
    return ids, valid, out, kept


def compute():
    tfs, x, y, z, (cx, cy, cz), b = scene()
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    out = {}
    u, v, d = ol.project_points(K, cx, cy, cz)
    for k in (4, 10):
        depths, d2 = ol.depth_for_bboxes(u, v, d, b, k)
        out[f"knn_depth_{k}"] = depths
        out[f"knn_d2_{k}"] = d2
    m, mask, coeff = ol.segment_ground_plane(cx, cy, cz)
    out["ground_n"] = np.array([m], np.int64)
    out["ground_mask_bits"] = np.packbits(mask)
    out["ground_coeff"] = coeff
    ids, valid, poses, kept = poses_of(cx, cy, cz, K, b)
    out["bbox_id"] = ids.astype(np.int8)
    out["pose_valid"] = valid
    out["poses"] = poses
    out["kept"] = kept
    g = mask == 0
    _, valid2, poses2, kept2 = poses_of(cx[g], cy[g], cz[g], K, b)
    out["pose_valid_ground_removed"] = valid2
    out["poses_ground_removed"] = poses2
    out["kept_ground_removed"] = kept2
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "pca_small.npz"), **compute())
    print("wrote", os.path.join(HERE, "pca_small.npz"))
