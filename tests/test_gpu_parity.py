"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Integer/index results are bit-exact; log-odds and
occupancy within 1e-5 (north_star), in practice bit-equal.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from gvamd import synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LOG_ODDS_TOL = 1e-5   # north_star: "log-odds within 1e-5"
OCC_TOL = 1e-5


@pytest.fixture(scope="module")
def gvamd():
    import gvamd as m
    m.load()
    return m


def make_handle(gvamd, config, perturbed=False):
    cfg = synth.CONFIGS[config]
    g = cfg["grid"]
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    tfs = synth.transforms(perturbed)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    return h, tfs


def oracle_frame(og, tfs, x, y, z, bboxes=None, poses=None, raymarch=True):
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    hits, cell = og.bin_points(m_base, x, y, z)
    miss, visits = og.raymarch(m_base, x, y, z) if raymarch else (None, 0)
    ids = None
    if bboxes is not None:
        cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
        K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
        ids = ol.extract_cloud_per_bbox(K, cx, cy, cz, bboxes, synth.IMG_W, synth.IMG_H)
    og.frame_update(poses, hits, miss)
    return hits, cell, miss, ids, visits


def check_grid(h, og):
    lo, occ = h.log_odds(), h.occupancy()
    assert np.max(np.abs(lo - og.log_odds)) <= LOG_ODDS_TOL
    assert np.max(np.abs(occ - og.occupancy)) <= OCC_TOL
    data, info = h.to_occupancy_grid()
    odata, oinfo = og.to_occupancy_grid()
    # int8 = trunc(p*100): may differ by 1 only where p*100 sits on an integer boundary
    diff = np.abs(data.astype(np.int16) - odata.astype(np.int16))
    assert diff.max() <= 1
    assert np.count_nonzero(diff) <= 1e-4 * data.size
    assert [info.width, info.height, info.resolution, info.origin_x, info.origin_y] == oinfo.tolist()
    return int(np.count_nonzero(lo != og.log_odds)), int(np.count_nonzero(occ != og.occupancy)), int(np.count_nonzero(diff))


@pytest.mark.parametrize("config,perturbed", [(1, False), (1, True), (2, False), (2, True)])
def test_frame_uniform_cloud(gvamd, config, perturbed):
    """configs[0]/[1]: binning + DDA ray-march, cell indices and counts bit-exact."""
    h, tfs = make_handle(gvamd, config, perturbed)
    cfg = synth.CONFIGS[config]
    g = cfg["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    x, y, z, _ = synth.cloud_uniform(config)
    h.upload_xyz(x, y, z)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_CELL_IDX | gvamd.FRAME_KEEP_COUNTS
    for frame in range(3):
        h.process_frame(flags)
        hits, cell, miss, _, visits = oracle_frame(og, tfs, x, y, z)
        assert np.array_equal(h.cell_idx(), cell)
        assert np.array_equal(h.hits(), hits)
        assert np.array_equal(h.miss(), miss.astype(np.int32))
        assert hits.sum() == np.count_nonzero(cell >= 0)
        nlo, nocc, ni8 = check_grid(h, og)
        assert nlo == 0, "log-odds are adds + clamp: expected bit-equal"
    h.close()


def test_frame_with_detections_and_poses(gvamd):
    """bbox first-match ids bit-exact; rectangle block adds; lidar-like contention."""
    config = 2
    h, tfs = make_handle(gvamd, config, perturbed=True)
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    bboxes = synth.detections(3, 50)
    poses = synth.lshape_poses(config, 50)
    flags = (gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_CELL_IDX
             | gvamd.FRAME_KEEP_COUNTS)
    for frame, gen in enumerate([synth.cloud_lidar_like, synth.cloud_uniform, synth.cloud_lidar_like]):
        x, y, z, _ = gen(config, 60_000, seed_extra=frame)
        h.upload_xyz(x, y, z)
        h.process_frame(flags, bboxes=bboxes, poses=poses)
        hits, cell, miss, ids, _ = oracle_frame(og, tfs, x, y, z, bboxes, poses)
        assert np.array_equal(h.cell_idx(), cell)
        assert np.array_equal(h.hits(), hits)
        assert np.array_equal(h.miss(), miss.astype(np.int32))
        assert np.array_equal(h.bbox_id(), ids)
        assert (ids >= 0).sum() > 100, "fixture must put points inside bboxes"
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
    h.close()


def test_counts_not_kept_matches_kept(gvamd):
    """the production path (counts cleared inside the finalise pass) gives the same grid."""
    config = 1
    x, y, z, _ = synth.cloud_uniform(config)
    outs = []
    for keep in (True, False):
        h, tfs = make_handle(gvamd, config)
        h.upload_xyz(x, y, z)
        flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | (gvamd.FRAME_KEEP_COUNTS if keep else 0)
        h.set_detections(flags)
        for _ in range(4):
            h.enqueue_frame()
        h.synchronize()
        outs.append((h.log_odds(), h.occupancy(), h.to_occupancy_grid()[0]))
        h.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_transform_lidar_to_camera_bit_exact(gvamd):
    h, tfs = make_handle(gvamd, 1, perturbed=True)
    x, y, z, _ = synth.cloud_uniform(1)
    h.upload_xyz(x, y, z)
    gx, gy, gz = h.transform_lidar_to_camera()
    ox, oy, oz = ol.transform_cloud(ol.tf_to_matrix4f(tfs["cam_lidar"]), x, y, z)
    assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gz, oz)
    h.close()


def test_extract_cloud_per_bbox_api(gvamd):
    h, tfs = make_handle(gvamd, 2, perturbed=False)
    x, y, z, _ = synth.cloud_lidar_like(2, 50_000)
    bboxes = synth.detections(3, 50)
    h.upload_xyz(x, y, z)
    ids, counts = h.extract_cloud_per_bbox(bboxes)
    cx, cy, cz = ol.transform_cloud(ol.tf_to_matrix4f(tfs["cam_lidar"]), x, y, z)
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    exp = ol.extract_cloud_per_bbox(K, cx, cy, cz, bboxes, synth.IMG_W, synth.IMG_H)
    assert np.array_equal(ids, exp)
    assert np.array_equal(counts, np.bincount(exp[exp >= 0], minlength=len(bboxes)))
    h.close()


def test_known_answers_through_abi(gvamd):
    """SURVEY 8(c) sequences through the C ABI (default yaml grid)."""
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        ka = json.load(f)
    h = gvamd.GridVisionHIP(50, 20, 0.1)
    assert (h.nx, h.ny, h.pos_x) == (500, 200, 16.0)
    assert np.all(h.log_odds() == 0.0) and np.all(h.occupancy() == 0.5)
    seq = []
    for k in range(12):
        h.update_map()
        lo = h.log_odds()
        assert np.all(lo == lo[0])
        seq.append(lo[0])
    assert np.array(seq, np.float32).tolist() == np.array(ka["empty_frame_log_odds"], np.float32).tolist()
    assert np.all(h.to_occupancy_grid()[0] == ka["empty_frame_saturated"]["int8"])
    assert abs(h.occupancy()[0] - ka["empty_frame_saturated"]["occupancy"]) < 1e-7
    h.reset()
    pose = np.zeros(1, dtype=synth.LSHAPE_DTYPE)
    pose["px"], pose["py"], pose["length"], pose["width"], pose["qw"] = 16.0, 0.0, 2.0, 1.0, 1.0
    k = 100 * 500 + 250
    seq = []
    for _ in range(7):
        h.update_map_poses(pose)
        seq.append(h.log_odds()[k])
    assert np.array(seq, np.float32).tolist() == np.array(ka["object_every_frame_log_odds"], np.float32).tolist()
    data, _ = h.to_occupancy_grid()
    assert data[h.G - 1 - k] == ka["object_saturated"]["int8"]
    h.close()


def test_update_map_poses_and_points_vs_oracle(gvamd):
    g = synth.CONFIGS[2]["grid"]
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    for frame in range(8):
        poses = synth.lshape_poses(2, 37, seed_extra=frame)
        h.update_map_poses(poses)
        og.update_map_poses(poses)
    nlo, nocc, ni8 = check_grid(h, og)
    assert nlo == 0
    # dead-code overload (occupancy_grid.cpp:33-63): centre points + class depth
    pts = np.stack([poses["px"], poses["py"], poses["pz"]], axis=1)
    bboxes = synth.detections(3, 37)
    h.update_map_points(pts, bboxes)
    og.update_map_points(pts, bboxes)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.close()


def test_edge_cases(gvamd):
    h, tfs = make_handle(gvamd, 1)
    g = synth.CONFIGS[1]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_CELL_IDX | gvamd.FRAME_KEEP_COUNTS
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    clouds = [
        # non-finite points, map edges (+edge inside, -edge outside), far-away points
        (np.array([nan, 1, inf, 83.0, -17.0, 0, 0, 1e30, -1e30, 5, 33.0], np.float32),
         np.array([0, nan, 0, 0, 0, 50.0, -50.0, 1e30, 3, -inf, 0.0], np.float32),
         np.array([0, 0, 0, 0, 0, 0, 0, 0, nan, 0, 0], np.float32)),
        # all points outside the map
        (np.full(100, 500.0, np.float32), np.linspace(-300, 300, 100).astype(np.float32), np.zeros(100, np.float32)),
        # every point in the sensor's own cell (zero-length rays)
        (np.zeros(64, np.float32), np.zeros(64, np.float32), np.zeros(64, np.float32)),
        # empty cloud
        (np.zeros(0, np.float32),) * 3,
    ]
    for x, y, z in clouds:
        h.upload_xyz(x, y, z)
        h.process_frame(flags)
        hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
        if len(x):
            assert np.array_equal(h.cell_idx(), cell)
        assert np.array_equal(h.hits(), hits)
        assert np.array_equal(h.miss(), miss.astype(np.int32))
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
    h.close()


def test_origin_outside_map_casts_no_rays(gvamd):
    g = synth.CONFIGS[1]["grid"]
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    tfs = synth.transforms()
    tfs["base_lidar"] = np.array([0, 0, 0, 1, -500.0, 0.0, 1.8])
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    x, y, z, _ = synth.cloud_uniform(1)
    x = x + np.float32(500.0)
    h.upload_xyz(x, y, z)
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS)
    hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
    assert miss.sum() == 0 and hits.sum() > 0
    assert np.array_equal(h.hits(), hits)
    assert h.miss().sum() == 0
    check_grid(h, og)
    h.close()


def test_pointcloud2_ingest(gvamd):
    """sensor_msgs/PointCloud2 bytes (PointXYZI, point_step 16 and a padded 32) -> same result as SoA."""
    h, tfs = make_handle(gvamd, 1)
    x, y, z, inten = synth.cloud_uniform(1)
    n = len(x)
    h.upload_xyz(x, y, z)
    ref = h.transform_lidar_to_camera()
    for step, offs in [(16, (0, 4, 8)), (32, (0, 4, 8)), (19, (1, 5, 9))]:
        raw = np.zeros((n, step), np.uint8)
        for arr, o in zip((x, y, z), offs):
            raw[:, o:o + 4] = arr.view(np.uint8).reshape(n, 4)
        h.upload_pointcloud2(raw.reshape(-1), n, step, *offs)
        got = h.transform_lidar_to_camera()
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
    h.close()


def test_vision_post_process_tolerance(gvamd):
    """A13/A14: 64-combination least squares per bbox, one wavefront each.  SURVEY 8(a) A14: location
    within 1e-4 (absolute + relative).  The device evaluates the float trig in fp64 rounded once and runs
    the oracle's QR operation order, so normally every box agrees to the last bits.  A box may pick another
    of the 64 constraint sets only if that was a proven near-tie: the oracle's own residual of the set the
    device chose must be within 1e-6 (relative) of the oracle's minimum."""
    h, tfs = make_handle(gvamd, 2)
    nb = 50
    bboxes = synth.detections(3, nb)
    orient, conf, dims = synth.network_outputs(nb)
    cam = ol.make_cam()
    exp = ol.post_process(cam, orient, conf, dims, bboxes)
    got = h.vision_post_process(orient, conf, dims, bboxes)
    assert len(got) == len(exp) > 0
    # oracle-side inputs of calcLocation per emitted pose (unknown classes are skipped, :496-499)
    bins = ol.generate_bins(2)
    kept = [i for i in range(nb) if int(bboxes[i]["label"]) in (9, 0, 1, 2)]
    assert len(kept) == len(exp)
    n_exact, n_ties, worst = 0, 0, 0.0
    for (gpose, epose), i in zip(zip(got, exp), kept):
        for k in ("length", "width", "height"):
            assert gpose[k] == epose[k]
        gl = np.array([gpose["px"], gpose["py"], gpose["pz"]])
        el = np.array([epose["px"], epose["py"], epose["pz"]])
        gq = np.array([gpose[q] for q in ("qx", "qy", "qz", "qw")])
        eq = np.array([epose[q] for q in ("qx", "qy", "qz", "qw")])
        assert np.allclose(gq, eq, atol=1e-6)
        if np.allclose(gl, el, rtol=1e-4, atol=1e-4):
            n_exact += 1
            worst = max(worst, float(np.max(np.abs(gl - el))))
            continue
        argmax = 1 if conf[i][1] > conf[i][0] else 0
        alpha = ol.compute_alpha(orient[i], argmax, bins)
        theta = ol.compute_theta_ray(cam, bboxes[i])
        loc64, err64 = ol.calc_location_all(cam, [epose["length"], epose["width"], epose["height"]], bboxes[i], alpha, theta)
        k = int(np.argmin(np.max(np.abs(loc64 - gl.astype(np.float32)), axis=1)))
        assert np.allclose(loc64[k], gl, rtol=1e-4, atol=1e-4), f"box {i}: device location is none of the 64 candidates"
        best = float(err64.min())
        assert float(err64[k]) - best <= 1e-6 * max(best, 1e-30), f"box {i}: set {k} residual {err64[k]} vs best {best} is no tie"
        n_ties += 1
    assert n_exact + n_ties == len(exp)
    print(f"A14: {n_exact}/{len(exp)} within 1e-4 (max abs diff {worst:.3g}), {n_ties} proven ties")
    # base-frame transform of the poses (A15) is host fp64: exact vs oracle
    tb = h.transform_lshape_objects(exp)
    for i, e in enumerate(exp):
        pose7 = [e[k] for k in ("px", "py", "pz", "qx", "qy", "qz", "qw")]
        o = ol.tf_pose(tfs["base_cam"], pose7)
        assert [tb[i][k] for k in ("px", "py", "pz", "qx", "qy", "qz", "qw")] == o.tolist()
    h.close()


def test_frame_vision_orient_path(gvamd):
    """use_vision_orientation=true path fused on device: net outputs -> poses -> rectangles."""
    config = 2
    h, tfs = make_handle(gvamd, config)
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    nb = 20
    bboxes = synth.detections(3, nb)
    orient, conf, dims = synth.network_outputs(nb)
    got = h.vision_post_process(orient, conf, dims, bboxes)      # device poses (camera frame)
    base = h.transform_lshape_objects(got)
    x, y, z, _ = synth.cloud_uniform(config, 20_000)
    h.upload_xyz(x, y, z)
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_VISION_ORIENT | gvamd.FRAME_KEEP_COUNTS,
                    bboxes=bboxes, net=(orient, conf, dims))
    # oracle grid update fed with the device's own poses: checks the fused plumbing
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    hits, _ = og.bin_points(m_base, x, y, z)
    miss, _ = og.raymarch(m_base, x, y, z)
    og.frame_update(base, hits, miss)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.close()


def test_object_detection_host_side(gvamd):
    st = synth.Stream(1234, 9)
    n, c = 300, 10
    ctr = st.uniform(2 * n, 0.2, 0.8).reshape(n, 2)
    wh = st.uniform(2 * n, 0.02, 0.3).reshape(n, 2)
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
    scores = st.uniform(n * c, 0.0, 1.0).reshape(n, c).astype(np.float32)
    got = gvamd.extract_bboxes(boxes, scores, 0.6, 0.6, 640, 480, 416)
    exp = ol.extract_bboxes(boxes, scores, 0.6, 0.6, 640, 480, 416)
    assert len(got) == len(exp) > 5
    assert got.tobytes() == exp.tobytes()
    gs, gd = gvamd.filter_bboxes(got)
    es, ed = ol.filter_bboxes(exp)
    assert gs.tobytes() == es.tobytes() and gd.tobytes() == ed.tobytes()
    h = gvamd.GridVisionHIP(50, 20, 0.1)
    k, ki = h.intrinsics()
    assert k.tolist() == ol.set_intrinsic(320.0, 320.0, 320.0, 240.0).tolist()
    assert ki.tolist() == ol.k_inverse(k).tolist()
    depths = np.linspace(3, 40, len(gs)).astype(np.float32)
    tfs = synth.transforms()
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    pts = h.convert_pixels_to_3d(gs, depths)
    for i, b in enumerate(gs):
        pcx = np.float32(b["x_min"] + ((b["x_max"] - b["x_min"]) / np.float32(2.0)))
        pcy = np.float32(b["y_min"] + ((b["y_max"] - b["y_min"]) / np.float32(2.0)))
        cam = ol.pixel_to_3d(pcx, pcy, depths[i], ki)
        assert pts[i].tolist() == ol.tf_point(tfs["base_cam"], cam).tolist()
    h.close()


@pytest.mark.parametrize("grid", [(50, 20, 0.3), (50, 20, 0.1), (120, 200, 0.25), (255, 255, 0.5)])
def test_odd_grid_shapes(gvamd, grid):
    """non-square grids, nx % 4 != 0 (generic kernels) and nx % 64 != 0 (partial tiles)."""
    gx, gy, res = grid
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    assert (h.nx, h.ny) == (og.nx, og.ny)
    tfs = synth.transforms(True)
    tfs["base_lidar"] = np.array([0.0, 0.0, 0.0, 1.0, gx / 3.0 + 1.7, -gy * 0.21, 1.8])
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    st = synth.Stream(99, gx)
    n = 40_000
    lx, ly = og.g.len_x, og.g.len_y
    x = st.uniform(n, -0.8 * lx, 0.8 * lx)
    y = st.uniform(n, -0.8 * ly, 0.8 * ly)
    z = st.uniform(n, -1.0, 1.0)
    poses = np.zeros(20, dtype=synth.LSHAPE_DTYPE)
    poses["px"] = st.uniform(20, og.g.pos_x - lx / 2, og.g.pos_x + lx / 2)
    poses["py"] = st.uniform(20, -ly / 2, ly / 2)
    poses["length"] = st.uniform(20, 0.5, 5.0)
    poses["width"] = st.uniform(20, 0.5, 2.5)
    h.upload_xyz(x, y, z)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_CELL_IDX | gvamd.FRAME_KEEP_COUNTS
    for frame in range(2):
        h.process_frame(flags, poses=poses)
        hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z, None, poses)
        assert np.array_equal(h.cell_idx(), cell)
        assert np.array_equal(h.hits(), hits)
        assert np.array_equal(h.miss(), miss.astype(np.int32))
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
    h.close()


def test_config3_full_size_properties(gvamd):
    """BASELINE configs[2] at full size (1M points, 2000x2000): size-independent
    properties instead of the oracle: count conservation, hits <-> cell_idx
    consistency, a hit cell is never counted free, every free cell lies on the
    Bresenham line of at least... (checked on a sample against the oracle)."""
    config = 3
    h, tfs = make_handle(gvamd, config, perturbed=True)
    g = synth.CONFIGS[config]["grid"]
    x, y, z, _ = synth.cloud_uniform(config)
    bboxes, poses = synth.detections(config), synth.lshape_poses(config)
    h.upload_xyz(x, y, z)
    flags = (gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_CELL_IDX
             | gvamd.FRAME_KEEP_COUNTS)
    h.process_frame(flags, bboxes=bboxes, poses=poses)
    cell, hits, miss, ids = h.cell_idx(), h.hits(), h.miss(), h.bbox_id()
    inmap = cell >= 0
    assert hits.sum() == inmap.sum()
    assert np.array_equal(np.bincount(cell[inmap], minlength=h.G).astype(np.int32), hits)
    assert set(np.unique(miss)) <= {0, 1}
    assert ids.min() >= -1 and ids.max() < len(bboxes)
    # the oracle on the same full-size inputs (a few seconds with dedupe)
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    ohits, ocell, omiss, oids, _ = oracle_frame(og, tfs, x, y, z, bboxes, poses)
    assert np.array_equal(cell, ocell)
    assert np.array_equal(hits, ohits)
    assert np.array_equal(miss, omiss.astype(np.int32))
    assert np.array_equal(ids, oids)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.close()


def test_simple_ray_impl_matches(gvamd, monkeypatch):
    """GV_RAY_IMPL=simple (literal per-ray march kernels) and the sector/gather
    kernels produce the same miss grid."""
    config = 2
    x, y, z, _ = synth.cloud_lidar_like(config, 80_000)
    outs = []
    for impl in ("simple", "sectors"):
        monkeypatch.setenv("GV_RAY_IMPL", impl)
        h, tfs = make_handle(gvamd, config, perturbed=True)
        h.upload_xyz(x, y, z)
        h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS)
        outs.append((h.hits(), h.miss(), h.log_odds()))
        h.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_sector_chunk_overflow_path(gvamd, monkeypatch):
    """few, fat sectors (GV_LOG2S=5) with a dense cloud: more ends per wedge than one
    LDS chunk holds, so the chunked scan/flush path of the sector kernel runs."""
    monkeypatch.setenv("GV_LOG2S", "5")
    monkeypatch.setenv("GV_CAP", "2048")
    config = 2
    h, tfs = make_handle(gvamd, config, perturbed=True)
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    x, y, z, _ = synth.cloud_uniform(config, 700_000)
    h.upload_xyz(x, y, z)
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS)
    hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
    assert np.array_equal(h.hits(), hits)
    assert np.array_equal(h.miss(), miss.astype(np.int32))
    n_rays, n_visits = h.ray_stats()
    assert n_rays > 32 * 8 * 2048 // 2   # some wedges must have overflowed one chunk
    h.close()


def test_bbox_test_many_fractional_boxes(gvamd):
    """150 overlapping boxes (3 mask words) with fractional fp64 bounds, some exactly on
    float values of projected pixels: the float-threshold form must keep the fp64 truth table."""
    h, tfs = make_handle(gvamd, 2, perturbed=True)
    x, y, z, _ = synth.cloud_lidar_like(2, 120_000)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    u, v, _ = ol.project_points(K, cx, cy, cz)
    st = synth.Stream(77, 3)
    nb = 150
    b = np.zeros(nb, dtype=synth.BBOX_DTYPE)
    x0 = st.uniform(nb, -20, 600).astype(np.float64) + 1e-7
    y0 = st.uniform(nb, -20, 440).astype(np.float64) - 1e-9
    b["x_min"], b["y_min"] = x0, y0
    b["x_max"] = x0 + st.uniform(nb, 1, 120).astype(np.float64)
    b["y_max"] = y0 + st.uniform(nb, 1, 120).astype(np.float64)
    # bounds sitting exactly on projected pixel coordinates (inclusive edges must match)
    pick = st.integers(40, 0, len(u))
    b["x_min"][:20] = u[pick[:20]].astype(np.float64)
    b["x_max"][20:40] = u[pick[20:40]].astype(np.float64)
    b["x_max"][:20] = b["x_min"][:20] + 50
    b["x_min"][20:40] = b["x_max"][20:40] - 50
    b[140]["x_min"] = np.nan          # never matches
    b[141]["x_max"] = b[141]["x_min"] - 1   # empty
    b["confidence"] = np.sort(st.uniform(nb, 0.5, 1.0))[::-1]
    b["label"] = st.integers(nb, 0, 10)
    h.upload_xyz(x, y, z)
    ids, counts = h.extract_cloud_per_bbox(b)
    exp = ol.extract_cloud_per_bbox(K, cx, cy, cz, b, synth.IMG_W, synth.IMG_H)
    assert np.array_equal(ids, exp)
    assert (exp >= 64).sum() > 0 and (exp >= 128).sum() > 0, "fixture must reach the 2nd and 3rd mask word"
    h.close()


def _cluster_scene(tfs, n_clusters=24, pts_per=400, seed=11):
    """clusters of points in front of the camera + a sparse background, and one integer
    pixel bbox around each cluster (what a detector would hand over)."""
    st = synth.Stream(seed, 21)
    cxs = st.uniform(n_clusters, 6.0, 45.0).astype(np.float64)
    cys = (st.uniform(n_clusters, -0.7, 0.7).astype(np.float64)) * cxs
    czs = st.uniform(n_clusters, -1.0, 0.5).astype(np.float64)
    xs, ys, zs = [], [], []
    for k in range(n_clusters):
        xs.append(cxs[k] + st.uniform(pts_per, -0.7, 0.7))
        ys.append(cys[k] + st.uniform(pts_per, -0.5, 0.5))
        zs.append(czs[k] + st.uniform(pts_per, -0.4, 0.4))
    nbg = 30_000
    xs.append(st.uniform(nbg, -20, 60)); ys.append(st.uniform(nbg, -40, 40)); zs.append(st.uniform(nbg, -2, 3))
    x = np.concatenate(xs).astype(np.float32); y = np.concatenate(ys).astype(np.float32); z = np.concatenate(zs).astype(np.float32)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    boxes = []
    for k in range(n_clusters):
        sl = slice(k * pts_per, (k + 1) * pts_per)
        zc = cz[sl]
        ok = zc > 0.1
        if ok.sum() < 10:
            continue
        u = synth.FX * cx[sl][ok] / zc[ok] + synth.CX
        v = synth.FY * cy[sl][ok] / zc[ok] + synth.CY
        x0, x1 = int(max(0, np.floor(u.min()) - 2)), int(min(639, np.ceil(u.max()) + 2))
        y0, y1 = int(max(0, np.floor(v.min()) - 2)), int(min(479, np.ceil(v.max()) + 2))
        if x1 > x0 and y1 > y0:
            boxes.append((x0, y0, x1, y1))
    b = np.zeros(len(boxes), dtype=synth.BBOX_DTYPE)
    for i, (x0, y0, x1, y1) in enumerate(boxes):
        b[i] = (x0, y0, x1, y1, 0.99 - 0.01 * i, [9, 2, 0, 1, 5][i % 5])
    return x, y, z, (cx, cy, cz), K, b


@pytest.mark.parametrize("k", [1, 4, 10, 32])
def test_compute_depth_for_bboxes_knn(gvamd, k):
    """buildKDTree + computeDepthForBoundingBoxes: exact k nearest in (u, v, depth),
    distances bit-equal, ties by lower index, upper-median depth equal."""
    h, tfs = make_handle(gvamd, 2, perturbed=True)
    x, y, z, (cx, cy, cz), K, b = _cluster_scene(tfs)
    h.upload_xyz(x, y, z)
    depths, d2 = h.compute_depth_for_bboxes(b, k)
    u, v, d = ol.project_points(K, cx, cy, cz)
    edepths, ed2 = ol.depth_for_bboxes(u, v, d, b, k)
    assert len(b) >= 15
    assert np.array_equal(d2, ed2)
    assert np.array_equal(depths, edepths)
    assert (depths > 0).all()
    h.close()


def test_compute_depth_empty_and_behind(gvamd):
    h, tfs = make_handle(gvamd, 1)
    b = synth.detections(3, 5)
    # every point behind the camera -> depth stays -1 (cloud_detections.cpp:49)
    x = np.full(100, -5.0, np.float32)
    h.upload_xyz(x, np.zeros(100, np.float32), np.zeros(100, np.float32))
    depths, d2 = h.compute_depth_for_bboxes(b, 4)
    assert (depths == -1.0).all() and np.isinf(d2).all()
    # fewer points than k: the reference's kNN returns what exists
    h.upload_xyz(np.array([10.0, 12.0], np.float32), np.zeros(2, np.float32), np.zeros(2, np.float32))
    depths, d2 = h.compute_depth_for_bboxes(b, 4)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, [10.0, 12.0], [0, 0], [0, 0])
    u, v, d = ol.project_points(ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY), cx, cy, cz)
    ed, ed2 = ol.depth_for_bboxes(u, v, d, b, 4)
    assert np.array_equal(depths, ed) and np.array_equal(d2, ed2)
    h.close()


def test_compute_bbox_pose_pca_path(gvamd):
    """use_vision_orientation=false path: extractCloudPerBBox + RadiusOutlierRemoval(0.4, 10)
    + centroid + PCA rectangle per bbox, all on the device (cell-hash neighbour counts, sixteen lanes per query;
    the rectangle's sums as order-independent integers).  The RANSAC ground removal is not part of this call.
    Against the oracle within SURVEY A11's 1e-4, against fp64 numpy much tighter (_check_pose_fp64); and the call
    is bit-reproducible (integer sums: no dependence on the order workgroups arrive in)."""
    h, tfs = make_handle(gvamd, 2, perturbed=True)
    x, y, z, (cx, cy, cz), K, b = _cluster_scene(tfs, n_clusters=20, pts_per=500, seed=5)
    h.upload_xyz(x, y, z)
    poses, valid = h.compute_bbox_pose(b)
    ids = ol.extract_cloud_per_bbox(K, cx, cy, cz, b, synth.IMG_W, synth.IMG_H)
    n_valid = 0
    for i in range(len(b)):
        sel = ids == i
        keep = ol.radius_outlier(cx[sel], cy[sel], cz[sel], 0.4, 10).astype(bool)
        ok, e = ol.pca_bbox(cx[sel][keep], cy[sel][keep], cz[sel][keep])
        assert bool(valid[i]) == ok
        if ok:
            n_valid += 1
            _check_pose(poses[i], e, i)
            _check_pose_fp64(poses[i], cx[sel][keep], cy[sel][keep], cz[sel][keep], i)
    assert n_valid >= 10
    for _ in range(3):   # bit-reproducible, call after call
        p2, v2 = h.compute_bbox_pose(b)
        assert p2.tobytes() == poses.tobytes() and v2.tobytes() == valid.tobytes()
    # and through the grid: transformLShapeObjects + updateMap(poses)
    g = synth.CONFIGS[2]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    base = h.transform_lshape_objects(poses[valid.astype(bool)])
    h.update_map_poses(base)
    og.update_map_poses(base)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.close()


@pytest.mark.parametrize("n_clusters,pts_per", [(90, 150), (170, 120)])
def test_compute_bbox_pose_many_boxes(gvamd, n_clusters, pts_per):
    """Up to 128 boxes a workgroup keeps the boxes' integer sums / extents in an LDS table and flushes it once;
    beyond that the atomics go to global memory directly: 90 and 170 clusters / boxes take the two forms.  Same
    comparison as test_compute_bbox_pose_pca_path."""
    h, tfs = make_handle(gvamd, 2, perturbed=True)
    x, y, z, (cx, cy, cz), K, b = _cluster_scene(tfs, n_clusters=n_clusters, pts_per=pts_per, seed=23)
    assert len(b) > (128 if n_clusters > 128 else 64)
    h.upload_xyz(x, y, z)
    poses, valid = h.compute_bbox_pose(b)
    ids = ol.extract_cloud_per_bbox(K, cx, cy, cz, b, synth.IMG_W, synth.IMG_H)
    n_valid = 0
    for i in range(len(b)):
        sel = ids == i
        keep = ol.radius_outlier(cx[sel], cy[sel], cz[sel], 0.4, 10).astype(bool)
        ok, e = ol.pca_bbox(cx[sel][keep], cy[sel][keep], cz[sel][keep])
        assert bool(valid[i]) == ok
        if ok:
            n_valid += 1
            _check_pose(poses[i], e, i)
            _check_pose_fp64(poses[i], cx[sel][keep], cy[sel][keep], cz[sel][keep], i)
    assert n_valid >= 30
    h.close()


def test_sharded_frame_world1_matches_plain(gvamd):
    """RCCL path with a 1-rank communicator (all this box has): send/recv group, slice OR, all-gather,
    band packing, band grid pass and band broadcast must reproduce the plain frame exactly.
    rank > 0 / world > 1 run on one device in test_sharded_frame_every_rank_emulated."""
    config = 2
    x, y, z, _ = synth.cloud_uniform(config)
    poses = synth.lshape_poses(config, 30)
    bboxes = synth.detections(3, 30)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    ha, tfs = make_handle(gvamd, config, perturbed=True)
    hb, _ = make_handle(gvamd, config, perturbed=True)
    hb.comm_init(gvamd.GridVisionHIP.comm_unique_id(), 0, 1)
    assert hb.comm_band() == (0, hb.G)
    assert hb.comm_info()[:2] == (1, 0)        # what RCCL says: one rank, this is rank 0
    for h in (ha, hb):
        h.upload_xyz(x, y, z)
    for frame in range(3):
        ha.process_frame(flags, bboxes=bboxes, poses=poses)
        hb.process_frame_sharded(flags, bboxes=bboxes, poses=poses)
        assert np.array_equal(ha.log_odds(), hb.log_odds())
        assert np.array_equal(ha.occupancy(), hb.occupancy())
        assert np.array_equal(ha.to_occupancy_grid()[0], hb.to_occupancy_grid()[0])
        assert np.array_equal(ha.bbox_id(), hb.bbox_id())
    # asynchronous form: frames in flight on the lanes / the exchange stream / the public stream, detections changing
    # in between, counts reduced by band (one band = the whole grid here)
    kf = flags | gvamd.FRAME_KEEP_COUNTS
    for frame in range(7):
        bb, pp = synth.detections(3, 10 + 3 * frame, seed_extra=frame), synth.lshape_poses(config, 5 + 2 * frame, seed_extra=frame)
        ha.set_detections(kf, bboxes=bb, poses=pp)
        hb.set_detections_async(kf, bboxes=bb, poses=pp)
        ha.enqueue_frame()
        hb.enqueue_frame_sharded()
        if frame % 3 == 2:
            hb.enqueue_frame_sharded()
            ha.enqueue_frame()
    ha.synchronize(); hb.synchronize()
    assert np.array_equal(ha.log_odds(), hb.log_odds())
    assert np.array_equal(ha.to_occupancy_grid()[0], hb.to_occupancy_grid()[0])
    assert np.array_equal(ha.hits(), hb.hits())
    assert np.array_equal(ha.bbox_id(), hb.bbox_id())
    st = hb.time_frame_sharded_stages(3)
    assert set(st) == {"bin", "exchange_ends", "sectors", "exchange_free", "grid_pass", "gather"} and all(v >= 0 for v in st.values())
    assert st["bin"] > 0 and st["grid_pass"] > 0
    ha.enqueue_frame(); ha.enqueue_frame(); ha.enqueue_frame(); ha.synchronize()   # the three timed frames
    assert np.array_equal(ha.log_odds(), hb.log_odds())
    hb.comm_destroy()
    ha.close(); hb.close()


def test_sharded_keep_counts_frames_in_flight(gvamd):
    """Round-3 advisor finding: step x3 of a sharded KEEP_COUNTS frame reduces the lane's hits[] in place on the
    exchange stream, and the lane's next tile pass (two frames later) rewrites every cell of it.  Six asynchronous
    KEEP_COUNTS frames over DIFFERENT clouds, nothing synchronised in between: the counts read afterwards are the
    last frame's, bit-exact against the oracle, and the grid is the six-frame sequence's."""
    config = 2
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    h, tfs = make_handle(gvamd, config, perturbed=True)
    h.comm_init(gvamd.GridVisionHIP.comm_unique_id(), 0, 1)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS
    h.set_detections(flags)
    clouds = []
    for f in range(6):
        gen = synth.cloud_lidar_like if f % 2 else synth.cloud_uniform
        x, y, z, _ = gen(config, 200_000 + 10_000 * f, seed_extra=40 + f)
        pin = [gvamd.PinnedF32(len(x)) for _ in range(3)]
        for p, a in zip(pin, (x, y, z)):
            p.array[:] = a
        clouds.append((x, y, z, pin))
    hits = None
    for x, y, z, pin in clouds:
        h.upload_xyz_async(pin[0].array, pin[1].array, pin[2].array)
        h.enqueue_frame_sharded()
        hits, _, _, _, _ = oracle_frame(og, tfs, x, y, z)
    h.synchronize()
    assert np.array_equal(h.hits(), hits)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.comm_destroy()
    h.close()
    for _, _, _, pin in clouds:
        for p in pin:
            p.close()


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("grid,n", [((200, 200, 0.2), 90_000), ((120, 200, 0.25), 50_001)])
def test_sharded_frame_every_rank_emulated(gvamd, world, grid, n):
    """SURVEY 8(e)-2 product code for rank > 0 / world > 1 on ONE device: every rank's binning of its
    point slice, the OR of the end-bitmap slices, every world-th sector workgroup, band packing, band OR
    and band grid pass run with their real (rank, world) -- only the RCCL transfers are device copies.
    ny = 800 (ny_pad 896, 14 blocks of 64 rows) gives bands of unequal size and a clipped last block.
    The union of the bands must equal the plain frame, i.e. the oracle, frame after frame."""
    gx, gy, res = grid
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    tfs = synth.transforms(True)
    tfs["base_lidar"] = np.array([0.0, 0.0, 0.0, 1.0, gx / 3.0 - 2.3, gy * 0.11, 1.8])
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    st = synth.Stream(991, world * 100 + n)
    lx, ly = og.g.len_x, og.g.len_y
    poses = np.zeros(9, dtype=synth.LSHAPE_DTYPE)
    poses["px"] = st.uniform(9, og.g.pos_x - lx / 2, og.g.pos_x + lx / 2)
    poses["py"] = st.uniform(9, -ly / 2, ly / 2)
    poses["length"] = st.uniform(9, 0.5, 5.0)
    poses["width"] = st.uniform(9, 0.5, 2.5)
    bboxes = synth.detections(3, 10)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_CELL_IDX
    for frame in range(2):
        x = st.uniform(n, -0.7 * lx, 0.7 * lx)
        y = st.uniform(n, -0.7 * ly, 0.7 * ly)
        z = st.uniform(n, -1.0, 1.0)
        h.upload_xyz(x, y, z)
        h.frame_sharded_emulated(world, flags, bboxes=bboxes, poses=poses)
        _, cell, _, ids, _ = oracle_frame(og, tfs, x, y, z, bboxes, poses)
        assert np.array_equal(h.cell_idx(), cell)
        assert np.array_equal(h.bbox_id(), ids)
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
    h.close()


def test_gpu_matches_frozen_fixture(gvamd):
    """tests/golden/frame_small.npz (made by tests/golden/make_frame_fixture.py): the HIP path
    against committed expected outputs, independent of the oracle binary on this machine."""
    gold = np.load(os.path.join(HERE, "golden", "frame_small.npz"))
    config, N, NDET = 1, 4000, 12
    h, tfs = make_handle(gvamd, config, perturbed=True)
    bboxes, poses = synth.detections(3, NDET), synth.lshape_poses(config, NDET)
    flags = (gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_CELL_IDX
             | gvamd.FRAME_KEEP_COUNTS)
    for f in range(3):
        gen = synth.cloud_lidar_like if f == 1 else synth.cloud_uniform
        x, y, z, _ = gen(config, N, seed_extra=f)
        h.upload_xyz(x, y, z)
        h.process_frame(flags, bboxes=bboxes, poses=poses)
        hits = h.hits()
        assert np.array_equal(h.cell_idx(), gold[f"cell_{f}"])
        assert np.array_equal(np.flatnonzero(hits).astype(np.int32), gold[f"hits_nz_{f}"])
        assert np.array_equal(hits[hits != 0], gold[f"hits_val_{f}"])
        assert np.array_equal(np.packbits(h.miss().astype(np.uint8)), gold[f"miss_bits_{f}"])
        assert np.array_equal(h.bbox_id().astype(np.int8), gold[f"bbox_id_{f}"])
        assert np.max(np.abs(h.log_odds() - gold[f"log_odds_{f}"])) <= LOG_ODDS_TOL
        assert np.max(np.abs(h.to_occupancy_grid()[0].astype(int) - gold[f"i8_{f}"].astype(int))) <= 1
    h.close()


def test_gpu_matches_pca_fixture(gvamd):
    """tests/golden/pca_small.npz: kNN depths / distances and the ground mask bit for bit, the plane coefficients
    equal, the PCA poses of computeBBoxPose (with and without ground removal) within the tolerances of _check_pose"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_pca_fixture as mk
    gold = np.load(os.path.join(HERE, "golden", "pca_small.npz"))
    tfs, x, y, z, _, b = mk.scene()
    h = gvamd.GridVisionHIP(50, 20, 0.1)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.upload_xyz(x, y, z)
    for k in (4, 10):
        depths, d2 = h.compute_depth_for_bboxes(b, k)
        assert np.array_equal(depths, gold[f"knn_depth_{k}"])
        assert np.array_equal(d2, gold[f"knn_d2_{k}"])
    m, mask, coeff = h.segment_ground_plane()
    ccx, ccy, ccz = ol.transform_cloud(ol.tf_to_matrix4f(tfs["cam_lidar"]), x, y, z)
    _check_plane(m, mask, coeff, int(gold["ground_n"][0]), np.unpackbits(gold["ground_mask_bits"])[:len(x)], gold["ground_coeff"],
                 ccx, ccy, ccz)
    for call, vkey, pkey in ((lambda: h.compute_bbox_pose(b), "pose_valid", "poses"),
                             (lambda: h.compute_bbox_pose_ground_removed(b)[:2], "pose_valid_ground_removed", "poses_ground_removed")):
        poses, valid = call()
        assert np.array_equal(valid, gold[vkey])
        for i in range(len(b)):
            if valid[i]:
                _check_pose(poses[i], dict(zip(mk.POSE_FIELDS, gold[pkey][i])), (pkey, i))
    assert np.array_equal(h.bbox_id()[mask == 0].astype(np.int8), gold["bbox_id"][mask == 0])
    h.close()


def test_cpp_demo(gvamd, tmp_path):
    """grid-vision_amd/examples/frame_demo.cpp: the reference's timerCallback flow in plain
    g++ host code over the C ABI (no hipcc, no torch).  Its checksums must equal the same
    sequence driven through the ctypes binding."""
    import re
    import subprocess
    root = os.path.dirname(HERE)
    pkg = os.path.join(root, "grid-vision_amd")
    exe = str(tmp_path / "frame_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(pkg, "examples", "frame_demo.cpp"), "-o", exe,
                           "-L" + pkg, "-lgridvision_hip", "-Wl,-rpath," + pkg])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    txt = out.stdout
    sum_i8, sum_hits = (int(v) for v in re.search(r"sum_i8 (-?\d+) sum_hits (\d+)", txt).groups())
    depth0 = float(re.search(r"depth0 (\S+)", txt).group(1))
    # the same flow from python
    n = 50000
    with np.errstate(over="ignore"):
        z = synth._mix64(np.uint64(42) + np.arange(1, 3 * n + 1, dtype=np.uint64) * synth.GOLDEN)
    u = ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(n, 3)
    x = np.float32(-12.0) + np.float32(56.0) * u[:, 0]
    y = np.float32(-12.0) + np.float32(24.0) * u[:, 1]
    zz = np.float32(-2.0) + np.float32(4.0) * u[:, 2]
    h = gvamd.GridVisionHIP(50, 20, 0.1)
    h.set_transforms([0.5, -0.5, 0.5, 0.5, 0.0, 0.4, -0.3], [0.5, -0.5, 0.5, -0.5, 0.3, 0.0, 2.2], [0, 0, 0, 1, 0, 0, 1.8])
    h.upload_xyz(x, y, zz)
    b = np.array([(100, 150, 220, 300, 0.95, 9), (300, 200, 380, 330, 0.9, 2), (420, 100, 470, 160, 0.8, 5),
                  (500, 250, 600, 400, 0.7, 0)], dtype=synth.BBOX_DTYPE)
    st, dy = gvamd.filter_bboxes(b)
    depths, _ = h.compute_depth_for_bboxes(st, 4)
    assert abs(depths[0] - depth0) < 1e-5
    orient = np.tile(np.array([0.8, 0.6, -0.6, 0.8], np.float32), (len(dy), 1))
    conf = np.tile(np.array([0.3, 0.7], np.float32), (len(dy), 1))
    dims = np.full((len(dy), 3), 0.1, np.float32)
    poses = h.transform_lshape_objects(h.vision_post_process(orient, conf, dims, dy))
    h.update_map_poses(poses)
    pp, valid = h.compute_bbox_pose(b)
    h.update_map_poses(h.transform_lshape_objects(pp[valid.astype(bool)]))
    h.update_map()
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_COUNTS,
                    bboxes=b, poses=poses)
    assert int(h.hits().sum()) == sum_hits
    assert int(h.to_occupancy_grid()[0].astype(np.int64).sum()) == sum_i8
    h.close()


def test_cpp_flow_demo(gvamd, tmp_path):
    """grid-vision_amd/include/grid_vision/frame_flow.hpp: the decision flow of the reference's timerCallback
    (grid_vision_node.cpp:108-244) as a ROS-free class, driven through all six branches by examples/flow_demo.cpp
    (plain g++ over the C ABI).  Branch taken, box / pose counts, first static depth and the published grid's
    checksum after every tick must equal the same sequence replayed through the ctypes binding."""
    import re
    import subprocess
    root = os.path.dirname(HERE)
    pkg = os.path.join(root, "grid-vision_amd")
    exe = str(tmp_path / "flow_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(pkg, "examples", "flow_demo.cpp"), "-o", exe,
                           "-L" + pkg, "-lgridvision_hip", "-Wl,-rpath," + pkg])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # the fused tick (default: one batch, one host wait) and the call-by-call sequence print the same lines
    out0 = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=120)
    assert out0.returncode == 0, out0.stderr
    assert out0.stdout == out.stdout
    ticks = []
    for ln in out.stdout.splitlines():
        m = re.match(r"tick (\d+) branch (\S+) bboxes (\d+) static (\d+) dynamic (\d+) depths (\d+) poses (\d+) "
                     r"publish_detections (\d) sum_i8 (-?\d+) depth0 (\S+)", ln)
        assert m, ln
        ticks.append(m.groups())
    assert [t[1] for t in ticks] == ["missing_inputs", "no_detections", "no_transform", "vision_orientation", "cloud_pca",
                                     "no_dynamic_objects"]
    # the same cloud and ticks from python
    n = 60000
    with np.errstate(over="ignore"):
        zz = synth._mix64(np.uint64(7) + np.arange(1, 3 * n + 1, dtype=np.uint64) * synth.GOLDEN)
    u = ((zz >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(n, 3)
    a, b_, c = u[:, 0], u[:, 1], u[:, 2]
    f = np.float32
    x, y, z = np.empty(n, f), np.empty(n, f), np.empty(n, f)
    s0, s1, s2, s3 = slice(0, 30000), slice(30000, 40000), slice(40000, 50000), slice(50000, n)
    x[s0] = f(2) + f(38) * a[s0]; y[s0] = f(-9) + f(18) * b_[s0]; z[s0] = f(-1.7) + f(0.02) * (c[s0] - f(0.5))
    x[s1] = f(11) + f(2.4) * a[s1]; y[s1] = f(-2.6) + f(1.2) * b_[s1]; z[s1] = f(-1.2) + f(1.3) * c[s1]
    x[s2] = f(17) + f(1.0) * a[s2]; y[s2] = f(2.5) + f(3.0) * b_[s2]; z[s2] = f(-1.2) + f(1.3) * c[s2]
    x[s3] = f(-8) + f(48) * a[s3]; y[s3] = f(-9.5) + f(19) * b_[s3]; z[s3] = f(-1.5) + f(4) * c[s3]
    h = gvamd.GridVisionHIP(50, 20, 0.1)
    h.upload_xyz(x, y, z)
    full = np.array([(330, 200, 460, 330, 0.95, 9), (150, 180, 280, 330, 0.9, 2), (420, 100, 470, 160, 0.8, 5),
                     (40, 60, 100, 120, 0.7, 7)], dtype=synth.BBOX_DTYPE)
    only_static = full[2:]

    def grid_sum():
        return int(h.to_occupancy_grid()[0].astype(np.int64).sum())

    def expect(t, bboxes, nst, ndy, ndepth, nposes, pub, depth0=None):
        assert (int(t[2]), int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7])) == (bboxes, nst, ndy, ndepth, nposes, pub), t
        assert int(t[8]) == grid_sum(), t
        if depth0 is not None:
            assert abs(float(t[9]) - depth0) < 1e-5, t

    expect(ticks[0], 0, 0, 0, 0, 0, 0)                     # nothing happened: the initial grid
    h.update_map()
    expect(ticks[1], 0, 0, 0, 0, 0, 0)
    expect(ticks[2], 4, 2, 2, 0, 0, 0)                     # tf failure: stale grid
    h.set_transforms([0.5, -0.5, 0.5, 0.5, 0.0, 0.4, -0.3], [0.5, -0.5, 0.5, -0.5, 0.3, 0.0, 2.2], [0, 0, 0, 1, 0, 0, 1.8])
    st, dy = gvamd.filter_bboxes(full)
    depths, _ = h.compute_depth_for_bboxes(st, 4)
    orient = np.tile(np.array([0.8, 0.6, -0.6, 0.8], np.float32), (len(dy), 1))
    conf = np.tile(np.array([0.3, 0.7], np.float32), (len(dy), 1))
    dims = np.full((len(dy), 3), 0.1, np.float32)
    poses = h.transform_lshape_objects(h.vision_post_process(orient, conf, dims, dy))
    h.update_map_poses(poses)
    expect(ticks[3], 4, 2, 2, 2, len(poses), 1, float(depths[0]))
    pp, valid, npz = h.compute_bbox_pose_ground_removed(full)
    kept = pp[valid.astype(bool)] if npz >= 0 else pp[:0]
    h.update_map_poses(h.transform_lshape_objects(kept))
    expect(ticks[4], 4, 2, 2, 2, len(kept), 1, float(depths[0]))
    assert len(kept) >= 2                                   # the two blobs give poses once the ground is gone
    d2, _ = h.compute_depth_for_bboxes(only_static, 4)
    h.update_map()
    expect(ticks[5], 2, 2, 0, 2, 0, 1, float(d2[0]))
    h.close()


@pytest.mark.timeout(600)
def test_config5_full_size(gvamd):
    """BASELINE configs[4] at full size on one GPU: 10M points, 4000x4000 @ 0.05 m
    (wedges longer than 2048 columns -> the 8-columns-per-thread sector kernel)."""
    config = 5
    h, tfs = make_handle(gvamd, config, perturbed=True)
    g = synth.CONFIGS[config]["grid"]
    assert (h.nx, h.ny) == (4000, 4000)
    x, y, z, _ = synth.cloud_lidar_like(config, 5_000_000)
    x2, y2, z2, _ = synth.cloud_uniform(config, 5_000_000)
    x, y, z = np.concatenate([x, x2]), np.concatenate([y, y2]), np.concatenate([z, z2])
    h.upload_xyz(x, y, z)
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_CELL_IDX | gvamd.FRAME_KEEP_COUNTS)
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
    assert np.array_equal(h.cell_idx(), cell)
    assert np.array_equal(h.hits(), hits)
    assert np.array_equal(h.miss(), miss.astype(np.int32))
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    # the production (pipelined, counts not kept) path on the same input gives the same grid
    h2, _ = make_handle(gvamd, config, perturbed=True)
    h2.upload_xyz(x, y, z)
    h2.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH)
    assert np.array_equal(h2.log_odds(), h.log_odds())
    h.close(); h2.close()


def _check_plane(m, mask, coeff, em, emask, ecoeff, cx, cy, cz, thr=0.04, tag=None):
    """RANSAC ground plane, specified by outcome: normally the device's refined plane equals the oracle's
    to the last bit and so do the masks.  The closed-form eigenvector evaluates atan2 / cos / sin in fp64, where the device
    library and glibc may differ in the last bit: a coefficient may then come out one fp32 ulp apart, and only points whose
    distance sits within 1e-5 of the threshold may change sides."""
    if np.array_equal(coeff, ecoeff):
        assert m == em and np.array_equal(mask, emask), tag
        return
    assert np.allclose(coeff, ecoeff, rtol=3e-7, atol=1e-9), (tag, coeff, ecoeff)
    d = np.abs(ecoeff[0] * cx + ecoeff[1] * cy + ecoeff[2] * cz + ecoeff[3])
    far = np.abs(d - thr) > 1e-5
    assert np.array_equal(mask[far], emask[far]), tag
    assert abs(int(m) - int(em)) <= int(np.count_nonzero(~far)), tag


def _ground_scene(tfs, seed=3):
    """ground plane (camera y ~ +1.6, slightly tilted, 2 cm noise) + the cluster scene on top"""
    x, y, z, _, K, b = _cluster_scene(tfs, n_clusters=16, pts_per=400, seed=seed)
    st = synth.Stream(seed, 33)
    ng = 60_000
    gx = st.uniform(ng, 1.0, 60.0)
    gy = st.uniform(ng, -40.0, 40.0)
    gz = (np.float32(-1.3) + np.float32(0.01) * gx + st.uniform(ng, -0.02, 0.02)).astype(np.float32)  # lidar z up
    x = np.concatenate([x, gx]); y = np.concatenate([y, gy]); z = np.concatenate([z, gz])
    return x, y, z, K, b


def test_segment_ground_plane_matches_oracle(gvamd):
    """A12 by outcome: same counter-based hypotheses, inlier counts of all of them in one pass, selection
    and the fp64 moment refinement (fixed 64-ary sum tree) on the device: mask and coefficients equal the
    oracle's; the recovered plane is the planted one."""
    h, tfs = make_handle(gvamd, 2, perturbed=True)
    x, y, z, K, b = _ground_scene(tfs)
    h.upload_xyz(x, y, z)
    m, mask, coeff = h.segment_ground_plane()
    cx, cy, cz = ol.transform_cloud(ol.tf_to_matrix4f(tfs["cam_lidar"]), x, y, z)
    em, emask, ecoeff = ol.segment_ground_plane(cx, cy, cz)
    assert em > 50_000
    _check_plane(m, mask, coeff, em, emask, ecoeff, cx, cy, cz)
    # planted ground points are (nearly all) found, cluster points above ground are not
    ng = 60_000
    assert mask[-ng:].mean() > 0.9
    assert mask[:16 * 400].mean() < 0.2
    # no plane in pure noise of 2 points / failure path
    h.upload_xyz(x[:2], y[:2], z[:2])
    m2, _, _ = h.segment_ground_plane()
    assert m2 == 0
    h.close()


def test_compute_bbox_pose_ground_removed(gvamd):
    """computeBBoxPose in full: segmentGroundPlane -> extractCloudPerBBox -> radius filter -> PCA"""
    h, tfs = make_handle(gvamd, 2, perturbed=True)
    x, y, z, K, b = _ground_scene(tfs, seed=9)
    h.upload_xyz(x, y, z)
    poses, valid, npz = h.compute_bbox_pose_ground_removed(b)
    cx, cy, cz = ol.transform_cloud(ol.tf_to_matrix4f(tfs["cam_lidar"]), x, y, z)
    em, ground, _ = ol.segment_ground_plane(cx, cy, cz)
    keep_pts = ground == 0
    ids = ol.extract_cloud_per_bbox(K, cx[keep_pts], cy[keep_pts], cz[keep_pts], b, synth.IMG_W, synth.IMG_H)
    sx, sy, sz = cx[keep_pts], cy[keep_pts], cz[keep_pts]
    nv = 0
    for i in range(len(b)):
        sel = ids == i
        kp = ol.radius_outlier(sx[sel], sy[sel], sz[sel], 0.4, 10).astype(bool)
        ok, e = ol.pca_bbox(sx[sel][kp], sy[sel][kp], sz[sel][kp])
        assert bool(valid[i]) == ok
        if ok:
            nv += 1
            _check_pose(poses[i], e, i)
    assert nv == npz >= 8
    h.close()


def _check_pose(p, e, tag):
    """centre and extents within SURVEY A11's 1e-4 of the oracle's arithmetic.  The oracle follows the reference's
    order-dependent sums (fp32 running sums for the means); the device takes the same sums exactly (fixed point,
    gv_cloudops.hip), so the two differ by the rounding the REFERENCE accumulates (~sqrt(n) ulp of fp32), not by
    anything the device approximates -- _check_pose_fp64 holds the device against fp64 numpy, much tighter.
    The orientation is setRPY(0, -angle, 0) with the angle in DEGREES handed over as radians
    (cloud_detections.cpp:227,236): an axis off by 1e-6 rad moves the fp32 angle by 6e-5, sin / cos of half of it by
    3e-5."""
    for f in ("px", "py", "pz", "length", "width"):
        assert p[f] == pytest.approx(e[f], rel=1e-4, abs=1e-4), (tag, f)
    for f in ("qx", "qy", "qz", "qw"):
        assert p[f] == pytest.approx(e[f], abs=2e-4), (tag, f)


def _check_pose_fp64(p, x, y, z, tag):
    """second opinion that does not go through the oracle: the kept points' means in fp64 numpy.  The device's sums
    are exact to 2^-28 m per term, so its centre is the fp64 mean rounded to fp32 (half an ulp + the quantisation)."""
    for f, a in (("px", x), ("py", y), ("pz", z)):
        m = float(np.mean(a.astype(np.float64)))
        assert abs(p[f] - m) <= 1.3e-7 * max(1.0, abs(m)) + 1e-8, (tag, f, p[f], m)
    D = np.stack([z.astype(np.float64), x.astype(np.float64)], axis=1)
    mean = D.mean(axis=0)
    cov = (D - mean).T @ (D - mean) / len(x)
    wv, V = np.linalg.eigh(cov)
    if not (wv[1] - wv[0] >= 1e-3 * wv[1] and wv[1] > 0):
        return   # (nearly) isotropic or degenerate: the axes are ill-conditioned, extents are compared through the oracle only
    major = V[:, 1] * (1.0 if V[0, 1] >= 0 else -1.0)
    minor = np.array([-major[1], major[0]])
    pl, pw = (D - mean) @ major, (D - mean) @ minor
    cond = wv[1] / (wv[1] - wv[0])
    tol = 1e-4 * max(1.0, float(pl.max() - pl.min())) * max(1.0, cond)
    assert abs(p["length"] - (pl.max() - pl.min())) <= tol, (tag, "length")
    assert abs(p["width"] - (pw.max() - pw.min())) <= tol, (tag, "width")


def _radius_keep_ckdtree(x, y, z, r=0.4, min_pts=10):
    """RadiusOutlierRemoval keep flags of one bbox cloud at sizes the oracle's all-pairs loop cannot reach:
    neighbour counts from scipy's cKDTree just inside and just outside the radius decide every point whose
    outcome does not hinge on a neighbour within 1e-6 (relative) of the radius; the few that do are counted
    exactly, in fp32 with the reference's operation order.  (Checked against the oracle itself below.)"""
    from scipy.spatial import cKDTree
    n = len(x)
    if n == 0:
        return np.zeros(0, dtype=bool)
    pts = np.stack([x, y, z], axis=1).astype(np.float64)
    tree = cKDTree(pts)
    lo = tree.query_ball_point(pts, r * (1 - 1e-6), return_length=True)
    hi = tree.query_ball_point(pts, r * (1 + 1e-6), return_length=True)
    keep = lo >= min_pts + 1
    r2 = np.float32(r * r)
    if float(r2) > r * r:
        r2 = np.nextafter(r2, np.float32(-np.inf))
    for i in np.nonzero((lo < min_pts + 1) & (hi >= min_pts + 1))[0]:
        d = x - x[i]; acc = d * d
        d = y - y[i]; acc = acc + d * d
        d = z - z[i]; acc = acc + d * d
        keep[i] = np.count_nonzero(acc <= r2) >= min_pts + 1
    return keep


def _pose_reference(cx, cy, cz, K, b, keep_fn):
    """per bbox: ids as extractCloudPerBBox, radius filter, oracle PCA rectangle on the kept points in cloud order"""
    ids = ol.extract_cloud_per_bbox(K, cx, cy, cz, b, synth.IMG_W, synth.IMG_H)
    out = []
    for i in range(len(b)):
        sel = ids == i
        sx, sy, sz = cx[sel], cy[sel], cz[sel]
        kp = keep_fn(sx, sy, sz)
        out.append(ol.pca_bbox(sx[kp], sy[kp], sz[kp]) + (int(kp.sum()), int(sel.sum())))
    return ids, out


def _large_scene(tfs, n_total=1_000_000, seed=17):
    """config-3 sized scene with structure: a ground plane, 40 dense objects in front of the camera (each inside
    its own pixel bbox), a lidar-like near field and uniform clutter; shuffled, as a sensor delivers it."""
    rng = np.random.default_rng(seed)
    n_obj, per = 40, 6000
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
    ng = 450_000
    gx_ = rng.uniform(1.0, 90.0, ng); gy_ = rng.uniform(-60.0, 60.0, ng)
    xs.append(gx_); ys.append(gy_); zs.append(-1.75 + 0.004 * gx_ + rng.normal(0, 0.012, ng))
    nl = 150_000
    lx, ly, lz, _ = synth.cloud_lidar_like(3, nl, seed_extra=seed)
    xs.append(lx); ys.append(ly); zs.append(lz)
    nu = n_total - n_obj * per - ng - nl
    xs.append(rng.uniform(-44, 176, nu)); ys.append(rng.uniform(-110, 110, nu)); zs.append(rng.uniform(-2, 4, nu))
    x = np.concatenate(xs).astype(np.float32); y = np.concatenate(ys).astype(np.float32); z = np.concatenate(zs).astype(np.float32)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
    boxes = []
    for k in range(n_obj):
        sl = slice(k * per, (k + 1) * per)
        ok = cz[sl] > 0.1
        if ok.sum() < 100:
            continue
        u = synth.FX * cx[sl][ok] / cz[sl][ok] + synth.CX
        v = synth.FY * cy[sl][ok] / cz[sl][ok] + synth.CY
        x0, x1 = max(0.0, np.percentile(u, 2)), min(639.0, np.percentile(u, 98))
        y0, y1 = max(0.0, np.percentile(v, 2)), min(479.0, np.percentile(v, 98))
        if x1 - x0 > 3 and y1 - y0 > 3:
            boxes.append((float(np.float32(x0 + 0.25)), float(np.float32(y0 + 0.5)), float(np.float32(x1 + 0.75)), float(np.float32(y1))))
    b = np.zeros(len(boxes), dtype=synth.BBOX_DTYPE)
    for i, (x0, y0, x1, y1) in enumerate(boxes):
        b[i] = (x0, y0, x1, y1, 0.99 - 0.01 * i, [9, 2, 0, 1, 5][i % 5])
    perm = rng.permutation(len(x))
    return x[perm], y[perm], z[perm], b


def test_radius_reference_helper_equals_oracle():
    """the cKDTree-based keep flags used at 1 M points equal the oracle's all-pairs filter (sizes it can do)"""
    rng = np.random.default_rng(2)
    cen = rng.uniform(-6, 6, (10, 3))
    pts = np.concatenate([cen[i] + rng.normal(0, rng.uniform(0.1, 0.5), (500, 3)) for i in range(10)]
                         + [rng.uniform(-8, 8, (3000, 3))]).astype(np.float32)
    x, y, z = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
    want = ol.radius_outlier(x, y, z, 0.4, 10).astype(bool)
    assert 0.2 < want.mean() < 0.9
    assert np.array_equal(_radius_keep_ckdtree(x, y, z), want)


@pytest.mark.timeout(900)
def test_pca_path_at_config3_size(gvamd):
    """SURVEY 8(a) A2/A3, A11, A12 at BASELINE configs[2] size (1 M points, ~40 bboxes; round-2 verdict: these
    kernels had only ever run below 70 k points): kNN depths and distances bit-equal to the oracle; RANSAC mask
    and coefficients equal; PCA poses of computeBBoxPose with and without ground removal within 1e-6 of the
    oracle's arithmetic on the same kept points (radius filter checked through cKDTree + exact fp32 counts)."""
    h, tfs = make_handle(gvamd, 3, perturbed=True)
    x, y, z, b = _large_scene(tfs)
    assert len(x) == 1_000_000 and len(b) >= 30
    h.upload_xyz(x, y, z)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    # --- kNN depth (cloud_detections.cpp:8-87)
    for k in (10, 32):
        depths, d2 = h.compute_depth_for_bboxes(b, k)
        front = cz > 0
        u, v, d = ol.project_points(K, cx, cy, cz)
        edepths, ed2 = ol.depth_for_bboxes(u, v, d, b, k)
        assert np.array_equal(d2, ed2), k
        assert np.array_equal(depths, edepths), k
    # --- RANSAC ground plane (:105-138, by outcome)
    m, mask, coeff = h.segment_ground_plane()
    em, emask, ecoeff = ol.segment_ground_plane(cx, cy, cz)
    assert em > 350_000
    _check_plane(m, mask, coeff, em, emask, ecoeff, cx, cy, cz)
    # --- computeBBoxPose without / with ground removal (:140-321)
    poses, valid = h.compute_bbox_pose(b)
    ids, ref = _pose_reference(cx, cy, cz, K, b, _radius_keep_ckdtree)
    assert np.array_equal(h.bbox_id(), ids)
    n_valid = 0
    for i, (ok, e, nk, ns) in enumerate(ref):
        assert bool(valid[i]) == ok, (i, nk, ns)
        if ok:
            n_valid += 1
            _check_pose(poses[i], e, (i, nk))
    assert n_valid == sum(1 for r in ref if r[0]) >= 15   # the oracle's own count; the scene gives a few dozen
    assert max(r[2] for r in ref) >= 4000   # a bbox with thousands of kept points
    poses2, valid2, npz = h.compute_bbox_pose_ground_removed(b)
    g = emask == 0
    _, ref2 = _pose_reference(cx[g], cy[g], cz[g], K, b, _radius_keep_ckdtree)
    nv = 0
    for i, (ok, e, nk, ns) in enumerate(ref2):
        assert bool(valid2[i]) == ok, (i, nk, ns)
        if ok:
            nv += 1
            _check_pose(poses2[i], e, (i, nk))
    assert nv == npz == sum(1 for r in ref2 if r[0]) >= 15
    h.close()


def test_bbox_pose_beyond_one_million_points(gvamd):
    """1.3 M points: more than 1024 blocks of the kept-point split, and the kNN / radius-filter grids stride
    over the cloud.  Poses of
    computeBBoxPose (no ground removal) against the oracle's arithmetic on the cKDTree-kept points."""
    h, tfs = make_handle(gvamd, 3, perturbed=True)
    x, y, z, b = _large_scene(tfs, n_total=1_300_000, seed=29)
    assert len(x) == 1_300_000 and len(b) >= 30
    h.upload_xyz(x, y, z)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    poses, valid = h.compute_bbox_pose(b)
    ids, ref = _pose_reference(cx, cy, cz, K, b, _radius_keep_ckdtree)
    assert np.array_equal(h.bbox_id(), ids)
    n_valid = 0
    for i, (ok, e, nk, ns) in enumerate(ref):
        assert bool(valid[i]) == ok, (i, nk, ns)
        if ok:
            n_valid += 1
            _check_pose(poses[i], e, (i, nk))
    assert n_valid == sum(1 for r in ref if r[0]) >= 10   # the oracle's own count (round-3 verdict: not a constant)
    h.close()


def _oracle_rect(og, p):
    """index rectangle updateGridCellsFast (occupancy_grid.cpp:140-183) derives from a base-frame pose, or None"""
    hx, hy = float(p["length"]) / 2.0, float(p["width"]) / 2.0
    px, py = float(p["px"]), float(p["py"])
    idx = []
    for cx, cy in ((px - hx, py - hy), (px + hx, py - hy), (px + hx, py + hy), (px - hx, py + hy)):
        ok, ix, iy = og.get_index(cx, cy)
        if not ok:
            return None
        idx.append((ix, iy))
    return (min(i[0] for i in idx), min(i[1] for i in idx), max(i[0] for i in idx), max(i[1] for i in idx))


def _base_poses(tfs, cam_poses):
    out = np.zeros(len(cam_poses), dtype=synth.LSHAPE_DTYPE)
    for i, e in enumerate(cam_poses):
        o = ol.tf_pose(tfs["base_cam"], [e[k] for k in ("px", "py", "pz", "qx", "qy", "qz", "qw")])
        out[i] = tuple(o.tolist()) + (e["length"], e["width"], e["height"])
    return out


@pytest.mark.timeout(1200)
def test_tick_fused_equals_call_sequence_and_oracle(gvamd):
    """The reference's timerCallback from filterBBoxes on (grid_vision_node.cpp:153-244) at BASELINE configs[2] size on
    a scene with objects, three ways:
      A  gv_tick: ONE batch of device work -- kNN depth on a lane, RANSAC + per-box clouds + radius filter + PCA (or
         the orientation-network geometry), camera->base, rectangles, map update, int8 pack, grid to pinned memory --
         and one host wait;
      B  the call-by-call sequence over the same library (one synchronous call per reference function);
      C  the oracle.
    A == B bit for bit (depths, points, poses, both layers, the packed grid).  Against C: depths bit-equal; poses
    within SURVEY A11 / A14's tolerance; the grid bit-equal to the oracle's updateMap fed with the device's poses, and
    the index rectangles of the oracle's own poses are the same ones (a differing rectangle would be a pose within
    tolerance whose corner sits on a cell border: counted, none expected)."""
    hA, tfs = make_handle(gvamd, 3, perturbed=True)
    hB, _ = make_handle(gvamd, 3, perturbed=True)
    g = synth.CONFIGS[3]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    x, y, z, b = _large_scene(tfs)
    for h in (hA, hB):
        h.upload_xyz(x, y, z)
    st, dy = gvamd.filter_bboxes(b)
    assert len(st) >= 5 and len(dy) >= 20
    orient, conf, dims = synth.network_outputs(len(dy))
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
    # --- oracle side, once: kNN depth of the static boxes, poses of both branches
    u, v, d = ol.project_points(K, cx, cy, cz)
    edepth, _ = ol.depth_for_bboxes(u, v, d, st, 4)
    kinv = ol.k_inverse(K)
    epts = np.array([ol.tf_point(tfs["base_cam"], ol.pixel_to_3d(
        np.float32(bb["x_min"] + ((bb["x_max"] - bb["x_min"]) / np.float32(2.0))),
        np.float32(bb["y_min"] + ((bb["y_max"] - bb["y_min"]) / np.float32(2.0))), dep, kinv)) for bb, dep in zip(st, edepth)])
    _, emask, _ = ol.segment_ground_plane(cx, cy, cz)
    keep = emask == 0
    _, ref = _pose_reference(cx[keep], cy[keep], cz[keep], K, b, _radius_keep_ckdtree)
    e_pca = _base_poses(tfs, [e for ok, e, _, _ in ref if ok])
    e_vis = _base_poses(tfs, ol.post_process(ol.make_cam(), orient, conf, dims, dy))
    assert len(e_pca) >= 25 and len(e_vis) == len(dy)
    pin = gvamd.PinnedI8(hA.G)
    ties = 0
    for tick, branch in enumerate(["pca", "vision", "pca", "static_only", "none", "vision_no_net"]):
        boxes = {"static_only": st, "none": b[:0]}.get(branch, b)
        vision = branch.startswith("vision")
        net = (orient, conf, dims) if branch == "vision" else None
        r = hA.tick(boxes, k_near=4, vision=vision, net=net, grid_out=pin.array)
        # --- B: the call sequence
        bst, bdy = gvamd.filter_bboxes(boxes) if len(boxes) else (boxes, boxes)
        bdepth = hB.compute_depth_for_bboxes(bst, 4)[0] if len(bst) else np.zeros(0, np.float32)
        bpts = hB.convert_pixels_to_3d(bst, bdepth) if len(bst) else np.zeros((0, 3))
        if len(bdy) == 0:
            bposes = np.zeros(0, dtype=synth.LSHAPE_DTYPE)
            hB.update_map()
        else:
            if branch == "vision":
                bposes = hB.transform_lshape_objects(hB.vision_post_process(orient, conf, dims, bdy))
            elif branch == "vision_no_net":
                bposes = np.zeros(0, dtype=synth.LSHAPE_DTYPE)
            else:
                pp, valid, npz = hB.compute_bbox_pose_ground_removed(boxes)
                bposes = hB.transform_lshape_objects(pp[valid.astype(bool)] if npz >= 0 else pp[:0])
            hB.update_map_poses(bposes)
        assert r["n_static"] == len(bst) and r["n_dynamic"] == len(bdy), branch
        assert r["depths"].tobytes() == bdepth.tobytes(), branch
        assert r["base_points"].tobytes() == np.ascontiguousarray(bpts).tobytes(), branch
        assert r["poses"].tobytes() == bposes.tobytes(), branch
        lo = hA.log_odds()
        assert np.array_equal(lo, hB.log_odds()), branch
        assert np.array_equal(hA.occupancy(), hB.occupancy()), branch
        i8 = hA.to_occupancy_grid()[0]
        assert np.array_equal(i8, hB.to_occupancy_grid()[0]), branch
        assert np.array_equal(pin.array, i8), branch          # the grid the tick itself delivered
        # --- C: the oracle
        if len(bst):
            assert np.array_equal(r["depths"], edepth), branch
            assert np.array_equal(r["base_points"], epts), branch
        want = {"pca": e_pca, "vision": e_vis}.get(branch, e_pca[:0])
        assert len(r["poses"]) == len(want), branch
        for i, (p, e) in enumerate(zip(r["poses"], want)):
            for f in ("px", "py", "pz", "length", "width"):
                assert p[f] == pytest.approx(e[f], rel=1e-4, abs=1e-4), (branch, i, f)
            ties += _oracle_rect(og, p) != _oracle_rect(og, e)
        if len(bdy) == 0:
            og.update_map()
        else:
            og.update_map_poses(r["poses"])
        nlo, _, _ = check_grid(hA, og)
        assert nlo == 0, branch
    assert ties == 0, f"{ties} rectangles differ between the oracle's poses and the device's (poses within tolerance, corner on a cell border)"
    pin.close()
    hA.close(); hB.close()


def test_tick_equals_call_sequence_at_ten_million_points(gvamd):
    """BASELINE configs[4]'s point count through the tick: a scene with objects of 10 M points (60 000 per object: boxes
    of tens of thousands of kept points, a bucket table of 4 M entries whose block offsets no longer fit the kernels'
    LDS copy).  The oracle does not finish at this size; the size-independent property is that the fused tick and the
    call-by-call sequence over the same library agree bit for bit -- depths, poses, both layers -- and that every dynamic
    box with an object yields a pose."""
    hA, tfs = make_handle(gvamd, 3, perturbed=True)
    hB, _ = make_handle(gvamd, 3, perturbed=True)
    x, y, z, b = synth.scene_with_objects(tfs, n_total=10_000_000, seed=23, per=60_000)
    for h in (hA, hB):
        h.upload_xyz(x, y, z)
    st, dy = gvamd.filter_bboxes(b)
    assert len(st) >= 5 and len(dy) >= 20
    for rep in range(2):
        r = hA.tick(b, k_near=4)
        bdepth = hB.compute_depth_for_bboxes(st, 4)[0]
        pp, valid, npz = hB.compute_bbox_pose_ground_removed(b)
        assert npz >= 20
        bposes = hB.transform_lshape_objects(pp[valid.astype(bool)])
        hB.update_map_poses(bposes)
        assert r["depths"].tobytes() == bdepth.tobytes()
        assert r["poses"].tobytes() == bposes.tobytes()
        assert np.array_equal(hA.log_odds(), hB.log_odds())
        assert np.array_equal(hA.to_occupancy_grid()[0], hB.to_occupancy_grid()[0])
    # (at ten times the density the clutter inside a box's frustum survives the radius filter: the rectangles span it)
    assert len(r["poses"]) >= 20 and np.all(r["poses"]["length"] > 0.5) and np.all(np.isfinite(r["poses"]["px"]))
    hA.close(); hB.close()


def test_tick_lidar_extension_and_error_paths(gvamd):
    """the tick with the [EXTENSION] map update (hit counts + free space inside the same batch) equals the fused frame
    fed with the tick's own poses; argument / state errors; a tick while frames are in flight drains them first"""
    config = 2
    hA, tfs = make_handle(gvamd, config, perturbed=True)
    hB, _ = make_handle(gvamd, config, perturbed=True)
    x, y, z, K, b = _ground_scene(tfs, seed=5)
    for h in (hA, hB):
        h.upload_xyz(x, y, z)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH
    hA.set_detections(flags)
    hB.set_detections(flags)
    for h in (hA, hB):
        h.enqueue_frame(); h.enqueue_frame()      # frames in flight when the tick arrives
    for vision in (False, True):
        st, dy = gvamd.filter_bboxes(b)
        net = synth.network_outputs(len(dy)) if vision else None
        r = hA.tick(b, k_near=4, vision=vision, net=net, lidar_bin=True, lidar_raymarch=True)
        assert len(r["poses"]) >= (len(dy) if vision else 5)
        hB.process_frame(flags | gvamd.FRAME_KEEP_COUNTS, poses=r["poses"])
        assert np.array_equal(hA.log_odds(), hB.log_odds())
        assert np.array_equal(hA.to_occupancy_grid()[0], hB.to_occupancy_grid()[0])
        assert np.array_equal(hA.hits(), hB.hits())
    # errors
    hA.tick_enqueue(b, k_near=4)
    with pytest.raises(gvamd.GVError) as e:
        hA.tick_enqueue(b, k_near=4)           # one tick at a time
    assert e.value.code == 5
    for call in (lambda: hA.compute_depth_for_bboxes(b, 4), lambda: hA.compute_bbox_pose(b), lambda: hA.segment_ground_plane()):
        with pytest.raises(gvamd.GVError) as e:
            call()                             # would reuse the pending tick's result block / detection set
        assert e.value.code == 5
    want = hA.tick_wait()
    assert len(want["depths"]) > 0 and len(want["poses"]) > 0
    hA.compute_depth_for_bboxes(b, 4)          # fine again
    with pytest.raises(gvamd.GVError) as e:
        hA.tick_wait()
    assert e.value.code == 5
    with pytest.raises(gvamd.GVError) as e:
        hA.tick_enqueue(b, k_near=0)
    assert e.value.code == 1
    with pytest.raises(gvamd.GVError) as e:
        hA.tick_enqueue(b, k_near=4, vision=True, net=synth.network_outputs(1))   # outputs for the wrong number of boxes
    assert e.value.code == 1
    hC = gvamd.GridVisionHIP(50, 20, 0.1)
    hC.upload_xyz(x, y, z)
    with pytest.raises(gvamd.GVError) as e:
        hC.tick_enqueue(b, k_near=4)           # transforms never set: the node publishes the stale grid (:160-164)
    assert e.value.code == 6
    assert hC.tick(b[:0])["n_static"] == 0     # no boxes: plain updateMap, needs no transform
    hA.close(); hB.close(); hC.close()


def test_ransac_tree_levels_and_failure_paths(gvamd):
    """the sum tree of the refinement at sizes that end after 1, 2 and 3 levels (n <= 64, <= 4096, > 4096 with
    more than one workgroup), a cloud that is all ground, and more hypotheses than one LDS batch holds"""
    h, tfs = make_handle(gvamd, 2, perturbed=True)
    m_cam = ol.tf_to_matrix4f(tfs["cam_lidar"])
    rng = np.random.default_rng(4)
    for n in (3, 40, 64, 65, 4096, 4097, 70_001, 300_000):
        gx_ = rng.uniform(1.0, 60.0, n); gy_ = rng.uniform(-30.0, 30.0, n)
        gz_ = -1.5 + 0.01 * gx_ + rng.normal(0, 0.01, n)
        k = n // 3
        gz_[:k] = rng.uniform(-1, 3, k)   # a third of the points off the plane
        x, y, z = gx_.astype(np.float32), gy_.astype(np.float32), gz_.astype(np.float32)
        h.upload_xyz(x, y, z)
        cx, cy, cz = ol.transform_cloud(m_cam, x, y, z)
        for iters, seed in ((50, 12345), (7, 99)) + (((2500, 5),) if n == 70_001 else ()):
            m, mask, coeff = h.segment_ground_plane(0.04, iters, seed)
            em, emask, ecoeff = ol.segment_ground_plane(cx, cy, cz, 0.04, iters, seed)
            _check_plane(m, mask, coeff, em, emask, ecoeff, cx, cy, cz, tag=(n, iters))
    # everything on one plane: computeBBoxPose's "empty segmented cloud" (:307-309)
    n = 20_000
    gx_ = rng.uniform(1.0, 60.0, n); gy_ = rng.uniform(-30.0, 30.0, n)
    x, y, z = gx_.astype(np.float32), gy_.astype(np.float32), np.full(n, -1.5, np.float32)
    h.upload_xyz(x, y, z)
    b = synth.detections(3, 12)
    poses, valid, npz = h.compute_bbox_pose_ground_removed(b)
    assert npz == -1 and not valid.any()
    h.close()


def test_pipelined_equals_serial(gvamd, monkeypatch):
    """two-stream frame pipelining (default) vs GV_PIPELINE=0, including the vision-orientation
    mode and detections that change between frames."""
    config = 2
    outs = []
    for pipe in ("1", "0"):
        monkeypatch.setenv("GV_PIPELINE", pipe)
        h, tfs = make_handle(gvamd, config, perturbed=True)
        res = []
        for f in range(6):
            x, y, z, _ = synth.cloud_uniform(config, 50_000, seed_extra=f % 3)
            h.upload_xyz(x, y, z)
            bboxes = synth.detections(3, 20, seed_extra=f)
            if f % 2 == 0:
                h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=bboxes,
                                 poses=synth.lshape_poses(config, 15, seed_extra=f))
            else:
                h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_VISION_ORIENT, bboxes=bboxes,
                                 net=synth.network_outputs(20, seed=f))
            for _ in range(3):
                h.enqueue_frame()      # several frames in flight
        h.synchronize()
        outs.append((h.log_odds(), h.occupancy(), h.to_occupancy_grid()[0]))
        h.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("grid,n", [((50, 20, 0.1), 60_000), ((120, 200, 0.25), 40_000), ((50, 20, 0.3), 20_000),
                                    ((200, 200, 0.1), 1_000_000)])
@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_production_frame_vs_oracle(gvamd, monkeypatch, grid, n, pipeline):
    """The production frame (no KEEP_* flags): int32 hit counts read back from the PIPELINED path are
    bit-exact, and the grid layers equal the oracle's frame after frame, serial and pipelined,
    including a grid with nx % 4 != 0 (generic kernels, no counts without KEEP_COUNTS) and config 3
    at full size."""
    monkeypatch.setenv("GV_PIPELINE", pipeline)
    gx, gy, res = grid
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    tfs = synth.transforms(True)
    tfs["base_lidar"] = np.array([0.0, 0.0, 0.0, 1.0, gx / 3.0 + 1.7, -gy * 0.21, 1.8])
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    st = synth.Stream(4242, gx + n)
    lx, ly = og.g.len_x, og.g.len_y
    poses = np.zeros(12, dtype=synth.LSHAPE_DTYPE)
    poses["px"] = st.uniform(12, og.g.pos_x - lx / 2, og.g.pos_x + lx / 2)
    poses["py"] = st.uniform(12, -ly / 2, ly / 2)
    poses["length"] = st.uniform(12, 0.5, 5.0)
    poses["width"] = st.uniform(12, 0.5, 2.5)
    bboxes = synth.detections(3, 10)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    for frame in range(3):
        x = st.uniform(n, -0.8 * lx, 0.8 * lx)
        y = st.uniform(n, -0.8 * ly, 0.8 * ly)
        z = st.uniform(n, -1.0, 1.0)
        h.upload_xyz(x, y, z)
        h.set_detections(flags, bboxes=bboxes, poses=poses)
        h.enqueue_frame()
        h.enqueue_frame()          # the same cloud twice: two frames in flight when pipelined
        h.synchronize()
        for _ in range(2):
            hits, _, _, ids, _ = oracle_frame(og, tfs, x, y, z, bboxes, poses)
        assert np.array_equal(h.bbox_id(), ids)
        if og.g.nx % 4 == 0:
            assert np.array_equal(h.hits(), hits), "int32 hit counts of the production (pipelined) frame"
            assert int(h.hits().sum()) == int(hits.sum())
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
    h.close()


def _crowded_cloud(n, seed, spread):
    """points packed around the sensor: most of them in one or two 128x128-cell tiles"""
    st = synth.Stream(seed, n)
    r = st.uniform(n, 0.0, 1.0) ** 2 * spread
    a = st.uniform(n, 0.0, 2.0 * np.pi)
    return (r * np.cos(a)).astype(np.float32), (r * np.sin(a)).astype(np.float32), st.uniform(n, -1.0, 1.0)


@pytest.mark.parametrize("n,spread", [(300_000, 6.0), (150_000, 25.0)])
def test_crowded_tiles_are_shared(gvamd, n, spread):
    """A tile that holds more than 32768 keys is histogrammed by several workgroups whose partial
    tiles are summed by the last one to arrive: counts bit-exact, hundreds of hits per cell, several
    frames so that the arrival tickets and per-tile totals have to reset themselves."""
    config = 3
    h, tfs = make_handle(gvamd, config, perturbed=True)
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_CELL_IDX | gvamd.FRAME_KEEP_COUNTS
    for frame in range(3):
        x, y, z = _crowded_cloud(n, 77 + frame, spread)
        h.upload_xyz(x, y, z)
        h.process_frame(flags)
        hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
        assert hits.max() > 50, "fixture must pile points up"
        assert np.array_equal(h.cell_idx(), cell)
        assert np.array_equal(h.hits(), hits)
        assert np.array_equal(h.miss(), miss.astype(np.int32))
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
    # a sparse frame afterwards: no tile is shared any more
    x, y, z, _ = synth.cloud_uniform(config, 50_000)
    h.upload_xyz(x, y, z)
    h.process_frame(flags)
    hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
    assert np.array_equal(h.hits(), hits)
    assert np.array_equal(h.miss(), miss.astype(np.int32))
    h.close()


def test_streaming_ingest_matches_oracle(gvamd):
    """grid_vision_node.cpp:103-106,108-244: a new cloud and new detections every frame.  Six distinct
    clouds / detection sets go through the asynchronous, double-buffered uploads and the pipelined
    frame with no host wait in between; the grid after every frame count must equal the oracle's (a
    frame that read a stale or half-copied cloud or detection set would change it), and the per-point
    outputs of the last frame are compared too."""
    config = 2
    g = synth.CONFIGS[config]["grid"]
    n = 80_000
    clouds, dets = [], []
    pins = []
    for f in range(6):
        gen = synth.cloud_uniform if f % 2 == 0 else synth.cloud_lidar_like
        x, y, z, _ = gen(config, n - 1000 * f, seed_extra=f)
        px, py, pz = (gvamd.PinnedF32(len(x)) for _ in range(3))
        px.array[:], py.array[:], pz.array[:] = x, y, z
        pins.append((px, py, pz))
        clouds.append((x, y, z))
        dets.append((synth.detections(3, 10 + 5 * f, seed_extra=f), synth.lshape_poses(config, 8 + 3 * f, seed_extra=f)))
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_CELL_IDX
    # frame counts 1, 2, 4, 3, 5, 12: the last frame ends on either lane, and 12 frames wrap the four buffer
    # sets three times (the host back-pressure of gv_frame_enqueue is exercised)
    for upto, reps in ((1, 1), (1, 2), (2, 2), (3, 1), (5, 1), (6, 2)):
        h, tfs = make_handle(gvamd, config, perturbed=True)
        og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
        for rep in range(reps):
            for f in range(upto):
                px, py, pz = pins[f]
                h.upload_xyz_async(px.array, py.array, pz.array)
                h.set_detections_async(flags, bboxes=dets[f][0], poses=dets[f][1])
                h.enqueue_frame()
        h.synchronize()
        for rep in range(reps):
            for f in range(upto):
                hits, cells, _, ids, _ = oracle_frame(og, tfs, *clouds[f], dets[f][0], dets[f][1])
        assert np.array_equal(h.hits(), hits)
        assert np.array_equal(h.bbox_id(), ids)
        assert np.array_equal(h.cell_idx(), cells)
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
        h.close()
    for p3 in pins:
        for p in p3:
            p.close()


def test_detection_set_kept_over_several_frames(gvamd):
    """Detections arrive slower than frames (the 2D detector runs below the 20 Hz timer,
    grid_vision_node.cpp:49-50,108-244): a detection set is read by frames on BOTH lanes before the set
    is re-uploaded.  The upload must come after every one of those readers, not only after the reader on
    its own lane (round-2 advisor finding).  Schedules: 2 frames on set A, 1 on B, then A', and longer runs
    with 1..3 frames per set; grids and the last frame's bbox ids must equal the oracle's."""
    config = 2
    g = synth.CONFIGS[config]["grid"]
    x, y, z, _ = synth.cloud_uniform(config, 70_000, seed_extra=3)
    dets = [(synth.detections(3, 12 + 7 * f, seed_extra=20 + f), synth.lshape_poses(config, 6 + 4 * f, seed_extra=20 + f))
            for f in range(7)]
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    for schedule in ((2, 1, 1), (2, 1, 2, 1), (1, 2, 3, 1, 2, 3, 1), (3, 3, 3, 3)):
        h, tfs = make_handle(gvamd, config, perturbed=True)
        og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
        h.upload_xyz(x, y, z)
        for d, reps in enumerate(schedule):
            h.set_detections_async(flags, bboxes=dets[d][0], poses=dets[d][1])
            for _ in range(reps):
                h.enqueue_frame()
        h.synchronize()
        ids = None
        for d, reps in enumerate(schedule):
            for _ in range(reps):
                _, _, _, ids, _ = oracle_frame(og, tfs, x, y, z, dets[d][0], dets[d][1])
        assert np.array_equal(h.bbox_id(), ids), schedule
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0, schedule
        h.close()


def test_stream_contract_public_stream(gvamd):
    """gv_stream contract: every frame's grid pass runs on gv_stream(h) behind the frame's other kernels
    (which run on the two internal lanes), so a copy the caller puts there between two gv_frame_enqueue
    calls sees exactly the frames enqueued before it -- without a host wait and without stalling the
    frames after it.  Seven frames are enqueued; the packed grid is read asynchronously after the 4th
    and after the 7th (gv_to_occupancy_grid_async) and ONLY gv_stream(h) is synchronised
    (hipStreamSynchronize through ctypes).  Both must equal fully drained handles run for 4 / 7 frames."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    config = 2
    x, y, z, _ = synth.cloud_lidar_like(config, 90_000)
    poses = synth.lshape_poses(config, 20)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH
    G = synth.CONFIGS[config]["grid"].nx * synth.CONFIGS[config]["grid"].ny
    pins = [gvamd.PinnedF32((G + 3) // 4) for _ in range(2)]
    h, tfs = make_handle(gvamd, config, perturbed=True)
    h.upload_xyz(x, y, z)
    h.set_detections(flags, poses=poses)
    bufs = [p.array.view(np.int8)[:h.G] for p in pins]
    for b in bufs:
        b[:] = 0
    for f in range(7):
        h.enqueue_frame()
        if f == 3:
            h.to_occupancy_grid_async(bufs[0])
    h.to_occupancy_grid_async(bufs[1])
    rc = hip.hipStreamSynchronize(ctypes.c_void_p(h.stream()))   # the public stream only: the lanes are not waited for by the host
    assert rc == 0
    got = [b.copy() for b in bufs]
    h.synchronize()
    h.close()
    for frames, g in zip((4, 7), got):
        r, _ = make_handle(gvamd, config, perturbed=True)
        r.upload_xyz(x, y, z)
        r.set_detections(flags, poses=poses)
        for _ in range(frames):
            r.enqueue_frame()
        r.synchronize()
        want = r.to_occupancy_grid()[0]
        r.close()
        assert np.array_equal(g, want), f"grid read on the public stream after {frames} frames differs"
    assert not np.array_equal(got[0], got[1])   # the scene does change between the two reads
    for p in pins:
        p.close()


def test_publish_grid_every_frame_while_clouds_stream(gvamd):
    """gv_publish_grid_async: the node's "publish the grid every tick" while clouds stream in.  Eight frames, each with a
    fresh cloud (asynchronous upload) and the grid of the frame written to pinned host memory by a kernel on the public
    stream; a ninth and tenth frame on the resident cloud; one grid into PAGEABLE memory (falls back to the copy
    command).  Every grid received equals the oracle's after that many frames, nothing synchronised in between."""
    config = 2
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    h, tfs = make_handle(gvamd, config, perturbed=True)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH
    h.set_detections(flags)
    n_frames = 10
    outs = [gvamd.PinnedI8(h.G) for _ in range(n_frames)]
    pageable = np.zeros(h.G, np.int8)
    clouds, want = [], []
    for f in range(8):
        x, y, z, _ = (synth.cloud_lidar_like if f % 2 else synth.cloud_uniform)(config, 150_000 + 7_000 * f, seed_extra=70 + f)
        pin = [gvamd.PinnedF32(len(x)) for _ in range(3)]
        for p, a in zip(pin, (x, y, z)):
            p.array[:] = a
        clouds.append((x, y, z, pin))
    for f in range(n_frames):
        x, y, z, pin = clouds[min(f, 7)]
        if f < 8:
            h.upload_xyz_async(pin[0].array, pin[1].array, pin[2].array)
        h.enqueue_frame()
        h.publish_grid_async(outs[f].array)
        if f == 5:
            h.publish_grid_async(pageable)
        oracle_frame(og, tfs, x, y, z)
        want.append(og.to_occupancy_grid()[0].copy())
    h.synchronize()
    assert np.array_equal(pageable, outs[5].array)
    for f in range(n_frames):
        diff = np.abs(outs[f].array.astype(np.int16) - want[f].astype(np.int16))
        assert diff.max() <= 1 and np.count_nonzero(diff) <= 1e-4 * h.G, f
    h.close()
    for o in outs:
        o.close()
    for _, _, _, pin in clouds:
        for p in pin:
            p.close()


def test_publish_grid_size_not_a_multiple_of_16(gvamd):
    """gv_publish_grid_async on a grid whose byte count is no multiple of 16 (the kernel moves 16 bytes per lane, the tail
    goes by a copy command) and off the tile path (nx % 4 != 0): equals gv_to_occupancy_grid byte for byte"""
    h = gvamd.GridVisionHIP(21, 19, 0.1)   # 210 x 190 cells = 39900 bytes = 16 * 2493 + 12
    assert h.G % 16 != 0
    tfs = synth.transforms(True)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    h.update_map_poses(synth.lshape_poses(1, 6))
    h.update_map_poses(synth.lshape_poses(1, 4, seed_extra=3))
    pin = gvamd.PinnedI8(h.G)
    pin.array[:] = 77
    h.publish_grid_async(pin.array)
    h.synchronize()
    want = h.to_occupancy_grid()[0]
    assert np.array_equal(pin.array, want) and not np.any(want == 77)
    # a destination inside pinned memory that is not 16-byte aligned takes the copy command
    big = gvamd.PinnedI8(h.G + 16)
    big.array[:] = 77
    h.publish_grid_async(big.array[3:3 + h.G])
    h.synchronize()
    assert np.array_equal(big.array[3:3 + h.G], want) and np.all(big.array[:3] == 77) and np.all(big.array[3 + h.G:] == 77)
    big.close()
    pin.close()
    h.close()


def test_device_layers_are_the_resident_grid(gvamd):
    """gv_device_layers: the device pointers a device-side consumer reads behind a frame on gv_stream hold what the
    host getters return"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    h, tfs = make_handle(gvamd, 1, perturbed=True)
    x, y, z, _ = synth.cloud_uniform(1)
    h.upload_xyz(x, y, z)
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH, poses=synth.lshape_poses(1, 6))
    occ, lo, pr = h.device_layers()
    assert occ and lo and pr
    for ptr, want in ((occ, h.to_occupancy_grid()[0]), (lo, h.log_odds()), (pr, h.occupancy())):
        got = np.empty_like(want)
        assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), C.c_void_p(ptr), C.c_size_t(got.nbytes), 2) == 0
        assert np.array_equal(got, want)
    h.close()


def test_standalone_calls_keep_frame_detections(gvamd):
    """The reference-surface calls (extractCloudPerBBox, updateMap(poses), ...) must not change what the
    next gv_frame_enqueue uses: set_detections(50 boxes) -> extract_cloud_per_bbox(3 other boxes) ->
    update_map_poses(other poses) -> enqueue_frame still runs with the 50."""
    config = 2
    h, tfs = make_handle(gvamd, config, perturbed=True)
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    x, y, z, _ = synth.cloud_uniform(config, 60_000)
    h.upload_xyz(x, y, z)
    bboxes, poses = synth.detections(3, 50), synth.lshape_poses(config, 50)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    h.set_detections(flags, bboxes=bboxes, poses=poses)
    other = synth.detections(3, 3, seed_extra=9)
    ids3, _ = h.extract_cloud_per_bbox(other)
    cx, cy, cz = ol.transform_cloud(ol.tf_to_matrix4f(tfs["cam_lidar"]), x, y, z)
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    assert np.array_equal(ids3, ol.extract_cloud_per_bbox(K, cx, cy, cz, other, synth.IMG_W, synth.IMG_H))
    other_poses = synth.lshape_poses(config, 5, seed_extra=4)
    h.update_map_poses(other_poses)
    og.update_map_poses(other_poses)
    h.enqueue_frame()
    h.synchronize()
    hits, _, _, ids, _ = oracle_frame(og, tfs, x, y, z, bboxes, poses)
    assert np.array_equal(h.bbox_id(), ids)
    assert np.array_equal(h.hits(), hits)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.close()


def test_diagnostic_switches_do_not_exist_in_the_product(gvamd, monkeypatch):
    """GV_ABLATE / GV_POINTS_ABLATE / GV_SECTOR_DBG only exist in the -DGV_DIAG build: the shipped
    library ignores them and the grid equals the oracle's."""
    for k, v in (("GV_ABLATE", "6"), ("GV_POINTS_ABLATE", "3"), ("GV_SECTOR_DBG", "1"), ("GV_HIT_COUNTS", "0")):
        monkeypatch.setenv(k, v)
    config = 2
    h, tfs = make_handle(gvamd, config, perturbed=True)
    g = synth.CONFIGS[config]["grid"]
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    x, y, z, _ = synth.cloud_uniform(config, 70_000)
    bboxes = synth.detections(3, 12)
    h.upload_xyz(x, y, z)
    h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST, bboxes=bboxes)
    for _ in range(2):
        h.enqueue_frame()
        hits, _, _, ids, _ = oracle_frame(og, tfs, x, y, z, bboxes)
    h.synchronize()
    assert np.array_equal(h.hits(), hits)
    assert np.array_equal(h.bbox_id(), ids)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.close()


@pytest.mark.parametrize("origin", ["corner++", "corner--", "edge_x", "edge_y", "near_corner", "centre"])
def test_sector_counts_follow_origin(gvamd, origin):
    """The sector count of every octant follows the length of its wedge (distance from the origin cell
    to the map edge): origins in a corner, on an edge and next to them give octants with zero, a
    handful and the full number of columns in one launch.  Miss grid and layers vs the oracle."""
    gx, gy, res = 120, 100, 0.25        # 480 x 400 cells
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    lx, ly = og.g.len_x, og.g.len_y
    hix, hiy = og.g.pos_x + lx / 2, ly / 2
    pos = {"corner++": (hix - 0.01, hiy - 0.01), "corner--": (hix - lx + 0.26, hiy - ly + 0.26),
           "edge_x": (hix - 0.1, 0.3), "edge_y": (og.g.pos_x + 1.0, hiy - ly + 0.3),
           "near_corner": (hix - 3.1, hiy - 2.2), "centre": (og.g.pos_x, 0.0)}[origin]
    tfs = synth.transforms(True)
    tfs["base_lidar"] = np.array([0.0, 0.0, 0.0, 1.0, pos[0], pos[1], 1.8])
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    st = synth.Stream(777, len(origin))
    n = 120_000
    # lidar-frame points = base-frame minus the translation: cover the map and 10 % beyond
    x = st.uniform(n, hix - 1.1 * lx - pos[0], hix + 0.1 * lx - pos[0])
    y = st.uniform(n, hiy - 1.1 * ly - pos[1], hiy + 0.1 * ly - pos[1])
    z = st.uniform(n, -1.0, 1.0)
    h.upload_xyz(x, y, z)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS
    for frame in range(2):
        h.process_frame(flags)
        hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
        assert np.array_equal(h.hits(), hits)
        assert np.array_equal(h.miss(), miss.astype(np.int32))
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0
    assert miss.sum() > 1000
    h.close()


def test_sector_path_random_scenes(gvamd):
    """Randomised scenes for the sector ray stage (seeded): grid shape and resolution, origin anywhere
    in or next to the map, cloud density from a few points to several per cell, clouds with angular
    gaps and axis-aligned walls (many ends on one slope).  Hits and miss grids vs the oracle."""
    st = synth.Stream(20260104, 7)
    for case in range(24):
        k = int(st.integers(1, 2, 60)[0]); gy = int(st.integers(1, 5, 120)[0])
        res = [0.25, 0.5, 0.2, 1.0][case % 4]
        gx = [k, 2 * k, 4 * k, 4 * k][case % 4]      # nx = gx / res is a multiple of 4: the tile (sector) path
        h = gvamd.GridVisionHIP(gx, gy, res)
        assert h.nx % 4 == 0
        og = ol.OGrid(gx, gy, res)
        lx, ly = og.g.len_x, og.g.len_y
        hix, hiy = og.g.pos_x + lx / 2, ly / 2
        ox = float(st.uniform(1, hix - lx - 0.05 * lx, hix + 0.05 * lx)[0])
        oy = float(st.uniform(1, hiy - ly - 0.05 * ly, hiy + 0.05 * ly)[0])
        tfs = synth.transforms(case % 2 == 0)
        tfs["base_lidar"] = np.array([0.0, 0.0, 0.0, 1.0, ox, oy, 1.8])
        h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
        n = [50, 2_000, 40_000, 200_000][case % 4]
        kind = case % 3
        if kind == 0:      # uniform over the map and 10 % beyond
            x = st.uniform(n, hix - 1.1 * lx - ox, hix + 0.1 * lx - ox)
            y = st.uniform(n, hiy - 1.1 * ly - oy, hiy + 0.1 * ly - oy)
        elif kind == 1:    # a 90 degree fan of ranges: most sectors see nothing
            r = st.uniform(n, 0.5, 1.2 * max(lx, ly)).astype(np.float64)
            az = st.uniform(n, 0.3, 0.3 + np.pi / 2).astype(np.float64)
            x = (r * np.cos(az)).astype(np.float32); y = (r * np.sin(az)).astype(np.float32)
        else:              # two walls: ends share columns / rows (slopes 0 and infinity dominate)
            half = n // 2
            x = np.concatenate([st.uniform(half, -0.4 * lx, 0.4 * lx), np.full(n - half, 0.31 * lx, np.float32)]).astype(np.float32)
            y = np.concatenate([np.full(half, 0.27 * ly, np.float32), st.uniform(n - half, -0.4 * ly, 0.4 * ly)]).astype(np.float32)
        z = st.uniform(n, -1.0, 1.0)
        h.upload_xyz(x, y, z)
        h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS)
        hits, cell, miss, _, _ = oracle_frame(og, tfs, x, y, z)
        assert np.array_equal(h.hits(), hits), case
        assert np.array_equal(h.miss(), miss.astype(np.int32)), (case, gx, gy, res, ox, oy, n, kind)
        nlo, _, _ = check_grid(h, og)
        assert nlo == 0, case
        h.close()


@pytest.mark.parametrize("grid", [(50, 20, 0.1), (100, 100, 0.5), (120, 60, 0.3), (200, 200, 0.05)])
def test_cell_index_on_cell_boundaries(gvamd, grid):
    """The points pass derives (int)(-(d / res)) from a reciprocal-multiply estimate and only divides
    when the estimate is within 1e-6 of an integer.  Points ON cell boundaries (base-frame x, y =
    k * res as fp32, and their fp32 neighbours) are exactly those cases; SURVEY 8(c)'s canary
    (getIndex(40.7) = 2, not 3, on the default grid) is among them.  cell_idx vs the oracle."""
    gx, gy, res = grid
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    tfs = synth.transforms(False)
    tfs["base_lidar"] = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])   # identity: lidar frame = base frame
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    lx, ly = og.g.len_x, og.g.len_y
    hix, hiy = og.g.pos_x + lx / 2, ly / 2
    kx = np.arange(og.nx + 1, dtype=np.float64)
    ky = np.arange(og.ny + 1, dtype=np.float64)
    bx = (hix - kx * res).astype(np.float32)            # every cell boundary along x, as fp32
    by = (hiy - ky * res).astype(np.float32)
    xs, ys = [], []
    for d in (0, 1, -1, 2, -2):                        # the boundary and its fp32 neighbours
        nbx = bx.copy(); nby = by.copy()
        for _ in range(abs(d)):
            nbx = np.nextafter(nbx, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
            nby = np.nextafter(nby, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
        xs.append(np.repeat(nbx, 3)); ys.append(np.tile(np.float32([0.0, by[1], by[len(by) // 2]]), len(nbx)))
        ys.append(np.repeat(nby, 3)); xs.append(np.tile(np.float32([og.g.pos_x, bx[1], bx[len(bx) // 2]]), len(nby)))
    x = np.concatenate(xs + [np.float32([40.7, 40.8, 40.9, 41.0, -9.0])])
    y = np.concatenate(ys + [np.float32([0.0, 0.0, 0.0, 0.0, 0.0])])
    z = np.zeros_like(x)
    h.upload_xyz(x, y, z)
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_KEEP_CELL_IDX | gvamd.FRAME_KEEP_COUNTS)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    hits, cell = og.bin_points(m_base, x, y, z)
    got = h.cell_idx()
    assert np.array_equal(got, cell), np.flatnonzero(got != cell)[:10]
    assert np.array_equal(h.hits(), hits)
    assert (cell >= 0).sum() > len(x) // 2
    h.close()


def test_projection_on_float_rounding_boundaries(gvamd):
    """The points pass forms (float)(n / iz) from a reciprocal-multiply estimate and only divides when
    the estimate is within 2^-45 of a float rounding boundary.  tests/golden/projection_canaries.npz
    holds camera-frame points whose exact fp64 quotient lies within 6 ulp of such a boundary (found by
    tests/golden/make_projection_canaries.py).  Each point gets a bbox that is exactly one pixel value
    wide in u and v: it is inside iff both projections round like the reference's division."""
    d = np.load(os.path.join(HERE, "golden", "projection_canaries.npz"))
    x, y, z = d["x"], d["y"], d["z"]
    K = ol.set_intrinsic(synth.FX, synth.FY, synth.CX, synth.CY)
    u = (((synth.FX * x.astype(np.float64) + 0.0 * y.astype(np.float64)) + synth.CX * z.astype(np.float64)) / z.astype(np.float64)).astype(np.float32)
    v = (((0.0 * x.astype(np.float64) + synth.FY * y.astype(np.float64)) + synth.CY * z.astype(np.float64)) / z.astype(np.float64)).astype(np.float32)
    inside = (u >= 0) & (u < synth.IMG_W) & (v >= 0) & (v < synth.IMG_H)
    x, y, z, u, v = x[inside], y[inside], z[inside], u[inside], v[inside]
    assert len(x) >= 24
    n = min(len(x), 60)
    x, y, z, u, v = x[:n], y[:n], z[:n], u[:n], v[:n]
    bboxes = np.zeros(n, dtype=synth.BBOX_DTYPE)
    bboxes["x_min"] = u.astype(np.float64); bboxes["x_max"] = u.astype(np.float64)
    bboxes["y_min"] = v.astype(np.float64); bboxes["y_max"] = v.astype(np.float64)
    bboxes["confidence"] = 0.9
    want = ol.extract_cloud_per_bbox(K, x, y, z, bboxes, synth.IMG_W, synth.IMG_H)
    assert (want >= 0).all(), "the oracle itself must put every canary into a one-value-wide bbox"
    g = synth.CONFIGS[1]["grid"]
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    ident = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])          # camera frame = lidar frame: the transform is exact
    h.set_transforms(ident, ident, ident)
    # the same canaries many times over, so that every lane position and wavefront sees them
    reps = 40
    h.upload_xyz(np.tile(x, reps), np.tile(y, reps), np.tile(z, reps))
    ids, counts = h.extract_cloud_per_bbox(bboxes)
    assert np.array_equal(ids, np.tile(want, reps))
    h.close()


def test_handle_lifecycle_with_frames_in_flight(gvamd):
    """gv_destroy with frames still in flight on both lanes (no gv_synchronize before close), twenty
    times over: nothing hangs or faults, and device memory comes back (hipMemGetInfo before / after)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
        return fr.value

    config = 2
    x, y, z, _ = synth.cloud_lidar_like(config, 60_000)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    bboxes, poses = synth.detections(3, 20), synth.lshape_poses(config, 12)
    h, _ = make_handle(gvamd, config, perturbed=True)   # first handle: runtime pools and code objects get allocated
    h.upload_xyz(x, y, z)
    h.set_detections(flags, bboxes=bboxes, poses=poses)
    h.enqueue_frame()
    h.close()
    before = free_bytes()
    for it in range(20):
        h, _ = make_handle(gvamd, config, perturbed=True)
        h.upload_xyz(x, y, z)
        h.set_detections(flags, bboxes=bboxes, poses=poses)
        for _ in range(3 + it % 5):
            h.enqueue_frame()
        h.close()   # frames in flight
    after = free_bytes()
    assert before - after < (64 << 20), f"device memory not returned: {before - after} bytes"


def test_streaming_growth_reallocates_under_load(gvamd):
    """Clouds and detection sets that GROW while frames are in flight: every per-point buffer, binning scratch,
    detection block and rectangle list is reallocated mid-stream (the growth paths drain the lanes first).
    Grid, hit counts and bbox ids after the last frame must equal the oracle's."""
    config = 2
    g = synth.CONFIGS[config]["grid"]
    sizes = [20_000, 45_000, 45_000, 130_000, 60_000, 260_000]
    nbox = [10, 70, 70, 150, 30, 300]
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    h, tfs = make_handle(gvamd, config, perturbed=True)
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    pins = []
    last = None
    for f, (n, nb) in enumerate(zip(sizes, nbox)):
        gen = synth.cloud_lidar_like if f % 2 else synth.cloud_uniform
        x, y, z, _ = gen(config, n, seed_extra=40 + f)
        bboxes, poses = synth.detections(3, nb, seed_extra=f), synth.lshape_poses(config, nb // 2 + 1, seed_extra=f)
        p3 = tuple(gvamd.PinnedF32(n) for _ in range(3))
        p3[0].array[:], p3[1].array[:], p3[2].array[:] = x, y, z
        pins.append(p3)
        h.upload_xyz_async(p3[0].array, p3[1].array, p3[2].array)
        h.set_detections_async(flags, bboxes=bboxes, poses=poses)
        h.enqueue_frame()
        h.enqueue_frame()   # a second frame on the same inputs: both lanes busy when the next upload grows things
        for _ in range(2):
            last = oracle_frame(og, tfs, x, y, z, bboxes, poses)
    h.synchronize()
    hits, _, _, ids, _ = last
    assert np.array_equal(h.hits(), hits)
    assert np.array_equal(h.bbox_id(), ids)
    nlo, _, _ = check_grid(h, og)
    assert nlo == 0
    h.close()
    for p3 in pins:
        for p in p3:
            p.close()


@pytest.mark.parametrize("n", [1, 63, 511, 2047, 2048, 2049, 4097, 6145])
def test_partition_ragged_cloud_sizes(gvamd, n):
    """The partition pass takes four points per lane and step (chunks of 2048): cloud sizes around the chunk and
    step boundaries, with non-finite points, points behind the camera and points outside the map mixed in, every
    per-point output kept (KEEP_CELL_IDX: the <ray, bbox, cell ids> instance of the kernel) -- cell ids, bbox ids,
    hit counts and the free cells bit-exact against the oracle."""
    gx, gy, res = 120, 80, 0.25
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    tfs = synth.transforms(True)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    st = synth.Stream(977, n)
    x = st.uniform(n, -0.7 * gx, 0.7 * gx)
    y = st.uniform(n, -0.7 * gy, 0.7 * gy)
    z = st.uniform(n, -1.5, 1.5)
    for k, v in ((5, np.nan), (17, np.inf), (40, -np.inf)):   # scattered over x, y, z
        if n > k:
            (x, y, z)[k % 3][k] = np.float32(v)
    if n > 2050:
        x[2040:2050] = np.float32(1e30)        # far outside, across the chunk boundary: the clip path
        y[100:140] = np.float32(np.nan)        # a whole run of dropped points
    bboxes = synth.detections(3, 40)
    poses = synth.lshape_poses(1, 6)
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST | gvamd.FRAME_KEEP_CELL_IDX | gvamd.FRAME_KEEP_COUNTS
    h.upload_xyz(x, y, z)
    h.process_frame(flags, bboxes=bboxes, poses=poses)
    hits, cell, miss, ids, _ = oracle_frame(og, tfs, x, y, z, bboxes, poses)
    assert np.array_equal(h.cell_idx(), cell)
    assert np.array_equal(h.bbox_id(), ids)
    assert np.array_equal(h.hits(), hits)
    assert np.array_equal(h.miss(), miss)
    assert check_grid(h, og)[0] == 0
    h.close()


def _walls_scene(og, n, seed):
    """ends along two walls and a sprinkle elsewhere: the sectors next to the axes and the diagonals get the long
    tails (hundreds of columns beyond the threshold column) that the tail policies differ on"""
    st = synth.Stream(seed, n)
    lx, ly = og.g.len_x, og.g.len_y
    x = st.uniform(n, -0.49 * lx, 0.49 * lx)
    y = st.uniform(n, -0.49 * ly, 0.49 * ly)
    k = n // 3
    x[:k] = np.float32(0.47 * lx)              # a wall across the +x octants
    y[k:2 * k] = x[k:2 * k] * np.float32(ly / lx)   # a diagonal wall
    return x, y, st.uniform(n, -0.5, 0.5)


@pytest.mark.parametrize("grid,n", [((200, 200, 0.25), 30_000), ((250, 100, 0.1), 120_000)])
def test_sector_tail_policies_agree(gvamd, monkeypatch, grid, n):
    """Beyond the threshold column a sector either marches its long rays, evaluates every cell there exactly after
    looking at the long rays, or (short tail) evaluates them without looking: GV_FLAT_DIRECT / GV_MARCH_LIMIT /
    GV_FLAT_K move the choice, the free cells must not move -- all variants equal to the oracle's literal march."""
    gx, gy, res = grid
    og = ol.OGrid(gx, gy, res)
    tfs = synth.transforms(False)
    x, y, z = _walls_scene(og, n, 31337)
    m_base = ol.tf_to_matrix4f(tfs["base_lidar"])
    want, _ = og.raymarch(m_base, x, y, z)
    assert 0 < int(want.sum()) < og.G
    for fd, ml, fk in (("0", None, None), ("1000000000", None, None), (None, "0", None), (None, "100000000", "0"), ("300", "3000", "2")):
        for name, v in (("GV_FLAT_DIRECT", fd), ("GV_MARCH_LIMIT", ml), ("GV_FLAT_K", fk)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, v)
        h = gvamd.GridVisionHIP(gx, gy, res)
        h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
        h.upload_xyz(x, y, z)
        h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS)
        got = h.miss()
        h.close()
        assert np.array_equal(got, want), (fd, ml, fk, int(np.count_nonzero(got != want)))


@pytest.mark.parametrize("lanes", ["3", "2"])
def test_partition_overtakes_sector_tail(gvamd, monkeypatch, lanes):
    """With unchanged inputs the partition pass of frame f + 2 is launched without the barrier bit and starts while
    the sector kernel of frame f (same lane, other buffers) still runs.  Config-3 size so that the kernels really
    overlap: a run of frames on one cloud, new detections in the middle (their upload goes on a lane: those frames
    keep the barrier), a new cloud, more frames -- hit counts, free cells, bbox ids and the grid layers against the
    oracle after every batch, and the whole sequence once more with GV_ANYORDER=0 for equal layers."""
    config = 3
    g = synth.CONFIGS[config]["grid"]
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    clouds = [synth.cloud_uniform(config)[:3], synth.cloud_lidar_like(config)[:3]]
    dets = [(synth.detections(config), synth.lshape_poses(config)), (synth.detections(config, seed_extra=1), synth.lshape_poses(config, seed_extra=1))]
    plan = [(0, 0, 7), (0, 1, 5), (1, 1, 6), (1, 0, 1), (0, 0, 4)]   # (cloud, detection set, frames)
    layers = {}
    monkeypatch.setenv("GV_LANES", lanes)
    for mode in ("1", "0"):
        monkeypatch.setenv("GV_ANYORDER", mode)
        h, tfs = make_handle(gvamd, config, True)
        og = ol.OGrid(g.grid_x, g.grid_y, g.resolution) if mode == "1" else None
        cur_cloud = None
        for ci, di, nf in plan:
            if ci != cur_cloud:
                h.upload_xyz(*clouds[ci])
                cur_cloud = ci
            bb, pp = dets[di]
            h.set_detections(flags, bboxes=bb, poses=pp)
            for _ in range(nf):
                h.enqueue_frame()
            h.synchronize()
            if og is not None:
                x, y, z = clouds[ci]
                for _ in range(nf):
                    hits, _, miss, ids, _ = oracle_frame(og, tfs, x, y, z, bb, pp)
                assert np.array_equal(h.hits(), hits)
                assert np.array_equal(h.miss(), miss)
                assert np.array_equal(h.bbox_id(), ids)
                assert check_grid(h, og)[0] == 0
        layers[mode] = (h.log_odds().copy(), h.hits().copy(), h.miss().copy())
        h.close()
    for a, b in zip(layers["1"], layers["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("seed,lanes", [(11, "3"), (29, "3"), (29, "2")])
def test_random_call_sequences_with_frames_in_flight(gvamd, monkeypatch, seed, lanes):
    """A caller that does things in no particular order: runs of frames on unchanged inputs (their partition passes
    start inside the previous sector kernel), new clouds through the asynchronous upload, new detections through
    the asynchronous form, both at once, host waits at random points.  300 k points on a 1200 x 1200 grid: long
    enough kernels for real overlap, short enough for ~60 oracle frames.  Whatever the interleaving, the grid
    after every wait equals the oracle's and the last frame's per-point outputs are that frame's."""
    monkeypatch.setenv("GV_LANES", lanes)
    gx, gy, res = 240, 240, 0.2
    n = 300_000
    rng = np.random.default_rng(seed)
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    tfs = synth.transforms(True)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    clouds, pins = [], []
    for c in range(3):
        st = synth.Stream(seed * 100 + c, n)
        spread = (0.6, 0.35, 0.8)[c]
        x = st.uniform(n, -spread * gx, spread * gx)
        y = st.uniform(n, -spread * gy, spread * gy)
        z = st.uniform(n, -1.0, 1.0)
        p3 = tuple(gvamd.PinnedF32(n) for _ in range(3))
        for p, a in zip(p3, (x, y, z)):
            p.array[:] = a
        pins.append(p3)
        clouds.append((x, y, z))
    dets = [(synth.detections(3, 12 + 9 * d, seed_extra=d), synth.lshape_poses(1, 5 + 4 * d, seed_extra=d)) for d in range(3)]
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    ci, di = 0, 0
    h.upload_xyz_async(*(p.array for p in pins[ci]))
    h.set_detections_async(flags, bboxes=dets[di][0], poses=dets[di][1])
    pending = []          # (cloud, detection set) of the frames enqueued since the last check
    total = 0
    while total < 60:
        op = rng.integers(0, 10)
        if op <= 4:       # a run of frames on whatever is current
            for _ in range(int(rng.integers(1, 6))):
                h.enqueue_frame()
                pending.append((ci, di))
                total += 1
        elif op == 5:
            ci = int(rng.integers(0, 3))
            h.upload_xyz_async(*(p.array for p in pins[ci]))
        elif op == 6:
            di = int(rng.integers(0, 3))
            h.set_detections_async(flags, bboxes=dets[di][0], poses=dets[di][1])
        elif op == 7:     # both, then one frame right behind
            ci, di = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            h.upload_xyz_async(*(p.array for p in pins[ci]))
            h.set_detections_async(flags, bboxes=dets[di][0], poses=dets[di][1])
            h.enqueue_frame()
            pending.append((ci, di))
            total += 1
        elif pending:     # host wait + check
            h.synchronize()
            for c, d in pending:
                hits, _, miss, ids, _ = oracle_frame(og, tfs, *clouds[c], dets[d][0], dets[d][1])
            pending = []
            assert np.array_equal(h.hits(), hits)
            assert np.array_equal(h.miss(), miss)
            assert np.array_equal(h.bbox_id(), ids)
            assert check_grid(h, og)[0] == 0
    h.synchronize()
    for c, d in pending:
        oracle_frame(og, tfs, *clouds[c], dets[d][0], dets[d][1])
    assert check_grid(h, og)[0] == 0
    h.close()
    for p3 in pins:
        for p in p3:
            p.close()


@pytest.mark.parametrize("extra", [0, 30_000, 400_000])
def test_sector_bucket_rows_overflow_on_shared_slopes(gvamd, extra):
    """Ends on exact rational slopes through the sensor cell (0, 1/3, 1/2, 2/3, 1, and their mirror images in all
    eight octants): hundreds of ends share one slope bucket, far more than a bucket's row of eight holds, so the
    row-mode walks have to go through the overflow list; with 30 k scattered points on top the wedges stay in row
    mode, with 400 k they switch to counting-sort placement.  Free cells against the literal march."""
    gx, gy, res = 200, 200, 0.2          # 1000 x 1000 cells
    h = gvamd.GridVisionHIP(gx, gy, res)
    og = ol.OGrid(gx, gy, res)
    tfs = synth.transforms(False)
    tfs["base_lidar"] = np.array([0.0, 0.0, 0.0, 1.0, float(og.g.pos_x) + 0.1, 0.1, 1.8])   # the middle of a cell near the centre
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    xs, ys = [], []
    steps = np.arange(1, 480, dtype=np.float64)
    for p, q in ((0, 1), (1, 3), (1, 2), (2, 3), (1, 1)):
        for sx in (1, -1):
            for sy in (1, -1):
                for swap in (False, True):
                    a = steps * q * res
                    b = steps * p * res
                    keep = (a < 0.49 * gx) & (b < 0.49 * gy)
                    a, b = a[keep][::2], b[keep][::2]          # every second lattice point of the line
                    xs.append(sx * (b if swap else a))
                    ys.append(sy * (a if swap else b))
    x = np.concatenate(xs).astype(np.float32)
    y = np.concatenate(ys).astype(np.float32)
    if extra:
        st = synth.Stream(4711, extra)
        x = np.concatenate([x, st.uniform(extra, -0.48 * gx, 0.48 * gx)])
        y = np.concatenate([y, st.uniform(extra, -0.48 * gy, 0.48 * gy)])
    z = np.zeros(len(x), np.float32)
    h.upload_xyz(x, y, z)
    h.process_frame(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_KEEP_COUNTS)
    hits, _, miss, _, _ = oracle_frame(og, tfs, x, y, z)
    assert int(miss.sum()) > 1000
    assert np.array_equal(h.hits(), hits)
    assert np.array_equal(h.miss(), miss.astype(np.int32)), int(np.count_nonzero(h.miss() != miss))
    h.close()


@pytest.mark.parametrize("extra", [0, 1, 2])
def test_upload_stream_gets_a_queue_of_its_own(gvamd, monkeypatch, capfd, extra):
    """A process gets four hardware queues; with other streams alive (what a host application or a framework owns) one
    of the handle's four streams has to share, and when that is the upload stream every cloud waits behind kernels
    (profiles/r03/h2d_notes.md 6).  gv_create probes for it and replaces the upload stream until a 4-byte memset no
    longer waits for the compute streams; GV_VERBOSE prints the last probe.  The streamed frames stay correct."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    others = []
    for _ in range(extra):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        others.append(s)
    monkeypatch.setenv("GV_VERBOSE", "1")
    config = 2
    g = synth.CONFIGS[config]["grid"]
    h, tfs = make_handle(gvamd, config, perturbed=False)
    err = capfd.readouterr().err
    line = [l for l in err.splitlines() if "upload stream probe" in l]
    assert line, err
    waited_us = float(line[-1].split("probe")[1].split("us")[0])
    assert waited_us < 90.0, line[-1]          # the idle kernels on the compute streams last 150 us
    og = ol.OGrid(g.grid_x, g.grid_y, g.resolution)
    x, y, z, _ = synth.cloud_uniform(config, 50_000)
    pins = [gvamd.PinnedF32(len(x)) for _ in range(3)]
    for p, a in zip(pins, (x, y, z)):
        p.array[:] = a
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH
    h.set_detections(flags)
    for _ in range(5):
        h.upload_xyz_async(pins[0].array, pins[1].array, pins[2].array)
        h.enqueue_frame()
    h.synchronize()
    for _ in range(5):
        hits, _, miss, _, _ = oracle_frame(og, tfs, x, y, z)
    assert np.array_equal(h.hits(), hits) and np.array_equal(h.miss(), miss)
    assert check_grid(h, og)[0] == 0
    h.close()
    for p in pins:
        p.close()
    for s in others:
        hip.hipStreamDestroy(s)
