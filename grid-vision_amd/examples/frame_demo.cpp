// frame_demo.cpp -- the reference's timerCallback data flow (src/grid_vision_node.cpp:108-244)
// written against the C++ mirror headers: plain g++ host code over the C ABI.
//   g++ -std=c++17 -O2 frame_demo.cpp -o frame_demo -L.. -lgridvision_hip -Wl,-rpath,$PWD/..
// Prints a few checksums; tests/test_gpu_parity.py::test_cpp_demo compares them with python.
#include <cinttypes>
#include <cstdio>
#include <vector>

#include "../include/grid_vision/hip_backend.hpp"

static uint64_t sm64(uint64_t &s)
{
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float u01(uint64_t &s) { return (float)(sm64(s) >> 40) * (1.0f / 16777216.0f); }

int main()
{
  try {
    const CAMParams cam{224, 224, 480, 640, 320.f, 320.f, 320.f, 240.f};   // config/grid_vision_cfg.yaml
    GridVisionContext ctx(50, 20, 0.1, cam);                                // grid_x, grid_y, resolution
    OccupancyGridMap occ_grid(ctx);
    const gv_transform cam_lidar{0.5, -0.5, 0.5, 0.5, 0.0, 0.4, -0.3};
    const gv_transform base_cam{0.5, -0.5, 0.5, -0.5, 0.3, 0.0, 2.2};
    const gv_transform base_lidar{0, 0, 0, 1, 0, 0, 1.8};
    ctx.setTransforms(&cam_lidar, &base_cam, &base_lidar);

    // cloudCallback: a seeded cloud
    uint64_t seed = 42;
    const size_t n = 50000;
    std::vector<float> x(n), y(n), z(n);
    for (size_t i = 0; i < n; ++i) { x[i] = -12.f + 56.f * u01(seed); y[i] = -12.f + 24.f * u01(seed); z[i] = -2.f + 4.f * u01(seed); }
    ctx.setCloud(x.data(), y.data(), z.data(), n);

    // precomputed detections
    std::vector<BoundingBox> bboxes = {{100, 150, 220, 300, 0.95f, 9}, {300, 200, 380, 330, 0.9f, 2},
                                       {420, 100, 470, 160, 0.8f, 5}, {500, 250, 600, 400, 0.7f, 0}};
    auto [static_bboxes, dynamic_bboxes] = object_detection::filterBBoxes(bboxes);

    // static branch (:168-184)
    std::vector<float> depth_vec = cloud_detections::computeDepthForBoundingBoxes(ctx, static_bboxes, 4);
    auto cam_points = cloud_detections::convertPixelsTo3D(ctx, static_bboxes, depth_vec);

    // dynamic branch, use_vision_orientation == true (:190-209), with made-up network outputs
    std::vector<float> orient(dynamic_bboxes.size() * 4), conf(dynamic_bboxes.size() * 2), dims(dynamic_bboxes.size() * 3, 0.1f);
    for (size_t i = 0; i < dynamic_bboxes.size(); ++i) {
      orient[4 * i] = 0.8f; orient[4 * i + 1] = 0.6f; orient[4 * i + 2] = -0.6f; orient[4 * i + 3] = 0.8f;
      conf[2 * i] = 0.3f; conf[2 * i + 1] = 0.7f;
    }
    VisionOrientation vision_orient(ctx);
    std::vector<LShapePose> bboxes_pose = vision_orient.postProcessOutputs(orient.data(), conf.data(), dims.data(), dynamic_bboxes);
    vision_orient.transformLShapeObjects(bboxes_pose);
    occ_grid.updateMap(bboxes_pose);

    // dynamic branch, PCA path (:210-231)
    std::vector<LShapePose> pca_pose = cloud_detections::computeBBoxPose(ctx, bboxes, /*remove_ground=*/false);
    vision_orient.transformLShapeObjects(pca_pose);
    occ_grid.updateMap(pca_pose);
    occ_grid.updateMap();

    // [EXTENSION] fused frame: bin + ray-march + bbox test + grid pass
    gv_frame_desc d{};
    d.flags = GV_FRAME_BIN | GV_FRAME_RAYMARCH | GV_FRAME_BBOX_TEST | GV_FRAME_KEEP_COUNTS;
    d.bboxes = bboxes.data();
    d.n_bboxes = (int32_t)bboxes.size();
    d.poses = bboxes_pose.data();
    d.n_poses = (int32_t)bboxes_pose.size();
    gv::check(gv_process_frame(ctx.handle(), &d), ctx.handle(), "gv_process_frame");

    gv_grid_info info{};
    std::vector<int8_t> grid = occ_grid.toOccupancyGrid(&info);   // publishOccupancyGrid (:265-278)
    std::vector<int32_t> hits(occ_grid.cells());
    gv::check(gv_get_hits(ctx.handle(), hits.data()), ctx.handle(), "gv_get_hits");
    long long sum_i8 = 0, sum_hits = 0;
    for (int8_t v : grid) sum_i8 += v;
    for (int32_t v : hits) sum_hits += v;
    std::printf("grid %ux%u res %.3f origin (%.3f, %.3f)\n", info.width, info.height, info.resolution, info.origin_x, info.origin_y);
    std::printf("static %zu dynamic %zu poses %zu pca_poses %zu\n", static_bboxes.size(), dynamic_bboxes.size(), bboxes_pose.size(), pca_pose.size());
    std::printf("depth0 %.6f cam_point0 %.6f %.6f %.6f\n", depth_vec.empty() ? -1.f : depth_vec[0],
                cam_points.empty() ? 0. : cam_points[0].x, cam_points.empty() ? 0. : cam_points[0].y, cam_points.empty() ? 0. : cam_points[0].z);
    std::printf("sum_i8 %lld sum_hits %lld\n", sum_i8, sum_hits);
    return 0;
  } catch (const gv::Error &e) {
    std::fprintf(stderr, "gv error %d: %s\n", e.code, e.what());
    return 2;
  }
}
