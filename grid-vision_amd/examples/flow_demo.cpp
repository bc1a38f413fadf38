// flow_demo.cpp -- drives grid_vision::FrameFlow (the reference's timerCallback decision flow,
// src/grid_vision_node.cpp:108-244) through every branch, in plain g++ host code over the C ABI:
//   g++ -std=c++17 -O2 flow_demo.cpp -o flow_demo -L.. -lgridvision_hip -Wl,-rpath,$PWD/..
// Prints one line per tick; tests/test_gpu_parity.py::test_cpp_flow_demo replays the same ticks through the
// ctypes binding and compares branch, counts and grid checksums.
#include <cinttypes>
#include <cstdio>
#include <vector>

#include "../include/grid_vision/frame_flow.hpp"

static uint64_t sm64(uint64_t &s)
{
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float u01(uint64_t &s) { return (float)(sm64(s) >> 40) * (1.0f / 16777216.0f); }

int main(int argc, char **argv)
{
  using namespace grid_vision;
  // "0": the reference's call-by-call sequence (one synchronous call per reference function); default: the fused
  // tick (gv_tick_*: one batch of device work, one host wait).  Both must print the same lines.
  const bool fused = !(argc > 1 && argv[1][0] == '0');
  try {
    const CAMParams cam{224, 224, 480, 640, 320.f, 320.f, 320.f, 240.f};   // config/grid_vision_cfg.yaml
    GridVisionContext ctx(50, 20, 0.1, cam);
    OccupancyGridMap occ_grid(ctx);
    const gv_transform cam_lidar{0.5, -0.5, 0.5, 0.5, 0.0, 0.4, -0.3};
    const gv_transform base_cam{0.5, -0.5, 0.5, -0.5, 0.3, 0.0, 2.2};
    const gv_transform base_lidar{0, 0, 0, 1, 0, 0, 1.8};

    // cloudCallback: a seeded cloud with a ground plane (lidar z = -1.7) and two dense blobs
    uint64_t seed = 7;
    const size_t n = 60000;
    std::vector<float> x(n), y(n), z(n);
    for (size_t i = 0; i < n; ++i) {
      const float a = u01(seed), b = u01(seed), c = u01(seed);
      if (i < 30000) { x[i] = 2.f + 38.f * a; y[i] = -9.f + 18.f * b; z[i] = -1.7f + 0.02f * (c - 0.5f); }
      else if (i < 40000) { x[i] = 11.f + 2.4f * a; y[i] = -2.6f + 1.2f * b; z[i] = -1.2f + 1.3f * c; }
      else if (i < 50000) { x[i] = 17.f + 1.0f * a; y[i] = 2.5f + 3.0f * b; z[i] = -1.2f + 1.3f * c; }
      else { x[i] = -8.f + 48.f * a; y[i] = -9.5f + 19.f * b; z[i] = -1.5f + 4.f * c; }
    }
    ctx.setCloud(x.data(), y.data(), z.data(), n);

    const std::vector<BoundingBox> full = {{330, 200, 460, 330, 0.95f, 9}, {150, 180, 280, 330, 0.9f, 2},
                                           {420, 100, 470, 160, 0.8f, 5}, {40, 60, 100, 120, 0.7f, 7}};
    const std::vector<BoundingBox> only_static = {{420, 100, 470, 160, 0.8f, 5}, {40, 60, 100, 120, 0.7f, 7}};
    const std::vector<BoundingBox> none;
    auto net = [](const std::vector<BoundingBox> &dyn, std::vector<float> &orient, std::vector<float> &conf, std::vector<float> &dims) {
      orient.assign(dyn.size() * 4, 0.f); conf.assign(dyn.size() * 2, 0.f); dims.assign(dyn.size() * 3, 0.1f);
      for (size_t i = 0; i < dyn.size(); ++i) {
        orient[4 * i] = 0.8f; orient[4 * i + 1] = 0.6f; orient[4 * i + 2] = -0.6f; orient[4 * i + 3] = 0.8f;
        conf[2 * i] = 0.3f; conf[2 * i + 1] = 0.7f;
      }
    };

    auto report = [&](int tick, const TickResult &r) {
      gv_grid_info info{};
      std::vector<int8_t> grid = occ_grid.toOccupancyGrid(&info);   // publishOccupancyGrid (:265-278)
      long long sum = 0;
      for (int8_t v : grid) sum += v;
      std::printf("tick %d branch %s bboxes %zu static %zu dynamic %zu depths %zu poses %zu publish_detections %d sum_i8 %lld depth0 %.6f\n",
                  tick, branch_name(r.branch), r.bboxes.size(), r.static_bboxes.size(), r.dynamic_bboxes.size(), r.depth_vec.size(),
                  r.bboxes_pose.size(), r.publish_detections ? 1 : 0, sum, r.depth_vec.empty() ? -1.f : r.depth_vec[0]);
    };

    FlowParams pv;                        // yaml defaults: vision orientation
    pv.fused = fused;
    FlowParams pp = pv;
    pp.use_vision_orientation = false;    // PCA branch
    FrameFlow flow_v(ctx, occ_grid, pv), flow_p(ctx, occ_grid, pp);
    int tick = 0;
    TickInput in;
    // 1. neither image nor cloud yet
    report(tick++, flow_v.tick(in));
    // 2. inputs present, the detector found nothing
    in.have_image = in.have_cloud = true;
    in.image_w = 640; in.image_h = 480;
    in.bboxes = &none;
    report(tick++, flow_v.tick(in));
    // 3. detections, but the tf lookup fails
    in.bboxes = &full;
    in.orientation_net = net;
    report(tick++, flow_v.tick(in));
    // 4. transforms known: vision-orientation branch
    ctx.setTransforms(&cam_lidar, &base_cam, &base_lidar);
    flow_v.setTransformsAvailable(true);
    flow_p.setTransformsAvailable(true);
    report(tick++, flow_v.tick(in));
    // 5. the same tick through the PCA branch (segmentGroundPlane -> extractCloudPerBBox -> radius filter -> PCA)
    report(tick++, flow_p.tick(in));
    // 6. only static detections
    in.bboxes = &only_static;
    report(tick++, flow_v.tick(in));
    return 0;
  } catch (const gv::Error &e) {
    std::fprintf(stderr, "gv error %d: %s\n", e.code, e.what());
    return 2;
  }
}
