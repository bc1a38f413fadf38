// grid_vision/frame_flow.hpp -- the decision flow of GridVision::timerCallback
// (src/grid_vision_node.cpp:108-244) without ROS: guards, static / dynamic split, depth -> 3-D points,
// vision-orientation versus PCA branch and the choice of updateMap overload, written once over the C++
// mirror (hip_backend.hpp).  What is left to a ROS2 node is message and tf conversion: it fills a
// TickInput from its subscriptions, calls tick(), and publishes what TickResult says (ros2/src/).
// Compiled and run without ROS by examples/flow_demo.cpp (tests/test_gpu_parity.py::test_cpp_flow_demo).
#pragma once

#include <functional>
#include <string>
#include <vector>

#include "hip_backend.hpp"

namespace grid_vision {

// config/grid_vision_cfg.yaml: the parameters the timer callback reads
struct FlowParams {
  double conf_threshold = 0.6;          // confidence_threshold (:14)
  double iou_threshold = 0.6;           // iou_threshold (:15)
  int resize = 416;                     // network input of the 2-D detector (yolov4_3l_416_416)
  uint16_t k_near = 4;                  // k_near (:20): neighbours of the depth estimate
  bool use_vision_orientation = true;   // use_vision_orientation (:24): orientation network or cloud PCA
  // [EXTENSION] (ros2/config: lidar_binning, lidar_raymarch) the map update of a tick is one fused frame -- decay,
  // rectangles, per-cell lidar hit counts and optionally the free-space ray stage -- instead of updateMap alone
  bool lidar_binning = false, lidar_raymarch = false;
  // true (default): everything from filterBBoxes to the packed grid is ONE batch of device work with one host wait
  // (gv_tick_*): the poses go from the kernel that computes them into the grid pass without leaving the device.
  // false: the reference's call-by-call sequence over the mirror (one synchronous call per reference function).
  bool fused = true;
};

// what one tick of the 50 ms timer (grid_vision_node.cpp:49-50) has to work with
struct TickInput {
  bool have_image = false;      // init_image_ not empty
  bool have_cloud = false;      // cloud_ not empty (already uploaded to the context by the cloud callback)
  int image_w = 0, image_h = 0;
  // raw outputs of the 2-D detector (run_inference, :124-125): boxes[n, 4], scores[n, classes] ...
  const float *det_boxes = nullptr, *det_scores = nullptr;
  int n_det = 0, n_classes = 0;
  // ... or boxes somebody already extracted (then the two arrays above are ignored)
  const std::vector<BoundingBox> *bboxes = nullptr;
  // the orientation network (VisionOrientation::runInference, :193): called with the dynamic boxes, fills
  // orient[n * 4], conf[n * 2], dims[n * 3]; only used when use_vision_orientation
  std::function<void(const std::vector<BoundingBox> &, std::vector<float> &, std::vector<float> &, std::vector<float> &)> orientation_net;
};

enum class TickBranch {
  MissingInputs,     // :111-116  grid published as it is
  NoDetections,      // :141-147  plain updateMap, grid published
  NoTransform,       // :160-164  tf lookup failed: grid published as it is
  VisionOrientation, // :190-209
  CloudPCA,          // :210-231
  NoDynamicObjects   // :233-236  plain updateMap
};

struct TickResult {
  TickBranch branch = TickBranch::MissingInputs;
  bool publish_grid = true;             // every path of the reference ends in publishOccupancyGrid
  bool publish_detections = false;      // :239, :242: only the full path publishes boxes and markers
  std::vector<BoundingBox> bboxes, static_bboxes, dynamic_bboxes;
  std::vector<float> depth_vec;                 // depth of every static box (:176-177)
  std::vector<geometry::Point> cam_points;      // their 3-D points in the base frame (:180)
  std::vector<LShapePose> bboxes_pose;          // dynamic objects in the base frame
  std::string warning;
};

class FrameFlow {
public:
  FrameFlow(GridVisionContext &ctx, OccupancyGridMap &grid, const FlowParams &p) : ctx_(ctx), grid_(grid), vision_(ctx), p_(p) {}

  // base <- camera and camera <- lidar known (transformLidarToCamera's lookup, :280-307)
  void setTransformsAvailable(bool ok) { have_tf_ = ok; }

  TickResult tick(const TickInput &in)
  {
    TickResult r;
    // :111 -- the reference tests `init_image_.empty() && cloud_.empty()` (both missing); kept as it is
    if (!in.have_image && !in.have_cloud) {
      r.branch = TickBranch::MissingInputs;
      r.warning = "Image or Pointcloud is missing in GridVision";
      return r;
    }
    // :138-139 extract_bboxes
    if (in.bboxes) r.bboxes = *in.bboxes;
    else if (in.n_det > 0)
      r.bboxes = object_detection::extract_bboxes(in.det_boxes, in.det_scores, in.n_det, in.n_classes, p_.conf_threshold,
                                                  p_.iou_threshold, in.image_w, in.image_h, p_.resize);
    if (r.bboxes.empty()) {   // :141-147
      update(r.bboxes_pose);
      r.branch = TickBranch::NoDetections;
      return r;
    }
    std::tie(r.static_bboxes, r.dynamic_bboxes) = object_detection::filterBBoxes(r.bboxes);   // :152
    if (!have_tf_) {   // :156-164 transformLidarToCamera returned nullptr
      r.branch = TickBranch::NoTransform;
      return r;
    }
    if (p_.fused && fusable()) return tickFused(in, r);
    if (!r.static_bboxes.empty()) {   // :168-184
      r.depth_vec = cloud_detections::computeDepthForBoundingBoxes(ctx_, r.static_bboxes, p_.k_near);
      r.cam_points = cloud_detections::convertPixelsTo3D(ctx_, r.static_bboxes, r.depth_vec);
    }
    if (!r.dynamic_bboxes.empty()) {
      if (p_.use_vision_orientation) {   // :190-209
        std::vector<float> orient, conf, dims;
        if (in.orientation_net) in.orientation_net(r.dynamic_bboxes, orient, conf, dims);
        if (orient.size() == r.dynamic_bboxes.size() * 4 && conf.size() == r.dynamic_bboxes.size() * 2 &&
            dims.size() == r.dynamic_bboxes.size() * 3)
          r.bboxes_pose = vision_.postProcessOutputs(orient.data(), conf.data(), dims.data(), r.dynamic_bboxes);
        vision_.transformLShapeObjects(r.bboxes_pose);
        update(r.bboxes_pose, true);
        r.branch = TickBranch::VisionOrientation;
      } else {   // :210-231 -- the reference hands ALL boxes to computeBBoxPose here, not only the dynamic ones
        r.bboxes_pose = cloud_detections::computeBBoxPose(ctx_, r.bboxes, /*remove_ground=*/true);
        vision_.transformLShapeObjects(r.bboxes_pose);
        update(r.bboxes_pose, true);
        r.branch = TickBranch::CloudPCA;
      }
    } else {   // :233-236
      update(r.bboxes_pose);
      r.branch = TickBranch::NoDynamicObjects;
    }
    r.publish_detections = true;   // :239-243
    return r;
  }

  // pinned landing place of OccupancyGrid.data (gv_host_alloc, G bytes): the fused tick copies the packed grid there
  // behind its grid pass; nullptr (default): the caller fetches it with toOccupancyGrid as before
  void setGridOut(int8_t *pinned) { grid_out_ = pinned; }

private:
  // the fused form needs the tile path when the lidar extension is on (gv_tick_enqueue says so too)
  bool fusable() const
  {
    if (!p_.lidar_binning) return true;
    int32_t nx = 0, ny = 0;
    gv_grid_geometry(ctx_.handle(), &nx, &ny, nullptr, nullptr);
    return nx % 4 == 0 && nx <= 8000 && ny <= 8000;
  }

  // :166-236 as one batch: same branches, same outputs, one host wait (GridVisionContext::tickWait)
  TickResult tickFused(const TickInput &in, TickResult &r)
  {
    std::vector<float> orient, conf, dims;
    uint32_t flags = 0;
    int32_t n_net = 0;
    const bool dyn = !r.dynamic_bboxes.empty();
    if (p_.use_vision_orientation) {
      flags |= GV_TICK_VISION_ORIENT;
      if (dyn && in.orientation_net) in.orientation_net(r.dynamic_bboxes, orient, conf, dims);   // :193
      if (dyn && orient.size() == r.dynamic_bboxes.size() * 4 && conf.size() == r.dynamic_bboxes.size() * 2 &&
          dims.size() == r.dynamic_bboxes.size() * 3)
        n_net = (int32_t)r.dynamic_bboxes.size();
    }
    if (p_.lidar_binning && ctx_.cloudSize() > 0) flags |= GV_TICK_LIDAR_BIN | (p_.lidar_raymarch ? GV_TICK_LIDAR_RAYMARCH : 0u);
    ctx_.tickEnqueue(r.bboxes, flags, p_.k_near, n_net ? orient.data() : nullptr, n_net ? conf.data() : nullptr,
                     n_net ? dims.data() : nullptr, n_net, grid_out_);
    GridVisionContext::TickOutput t = ctx_.tickWait();
    r.depth_vec = std::move(t.depths);
    r.cam_points = std::move(t.base_points);
    r.bboxes_pose = std::move(t.poses);
    r.branch = !dyn ? TickBranch::NoDynamicObjects : (p_.use_vision_orientation ? TickBranch::VisionOrientation : TickBranch::CloudPCA);
    r.publish_detections = true;   // :239-243
    return r;
  }

  // updateMap(grid) / updateMap(grid, poses) of the reference (the poses overload is called even with an empty
  // vector, :206,:230), or the fused frame when the lidar extension is on and a cloud is resident
  void update(const std::vector<LShapePose> &poses, bool poses_overload = false)
  {
    if (p_.lidar_binning && ctx_.cloudSize() > 0 && have_tf_) {
      gv_frame_desc d{};
      d.flags = GV_FRAME_BIN | (p_.lidar_raymarch ? GV_FRAME_RAYMARCH : 0u);
      d.poses = poses.data();
      d.n_poses = (int32_t)poses.size();
      gv::check(gv_process_frame(ctx_.handle(), &d), ctx_.handle(), "gv_process_frame");
    } else if (poses_overload) {
      grid_.updateMap(poses);
    } else {
      grid_.updateMap();
    }
  }

  GridVisionContext &ctx_;
  OccupancyGridMap &grid_;
  VisionOrientation vision_;
  FlowParams p_;
  bool have_tf_ = false;
  int8_t *grid_out_ = nullptr;
};

inline const char *branch_name(TickBranch b)
{
  switch (b) {
  case TickBranch::MissingInputs: return "missing_inputs";
  case TickBranch::NoDetections: return "no_detections";
  case TickBranch::NoTransform: return "no_transform";
  case TickBranch::VisionOrientation: return "vision_orientation";
  case TickBranch::CloudPCA: return "cloud_pca";
  case TickBranch::NoDynamicObjects: return "no_dynamic_objects";
  }
  return "?";
}

}  // namespace grid_vision
