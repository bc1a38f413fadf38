// grid_vision/hip_backend.hpp -- C++ host-side mirror of the reference's per-frame
// interface, implemented over the C ABI of libgridvision_hip.so (include/gridvision_hip.h).
//
// Same names, argument meaning and error behaviour as the reference classes/namespaces:
//   OccupancyGridMap            include/grid_vision/occupancy_grid.hpp:13-40
//   namespace cloud_detections  include/grid_vision/cloud_detections.hpp:27-56
//   namespace object_detection  include/grid_vision/object_detection.hpp:34-67 (post-processing)
//   VisionOrientation           include/grid_vision/vision_orientation.hpp:41-99 (geometry half)
// ROS message types are replaced by layout-compatible PODs; on a ROS2 machine the node shim
// converts.  Errors: the reference logs and carries on; here
// every call returns/raises gv::Error only for programming errors and HIP/RCCL failures,
// out-of-map rectangles/points are silently skipped exactly like the reference.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <cstring>
#include <vector>

#include "../../../include/gridvision_hip.h"

namespace gv {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc, gv_handle h, const char *where)
{
  if (rc != GV_OK) throw Error(rc, std::string(where) + ": status " + std::to_string(rc) + " " + (h ? gv_last_error(h) : ""));
}

}  // namespace gv

// ObjectClass / BoundingBox  object_detection.hpp:12-32
enum class ObjectClass : int32_t {
  BIKE = 0, MOTORBIKE = 1, PERSON = 2, TRAFFIC_LIGHT_GREEN = 3, TRAFFIC_LIGHT_ORANGE = 4, TRAFFIC_LIGHT_RED = 5,
  TRAFFIC_SIGN_30 = 6, TRAFFIC_SIGN_60 = 7, TRAFFIC_SIGN_90 = 8, VEHICLE = 9, UNKNOWN = 10
};
using BoundingBox = gv_bbox;          // {x_min,y_min,x_max,y_max:f64, confidence:f32, label:i32}
using LShapePose = gv_lshape_pose;    // geometry_msgs/Pose + length, width, height
using CAMParams = gv_cam_params;      // vision_orientation.hpp:18-25

namespace geometry {
struct Point { double x, y, z; };
}

// One context = one GPU + one stream + one resident grid (the node owns exactly one).
class GridVisionContext {
public:
  GridVisionContext(uint8_t grid_x, uint8_t grid_y, double resolution, const CAMParams &cam, int device = -1)
  {
    gv::check(gv_create(&h_, grid_x, grid_y, resolution, &cam, device), nullptr, "gv_create");
  }
  ~GridVisionContext()
  {
    if (h_) gv_destroy(h_);
    for (auto &st : stage_)
      if (st.p) gv_host_free(st.p);
  }
  GridVisionContext(const GridVisionContext &) = delete;
  GridVisionContext &operator=(const GridVisionContext &) = delete;
  gv_handle handle() const { return h_; }

  // tf lookups of the node (grid_vision_node.cpp:290,348,371) handed over as transforms
  void setTransforms(const gv_transform *camera_from_lidar, const gv_transform *base_from_camera,
                     const gv_transform *base_from_lidar)
  {
    gv::check(gv_set_transforms(h_, camera_from_lidar, base_from_camera, base_from_lidar), h_, "gv_set_transforms");
  }
  // GridVision::cloudCallback (grid_vision_node.cpp:103-106)
  void setCloud(const float *x, const float *y, const float *z, size_t n)
  {
    gv::check(gv_cloud_upload_xyz(h_, x, y, z, n), h_, "gv_cloud_upload_xyz");
    n_ = n;
  }
  void setCloudPointCloud2(const uint8_t *data, size_t n, uint32_t point_step, uint32_t ox, uint32_t oy, uint32_t oz)
  {
    gv::check(gv_cloud_upload_pointcloud2(h_, data, n, point_step, ox, oy, oz), h_, "gv_cloud_upload_pointcloud2");
    n_ = n;
  }
  // Streaming form: the copy of the next cloud is enqueued on the handle's copy stream and overlaps the frames in
  // flight (three resident clouds rotate).  The ROS message is released when the callback returns, so the
  // bytes are staged in the context's own pinned buffers first (two, alternating); the host never waits for the
  // device here.
  void setCloudPointCloud2Async(const uint8_t *data, size_t n, uint32_t point_step, uint32_t ox, uint32_t oy, uint32_t oz)
  {
    const size_t bytes = n * (size_t)point_step;
    Staging &st = stage_[flip_ ^= 1];
    if (bytes > st.cap) {
      gv::check(gv_cloud_upload_wait(h_), h_, "gv_cloud_upload_wait");   // the old buffer may still be in flight
      if (st.p) gv_host_free(st.p);
      st.p = nullptr;
      gv::check(gv_host_alloc(&st.p, bytes + bytes / 8 + 4096), nullptr, "gv_host_alloc");
      st.cap = bytes + bytes / 8 + 4096;
    }
    // The copy that last read this buffer was enqueued two clouds ago.  At the node's 20 Hz it has long
    // finished and the wait returns at once; behind a held-back copy stream (back-pressure, a burst of clouds)
    // it keeps the bytes under an in-flight DMA unchanged, as the header demands.
    if (st.submitted) gv::check(gv_cloud_upload_wait(h_), h_, "gv_cloud_upload_wait");
    std::memcpy(st.p, data, bytes);
    st.submitted = true;
    gv::check(gv_cloud_upload_pointcloud2_async(h_, static_cast<const uint8_t *>(st.p), n, point_step, ox, oy, oz), h_,
              "gv_cloud_upload_pointcloud2_async");
    n_ = n;
  }
  // the fused frame without a host wait (gv_frame_set_detections + gv_frame_enqueue); results through
  // occupancyGridAsync / gv_synchronize
  void enqueueFrame(const gv_frame_desc &d)
  {
    gv::check(gv_frame_set_detections_async(h_, &d), h_, "gv_frame_set_detections_async");
    gv::check(gv_frame_enqueue(h_), h_, "gv_frame_enqueue");
  }
  void synchronize() { gv::check(gv_synchronize(h_), h_, "gv_synchronize"); }
  size_t cloudSize() const { return n_; }

  // The device half of one timerCallback as a single batch (gv_tick_enqueue / gv_tick_wait): kNN depth of the static
  // boxes, poses of the dynamic ones (orientation-network geometry or cloud PCA), rectangles, map update, int8 pack.
  // tickWait is the tick's only host wait.
  struct TickOutput {
    std::vector<BoundingBox> static_bboxes;
    std::vector<float> depths;                   // :176-177
    std::vector<geometry::Point> base_points;    // :180
    std::vector<LShapePose> poses;               // base frame (:204, :227)
    bool pca_empty = false;                      // computeBBoxPose returned {} (:307-309)
  };
  void tickEnqueue(const std::vector<BoundingBox> &bboxes, uint32_t flags, uint16_t k_near, const float *orient = nullptr,
                   const float *conf = nullptr, const float *dims = nullptr, int32_t n_net = 0, int8_t *grid_out = nullptr)
  {
    gv_tick_desc d{};
    d.flags = flags;
    d.bboxes = bboxes.data();
    d.n_bboxes = (int32_t)bboxes.size();
    d.orient = orient; d.conf = conf; d.dims = dims;
    d.n_net = n_net;
    d.k_near = k_near;
    d.grid_out = grid_out;
    tick_n_ = bboxes.size();
    gv::check(gv_tick_enqueue(h_, &d), h_, "gv_tick_enqueue");
  }
  TickOutput tickWait()
  {
    TickOutput o;
    const size_t cap = tick_n_ ? tick_n_ : 1;
    o.static_bboxes.resize(cap); o.depths.resize(cap); o.base_points.resize(cap); o.poses.resize(cap);
    gv_tick_result r{};
    r.static_bboxes = o.static_bboxes.data();
    r.depths = o.depths.data();
    r.base_points_xyz = reinterpret_cast<double *>(o.base_points.data());
    r.poses = o.poses.data();
    gv::check(gv_tick_wait(h_, &r), h_, "gv_tick_wait");
    o.static_bboxes.resize((size_t)r.n_static); o.depths.resize((size_t)r.n_static); o.base_points.resize((size_t)r.n_static);
    o.poses.resize((size_t)r.n_poses);
    o.pca_empty = r.pca_empty != 0;
    return o;
  }

private:
  struct Staging {
    void *p = nullptr;
    size_t cap = 0;
    bool submitted = false;   // a copy out of this buffer has been enqueued
  };
  Staging stage_[2];
  int flip_ = 0;
  gv_handle h_ = nullptr;
  size_t n_ = 0;
  size_t tick_n_ = 0;
};

// OccupancyGridMap  occupancy_grid.hpp:13-40.  grid_map_ is the device-resident grid.
class OccupancyGridMap {
public:
  explicit OccupancyGridMap(GridVisionContext &ctx) : ctx_(ctx)
  {
    int32_t nx, ny;
    gv::check(gv_grid_geometry(ctx_.handle(), &nx, &ny, nullptr, nullptr), ctx_.handle(), "gv_grid_geometry");
    cells_ = (size_t)nx * ny;
  }
  // updateMap(GridMap&)  occupancy_grid.cpp:16-31
  void updateMap() { gv::check(gv_update_map(ctx_.handle()), ctx_.handle(), "gv_update_map"); }
  // updateMap(GridMap&, vector<LShapePose>)  :65-105
  void updateMap(const std::vector<LShapePose> &bboxes_pose)
  {
    gv::check(gv_update_map_poses(ctx_.handle(), bboxes_pose.data(), (int32_t)bboxes_pose.size()), ctx_.handle(),
              "gv_update_map_poses");
  }
  // updateMap(GridMap&, vector<Point>, vector<BoundingBox>)  :33-63
  void updateMap(const std::vector<geometry::Point> &base_points, const std::vector<BoundingBox> &bboxes)
  {
    gv::check(gv_update_map_points(ctx_.handle(), reinterpret_cast<const double *>(base_points.data()), bboxes.data(),
                                   (int32_t)bboxes.size()), ctx_.handle(), "gv_update_map_points");
  }
  // GridMapRosConverter::toOccupancyGrid(map, "occupancy", 0, 1, msg)  grid_vision_node.cpp:270-271
  std::vector<int8_t> toOccupancyGrid(gv_grid_info *info = nullptr) const
  {
    std::vector<int8_t> data(cells_);
    gv::check(gv_to_occupancy_grid(ctx_.handle(), data.data(), info), ctx_.handle(), "gv_to_occupancy_grid");
    return data;
  }
  std::vector<float> layer(const char *name) const   // "log_odds" | "occupancy"
  {
    std::vector<float> v(cells_);
    const bool lo = std::string(name) == "log_odds";
    gv::check(lo ? gv_get_log_odds(ctx_.handle(), v.data()) : gv_get_occupancy(ctx_.handle(), v.data()), ctx_.handle(), name);
    return v;
  }
  size_t cells() const { return cells_; }

private:
  GridVisionContext &ctx_;
  size_t cells_ = 0;
};

namespace cloud_detections {

// transformLidarToCamera  grid_vision_node.hpp:95-97; empty result where the reference returns nullptr
inline bool transformLidarToCamera(GridVisionContext &ctx, std::vector<float> &x, std::vector<float> &y,
                                   std::vector<float> &z)
{
  x.resize(ctx.cloudSize()); y.resize(ctx.cloudSize()); z.resize(ctx.cloudSize());
  const int rc = gv_transform_lidar_to_camera(ctx.handle(), x.data(), y.data(), z.data());
  if (rc == GV_ERR_TF) return false;   // tf lookup failed: the node publishes the stale grid (:160-164)
  gv::check(rc, ctx.handle(), "gv_transform_lidar_to_camera");
  return true;
}

// buildKDTree + computeDepthForBoundingBoxes  cloud_detections.hpp:29-35
inline std::vector<float> computeDepthForBoundingBoxes(GridVisionContext &ctx, const std::vector<BoundingBox> &bboxes,
                                                       uint16_t k = 10)
{
  std::vector<float> depths(bboxes.size(), -1.0f);
  if (!bboxes.empty())
    gv::check(gv_compute_depth_for_bboxes(ctx.handle(), bboxes.data(), (int32_t)bboxes.size(), k, depths.data(), nullptr),
              ctx.handle(), "gv_compute_depth_for_bboxes");
  return depths;
}

// convertPixelsTo3D -> pixelTo3D -> transformPointToBaseFrame  grid_vision_node.cpp:309-359
inline std::vector<geometry::Point> convertPixelsTo3D(GridVisionContext &ctx, const std::vector<BoundingBox> &bboxes,
                                                      const std::vector<float> &depths)
{
  std::vector<geometry::Point> out(bboxes.size());
  if (!bboxes.empty())
    gv::check(gv_convert_pixels_to_3d(ctx.handle(), bboxes.data(), depths.data(), (int32_t)bboxes.size(),
                                      reinterpret_cast<double *>(out.data())), ctx.handle(), "gv_convert_pixels_to_3d");
  return out;
}

// extractCloudPerBBox  cloud_detections.hpp:46-48: bbox id of every point instead of copies
inline std::vector<int32_t> extractCloudPerBBox(GridVisionContext &ctx, const std::vector<BoundingBox> &bboxes,
                                                std::vector<int32_t> *counts = nullptr)
{
  std::vector<int32_t> ids(ctx.cloudSize());
  std::vector<int32_t> cnt(bboxes.size());
  gv::check(gv_extract_cloud_per_bbox(ctx.handle(), bboxes.data(), (int32_t)bboxes.size(), ids.data(), cnt.data()),
            ctx.handle(), "gv_extract_cloud_per_bbox");
  if (counts) *counts = cnt;
  return ids;
}

// segmentGroundPlane  cloud_detections.hpp:40-41: mask of the ground points of the camera-frame cloud
// (empty vector where the reference returns an empty cloud: no plane found)
inline std::vector<uint8_t> segmentGroundPlane(GridVisionContext &ctx, float coeff[4] = nullptr)
{
  std::vector<uint8_t> mask(ctx.cloudSize());
  int64_t m = 0;
  gv::check(gv_segment_ground_plane(ctx.handle(), 0.04, 50, 12345ull, mask.data(), coeff, &m), ctx.handle(),
            "gv_segment_ground_plane");
  if (m == 0) mask.clear();
  return mask;
}

// computeBBoxPose  cloud_detections.hpp:50-52.  remove_ground = true is the reference's flow
// (segmentGroundPlane first, :306-314); false skips the RANSAC step.
inline std::vector<LShapePose> computeBBoxPose(GridVisionContext &ctx, const std::vector<BoundingBox> &bboxes,
                                               bool remove_ground = true)
{
  std::vector<LShapePose> all(bboxes.size()), out;
  std::vector<uint8_t> valid(bboxes.size());
  if (!bboxes.empty()) {
    if (remove_ground) {
      int32_t np = 0;
      gv::check(gv_compute_bbox_pose_ground_removed(ctx.handle(), bboxes.data(), (int32_t)bboxes.size(), all.data(),
                                                    valid.data(), &np), ctx.handle(), "gv_compute_bbox_pose_ground_removed");
      if (np < 0) return out;   // empty segmented cloud: the reference returns {} (:307-309)
    } else
      gv::check(gv_compute_bbox_pose(ctx.handle(), bboxes.data(), (int32_t)bboxes.size(), all.data(), valid.data()),
                ctx.handle(), "gv_compute_bbox_pose");
  }
  for (size_t i = 0; i < all.size(); ++i)
    if (valid[i]) out.push_back(all[i]);   // the reference appends only non-empty clouds (:174-181)
  return out;
}

}  // namespace cloud_detections

namespace object_detection {

// extract_bboxes  object_detection.hpp:48-49 on precomputed detector outputs
inline std::vector<BoundingBox> extract_bboxes(const float *boxes, const float *scores, int num_detections, int num_classes,
                                               double conf_threshold, double iou_threshold, int orig_w, int orig_h,
                                               int resize)
{
  std::vector<BoundingBox> out((size_t)num_detections);
  int32_t n = 0;
  gv::check(gv_extract_bboxes(boxes, scores, num_detections, num_classes, conf_threshold, iou_threshold, orig_w, orig_h,
                              resize, out.data(), &n), nullptr, "gv_extract_bboxes");
  out.resize((size_t)n);
  return out;
}

// GridVision::filterBBoxes  grid_vision_node.cpp:384-403 -> (static, dynamic)
inline std::tuple<std::vector<BoundingBox>, std::vector<BoundingBox>> filterBBoxes(const std::vector<BoundingBox> &bboxes)
{
  std::vector<BoundingBox> st(bboxes.size()), dy(bboxes.size());
  int32_t ns = 0, nd = 0;
  gv::check(gv_filter_bboxes(bboxes.data(), (int32_t)bboxes.size(), st.data(), &ns, dy.data(), &nd), nullptr,
            "gv_filter_bboxes");
  st.resize((size_t)ns);
  dy.resize((size_t)nd);
  return {st, dy};
}

}  // namespace object_detection

// VisionOrientation, geometry half (postProcessOutputs and below, vision_orientation.hpp:90-98)
class VisionOrientation {
public:
  explicit VisionOrientation(GridVisionContext &ctx) : ctx_(ctx) {}
  // orient[nb*4], conf[nb*2], dims[nb*3] are the network outputs runInference copies back (:220-225)
  std::vector<LShapePose> postProcessOutputs(const float *orient, const float *conf, const float *dims,
                                             const std::vector<BoundingBox> &bboxes)
  {
    std::vector<LShapePose> out(bboxes.size());
    int32_t n = 0;
    if (!bboxes.empty())
      gv::check(gv_vision_post_process(ctx_.handle(), orient, conf, dims, bboxes.data(), (int32_t)bboxes.size(), out.data(),
                                       &n), ctx_.handle(), "gv_vision_post_process");
    out.resize((size_t)n);
    return out;
  }
  // GridVision::transformLShapeObjects  grid_vision_node.cpp:525-531
  void transformLShapeObjects(std::vector<LShapePose> &poses)
  {
    gv::check(gv_transform_lshape_objects(ctx_.handle(), poses.data(), (int32_t)poses.size()), ctx_.handle(),
              "gv_transform_lshape_objects");
  }

private:
  GridVisionContext &ctx_;
};
