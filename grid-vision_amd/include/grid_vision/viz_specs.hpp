// grid_vision/viz_specs.hpp -- what GridVision::publishObjectVisualizations (src/grid_vision_node.cpp:405-523) and
// GridVision::publishObjectDetections -> object_detection::draw_bboxes (:246-263, src/object_detection.cpp:213-224)
// put on the wire, without ROS or OpenCV: every field of every visualization_msgs/Marker and every rectangle / label of
// the detection overlay as plain data.  A ROS2 node copies a MarkerSpec field by field into a Marker and hands an
// OverlaySpec to cv::rectangle / cv::putText (ros2/src/grid_vision_hip_node.cpp); the decisions -- ids, types, scales,
// colours, lifetimes, which box gets which marker, integer truncations, label text -- are made here, compiled and tested
// without either library (examples/viz_demo.cpp, tests/test_host_side.py).
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "hip_backend.hpp"

namespace grid_vision {

// visualization_msgs/msg/Marker constants used by the reference
enum MarkerType : int32_t { MARKER_CUBE = 1, MARKER_SPHERE = 2, MARKER_TEXT_VIEW_FACING = 9 };
enum MarkerAction : int32_t { MARKER_ADD = 0 };

struct MarkerSpec {
  std::string frame_id;   // header.frame_id = base_frame_ (:426, :463, :501)
  std::string ns;
  int32_t id = 0;
  int32_t type = 0;
  int32_t action = MARKER_ADD;
  double lifetime_s = 0.0;                       // rclcpp::Duration::from_seconds
  double px = 0, py = 0, pz = 0;                 // pose.position
  double qx = 0, qy = 0, qz = 0, qw = 0;         // pose.orientation (a default-constructed message: all zero)
  double sx = 0, sy = 0, sz = 0;                 // scale
  float r = 0, g = 0, b = 0, a = 0;              // color
  std::string text;
};

// object_detection::objectClassToString  src/object_detection.cpp:272-290
inline const char *objectClassToString(int32_t label)
{
  switch (static_cast<ObjectClass>(label)) {
  case ObjectClass::BIKE: return "Bike";
  case ObjectClass::MOTORBIKE: return "Motorbike";
  case ObjectClass::PERSON: return "Person";
  case ObjectClass::TRAFFIC_LIGHT_GREEN: return "Light Green";
  case ObjectClass::TRAFFIC_LIGHT_ORANGE: return "Light Orange";
  case ObjectClass::TRAFFIC_LIGHT_RED: return "Light Red";
  case ObjectClass::TRAFFIC_SIGN_30: return "Sign 30";
  case ObjectClass::TRAFFIC_SIGN_60: return "Sign 60";
  case ObjectClass::TRAFFIC_SIGN_90: return "Sign 90";
  case ObjectClass::VEHICLE: return "Vehicle";
  default: return "Unknown";
  }
}

// GridVision::publishObjectVisualizations  src/grid_vision_node.cpp:405-523.  One id counter over all markers (:412);
// static objects first, in the order of static_positions (which is the order of static_bboxes, :415-418), a traffic
// light as a 0.3 m sphere in its colour (:421-457), a speed sign as white text one metre above its position (:460-495),
// anything else static (an UNKNOWN class) nothing; then every L-shape box as a cube (:499-520) whose scale.z is
// box.height -- never set on the PCA path of the reference (cloud_detections.cpp:187-247; this library hands 0.0 there).
inline std::vector<MarkerSpec> buildObjectVisualizations(const std::vector<LShapePose> &lshape_boxes,
                                                         const std::vector<geometry::Point> &static_positions,
                                                         const std::vector<BoundingBox> &static_bboxes,
                                                         const std::string &base_frame)
{
  std::vector<MarkerSpec> out;
  int id = 0;
  for (size_t i = 0; i < static_positions.size() && i < static_bboxes.size(); ++i) {
    const geometry::Point &pos = static_positions[i];
    const auto label = static_cast<ObjectClass>(static_bboxes[i].label);
    if (label == ObjectClass::TRAFFIC_LIGHT_RED || label == ObjectClass::TRAFFIC_LIGHT_ORANGE ||
        label == ObjectClass::TRAFFIC_LIGHT_GREEN) {
      MarkerSpec m;
      m.frame_id = base_frame;
      m.ns = "traffic_light";
      m.id = id++;
      m.type = MARKER_SPHERE;
      m.action = MARKER_ADD;
      m.lifetime_s = 0.2;
      m.px = pos.x; m.py = pos.y; m.pz = pos.z;
      m.qw = 1.0;
      m.sx = m.sy = m.sz = 0.3;
      m.a = 1.0f;
      if (label == ObjectClass::TRAFFIC_LIGHT_RED) m.r = 1.0f;
      else if (label == ObjectClass::TRAFFIC_LIGHT_ORANGE) { m.r = 1.0f; m.g = 1.0f; }
      else m.g = 1.0f;
      out.push_back(m);
    }
    if (label == ObjectClass::TRAFFIC_SIGN_30 || label == ObjectClass::TRAFFIC_SIGN_60 || label == ObjectClass::TRAFFIC_SIGN_90) {
      MarkerSpec m;
      m.frame_id = base_frame;
      m.ns = "traffic_sign";
      m.id = id++;
      m.type = MARKER_TEXT_VIEW_FACING;
      m.action = MARKER_ADD;
      m.lifetime_s = 0.2;
      m.px = pos.x; m.py = pos.y; m.pz = pos.z;
      m.pz += 1.0;   // :471
      m.qw = 1.0;
      m.sz = 0.5;    // :474 (scale.x / .y stay 0)
      m.r = m.g = m.b = m.a = 1.0f;
      m.text = label == ObjectClass::TRAFFIC_SIGN_30 ? "SPEED LIMIT: 30 KMPH"
               : label == ObjectClass::TRAFFIC_SIGN_60 ? "SPEED LIMIT: 60 KMPH" : "SPEED LIMIT: 90 KMPH";
      out.push_back(m);
    }
  }
  for (const LShapePose &box : lshape_boxes) {
    MarkerSpec m;
    m.frame_id = base_frame;
    m.ns = "lshape_bbox";
    m.id = id++;
    m.type = MARKER_CUBE;
    m.action = MARKER_ADD;
    m.lifetime_s = 0.1;
    m.px = box.px; m.py = box.py; m.pz = box.pz;                       // box_marker.pose = box.pose (:509)
    m.qx = box.qx; m.qy = box.qy; m.qz = box.qz; m.qw = box.qw;
    m.sx = box.length; m.sy = box.width; m.sz = box.height;            // :510-512 (the height quirk at :512)
    m.r = 0.0f; m.g = 0.5f; m.b = 1.0f; m.a = 1.0f;
    out.push_back(m);
  }
  return out;
}

// one rectangle + label of the detection overlay: cv::rectangle(image, Rect(x, y, w, h), Scalar(0, 255, 0), 2) and
// cv::putText(image, label, Point(text_x, text_y), FONT_HERSHEY_SIMPLEX, 0.5, Scalar(0, 255, 0), 1)
struct OverlaySpec {
  int x = 0, y = 0, w = 0, h = 0;   // cv::Rect(box.x_min, box.y_min, box.x_max - box.x_min, box.y_max - box.y_min): doubles
                                    // narrowed to the int fields of cv::Rect (truncation towards zero)
  int text_x = 0, text_y = 0;       // cv::Point(box.x_min, box.y_min - 5)
  std::string label;                // objectClassToString(label) + " (" + std::to_string(confidence) + ")"
  uint8_t r = 0, g = 255, b = 0;    // the image is rgb8 (:258), cv::Scalar(0, 255, 0)
  int box_thickness = 2, text_thickness = 1;
  double font_scale = 0.5;
};

// object_detection::draw_bboxes  src/object_detection.cpp:213-224
inline std::vector<OverlaySpec> buildDetectionOverlay(const std::vector<BoundingBox> &bboxes)
{
  std::vector<OverlaySpec> out;
  out.reserve(bboxes.size());
  for (const BoundingBox &box : bboxes) {
    OverlaySpec o;
    o.x = static_cast<int>(box.x_min);
    o.y = static_cast<int>(box.y_min);
    o.w = static_cast<int>(box.x_max - box.x_min);
    o.h = static_cast<int>(box.y_max - box.y_min);
    o.text_x = static_cast<int>(box.x_min);
    o.text_y = static_cast<int>(box.y_min - 5);
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%f", static_cast<double>(box.confidence));   // std::to_string(float): "%f" of the double
    o.label = std::string(objectClassToString(box.label)) + " (" + buf + ")";
    out.push_back(o);
  }
  return out;
}

// The rectangle outlines drawn into an rgb8 image without OpenCV (a node without cv_bridge, tests): the pixels within
// box_thickness / 2 of the rectangle's border lines, clipped to the image.  cv::rectangle rounds the joins of thick
// lines slightly differently at the corners; the labels need OpenCV's Hershey font and are not drawn here.
inline void drawOverlayRectangles(uint8_t *rgb, int width, int height, const std::vector<OverlaySpec> &specs)
{
  auto put = [&](int px, int py, const OverlaySpec &o) {
    if (px < 0 || py < 0 || px >= width || py >= height) return;
    uint8_t *p = rgb + (static_cast<size_t>(py) * width + px) * 3;
    p[0] = o.r; p[1] = o.g; p[2] = o.b;
  };
  for (const OverlaySpec &o : specs) {
    const int t0 = -(o.box_thickness / 2), t1 = (o.box_thickness - 1) / 2;   // thickness 2: offsets -1, 0
    const int x0 = o.x, y0 = o.y, x1 = o.x + o.w - 1, y1 = o.y + o.h - 1;    // cv::Rect covers [x, x + w) x [y, y + h)
    if (o.w <= 0 || o.h <= 0) continue;
    for (int t = t0; t <= t1; ++t) {
      for (int px = x0 + t0; px <= x1 + t1; ++px) { put(px, y0 + t, o); put(px, y1 + t, o); }
      for (int py = y0 + t0; py <= y1 + t1; ++py) { put(x0 + t, py, o); put(x1 + t, py, o); }
    }
  }
}

}  // namespace grid_vision
