// grid_vision_hip_node.cpp -- ROS2 node shim over libgridvision_hip.so.
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no ROS2 here); see ros2/README.md.
// Same ROS surface as the reference node (src/grid_vision_node.cpp): parameters :8-32,
// subscriptions :43-47, 50 ms wall timer :49-50, publishers :52-54, tf frames :290,:348,:371.
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include <geometry_msgs/msg/transform_stamped.hpp>
#include <nav_msgs/msg/occupancy_grid.hpp>
#include <rclcpp/rclcpp.hpp>
#include <sensor_msgs/msg/image.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>
#include <tf2_ros/buffer.h>
#include <tf2_ros/transform_listener.h>
#include <visualization_msgs/msg/marker_array.hpp>

#include <grid_vision/hip_backend.hpp>

namespace {
gv_transform toGv(const geometry_msgs::msg::Transform &t)
{
  return gv_transform{t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w, t.translation.x, t.translation.y,
                      t.translation.z};
}
}  // namespace

class GridVision : public rclcpp::Node {
public:
  // L1 hooks (out of scope here): detector outputs boxes[n,4] + scores[n,c]; orientation net
  // outputs orient[nb*4], conf[nb*2], dims[nb*3] for the given dynamic bboxes.
  using DetectorFn = std::function<bool(const sensor_msgs::msg::Image &, std::vector<float> &boxes,
                                        std::vector<float> &scores, int &n, int &c)>;
  using OrientationFn = std::function<bool(const sensor_msgs::msg::Image &, const std::vector<BoundingBox> &,
                                           std::vector<float> &orient, std::vector<float> &conf,
                                           std::vector<float> &dims)>;

  GridVision(DetectorFn detector, OrientationFn orientation)
      : Node("grid_vision_node"), detector_(std::move(detector)), orientation_(std::move(orientation))
  {
    image_topic_ = declare_parameter<std::string>("image_topic", "");
    lidar_topic_ = declare_parameter<std::string>("lidar_topic", "");
    declare_parameter<std::string>("detection_weights_file", "");
    declare_parameter<std::string>("vision_weights_file", "");
    lidar_frame_ = declare_parameter<std::string>("lidar_frame", "");
    camera_frame_ = declare_parameter<std::string>("camera_frame", "");
    base_frame_ = declare_parameter<std::string>("base_frame", "");
    conf_threshold_ = declare_parameter("confidence_threshold", 0.5);
    iou_threshold_ = declare_parameter("iou_threshold", 0.4);
    resize_ = static_cast<uint16_t>(declare_parameter("detection_network_input_size", 416));
    k_near_ = static_cast<uint16_t>(declare_parameter("k_near", 10));
    const auto grid_x = static_cast<uint8_t>(declare_parameter("grid_x", 50));
    const auto grid_y = static_cast<uint8_t>(declare_parameter("grid_y", 10));
    const double resolution = declare_parameter("resolution", 0.1);
    use_vision_orientation_ = declare_parameter("use_vision_orientation", false);
    CAMParams cam{};
    cam.orig_h = declare_parameter("camera_image_height", 480);
    cam.orig_w = declare_parameter("camera_image_width", 640);
    cam.network_h = declare_parameter("network_height", 224);
    cam.network_w = declare_parameter("network_width", 224);
    cam.fx = static_cast<float>(declare_parameter("fx", 0.0));
    cam.fy = static_cast<float>(declare_parameter("fy", 0.0));
    cam.cx = static_cast<float>(declare_parameter("cx", 0.0));
    cam.cy = static_cast<float>(declare_parameter("cy", 0.0));
    lidar_binning_ = declare_parameter("lidar_binning", false);
    lidar_raymarch_ = declare_parameter("lidar_raymarch", false);

    ctx_ = std::make_unique<GridVisionContext>(grid_x, grid_y, resolution, cam);   // OccupancyGridMap ctor, K, K^-1
    occ_grid_ = std::make_unique<OccupancyGridMap>(*ctx_);
    vision_ = std::make_unique<VisionOrientation>(*ctx_);

    if (image_topic_.empty() || lidar_topic_.empty()) {
      RCLCPP_ERROR(get_logger(), "Check if topic name or weight file is assigned");
      return;
    }
    image_sub_ = create_subscription<sensor_msgs::msg::Image>(
      image_topic_, 1, [this](sensor_msgs::msg::Image::ConstSharedPtr msg) { image_ = msg; });
    cloud_sub_ = create_subscription<sensor_msgs::msg::PointCloud2>(
      lidar_topic_, 1, [this](sensor_msgs::msg::PointCloud2::ConstSharedPtr msg) { cloudCallback(*msg); });
    timer_ = create_wall_timer(std::chrono::milliseconds(50), [this] { timerCallback(); });
    occupancy_pub_ = create_publisher<nav_msgs::msg::OccupancyGrid>("occupancy_grid", 10);
    viz_pub_ = create_publisher<visualization_msgs::msg::MarkerArray>("objects_viz", 10);
    tf_buffer_ = std::make_unique<tf2_ros::Buffer>(get_clock());
    tf_listener_ = std::make_unique<tf2_ros::TransformListener>(*tf_buffer_);
  }

private:
  void cloudCallback(const sensor_msgs::msg::PointCloud2 &msg)
  {
    uint32_t ox = 0, oy = 4, oz = 8;
    for (const auto &f : msg.fields) {
      if (f.name == "x") ox = f.offset;
      if (f.name == "y") oy = f.offset;
      if (f.name == "z") oz = f.offset;
    }
    // bytes go straight to the device (no pcl::fromROSMsg AoS copy), asynchronously: the copy overlaps the frame
    // the timer callback may still have in flight, and the de-interleave runs on the device
    ctx_->setCloudPointCloud2Async(msg.data.data(), static_cast<size_t>(msg.width) * msg.height, msg.point_step, ox, oy, oz);
    have_cloud_ = true;
  }

  bool lookup(const std::string &target, const std::string &source, gv_transform &out)
  {
    try {
      out = toGv(tf_buffer_->lookupTransform(target, source, tf2::TimePointZero).transform);
      return true;
    } catch (const tf2::TransformException &ex) {
      RCLCPP_ERROR(get_logger(), "Could not transform %s to %s: %s", source.c_str(), target.c_str(), ex.what());
      return false;
    }
  }

  void timerCallback()
  {
    if (!image_ && !have_cloud_) {   // same guard as the reference (&&, grid_vision_node.cpp:111)
      publishOccupancyGrid();
      return;
    }
    std::vector<float> boxes, scores;
    int n = 0, c = 0;
    std::vector<BoundingBox> bboxes;
    if (image_ && detector_ && detector_(*image_, boxes, scores, n, c))
      bboxes = object_detection::extract_bboxes(boxes.data(), scores.data(), n, c, conf_threshold_, iou_threshold_,
                                                image_->width, image_->height, resize_);
    if (bboxes.empty() && !lidar_binning_) {
      occ_grid_->updateMap();
      publishOccupancyGrid();
      return;
    }
    gv_transform cl{}, bc{}, bl{};
    if (!lookup(camera_frame_, lidar_frame_, cl) || !lookup(base_frame_, camera_frame_, bc)
        || !lookup(base_frame_, lidar_frame_, bl)) {
      publishOccupancyGrid();   // tf failure: publish the stale grid (:160-164)
      return;
    }
    ctx_->setTransforms(&cl, &bc, &bl);
    auto [static_bboxes, dynamic_bboxes] = object_detection::filterBBoxes(bboxes);
    if (!static_bboxes.empty()) {
      depth_vec_ = cloud_detections::computeDepthForBoundingBoxes(*ctx_, static_bboxes, k_near_);
      static_points_ = cloud_detections::convertPixelsTo3D(*ctx_, static_bboxes, depth_vec_);
    }
    std::vector<LShapePose> bboxes_pose;
    if (!dynamic_bboxes.empty()) {
      if (use_vision_orientation_) {
        std::vector<float> orient, conf, dims;
        if (orientation_ && orientation_(*image_, dynamic_bboxes, orient, conf, dims))
          bboxes_pose = vision_->postProcessOutputs(orient.data(), conf.data(), dims.data(), dynamic_bboxes);
      } else {
        bboxes_pose = cloud_detections::computeBBoxPose(*ctx_, bboxes);   // the reference passes ALL bboxes (:215-216)
      }
      vision_->transformLShapeObjects(bboxes_pose);
    }
    if (lidar_binning_) {   // [EXTENSION] one fused frame: bin + ray-march + rectangles + grid pass
      gv_frame_desc d{};
      d.flags = GV_FRAME_BIN | (lidar_raymarch_ ? GV_FRAME_RAYMARCH : 0u);
      d.poses = bboxes_pose.data();
      d.n_poses = static_cast<int32_t>(bboxes_pose.size());
      gv::check(gv_process_frame(ctx_->handle(), &d), ctx_->handle(), "gv_process_frame");
    } else if (!bboxes_pose.empty()) {
      occ_grid_->updateMap(bboxes_pose);
    } else {
      occ_grid_->updateMap();
    }
    publishOccupancyGrid();
  }

  void publishOccupancyGrid()
  {
    nav_msgs::msg::OccupancyGrid msg;
    gv_grid_info info{};
    const std::vector<int8_t> data = occ_grid_->toOccupancyGrid(&info);
    msg.header.stamp = rclcpp::Clock(RCL_ROS_TIME).now();
    msg.header.frame_id = base_frame_;
    msg.info.resolution = static_cast<float>(info.resolution);
    msg.info.width = info.width;
    msg.info.height = info.height;
    msg.info.origin.position.x = info.origin_x;
    msg.info.origin.position.y = info.origin_y;
    msg.info.origin.orientation.w = 1.0;
    msg.data.assign(data.begin(), data.end());
    occupancy_pub_->publish(msg);
  }

  DetectorFn detector_;
  OrientationFn orientation_;
  std::string image_topic_, lidar_topic_, lidar_frame_, camera_frame_, base_frame_;
  double conf_threshold_ = 0.5, iou_threshold_ = 0.4;
  uint16_t resize_ = 416, k_near_ = 10;
  bool use_vision_orientation_ = false, lidar_binning_ = false, lidar_raymarch_ = false, have_cloud_ = false;
  std::unique_ptr<GridVisionContext> ctx_;
  std::unique_ptr<OccupancyGridMap> occ_grid_;
  std::unique_ptr<VisionOrientation> vision_;
  sensor_msgs::msg::Image::ConstSharedPtr image_;
  std::vector<float> depth_vec_;
  std::vector<geometry::Point> static_points_;
  rclcpp::Subscription<sensor_msgs::msg::Image>::SharedPtr image_sub_;
  rclcpp::Subscription<sensor_msgs::msg::PointCloud2>::SharedPtr cloud_sub_;
  rclcpp::TimerBase::SharedPtr timer_;
  rclcpp::Publisher<nav_msgs::msg::OccupancyGrid>::SharedPtr occupancy_pub_;
  rclcpp::Publisher<visualization_msgs::msg::MarkerArray>::SharedPtr viz_pub_;
  std::unique_ptr<tf2_ros::Buffer> tf_buffer_;
  std::unique_ptr<tf2_ros::TransformListener> tf_listener_;
};

int main(int argc, char *argv[])
{
  rclcpp::init(argc, argv);
  // plug the L1 networks in here (ONNX detector, TensorRT orientation net: out of scope of this repo)
  auto node = std::make_shared<GridVision>(GridVision::DetectorFn{}, GridVision::OrientationFn{});
  rclcpp::spin(node);
  rclcpp::shutdown();
  return 0;
}
