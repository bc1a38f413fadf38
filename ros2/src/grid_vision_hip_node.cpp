// grid_vision_hip_node.cpp -- ROS2 node shim over libgridvision_hip.so.
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no ROS2 here); see ros2/README.md.  Build: cmake -DGV_WITH_ROS2=ON (CMakeLists.txt).
// Everything this file decides is message / tf conversion; the callback's decisions live in grid_vision/frame_flow.hpp
// and the contents of the markers and of the detection overlay in grid_vision/viz_specs.hpp, both compiled and tested
// without ROS (tests/test_gpu_parity.py::test_cpp_flow_demo, tests/test_host_side.py).
// Same ROS surface as the reference node (src/grid_vision_node.cpp): parameters :8-32,
// subscriptions :43-47, 50 ms wall timer :49-50, publishers :52-54, tf frames :290,:348,:371.
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include <cv_bridge/cv_bridge.h>
#include <geometry_msgs/msg/transform_stamped.hpp>
#include <image_transport/image_transport.hpp>
#include <nav_msgs/msg/occupancy_grid.hpp>
#include <opencv2/imgproc.hpp>
#include <rclcpp/rclcpp.hpp>
#include <sensor_msgs/msg/image.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>
#include <tf2_ros/buffer.h>
#include <tf2_ros/transform_listener.h>
#include <visualization_msgs/msg/marker_array.hpp>

#include <grid_vision/frame_flow.hpp>
#include <grid_vision/viz_specs.hpp>

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
    grid_vision::FlowParams fp;
    fp.conf_threshold = conf_threshold_;
    fp.iou_threshold = iou_threshold_;
    fp.resize = resize_;
    fp.k_near = k_near_;
    fp.use_vision_orientation = use_vision_orientation_;
    fp.lidar_binning = lidar_binning_;
    fp.lidar_raymarch = lidar_raymarch_;
    flow_ = std::make_unique<grid_vision::FrameFlow>(*ctx_, *occ_grid_, fp);

    if (image_topic_.empty() || lidar_topic_.empty()) {
      RCLCPP_ERROR(get_logger(), "Check if topic name or weight file is assigned");
      return;
    }
    image_sub_ = create_subscription<sensor_msgs::msg::Image>(
      image_topic_, 1, [this](sensor_msgs::msg::Image::ConstSharedPtr msg) { image_ = msg; });
    cloud_sub_ = create_subscription<sensor_msgs::msg::PointCloud2>(
      lidar_topic_, 1, [this](sensor_msgs::msg::PointCloud2::ConstSharedPtr msg) { cloudCallback(*msg); });
    timer_ = create_wall_timer(std::chrono::milliseconds(50), [this] { timerCallback(); });
    detection_pub_ = image_transport::create_publisher(this, "carla/front/detections");             // :52
    occupancy_pub_ = create_publisher<nav_msgs::msg::OccupancyGrid>("occupancy_grid", 10);           // :53
    viz_pub_ = create_publisher<visualization_msgs::msg::MarkerArray>("objects_viz", 10);            // :54
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

  // The decisions of the reference's timerCallback (grid_vision_node.cpp:108-244) live in grid_vision::FrameFlow
  // (grid-vision_amd/include/grid_vision/frame_flow.hpp, compiled and tested without ROS); this callback only
  // converts: subscriptions -> TickInput, tf -> gv_transform, TickResult -> publishers.
  void timerCallback()
  {
    grid_vision::TickInput in;
    in.have_image = static_cast<bool>(image_);
    in.have_cloud = have_cloud_;
    std::vector<float> boxes, scores;
    int n = 0, c = 0;
    if (image_) {
      in.image_w = static_cast<int>(image_->width);
      in.image_h = static_cast<int>(image_->height);
      if (detector_ && detector_(*image_, boxes, scores, n, c)) {   // run_inference (:124-125)
        in.det_boxes = boxes.data();
        in.det_scores = scores.data();
        in.n_det = n;
        in.n_classes = c;
      }
      in.orientation_net = [this](const std::vector<BoundingBox> &dyn, std::vector<float> &orient, std::vector<float> &conf,
                                  std::vector<float> &dims) {
        if (!orientation_ || !orientation_(*image_, dyn, orient, conf, dims)) { orient.clear(); conf.clear(); dims.clear(); }
      };
    }
    gv_transform cl{}, bc{}, bl{};
    const bool tf_ok = lookup(camera_frame_, lidar_frame_, cl) && lookup(base_frame_, camera_frame_, bc)
                       && lookup(base_frame_, lidar_frame_, bl);
    if (tf_ok) ctx_->setTransforms(&cl, &bc, &bl);
    flow_->setTransformsAvailable(tf_ok);
    const grid_vision::TickResult r = flow_->tick(in);
    if (!r.warning.empty()) RCLCPP_WARN(get_logger(), "%s", r.warning.c_str());
    depth_vec_ = r.depth_vec;
    static_points_ = r.cam_points;
    // the reference's order: detections, grid, markers (:239-243); the early-return paths publish the grid only
    if (r.publish_detections && image_) publishObjectDetections(r.bboxes);
    if (r.publish_grid) publishOccupancyGrid();
    if (r.publish_detections) publishObjectVisualizations(r.bboxes_pose, r.cam_points, r.static_bboxes);
  }

  // GridVision::publishObjectDetections (:246-263): the boxes drawn into a copy of the camera image.  What is drawn
  // where is grid_vision::buildDetectionOverlay (object_detection::draw_bboxes, src/object_detection.cpp:213-224)
  void publishObjectDetections(const std::vector<BoundingBox> &bboxes)
  {
    cv_bridge::CvImagePtr img;
    try {
      img = cv_bridge::toCvCopy(image_, "rgb8");   // imageCallback's conversion (:84), on a copy (:251)
    } catch (const cv_bridge::Exception &e) {
      RCLCPP_ERROR(get_logger(), "cv_bridge error: %s", e.what());
      return;
    }
    for (const grid_vision::OverlaySpec &o : grid_vision::buildDetectionOverlay(bboxes)) {
      const cv::Scalar colour(o.r, o.g, o.b);
      cv::rectangle(img->image, cv::Rect(o.x, o.y, o.w, o.h), colour, o.box_thickness);
      cv::putText(img->image, o.label, cv::Point(o.text_x, o.text_y), cv::FONT_HERSHEY_SIMPLEX, o.font_scale, colour, o.text_thickness);
    }
    std_msgs::msg::Header header;
    header.stamp = rclcpp::Clock(RCL_ROS_TIME).now();   // :257
    detection_pub_.publish(*cv_bridge::CvImage(header, "rgb8", img->image).toImageMsg());
  }

  // GridVision::publishObjectVisualizations (:405-523): one Marker per grid_vision::MarkerSpec, field by field
  void publishObjectVisualizations(const std::vector<LShapePose> &lshape_boxes, const std::vector<geometry::Point> &static_positions,
                                   const std::vector<BoundingBox> &static_bboxes)
  {
    visualization_msgs::msg::MarkerArray arr;
    for (const grid_vision::MarkerSpec &m : grid_vision::buildObjectVisualizations(lshape_boxes, static_positions, static_bboxes, base_frame_)) {
      visualization_msgs::msg::Marker k;
      k.header.frame_id = m.frame_id;
      k.header.stamp = rclcpp::Clock().now();   // :427 (system clock, as the reference)
      k.ns = m.ns;
      k.id = m.id;
      k.type = m.type;
      k.action = m.action;
      k.lifetime = rclcpp::Duration::from_seconds(m.lifetime_s);
      k.pose.position.x = m.px; k.pose.position.y = m.py; k.pose.position.z = m.pz;
      k.pose.orientation.x = m.qx; k.pose.orientation.y = m.qy; k.pose.orientation.z = m.qz; k.pose.orientation.w = m.qw;
      k.scale.x = m.sx; k.scale.y = m.sy; k.scale.z = m.sz;
      k.color.r = m.r; k.color.g = m.g; k.color.b = m.b; k.color.a = m.a;
      k.text = m.text;
      arr.markers.push_back(k);
    }
    viz_pub_->publish(arr);
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
  std::unique_ptr<grid_vision::FrameFlow> flow_;
  sensor_msgs::msg::Image::ConstSharedPtr image_;
  std::vector<float> depth_vec_;
  std::vector<geometry::Point> static_points_;
  rclcpp::Subscription<sensor_msgs::msg::Image>::SharedPtr image_sub_;
  rclcpp::Subscription<sensor_msgs::msg::PointCloud2>::SharedPtr cloud_sub_;
  rclcpp::TimerBase::SharedPtr timer_;
  image_transport::Publisher detection_pub_;
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
