/*
 * gridvision_hip.h -- C ABI of the MI355X (gfx950) implementation of
 * grid-vision's per-frame hot path.  libgridvision_hip.so exports exactly the
 * functions declared here; plain pointers and sizes, no C++/torch types, no
 * exception ever crosses this boundary.
 *
 * The reference (rohankhaire-work/grid-vision) has no FFI boundary of its own:
 * the hot path is called in-process from GridVision::timerCallback
 * (src/grid_vision_node.cpp:108-244).  Each entry point below cites the
 * reference interface it replaces (file:line relative to the reference root).
 * INTEGRATION.md shows the node-side binding.
 *
 * Conventions
 *   - every function returns a gv_status (0 = ok); out-of-map rectangles and
 *     points are NOT errors (the reference skips them silently,
 *     src/occupancy_grid.cpp:152-156,171-172);
 *   - one handle = one GPU + one resident grid + the HIP streams of its frame
 *     pipeline (gv_stream returns the public one: every frame finishes there, in
 *     order, and callers may order their own work on it); a handle is used by
 *     one thread at a time; handles are independent;
 *   - host pointers are caller owned and may be pageable; every call returns
 *     after its results are complete in the caller's buffers (synchronous),
 *     except the streaming calls gv_frame_enqueue, gv_cloud_upload_*_async and
 *     gv_frame_set_detections_async (see there);
 *   - grid layers use the reference's storage order: grid_map's column-major
 *     Eigen::MatrixXf(size0,size1) with row = x index, i.e. linear = iy*nx+ix;
 *   - [EXTENSION] marks what north_star asks for and the reference lacks.
 */
#ifndef GRIDVISION_HIP_H_
#define GRIDVISION_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gv_context *gv_handle;

typedef enum {
  GV_OK = 0,
  GV_ERR_BAD_ARG = 1,     /* null pointer, size out of range, bad flag        */
  GV_ERR_HIP = 2,         /* a HIP runtime call failed (gv_last_error)        */
  GV_ERR_RCCL = 3,        /* an RCCL call failed                               */
  GV_ERR_NO_DEVICE = 4,   /* no gfx950 device / device id out of range        */
  GV_ERR_STATE = 5,       /* call order violated (e.g. no cloud uploaded)     */
  GV_ERR_TF = 6           /* a required transform was never set (the          */
                          /* reference returns nullptr: grid_vision_node.cpp:292-297) */
} gv_status;

/* BoundingBox  include/grid_vision/object_detection.hpp:27-32 (40 bytes) */
typedef struct {
  double x_min, y_min, x_max, y_max;
  float confidence;
  int32_t label;          /* ObjectClass, object_detection.hpp:12-25 */
} gv_bbox;

/* LShapePose  include/grid_vision/cloud_detections.hpp:19-25 (80 bytes):
 * geometry_msgs/Pose (position xyz, orientation xyzw) + length, width, height */
typedef struct {
  double px, py, pz;
  double qx, qy, qz, qw;
  double length, width, height;
} gv_lshape_pose;

/* geometry_msgs/Transform as tf2_ros::Buffer::lookupTransform returns it
 * (src/grid_vision_node.cpp:290,348,371) */
typedef struct {
  double qx, qy, qz, qw;
  double tx, ty, tz;
} gv_transform;

/* CAMParams  include/grid_vision/vision_orientation.hpp:18-25 */
typedef struct {
  int32_t network_h, network_w, orig_h, orig_w;
  float fx, fy, cx, cy;
} gv_cam_params;

/* nav_msgs/OccupancyGrid.info as GridMapRosConverter::toOccupancyGrid fills it */
typedef struct {
  uint32_t width, height;     /* size(0), size(1) */
  double resolution;
  double origin_x, origin_y;  /* position - length/2 */
} gv_grid_info;

/* ------------------------------------------------------------ lifecycle -- */
/* Replaces OccupancyGridMap::OccupancyGridMap(base_link, uint8_t grid_x,
 * uint8_t grid_y, double resolution)  include/grid_vision/occupancy_grid.hpp:16,
 * src/occupancy_grid.cpp:4-14, plus object_detection::setIntrinsicMatrix /
 * computeKInverse (src/object_detection.cpp:241-249) from cam.  device_id < 0
 * picks the current device.  The call also makes sure the handle's upload stream has a hardware queue of its own
 * (a process gets four; profiles/r03/h2d_notes.md 6): three 150 us idle kernels + a timed 4-byte memset, only the
 * handle's own streams are waited for; 0.3-1 ms per handle, GV_QUEUE_PROBE=0 in the environment skips it. */
int gv_create(gv_handle *out, uint8_t grid_x, uint8_t grid_y, double resolution,
              const gv_cam_params *cam, int device_id);
int gv_destroy(gv_handle h);
/* text of the last HIP/RCCL failure on this handle ("" if none) */
const char *gv_last_error(gv_handle h);
/* ABI version of the library (this header: 4) */
int gv_abi_version(void);
/* geometry read-back: nx, ny, pos_x, pos_y (grid_map size / position) */
int gv_grid_geometry(gv_handle h, int32_t *nx, int32_t *ny, double *pos_x, double *pos_y);
/* Reset both layers to the constructor state (log_odds 0.0, occupancy 0.5). */
int gv_reset(gv_handle h);

/* Replaces the three tf lookups of the node: camera<-lidar
 * (grid_vision_node.cpp:290), base<-camera (:348,:371) and, [EXTENSION] for X1/X2,
 * base<-lidar.  A NULL pointer leaves that transform unset; calls that need it
 * then return GV_ERR_TF. */
int gv_set_transforms(gv_handle h, const gv_transform *camera_from_lidar,
                      const gv_transform *base_from_camera, const gv_transform *base_from_lidar);

/* ---------------------------------------------------------------- cloud -- */
/* Replaces GridVision::cloudCallback's pcl::fromROSMsg (grid_vision_node.cpp:103-106)
 * for a caller that already holds SoA x/y/z (fp32, lidar frame).  The cloud
 * stays resident in HBM until the next upload. */
int gv_cloud_upload_xyz(gv_handle h, const float *x, const float *y, const float *z, size_t n);
/* Same, from sensor_msgs/PointCloud2 bytes: n points of point_step bytes with
 * fp32 fields at off_x/off_y/off_z; de-interleaved on the device. */
int gv_cloud_upload_pointcloud2(gv_handle h, const uint8_t *data, size_t n, uint32_t point_step,
                                uint32_t off_x, uint32_t off_y, uint32_t off_z);
/* Streaming ingest: the node receives a new cloud (cloudCallback, grid_vision_node.cpp:103-106)
 * while the previous frame is still being processed (timerCallback, :108-244).  The handle
 * keeps THREE resident clouds in rotation: the *_async calls enqueue the host-to-device copy of the next one
 * on a copy stream, ordered after the last frame that reads the buffer being replaced, and return
 * at once; frames enqueued afterwards use the new cloud (stream-ordered, no host wait).  The copy
 * is truly asynchronous when the host buffers are pinned (gv_host_alloc); pageable buffers work
 * too (the HIP runtime then stages them before returning).  The host buffers must stay unchanged
 * until gv_cloud_upload_wait (or gv_synchronize) returns.  The synchronous gv_cloud_upload_*
 * calls above are the same upload followed by that wait; neither kind drains the frame pipeline. */
int gv_cloud_upload_xyz_async(gv_handle h, const float *x, const float *y, const float *z, size_t n);
int gv_cloud_upload_pointcloud2_async(gv_handle h, const uint8_t *data, size_t n, uint32_t point_step,
                                      uint32_t off_x, uint32_t off_y, uint32_t off_z);
/* wait until every upload enqueued so far has left the host buffers (the clouds' own completion events: frames
 * that run on the upload stream -- the third lane -- are not waited for) */
int gv_cloud_upload_wait(gv_handle h);
/* page-locked host memory for the *_async uploads (hipHostMalloc / hipHostFree) */
int gv_host_alloc(void **ptr, size_t bytes);
int gv_host_free(void *ptr);
/* Replaces GridVision::transformLidarToCamera (grid_vision_node.cpp:280-307,
 * include/grid_vision/grid_vision_node.hpp:95-97): camera-frame copy of the
 * resident cloud written to caller SoA buffers (each n floats). */
int gv_transform_lidar_to_camera(gv_handle h, float *x_cam, float *y_cam, float *z_cam);

/* ------------------------------------------------------ cloud_detections -- */
/* Replaces cloud_detections::extractCloudPerBBox (cloud_detections.hpp:46-48,
 * src/cloud_detections.cpp:250-298) on the resident cloud: bbox_id[i] is the
 * index of the first bbox containing the projection of point i, or -1.
 * counts (optional, nb ints) receives the per-bbox point counts. */
int gv_extract_cloud_per_bbox(gv_handle h, const gv_bbox *bboxes, int32_t nb,
                              int32_t *bbox_id, int32_t *counts);
/* Replaces cloud_detections::buildKDTree + computeDepthForBoundingBoxes
 * (cloud_detections.hpp:29-35, src/cloud_detections.cpp:8-87): exact k nearest
 * projected points of each bbox centre in (u,v,depth), upper median depth;
 * depths[i] = -1 when no point qualifies.  knn_d2 (optional, nb*k) receives the
 * sorted squared distances.  1 <= k <= 32. */
int gv_compute_depth_for_bboxes(gv_handle h, const gv_bbox *bboxes, int32_t nb, int32_t k,
                                float *depths, float *knn_d2);
/* Replaces GridVision::convertPixelsTo3D -> cloud_detections::pixelTo3D ->
 * transformPointToBaseFrame (grid_vision_node.cpp:309-359,
 * src/cloud_detections.cpp:89-103): base-frame points, 3 doubles per bbox. */
int gv_convert_pixels_to_3d(gv_handle h, const gv_bbox *bboxes, const float *depths, int32_t nb,
                            double *base_points_xyz);
/* Replaces cloud_detections::computeBBoxPose without the RANSAC ground removal
 * (cloud_detections.hpp:50-52, src/cloud_detections.cpp:140-247,300-321; see
 * DESIGN.md): extractCloudPerBBox + RadiusOutlierRemoval(0.4, 10) + centroid +
 * PCA rectangle per bbox.  poses_out holds nb entries; valid[i] = 0 where the
 * reference would have skipped the bbox (empty cloud, :174-175). */
int gv_compute_bbox_pose(gv_handle h, const gv_bbox *bboxes, int32_t nb,
                         gv_lshape_pose *poses_out, uint8_t *valid);

/* Replaces cloud_detections::segmentGroundPlane (cloud_detections.hpp:40-41,
 * src/cloud_detections.cpp:105-138: pcl SACSegmentation, plane, RANSAC, threshold 0.04,
 * optimised coefficients) on the camera-frame view of the resident cloud.  PCL's sample
 * sequence cannot be reproduced, so this is specified BY OUTCOME:
 * `iterations` hypotheses from a counter-based RNG (seed), inliers |n.p+d| < threshold
 * counted on the device, least-squares refinement, inliers re-selected.  is_ground
 * (optional, n bytes) marks the plane's points; *n_inliers == 0 means "could not
 * estimate a planar model" (the reference then returns an empty cloud, :122-126). */
int gv_segment_ground_plane(gv_handle h, double threshold, int32_t iterations, uint64_t seed,
                            uint8_t *is_ground, float coeff[4], int64_t *n_inliers);
/* Replaces cloud_detections::computeBBoxPose in full (src/cloud_detections.cpp:300-321):
 * segmentGroundPlane(0.04, 50 iterations) -> extractCloudPerBBox -> bboxPoseEstimation.
 * *n_poses = number of valid poses, or -1 where the reference returns {} (empty segmented cloud). */
int gv_compute_bbox_pose_ground_removed(gv_handle h, const gv_bbox *bboxes, int32_t nb,
                                        gv_lshape_pose *poses_out, uint8_t *valid, int32_t *n_poses);

/* ---------------------------------------------------- vision_orientation -- */
/* Replaces VisionOrientation::postProcessOutputs (+ computeAlpha,
 * computeThetaRay, calcLocation; vision_orientation.hpp:90-98,
 * src/vision_orientation.cpp:241-519) on precomputed network outputs
 * orient[nb*4], conf[nb*2], dims[nb*3].  Writes *n_out <= nb camera-frame poses
 * (unknown classes are skipped, :496-499). */
int gv_vision_post_process(gv_handle h, const float *orient, const float *conf, const float *dims,
                           const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out,
                           int32_t *n_out);
/* Replaces GridVision::transformLShapeObjects (grid_vision_node.cpp:525-531,
 * :361-382): pose camera -> base, in place. */
int gv_transform_lshape_objects(gv_handle h, gv_lshape_pose *poses, int32_t n);

/* ------------------------------------------------------ object_detection -- */
/* Host-side post-processing on precomputed detector outputs; no GPU work.
 * Replaces object_detection::extract_bboxes (object_detection.hpp:48-49,
 * src/object_detection.cpp:94-146) incl. fast_non_max_suppression (:166-211),
 * denormalizeAndScaleBoundingBox (:226-239), getObjectClass (:252-269).
 * boxes[n*4], scores[n*c]; out must hold n entries; returns count in *n_out. */
int gv_extract_bboxes(const float *boxes, const float *scores, int32_t n, int32_t c,
                      double conf_threshold, double iou_threshold, int32_t orig_w, int32_t orig_h,
                      int32_t resize, gv_bbox *out, int32_t *n_out);
/* Replaces GridVision::filterBBoxes (grid_vision_node.cpp:384-403). */
int gv_filter_bboxes(const gv_bbox *in, int32_t n, gv_bbox *static_out, int32_t *n_static,
                     gv_bbox *dynamic_out, int32_t *n_dynamic);
/* Replaces setIntrinsicMatrix / computeKInverse (src/object_detection.cpp:241-249):
 * row-major 3x3 K and K^-1 of the handle. */
int gv_get_intrinsics(gv_handle h, double K[9], double K_inv[9]);

/* -------------------------------------------------------- occupancy grid -- */
/* Replaces OccupancyGridMap::updateMap(GridMap&)  occupancy_grid.hpp:20,
 * src/occupancy_grid.cpp:16-31 */
int gv_update_map(gv_handle h);
/* Replaces OccupancyGridMap::updateMap(GridMap&, vector<LShapePose>)
 * occupancy_grid.hpp:19, src/occupancy_grid.cpp:65-105,140-183 (poses in the
 * base frame) */
int gv_update_map_poses(gv_handle h, const gv_lshape_pose *poses, int32_t n);
/* Replaces OccupancyGridMap::updateMap(GridMap&, vector<Point>, vector<BoundingBox>)
 * occupancy_grid.hpp:17-18, src/occupancy_grid.cpp:33-63,107-138,185-196
 * (never called by the node; kept for the class surface) */
int gv_update_map_points(gv_handle h, const double *base_points_xyz, const gv_bbox *bboxes,
                         int32_t n);
/* Replaces GridVision::publishOccupancyGrid's
 * GridMapRosConverter::toOccupancyGrid(map,"occupancy",0,1,msg)
 * (grid_vision_node.cpp:265-278): data[G] int8 in OccupancyGrid order. */
int gv_to_occupancy_grid(gv_handle h, int8_t *data, gv_grid_info *info);
/* The same without stalling the frame pipeline: an asynchronous device-to-host copy of data[G] on
 * gv_stream(h), behind the grid pass of the last enqueued frame and ahead of the next one's.  data
 * should be pinned (gv_host_alloc); it is complete once an event recorded on gv_stream(h) after this
 * call has passed, or after gv_synchronize. */
int gv_to_occupancy_grid_async(gv_handle h, int8_t *data);
/* The same for a node that publishes the grid EVERY tick while clouds stream in (the reference does:
 * grid_vision_node.cpp:240, :265-278).  data must be pinned (gv_host_alloc): the grid is then written there by a small
 * kernel on gv_stream(h) instead of a copy command, so the copy engines stay with the cloud uploads and PCIe carries
 * both directions at once -- 262 us per frame, steady, against 292-349 us and erratic for the copy command beside an
 * upload (profiles/r04/publish_variants.txt).  Pageable memory, or a destination that is not 16-byte aligned, falls back to the
 * copy command.  Same completion rule as
 * gv_to_occupancy_grid_async. */
int gv_publish_grid_async(gv_handle h, int8_t *data);
/* Layer read-back (grid_map_["log_odds"], ["occupancy"]; occupancy_grid.hpp:22) */
int gv_get_log_odds(gv_handle h, float *out);
int gv_get_occupancy(gv_handle h, float *out);
/* Layer write (tests / checkpoint restore): G floats */
int gv_set_log_odds(gv_handle h, const float *in);

/* ------------------------------------------------------ [EXTENSION] frame -- */
/* One fused per-frame pass over the resident cloud (SURVEY rows X1, X2, A5, A8,
 * A7, A18):  bin points into hit counts, ray-march free space from the sensor
 * origin, first-match bbox id per point, then one grid pass: decay, rectangle
 * adds, hit/miss rule, clamp, sigmoid, int8 pack. */
enum {
  GV_FRAME_BIN       = 1 << 0,   /* X1: hits, optional cell_idx                 */
  GV_FRAME_RAYMARCH  = 1 << 1,   /* X2: miss (needs GV_FRAME_BIN)               */
  GV_FRAME_BBOX_TEST = 1 << 2,   /* A5: bbox_id per point                       */
  GV_FRAME_KEEP_CELL_IDX = 1 << 3,  /* write cell_idx[N] (debug/parity output)  */
  GV_FRAME_KEEP_COUNTS   = 1 << 4,  /* keep hits/miss of this frame for getters */
  GV_FRAME_VISION_ORIENT = 1 << 5   /* poses come from net outputs (A13/A14/A15)
                                       instead of base-frame poses             */
};
typedef struct {
  uint32_t flags;
  const gv_bbox *bboxes;          /* nb bboxes (GV_FRAME_BBOX_TEST / VISION_ORIENT) */
  int32_t n_bboxes;
  const gv_lshape_pose *poses;    /* base-frame poses for the rectangle adds     */
  int32_t n_poses;
  const float *orient, *conf, *dims;  /* GV_FRAME_VISION_ORIENT: nb*4, nb*2, nb*3 */
} gv_frame_desc;
/* Upload the small per-frame detection inputs (bboxes, poses / net outputs).  Two sets alternate: the
 * arrays are copied into pinned staging (the caller's arrays are free on return), go to the device in
 * one copy on the stream of the frame that reads them first -- in order ahead of it -- and are turned
 * into the bbox-test tables there.
 * Neither form waits for the copy or drains the frame pipeline (the _async name is kept for symmetry
 * with the cloud uploads).  The standalone entry points above (gv_extract_cloud_per_bbox,
 * gv_update_map_poses, ...) keep their inputs in a set of their own and never change what
 * gv_frame_enqueue uses. */
int gv_frame_set_detections(gv_handle h, const gv_frame_desc *desc);
int gv_frame_set_detections_async(gv_handle h, const gv_frame_desc *desc);
/* Enqueue one frame using the resident cloud and the last detections set (asynchronous).
 * GV_ERR_STATE before the first gv_frame_set_detections.  Two or three frames run side by side (the third lane is
 * the upload stream while no cloud has been uploaded for a while; GV_LANES=2: never): binning and ray stage of
 * frame f on an internal stream, its grid pass on gv_stream(h) behind them, so
 * the grid passes -- and anything the caller puts on gv_stream(h) between two frames -- execute in
 * enqueue order and see every result of the frames before them.  At most SIX frames are in flight (four
 * with two lanes): the call waits on the host for the frame six back when the caller runs further ahead. */
int gv_frame_enqueue(gv_handle h);
/* Make gv_stream(h) wait (on the device, not the host) for every upload enqueued so far as well
 * (frames are ordered on gv_stream(h) by construction): afterwards an event recorded or a kernel
 * launched there sees the results of everything enqueued on the handle. */
int gv_frame_fence(gv_handle h);
/* Wait (host) for everything enqueued on the handle. */
int gv_synchronize(gv_handle h);
/* gv_frame_set_detections + gv_frame_enqueue + gv_synchronize */
int gv_process_frame(gv_handle h, const gv_frame_desc *desc);
/* Per-frame outputs of the last frame (need GV_FRAME_KEEP_* where noted). */
int gv_get_hits(gv_handle h, int32_t *out);          /* G ints; any BIN frame (generic grids: KEEP_COUNTS) */
int gv_get_miss(gv_handle h, int32_t *out);          /* G ints in {0,1}, KEEP_COUNTS */
int gv_get_cell_idx(gv_handle h, int32_t *out);      /* N ints, KEEP_CELL_IDX */
int gv_get_bbox_id(gv_handle h, int32_t *out);       /* N ints, BBOX_TEST     */
/* number of grid cells visited by the last ray-march (sum over marched rays) */
int gv_get_ray_stats(gv_handle h, uint64_t *n_rays, uint64_t *n_visits);

/* ------------------------------------------------------------ the node's tick -- */
/* Replaces GridVision::timerCallback from filterBBoxes to publishOccupancyGrid (src/grid_vision_node.cpp:153-244) as
 * ONE batch of device work with ONE host wait:
 *   filterBBoxes (:384-403, host)                                   -> static / dynamic boxes
 *   static boxes: buildKDTree + computeDepthForBoundingBoxes (:168-184, cloud_detections.cpp:8-87) + convertPixelsTo3D
 *   dynamic boxes, GV_TICK_VISION_ORIENT: VisionOrientation::postProcessOutputs on the network outputs (:190-209)
 *                  otherwise:             cloud_detections::computeBBoxPose on ALL boxes (:210-231, cloud_detections.cpp:300-321)
 *   transformLShapeObjects (:204, :227), updateMap(grid, poses) / updateMap(grid) (:145, :206, :230, :235),
 *   GridMapRosConverter::toOccupancyGrid (:265-278).
 * The poses go from the kernel that computes them through the camera->base transform and the rectangle kernel into the
 * grid pass without leaving the device; the depths, the poses (for the markers, :243) and optionally the packed grid
 * come back through pinned memory and are complete when gv_tick_wait returns.  The static boxes' kNN runs on a second
 * stream beside the pose branch.  The caller runs extract_bboxes and, for GV_TICK_VISION_ORIENT, the orientation
 * network on the dynamic boxes (gv_filter_bboxes gives their order) first; a tick with no boxes is the :141-147 path.
 * GV_ERR_TF when a transform the tick needs was never set (the node publishes the stale grid, :160-164).
 * One tick may be pending per handle; frames in flight (gv_frame_enqueue) are drained first. */
enum {
  GV_TICK_VISION_ORIENT  = 1 << 0,  /* use_vision_orientation (config/grid_vision_cfg.yaml:24) */
  GV_TICK_LIDAR_BIN      = 1 << 1,  /* [EXTENSION] the map update also bins the resident cloud (X1) ...  */
  GV_TICK_LIDAR_RAYMARCH = 1 << 2   /* [EXTENSION] ... and marks free space (X2); tile-path grids only   */
};
typedef struct {
  uint32_t flags;
  const gv_bbox *bboxes;             /* what extract_bboxes returned (:138-139): static and dynamic mixed */
  int32_t n_bboxes;
  const float *orient, *conf, *dims; /* GV_TICK_VISION_ORIENT: n_net * 4 / 2 / 3 network outputs, dynamic-box order */
  int32_t n_net;                     /* 0 (no poses this tick, the reference's empty vector) or the number of dynamic boxes */
  int32_t k_near;                    /* k_near (grid_vision_cfg.yaml:20), 1..32 */
  int8_t *grid_out;                  /* optional: G bytes (gv_host_alloc for a true DMA) receive OccupancyGrid.data */
} gv_tick_desc;
typedef struct {
  int32_t n_static, n_dynamic;       /* out */
  gv_bbox *static_bboxes;            /* optional, room for n_bboxes: the static boxes (marker labels, :413-480)      */
  float *depths;                     /* optional, room for n_bboxes: depth of every static box (:176-177)            */
  double *base_points_xyz;           /* optional, room for 3 * n_bboxes: their base-frame points (:180)              */
  gv_lshape_pose *poses;             /* optional, room for n_bboxes: the dynamic objects' base-frame poses           */
  int32_t n_poses;                   /* out */
  int32_t pca_empty;                 /* out: 1 = computeBBoxPose returned {} (no plane / empty segmented cloud)      */
} gv_tick_result;
/* One tick at a time (the reference's timer is single threaded): a second gv_tick_enqueue before gv_tick_wait returns
 * GV_ERR_STATE, and so do, between the two, the synchronous calls that would reuse the tick's result block or its
 * detection set (gv_compute_depth_for_bboxes, gv_compute_bbox_pose*, gv_segment_ground_plane, gv_extract_cloud_per_bbox,
 * ...).  Cloud uploads, gv_frame_* and the grid getters may be called; they are ordered behind the tick on gv_stream(h). */
int gv_tick_enqueue(gv_handle h, const gv_tick_desc *d);
int gv_tick_wait(gv_handle h, gv_tick_result *r);
int gv_tick(gv_handle h, const gv_tick_desc *d, gv_tick_result *r);   /* = enqueue + wait */

/* ------------------------------------------------ raw stream / timing hooks -- */
/* The public HIP stream of the handle (hipStream_t as void*), for callers that record their own
 * events around gv_frame_enqueue or consume the grid layers on the device: every frame's grid pass
 * runs on it, behind the frame's other kernels. */
void *gv_stream(gv_handle h);
/* Device pointers of the resident grid for consumers on the device (stream-ordered behind a frame on gv_stream):
 * the packed OccupancyGrid.data bytes (G int8, `OccupancyGrid.data` order) and the two float layers (G floats,
 * grid_map order).  Any of the three may be null.  Read-only for the caller. */
int gv_device_layers(gv_handle h, int8_t **occ_i8, float **log_odds, float **occupancy);
/* Time `frames` back-to-back gv_frame_enqueue calls with HIP events on the
 * handle's stream; *ms_total is the elapsed device time. */
int gv_time_frames(gv_handle h, int32_t frames, float *ms_total);
/* Per-stage device time of one frame, averaged over `frames` serial frames.  On the tile path
 * the four kernels carry their own start / end events (dispatch-packet timestamps); other stages are
 * intervals between HIP events recorded on the handle's stream.
 * stage_ms has GV_NUM_STAGES entries. */
enum {
  GV_STAGE_DETECTIONS = 0,   /* vision-orientation geometry + rectangles       */
  GV_STAGE_POINTS = 1,       /* transform + cell index + ray ends + bbox test, keys partitioned by tile */
  GV_STAGE_RAY_COMPACT = 2,  /* per-tile hit histogram -> hits[] + ray-end bitmaps */
  GV_STAGE_RAY_MARCH = 3,    /* Bresenham free-space march                     */
  GV_STAGE_FINALIZE = 4,     /* decay/rect/hit-miss/clamp/sigmoid/int8 pass    */
  GV_NUM_STAGES = 5
};
int gv_time_frame_stages(gv_handle h, int32_t frames, float *stage_ms);

/* ---------------------------------------- [EXTENSION] multi-GPU (RCCL/xGMI) -- */
/* One large frame sharded by POINTS over `world` ranks: every rank bins its
 * slice into private ray-end bitmaps; the bitmaps are OR-ed across ranks (all-to-all of slices +
 * local OR + all-gather); every rank runs every world-th workgroup of the ray stage; the free-cell
 * bitmaps are OR-ed by row band (rank r receives band r), each rank finalises its band and the
 * packed int8 bands are broadcast.  After the frame log_odds/occupancy are valid for the rank's own
 * band only (gv_comm_band), the int8 grid everywhere.  With GV_FRAME_KEEP_COUNTS the hit counts are
 * reduce-scattered by band: gv_get_hits then holds the SUMMED counts at the rank's band
 * and this rank's partial counts elsewhere; gv_get_miss returns GV_ERR_STATE.
 * gv_comm_unique_id fills a 128-byte RCCL id on rank 0 (broadcast it by any
 * means); gv_comm_init joins the communicator (and creates the handle's exchange stream). */
int gv_comm_unique_id(uint8_t id_out[128]);
int gv_comm_init(gv_handle h, const uint8_t id[128], int32_t rank, int32_t world);
int gv_comm_destroy(gv_handle h);
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): the number of
 * ranks that joined, this rank, and the device it runs on -- bench.py prints them, so that a scaling line proves that
 * N ranks on N devices took part.  GV_ERR_STATE before gv_comm_init. */
int gv_comm_info(gv_handle h, int32_t *n_ranks, int32_t *rank, int32_t *device);
/* Sharded frame, asynchronous: the counterpart of gv_frame_enqueue (same detection sets, same back-pressure of
 * four frames in flight on two lanes, results on the public stream behind it) for a resident cloud that is this rank's
 * slice.  Binning, sector share and band packing run on the frame's lane, the RCCL exchanges on the handle's
 * exchange stream, the band's grid pass on the public stream: frame f's exchanges overlap frame f+1's binning.
 * Collective: every rank of the communicator must enqueue the same sequence of frames. */
int gv_frame_enqueue_sharded(gv_handle h);
/* = gv_frame_set_detections + gv_frame_enqueue_sharded + gv_synchronize */
int gv_process_frame_sharded(gv_handle h, const gv_frame_desc *desc);
/* Device time of the six steps of the sharded frame -- binning, ends exchange, sector share + packing, free-band
 * exchange, band grid pass, band broadcast (+ count reduce) -- averaged over `frames` frames run one at a time.
 * Collective. */
int gv_time_frame_sharded_stages(gv_handle h, int32_t frames, float stage_ms[6]);
/* Pure host helpers (no handle, no GPU): rows [*y0, *y1) rank `rank` of `world` finalises in a grid of `ny` rows
 * (whole 64-row blocks of the grid padded to 128 rows, clipped to ny), and the words of one of the `world`
 * equal slices a bitmap of `words` words is exchanged in.  The multi-rank CPU tests use the same functions. */
int gv_shard_band_rows(int32_t rank, int32_t world, int32_t ny, int32_t *y0, int32_t *y1);
int64_t gv_shard_slice_words(int64_t words, int32_t world);
/* Band of cells [begin,end) this rank finalises (linear cell indices): whole 64-row blocks. */
int gv_comm_band(gv_handle h, int64_t *begin, int64_t *end);
#ifdef __cplusplus
}
#endif
#endif /* GRIDVISION_HIP_H_ */
