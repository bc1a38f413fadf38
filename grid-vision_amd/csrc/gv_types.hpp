// gv_types.hpp -- POD parameter blocks shared by the host side (gv_api.hip)
// and the gfx950 kernels (gv_kernels.hip).  Passed by value as kernel arguments.
#pragma once

#include <stdint.h>

#include "../../include/gridvision_hip.h"

namespace gv {

// Geometry of the resident grid (grid_map conventions, src/occupancy_grid.cpp:4-14).
// Layers are stored exactly like grid_map's column-major MatrixXf(size0,size1):
// linear cell = iy*nx + ix, ix = x index (fastest), (0,0) = +x,+y corner.
struct GridParams {
  int32_t nx, ny, G;
  double res;
  double len_x, len_y;   // size * res
  double pos_x, pos_y;   // map centre
  double off_x, off_y;   // 0.5 * len
  double inv_res;        // fl64(1 / res): quotient estimate of the points pass (never the result itself)
};

// Row-major 3x4 fp32 rigid transform (the top of PCL's 4x4).
struct Mat34f {
  float m[12];
};

// fp64 rigid transform as tf2 applies it (basis rows + origin).
struct Xform64 {
  double b[9];
  double o[3];
};

struct CamK {
  double k[9];   // row-major K
  int32_t W, H;  // image width / height
};

// sensor origin for the X2 ray-march
struct RayOrigin {
  double ox, oy;      // base-frame position of the lidar origin
  int32_t cx, cy;     // its cell
  int32_t valid;      // 0: origin outside the map -> no rays this frame
};

// Inclusive index rectangle of one object (updateGridCellsFast block).
struct Rect {
  int32_t x0, y0, x1, y1;
  int32_t valid;
};

// device-side result of the vision-orientation geometry for one bbox
struct VisionOut {
  float loc[3];
  float orient;     // alpha + theta_ray
  float err;
  float dims[3];    // length, width, height (fp32 sums, :474-476)
  int32_t valid;    // 0 for classes the reference skips (:496-499)
};

// reference constants, include/grid_vision/occupancy_grid.hpp:25-31, occupancy_grid.cpp:182
constexpr float kLogOddsFree = -0.4f;
constexpr float kLogOddsOccupied = 1.2f;
constexpr float kLogOddsPrior = 0.0f;
constexpr float kInitProbability = 0.5f;
constexpr float kLogOddsDecay = -0.2f;
constexpr float kMinLogOdds = -2.0f;
constexpr float kMaxLogOdds = 3.6f;
constexpr float kRectIncrement = 0.85f;

enum Stage : int {
  kStageDetections = 0,  // vision-orientation geometry + rectangles
  kStagePoints = 1,      // transform + bin + ray ends + bbox test
  kStageRayCompact = 2,
  kStageRayMarch = 3,
  kStageFinalize = 4,
  kNumStages = 5
};

}  // namespace gv
