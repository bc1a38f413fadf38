/*
 * oracle/grid.c -- CPU ORACLE (test infrastructure; see gv_oracle.h header).
 * PARITY UNPINNED: follows src/occupancy_grid.cpp line by line plus
 * [UPSTREAM-RECALL] grid_map_core / grid_map_ros behaviour.
 */
#include "gv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* OccupancyGridMap::OccupancyGridMap  src/occupancy_grid.cpp:4-14 */
int gvo_grid_init(gvo_grid *g, uint8_t grid_x, uint8_t grid_y, double resolution)
{
  memset(g, 0, sizeof(*g));
  if (!(resolution > 0.0) || grid_x == 0 || grid_y == 0) return -1;
  /* :10 setGeometry(Length(grid_x, grid_y), resolution)
   * [UPSTREAM-RECALL] GridMap::setGeometry: size = (int)round(length/res);
   * length_ = size * res; startIndex = 0. */
  g->res = resolution;
  g->nx = (int32_t)round((double)grid_x / resolution);
  g->ny = (int32_t)round((double)grid_y / resolution);
  if (g->nx <= 0 || g->ny <= 0) return -1;
  g->len_x = (double)g->nx * resolution;
  g->len_y = (double)g->ny * resolution;
  /* :11 setPosition(Position(grid_x / 3, 0.0)) -- uint8_t/int => INTEGER division */
  g->pos_x = (double)(grid_x / 3);
  g->pos_y = 0.0;
  size_t G = (size_t)g->nx * (size_t)g->ny;
  g->log_odds = (float *)malloc(G * sizeof(float));
  g->occupancy = (float *)malloc(G * sizeof(float));
  if (!g->log_odds || !g->occupancy) { gvo_grid_free(g); return -2; }
  for (size_t i = 0; i < G; ++i) {
    g->log_odds[i] = GVO_LOG_ODDS_PRIOR;     /* :12 */
    g->occupancy[i] = GVO_INIT_PROBABILITY;  /* :13 */
  }
  return 0;
}

void gvo_grid_free(gvo_grid *g)
{
  free(g->log_odds);
  free(g->occupancy);
  g->log_odds = g->occupancy = NULL;
}

/* grid_map::GridMap::getIndex -> getIndexFromPosition [UPSTREAM-RECALL]
 *   offset      = 0.5 * mapLength
 *   indexVector = (position - offset - mapPosition) / resolution      (fp64)
 *   index       = (int)(-indexVector)     (truncation; startIndex == 0)
 *   inside  iff  t = -(position - mapPosition - offset);  0 <= t < length
 * so index (0,0) is the +x,+y corner, the + edge is inside, the - edge is not. */
int gvo_get_index(const gvo_grid *g, double x, double y, int32_t *ix, int32_t *iy)
{
  *ix = -1; *iy = -1;
  if (!isfinite(x) || !isfinite(y)) return 0;
  const double offx = 0.5 * g->len_x, offy = 0.5 * g->len_y;
  const double tx = -((x - g->pos_x) - offx);
  const double ty = -((y - g->pos_y) - offy);
  if (!(tx >= 0.0 && ty >= 0.0 && tx < g->len_x && ty < g->len_y)) return 0;
  const double vx = ((x - offx) - g->pos_x) / g->res;
  const double vy = ((y - offy) - g->pos_y) / g->res;
  const int32_t jx = (int32_t)(-vx);
  const int32_t jy = (int32_t)(-vy);
  /* defensive range check (unreachable for in-map positions) */
  if (jx < 0 || jy < 0 || jx >= g->nx || jy >= g->ny) return 0;
  *ix = jx; *iy = jy;
  return 1;
}

/* :19 / :37 / :69  grid_map["log_odds"].array() += log_odds_decay_ */
void gvo_decay(gvo_grid *g)
{
  const size_t G = (size_t)g->nx * (size_t)g->ny;
  for (size_t i = 0; i < G; ++i) g->log_odds[i] = g->log_odds[i] + GVO_LOG_ODDS_DECAY;
}

/* :21-30  cwiseMax(min).cwiseMin(max), then p = 1/(1+exp(-l)) per cell */
void gvo_clamp_and_sigmoid(gvo_grid *g)
{
  const size_t G = (size_t)g->nx * (size_t)g->ny;
  for (size_t i = 0; i < G; ++i) {
    float l = g->log_odds[i];
    l = (l < GVO_MIN_LOG_ODDS) ? GVO_MIN_LOG_ODDS : l;   /* cwiseMax(min_log_odds_) */
    l = (l > GVO_MAX_LOG_ODDS) ? GVO_MAX_LOG_ODDS : l;   /* cwiseMin(max_log_odds_) */
    g->log_odds[i] = l;
    g->occupancy[i] = 1.0f / (1.0f + expf(-l));          /* std::exp(float) */
  }
}

/* updateMap(GridMap&)  src/occupancy_grid.cpp:16-31 */
void gvo_update_map(gvo_grid *g)
{
  gvo_decay(g);
  gvo_clamp_and_sigmoid(g);
}

/* updateGridCellsFast  src/occupancy_grid.cpp:140-183 */
int gvo_update_grid_cells_fast(gvo_grid *g, const double c[8])
{
  int32_t minx = 0, miny = 0, maxx = 0, maxy = 0;
  for (int i = 0; i < 4; ++i) {
    int32_t ix, iy;
    if (!gvo_get_index(g, c[2 * i], c[2 * i + 1], &ix, &iy)) return 0;  /* :152-156,171-172 */
    if (i == 0) { minx = maxx = ix; miny = maxy = iy; }
    else {
      if (ix < minx) minx = ix;
      if (iy < miny) miny = iy;
      if (ix > maxx) maxx = ix;
      if (iy > maxy) maxy = iy;
    }
  }
  /* :175-182  block(min.x, min.y, dx+1, dy+1).array() += 0.85f */
  for (int32_t iy = miny; iy <= maxy; ++iy)
    for (int32_t ix = minx; ix <= maxx; ++ix) {
      size_t k = (size_t)iy * (size_t)g->nx + (size_t)ix;
      g->log_odds[k] = g->log_odds[k] + GVO_RECT_INCREMENT;
    }
  return 1;
}

/* corners of one LShapePose, src/occupancy_grid.cpp:79-90,
 * order {left_back, left_front, right_front, right_back} */
static void pose_corners(const gvo_lshape *p, double c[8])
{
  const double hx = p->length / 2.0, hy = p->width / 2.0;
  c[0] = p->px - hx; c[1] = p->py - hy;   /* left_back   :83-84 */
  c[2] = p->px + hx; c[3] = p->py - hy;   /* left_front  :79-80 */
  c[4] = p->px + hx; c[5] = p->py + hy;   /* right_front :81-82 */
  c[6] = p->px - hx; c[7] = p->py + hy;   /* right_back  :85-86 */
}

/* updateMap(GridMap&, vector<LShapePose>)  src/occupancy_grid.cpp:65-105 */
void gvo_update_map_poses(gvo_grid *g, const gvo_lshape *poses, int32_t n)
{
  gvo_decay(g);
  for (int32_t i = 0; i < n; ++i) {
    double c[8];
    pose_corners(&poses[i], c);
    gvo_update_grid_cells_fast(g, c);
  }
  gvo_clamp_and_sigmoid(g);
}

/* getEstimatedDepth  src/occupancy_grid.cpp:185-196 */
float gvo_estimated_depth(int32_t label)
{
  switch (label) {
  case GVO_VEHICLE: return 3.5f;
  case GVO_PERSON: return 0.6f;
  case GVO_BIKE: return 2.5f;
  case GVO_MOTORBIKE: return 2.5f;
  default: return -1.0f;
  }
}

/* computeBoundingBox3D  src/occupancy_grid.cpp:107-138 (x,y only; z is copied) */
void gvo_bounding_box_3d(const double ctr[3], int32_t label, double c[8])
{
  const float d = gvo_estimated_depth(label);
  c[0] = ctr[0] + d; c[1] = ctr[1] + (d / 2);   /* left-front  :118-119 */
  c[2] = ctr[0] + d; c[3] = ctr[1] - (d / 2);   /* right-front :123-124 */
  c[4] = ctr[0];     c[5] = ctr[1] - (d / 2);   /* right-back  :128-129 */
  c[6] = ctr[0];     c[7] = ctr[1] + (d / 2);   /* left-back   :133-134 */
}

/* updateMap(GridMap&, vector<Point>, vector<BoundingBox>) :33-63 (never called
 * by the node; part of the class surface) */
void gvo_update_map_points(gvo_grid *g, const double *pts, const gvo_bbox *bboxes, int32_t n)
{
  gvo_decay(g);
  for (int32_t i = 0; i < n; ++i) {
    double c[8];
    gvo_bounding_box_3d(&pts[3 * i], bboxes[i].label, c);
    gvo_update_grid_cells_fast(g, c);
  }
  gvo_clamp_and_sigmoid(g);
}

/* GridMapRosConverter::toOccupancyGrid(map, "occupancy", 0.0, 1.0, msg)
 * [UPSTREAM-RECALL]: width=size0, height=size1, origin = pos - length/2,
 * value = (occ-0)/(1-0); NaN -> -1 else 0 + clamp01(value)*100; stored as int8
 * (truncation) at data[G-1-(iy*size0+ix)]. */
void gvo_to_occupancy_grid(const gvo_grid *g, int8_t *data, double info[5])
{
  const size_t G = (size_t)g->nx * (size_t)g->ny;
  const float dmin = 0.0f, dmax = 1.0f, cmin = 0.0f, cmax = 100.0f;
  const float range = cmax - cmin;
  for (size_t k = 0; k < G; ++k) {
    float v = (g->occupancy[k] - dmin) / (dmax - dmin);
    if (isnan(v)) v = -1.0f;
    else {
      float c = v < 0.0f ? 0.0f : v;
      c = c > 1.0f ? 1.0f : c;
      v = cmin + c * range;
    }
    data[G - 1 - k] = (int8_t)v;
  }
  if (info) {
    info[0] = (double)g->nx;
    info[1] = (double)g->ny;
    info[2] = g->res;
    info[3] = g->pos_x - 0.5 * g->len_x;
    info[4] = g->pos_y - 0.5 * g->len_y;
  }
}

/* [EXTENSION] fused frame update: the reference's decay / rectangle / clamp /
 * sigmoid sequence (:69-104) with the X2 hit/miss rule inserted after the
 * rectangle adds and before the clamp:
 *   hits>0 -> l += log_odds_occupied_ (1.2f) ; else miss>0 -> l += log_odds_free_ (-0.4f) */
void gvo_frame_update(gvo_grid *g, const gvo_lshape *poses, int32_t n_poses,
                      const int32_t *hits, const uint8_t *miss)
{
  gvo_decay(g);
  for (int32_t i = 0; i < n_poses; ++i) {
    double c[8];
    pose_corners(&poses[i], c);
    gvo_update_grid_cells_fast(g, c);
  }
  const size_t G = (size_t)g->nx * (size_t)g->ny;
  if (hits || miss) {
    for (size_t k = 0; k < G; ++k) {
      if (hits && hits[k] > 0) g->log_odds[k] = g->log_odds[k] + GVO_LOG_ODDS_OCCUPIED;
      else if (miss && miss[k] > 0) g->log_odds[k] = g->log_odds[k] + GVO_LOG_ODDS_FREE;
    }
  }
  gvo_clamp_and_sigmoid(g);
}
