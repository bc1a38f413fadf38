/*
 * oracle/extension.c -- CPU ORACLE (test infrastructure; see gv_oracle.h).
 *
 * [EXTENSION] The reference contains NO per-point binning and NO ray-march
 * (SURVEY.md 0.3, rows X1/X2): include/grid_vision/occupancy_grid.hpp:25-26
 * declares log_odds_free_/log_odds_occupied_ and never reads them.  This file
 * is the DEFINITION the HIP kernels are checked against, not a restatement.
 * It reuses the reference's cell convention (getIndex, grid.c), its fp32 rigid
 * transform op order (transforms.c) and grid_map::LineIterator's integer
 * Bresenham stepping [UPSTREAM-RECALL].
 *
 * X1  hits[cell] += 1 for every finite lidar point whose base-frame position
 *     lies inside the map.
 * X2  For every finite point a ray is cast from the sensor-origin cell O (the
 *     cell of the translation of base<-lidar; it must lie inside the map,
 *     otherwise no ray is cast this frame):
 *       - point inside the map: end cell E = its cell; traversed cells are
 *         LineIterator(O,E) WITHOUT E (E is the hit).
 *       - point outside the map: the segment origin->point is clipped to the
 *         map rectangle in fp64 (slab clip, fixed op order below), E = the
 *         clamped floor cell of the clip point; traversed cells are
 *         LineIterator(O,E) INCLUDING E (no hit).
 *     miss[cell] = 1 on every traversed cell.  The update rule is binary per
 *     frame and order independent (gvo_frame_update in grid.c).
 */
#include "gv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline void xf_point(const float m[16], float px, float py, float pz, float o[3])
{
  for (int r = 0; r < 3; ++r) {
    const float p0 = px * m[r * 4 + 0];
    const float p1 = py * m[r * 4 + 1];
    const float p2 = pz * m[r * 4 + 2];
    o[r] = p0 + (p1 + (p2 + m[r * 4 + 3]));
  }
}

void gvo_bin_points(const gvo_grid *g, const float m_base[16], const float *x, const float *y,
                    const float *z, size_t n, int32_t *hits, int32_t *cell_idx)
{
  for (size_t i = 0; i < n; ++i) {
    float b[3];
    xf_point(m_base, x[i], y[i], z[i], b);
    int32_t ix, iy, cell = -1;
    if (isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2])
        && gvo_get_index(g, (double)b[0], (double)b[1], &ix, &iy)) {
      cell = iy * g->nx + ix;
      if (hits) hits[cell] += 1;
    }
    if (cell_idx) cell_idx[i] = cell;
  }
}

int gvo_ray_end(const gvo_grid *g, double ox, double oy, float pxf, float pyf, float pzf,
                int32_t *ex, int32_t *ey)
{
  if (!isfinite(pxf) || !isfinite(pyf) || !isfinite(pzf)) return 0;
  const double px = (double)pxf, py = (double)pyf;
  if (gvo_get_index(g, px, py, ex, ey)) return 1;
  /* slab clip of origin + t*(p - origin), t in [0,1], against the map rectangle
   * x in [hi_x - len_x, hi_x], y likewise; hi = pos + 0.5*len */
  const double offx = 0.5 * g->len_x, offy = 0.5 * g->len_y;
  const double hix = g->pos_x + offx, hiy = g->pos_y + offy;
  const double lox = hix - g->len_x, loy = hiy - g->len_y;
  const double dx = px - ox, dy = py - oy;
  double t = 1.0;
  if (dx > 0.0) { const double tx = (hix - ox) / dx; if (tx < t) t = tx; }
  if (dx < 0.0) { const double tx = (lox - ox) / dx; if (tx < t) t = tx; }
  if (dy > 0.0) { const double ty = (hiy - oy) / dy; if (ty < t) t = ty; }
  if (dy < 0.0) { const double ty = (loy - oy) / dy; if (ty < t) t = ty; }
  if (t < 0.0) t = 0.0;
  const double qx = ox + t * dx;
  const double qy = oy + t * dy;
  double fx = floor(-(((qx - offx) - g->pos_x) / g->res));
  double fy = floor(-(((qy - offy) - g->pos_y) / g->res));
  if (!(fx >= 0.0)) fx = 0.0;
  if (!(fy >= 0.0)) fy = 0.0;
  if (fx > (double)(g->nx - 1)) fx = (double)(g->nx - 1);
  if (fy > (double)(g->ny - 1)) fy = (double)(g->ny - 1);
  *ex = (int32_t)fx;
  *ey = (int32_t)fy;
  return 2;
}

/* grid_map::LineIterator [UPSTREAM-RECALL]: delta = |end-start|; the major axis
 * (x when delta.x >= delta.y) steps every iteration; numerator = major/2;
 * numerator += minor; if (numerator >= major) { numerator -= major; step minor }
 * nCells = major + 1.  Marks the first n_mark cells. */
static uint64_t march(const gvo_grid *g, int32_t sx, int32_t sy, int32_t ex, int32_t ey,
                      int include_end, uint8_t *miss)
{
  const int32_t ddx = abs(ex - sx), ddy = abs(ey - sy);
  const int32_t stepx = (ex >= sx) ? 1 : -1, stepy = (ey >= sy) ? 1 : -1;
  int32_t inc1x = stepx, inc1y = stepy, inc2x = stepx, inc2y = stepy;
  int32_t den, num, add, ncells;
  if (ddx >= ddy) { inc1x = 0; inc2y = 0; den = ddx; num = ddx / 2; add = ddy; ncells = ddx + 1; }
  else            { inc2x = 0; inc1y = 0; den = ddy; num = ddy / 2; add = ddx; ncells = ddy + 1; }
  const int32_t nmark = include_end ? ncells : ncells - 1;
  int32_t cx = sx, cy = sy;
  for (int32_t i = 0; i < nmark; ++i) {
    miss[(size_t)cy * (size_t)g->nx + (size_t)cx] = 1;
    num += add;
    if (num >= den) { num -= den; cx += inc1x; cy += inc1y; }
    cx += inc2x; cy += inc2y;
  }
  return (uint64_t)(nmark > 0 ? nmark : 0);
}

/* [EXTENSION] X2 for ray ends that are already known (ex, ey, kind as gvo_ray_end returns them; kind 0 entries
 * are skipped): the march of gvo_raymarch without the points.  The multi-rank tests use it: after the first
 * exchange of the sharded frame a rank holds END BITMAPS, not points. */
void gvo_march_ends(const gvo_grid *g, double ox, double oy, const int32_t *ex, const int32_t *ey,
                    const uint8_t *kind, size_t n, uint8_t *miss, uint64_t *visits)
{
  uint64_t v = 0;
  int32_t ocx, ocy;
  if (!gvo_get_index(g, ox, oy, &ocx, &ocy)) { if (visits) *visits = 0; return; }
  for (size_t i = 0; i < n; ++i) {
    if (!kind[i]) continue;
    v += march(g, ocx, ocy, ex[i], ey[i], kind[i] == 2, miss);
  }
  if (visits) *visits = v;
}

void gvo_raymarch(const gvo_grid *g, const float m_base[16], const float *x, const float *y,
                  const float *z, size_t n, uint8_t *miss, int dedupe, uint64_t *visits)
{
  uint64_t v = 0;
  /* sensor origin = image of (0,0,0) = translation column, as fp32 */
  const double ox = (double)m_base[3], oy = (double)m_base[7];
  int32_t ocx, ocy;
  if (!gvo_get_index(g, ox, oy, &ocx, &ocy)) { if (visits) *visits = 0; return; }
  const size_t G = (size_t)g->nx * (size_t)g->ny;
  uint8_t *seen = dedupe ? (uint8_t *)calloc(G, 1) : NULL;   /* bit0 hit-end, bit1 clip-end */
  for (size_t i = 0; i < n; ++i) {
    float b[3];
    xf_point(m_base, x[i], y[i], z[i], b);
    int32_t ex, ey;
    const int kind = gvo_ray_end(g, ox, oy, b[0], b[1], b[2], &ex, &ey);
    if (!kind) continue;
    if (seen) {
      uint8_t *s = &seen[(size_t)ey * (size_t)g->nx + (size_t)ex];
      if (*s & (uint8_t)kind) continue;
      *s |= (uint8_t)kind;
    }
    v += march(g, ocx, ocy, ex, ey, kind == 2, miss);
  }
  free(seen);
  if (visits) *visits = v;
}
