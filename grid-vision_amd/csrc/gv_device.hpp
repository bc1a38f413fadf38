// gv_device.hpp -- device-side arithmetic shared by the gfx950 kernels: the reference's
// float transform, grid_map's getIndex, the pinhole projection and the X2 ray clip.
//
// Every translation unit that includes this is built with -ffp-contract=off: the reference
// arithmetic keeps separate multiply/add roundings (PCL's SSE transform, grid_map's getIndex)
// and cell indices must be bit-exact, so no FMA contraction; fp64 and fp32 divisions are the
// compiler's correctly rounded sequences.  Reference lines are cited as file:line relative to
// the reference root.
#pragma once

#include <hip/hip_runtime.h>

#include <float.h>
#include <stddef.h>
#include <math.h>

#include "gv_kernels.hpp"

namespace gv {

// A kernel argument field, loaded from the kernarg segment AT THE POINT OF USE.  The compiler loads every
// field of a by-value argument struct at the top of the kernel and keeps it in SGPRs for as long as some
// loop uses it; a kernel whose loops need more constants than the ~96 SGPRs of a fully occupied SIMD then
// spills them to VGPR lanes and pays a v_readlane (a VALU issue slot) per use.  Reading the field through a
// pointer the optimiser cannot see through turns that into a scalar load (SMEM, its own issue port, served
// from the scalar cache) whose registers die after the use.  `off` = offsetof(the kernel's single argument
// struct, field).
#define GV_AS4 __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ T load_karg(size_t off)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const GV_AS4 char *kp = (const GV_AS4 char *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *(const GV_AS4 T *)(kp + off);
#else
  (void)off;
  return T{};
#endif
}

// pcl::detail::Transformer<float>::se3 (SSE2 path): x*c0 + (y*c1 + (z*c2 + c3)),
// fp32, no FMA.  Call site: src/grid_vision_node.cpp:304.
__device__ __forceinline__ void xform34(const Mat34f &m, float px, float py, float pz, float &ox,
                                        float &oy, float &oz)
{
  ox = __fadd_rn(__fmul_rn(px, m.m[0]), __fadd_rn(__fmul_rn(py, m.m[1]), __fadd_rn(__fmul_rn(pz, m.m[2]), m.m[3])));
  oy = __fadd_rn(__fmul_rn(px, m.m[4]), __fadd_rn(__fmul_rn(py, m.m[5]), __fadd_rn(__fmul_rn(pz, m.m[6]), m.m[7])));
  oz = __fadd_rn(__fmul_rn(px, m.m[8]), __fadd_rn(__fmul_rn(py, m.m[9]), __fadd_rn(__fmul_rn(pz, m.m[10]), m.m[11])));
}

// grid_map::GridMap::getIndex (called at src/occupancy_grid.cpp:152):
//   indexVector = (position - 0.5*length - mapPosition) / resolution, index = (int)(-indexVector)
//   inside iff t = -(position - mapPosition - 0.5*length), 0 <= t < length
__device__ __forceinline__ bool get_index(const GridParams &g, double x, double y, int &ix, int &iy)
{
  const double tx = -((x - g.pos_x) - g.off_x);
  const double ty = -((y - g.pos_y) - g.off_y);
  if (!(tx >= 0.0 && ty >= 0.0 && tx < g.len_x && ty < g.len_y)) return false;  // NaN/inf land here
  const double vx = ((x - g.off_x) - g.pos_x) / g.res;
  const double vy = ((y - g.off_y) - g.pos_y) / g.res;
  const int jx = (int)(-vx);
  const int jy = (int)(-vy);
  if (jx < 0 || jy < 0 || jx >= g.nx || jy >= g.ny) return false;
  ix = jx;
  iy = jy;
  return true;
}

// Same result as get_index without the two fp64 divisions (the points pass is bound by fp64
// issue, a division is ~15 dependent fp64 ops): (int)(-(d / res)) only depends on which side
// of an integer the correctly rounded quotient lies.  q' = (-d) * fl(1/res) is within 3 roundings
// (< 1e-11 absolute for |q| < 2^14) of that quotient, so whenever q' is further than 1e-6 from
// every integer both truncate to the same cell; otherwise the exact division decides.
__device__ __forceinline__ bool get_index_fast(const GridParams &g, double x, double y, int &ix, int &iy)
{
  const double tx = -((x - g.pos_x) - g.off_x);
  const double ty = -((y - g.pos_y) - g.off_y);
  if (!(tx >= 0.0 && ty >= 0.0 && tx < g.len_x && ty < g.len_y)) return false;  // NaN/inf land here
  const double dx = (x - g.off_x) - g.pos_x;
  const double dy = (y - g.off_y) - g.pos_y;
  double qx = -dx * g.inv_res;
  double qy = -dy * g.inv_res;
  if (fabs(qx - rint(qx)) < 1e-6) qx = -(dx / g.res);
  if (fabs(qy - rint(qy)) < 1e-6) qy = -(dy / g.res);
  const int jx = (int)qx;
  const int jy = (int)qy;
  if (jx < 0 || jy < 0 || jx >= g.nx || jy >= g.ny) return false;
  ix = jx;
  iy = jy;
  return true;
}

// (float)(n / d) for finite n, d without the division: r ~ 1/d by v_rcp_f64 + two Newton steps
// (<= 1 ulp), q' = n * r is within a few ulp64 of the correctly rounded quotient, and both round
// to the same float unless q' sits within 2^-45 (relative) of a float rounding boundary, i.e. its
// 29 discarded mantissa bits are within 128 of the midpoint pattern; then the exact division
// decides.  Non-finite or subnormal-float results take the exact path too.
__device__ __forceinline__ double rcp_newton(double d)
{
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ bool div_risky(double q)
{
  const unsigned long long bits = (unsigned long long)__double_as_longlong(q);
  const unsigned low = (unsigned)bits & 0x1FFFFFFFu;
  const unsigned ex = (unsigned)(bits >> 52) & 0x7FFu;
  return (low - (0x10000000u - 128u)) <= 256u || ex < 1023u - 120u || ex > 1023u + 120u;
}
// (float)(n0 / d), (float)(n1 / d) given r ~ 1/d.  The exact divisions sit behind a WAVEFRONT-uniform branch: a
// per-lane select lets the compiler speculate both divisions into the common path (13 fp64 instructions each,
// executed for every point -- seen in the ISA of round 2's k_bin_partition); about one wavefront in 10^4 has a
// risky lane.
__device__ __forceinline__ void div2_to_float(double n0, double n1, double d, double r, float &o0, float &o1)
{
  const double q0 = n0 * r, q1 = n1 * r;
  const bool k0 = div_risky(q0), k1 = div_risky(q1);
  o0 = (float)q0;
  o1 = (float)q1;
  if (__ballot(k0 || k1) != 0ull) {
    if (k0) o0 = (float)(n0 / d);
    if (k1) o1 = (float)(n1 / d);
  }
}

// [EXTENSION] X2 ray end of an out-of-map point: fp64 slab clip of
// origin + t*(p - origin) against the map rectangle, then the clamped floor cell.
__device__ __forceinline__ void clip_ray_end(const GridParams &g, const RayOrigin &o, double px, double py,
                                             int &ex, int &ey)
{
  const double hix = g.pos_x + g.off_x, hiy = g.pos_y + g.off_y;
  const double lox = hix - g.len_x, loy = hiy - g.len_y;
  const double dx = px - o.ox, dy = py - o.oy;
  double t = 1.0;
  // one division per axis: the slab a ray can leave through is chosen by the sign of its direction first (the
  // quotient is the same one the two-branch form computes; dx == 0 constrains nothing)
  if (dx != 0.0) { const double tx = ((dx > 0.0 ? hix : lox) - o.ox) / dx; if (tx < t) t = tx; }
  if (dy != 0.0) { const double ty = ((dy > 0.0 ? hiy : loy) - o.oy) / dy; if (ty < t) t = ty; }
  if (t < 0.0) t = 0.0;
  const double qx = o.ox + t * dx;
  const double qy = o.oy + t * dy;
  double fx = floor(-(((qx - g.off_x) - g.pos_x) / g.res));
  double fy = floor(-(((qy - g.off_y) - g.pos_y) / g.res));
  if (!(fx >= 0.0)) fx = 0.0;
  if (!(fy >= 0.0)) fy = 0.0;
  if (fx > (double)(g.nx - 1)) fx = (double)(g.nx - 1);
  if (fy > (double)(g.ny - 1)) fy = (double)(g.ny - 1);
  ex = (int)fx;
  ey = (int)fy;
}

// Eigen Matrix3d * Vector3d, coefficient r: (K(r,0)*x + K(r,1)*y) + K(r,2)*z
__device__ __forceinline__ double krow(const double *k, int r, double x, double y, double z)
{
  return (k[r * 3 + 0] * x + k[r * 3 + 1] * y) + k[r * 3 + 2] * z;
}


// ------------------------------------------------------------- rectangles --
// updateMap(GridMap&, vector<LShapePose>) corners (src/occupancy_grid.cpp:79-90) and the
// index half of updateGridCellsFast (:147-172): any corner outside -> box skipped.
__device__ __forceinline__ Rect rect_from_corners(const GridParams &g, const double c[8])
{
  Rect r;
  r.valid = 1;
  int minx = 0, miny = 0, maxx = 0, maxy = 0;
  for (int i = 0; i < 4; ++i) {
    int ix, iy;
    if (!get_index(g, c[2 * i], c[2 * i + 1], ix, iy)) { r.valid = 0; break; }
    if (i == 0) { minx = maxx = ix; miny = maxy = iy; }
    else {
      minx = min(minx, ix); miny = min(miny, iy);
      maxx = max(maxx, ix); maxy = max(maxy, iy);
    }
  }
  r.x0 = minx; r.y0 = miny; r.x1 = maxx; r.y1 = maxy;
  return r;
}

// rectangle of one base-frame pose: corners {left_back, left_front, right_front, right_back} (:79-90)
__device__ __forceinline__ Rect rect_from_pose(const GridParams &g, const gv_lshape_pose &p)
{
  const double hx = p.length / 2.0, hy = p.width / 2.0;
  const double c[8] = {p.px - hx, p.py - hy, p.px + hx, p.py - hy, p.px + hx, p.py + hy, p.px - hx, p.py + hy};
  Rect r = rect_from_corners(g, c);
  // a pose the vision kernel marked invalid carries length < 0
  if (!(p.length >= 0.0)) r.valid = 0;
  return r;
}

// extractCloudPerBBox (src/cloud_detections.cpp:264-288) for one camera-frame point: index of
// the first bbox containing its projection, or -1.  The reference compares (double)u against
// the double bounds; bbox_f holds the float thresholds with the identical truth table (host:
// smallest float >= x_min, largest float <= x_max), and the 16x16-pixel tile masks only prune
// boxes that cannot contain this pixel, in index order.
__device__ __forceinline__ int first_bbox(const CamK &cam, const BBoxTest &t, float cx, float cy, float cz)
{
  int id = -1;
  // :264 pcl::isFinite(pt) && pt.z > 0.001f
  if (isfinite(cx) && isfinite(cy) && isfinite(cz) && !(cz <= 0.001f)) {
    // K = [[fx,0,cx],[0,fy,cy],[0,0,1]] (object_detection::setIntrinsicMatrix, :241-247: the only way a K
    // is made).  With its zero and unit entries Eigen's (K(r,0)*X + K(r,1)*Y) + K(r,2)*Z collapses exactly
    // for finite X, Y, Z: a product with 0 is +-0, x + (+-0) = x, and 1*Z = Z -- the same roundings in
    // the same places, 10 fp64 operations fewer per point.
    const double X = (double)cx, Y = (double)cy, Z = (double)cz;
    const double iz = Z;
    const double riz = rcp_newton(iz);
    float u, v;
    div2_to_float(cam.k[0] * X + cam.k[2] * Z, cam.k[4] * Y + cam.k[5] * Z, iz, riz, u, v);   // :268-273  (float)(n / iz)
    if (!(u < 0 || u >= (float)cam.W || v < 0 || v >= (float)cam.H)) {   // :276
      // :280-288 first match wins
      const int tx = (int)u >> 4, ty = (int)v >> 4;
      const unsigned long long *tm = t.tile_mask + ((size_t)ty * t.tiles_x + tx) * t.mask_words;
      for (int wd = 0; wd < t.mask_words && id < 0; ++wd) {
        unsigned long long m = tm[wd];
        while (m) {
          const int b = wd * 64 + (__ffsll((long long)m) - 1);
          m &= m - 1;
          const float4 f = t.bbox_f[b];
          if (u >= f.x && u <= f.z && v >= f.y && v <= f.w) {
            id = b;
            break;
          }
        }
      }
    }
  }
  return id;
}

}  // namespace gv
