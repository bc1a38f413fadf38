// gv_host_math.hpp -- small host-side arithmetic of the hot path: what the
// reference delegates to tf2, pcl_ros and Eigen around its per-frame loop.
// Product code (no dependency on oracle/).  Compiled with -ffp-contract=off.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "gv_types.hpp"

namespace gv {
namespace host {

struct Quat {
  double x, y, z, w;
};

// 3x3 fp64 rotation, row-major (tf2::Matrix3x3)
struct Basis {
  double m[9];

  // tf2::Matrix3x3::setRotation(const Quaternion&)
  static Basis from_quat(const Quat &q)
  {
    const double d = ((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w;
    const double s = 2.0 / d;
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    const double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    Basis b;
    b.m[0] = 1.0 - (yy + zz); b.m[1] = xy - wz;         b.m[2] = xz + wy;
    b.m[3] = xy + wz;         b.m[4] = 1.0 - (xx + zz); b.m[5] = yz - wx;
    b.m[6] = xz - wy;         b.m[7] = yz + wx;         b.m[8] = 1.0 - (xx + yy);
    return b;
  }

  // tf2::Matrix3x3::getRotation(Quaternion&)
  Quat to_quat() const
  {
    const double trace = (m[0] + m[4]) + m[8];
    double t[4];
    if (trace > 0.0) {
      double s = std::sqrt(trace + 1.0);
      t[3] = s * 0.5;
      s = 0.5 / s;
      t[0] = (m[7] - m[5]) * s;
      t[1] = (m[2] - m[6]) * s;
      t[2] = (m[3] - m[1]) * s;
    } else {
      const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      double s = std::sqrt(((m[i * 3 + i] - m[j * 3 + j]) - m[k * 3 + k]) + 1.0);
      t[i] = s * 0.5;
      s = 0.5 / s;
      t[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
      t[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
      t[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
    }
    return Quat{t[0], t[1], t[2], t[3]};
  }

  Basis operator*(const Basis &o) const
  {
    Basis r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        r.m[i * 3 + j] = (m[i * 3 + 0] * o.m[0 * 3 + j] + m[i * 3 + 1] * o.m[1 * 3 + j]) + m[i * 3 + 2] * o.m[2 * 3 + j];
    return r;
  }
};

inline Xform64 xform_from_tf(const gv_transform &t)
{
  const Basis b = Basis::from_quat(Quat{t.qx, t.qy, t.qz, t.qw});
  Xform64 x;
  for (int i = 0; i < 9; ++i) x.b[i] = b.m[i];
  x.o[0] = t.tx; x.o[1] = t.ty; x.o[2] = t.tz;
  return x;
}

// tf2::Transform::operator()(Vector3): basis[r].dot(v) + origin[r]
inline void apply(const Xform64 &x, const double v[3], double out[3])
{
  for (int r = 0; r < 3; ++r) out[r] = ((x.b[r * 3] * v[0] + x.b[r * 3 + 1] * v[1]) + x.b[r * 3 + 2] * v[2]) + x.o[r];
}

// pcl_ros::transformPointCloud(in, out, tf2::Transform) matrix construction:
// quaternion read back from the tf2 basis, narrowed to fp32,
// Eigen::Quaternionf::toRotationMatrix, translation narrowed to fp32.
inline Mat34f pcl_matrix_from_tf(const gv_transform &t)
{
  const Quat q = Basis::from_quat(Quat{t.qx, t.qy, t.qz, t.qw}).to_quat();
  const float x = (float)q.x, y = (float)q.y, z = (float)q.z, w = (float)q.w;
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Mat34f m;
  m.m[0] = 1.0f - (tyy + tzz); m.m[1] = txy - twz;          m.m[2] = txz + twy;           m.m[3] = (float)t.tx;
  m.m[4] = txy + twz;          m.m[5] = 1.0f - (txx + tzz); m.m[6] = tyz - twx;           m.m[7] = (float)t.ty;
  m.m[8] = txz - twy;          m.m[9] = tyz + twx;          m.m[10] = 1.0f - (txx + tyy); m.m[11] = (float)t.tz;
  return m;
}

// tf2::Quaternion::setRPY
inline Quat quat_from_rpy(double roll, double pitch, double yaw)
{
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = std::cos(hy), sy = std::sin(hy);
  const double cp = std::cos(hp), sp = std::sin(hp);
  const double cr = std::cos(hr), sr = std::sin(hr);
  return Quat{sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
              cr * cp * cy + sr * sp * sy};
}

// tf2::doTransform(Pose): Transform(t) * Transform(r, v)
inline void transform_pose(const gv_transform &t, gv_lshape_pose &p)
{
  const Xform64 x = xform_from_tf(t);
  const double v[3] = {p.px, p.py, p.pz};
  double o[3];
  apply(x, v, o);
  const Basis prod = Basis::from_quat(Quat{t.qx, t.qy, t.qz, t.qw}) * Basis::from_quat(Quat{p.qx, p.qy, p.qz, p.qw});
  const Quat q = prod.to_quat();
  p.px = o[0]; p.py = o[1]; p.pz = o[2];
  p.qx = q.x; p.qy = q.y; p.qz = q.z; p.qw = q.w;
}

// setIntrinsicMatrix / K.inverse() (Eigen 3x3 cofactor inverse)
inline void intrinsics(double fx, double fy, double cx, double cy, double K[9], double Ki[9])
{
  const double k[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
  for (int i = 0; i < 9; ++i) K[i] = k[i];
  auto M = [&](int r, int c) { return k[r * 3 + c]; };
  double cof[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      cof[i * 3 + j] = M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
    }
  const double det = (cof[0] * M(0, 0) + cof[3] * M(1, 0)) + cof[6] * M(2, 0);
  const double invdet = 1.0 / det;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Ki[r * 3 + c] = cof[c * 3 + r] * invdet;
}

// ---- object_detection post-processing (src/object_detection.cpp:94-269) ----
inline int32_t object_class(int32_t label) { return (label >= 0 && label <= 9) ? label : 10; }

inline float iou(const gv_bbox &box, const float r[4])
{
  const float bx0 = (float)box.x_min, by0 = (float)box.y_min, bx1 = (float)box.x_max, by1 = (float)box.y_max;
  const float x1 = std::max(r[0], bx0), y1 = std::max(r[1], by0);
  const float x2 = std::min(r[2], bx1), y2 = std::min(r[3], by1);
  const float w = std::max(x2 - x1, 0.0f), h = std::max(y2 - y1, 0.0f);
  const float inter = w * h;
  const float area1 = (r[2] - r[0]) * (r[3] - r[1]);
  const float area2 = (float)((box.x_max - box.x_min) * (box.y_max - box.y_min));
  return inter / ((area1 + area2) - inter);
}

// fast_non_max_suppression :166-211 (stable sort: equal confidences keep input order)
inline std::vector<gv_bbox> nms(std::vector<gv_bbox> b, float iou_threshold)
{
  std::vector<gv_bbox> out;
  if (b.empty()) return out;
  std::stable_sort(b.begin(), b.end(), [](const gv_bbox &a, const gv_bbox &c) { return a.confidence > c.confidence; });
  const size_t n = b.size();
  std::vector<float> mat(n * 4);
  std::vector<char> keep(n, 1);
  for (size_t i = 0; i < n; ++i) {
    mat[i * 4 + 0] = (float)b[i].x_min; mat[i * 4 + 1] = (float)b[i].y_min;
    mat[i * 4 + 2] = (float)b[i].x_max; mat[i * 4 + 3] = (float)b[i].y_max;
  }
  for (size_t i = 0; i < n; ++i) {
    if (!keep[i]) continue;
    out.push_back(b[i]);
    for (size_t j = i + 1; j < n; ++j)
      if (iou(b[i], &mat[j * 4]) > iou_threshold) keep[j] = 0;
  }
  return out;
}

// denormalizeAndScaleBoundingBox :226-239
inline void denormalize(std::vector<gv_bbox> &b, int orig_w, int orig_h, int resize)
{
  const float scale_x = static_cast<float>(orig_w) / resize;
  const float scale_y = static_cast<float>(orig_h) / resize;
  for (auto &box : b) {
    box.x_min = static_cast<int>(box.x_min * resize * scale_x);
    box.y_min = static_cast<int>(box.y_min * resize * scale_y);
    box.x_max = static_cast<int>(box.x_max * resize * scale_x);
    box.y_max = static_cast<int>(box.y_max * resize * scale_y);
  }
}

// (double)u >= lo  <=>  u >= ceil_f(lo);   (double)u <= hi  <=>  u <= floor_f(hi)
// for every float u (NaN bounds stay NaN: both forms are then always false).
inline float ceil_to_float(double v)
{
  float f = (float)v;
  if ((double)f < v) f = std::nextafterf(f, INFINITY);
  return f;
}
inline float floor_to_float(double v)
{
  float f = (float)v;
  if ((double)f > v) f = std::nextafterf(f, -INFINITY);
  return f;
}

// host getIndex (same arithmetic as the device one) for geometry-only queries
inline bool get_index(const GridParams &g, double x, double y, int &ix, int &iy)
{
  const double tx = -((x - g.pos_x) - g.off_x);
  const double ty = -((y - g.pos_y) - g.off_y);
  if (!(tx >= 0.0 && ty >= 0.0 && tx < g.len_x && ty < g.len_y)) return false;
  const double vx = ((x - g.off_x) - g.pos_x) / g.res;
  const double vy = ((y - g.off_y) - g.pos_y) / g.res;
  const int jx = (int)(-vx), jy = (int)(-vy);
  if (jx < 0 || jy < 0 || jx >= g.nx || jy >= g.ny) return false;
  ix = jx;
  iy = jy;
  return true;
}

}  // namespace host
}  // namespace gv
