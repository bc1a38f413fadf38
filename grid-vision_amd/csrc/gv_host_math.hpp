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

// bboxPoseEstimation :156-181 + computePCABoundingBox :187-247 for one (already filtered)
// bbox cloud, in the reference's order: pcl::compute3DCentroid (fp32 running sum / n),
// cv::PCA(DATA_AS_ROW, CV_32F) on rows (z, x): fp32 mean, fp64 covariance of the
// fp32-centred samples scaled by 1/n and stored fp32, eigenvectors of the symmetric 2x2,
// projections min/max.  The eigenvector sign is arbitrary upstream; major.x >= 0 here.
inline bool pca_bbox(const float *x, const float *y, const float *z, size_t n, gv_lshape_pose &out)
{
  out = gv_lshape_pose{};
  if (n == 0) return false;   // :174-175
  float cy = 0.0f;
  for (size_t i = 0; i < n; ++i) cy += y[i];
  cy /= (float)n;
  float m0 = 0.0f, m1 = 0.0f;
  for (size_t i = 0; i < n; ++i) { m0 += z[i]; m1 += x[i]; }
  m0 = m0 * (float)(1.0 / (double)n);
  m1 = m1 * (float)(1.0 / (double)n);
  double c00 = 0, c01 = 0, c11 = 0;
  for (size_t i = 0; i < n; ++i) {
    const float a = z[i] - m0, b = x[i] - m1;
    c00 += (double)a * a; c01 += (double)a * b; c11 += (double)b * b;
  }
  const double sc = 1.0 / (double)n;
  const double a = (double)(float)(c00 * sc), b = (double)(float)(c01 * sc), d = (double)(float)(c11 * sc);
  double mjx, mjy;
  if (b == 0.0) {
    if (a >= d) { mjx = 1; mjy = 0; } else { mjx = 0; mjy = 1; }
  } else {
    const double tr = a + d, df = a - d;
    const double root = std::sqrt(df * df + 4.0 * b * b);
    const double l1 = 0.5 * (tr + root);
    mjx = b; mjy = l1 - a;
    if (std::fabs(l1 - d) > std::fabs(mjy)) { mjx = l1 - d; mjy = b; }
    const double nn = std::sqrt(mjx * mjx + mjy * mjy);
    mjx /= nn; mjy /= nn;
  }
  if (mjx < 0 || (mjx == 0 && mjy < 0)) { mjx = -mjx; mjy = -mjy; }
  const float Mx = (float)mjx, My = (float)mjy, Nx = (float)(-mjy), Ny = (float)mjx;
  float minL = 3.402823466e+38f, maxL = -3.402823466e+38f, minW = 3.402823466e+38f, maxW = -3.402823466e+38f;
  for (size_t i = 0; i < n; ++i) {   // :203-216
    const float dx = z[i] - m0, dy = x[i] - m1;
    const float pl = dx * Mx + dy * My, pw = dx * Nx + dy * Ny;
    minL = std::min(minL, pl); maxL = std::max(maxL, pl);
    minW = std::min(minW, pw); maxW = std::max(maxW, pw);
  }
  const float angle = std::atan2(My, Mx) * 180.0f / (float)3.14159265358979323846;   // :227 (degrees)
  out.px = m1;    // :230 center.y
  out.py = cy;    // :231 then :181
  out.pz = m0;    // :232 center.x
  const Quat q = quat_from_rpy(0, -angle, 0);   // :236 (degrees passed as radians, as the reference does)
  out.qx = q.x; out.qy = q.y; out.qz = q.z; out.qw = q.w;
  out.length = maxL - minL;   // :218,:243
  out.width = maxW - minW;    // :219,:244
  out.height = 0.0;           // never set on this path in the reference
  return true;
}

// ---- RANSAC ground plane, host half (segmentGroundPlane, cloud_detections.cpp:105-138) ----
inline uint64_t splitmix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// SampleConsensusModelPlane::isSampleGood + computeModelCoefficients (fp32)
inline bool plane_from_sample(const float p0[3], const float p1[3], const float p2[3], float c[4])
{
  for (int k = 0; k < 3; ++k)
    if (!std::isfinite(p0[k]) || !std::isfinite(p1[k]) || !std::isfinite(p2[k])) return false;
  const float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  const float r0 = a[0] / b[0], r1 = a[1] / b[1], r2 = a[2] / b[2];
  if (!((r0 != r1) || (r2 != r1))) return false;
  float n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  const float len = std::sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
  if (!(len > 0.0f) || !std::isfinite(len)) return false;
  n[0] /= len; n[1] /= len; n[2] /= len;
  c[0] = n[0]; c[1] = n[1]; c[2] = n[2];
  c[3] = -1.0f * (((n[0] * p0[0]) + n[1] * p0[1]) + n[2] * p0[2]);
  return true;
}

inline bool plane_inlier(const float c[4], float x, float y, float z, double thr)
{
  const float d = (((c[0] * x) + c[1] * y) + c[2] * z) + c[3];
  return (double)std::fabs(d) < thr;
}

// unit eigenvector of the smallest eigenvalue of a symmetric 3x3 (cyclic Jacobi, fp64)
inline void smallest_eigenvector3(const double cov[6], double v[3])
{
  double a[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
  double e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double ekp = e[k][p], ekq = e[k][q];
          e[k][p] = c * ekp - s * ekq;
          e[k][q] = s * ekp + c * ekq;
        }
      }
  }
  int m = 0;
  if (a[1][1] < a[m][m]) m = 1;
  if (a[2][2] < a[m][m]) m = 2;
  const double n[3] = {e[0][m], e[1][m], e[2][m]};
  const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  int big = 0;
  if (std::fabs(n[1]) > std::fabs(n[big])) big = 1;
  if (std::fabs(n[2]) > std::fabs(n[big])) big = 2;
  const double sg = (n[big] < 0) ? -1.0 / len : 1.0 / len;
  v[0] = n[0] * sg; v[1] = n[1] * sg; v[2] = n[2] * sg;
}

// optimizeModelCoefficients: least-squares plane of the inliers of c (cloud order, fp64)
inline size_t refine_plane(const float *x, const float *y, const float *z, size_t n, const float c[4], double thr,
                           float refined[4])
{
  double sx = 0, sy = 0, sz = 0;
  size_t m = 0;
  for (size_t i = 0; i < n; ++i)
    if (plane_inlier(c, x[i], y[i], z[i], thr)) { sx += x[i]; sy += y[i]; sz += z[i]; ++m; }
  for (int k = 0; k < 4; ++k) refined[k] = c[k];
  if (m < 3) return m;
  const double cx = sx / (double)m, cy = sy / (double)m, cz = sz / (double)m;
  double cov[6] = {0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < n; ++i)
    if (plane_inlier(c, x[i], y[i], z[i], thr)) {
      const double dx = x[i] - cx, dy = y[i] - cy, dz = z[i] - cz;
      cov[0] += dx * dx; cov[1] += dx * dy; cov[2] += dx * dz;
      cov[3] += dy * dy; cov[4] += dy * dz; cov[5] += dz * dz;
    }
  double nv[3];
  smallest_eigenvector3(cov, nv);
  refined[0] = (float)nv[0]; refined[1] = (float)nv[1]; refined[2] = (float)nv[2];
  refined[3] = (float)(-((nv[0] * cx + nv[1] * cy) + nv[2] * cz));
  return m;
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
