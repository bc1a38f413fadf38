/*
 * oracle/transforms.c -- CPU ORACLE (test infrastructure; see gv_oracle.h).
 * PARITY UNPINNED: tf2 / pcl_ros / PCL / Eigen are absent from this image; the
 * bodies below restate their published algorithms [UPSTREAM-RECALL] at the
 * reference call sites src/grid_vision_node.cpp:280-307, :337-382.
 */
#include "gv_oracle.h"

#include <math.h>

/* tf2::Matrix3x3::setRotation(Quaternion) [UPSTREAM-RECALL], fp64, row-major */
static void tf2_set_rotation(const double q[4], double m[9])
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double d = ((x * x + y * y) + z * z) + w * w;    /* length2() */
  const double s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs;
  const double xx = x * xs, xy = x * ys, xz = x * zs;
  const double yy = y * ys, yz = y * zs, zz = z * zs;
  m[0] = 1.0 - (yy + zz); m[1] = xy - wz;         m[2] = xz + wy;
  m[3] = xy + wz;         m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy;         m[7] = yz + wx;         m[8] = 1.0 - (xx + yy);
}

/* tf2::Matrix3x3::getRotation(Quaternion&) [UPSTREAM-RECALL] */
static void tf2_get_rotation(const double m[9], double q[4])
{
  const double trace = (m[0] + m[4]) + m[8];
  double t[4];
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    t[3] = s * 0.5;
    s = 0.5 / s;
    t[0] = (m[7] - m[5]) * s;
    t[1] = (m[2] - m[6]) * s;
    t[2] = (m[3] - m[1]) * s;
  } else {
    const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    const int j = (i + 1) % 3;
    const int k = (i + 2) % 3;
    double s = sqrt(((m[i * 3 + i] - m[j * 3 + j]) - m[k * 3 + k]) + 1.0);
    t[i] = s * 0.5;
    s = 0.5 / s;
    t[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
    t[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
    t[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
  }
  q[0] = t[0]; q[1] = t[1]; q[2] = t[2]; q[3] = t[3];
}

/* src/grid_vision_node.cpp:302-304
 *   tf2::fromMsg(transform_stamped.transform, tf_transform);      // quat -> Matrix3x3
 *   pcl_ros::transformPointCloud(lidar_cloud, *out, tf_transform);
 * [UPSTREAM-RECALL] pcl_ros: q = transform.getRotation() (Matrix3x3 -> quat, fp64);
 * Eigen::Quaternionf rotation(q.w(),q.x(),q.y(),q.z()); Eigen::Vector3f origin(v);
 * pcl::transformPointCloud(in, out, origin, rotation) builds
 * Eigen::Affine3f t(Translation3f(origin) * rotation), i.e. linear part =
 * Quaternionf::toRotationMatrix() in fp32, last column = origin. */
void gvo_tf_to_matrix4f(const gvo_tf *tf, float m[16])
{
  const double q_in[4] = {tf->qx, tf->qy, tf->qz, tf->qw};
  double basis[9], q[4];
  tf2_set_rotation(q_in, basis);
  tf2_get_rotation(basis, q);
  const float x = (float)q[0], y = (float)q[1], z = (float)q[2], w = (float)q[3];
  /* Eigen::QuaternionBase::toRotationMatrix, Scalar = float */
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  m[0] = 1.0f - (tyy + tzz); m[1] = txy - twz;          m[2]  = txz + twy;          m[3]  = (float)tf->tx;
  m[4] = txy + twz;          m[5] = 1.0f - (txx + tzz); m[6]  = tyz - twx;          m[7]  = (float)tf->ty;
  m[8] = txz - twy;          m[9] = tyz + twx;          m[10] = 1.0f - (txx + tyy); m[11] = (float)tf->tz;
  m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;
}

/* pcl::detail::Transformer<float>::se3, SSE2 path [UPSTREAM-RECALL]:
 *   p0 = x*c0; p1 = y*c1; p2 = z*c2;  out = p0 + (p1 + (p2 + c3))   (fp32, no FMA)
 * where c_k is column k of the 4x4.  Non-dense handling is irrelevant here:
 * arithmetic on non-finite inputs stays non-finite. */
void gvo_transform_cloud(const float m[16], const float *x, const float *y, const float *z,
                         float *ox, float *oy, float *oz, size_t n)
{
  for (size_t i = 0; i < n; ++i) {
    const float px = x[i], py = y[i], pz = z[i];
    for (int r = 0; r < 3; ++r) {
      const float p0 = px * m[r * 4 + 0];
      const float p1 = py * m[r * 4 + 1];
      const float p2 = pz * m[r * 4 + 2];
      const float o = p0 + (p1 + (p2 + m[r * 4 + 3]));
      if (r == 0) ox[i] = o; else if (r == 1) oy[i] = o; else oz[i] = o;
    }
  }
}

/* tf2::doTransform(Point, Point, TransformStamped) [UPSTREAM-RECALL]:
 *   t = Transform(Quaternion, Vector3);  out = t * v
 *   = (basis[r].dot(v) + origin[r]),  dot = x*x' + y*y' + z*z'  (fp64) */
void gvo_tf_point(const gvo_tf *tf, const double in[3], double out[3])
{
  const double q[4] = {tf->qx, tf->qy, tf->qz, tf->qw};
  const double o[3] = {tf->tx, tf->ty, tf->tz};
  double b[9];
  tf2_set_rotation(q, b);
  for (int r = 0; r < 3; ++r)
    out[r] = ((b[r * 3 + 0] * in[0] + b[r * 3 + 1] * in[1]) + b[r * 3 + 2] * in[2]) + o[r];
}

/* tf2::doTransform(Pose, Pose, TransformStamped) [UPSTREAM-RECALL]:
 *   v_out = t * Transform(r, v) = Transform(t.basis * R(r), t(v)); toMsg -> getRotation */
void gvo_tf_pose(const gvo_tf *tf, const double in[7], double out[7])
{
  const double q[4] = {tf->qx, tf->qy, tf->qz, tf->qw};
  const double r[4] = {in[3], in[4], in[5], in[6]};
  double b[9], rb[9], prod[9];
  tf2_set_rotation(q, b);
  tf2_set_rotation(r, rb);
  gvo_tf_point(tf, in, out);
  /* Matrix3x3 operator*: m[i][j] = m1[i].dot(column j of m2) = tdotx/y/z */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      prod[i * 3 + j] = (b[i * 3 + 0] * rb[0 * 3 + j] + b[i * 3 + 1] * rb[1 * 3 + j])
                        + b[i * 3 + 2] * rb[2 * 3 + j];
  tf2_get_rotation(prod, &out[3]);
}

/* tf2::Quaternion::setRPY(roll, pitch, yaw) [UPSTREAM-RECALL] */
void gvo_set_rpy(double roll, double pitch, double yaw, double q[4])
{
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = cos(hy), sy = sin(hy);
  const double cp = cos(hp), sp = sin(hp);
  const double cr = cos(hr), sr = sin(hr);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}
