/*
 * oracle/ransac.c -- CPU ORACLE (test infrastructure; see gv_oracle.h).
 * cloud_detections::segmentGroundPlane  src/cloud_detections.cpp:105-138:
 * pcl::SACSegmentation, SACMODEL_PLANE, SAC_RANSAC, distance threshold 0.04,
 * optimizeCoefficients = true, inliers removed.
 *
 * PARITY UNPINNED and NOT bit-reproducible against PCL: PCL draws its samples from
 * boost::mt19937 (and stops adaptively); SURVEY 8(f)-2 specifies this step BY OUTCOME.
 * This file defines the variant the HIP path implements:
 *   - `iters` hypotheses (PCL default max_iterations_ = 50), sample t draws three point
 *     indices from the counter-based splitmix64 stream seed + 3t + {0,1,2} (mod n);
 *   - plane through the 3 points and the collinearity test as PCL's
 *     SampleConsensusModelPlane::computeModelCoefficients / isSampleGood (fp32);
 *   - inlier iff |n.p + d| < threshold (fp32 distance vs fp64 threshold), best = most
 *     inliers, first wins ties;
 *   - refinement: one pass of fp64 raw moments of the inliers (as pcl::computeMeanAndCovarianceMatrix);
 *     every sum is taken by a fixed 64-ary tree over the cloud order (below), so that a parallel
 *     implementation rounds the same way as this one; covariance = Q/m - c c^T; normal = eigenvector
 *     of the smallest eigenvalue (closed form, as pcl::eigen33), d = -n.centroid; the final inlier set is
 *     re-selected with the refined coefficients.
 */
#include "gv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint64_t sm64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* returns 1 and the fp32 plane (a,b,c,d) when the sample is usable */
int gvo_plane_from_sample(const float p0[3], const float p1[3], const float p2[3], float coeff[4])
{
  for (int k = 0; k < 3; ++k)
    if (!isfinite(p0[k]) || !isfinite(p1[k]) || !isfinite(p2[k])) return 0;
  const float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  /* isSampleGood: dy1dy2 = p1p0 / p2p0 elementwise; good iff not all three ratios equal */
  const float r0 = a[0] / b[0], r1 = a[1] / b[1], r2 = a[2] / b[2];
  if (!((r0 != r1) || (r2 != r1))) return 0;
  float n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
  if (!(len > 0.0f) || !isfinite(len)) return 0;
  n[0] /= len; n[1] /= len; n[2] /= len;
  coeff[0] = n[0]; coeff[1] = n[1]; coeff[2] = n[2];
  coeff[3] = -1.0f * (((n[0] * p0[0]) + n[1] * p0[1]) + n[2] * p0[2]);
  return 1;
}

static inline int is_inlier(const float c[4], float x, float y, float z, double thr)
{
  const float d = (((c[0] * x) + c[1] * y) + c[2] * z) + c[3];
  return (double)fabsf(d) < thr;   /* NaN -> 0 */
}

/* Unit eigenvector of the smallest eigenvalue of a symmetric 3x3 (fp64), closed form as pcl::eigen33(mat, eigenvalue,
 * eigenvector) [UPSTREAM-RECALL], which SampleConsensusModelPlane::optimizeModelCoefficients calls: the matrix is scaled
 * by its largest |entry|, the roots of the characteristic polynomial come from the trigonometric solution (computeRoots),
 * and the eigenvector is the largest of the three cross products of rows of (A - lambda I).  Sign: the component of
 * largest magnitude (the first of equals) is made positive.
 * (Rounds 2-3 ran a cyclic Jacobi here -- ninety dependent fp64 divisions and square roots, 14 us of single-lane tail
 * on the device; the step is specified by outcome, SURVEY 8(f)-2, and tests/test_oracle_second_opinions.py holds the
 * refined plane against numpy's SVD either way.) */
void gvo_smallest_eigenvector3(const double cov[6], double v[3])
{
  double a[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
  double scale = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (fabs(a[i][j]) > scale) scale = fabs(a[i][j]);
  if (!(scale > 2.2250738585072014e-308)) scale = 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = a[i][j] / scale;
  /* computeRoots: characteristic polynomial x^3 - c2 x^2 + c1 x - c0 */
  const double c0 = (((a[0][0] * a[1][1]) * a[2][2] + (2.0 * a[0][1]) * a[0][2] * a[1][2]) - (a[0][0] * a[1][2]) * a[1][2]
                     - (a[1][1] * a[0][2]) * a[0][2]) - (a[2][2] * a[0][1]) * a[0][1];
  const double c1 = ((((a[0][0] * a[1][1] - a[0][1] * a[0][1]) + a[0][0] * a[2][2]) - a[0][2] * a[0][2]) + a[1][1] * a[2][2])
                    - a[1][2] * a[1][2];
  const double c2 = (a[0][0] + a[1][1]) + a[2][2];
  double lam;
  {
    const double c2_over_3 = c2 * (1.0 / 3.0);
    double a_over_3 = (c2 * c2_over_3 - c1) * (1.0 / 3.0);
    if (a_over_3 < 0.0) a_over_3 = 0.0;
    const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
    double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    if (q < 0.0) q = 0.0;
    const double rho = sqrt(a_over_3);
    const double theta = atan2(sqrt(q), half_b) * (1.0 / 3.0);
    const double cos_theta = cos(theta), sin_theta = sin(theta);
    const double r0 = c2_over_3 + 2.0 * rho * cos_theta;
    const double r1 = c2_over_3 - rho * (cos_theta + 1.7320508075688772 * sin_theta);
    const double r2 = c2_over_3 - rho * (cos_theta - 1.7320508075688772 * sin_theta);
    lam = r0;
    if (r1 < lam) lam = r1;
    if (r2 < lam) lam = r2;
    if (lam <= 0.0) lam = 0.0;   /* a covariance matrix is positive semi-definite: eigen33 recomputes a root <= 0 as 0 */
  }
  a[0][0] -= lam; a[1][1] -= lam; a[2][2] -= lam;
  const double v1[3] = {a[0][1] * a[1][2] - a[0][2] * a[1][1], a[0][2] * a[1][0] - a[0][0] * a[1][2], a[0][0] * a[1][1] - a[0][1] * a[1][0]};
  const double v2[3] = {a[0][1] * a[2][2] - a[0][2] * a[2][1], a[0][2] * a[2][0] - a[0][0] * a[2][2], a[0][0] * a[2][1] - a[0][1] * a[2][0]};
  const double v3[3] = {a[1][1] * a[2][2] - a[1][2] * a[2][1], a[1][2] * a[2][0] - a[1][0] * a[2][2], a[1][0] * a[2][1] - a[1][1] * a[2][0]};
  const double l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2];
  const double l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2];
  const double l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
  double n[3], len2;
  if (l1 >= l2 && l1 >= l3) { n[0] = v1[0]; n[1] = v1[1]; n[2] = v1[2]; len2 = l1; }
  else if (l2 >= l1 && l2 >= l3) { n[0] = v2[0]; n[1] = v2[1]; n[2] = v2[2]; len2 = l2; }
  else { n[0] = v3[0]; n[1] = v3[1]; n[2] = v3[2]; len2 = l3; }
  const double len = sqrt(len2);
  int big = 0;
  if (fabs(n[1]) > fabs(n[big])) big = 1;
  if (fabs(n[2]) > fabs(n[big])) big = 2;
  const double sg = (n[big] < 0) ? -1.0 / len : 1.0 / len;
  v[0] = n[0] * sg; v[1] = n[1] * sg; v[2] = n[2] * sg;
}

/* Sum of v[0..n) by the 64-ary tree of the definition: groups of 64 consecutive values (the last one padded
 * with +0.0) are reduced by the butterfly  t[i] += t[i + off], off = 32, 16, 8, 4, 2, 1;  the group sums form
 * the next level, until one value is left (at least one level is always taken).  v is overwritten. */
static double tree_sum64(double *v, size_t n)
{
  if (n == 0) return 0.0;
  do {
    const size_t groups = (n + 63) / 64;
    for (size_t g = 0; g < groups; ++g) {
      double t[64];
      for (int i = 0; i < 64; ++i) t[i] = (g * 64 + (size_t)i < n) ? v[g * 64 + (size_t)i] : 0.0;
      for (int off = 32; off > 0; off >>= 1)
        for (int i = 0; i < off; ++i) t[i] = t[i] + t[i + off];
      v[g] = t[0];
    }
    n = groups;
  } while (n > 1);
  return v[0];
}

/* refined plane from the inliers of `coeff`: ONE pass of fp64 raw moments (count, sums of x y z, sums of the
 * six products -- each product of two fp32 coordinates is exact in fp64), every sum taken by the 64-ary tree
 * over the cloud order (a non-inlier contributes +0.0); centroid c = S / m, covariance C_ij = Q_ij / m - c_i c_j
 * (as pcl::computeMeanAndCovarianceMatrix, which SACSegmentation's optimizeModelCoefficients calls, accumulates
 * [UPSTREAM-RECALL]); normal = eigenvector of the smallest eigenvalue, d = -n.c.  Returns the inlier count.
 * (Round 3: the round-2 definition took the covariance in a second pass around the centroid; one pass is what
 * PCL does and is one sweep of the cloud on the device.  tests/test_oracle_second_opinions.py holds the plane
 * against numpy's SVD of the same inliers.) */
size_t gvo_refine_plane(const float *x, const float *y, const float *z, size_t n, const float coeff[4],
                        double thr, float refined[4])
{
  memcpy(refined, coeff, 4 * sizeof(float));
  double *buf = (double *)malloc((n ? n : 1) * sizeof(double));
  if (!buf) return 0;
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) m += (size_t)is_inlier(coeff, x[i], y[i], z[i], thr);
  if (m < 3) { free(buf); return m; }
  double mom[9];   /* Sx Sy Sz Qxx Qxy Qxz Qyy Qyz Qzz */
  for (int k = 0; k < 9; ++k) {
    for (size_t i = 0; i < n; ++i) {
      double t = 0.0;
      if (is_inlier(coeff, x[i], y[i], z[i], thr)) {
        const double dx = x[i], dy = y[i], dz = z[i];
        t = (k == 0) ? dx : (k == 1) ? dy : (k == 2) ? dz : (k == 3) ? dx * dx : (k == 4) ? dx * dy : (k == 5) ? dx * dz
          : (k == 6) ? dy * dy : (k == 7) ? dy * dz : dz * dz;
      }
      buf[i] = t;
    }
    mom[k] = tree_sum64(buf, n);
  }
  free(buf);
  const double dm = (double)m;
  const double cx = mom[0] / dm, cy = mom[1] / dm, cz = mom[2] / dm;
  const double cov[6] = {mom[3] / dm - cx * cx, mom[4] / dm - cx * cy, mom[5] / dm - cx * cz,
                         mom[6] / dm - cy * cy, mom[7] / dm - cy * cz, mom[8] / dm - cz * cz};
  double nv[3];
  gvo_smallest_eigenvector3(cov, nv);
  refined[0] = (float)nv[0]; refined[1] = (float)nv[1]; refined[2] = (float)nv[2];
  refined[3] = (float)(-((nv[0] * cx + nv[1] * cy) + nv[2] * cz));
  return m;
}

size_t gvo_segment_ground_plane(const float *x, const float *y, const float *z, size_t n, double thr,
                                int32_t iters, uint64_t seed, uint8_t *inlier, float coeff_out[4])
{
  memset(inlier, 0, n);
  memset(coeff_out, 0, 4 * sizeof(float));
  if (n < 3) return 0;
  size_t best = 0;
  float bestc[4] = {0, 0, 0, 0};
  for (int32_t t = 0; t < iters; ++t) {
    size_t id[3];
    for (int k = 0; k < 3; ++k) id[k] = (size_t)(sm64(seed + 3ull * (uint64_t)t + (uint64_t)k) % (uint64_t)n);
    const float p0[3] = {x[id[0]], y[id[0]], z[id[0]]}, p1[3] = {x[id[1]], y[id[1]], z[id[1]]},
                p2[3] = {x[id[2]], y[id[2]], z[id[2]]};
    float c[4];
    if (!gvo_plane_from_sample(p0, p1, p2, c)) continue;
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) cnt += (size_t)is_inlier(c, x[i], y[i], z[i], thr);
    if (cnt > best) { best = cnt; memcpy(bestc, c, sizeof(c)); }
  }
  if (best == 0) return 0;   /* :122-126 "Could not estimate a planar model" */
  float refined[4];
  gvo_refine_plane(x, y, z, n, bestc, thr, refined);
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    inlier[i] = (uint8_t)is_inlier(refined, x[i], y[i], z[i], thr);
    m += inlier[i];
  }
  memcpy(coeff_out, refined, 4 * sizeof(float));
  return m;
}
