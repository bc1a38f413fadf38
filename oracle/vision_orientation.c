/*
 * oracle/vision_orientation.c -- CPU ORACLE (test infrastructure; see gv_oracle.h).
 * PARITY UNPINNED.  Follows src/vision_orientation.cpp:241-519 (geometry half);
 * Eigen's ColPivHouseholderQR is restated [UPSTREAM-RECALL] (its vectorised
 * reduction order is not reproducible without Eigen, hence tolerance checks).
 */
#include "gv_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* generateBins  :241-258 */
void gvo_generate_bins(int32_t bins, float *out)
{
  /* Promotion: M_PI is a double, so 2.0f * M_PI and the division by the int are fp64; narrowed once on assignment. */
  const float interval = (float)(2.0f * M_PI / bins);          /* :244 */
  for (int32_t i = 0; i < bins; ++i) out[i] = 0.0f;
  for (int32_t i = 1; i < bins; ++i) out[i] = i * interval;    /* :248 */
  for (int32_t i = 0; i < bins; ++i) out[i] += interval / 2.0f; /* :254 */
}

/* computeAlpha  :260-275 */
float gvo_compute_alpha(const float orient[4], int32_t argmax, const float *bins)
{
  const float cos_val = orient[argmax * 2 + 0];
  const float sin_val = orient[argmax * 2 + 1];
  float alpha = atan2f(sin_val, cos_val);                      /* :268 */
  alpha += bins[argmax];                                       /* :271 */
  alpha -= (float)M_PI;                                        /* :272 */
  return alpha;
}

/* computeThetaRay  :277-292 */
float gvo_compute_theta_ray(const gvo_cam *cam, const gvo_bbox *b)
{
  const float fx = cam->fx;                                    /* proj_mat_(0,0) */
  /* Promotion: orig_w_ is an int and fx a float: int / float -> float, std::atan(float) is the float overload: fp32
   * throughout.  :282 adds two DOUBLE box fields and divides by 2.0f in fp64, narrowed on assignment; the rest is fp32. */
  const float fovx = 2.0f * atanf(cam->orig_w / (2.0f * fx));  /* :280 */
  const float box_center_x = (float)((b->x_min + b->x_max) / 2.0f);   /* :282 */
  float dx = box_center_x - (cam->orig_w / 2.0f);              /* :283 */
  const float sign = (dx < 0) ? -1.0f : 1.0f;                  /* :285 */
  dx = fabsf(dx);
  float angle = atanf((2.0f * dx * tanf(fovx / 2.0f)) / cam->orig_w);  /* :288 */
  angle *= sign;
  return angle;
}

/* ---- Eigen::ColPivHouseholderQR<Matrix<float,4,3>>::solve  [UPSTREAM-RECALL] ---- */
static void qr_solve_4x3(const float Ain[12], const float bin[4], float x[3])
{
  enum { R = 4, C = 3 };
  float a[R][C], c[R], hc[C];
  int perm[C];
  float ncu[C], ncd[C];
  for (int i = 0; i < R; ++i) { for (int j = 0; j < C; ++j) a[i][j] = Ain[i * C + j]; c[i] = bin[i]; }
  float maxn = 0.0f;
  for (int j = 0; j < C; ++j) {
    float s = 0.0f;
    for (int i = 0; i < R; ++i) s += a[i][j] * a[i][j];
    ncu[j] = ncd[j] = sqrtf(s);
    if (ncu[j] > maxn) maxn = ncu[j];
  }
  const float eps = FLT_EPSILON;
  float th = maxn * eps / (float)R;
  const float threshold_helper = th * th;
  const float downdate_thr = sqrtf(eps);
  int nonzero = C;
  float maxpivot = 0.0f;
  int transp[C];
  for (int k = 0; k < C; ++k) {
    int big = k; float bigv = ncu[k];
    for (int j = k + 1; j < C; ++j) if (ncu[j] > bigv) { bigv = ncu[j]; big = j; }
    const float big_sq = bigv * bigv;
    if (nonzero == C && big_sq < threshold_helper * (float)(R - k)) nonzero = k;
    transp[k] = big;
    if (big != k) {
      for (int i = 0; i < R; ++i) { float t = a[i][k]; a[i][k] = a[i][big]; a[i][big] = t; }
      float t = ncu[k]; ncu[k] = ncu[big]; ncu[big] = t;
      t = ncd[k]; ncd[k] = ncd[big]; ncd[big] = t;
    }
    /* makeHouseholderInPlace on a[k..R-1][k] */
    float tail = 0.0f;
    for (int i = k + 1; i < R; ++i) tail += a[i][k] * a[i][k];
    const float c0 = a[k][k];
    float tau, beta;
    if (tail <= FLT_MIN) {
      tau = 0.0f; beta = c0;
      for (int i = k + 1; i < R; ++i) a[i][k] = 0.0f;
    } else {
      beta = sqrtf(c0 * c0 + tail);
      if (c0 >= 0.0f) beta = -beta;
      for (int i = k + 1; i < R; ++i) a[i][k] = a[i][k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hc[k] = tau;
    a[k][k] = beta;
    if (fabsf(beta) > maxpivot) maxpivot = fabsf(beta);
    /* applyHouseholderOnTheLeft to a[k..][k+1..] */
    if (tau != 0.0f) {
      for (int j = k + 1; j < C; ++j) {
        float tmp = 0.0f;
        for (int i = k + 1; i < R; ++i) tmp += a[i][k] * a[i][j];
        tmp += a[k][j];
        a[k][j] -= tau * tmp;
        for (int i = k + 1; i < R; ++i) a[i][j] -= tau * a[i][k] * tmp;
      }
    }
    for (int j = k + 1; j < C; ++j) {
      if (ncu[j] != 0.0f) {
        float t = fabsf(a[k][j]) / ncu[j];
        t = (1.0f + t) * (1.0f - t);
        t = t < 0.0f ? 0.0f : t;
        const float r = ncu[j] / ncd[j];
        const float t2 = t * r * r;
        if (t2 <= downdate_thr) {
          float s = 0.0f;
          for (int i = k + 1; i < R; ++i) s += a[i][j] * a[i][j];
          ncd[j] = sqrtf(s);
          ncu[j] = ncd[j];
        } else ncu[j] *= sqrtf(t);
      }
    }
  }
  for (int j = 0; j < C; ++j) perm[j] = j;
  for (int k = 0; k < C; ++k) { int t = perm[k]; perm[k] = perm[transp[k]]; perm[transp[k]] = t; }
  /* rank: |R(i,i)| > maxpivot * (eps * diagSize) */
  const float prethr = fabsf(maxpivot) * (eps * (float)C);
  int rank = 0;
  for (int i = 0; i < nonzero; ++i) if (fabsf(a[i][i]) > prethr) ++rank;
  /* c = Q^T b */
  for (int k = 0; k < C; ++k) {
    if (hc[k] != 0.0f) {
      float tmp = 0.0f;
      for (int i = k + 1; i < R; ++i) tmp += a[i][k] * c[i];
      tmp += c[k];
      c[k] -= hc[k] * tmp;
      for (int i = k + 1; i < R; ++i) c[i] -= hc[k] * a[i][k] * tmp;
    }
  }
  /* back-substitute the leading rank x rank upper triangle */
  float y[C] = {0, 0, 0};
  for (int i = rank - 1; i >= 0; --i) {
    float s = c[i];
    for (int j = i + 1; j < rank; ++j) s -= a[i][j] * y[j];
    y[i] = s / a[i][i];
  }
  for (int i = 0; i < C; ++i) x[i] = 0.0f;
  for (int i = 0; i < rank; ++i) x[perm[i]] = y[i];
}

/* calcLocation  :294-447.  all_loc / all_err (optional, 64 x 3 / 64 floats): the solution and the
 * residual of every constraint set in the reference's loop order -- the parity tests use them to prove
 * that a different arg-min on the device was a near-tie. */
static void calc_location_impl(const gvo_cam *cam, const double dimension[3], const gvo_bbox *bbox,
                               float alpha, float theta_ray, double pose_out[7], float *best_err_out,
                               float *all_loc, float *all_err)
{
  const float orient = alpha + theta_ray;                      /* :298 */
  const float c = cosf(orient), s = sinf(orient);              /* rotationMatrix :512-519 */
  const float Rm[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  const float box[4] = {(float)bbox->x_min, (float)bbox->y_min,
                        (float)bbox->x_max, (float)bbox->y_max};   /* :301-303 */
  /* Promotion: dimension is std::array<double, 3>: double / 2.0f is an fp64 division, narrowed on assignment;
   * 88 * M_PI / 180.0f is int * double / float = fp64, narrowed once (:311-313). */
  const float dx = (float)(dimension[0] / 2.0f);               /* :306 */
  const float dy = (float)(dimension[1] / 2.0f);               /* :307 */
  const float dz = (float)(dimension[2] / 2.0f);               /* :308 */
  int left_mult = 1, right_mult = -1;                          /* :311 */
  const float deg88 = (float)(88 * M_PI / 180.0f);
  const float deg90 = (float)(90 * M_PI / 180.0f);
  const float deg92 = (float)(92 * M_PI / 180.0f);
  if (alpha < deg92 && alpha > deg88) { left_mult = 1; right_mult = 1; }
  else if (alpha < -deg88 && alpha > -deg92) { left_mult = -1; right_mult = -1; }
  else if (alpha < deg90 && alpha > -deg90) { left_mult = -1; right_mult = 1; }
  const int switch_mult = (alpha > 0) ? 1 : -1;                /* :332 */

  double L[2][3], Rg[2][3], T[4][3], B[4][3];                  /* Vec3 = doubles */
  {
    int n = 0;
    for (int i = -1; i <= 1; i += 2) {                         /* :340-346 */
      L[n][0] = left_mult * dx;  L[n][1] = (float)i * dy; L[n][2] = -switch_mult * dz;
      Rg[n][0] = right_mult * dx; Rg[n][1] = (float)i * dy; Rg[n][2] = switch_mult * dz;
      ++n;
    }
    n = 0;
    for (int i = -1; i <= 1; i += 2)                           /* :348-357 */
      for (int j = -1; j <= 1; j += 2) {
        T[n][0] = (float)i * dx; T[n][1] = -dy; T[n][2] = (float)j * dz;
        B[n][0] = (float)i * dx; B[n][1] = dy;  B[n][2] = (float)j * dz;
        ++n;
      }
  }
  /* proj_mat_  :19-20 (3x4 float) */
  const float P[3][4] = {{cam->fx, 0.0f, cam->cx, 0.0f}, {0.0f, cam->fy, cam->cy, 0.0f},
                         {0.0f, 0.0f, 1.0f, 0.0f}};
  static const int indices[4] = {0, 1, 0, 1};                  /* :379 */
  float best_loc[3] = {0, 0, 0};
  float best_error = FLT_MAX;                                  /* :382 */
  for (int l = 0; l < 2; ++l)                                  /* :363-374 order */
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 2; ++r)
        for (int b = 0; b < 4; ++b) {
          const double *X[4] = {L[l], T[t], Rg[r], B[b]};
          float A[12], bb[4];
          for (int row = 0; row < 4; ++row) {
            const float xv = (float)X[row][0], yv = (float)X[row][1], zv = (float)X[row][2];
            float RX[3];
            for (int q = 0; q < 3; ++q) RX[q] = (Rm[q * 3] * xv + Rm[q * 3 + 1] * yv) + Rm[q * 3 + 2] * zv;
            /* projected_M = proj_mat_ * M, M = I with column 3 = (RX,1)  :393-399 */
            float pM[3][4];
            for (int q = 0; q < 3; ++q) {
              const float Mcol3[4] = {RX[0], RX[1], RX[2], 1.0f};
              for (int cc = 0; cc < 3; ++cc) {
                /* M(:,cc) = e_cc */
                float acc = 0.0f;
                for (int kk = 0; kk < 4; ++kk) {
                  const float term = P[q][kk] * ((kk == cc) ? 1.0f : 0.0f);
                  acc = (kk == 0) ? term : acc + term;
                }
                pM[q][cc] = acc;
              }
              pM[q][3] = ((P[q][0] * Mcol3[0] + P[q][1] * Mcol3[1]) + P[q][2] * Mcol3[2]) + P[q][3] * Mcol3[3];
            }
            const int idx = indices[row];
            const float bv = box[row];
            for (int cc = 0; cc < 3; ++cc) A[row * 3 + cc] = pM[idx][cc] - bv * pM[2][cc];  /* :412 */
            bb[row] = bv * pM[2][3] - pM[idx][3];                                           /* :415 */
          }
          float loc[3];
          qr_solve_4x3(A, bb, loc);                            /* :419 */
          float err = 0.0f;                                    /* :422 */
          for (int row = 0; row < 4; ++row) {
            const float rr = ((A[row * 3] * loc[0] + A[row * 3 + 1] * loc[1]) + A[row * 3 + 2] * loc[2]) - bb[row];
            err += rr * rr;
          }
          {
            const int id = ((l * 4 + t) * 2 + r) * 4 + b;
            if (all_loc) { all_loc[id * 3] = loc[0]; all_loc[id * 3 + 1] = loc[1]; all_loc[id * 3 + 2] = loc[2]; }
            if (all_err) all_err[id] = err;
          }
          if (err < best_error) {                              /* :424-429 */
            best_error = err;
            best_loc[0] = loc[0]; best_loc[1] = loc[1]; best_loc[2] = loc[2];
          }
        }
  pose_out[0] = best_loc[0]; pose_out[1] = best_loc[1]; pose_out[2] = best_loc[2];   /* :434-436 */
  gvo_set_rpy(0, -orient, 0, &pose_out[3]);                    /* :440 */
  if (best_err_out) *best_err_out = best_error;
}

void gvo_calc_location(const gvo_cam *cam, const double dimension[3], const gvo_bbox *bbox,
                       float alpha, float theta_ray, double pose_out[7], float *best_err_out)
{
  calc_location_impl(cam, dimension, bbox, alpha, theta_ray, pose_out, best_err_out, NULL, NULL);
}

void gvo_calc_location_all(const gvo_cam *cam, const double dimension[3], const gvo_bbox *bbox,
                           float alpha, float theta_ray, float all_loc[192], float all_err[64])
{
  double pose[7];
  calc_location_impl(cam, dimension, bbox, alpha, theta_ray, pose, NULL, all_loc, all_err);
}

/* postProcessOutputs  :449-510 ; class averages include/grid_vision/vision_orientation.hpp:58-69 */
int32_t gvo_post_process(const gvo_cam *cam, const float *orient, const float *conf,
                         const float *dims, const gvo_bbox *bboxes, int32_t nb, gvo_lshape *out)
{
  float bins[2];
  gvo_generate_bins(2, bins);                                  /* :43 */
  int32_t m = 0;
  for (int32_t i = 0; i < nb; ++i) {
    const float *cs = &conf[i * 2], *os = &orient[i * 4], *ds = &dims[i * 3];
    const int argmax = (cs[1] > cs[0]) ? 1 : 0;                /* :466-467 max_element: first max */
    const float alpha = gvo_compute_alpha(os, argmax, bins);
    const float theta_ray = gvo_compute_theta_ray(cam, &bboxes[i]);
    float al, aw, ah;
    switch (bboxes[i].label) {
    case GVO_VEHICLE:   al = 3.884f; aw = 1.629f; ah = 1.526f; break;
    case GVO_BIKE:      al = 1.763f; aw = 0.597f; ah = 1.737f; break;
    case GVO_MOTORBIKE: al = 2.2f;   aw = 0.8f;   ah = 1.5f;   break;
    case GVO_PERSON:    al = 0.842f; aw = 0.660f; ah = 1.761f; break;
    default: continue;                                         /* :496-499 */
    }
    gvo_lshape r;
    memset(&r, 0, sizeof(r));
    r.length = ds[2] + al;                                     /* :474 */
    r.width = ds[0] + aw;                                      /* :475 */
    r.height = ds[1] + ah;                                     /* :476 */
    const double lwh[3] = {r.length, r.width, r.height};       /* :501 */
    double pose[7];
    gvo_calc_location(cam, lwh, &bboxes[i], alpha, theta_ray, pose, NULL);
    r.px = pose[0]; r.py = pose[1]; r.pz = pose[2];
    r.qx = pose[3]; r.qy = pose[4]; r.qz = pose[5]; r.qw = pose[6];
    out[m++] = r;
  }
  return m;
}
