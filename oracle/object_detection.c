/*
 * oracle/object_detection.c -- CPU ORACLE (test infrastructure; see gv_oracle.h).
 * PARITY UNPINNED.  Follows src/object_detection.cpp:94-269 and
 * src/grid_vision_node.cpp:384-403.
 */
#include "gv_oracle.h"

#include <stdlib.h>
#include <string.h>

/* setIntrinsicMatrix  src/object_detection.cpp:241-247 */
void gvo_set_intrinsic(double fx, double fy, double cx, double cy, double K[9])
{
  K[0] = fx; K[1] = 0;  K[2] = cx;
  K[3] = 0;  K[4] = fy; K[5] = cy;
  K[6] = 0;  K[7] = 0;  K[8] = 1;
}

/* computeKInverse :249  K.inverse().  [UPSTREAM-RECALL] Eigen 3x3 fixed-size
 * inverse: cofactors, det = col0(cofactors) . row0(K)... restated as
 * inv = cofactor^T * (1/det) with det expanded along the first column. */
void gvo_k_inverse(const double K[9], double Ki[9])
{
#define M(r, c) K[(r) * 3 + (c)]
  /* cofactor(i,j) of Eigen's cofactor_3x3<i,j>:
   *   m((i+1)%3,(j+1)%3)*m((i+2)%3,(j+2)%3) - m((i+1)%3,(j+2)%3)*m((i+2)%3,(j+1)%3) */
  double cof[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      cof[i * 3 + j] = M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
    }
  /* cofactors_col0 = (cof(0,0), cof(1,0), cof(2,0)); det = cofactors_col0 . col(0) */
  const double det = (cof[0] * M(0, 0) + cof[3] * M(1, 0)) + cof[6] * M(2, 0);
  const double invdet = 1.0 / det;
  /* result(r,c) = cofactor(c,r) * invdet */
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Ki[r * 3 + c] = cof[c * 3 + r] * invdet;
#undef M
}

/* getObjectClass :252-269 */
int32_t gvo_get_object_class(int32_t label)
{
  return (label >= 0 && label <= 9) ? label : GVO_UNKNOWN;
}

/* computeIoU_Eigen :148-164 for one candidate row (fp32) */
static float iou_f32(const gvo_bbox *box, const float r[4])
{
  const float bx0 = (float)box->x_min, by0 = (float)box->y_min;
  const float bx1 = (float)box->x_max, by1 = (float)box->y_max;
  const float x1 = r[0] < bx0 ? bx0 : r[0];          /* cwiseMax */
  const float y1 = r[1] < by0 ? by0 : r[1];
  const float x2 = bx1 < r[2] ? bx1 : r[2];          /* cwiseMin */
  const float y2 = by1 < r[3] ? by1 : r[3];
  float w = x2 - x1; w = w < 0.0f ? 0.0f : w;
  float h = y2 - y1; h = h < 0.0f ? 0.0f : h;
  const float inter = w * h;
  const float area1 = (r[2] - r[0]) * (r[3] - r[1]);
  const float area2 = (float)((box->x_max - box->x_min) * (box->y_max - box->y_min));
  return inter / ((area1 + area2) - inter);
}

/* fast_non_max_suppression :166-211.  The reference's std::sort is unstable;
 * a stable descending sort is used here, so equal-confidence order is the
 * input order (ties are order-dependent in the reference: SURVEY 8(a) A16). */
int32_t gvo_nms(gvo_bbox *b, int32_t n, float iou_threshold, gvo_bbox *out)
{
  if (n <= 0) return 0;
  for (int32_t i = 1; i < n; ++i) {          /* stable insertion sort, desc */
    gvo_bbox t = b[i];
    int32_t j = i - 1;
    while (j >= 0 && b[j].confidence < t.confidence) { b[j + 1] = b[j]; --j; }
    b[j + 1] = t;
  }
  float *mat = (float *)malloc((size_t)n * 4 * sizeof(float));
  unsigned char *keep = (unsigned char *)malloc((size_t)n);
  for (int32_t i = 0; i < n; ++i) {          /* :183-190 (double -> float) */
    mat[i * 4 + 0] = (float)b[i].x_min; mat[i * 4 + 1] = (float)b[i].y_min;
    mat[i * 4 + 2] = (float)b[i].x_max; mat[i * 4 + 3] = (float)b[i].y_max;
    keep[i] = 1;
  }
  int32_t m = 0;
  for (int32_t i = 0; i < n; ++i) {          /* :193-208 */
    if (!keep[i]) continue;
    out[m++] = b[i];
    for (int32_t j = i + 1; j < n; ++j)
      if (iou_f32(&b[i], &mat[j * 4]) > iou_threshold) keep[j] = 0;
  }
  free(mat);
  free(keep);
  return m;
}

/* denormalizeAndScaleBoundingBox :226-239 */
void gvo_denormalize(gvo_bbox *b, int32_t n, int32_t orig_w, int32_t orig_h, int32_t resize)
{
  /* Promotion (:229-237): float / int -> float scale; box.x_min is a DOUBLE field, so x_min * resize * scale_x is
   * double * int * float = fp64 throughout; static_cast<int> truncates towards zero; the int goes back into the double. */
  const float scale_x = (float)orig_w / resize;
  const float scale_y = (float)orig_h / resize;
  for (int32_t i = 0; i < n; ++i) {
    b[i].x_min = (int)(b[i].x_min * resize * scale_x);
    b[i].y_min = (int)(b[i].y_min * resize * scale_y);
    b[i].x_max = (int)(b[i].x_max * resize * scale_x);
    b[i].y_max = (int)(b[i].y_max * resize * scale_y);
  }
}

/* extract_bboxes :94-146 */
int32_t gvo_extract_bboxes(const float *boxes, const float *scores, int32_t n, int32_t c,
                           double conf_threshold, double iou_threshold,
                           int32_t orig_w, int32_t orig_h, int32_t resize, gvo_bbox *out)
{
  gvo_bbox *cand = (gvo_bbox *)malloc((size_t)(n > 0 ? n : 1) * sizeof(gvo_bbox));
  int32_t m = 0;
  for (int32_t i = 0; i < n; ++i) {
    int32_t best = 0;                         /* :121-122 maxCoeff: first maximum */
    float mx = scores[(size_t)i * c];
    for (int32_t k = 1; k < c; ++k)
      if (scores[(size_t)i * c + k] > mx) { mx = scores[(size_t)i * c + k]; best = k; }
    if (mx >= conf_threshold) {               /* :125 float vs double */
      gvo_bbox b;
      b.confidence = mx;
      b.label = gvo_get_object_class(best);
      b.x_min = boxes[i * 4 + 0]; b.y_min = boxes[i * 4 + 1];
      b.x_max = boxes[i * 4 + 2]; b.y_max = boxes[i * 4 + 3];
      cand[m++] = b;
    }
  }
  /* :142 iou_threshold (double) narrows to the float parameter */
  const int32_t k = gvo_nms(cand, m, (float)iou_threshold, out);
  gvo_denormalize(out, k, orig_w, orig_h, resize);   /* :143 */
  free(cand);
  return k;
}

/* GridVision::filterBBoxes  src/grid_vision_node.cpp:384-403 */
int32_t gvo_filter_bboxes(const gvo_bbox *in, int32_t n, gvo_bbox *stat, gvo_bbox *dyn,
                          int32_t *n_dyn)
{
  int32_t ns = 0, nd = 0;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t l = in[i].label;
    if (l == GVO_VEHICLE || l == GVO_BIKE || l == GVO_MOTORBIKE || l == GVO_PERSON) dyn[nd++] = in[i];
    else stat[ns++] = in[i];
  }
  *n_dyn = nd;
  return ns;
}
