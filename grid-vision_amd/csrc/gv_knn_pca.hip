// gv_knn_pca.hip -- the two "next tier" cloud_detections kernels:
//   * exact brute-force kNN of each bbox centre over the projected cloud
//     (buildKDTree + computeDepthForBoundingBoxes, src/cloud_detections.cpp:8-87)
//   * RadiusOutlierRemoval neighbour counting per bbox cloud
//     (src/cloud_detections.cpp:150-154)
// gfx950, wave64, built with -ffp-contract=off.
#include "gv_kernels.hpp"

#include <float.h>
#include <math.h>

namespace gv {

__device__ __forceinline__ void xform34k(const Mat34f &m, float px, float py, float pz, float &ox, float &oy,
                                         float &oz)
{
  ox = __fadd_rn(__fmul_rn(px, m.m[0]), __fadd_rn(__fmul_rn(py, m.m[1]), __fadd_rn(__fmul_rn(pz, m.m[2]), m.m[3])));
  oy = __fadd_rn(__fmul_rn(px, m.m[4]), __fadd_rn(__fmul_rn(py, m.m[5]), __fadd_rn(__fmul_rn(pz, m.m[6]), m.m[7])));
  oz = __fadd_rn(__fmul_rn(px, m.m[8]), __fadd_rn(__fmul_rn(py, m.m[9]), __fadd_rn(__fmul_rn(pz, m.m[10]), m.m[11])));
}

// buildKDTree projection (src/cloud_detections.cpp:14-33): camera transform, skip z <= 0,
// (u, v) = float(K*p / z) in fp64, depth = z.  Skipped points get NaN so they never rank.
__global__ void __launch_bounds__(256) k_project_uvd(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, uint32_t n, Mat34f m, CamK cam,
                                                     float *__restrict__ pu, float *__restrict__ pv,
                                                     float *__restrict__ pd)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  const float qnan = __uint_as_float(0x7fc00000u);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float cx, cy, cz;
    xform34k(m, x[i], y[i], z[i], cx, cy, cz);
    float u = qnan, v = qnan, d = qnan;
    if (!(cz <= 0)) {   // :16 (a NaN z passes this test in the reference too)
      const double X = cx, Y = cy, Z = cz;
      const double ix = (cam.k[0] * X + cam.k[1] * Y) + cam.k[2] * Z;
      const double iy = (cam.k[3] * X + cam.k[4] * Y) + cam.k[5] * Z;
      const double iz = (cam.k[6] * X + cam.k[7] * Y) + cam.k[8] * Z;
      u = (float)(ix / iz);
      v = (float)(iy / iz);
      d = cz;
    }
    pu[i] = u;
    pv[i] = v;
    pd[i] = d;
  }
}

void launch_project_uvd(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m,
                        const CamK &cam, float *pu, float *pv, float *pd, hipStream_t s)
{
  if (!n) return;
  const uint32_t blocks = (uint32_t)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_project_uvd, dim3(blocks), dim3(256), 0, s, x, y, z, n, m, cam, pu, pv, pd);
}

// (d2, index) candidates ordered lexicographically: equal distances resolve to the
// lower point index (the oracle's stable insertion has the same rule; FLANN's own tie
// order is tree dependent).
struct Cand {
  float d2;
  uint32_t idx;
};
__device__ __forceinline__ bool cand_less(const Cand &a, const Cand &b)
{
  return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx);
}

constexpr int kKnnMaxK = 32;
constexpr int kKnnThreads = 256;

// Stage 1: grid (chunks, nb).  Each thread keeps the k best of its strided points in a
// private sorted LDS list; the block then extracts its k best by k rounds of a block-wide
// arg-min.  FLANN L2_Simple distance: ((du*du) + dv*dv) + dz*dz in fp32, query (cx, cy, 0).
__global__ void __launch_bounds__(kKnnThreads) k_knn_stage1(const float *__restrict__ pu,
                                                            const float *__restrict__ pv,
                                                            const float *__restrict__ pd, uint32_t n,
                                                            const gv_bbox *__restrict__ bboxes, int k,
                                                            Cand *__restrict__ partial)
{
  extern __shared__ Cand s_list[];   // [threads][k]
  __shared__ Cand s_red[kKnnThreads];
  __shared__ int s_head[kKnnThreads];
  const int tid = threadIdx.x, b = blockIdx.y;
  const gv_bbox bb = bboxes[b];
  const float qx = (float)(bb.x_min + ((bb.x_max - bb.x_min) / 2.0f));   // :57
  const float qy = (float)(bb.y_min + ((bb.y_max - bb.y_min) / 2.0f));   // :58
  const float qz = 0.0f;                                                  // :59
  Cand *mine = s_list + (size_t)tid * k;
  int cnt = 0;
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = min(n, lo + per);
  for (uint32_t i = lo + tid; i < hi; i += kKnnThreads) {
    float d, r = 0.0f;
    d = pu[i] - qx; r = __fadd_rn(r, __fmul_rn(d, d));
    d = pv[i] - qy; r = __fadd_rn(r, __fmul_rn(d, d));
    d = pd[i] - qz; r = __fadd_rn(r, __fmul_rn(d, d));
    if (!(r == r)) continue;   // NaN never ranks
    const Cand c{r, i};
    if (cnt < k) {
      int j = cnt++;
      while (j > 0 && cand_less(c, mine[j - 1])) { mine[j] = mine[j - 1]; --j; }
      mine[j] = c;
    } else if (cand_less(c, mine[k - 1])) {
      int j = k - 1;
      while (j > 0 && cand_less(c, mine[j - 1])) { mine[j] = mine[j - 1]; --j; }
      mine[j] = c;
    }
  }
  // k rounds of block arg-min over the heads of the private sorted lists
  s_head[tid] = 0;
  const Cand none{INFINITY, 0xFFFFFFFFu};
  for (int round = 0; round < k; ++round) {
    const int h = s_head[tid];
    s_red[tid] = (h < cnt) ? mine[h] : none;
    __syncthreads();
    for (int off = kKnnThreads / 2; off > 0; off >>= 1) {
      if (tid < off && cand_less(s_red[tid + off], s_red[tid])) s_red[tid] = s_red[tid + off];
      __syncthreads();
    }
    const Cand best = s_red[0];
    if (tid == 0) partial[((size_t)b * gridDim.x + blockIdx.x) * k + round] = best;
    if (h < cnt && mine[h].idx == best.idx) s_head[tid] = h + 1;   // the owner advances
    __syncthreads();
  }
}

// Stage 2: one block per bbox merges the chunk winners, writes the sorted squared
// distances and the upper-median depth (nth_element at size/2, :78-81).
__global__ void __launch_bounds__(kKnnThreads) k_knn_stage2(const Cand *__restrict__ partial, int nchunks, int k,
                                                            const float *__restrict__ pd,
                                                            float *__restrict__ depths,
                                                            float *__restrict__ knn_d2)
{
  __shared__ Cand s_red[kKnnThreads];
  __shared__ Cand s_best[kKnnMaxK];
  const int tid = threadIdx.x, b = blockIdx.x;
  const Cand *p = partial + (size_t)b * nchunks * k;
  const int total = nchunks * k;
  const Cand none{INFINITY, 0xFFFFFFFFu};
  Cand last{-INFINITY, 0};
  bool have_last = false;
  for (int round = 0; round < k; ++round) {
    Cand m = none;
    for (int i = tid; i < total; i += kKnnThreads) {
      const Cand c = p[i];
      if (c.idx == 0xFFFFFFFFu) continue;
      if (have_last && !cand_less(last, c)) continue;   // already taken
      if (cand_less(c, m)) m = c;
    }
    s_red[tid] = m;
    __syncthreads();
    for (int off = kKnnThreads / 2; off > 0; off >>= 1) {
      if (tid < off && cand_less(s_red[tid + off], s_red[tid])) s_red[tid] = s_red[tid + off];
      __syncthreads();
    }
    last = s_red[0];
    have_last = true;
    if (tid == 0) s_best[round] = last;
    __syncthreads();
  }
  if (tid == 0) {
    int cnt = 0;
    float dv[kKnnMaxK];
    for (int j = 0; j < k; ++j) {
      const Cand c = s_best[j];
      if (knn_d2) knn_d2[(size_t)b * k + j] = (c.idx == 0xFFFFFFFFu) ? INFINITY : c.d2;
      if (c.idx != 0xFFFFFFFFu) dv[cnt++] = pd[c.idx];
    }
    float out = -1.0f;   // :49
    if (cnt > 0) {
      for (int a = 1; a < cnt; ++a) {
        const float t = dv[a];
        int j = a - 1;
        while (j >= 0 && dv[j] > t) { dv[j + 1] = dv[j]; --j; }
        dv[j + 1] = t;
      }
      out = dv[cnt / 2];
    }
    depths[b] = out;
  }
}

void launch_knn(const float *pu, const float *pv, const float *pd, uint32_t n, const gv_bbox *bboxes, int nb, int k,
                int nchunks, Cand2 *partial, float *depths, float *knn_d2, hipStream_t s)
{
  if (nb <= 0) return;
  static_assert(sizeof(Cand2) == sizeof(Cand), "candidate layout");
  hipLaunchKernelGGL(k_knn_stage1, dim3(nchunks, nb), dim3(kKnnThreads), (size_t)kKnnThreads * k * sizeof(Cand), s, pu,
                     pv, pd, n, bboxes, k, reinterpret_cast<Cand *>(partial));
  hipLaunchKernelGGL(k_knn_stage2, dim3(nb), dim3(kKnnThreads), 0, s, reinterpret_cast<const Cand *>(partial), nchunks, k,
                     pd, depths, knn_d2);
}

// ---- RadiusOutlierRemoval neighbour counts --------------------------------------
// Points are given grouped by bbox (segments).  keep[i] = 1 iff at least min_pts+1 points
// of the same segment (the point itself included) lie within d2 <= r2f, where r2f is the
// largest float not above the fp64 radius*radius (PCL >= 1.11 dense path).
__global__ void __launch_bounds__(256) k_radius_count(const float *__restrict__ x, const float *__restrict__ y,
                                                      const float *__restrict__ z,
                                                      const int32_t *__restrict__ seg_of,
                                                      const int32_t *__restrict__ seg_start, int32_t n_max,
                                                      const int32_t *__restrict__ n_dev, float r2f,
                                                      int32_t min_pts, uint8_t *__restrict__ keep)
{
  const int32_t n = n_dev ? min(n_max, *n_dev) : n_max;   // the number of gathered points may live on the device
  if ((int)blockIdx.x * 256 >= n) return;
  __shared__ float sx[256], sy[256], sz[256];
  // one block per 256 consecutive points; a block may straddle segments, so each thread
  // walks its own segment in tiles of 256 shared by the block when the tile ranges agree
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < n;
  int s0 = 0, s1 = 0;
  float px = 0, py = 0, pz = 0;
  if (active) {
    const int sg = seg_of[i];
    s0 = seg_start[sg];
    s1 = seg_start[sg + 1];
    px = x[i]; py = y[i]; pz = z[i];
  }
  // the block scans the union range of its threads' segments
  __shared__ int s_lo, s_hi;
  if (threadIdx.x == 0) { s_lo = 0x7fffffff; s_hi = 0; }
  __syncthreads();
  if (active) { atomicMin(&s_lo, s0); atomicMax(&s_hi, s1); }
  __syncthreads();
  const int lo = s_lo, hi = s_hi;
  int cnt = 0;
  for (int t0 = lo; t0 < hi; t0 += 256) {
    const int j = t0 + threadIdx.x;
    if (j < hi) { sx[threadIdx.x] = x[j]; sy[threadIdx.x] = y[j]; sz[threadIdx.x] = z[j]; }
    __syncthreads();
    if (active && cnt <= min_pts) {
      const int jb = max(t0, s0), je = min(min(t0 + 256, hi), s1);
      for (int jj = jb; jj < je; ++jj) {
        float d, r = 0.0f;
        d = sx[jj - t0] - px; r = __fadd_rn(r, __fmul_rn(d, d));
        d = sy[jj - t0] - py; r = __fadd_rn(r, __fmul_rn(d, d));
        d = sz[jj - t0] - pz; r = __fadd_rn(r, __fmul_rn(d, d));
        cnt += (r <= r2f);
      }
    }
    __syncthreads();
  }
  if (active) keep[i] = (uint8_t)(cnt >= min_pts + 1);
}

void launch_radius_count(const float *x, const float *y, const float *z, const int32_t *seg_of,
                         const int32_t *seg_start, int32_t n_max, const int32_t *n_dev, float r2f, int32_t min_pts,
                         uint8_t *keep, hipStream_t s)
{
  if (n_max <= 0) return;
  hipLaunchKernelGGL(k_radius_count, dim3((n_max + 255) / 256), dim3(256), 0, s, x, y, z, seg_of, seg_start, n_max, n_dev,
                     r2f, min_pts, keep);
}

}  // namespace gv
