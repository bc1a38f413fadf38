// gv_cloudops.hip -- device-resident halves of the "next tier" cloud_detections calls (SURVEY 8(f)-2,
// 8(f)-3): RANSAC ground plane (segmentGroundPlane, src/cloud_detections.cpp:105-138) and the per-bbox
// cloud split + PCA rectangle (extractCloudPerBBox / bboxPoseEstimation / computePCABoundingBox,
// :140-298).  Nothing here moves O(N) bytes to the host: hypotheses are drawn, counted, selected and
// refined on the device, bbox clouds are split by a stable device partition, and one wavefront per bbox
// reproduces the reference's sequential fp32 / fp64 accumulation order.  gfx950, wave64, built with
// -ffp-contract=off.
#include "gv_kernels.hpp"
#include "gv_device.hpp"

#include <algorithm>

namespace gv {

// ------------------------------------------------------------------ RANSAC --
__device__ __forceinline__ unsigned long long splitmix64_dev(unsigned long long z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// hypothesis t: three draws of the counter-based stream seed + 3t + {0,1,2} (mod n), plane through the
// three camera-frame points as PCL's SampleConsensusModelPlane (isSampleGood + computeModelCoefficients,
// fp32).  An unusable sample gets a NaN plane: it can never collect an inlier.
__global__ void __launch_bounds__(64) k_ransac_hypotheses(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ z, uint32_t n, Mat34f m,
                                                          unsigned long long seed, int iters, float4 *__restrict__ planes)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= iters) return;
  float p[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t id = (uint32_t)(splitmix64_dev(seed + 3ull * (unsigned long long)t + (unsigned long long)k) % (unsigned long long)n);
    xform34(m, x[id], y[id], z[id], p[k][0], p[k][1], p[k][2]);
  }
  const float qnan = __uint_as_float(0x7fc00000u);
  float4 out = make_float4(qnan, qnan, qnan, qnan);
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) ok = ok && isfinite(p[0][k]) && isfinite(p[1][k]) && isfinite(p[2][k]);
  if (ok) {
    const float a[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
    const float b[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
    const float r0 = a[0] / b[0], r1 = a[1] / b[1], r2 = a[2] / b[2];
    if ((r0 != r1) || (r2 != r1)) {
      float nn[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
      const float len = sqrtf((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]);
      if (len > 0.0f && isfinite(len)) {
        nn[0] /= len; nn[1] /= len; nn[2] /= len;
        out = make_float4(nn[0], nn[1], nn[2], -1.0f * (((nn[0] * p[0][0]) + nn[1] * p[0][1]) + nn[2] * p[0][2]));
      }
    }
  }
  planes[t] = out;
}

__device__ __forceinline__ bool plane_inlier_dev(const float4 &pl, float px, float py, float pz, double thr)
{
  const float d = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pl.x, px), __fmul_rn(pl.y, py)), __fmul_rn(pl.z, pz)), pl.w);
  return (double)fabsf(d) < thr;   // NaN -> false
}

// counts[h] += inliers of hypothesis h in this block's chunk of the (camera-frame) cloud
__global__ void __launch_bounds__(256) k_ransac_count(const float *__restrict__ x, const float *__restrict__ y,
                                                      const float *__restrict__ z, uint32_t n, Mat34f m,
                                                      const float4 *__restrict__ planes, double thr,
                                                      unsigned *__restrict__ counts)
{
  __shared__ unsigned s_w[4];
  const float4 pl = planes[blockIdx.y];
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = min(n, lo + per);
  unsigned c = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
    float cx, cy, cz;
    xform34(m, x[i], y[i], z[i], cx, cy, cz);
    c += plane_inlier_dev(pl, cx, cy, cz, thr) ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (t) atomicAdd(&counts[blockIdx.y], t);
  }
}

// best = most inliers, first hypothesis wins ties; state->best_count == 0: "could not estimate a planar model"
__global__ void k_ransac_select(const unsigned *__restrict__ counts, const float4 *__restrict__ planes, int iters,
                                RansacState *__restrict__ st)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned bestc = 0;
  int best = 0;
  for (int k = 0; k < iters; ++k)
    if (counts[k] > bestc) { bestc = counts[k]; best = k; }
  st->best_count = bestc;
  st->plane = bestc ? planes[best] : make_float4(0.f, 0.f, 0.f, 0.f);
  st->refined = st->plane;
  st->n_inliers = 0;
  st->m = 0;
}

// Deterministic fp64 sums for the refinement (optimizeModelCoefficients): a 64-ary tree in cloud order.
// Level 0: every wavefront reduces 64 consecutive points with the butterfly  v[i] += v[i + off],
// off = 32, 16, ..., 1  (non-inliers and padding contribute +0.0); every further level does the same over 64
// consecutive partial sums, until one value is left.  oracle/ransac.c states the same tree, so both sides
// round identically whatever the number of workgroups.
template <int K>
__device__ __forceinline__ void butterfly_store(double (&v)[K], double *__restrict__ out, size_t group, size_t ngroups)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = v[k] + __shfl_xor(v[k], off);
  }
  if ((threadIdx.x & 63) == 0 && group < ngroups) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[group * K + k] = v[k];
  }
}

// PASS 1: sums of x, y, z of the inliers of st->plane (+ their number);  PASS 2: the six covariance sums
// around st->centroid
template <int PASS>
__global__ void __launch_bounds__(256) k_ransac_level0(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ z, uint32_t n, Mat34f m, double thr,
                                                       const RansacState *__restrict__ st, double *__restrict__ out,
                                                       unsigned long long *__restrict__ m_count)
{
  constexpr int K = (PASS == 1) ? 3 : 6;
  const float4 pl = st->plane;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  bool in = false;
  if (i < n && st->best_count) {
    float cx, cy, cz;
    xform34(m, x[i], y[i], z[i], cx, cy, cz);
    in = plane_inlier_dev(pl, cx, cy, cz, thr);
    if (in) {
      if (PASS == 1) {
        v[0] = (double)cx; v[1] = (double)cy; v[2] = (double)cz;
      } else {
        const double dx = (double)cx - st->centroid[0], dy = (double)cy - st->centroid[1], dz = (double)cz - st->centroid[2];
        v[0] = dx * dx; v[1] = dx * dy; v[2] = dx * dz;
        if constexpr (K == 6) { v[3] = dy * dy; v[4] = dy * dz; v[5] = dz * dz; }
      }
    }
  }
  if (PASS == 1) {
    const unsigned long long bm = __ballot(in);
    if ((threadIdx.x & 63) == 0 && bm) atomicAdd(m_count, (unsigned long long)__popcll(bm));
  }
  butterfly_store<K>(v, out, i >> 6, ((size_t)n + 63) >> 6);
}

template <int K>
__global__ void __launch_bounds__(256) k_tree_level(const double *__restrict__ in, size_t count, double *__restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = (i < count) ? in[i * K + k] : 0.0;
  butterfly_store<K>(v, out, i >> 6, (count + 63) >> 6);
}

// unit eigenvector of the smallest eigenvalue of a symmetric 3x3 (cyclic Jacobi, fp64): oracle/ransac.c
__device__ void smallest_eigenvector3_dev(const double cov[6], double v[3])
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
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
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
  int mm = 0;
  if (a[1][1] < a[mm][mm]) mm = 1;
  if (a[2][2] < a[mm][mm]) mm = 2;
  const double nn[3] = {e[0][mm], e[1][mm], e[2][mm]};
  const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
  int big = 0;
  if (fabs(nn[1]) > fabs(nn[big])) big = 1;
  if (fabs(nn[2]) > fabs(nn[big])) big = 2;
  const double sg = (nn[big] < 0) ? -1.0 / len : 1.0 / len;
  v[0] = nn[0] * sg; v[1] = nn[1] * sg; v[2] = nn[2] * sg;
}

// after pass 1: centroid;  after pass 2: refined plane (kept = the sampled plane when fewer than 3 inliers)
template <int PASS>
__global__ void k_ransac_finish(const double *__restrict__ sums, RansacState *__restrict__ st)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!st->best_count) return;
  if (PASS == 1) {
    const double mm = (double)st->m;
    if (st->m >= 3) { st->centroid[0] = sums[0] / mm; st->centroid[1] = sums[1] / mm; st->centroid[2] = sums[2] / mm; }
  } else if (st->m >= 3) {
    const double cov[6] = {sums[0], sums[1], sums[2], sums[3], sums[4], sums[5]};
    double nv[3];
    smallest_eigenvector3_dev(cov, nv);
    st->refined = make_float4((float)nv[0], (float)nv[1], (float)nv[2],
                              (float)(-((nv[0] * st->centroid[0] + nv[1] * st->centroid[1]) + nv[2] * st->centroid[2])));
  }
}

// inliers of the refined plane: mask[i] (device resident) and their number
__global__ void __launch_bounds__(256) k_ransac_mask(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, uint32_t n, Mat34f m, double thr,
                                                     RansacState *__restrict__ st, uint8_t *__restrict__ mask)
{
  const float4 pl = st->refined;
  const bool have = st->best_count != 0;
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long cnt = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float cx, cy, cz;
    xform34(m, x[i], y[i], z[i], cx, cy, cz);
    const bool in = have && plane_inlier_dev(pl, cx, cy, cz, thr);
    mask[i] = in ? 1 : 0;
    cnt += in ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&st->n_inliers, cnt);
}

size_t ransac_scratch_doubles(size_t n) { return 6 * ((n + 63) / 64) + 6 * ((n + 4095) / 4096) + 64; }

void launch_ransac(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, double thr, int iters,
                   unsigned long long seed, float4 *planes, unsigned *counts, double *scratch, RansacState *st,
                   uint8_t *mask, hipStream_t s)
{
  hipLaunchKernelGGL(k_ransac_hypotheses, dim3((iters + 63) / 64), dim3(64), 0, s, x, y, z, n, m_cam, seed, iters, planes);
  (void)hipMemsetAsync(counts, 0, (size_t)iters * sizeof(unsigned), s);
  hipLaunchKernelGGL(k_ransac_count, dim3(64, iters), dim3(256), 0, s, x, y, z, n, m_cam, planes, thr, counts);
  hipLaunchKernelGGL(k_ransac_select, dim3(1), dim3(64), 0, s, counts, planes, iters, st);
  const size_t g0 = ((size_t)n + 63) / 64;
  double *bufA = scratch, *bufB = scratch + 6 * g0;   // ping-pong: level 0 -> A, then A -> B -> A ...
  auto tree = [&](int K) {
    size_t cnt = g0;
    double *src = bufA, *dst = bufB;
    while (cnt > 1) {
      const size_t groups = (cnt + 63) / 64;
      const uint32_t blocks = (uint32_t)((cnt + 255) / 256);
      if (K == 3) hipLaunchKernelGGL(k_tree_level<3>, dim3(blocks), dim3(256), 0, s, src, cnt, dst);
      else hipLaunchKernelGGL(k_tree_level<6>, dim3(blocks), dim3(256), 0, s, src, cnt, dst);
      std::swap(src, dst);
      cnt = groups;
    }
    return src;   // the total sits at src[0..K)
  };
  const uint32_t b0 = (uint32_t)(((size_t)n + 255) / 256);
  hipLaunchKernelGGL(k_ransac_level0<1>, dim3(b0), dim3(256), 0, s, x, y, z, n, m_cam, thr, st, bufA, &st->m);
  const double *s1 = tree(3);
  hipLaunchKernelGGL(k_ransac_finish<1>, dim3(1), dim3(64), 0, s, s1, st);
  hipLaunchKernelGGL(k_ransac_level0<2>, dim3(b0), dim3(256), 0, s, x, y, z, n, m_cam, thr, st, bufA, &st->m);
  const double *s2 = tree(6);
  hipLaunchKernelGGL(k_ransac_finish<2>, dim3(1), dim3(64), 0, s, s2, st);
  const uint32_t mb = (uint32_t)std::min<size_t>(((size_t)n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_ransac_mask, dim3(mb), dim3(256), 0, s, x, y, z, n, m_cam, thr, st, mask);
}

// --------------------------------------------------- per-bbox clouds + PCA --
// Stable split of the cloud by bbox id (extractCloudPerBBox appends in cloud order, :286): blocks of 1024
// points are counted, a column scan gives every (block, bbox) its first slot, and one wavefront per block
// places its points in order.  skip[i] != 0 (ground points, :306-314) drops the point.
constexpr int kSegBlock = 1024;

__global__ void __launch_bounds__(256) k_seg_count(const int16_t *__restrict__ ids, const uint8_t *__restrict__ skip,
                                                   uint32_t n, int nb, uint32_t *__restrict__ block_counts)
{
  extern __shared__ unsigned s_h[];
  for (int b = threadIdx.x; b < nb; b += 256) s_h[b] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSegBlock;
  for (int k = threadIdx.x; k < kSegBlock; k += 256) {
    const size_t i = base + k;
    if (i < n) {
      const int id = (skip && skip[i]) ? -1 : (int)ids[i];
      if (id >= 0 && id < nb) atomicAdd(&s_h[id], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += 256) block_counts[(size_t)blockIdx.x * nb + b] = s_h[b];
}

// one thread per bbox: exclusive prefix of its counts over the blocks (in place) and its total; then
// thread 0 turns the totals into seg_start[0..nb]
__global__ void __launch_bounds__(1024) k_seg_scan(uint32_t *__restrict__ block_counts, int nblocks, int nb,
                                                   int32_t *__restrict__ seg_start)
{
  __shared__ unsigned s_tot[1024];
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + (int)threadIdx.x;
    unsigned run = 0;
    if (b < nb)
      for (int k = 0; k < nblocks; ++k) {
        const unsigned c = block_counts[(size_t)k * nb + b];
        block_counts[(size_t)k * nb + b] = run;
        run += c;
      }
    s_tot[threadIdx.x] = run;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned acc = (b0 == 0) ? 0u : (unsigned)seg_start[b0];
      for (int q = 0; q < 1024 && b0 + q < nb; ++q) {
        seg_start[b0 + q] = (int32_t)acc;
        acc += s_tot[q];
      }
      seg_start[min(nb, b0 + 1024)] = (int32_t)acc;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(64) k_seg_scatter(const int16_t *__restrict__ ids, const uint8_t *__restrict__ skip,
                                                    uint32_t n, int nb, const uint32_t *__restrict__ block_off,
                                                    const int32_t *__restrict__ seg_start, int32_t *__restrict__ idx,
                                                    int32_t *__restrict__ seg_of)
{
  extern __shared__ unsigned s_cur[];   // slots this block has already filled, per bbox
  const int lane = threadIdx.x;
  for (int b = lane; b < nb; b += 64) s_cur[b] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSegBlock;
  for (int k0 = 0; k0 < kSegBlock; k0 += 64) {
    const size_t i = base + k0 + lane;
    int id = -1;
    if (i < n) {
      id = (skip && skip[i]) ? -1 : (int)ids[i];
      if (id >= nb) id = -1;
    }
    unsigned long long todo = __ballot(id >= 0);
    while (todo) {   // one round per distinct bbox id of this batch
      const int leader = __ffsll((long long)todo) - 1;
      const int idl = __builtin_amdgcn_readlane(id, leader);
      const unsigned long long same = __ballot(id == idl);
      if (id == idl) {
        const unsigned rank = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        const unsigned pos = (unsigned)seg_start[idl] + block_off[(size_t)blockIdx.x * nb + idl] + s_cur[idl] + rank;
        idx[pos] = (int32_t)i;
        seg_of[pos] = idl;
      }
      __syncthreads();   // (one wavefront: orders the cursor read above against the update below)
      if (lane == leader) s_cur[idl] += (unsigned)__popcll(same);
      __syncthreads();
      todo &= ~same;
    }
  }
}

// camera-frame coordinates of the selected points, in segment order
__global__ void __launch_bounds__(256) k_gather_cam(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, Mat34f m, const int32_t *__restrict__ idx,
                                                    const int32_t *__restrict__ seg_start, int nb, float *__restrict__ ox,
                                                    float *__restrict__ oy, float *__restrict__ oz)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= seg_start[nb]) return;
  const int j = idx[i];
  float a, b, c;
  xform34(m, x[j], y[j], z[j], a, b, c);
  ox[i] = a; oy[i] = b; oz[i] = c;
}

// bboxPoseEstimation :156-181 + computePCABoundingBox :187-247 for one bbox cloud, one wavefront each.
// The reference accumulates in cloud order -- pcl::compute3DCentroid and cv::PCA's mean in fp32, the
// covariance in fp64 -- so the sums are order dependent: the wavefront loads 64 points at a time and every
// lane adds them in order from cross-lane reads (all lanes hold the same running sums); the projections'
// min / max are order free and reduced in parallel.
__device__ __forceinline__ float lane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__global__ void __launch_bounds__(64) k_pca_bbox(const float *__restrict__ gx, const float *__restrict__ gy,
                                                 const float *__restrict__ gz, const uint8_t *__restrict__ keep,
                                                 const int32_t *__restrict__ seg_start, int nb,
                                                 gv_lshape_pose *__restrict__ poses, uint8_t *__restrict__ valid)
{
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= nb) return;
  const int s0 = seg_start[b], s1 = seg_start[b + 1];
  // pass 1: fp32 running sums in order (y: the centroid's y; z, x: the PCA rows (z, x), :167-172)
  float cy = 0.0f, m0 = 0.0f, m1 = 0.0f;
  int cnt = 0;
  for (int j0 = s0; j0 < s1; j0 += 64) {
    const int j = j0 + lane;
    const bool k = (j < s1) && keep[j];
    const float xv = k ? gx[j] : 0.f, yv = k ? gy[j] : 0.f, zv = k ? gz[j] : 0.f;
    unsigned long long mk = __ballot(k);
    while (mk) {
      const int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      cy = cy + lane_f32(yv, l);
      m0 = m0 + lane_f32(zv, l);
      m1 = m1 + lane_f32(xv, l);
      ++cnt;
    }
  }
  if (cnt == 0) {   // :174-175 empty cloud: no pose
    if (lane == 0) { valid[b] = 0; poses[b] = gv_lshape_pose{}; }
    return;
  }
  cy = cy / (float)cnt;
  const float inv = (float)(1.0 / (double)cnt);
  m0 = m0 * inv;
  m1 = m1 * inv;
  // pass 2: fp64 covariance sums of the fp32-centred samples, in order
  double c00 = 0.0, c01 = 0.0, c11 = 0.0;
  for (int j0 = s0; j0 < s1; j0 += 64) {
    const int j = j0 + lane;
    const bool k = (j < s1) && keep[j];
    const float av = k ? (gz[j] - m0) : 0.f, bv = k ? (gx[j] - m1) : 0.f;
    unsigned long long mk = __ballot(k);
    while (mk) {
      const int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      const float a = lane_f32(av, l), bb = lane_f32(bv, l);
      c00 = c00 + (double)a * a;
      c01 = c01 + (double)a * bb;
      c11 = c11 + (double)bb * bb;
    }
  }
  const double sc = 1.0 / (double)cnt;
  const double a = (double)(float)(c00 * sc), bq = (double)(float)(c01 * sc), d = (double)(float)(c11 * sc);
  double mjx, mjy;
  if (bq == 0.0) {
    if (a >= d) { mjx = 1; mjy = 0; } else { mjx = 0; mjy = 1; }
  } else {
    const double tr = a + d, df = a - d;
    const double root = sqrt(df * df + 4.0 * bq * bq);
    const double l1 = 0.5 * (tr + root);
    mjx = bq; mjy = l1 - a;
    if (fabs(l1 - d) > fabs(mjy)) { mjx = l1 - d; mjy = bq; }
    const double nn = sqrt(mjx * mjx + mjy * mjy);
    mjx /= nn; mjy /= nn;
  }
  if (mjx < 0 || (mjx == 0 && mjy < 0)) { mjx = -mjx; mjy = -mjy; }
  const float Mx = (float)mjx, My = (float)mjy, Nx = (float)(-mjy), Ny = (float)mjx;
  // pass 3: extent of the projections (:203-216), order free
  float minL = 3.402823466e+38f, maxL = -3.402823466e+38f, minW = 3.402823466e+38f, maxW = -3.402823466e+38f;
  for (int j = s0 + lane; j < s1; j += 64) {
    if (!keep[j]) continue;
    const float dx = gz[j] - m0, dy = gx[j] - m1;
    const float pl = dx * Mx + dy * My, pw = dx * Nx + dy * Ny;
    minL = fminf(minL, pl); maxL = fmaxf(maxL, pl);
    minW = fminf(minW, pw); maxW = fmaxf(maxW, pw);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    minL = fminf(minL, __shfl_xor(minL, off)); maxL = fmaxf(maxL, __shfl_xor(maxL, off));
    minW = fminf(minW, __shfl_xor(minW, off)); maxW = fmaxf(maxW, __shfl_xor(maxW, off));
  }
  if (lane == 0) {
    // :227 degrees, :236 passed to setRPY as if radians (as the reference does); atan2f evaluated in fp64, rounded once
    const float angle = (float)atan2((double)My, (double)Mx) * 180.0f / (float)3.14159265358979323846;
    const double hp = (double)(-angle) * 0.5;   // tf2 setRPY(0, pitch, 0): (0, sin(p/2), 0, cos(p/2))
    gv_lshape_pose p{};
    p.px = m1;    // :230 center.y
    p.py = cy;    // :231 then :181
    p.pz = m0;    // :232 center.x
    p.qx = 0.0; p.qy = sin(hp); p.qz = 0.0; p.qw = cos(hp);
    p.length = maxL - minL;   // :218,:243
    p.width = maxW - minW;    // :219,:244
    p.height = 0.0;           // never set on this path in the reference
    poses[b] = p;
    valid[b] = 1;
  }
}

void launch_split_by_bbox(const int16_t *ids, const uint8_t *skip, uint32_t n, int nb, uint32_t *block_counts,
                          int32_t *seg_start, int32_t *idx, int32_t *seg_of, hipStream_t s)
{
  const int nblocks = (int)(((size_t)n + kSegBlock - 1) / kSegBlock);
  if (nblocks == 0 || nb <= 0) return;
  hipLaunchKernelGGL(k_seg_count, dim3(nblocks), dim3(256), (size_t)nb * sizeof(unsigned), s, ids, skip, n, nb, block_counts);
  hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, s, block_counts, nblocks, nb, seg_start);
  hipLaunchKernelGGL(k_seg_scatter, dim3(nblocks), dim3(64), (size_t)nb * sizeof(unsigned), s, ids, skip, n, nb, block_counts,
                     seg_start, idx, seg_of);
}

void launch_gather_cam(const float *x, const float *y, const float *z, const Mat34f &m, const int32_t *idx,
                       const int32_t *seg_start, int nb, uint32_t n_max, float *ox, float *oy, float *oz, hipStream_t s)
{
  if (!n_max) return;
  hipLaunchKernelGGL(k_gather_cam, dim3((n_max + 255) / 256), dim3(256), 0, s, x, y, z, m, idx, seg_start, nb, ox, oy, oz);
}

void launch_pca_bbox(const float *gx, const float *gy, const float *gz, const uint8_t *keep, const int32_t *seg_start, int nb,
                     gv_lshape_pose *poses, uint8_t *valid, hipStream_t s)
{
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_pca_bbox, dim3(nb), dim3(64), 0, s, gx, gy, gz, keep, seg_start, nb, poses, valid);
}

}  // namespace gv
