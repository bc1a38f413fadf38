// gv_cloudops.hip -- device-resident halves of the "next tier" cloud_detections calls (SURVEY 8(f)-2,
// 8(f)-3): RANSAC ground plane (segmentGroundPlane, src/cloud_detections.cpp:105-138) and the per-bbox
// cloud split + radius filter + PCA rectangle (extractCloudPerBBox / bboxPoseEstimation /
// computePCABoundingBox, :140-298).  Nothing here moves O(N) bytes to the host.
//
// Round 3: every step is one sweep of the cloud or less --
//   RANSAC: all hypotheses are counted in ONE pass (planes in LDS, one transform per point); the
//     refinement is ONE pass of fp64 raw moments whose fixed 64-ary tree is finished by the last workgroup
//     to arrive (no tree-level launches);
//   radius filter: the selected points are counting-sorted by (0.5 m cell, bbox) bucket and queried in that
//     order (<= 27 cells per query, early exit, wavefront-coherent loads) instead of all pairs of a bbox;
//   PCA: only the points the filter keeps are split by bbox (stable, cloud order), and the reference's
//     order-dependent fp32 / fp64 sums run as three sequential lane chains fed through LDS.
// gfx950, wave64, built with -ffp-contract=off.
#include "gv_kernels.hpp"
#include "gv_device.hpp"

#include <algorithm>

namespace gv {

constexpr int kCoThreads = 1024;          // workgroup of the sweep kernels
constexpr int kCoPts = 4;                 // points per thread
constexpr int kCoBlock = kCoThreads * kCoPts;   // 4096 points = 64 level-0 groups of the sum tree

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
// fp32).  An unusable sample gets a NaN plane: it can never collect an inlier.  Every workgroup of the counting
// pass makes the planes itself (150 loads: cheaper than a launch of its own and the 4 us of a one-wavefront kernel).
__device__ __forceinline__ float4 ransac_hypothesis(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, uint32_t n, const Mat34f &m,
                                                    unsigned long long seed, int t)
{
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
  return out;
}

// |n.p + d| < thr with the oracle's operation order.  thr_f is the smallest float >= the fp64 threshold:
// for a float f, f < thr_f <=> (double)f < thr (the reference compares the fp32 distance with the double).
__device__ __forceinline__ bool plane_inlier_dev(const float4 &pl, float px, float py, float pz, float thr_f)
{
  const float d = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pl.x, px), __fmul_rn(pl.y, py)), __fmul_rn(pl.z, pz)), pl.w);
  return fabsf(d) < thr_f;   // NaN -> false
}

// counts[h] += inliers of hypothesis h, ALL hypotheses in one pass over the cloud: a workgroup transforms its
// 4096 points once (4 per thread, in registers) and walks the planes, which sit in LDS; per plane and
// wavefront the four ballots are counted on the scalar unit and one lane adds them to the plane's LDS counter.
__global__ void __launch_bounds__(kCoThreads) k_ransac_count_all(const float *__restrict__ x, const float *__restrict__ y,
                                                                 const float *__restrict__ z, uint32_t n, Mat34f m,
                                                                 unsigned long long seed, int h0, int iters, float thr_f,
                                                                 unsigned *__restrict__ counts, int stride)
{
  extern __shared__ float4 s_pl[];   // iters planes, then iters counters
  unsigned *s_cnt = reinterpret_cast<unsigned *>(s_pl + iters);
  const int tid = threadIdx.x;
  for (int t = tid; t < iters; t += kCoThreads) { s_pl[t] = ransac_hypothesis(x, y, z, n, m, seed, h0 + t); s_cnt[t] = 0u; }
  const float qnan = __uint_as_float(0x7fc00000u);
  float px[kCoPts], py[kCoPts], pz[kCoPts];
  const size_t base = (size_t)blockIdx.x * kCoBlock;
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    const size_t i = base + (size_t)j * kCoThreads + tid;
    px[j] = py[j] = pz[j] = qnan;   // a NaN point is nobody's inlier
    if (i < n) xform34(m, x[i], y[i], z[i], px[j], py[j], pz[j]);
  }
  __syncthreads();
  // two points per packed-fp32 instruction (v_pk_mul_f32 / v_pk_add_f32: the same separately rounded products and
  // sums as plane_inlier_dev, at twice the rate)
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f ax = {px[0], px[1]}, ay = {py[0], py[1]}, az = {pz[0], pz[1]};
  const v2f bx = {px[2], px[3]}, by = {py[2], py[3]}, bz = {pz[2], pz[3]};
  static_assert(kCoPts == 4, "two packed pairs per thread");
  for (int h = 0; h < iters; ++h) {
    const float4 pl = s_pl[h];
    const v2f nx = {pl.x, pl.x}, ny = {pl.y, pl.y}, nz = {pl.z, pl.z}, nw = {pl.w, pl.w};
    const v2f da = ((nx * ax + ny * ay) + nz * az) + nw;
    const v2f db = ((nx * bx + ny * by) + nz * bz) + nw;
    unsigned c = (unsigned)__popcll(__ballot(fabsf(da.x) < thr_f)) + (unsigned)__popcll(__ballot(fabsf(da.y) < thr_f)) +
                 (unsigned)__popcll(__ballot(fabsf(db.x) < thr_f)) + (unsigned)__popcll(__ballot(fabsf(db.y) < thr_f));
    if ((tid & 63) == 0 && c) atomicAdd(&s_cnt[h], c);
  }
  __syncthreads();
  for (int t = tid; t < iters; t += kCoThreads)
    if (s_cnt[t]) atomicAdd(&counts[(size_t)(blockIdx.x % kRansacCountSlices) * stride + t], s_cnt[t]);   // sliced: see gv_kernels.hpp
}

// unit eigenvector of the smallest eigenvalue of a symmetric 3x3 (fp64): oracle/ransac.c, operation for operation -- the
// closed form of pcl::eigen33 (scaled matrix, trigonometric roots, largest cross product of rows of A - lambda I).
// Rounds 2-3 ran a cyclic Jacobi here: ninety dependent fp64 divisions and square roots on one lane, 14 us of the
// kernel's 27.  (atan2 / cos / sin may differ from glibc's in the last bit: 1e-16 on lambda, far below the fp32 the
// plane is stored in; the tests accept a last-bit difference of the coefficients and say so.)
__device__ __forceinline__ void smallest_eigenvector3_dev(const double cov[6], double v[3])
{
  double a00 = cov[0], a01 = cov[1], a02 = cov[2], a11 = cov[3], a12 = cov[4], a22 = cov[5];
  double scale = fmax(fmax(fmax(fabs(a00), fabs(a01)), fmax(fabs(a02), fabs(a11))), fmax(fabs(a12), fabs(a22)));
  if (!(scale > 2.2250738585072014e-308)) scale = 1.0;
  a00 = a00 / scale; a01 = a01 / scale; a02 = a02 / scale; a11 = a11 / scale; a12 = a12 / scale; a22 = a22 / scale;
  const double c0 = (((a00 * a11) * a22 + (2.0 * a01) * a02 * a12) - (a00 * a12) * a12 - (a11 * a02) * a02) - (a22 * a01) * a01;
  const double c1 = ((((a00 * a11 - a01 * a01) + a00 * a22) - a02 * a02) + a11 * a22) - a12 * a12;
  const double c2 = (a00 + a11) + a22;
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
  double lam = r0;
  if (r1 < lam) lam = r1;
  if (r2 < lam) lam = r2;
  if (lam <= 0.0) lam = 0.0;
  a00 -= lam; a11 -= lam; a22 -= lam;
  const double v1[3] = {a01 * a12 - a02 * a11, a02 * a01 - a00 * a12, a00 * a11 - a01 * a01};
  const double v2[3] = {a01 * a22 - a02 * a12, a02 * a02 - a00 * a22, a00 * a12 - a01 * a02};
  const double v3[3] = {a11 * a22 - a12 * a12, a12 * a02 - a01 * a22, a01 * a12 - a11 * a02};
  const double l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2];
  const double l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2];
  const double l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
  const bool p1 = l1 >= l2 && l1 >= l3, p2 = !p1 && l2 >= l1 && l2 >= l3;
  double nn[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) nn[k] = p1 ? v1[k] : (p2 ? v2[k] : v3[k]);
  const double len = sqrt(p1 ? l1 : (p2 ? l2 : l3));
  // sign: the component of largest magnitude (the first of equals) is made positive
  double big = nn[0];
  if (fabs(nn[1]) > fabs(big)) big = nn[1];
  if (fabs(nn[2]) > fabs(big)) big = nn[2];
  const double sg = (big < 0) ? -1.0 / len : 1.0 / len;
  v[0] = nn[0] * sg; v[1] = nn[1] * sg; v[2] = nn[2] * sg;
}

constexpr int kMom = 10;   // Sx Sy Sz Qxx Qxy Qxz Qyy Qyz Qzz + the inlier count (as a double: exact)

// lane i <- lane i + N inside its row of 16 (DPP row_shl:N; lanes whose source falls outside read 0.0 and hold
// values nobody uses afterwards)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// The tree's butterfly v[i] += v[i + off], off = 32 .. 1; lane 0 ends with the sum of the 64 values in the
// oracle's order (tree_sum64, oracle/ransac.c).  Only lanes below `off` matter at each step, so the four
// in-row steps are DPP shifts on the operand (no LDS traffic); the two cross-row steps go through the
// permute network.
__device__ __forceinline__ double wave_butterfly(double v)
{
  v = v + __shfl_down(v, 32);
  v = v + __shfl_down(v, 16);
  v = v + dpp_f64<0x108>(v);   // row_shl:8
  v = v + dpp_f64<0x104>(v);   // row_shl:4
  v = v + dpp_f64<0x102>(v);   // row_shl:2
  v = v + dpp_f64<0x101>(v);   // row_shl:1
  return v;
}

// Selection + refinement (optimizeModelCoefficients) in one pass.  Every workgroup finds the best hypothesis
// (most inliers, first wins ties) from the counts, then accumulates the fp64 raw moments of that plane's
// inliers by the fixed 64-ary tree of oracle/ransac.c: level 0 = 64 consecutive points reduced by the
// wavefront butterfly v[i] += v[i + off], off = 32 .. 1 (non-inliers and padding contribute +0.0); a
// workgroup holds 64 level-0 groups, so its level-1 sum is one more butterfly out of LDS.  The workgroup
// whose ticket comes last runs the remaining levels and solves for the plane.  Same tree on both sides:
// the device rounds exactly as the oracle does, whatever the number of workgroups.
__global__ void __launch_bounds__(kCoThreads) k_ransac_moments(const float *__restrict__ x, const float *__restrict__ y,
                                                               const float *__restrict__ z, uint32_t n, Mat34f m,
                                                               unsigned long long seed, unsigned *__restrict__ counts,
                                                               int iters, float thr_f, double *__restrict__ part_a,
                                                               double *__restrict__ part_b, RansacState *__restrict__ st)
{
  __shared__ double s_l0[64][kMom];
  __shared__ unsigned s_bestc, s_last;
  __shared__ int s_best;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 64) {
    unsigned bc = 0;
    int bi = 0x7fffffff;
    for (int t = tid; t < iters; t += 64) {
      unsigned c = 0;
#pragma unroll
      for (int sl = 0; sl < kRansacCountSlices; ++sl) c += counts[(size_t)sl * iters + t];
      if (c > bc) { bc = c; bi = t; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned oc = __shfl_xor(bc, off);
      const int oi = __shfl_xor(bi, off);
      if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (tid == 0) { s_bestc = bc; s_best = bc ? bi : 0; }
  }
  __syncthreads();
  const unsigned bestc = s_bestc;
  const float4 pl = bestc ? ransac_hypothesis(x, y, z, n, m, seed, s_best) : make_float4(0.f, 0.f, 0.f, 0.f);
  // A wavefront owns four level-0 groups (256 consecutive points).  Row q of 16 lanes takes group 4 w + q, and
  // lane r of the row the group's points r, r + 16, r + 32, r + 48: the butterfly's steps 32 and 16 are then adds
  // between the lane's own four values, and steps 8 .. 1 are DPP shifts inside the row -- for the four groups at
  // once, nothing through the permute network.  Same pairs, same order of additions as tree_sum64.
  {
    const int q = lane >> 4, r = lane & 15;
    const int g = w * kCoPts + q;   // level-0 group of this workgroup: points [64 g, 64 g + 64) of its 4096
    const size_t i0 = (size_t)blockIdx.x * kCoBlock + (size_t)g * 64 + r;
    float px[4], py[4], pz[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t i = i0 + 16 * t;
      const bool in = i < n && bestc;
      px[t] = in ? x[i] : 0.f;
      py[t] = in ? y[i] : 0.f;
      pz[t] = in ? z[i] : 0.f;
    }
    auto moments = [&](int t, double *v) {
#pragma unroll
      for (int k = 0; k < kMom; ++k) v[k] = 0.0;
      float cx, cy, cz;
      xform34(m, px[t], py[t], pz[t], cx, cy, cz);
      if (i0 + 16 * t < n && bestc && plane_inlier_dev(pl, cx, cy, cz, thr_f)) {
        const double dx = (double)cx, dy = (double)cy, dz = (double)cz;
        v[0] = dx; v[1] = dy; v[2] = dz;
        v[3] = dx * dx; v[4] = dx * dy; v[5] = dx * dz; v[6] = dy * dy; v[7] = dy * dz; v[8] = dz * dz;
        v[9] = 1.0;
      }
    };
    double a[kMom], b[kMom], c[kMom];
    moments(0, a);
    moments(2, c);
#pragma unroll
    for (int k = 0; k < kMom; ++k) a[k] = a[k] + c[k];   // step 32: v[r] += v[r + 32]
    moments(1, b);
    moments(3, c);
#pragma unroll
    for (int k = 0; k < kMom; ++k) b[k] = b[k] + c[k];   //          v[r + 16] += v[r + 48]
#pragma unroll
    for (int k = 0; k < kMom; ++k) {
      double v = a[k] + b[k];        // step 16
      v = v + dpp_f64<0x108>(v);     // row_shl:8
      v = v + dpp_f64<0x104>(v);     // row_shl:4
      v = v + dpp_f64<0x102>(v);     // row_shl:2
      v = v + dpp_f64<0x101>(v);     // row_shl:1
      if (r == 0) s_l0[g][k] = v;
    }
  }
  __syncthreads();
  if (w < kMom) {   // level 1: one wavefront per quantity
    const double t = wave_butterfly(s_l0[lane][w]);
    // (agent-scope atomic store: written through to the device's coherence point, where the last workgroup's atomic
    //  loads read it -- see the note at the ticket)
    if (lane == 0) __hip_atomic_store(&part_a[(size_t)blockIdx.x * kMom + w], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // The ticket.  What the last workgroup reads -- the partial sums -- was written by agent-scope atomic stores and is
  // read by agent-scope atomic loads; the ticket itself only has to come after this workgroup's stores have been
  // acknowledged (vmcnt, then the barrier).  A release FENCE instead (rounds 2-3) is a write-back of the whole L2 of the
  // XCD by every workgroup (buffer_wbl2): 10 us of this kernel, 15 us of k_pca_extent (profiles/r04/ticket_fence_ab.txt).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == gridDim.x - 1u;
    if (last) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    s_last = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // levels 2..: groups of 64 partial sums per wavefront.
  size_t cnt = gridDim.x;
  __shared__ double s_tot[kMom];
  const bool short_tree = cnt > 1 && cnt <= 4096;   // up to 16.7 M points: levels 2 and 3 without a trip through memory
  if (short_tree) {
    // ONE round of loads (every wavefront its groups, all ten quantities in flight), level-2 sums into LDS, the
    // level-3 butterfly out of LDS: the tail used to be store -> wait -> barrier -> load per level (10 us of
    // dependent round trips for 245 partial sums)
    const int groups = (int)((cnt + 63) / 64);
    for (int g = w; g < groups; g += kCoThreads / 64) {
      const size_t i = (size_t)g * 64 + lane;
      double v[kMom];
#pragma unroll
      for (int k = 0; k < kMom; ++k)
        v[k] = (i < cnt) ? __hip_atomic_load(&part_a[i * kMom + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
      for (int k = 0; k < kMom; ++k) v[k] = wave_butterfly(v[k]);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kMom; ++k) s_l0[g][k] = v[k];
      }
    }
    __syncthreads();
    if (groups > 1) {   // level 3: the group sums and zeros, as the oracle's tree pads them
      if (w < kMom) {
        const double t = wave_butterfly(lane < groups ? s_l0[lane][w] : 0.0);
        if (lane == 0) s_tot[w] = t;
      }
    } else if (tid < kMom) {
      s_tot[tid] = s_l0[0][tid];
    }
    __syncthreads();
  } else {
    // the general tree: ping-pong between the two scratch arrays (agent-scope accesses: other wavefronts of this
    // workgroup wrote them)
    const double *src = part_a;
    double *dst = part_b;
    while (cnt > 1) {
      const size_t groups = (cnt + 63) / 64;
      for (size_t g = (size_t)w; g < groups; g += kCoThreads / 64) {
        const size_t i = g * 64 + lane;
        double v[kMom];
#pragma unroll
        for (int k = 0; k < kMom; ++k)
          v[k] = (i < cnt) ? __hip_atomic_load(&src[i * kMom + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
        for (int k = 0; k < kMom; ++k) v[k] = wave_butterfly(v[k]);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < kMom; ++k) __hip_atomic_store(&dst[g * kMom + k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const double *t = src;
      src = dst;
      dst = const_cast<double *>(t);
      cnt = groups;
    }
    // one workgroup: its level-1 sums ARE the totals only if the tree had a single level-1 group; the oracle's
    // tree always takes at least one level per 64 values, which the loop above reproduces for cnt > 1 and which
    // is the identity (sum of one value and 63 zeros) for cnt == 1
    if (tid < kMom) s_tot[tid] = __hip_atomic_load(&src[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  // every workgroup has read the counts: cleared here for the next call (no memset launch; nobody waits for these)
  for (int t = tid; t < iters * kRansacCountSlices; t += kCoThreads) counts[t] = 0u;
  if (tid == 0) {
    double mom[kMom];
    for (int k = 0; k < kMom; ++k) mom[k] = s_tot[k];
    const unsigned long long mi = (unsigned long long)mom[9];
    float4 refined = pl;
    if (bestc && mi >= 3) {
      const double dm = (double)mi;
      const double cx = mom[0] / dm, cy = mom[1] / dm, cz = mom[2] / dm;
      const double cov[6] = {mom[3] / dm - cx * cx, mom[4] / dm - cx * cy, mom[5] / dm - cx * cz,
                             mom[6] / dm - cy * cy, mom[7] / dm - cy * cz, mom[8] / dm - cz * cz};
      double nv[3];
      smallest_eigenvector3_dev(cov, nv);
      refined = make_float4((float)nv[0], (float)nv[1], (float)nv[2], (float)(-((nv[0] * cx + nv[1] * cy) + nv[2] * cz)));
    }
    st->best_count = bestc;
    st->plane = pl;
    st->refined = refined;
    st->m = bestc ? mi : 0ull;
    st->n_inliers = 0ull;   // counted by the mask / classify pass that follows
  }
}

// inliers of the refined plane: mask[i] (device resident) and their number
__global__ void __launch_bounds__(kCoThreads) k_ransac_mask(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, uint32_t n, Mat34f m, float thr_f,
                                                            RansacState *__restrict__ st, uint8_t *__restrict__ mask,
                                                            RansacState *__restrict__ st_copy, CallDone done)
{
  __shared__ unsigned s_n;
  const float4 pl = st->refined;
  const bool have = st->best_count != 0;
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    const size_t i = (size_t)blockIdx.x * kCoBlock + (size_t)j * kCoThreads + threadIdx.x;
    bool in = false;
    if (i < n) {
      float cx, cy, cz;
      xform34(m, x[i], y[i], z[i], cx, cy, cz);
      in = have && plane_inlier_dev(pl, cx, cy, cz, thr_f);
      mask[i] = in ? 1 : 0;
    }
    c += (unsigned)__popcll(__ballot(in));
  }
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_n, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_n) atomicAdd(&st->n_inliers, (unsigned long long)s_n);
    if (st_copy) {   // the workgroup that finishes last hands the final state out (the count is an agent-scope atomic:
      // no fence, only this thread's atomic acknowledged before its ticket)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gridDim.x - 1u) {
        __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RansacState o = *st;
        o.n_inliers = __hip_atomic_load(&st->n_inliers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o.ticket = 0u;
        *st_copy = o;
        call_done(done, 1u);
      }
    }
  }
}

size_t ransac_scratch_doubles(size_t n)
{
  const size_t nblk = (n + kCoBlock - 1) / kCoBlock;
  return (size_t)kMom * (nblk + (nblk + 63) / 64 + 2) + 64;
}

void launch_ransac_plane(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, float thr_f, int iters,
                         unsigned long long seed, float4 *planes, unsigned *counts, double *scratch, RansacState *st,
                         hipStream_t s)
{
  const uint32_t nblk = (uint32_t)(((size_t)n + kCoBlock - 1) / kCoBlock);
  (void)planes;   // (the hypotheses are no longer materialised: every workgroup makes the ones it needs)
  for (int h0 = 0; h0 < iters; h0 += 2048) {   // planes + counters of one launch sit in LDS (20 bytes per hypothesis)
    const int hn = std::min(2048, iters - h0);
    hipLaunchKernelGGL(k_ransac_count_all, dim3(nblk), dim3(kCoThreads), (size_t)hn * (sizeof(float4) + sizeof(unsigned)), s, x, y, z,
                       n, m_cam, seed, h0, hn, thr_f, counts + h0, iters);
  }
  double *part_a = scratch, *part_b = scratch + (size_t)kMom * (nblk + 1);
  hipLaunchKernelGGL(k_ransac_moments, dim3(nblk), dim3(kCoThreads), 0, s, x, y, z, n, m_cam, seed, counts, iters, thr_f, part_a,
                     part_b, st);
}

void launch_ransac_mask(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, float thr_f,
                        RansacState *st, uint8_t *mask, RansacState *st_copy, const CallDone &done, hipStream_t s)
{
  const uint32_t nblk = (uint32_t)(((size_t)n + kCoBlock - 1) / kCoBlock);
  hipLaunchKernelGGL(k_ransac_mask, dim3(nblk), dim3(kCoThreads), 0, s, x, y, z, n, m_cam, thr_f, st, mask, st_copy, done);
}

// --------------------------------------------- per-bbox clouds: classify + cell hash --
// Cells of the radius filter: cubes of kCellSize over the camera frame, points chained per (cell, bbox id)
// bucket.  The bucket keeps the low three bits of each cell coordinate, so cells that differ by less than 8 in
// every coordinate -- all a query visits -- always land in DIFFERENT buckets and no point can be met twice;
// whatever else shares a bucket (another bbox, a far cell) fails the id or the distance test.
constexpr float kCellSize = 0.5f;
__device__ __forceinline__ int cell_of(float c)
{
  const float q = floorf(__fmul_rn(c, 1.0f / kCellSize));
  return (int)fminf(fmaxf(q, -1.0e9f), 1.0e9f);
}
__device__ __forceinline__ uint32_t bucket_of(int ix, int iy, int iz, int id, uint32_t hi_mask)
{
  const uint32_t lo = ((uint32_t)ix & 7u) | (((uint32_t)iy & 7u) << 3) | (((uint32_t)iz & 7u) << 6);
  uint32_t h = (uint32_t)(ix >> 3) * 0x9E3779B1u ^ (uint32_t)(iy >> 3) * 0x85EBCA77u ^ (uint32_t)(iz >> 3) * 0xC2B2AE3Du ^
               (uint32_t)id * 0x27D4EB2Fu;
  h ^= h >> 15;
  return lo | ((h & hi_mask) << 9);
}
// Cells [lo, hi] along one axis that can hold a point within the radius of coordinate c.  The fp32 distance
// test accepts |dc| <= 0.4 (1 + 2e-7); rr adds that, the rounding of c + rr (6e-8 |c|) and of the cell product,
// and cell_of is monotone, so every accepted neighbour's cell lies in the range for any finite c (at most
// 3 cells for |c| < 1e5 m; the range is capped at own cell +- 3, exact below 1e6 m).
__device__ __forceinline__ void cell_range(float c, int own, int &lo, int &hi)
{
  const float rr = 0.4000005f + 2.5e-7f * fabsf(c);
  lo = max(cell_of(c - rr), own - 3);
  hi = min(cell_of(c + rr), own + 3);
}

// extractCloudPerBBox (:250-298) on the cloud with the ground removed (:306-314): per point the camera
// transform, the ground test against the refined plane (use_plane) and the first-match bbox.  A selected point
// is counted into its (cell, id) bucket.  Also counts the ground points (st->n_inliers) when the plane is in use.
// Phase by phase over the thread's four points (loads, transforms + ground test + projection, bbox test, stores +
// atomics) so that the four dependent chains overlap; the bbox test's tables (float thresholds, 16x16-pixel tile
// masks) are staged in LDS when they fit (LDS_TAB): the test's two dependent look-ups per candidate are then LDS
// round trips, not trips to the L2.  29 -> 17 us on the objects scene (profiles/r04/).
template <bool LDS_TAB>
__global__ void __launch_bounds__(kCoThreads) k_pose_classify(const float *__restrict__ x, const float *__restrict__ y,
                                                              const float *__restrict__ z, uint32_t n, Mat34f m, CamK cam,
                                                              BBoxTest bt, int nb, int use_plane, float thr_f,
                                                              RansacState *__restrict__ st, int16_t *__restrict__ ids,
                                                              uint32_t *__restrict__ cell_cnt, uint32_t *__restrict__ ticket_of,
                                                              uint32_t hi_mask)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char s_tab[];   // LDS_TAB: nb_pad float4 thresholds, then the tile masks
  __shared__ unsigned s_n;
  const int nb_pad = (nb + 3) & ~3;
  float4 *s_bf = reinterpret_cast<float4 *>(s_tab);
  unsigned long long *s_tm = reinterpret_cast<unsigned long long *>(s_tab + (size_t)nb_pad * sizeof(float4));
  if (LDS_TAB) {
    for (int b = threadIdx.x; b < nb; b += kCoThreads) s_bf[b] = bt.bbox_f[b];
    const int nmask = bt.tiles_x * bt.tiles_y * bt.mask_words;
    for (int i = threadIdx.x; i < nmask; i += kCoThreads) s_tm[i] = bt.tile_mask[i];
  }
  float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
  bool have = false;
  if (use_plane) {
    pl = st->refined;
    have = st->best_count != 0;
  }
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  // ---- loads
  size_t idx[kCoPts];
  bool in[kCoPts];
  float lx[kCoPts], ly[kCoPts], lz[kCoPts];
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    idx[j] = (size_t)blockIdx.x * kCoBlock + (size_t)j * kCoThreads + threadIdx.x;
    in[j] = idx[j] < n;
    const size_t i = in[j] ? idx[j] : (size_t)0;
    lx[j] = x[i]; ly[j] = y[i]; lz[j] = z[i];
  }
  // ---- camera transform, ground test, projection (first_bbox's first half, gv_device.hpp)
  float cx[kCoPts], cy[kCoPts], cz[kCoPts], u[kCoPts], v[kCoPts];
  bool ground[kCoPts], cand[kCoPts];
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    xform34(m, lx[j], ly[j], lz[j], cx[j], cy[j], cz[j]);
    ground[j] = in[j] && have && plane_inlier_dev(pl, cx[j], cy[j], cz[j], thr_f);
    c += (unsigned)__popcll(__ballot(ground[j]));
    cand[j] = in[j] && !ground[j] && isfinite(cx[j]) && isfinite(cy[j]) && isfinite(cz[j]) && !(cz[j] <= 0.001f);   // :264
    u[j] = v[j] = -1.0f;
  }
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    if (cand[j]) {
      const double X = (double)cx[j], Y = (double)cy[j], Z = (double)cz[j];
      const double riz = rcp_newton(Z);
      div2_to_float(cam.k[0] * X + cam.k[2] * Z, cam.k[4] * Y + cam.k[5] * Z, Z, riz, u[j], v[j]);   // :268-273
      cand[j] = !(u[j] < 0 || u[j] >= (float)cam.W || v[j] < 0 || v[j] >= (float)cam.H);             // :276
    }
  }
  // ---- first match (:280-288)
  int id[kCoPts];
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    id[j] = -1;
    if (cand[j]) {
      const int tx = (int)u[j] >> 4, ty = (int)v[j] >> 4;
      const size_t off = ((size_t)ty * bt.tiles_x + tx) * bt.mask_words;
      for (int wd = 0; wd < bt.mask_words && id[j] < 0; ++wd) {
        unsigned long long mk = LDS_TAB ? s_tm[off + wd] : bt.tile_mask[off + wd];
        while (mk) {
          const int b = wd * 64 + (__ffsll((long long)mk) - 1);
          mk &= mk - 1;
          const float4 f = LDS_TAB ? s_bf[b] : bt.bbox_f[b];
          if (u[j] >= f.x && u[j] <= f.z && v[j] >= f.y && v[j] <= f.w) {
            id[j] = b;
            break;
          }
        }
      }
      if (id[j] >= nb) id[j] = -1;
    }
  }
  // ---- bucket counts: the value the atomic returns is the point's slot inside its bucket -- kept, so that the
  // scatter pass needs no atomic of its own (one global atomic per selected point instead of two: they bound both
  // passes, ~10 G/s on random addresses) -- then the stores
  uint32_t tk[kCoPts];
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    tk[j] = 0u;
    if (id[j] >= 0) tk[j] = atomicAdd(&cell_cnt[bucket_of(cell_of(cx[j]), cell_of(cy[j]), cell_of(cz[j]), id[j], hi_mask)], 1u);
  }
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    if (in[j]) {
      ids[idx[j]] = (int16_t)id[j];
      if (id[j] >= 0) ticket_of[idx[j]] = tk[j];
    }
  }
  if (use_plane) {
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&st->n_inliers, (unsigned long long)s_n);
  }
}

// Exclusive prefix of the bucket counts, one launch: every workgroup scans its 4096 buckets (pre[b] = points
// of its earlier buckets) and publishes its total; the workgroup whose ticket comes last turns the totals into
// block offsets: first slot of bucket b = pre[b] + blk_off[b >> 12] (bucket_start below), blk_off[n_buckets >> 12] =
// the number of selected points, pre[n_buckets] = 0 so that the formula also gives the end of the last bucket.
constexpr int kScanBlock = 4096;
__global__ void __launch_bounds__(1024) k_cell_scan(uint32_t *__restrict__ cell_cnt, uint32_t n_buckets,
                                                    uint32_t *__restrict__ pre, uint32_t *__restrict__ blk_off,
                                                    unsigned *__restrict__ ticket)
{
  __shared__ unsigned s_w[16];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t b0 = (size_t)blockIdx.x * kScanBlock + (size_t)tid * 4;
  const uint4 c = *reinterpret_cast<const uint4 *>(cell_cnt + b0);   // n_buckets is a multiple of 4096
  *reinterpret_cast<uint4 *>(cell_cnt + b0) = make_uint4(0u, 0u, 0u, 0u);   // ready for the next call (nobody else reads them)
  const unsigned t4 = c.x + c.y + c.z + c.w;
  unsigned inc = t4;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  unsigned wbase = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const unsigned v = s_w[q];
    if (q < w) wbase += v;
    total += v;
  }
  const unsigned e = wbase + inc - t4;
  *reinterpret_cast<uint4 *>(pre + b0) = make_uint4(e, e + c.x, e + c.x + c.y, e + c.x + c.y + c.z);
  if (tid == 0) {
    // (the last workgroup reads only these totals, by agent-scope atomic loads: no fence -- see k_ransac_moments)
    __hip_atomic_store(&blk_off[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = tk == gridDim.x - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // totals -> exclusive offsets (at most 2^25 / 4096 = 8192 workgroups: eight per thread)
  const int nblk = (int)gridDim.x;
  const int per = (nblk + 1023) / 1024;
  unsigned loc[8];
  unsigned run = 0;
  for (int q = 0; q < per; ++q) {
    const int k = tid * per + q;
    loc[q] = (k < nblk) ? __hip_atomic_load(&blk_off[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    run += loc[q];
  }
  unsigned inc2 = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(inc2, off);
    if (lane >= off) inc2 += o;
  }
  __syncthreads();
  if (lane == 63) s_w[w] = inc2;
  __syncthreads();
  unsigned wb2 = 0, tot2 = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const unsigned v = s_w[q];
    if (q < w) wb2 += v;
    tot2 += v;
  }
  unsigned acc = wb2 + inc2 - run;
  for (int q = 0; q < per; ++q) {
    const int k = tid * per + q;
    if (k < nblk) blk_off[k] = acc;
    acc += loc[q];
  }
  if (tid == 0) {
    blk_off[nblk] = tot2;
    pre[n_buckets] = 0u;
  }
}

// first slot of bucket b = its offset inside its scan block + the block's offset (the scan leaves the two apart: a pass
// that adds them up was a launch of its own, 5 us + a gap; the table of block offsets has n_buckets / 4096 entries and
// sits in the caches of every consumer); bucket n_buckets = the number of selected points
__device__ __forceinline__ uint32_t bucket_start(const uint32_t *__restrict__ pre, const uint32_t *__restrict__ blk_off, uint32_t b)
{
  return pre[b] + blk_off[b >> 12];
}

// selected points -> bucket order: slot = first slot of the bucket + the ticket the classify pass drew from the bucket's
// count.  The order inside a bucket is arbitrary: nothing downstream depends on it (neighbour counts, integer sums,
// min / max).
__global__ void __launch_bounds__(kCoThreads) k_cell_scatter(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, uint32_t n, Mat34f m,
                                                             const int16_t *__restrict__ ids, const uint32_t *__restrict__ ticket_of,
                                                             const uint32_t *__restrict__ pre, const uint32_t *__restrict__ blk_off,
                                                             uint32_t hi_mask, CellNode *__restrict__ sorted)
{
  // phase by phase over the thread's four points: ids, coordinates + tickets, bucket starts, stores
  size_t idx[kCoPts];
  int id[kCoPts];
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    idx[j] = (size_t)blockIdx.x * kCoBlock + (size_t)j * kCoThreads + threadIdx.x;
    id[j] = (idx[j] < n) ? (int)ids[idx[j]] : -1;
  }
  float cx[kCoPts], cy[kCoPts], cz[kCoPts];
  uint32_t b[kCoPts], k[kCoPts];
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    const size_t i = (id[j] >= 0) ? idx[j] : (size_t)0;
    k[j] = ticket_of[i];
    xform34(m, x[i], y[i], z[i], cx[j], cy[j], cz[j]);
    b[j] = bucket_of(cell_of(cx[j]), cell_of(cy[j]), cell_of(cz[j]), id[j] >= 0 ? id[j] : 0, hi_mask);
  }
  uint32_t st[kCoPts];
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) st[j] = (id[j] >= 0) ? bucket_start(pre, blk_off, b[j]) : 0u;
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    if (id[j] >= 0) {
      CellNode nd;
      nd.x = cx[j]; nd.y = cy[j]; nd.z = cz[j];
      nd.id = id[j];
      sorted[st[j] + k[j]] = nd;
    }
  }
}

// ---- the PCA rectangle's sums: fixed point, so that they do not depend on the order of the additions ----
// The reference adds in cloud order (pcl::compute3DCentroid and cv::PCA's mean in fp32, the covariance in fp64);
// round 3 kept that order with one dependent add per kept point on a single wavefront (153-250 us).  SURVEY A11 asks
// for 1e-4, not for that rounding, so the sums are taken EXACTLY instead: every term is rounded once to a fixed-point
// grid (2^-28 m for coordinates, 2^-26 m^2 for products of centred samples) and added as a 64-bit integer --
// associative, hence bit-reproducible whatever order the chip's workgroups arrive in, and closer to the real-number
// sums than the reference's fp32 running sum.  Everything downstream (mean to fp32, fp32-centred samples, covariance
// scaled and stored fp32, 2x2 eigenvectors, fp32 projections) follows the reference line by line.
constexpr double kFixCoord = 268435456.0;    // 2^28: |coordinate| < 2^11 m and < 2^23 points keep the sum below 2^62
constexpr double kFixProd = 67108864.0;      // 2^26: |centred sample| < 2^7 m
constexpr float kCoordClamp = 2047.0f, kCentredClamp = 127.0f;
constexpr int kAccStride = 16;               // per bbox: sum y, z, x, count, sum aa, ab, bb -- one 128-byte line each: the
                                             // workgroups' flushes are same-line atomics otherwise (k_pca_extent with four
                                             // boxes per line: 60 us of queueing at the L2)
constexpr int kExtStride = 32;               // per bbox: four extent keys, one line each
constexpr int kPcaTab = 128;                 // bboxes whose accumulators a workgroup keeps in LDS (more: global atomics)

__device__ __forceinline__ long long fix_coord(float v) { return __double2ll_rn((double)fminf(fmaxf(v, -kCoordClamp), kCoordClamp) * kFixCoord); }
__device__ __forceinline__ long long fix_prod(float a, float b)
{
  a = fminf(fmaxf(a, -kCentredClamp), kCentredClamp);
  b = fminf(fmaxf(b, -kCentredClamp), kCentredClamp);
  return __double2ll_rn(((double)a * (double)b) * kFixProd);   // the product of two floats is exact in fp64
}
// order-preserving unsigned key of a float (min / max by integer atomics)
__device__ __forceinline__ unsigned fkey(float f)
{
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k)
{
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// mean of bbox b's kept points from the integer sums (the reference: :157-158 centroid[1], cv::PCA's mean of the rows
// (z, x), each narrowed to fp32)
__device__ __forceinline__ void pca_means(const long long *__restrict__ a, float &cy, float &m0, float &m1, unsigned long long &cnt)
{
  cnt = (unsigned long long)a[3];
  const double inv = 1.0 / ((double)cnt * kFixCoord);
  cy = (float)((double)a[0] * inv);
  m0 = (float)((double)a[1] * inv);
  m1 = (float)((double)a[2] * inv);
}

// RadiusOutlierRemoval(0.4, 10) (:150-154) per bbox cloud: keep[t] = 1 iff at least min_pts + 1 points of the same
// bbox (the point itself included) lie within d2 <= r2f (FLANN L2_Simple in fp32: ((dx*dx) + dy*dy) + dz*dz),
// r2f = the largest float not above the fp64 radius^2 (PCL >= 1.11 dense path).  Points are taken IN BUCKET ORDER.
// Cells that are neighbours along x and share ix >> 3 are neighbouring BUCKETS (the low bucket bits are ix & 7), so a
// row of up to three cells is one contiguous run of the sorted array (two when it straddles a multiple of 8).
//
// Two phases per pass of 32 points and wavefront:
//   A  one LANE per point: its cell, the cell range the fp32 test can reach, and its runs -- the point's OWN cell
//      first (six candidates in ten of it are hits, a quarter in the cells beside it), then the cells beside it in
//      its row, then the (up to two) runs of each of the eight rows around -- every bound requested before any is
//      used; the non-empty runs are parked in LDS in that order;
//   B  kRadLanes lanes per point, 64 / kRadLanes points at a time: the runs are one virtual list that the lanes
//      stride together, kRadCand candidates per lane and step (all requested before any is looked at); the sum of
//      hits and of lanes that still have candidates over the point's lanes is one DPP reduction, and the walk ends as
//      soon as min_pts + 1 hits are in -- for a point inside an object after the first step.
// (Round 3: four lanes per point walked the nine rows one after the other, x-neighbour cell first, every step a
// dependent trip to the L2 behind the running count: 65-116 us.  The variants tried on the way here are in
// profiles/r04/radius_filter_ab.txt.)  The kept points' coordinates are added to their bbox's integer sums on the way
// out (LDS table per workgroup, flushed once; consecutive points of the bucket order belong to the same box, so a
// workgroup flushes a handful of entries).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
// every lane of a group of LANES consecutive lanes (8 or 16, aligned) ends with the group's sum
template <int LANES>
__device__ __forceinline__ int group_sum(int v)
{
  if constexpr (LANES == 16) {
    v += dpp_i32<0x128>(v);   // row_ror:8
    v += dpp_i32<0x124>(v);   // row_ror:4
    v += dpp_i32<0x122>(v);   // row_ror:2
    v += dpp_i32<0x121>(v);   // row_ror:1
  } else {
    v += dpp_i32<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of its half row
    v += dpp_i32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_i32<0x4E>(v);    // quad_perm [2,3,0,1]
  }
  return v;
}

#ifndef GV_RAD_LANES
#define GV_RAD_LANES 16
#endif
#ifndef GV_RAD_CAND
#define GV_RAD_CAND 4
#endif
#ifndef GV_RAD_GRID
#define GV_RAD_GRID 2048
#endif
#ifndef GV_RAD_PTS
#define GV_RAD_PTS 32
#endif
#ifndef GV_RAD_OCC
#define GV_RAD_OCC 1
#endif
constexpr int kRadPts = GV_RAD_PTS;         // points per wavefront and pass
constexpr int kRadLanes = GV_RAD_LANES;     // lanes per point in phase B
constexpr int kRadCand = GV_RAD_CAND;       // candidates per lane and step
constexpr int kRadRuns = 20;                // own cell + 2 beside it + 8 rows x 2 runs, the empty ones dropped
constexpr int kRadTab = 64;                 // boxes whose sums a workgroup keeps in LDS (more: global atomics); 25.6 KB of LDS in all:
                                            // six workgroups per CU (80 VGPRs: six wavefronts per SIMD)
constexpr int kBoffLds = 256;               // block offsets of the bucket scan kept in LDS (n_buckets < 1 M: clouds up to
                                            // 2 M points; beyond that they are read from global memory)

__global__ void __launch_bounds__(256, GV_RAD_OCC) k_radius_sorted(const CellNode *__restrict__ sorted, const uint32_t *__restrict__ pre,
                                                       const uint32_t *__restrict__ blk_off, uint32_t n_buckets, uint32_t hi_mask,
                                                       float r2f, int min_pts, uint8_t *__restrict__ keep,
                                                       long long *__restrict__ acc, int nb)
{
  __shared__ long long s_acc[kRadTab][4];
  __shared__ uint2 s_run[4][kRadRuns][kRadPts];   // [begin, end) of a point's non-empty runs in visiting order
  __shared__ float4 s_pt[4][kRadPts];
  __shared__ uint32_t s_boff[kBoffLds];
  const bool tab = nb <= kRadTab;
  const uint32_t n_off = n_buckets >> 12;
  // n_off + 1 entries: the end of the LAST bucket is start_of(n_buckets) = pre[n_buckets] (0) + the total
  const bool staged = n_off + 1u <= (uint32_t)kBoffLds;
  if (tab)
    for (int i = threadIdx.x; i < nb * 4; i += 256) (&s_acc[0][0])[i] = 0;
  if (staged)
    for (uint32_t i = threadIdx.x; i <= n_off; i += 256) s_boff[i] = blk_off[i];
  __syncthreads();
  auto start_of = [&](uint32_t b) -> uint32_t { return pre[b] + (staged ? s_boff[b >> 12] : blk_off[b >> 12]); };
  const uint32_t n_sel = blk_off[n_off];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t g = (uint32_t)lane & (uint32_t)(kRadLanes - 1);
  const float4 *nodes = reinterpret_cast<const float4 *>(sorted);
  const uint32_t n_pass = (n_sel + kRadPts - 1) / kRadPts;
  // (Passes handed out through a counter to a launch as large as the chip holds, instead of this fixed stride: 131 us
  // against 59, profiles/r04/radius_filter_ab.txt.)
  for (uint32_t pass = blockIdx.x * 4u + (uint32_t)w; pass < n_pass; pass += gridDim.x * 4u) {
    const uint32_t t0 = pass * kRadPts;
    // ---- phase A: lane p < 32 prepares point t0 + p
    if (lane < kRadPts) {
      const uint32_t t = t0 + (uint32_t)lane;
      float4 pv = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
      if (t < n_sel) pv = nodes[t];
      const int myid = __float_as_int(pv.w);
      const int ix = cell_of(pv.x), iy = cell_of(pv.y), iz = cell_of(pv.z);
      int x0, x1, y0, y1, z0, z1;
      cell_range(pv.x, ix, x0, x1);
      cell_range(pv.y, iy, y0, y1);
      cell_range(pv.z, iz, z0, z1);
      const bool fast = x0 >= ix - 1 && x1 <= ix + 1 && y0 >= iy - 1 && y1 <= iy + 1 && z0 >= iz - 1 && z1 <= iz + 1;
#ifndef GV_RAD_ABLATE
#define GV_RAD_ABLATE 0
#endif
      const bool live = fast && t < n_sel && !(GV_RAD_ABLATE & 2);   // (timing experiments: 2 = no look-ups, 1 = no walk)
      // cells x0..x1 are one run, or two when they straddle a multiple of 8
      const int xs = ((x0 >> 3) != (x1 >> 3)) ? (x1 & ~7) : x1 + 1;   // first cell of the second run (none: x1 + 1)
      const bool two = xs <= x1;
      // bucket_of for the (up to 35) cells a point looks up, without its four 32-bit multiplies per cell: the hash is an
      // xor of per-axis terms, and every cell here lies within one of the point's own in each axis -- ten products per
      // point instead of a hundred (v_mul_lo_u32 issues at half rate)
      const uint32_t hid = (uint32_t)myid * 0x27D4EB2Fu;
      const uint32_t hxv[3] = {(uint32_t)((ix - 1) >> 3) * 0x9E3779B1u, (uint32_t)(ix >> 3) * 0x9E3779B1u, (uint32_t)((ix + 1) >> 3) * 0x9E3779B1u};
      const uint32_t hyv[3] = {(uint32_t)((iy - 1) >> 3) * 0x85EBCA77u, (uint32_t)(iy >> 3) * 0x85EBCA77u, (uint32_t)((iy + 1) >> 3) * 0x85EBCA77u};
      const uint32_t hzv[3] = {(uint32_t)((iz - 1) >> 3) * 0xC2B2AE3Du, (uint32_t)(iz >> 3) * 0xC2B2AE3Du, (uint32_t)((iz + 1) >> 3) * 0xC2B2AE3Du};
      auto bkt = [&](int cx, int dy, int dz) -> uint32_t {   // = bucket_of(cx, iy + dy, iz + dz, myid, hi_mask) for |cx - ix| <= 1
        const int cy = iy + dy, cz = iz + dz;
        const uint32_t lo = ((uint32_t)cx & 7u) | (((uint32_t)cy & 7u) << 3) | (((uint32_t)cz & 7u) << 6);
        uint32_t h = (cx < ix ? hxv[0] : (cx > ix ? hxv[2] : hxv[1])) ^ hyv[dy + 1] ^ hzv[dz + 1] ^ hid;
        h ^= h >> 15;
        return lo | ((h & hi_mask) << 9);
      };
      uint2 own = make_uint2(0u, 0u), lft = make_uint2(0u, 0u), rgt = make_uint2(0u, 0u);
      if (live) {
        const uint32_t b = bkt(ix, 0, 0);
        own = make_uint2(start_of(b), start_of(b + 1u));
        if (x0 < ix) { const uint32_t bl = bkt(ix - 1, 0, 0); lft = make_uint2(start_of(bl), start_of(bl + 1u)); }
        if (x1 > ix) { const uint32_t br = bkt(ix + 1, 0, 0); rgt = make_uint2(start_of(br), start_of(br + 1u)); }
      }
      uint4 rr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int q = k + (k >= 4 ? 1 : 0);   // the eight rows around the centre
        const int dy = (q % 3) - 1, dz = (q / 3) - 1;
        const int cy = iy + dy, cz = iz + dz;
        const bool in = live && cy >= y0 && cy <= y1 && cz >= z0 && cz <= z1;
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (in) {
          r.x = start_of(bkt(x0, dy, dz));
          r.y = start_of(bkt(min(xs - 1, x1), dy, dz) + 1u);
          if (two) {
            r.z = start_of(bkt(xs, dy, dz));
            r.w = start_of(bkt(x1, dy, dz) + 1u);
          }
        }
        rr[k] = r;
      }
      int nr = 0;
      auto put_run = [&](uint32_t a, uint32_t e) {
        if (e > a) { s_run[w][nr][lane] = make_uint2(a, e); ++nr; }
      };
      put_run(own.x, own.y);
      put_run(lft.x, lft.y);
      put_run(rgt.x, rgt.y);
#pragma unroll
      for (int k = 0; k < 8; ++k) { put_run(rr[k].x, rr[k].y); put_run(rr[k].z, rr[k].w); }
      // (ids are below 2^15) id | runs << 16 | "cell by cell" << 30; negative: past the last selected point
      if (t < n_sel) pv.w = __int_as_float(myid | (nr << 16) | (fast ? 0 : 0x40000000));
      s_pt[w][lane] = pv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- phase B: kRadLanes lanes per point, 64 / kRadLanes points at a time
    constexpr int kAtOnce = 64 / kRadLanes;
    for (int sub = 0; sub < kRadPts / kAtOnce; ++sub) {
      const int pi = sub * kAtOnce + lane / kRadLanes;
      const uint32_t t = t0 + (uint32_t)pi;
      const float4 pv = s_pt[w][pi];
      const float px = pv.x, py = pv.y, pz = pv.z;
      const int idw = __float_as_int(pv.w);
      if (idw < 0) continue;   // past the last selected point (a point's lanes take this together)
      const int myid = idw & 0xffff;
      int cnt = 0;   // the same in all lanes of the point
      auto hit = [&](const float4 &c, bool on) -> int {
        float d = c.x - px;
        float r = __fmul_rn(d, d);
        d = c.y - py; r = __fadd_rn(r, __fmul_rn(d, d));
        d = c.z - pz; r = __fadd_rn(r, __fmul_rn(d, d));
        return (on && __float_as_int(c.w) == myid && r <= r2f) ? 1 : 0;
      };
      if (GV_RAD_ABLATE & 1) {
        cnt = min_pts + 1;
      } else if (!(idw & 0x40000000)) {
        // A step has kRadCand SLOTS of kRadLanes consecutive candidates each; a slot takes the next kRadLanes entries of the
        // current run and hands over to the next run when that one is exhausted (a short run leaves lanes of its slot
        // idle).  Which run and where in it is the same for all lanes of the point, so the bookkeeping is a handful of
        // selects -- the per-lane cursor over one virtual list of this kernel's first form cost 65 instructions per
        // candidate, six times the distance test.
        const int nr = (idw >> 16) & 31;
        int r = 0;
        uint32_t a = 0u, e = 0u, off = 0u;
        if (nr) { const uint2 ae = s_run[w][0][pi]; a = ae.x; e = ae.y; }
        for (;;) {
          uint32_t jj[kRadCand];
          bool on[kRadCand];
#pragma unroll
          for (int u = 0; u < kRadCand; ++u) {
            const uint32_t c = a + off + g;
            on[u] = r < nr && c < e;
            jj[u] = c;
            off += (uint32_t)kRadLanes;
            const bool adv = r < nr && a + off >= e;
            const uint2 nx = s_run[w][min(r + 1, kRadRuns - 1)][pi];
            r += adv ? 1 : 0;
            a = adv ? nx.x : a;
            e = adv ? nx.y : e;
            off = adv ? 0u : off;
          }
          float4 c4[kRadCand];
#pragma unroll
          for (int u = 0; u < kRadCand; ++u) c4[u] = nodes[on[u] ? jj[u] : t];
          int hs = 0;
#pragma unroll
          for (int u = 0; u < kRadCand; ++u) hs += hit(c4[u], on[u]);
          cnt += group_sum<kRadLanes>(hs);
          if (cnt > min_pts || r >= nr) break;
        }
      } else {   // coordinates so large that fp32 spacing widens the range (capped at +-3 cells): cell by cell
        const int ix = cell_of(px), iy = cell_of(py), iz = cell_of(pz);
        int x0, x1, y0, y1, z0, z1;
        cell_range(px, ix, x0, x1);
        cell_range(py, iy, y0, y1);
        cell_range(pz, iz, z0, z1);
        for (int cz = z0; cz <= z1 && cnt <= min_pts; ++cz)
          for (int cy = y0; cy <= y1 && cnt <= min_pts; ++cy)
            for (int cx = x0; cx <= x1 && cnt <= min_pts; ++cx) {
              const uint32_t b = bucket_of(cx, cy, cz, myid, hi_mask);
              const uint32_t e = start_of(b + 1u);
              for (uint32_t j = start_of(b); j < e && cnt <= min_pts; j += (uint32_t)kRadLanes) {
                const bool on = j + g < e;
                cnt += group_sum<kRadLanes>(hit(nodes[on ? j + g : t], on));
              }
            }
      }
      if (g == 0u) {
        const bool kept = cnt >= min_pts + 1;
        keep[t] = kept ? 1 : 0;
        if (kept && myid < nb) {
          const long long fy = fix_coord(py), fz = fix_coord(pz), fx = fix_coord(px);
          unsigned long long *a = tab ? reinterpret_cast<unsigned long long *>(&s_acc[myid][0])
                                      : reinterpret_cast<unsigned long long *>(acc + (size_t)myid * kAccStride);
          atomicAdd(a + 0, (unsigned long long)fy);
          atomicAdd(a + 1, (unsigned long long)fz);
          atomicAdd(a + 2, (unsigned long long)fx);
          atomicAdd(a + 3, 1ull);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (!tab) return;
  __syncthreads();
  for (int i = threadIdx.x; i < nb * 4; i += 256) {
    const long long v = (&s_acc[0][0])[i];
    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(acc + (size_t)(i >> 2) * kAccStride + (i & 3)), (unsigned long long)v);
  }
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
  return v;
}

// covariance sums of the fp32-centred samples (cv::PCA: rows (z, x) minus the fp32 mean, products accumulated in
// fp64 -- here: each product, exact in fp64, rounded once to 2^-26 m^2 and added as an integer)
__global__ void __launch_bounds__(256) k_pca_cov(const CellNode *__restrict__ sorted, const uint32_t *__restrict__ n_sel_p,
                                                 const uint8_t *__restrict__ keep, long long *__restrict__ acc, int nb)
{
  __shared__ long long s_acc[kPcaTab][3];
  __shared__ float s_m[kPcaTab][2];
  const bool tab = nb <= kPcaTab;
  if (tab) {
    for (int i = threadIdx.x; i < nb * 3; i += 256) (&s_acc[0][0])[i] = 0;
    for (int b = threadIdx.x; b < nb; b += 256) {
      float cy, m0 = 0.f, m1 = 0.f;
      unsigned long long cnt;
      if (acc[(size_t)b * kAccStride + 3] != 0) pca_means(acc + (size_t)b * kAccStride, cy, m0, m1, cnt);
      s_m[b][0] = m0;
      s_m[b][1] = m1;
    }
  }
  __syncthreads();
  const uint32_t n_sel = *n_sel_p;
  const float4 *nodes = reinterpret_cast<const float4 *>(sorted);
  const int lane = threadIdx.x & 63;
  // whole wavefronts per trip (the reductions below need them); the points of a wavefront nearly always belong to ONE box
  // (bucket order: same cell block, same box), and 64 lanes adding to the same three LDS words are 192 serialised
  // atomics: the wavefront adds them up first and one lane does three
  for (uint32_t t0 = blockIdx.x * 256u + (threadIdx.x & ~63u); t0 < n_sel; t0 += gridDim.x * 256u) {
    const uint32_t t = t0 + (uint32_t)lane;
    int id = -1;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < n_sel && keep[t]) {
      p = nodes[t];
      id = __float_as_int(p.w);
      if (id >= nb) id = -1;
    }
    const unsigned long long have = __ballot(id >= 0);
    if (!have) continue;
    const int id0 = __builtin_amdgcn_readlane(id, __ffsll((long long)have) - 1);
    const bool uniform = __ballot(id >= 0 && id != id0) == 0ull;
    long long aa = 0, ab = 0, bb = 0;
    if (id >= 0) {
      float m0, m1;
      if (tab) { m0 = s_m[id][0]; m1 = s_m[id][1]; }
      else {
        float cy;
        unsigned long long cnt;
        pca_means(acc + (size_t)id * kAccStride, cy, m0, m1, cnt);
      }
      const float a = p.z - m0, b = p.x - m1;
      aa = fix_prod(a, a); ab = fix_prod(a, b); bb = fix_prod(b, b);
    }
    if (uniform) {
      aa = wave_sum_i64(aa); ab = wave_sum_i64(ab); bb = wave_sum_i64(bb);
      if (lane == 0) {
        unsigned long long *d = tab ? reinterpret_cast<unsigned long long *>(&s_acc[id0][0])
                                    : reinterpret_cast<unsigned long long *>(acc + (size_t)id0 * kAccStride + 4);
        atomicAdd(d + 0, (unsigned long long)aa);
        atomicAdd(d + 1, (unsigned long long)ab);
        atomicAdd(d + 2, (unsigned long long)bb);
      }
    } else if (id >= 0) {
      unsigned long long *d = tab ? reinterpret_cast<unsigned long long *>(&s_acc[id][0])
                                  : reinterpret_cast<unsigned long long *>(acc + (size_t)id * kAccStride + 4);
      atomicAdd(d + 0, (unsigned long long)aa);
      atomicAdd(d + 1, (unsigned long long)ab);
      atomicAdd(d + 2, (unsigned long long)bb);
    }
  }
  if (!tab) return;
  __syncthreads();
  for (int i = threadIdx.x; i < nb * 3; i += 256) {
    const long long v = (&s_acc[0][0])[i];
    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(acc + (size_t)(i / 3) * kAccStride + 4 + (i % 3)), (unsigned long long)v);
  }
}

// mean, principal axes of one bbox from its integer sums: cv::PCA (:191-200) -- covariance scaled by 1 / n and stored
// fp32, eigenvectors of the symmetric 2x2 as rows, eigenvalues descending; major.x >= 0 (the sign is arbitrary
// upstream; length, width and centre do not depend on it)
struct PcaAxes {
  float cy, m0, m1, Mx, My, Nx, Ny;
  unsigned long long cnt;
};
__device__ __forceinline__ PcaAxes pca_axes(const long long *__restrict__ a)
{
  PcaAxes r{};
  pca_means(a, r.cy, r.m0, r.m1, r.cnt);
  const double sc = 1.0 / ((double)r.cnt * kFixProd);
  const double c00 = (double)(float)((double)a[4] * sc), c01 = (double)(float)((double)a[5] * sc), c11 = (double)(float)((double)a[6] * sc);
  double mjx, mjy;
  if (c01 == 0.0) {
    if (c00 >= c11) { mjx = 1; mjy = 0; } else { mjx = 0; mjy = 1; }
  } else {
    const double tr = c00 + c11, df = c00 - c11;
    const double root = sqrt(df * df + 4.0 * c01 * c01);
    const double l1 = 0.5 * (tr + root);
    mjx = c01; mjy = l1 - c00;
    if (fabs(l1 - c11) > fabs(mjy)) { mjx = l1 - c11; mjy = c01; }
    const double nn = sqrt(mjx * mjx + mjy * mjy);
    mjx /= nn; mjy /= nn;
  }
  if (mjx < 0 || (mjx == 0 && mjy < 0)) { mjx = -mjx; mjy = -mjy; }
  r.Mx = (float)mjx; r.My = (float)mjy; r.Nx = (float)(-mjy); r.Ny = (float)mjx;
  return r;
}

// extents of the projections on the principal axes (:203-216; min / max: order free), and -- by the workgroup whose
// ticket comes last -- the poses of all bboxes (:218-247), stored straight into the caller's block; the accumulators
// are left zero for the next call.  ext[b] = {~key(minL), key(maxL), ~key(minW), key(maxW)}: zero = "nothing yet".
__global__ void __launch_bounds__(256) k_pca_extent(const CellNode *__restrict__ sorted, const uint32_t *__restrict__ n_sel_p,
                                                    const uint8_t *__restrict__ keep, long long *__restrict__ acc,
                                                    unsigned *__restrict__ ext, unsigned *__restrict__ ticket, int nb,
                                                    const RansacState *__restrict__ st, int use_plane, uint32_t n_cloud,
                                                    gv_lshape_pose *__restrict__ poses, uint8_t *__restrict__ valid,
                                                    RansacState *__restrict__ st_copy, CallDone done,
                                                    gv_lshape_pose *__restrict__ poses_dev)
{
  __shared__ PcaAxes s_ax[kPcaTab];
  __shared__ unsigned s_ext[kPcaTab][4];
  __shared__ unsigned s_last;
  const bool tab = nb <= kPcaTab;
  if (tab)
    for (int b = threadIdx.x; b < nb; b += 256) {
      PcaAxes ax{};
      if (acc[(size_t)b * kAccStride + 3] != 0) ax = pca_axes(acc + (size_t)b * kAccStride);
      s_ax[b] = ax;
      s_ext[b][0] = s_ext[b][1] = s_ext[b][2] = s_ext[b][3] = 0u;
    }
  __syncthreads();
  const uint32_t n_sel = *n_sel_p;
  const float4 *nodes = reinterpret_cast<const float4 *>(sorted);
  const int lane = threadIdx.x & 63;
  for (uint32_t t0 = blockIdx.x * 256u + (threadIdx.x & ~63u); t0 < n_sel; t0 += gridDim.x * 256u) {   // (as in k_pca_cov)
    const uint32_t t = t0 + (uint32_t)lane;
    int id = -1;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < n_sel && keep[t]) {
      p = nodes[t];
      id = __float_as_int(p.w);
      if (id >= nb) id = -1;
    }
    const unsigned long long have = __ballot(id >= 0);
    if (!have) continue;
    const int id0 = __builtin_amdgcn_readlane(id, __ffsll((long long)have) - 1);
    const bool uniform = __ballot(id >= 0 && id != id0) == 0ull;
    unsigned k0 = 0u, k1 = 0u, k2 = 0u, k3 = 0u;   // zero = "nothing": the identity of the max below
    if (id >= 0) {
      const PcaAxes ax = tab ? s_ax[id] : pca_axes(acc + (size_t)id * kAccStride);
      const float dx = p.z - ax.m0, dy = p.x - ax.m1;
      const float pl = dx * ax.Mx + dy * ax.My, pw = dx * ax.Nx + dy * ax.Ny;
      const unsigned kl = fkey(pl), kw = fkey(pw);
      k0 = ~kl; k1 = kl; k2 = ~kw; k3 = kw;
    }
    if (uniform) {
      k0 = wave_max_u32(k0); k1 = wave_max_u32(k1); k2 = wave_max_u32(k2); k3 = wave_max_u32(k3);
      if (lane == 0) {
        unsigned *e = tab ? &s_ext[id0][0] : ext + (size_t)id0 * kExtStride;
        atomicMax(e + 0, k0); atomicMax(e + 1, k1); atomicMax(e + 2, k2); atomicMax(e + 3, k3);
      }
    } else if (id >= 0) {
      unsigned *e = tab ? &s_ext[id][0] : ext + (size_t)id * kExtStride;
      atomicMax(e + 0, k0); atomicMax(e + 1, k1); atomicMax(e + 2, k2); atomicMax(e + 3, k3);
    }
  }
  __syncthreads();
  if (tab)
    for (int i = threadIdx.x; i < nb * 4; i += 256) {
      const unsigned v = (&s_ext[0][0])[i];
      if (v != 0u) atomicMax(ext + (size_t)(i >> 2) * kExtStride + (i & 3), v);
    }
  // ---- the last workgroup to arrive writes the poses.  Everything it reads was written by agent-scope ATOMICS (the
  // flushes above) and is read by agent-scope atomic loads: those meet at the device's coherence point whatever L2 the
  // workgroups sit behind, so all the ticket needs is that this workgroup's atomics have been acknowledged (vmcnt) --
  // a release FENCE here makes every workgroup write its L2 back first: 20 us of this kernel with 256 workgroups
  // (profiles/r04/ticket_fence_ab.txt).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = tk == gridDim.x - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // computeBBoxPose :307-309: an empty segmented cloud (no plane found, or everything is ground) -> no poses
  const bool none = use_plane && (st->best_count == 0 || st->n_inliers == 0ull || st->n_inliers == (unsigned long long)n_cloud);
  // The poses are staged in LDS and leave as consecutive 16-byte stores of consecutive lanes: `poses` is usually the
  // host's pinned result block, and forty lanes storing ten doubles each at a stride of 80 bytes are four hundred
  // partial writes over PCIe (23 us of this kernel, measured with nothing selected at all).
  __shared__ __attribute__((aligned(16))) double s_out[kPcaTab][10];
  __shared__ uint8_t s_ok[kPcaTab];
  static_assert(sizeof(gv_lshape_pose) == 80, "ten doubles");
  for (int b0 = 0; b0 < nb; b0 += kPcaTab) {
    const int nbc = min(kPcaTab, nb - b0);
    for (int bl = threadIdx.x; bl < nbc; bl += 256) {
      const int b = b0 + bl;
      long long a[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        a[k] = __hip_atomic_load(acc + (size_t)b * kAccStride + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc[(size_t)b * kAccStride + k] = 0;
      }
      unsigned e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e[k] = __hip_atomic_load(ext + (size_t)b * kExtStride + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ext[(size_t)b * kExtStride + k] = 0u;
      }
      double o[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      bool ok = false;
      if (!none && a[3] != 0) {   // :174-175 an empty cloud has no pose
        const PcaAxes ax = pca_axes(a);
        const float minL = fkey_inv(~e[0]), maxL = fkey_inv(e[1]), minW = fkey_inv(~e[2]), maxW = fkey_inv(e[3]);
        // :227 degrees, :236 passed to setRPY as if radians (as the reference does); atan2f evaluated in fp64, rounded once.
        // `std::atan2(float, float) * 180.0f / CV_PI`: float product, widened for the division by the DOUBLE CV_PI,
        // narrowed once on the assignment to `float angle`.
        const float a32 = (float)atan2((double)ax.My, (double)ax.Mx);
        const float angle = (float)((double)(a32 * 180.0f) / 3.1415926535897932384626433832795);
        const double hp = (double)(-angle) * 0.5;   // tf2 setRPY(0, pitch, 0): (0, sin(p/2), 0, cos(p/2))
        o[0] = ax.m1;    // position.x  :230 center.y
        o[1] = ax.cy;    // position.y  :231 then :181
        o[2] = ax.m0;    // position.z  :232 center.x
        o[3] = 0.0; o[4] = sin(hp); o[5] = 0.0; o[6] = cos(hp);
        o[7] = maxL - minL;   // length :218,:243
        o[8] = maxW - minW;   // width  :219,:244
        o[9] = 0.0;           // height: never set on this path in the reference
        ok = true;
      }
#pragma unroll
      for (int k = 0; k < 10; ++k) s_out[bl][k] = o[k];
      s_ok[bl] = ok ? 1 : 0;
    }
    __syncthreads();
    const double2 *src = reinterpret_cast<const double2 *>(&s_out[0][0]);
    double2 *dst = reinterpret_cast<double2 *>(poses + b0);
    for (int i = threadIdx.x; i < nbc * 5; i += 256) dst[i] = src[i];
    for (int i = threadIdx.x; i < nbc; i += 256) valid[b0 + i] = s_ok[i];
    if (poses_dev) {
      double2 *dd = reinterpret_cast<double2 *>(poses_dev + b0);
      for (int i = threadIdx.x; i < nbc * 5; i += 256) {
        double2 v = src[i];
        if (i % 5 == 3 && !s_ok[i / 5]) v.y = -1.0;   // length (double 7 of 10) = -1: k_rects_from_poses skips the box
        dd[i] = v;
      }
    }
    __syncthreads();
  }
  if (st_copy) {   // the state rides home in the same block as the poses
    static_assert(sizeof(RansacState) % 8 == 0, "copied as 64-bit words");
    const unsigned long long *ss = reinterpret_cast<const unsigned long long *>(st);
    unsigned long long *sd = reinterpret_cast<unsigned long long *>(st_copy);
    for (int i = threadIdx.x; i < (int)(sizeof(RansacState) / 8); i += 256) sd[i] = ss[i];
  }
  if (done.flag) {
    __threadfence_system();   // every thread's result stores (they go to the host: call_done)
    __syncthreads();
    if (threadIdx.x == 0) call_done(done, 1u);
  }
}

void launch_radius_filter(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, const CamK &cam,
                          const BBoxTest &bt, int nb, bool use_plane, float thr_f, RansacState *st, int16_t *ids,
                          uint32_t *cell_cnt, uint32_t *pre, uint32_t *blk_off, unsigned *ticket, CellNode *sorted, uint8_t *keep,
                          uint32_t *ticket_of, long long *acc, uint32_t n_buckets, float r2f, int min_pts, hipStream_t s)
{
  if (!n) return;
  const uint32_t nblk = (uint32_t)(((size_t)n + kCoBlock - 1) / kCoBlock);
  const uint32_t hi_mask = n_buckets / 512u - 1u;
  // the bbox test's tables in LDS when they fit (40 boxes, 640 x 480: 10 KB)
  const size_t tab_bytes = (size_t)((nb + 3) & ~3) * sizeof(float4) + (size_t)bt.tiles_x * bt.tiles_y * bt.mask_words * sizeof(unsigned long long);
  if (tab_bytes <= 48 * 1024)
    hipLaunchKernelGGL(k_pose_classify<true>, dim3(nblk), dim3(kCoThreads), tab_bytes, s, x, y, z, n, m_cam, cam, bt, nb, use_plane ? 1 : 0,
                       thr_f, st, ids, cell_cnt, ticket_of, hi_mask);
  else
    hipLaunchKernelGGL(k_pose_classify<false>, dim3(nblk), dim3(kCoThreads), 0, s, x, y, z, n, m_cam, cam, bt, nb, use_plane ? 1 : 0,
                       thr_f, st, ids, cell_cnt, ticket_of, hi_mask);
  hipLaunchKernelGGL(k_cell_scan, dim3(n_buckets / kScanBlock), dim3(1024), 0, s, cell_cnt, n_buckets, pre, blk_off, ticket);
  hipLaunchKernelGGL(k_cell_scatter, dim3(nblk), dim3(kCoThreads), 0, s, x, y, z, n, m_cam, ids, ticket_of, pre, blk_off, hi_mask, sorted);
  // 32 selected points per wavefront and pass; their number is only known on the device: a fixed grid strides over them
  const uint32_t rblk = (uint32_t)std::max<size_t>(1, std::min<size_t>(((size_t)n + 4 * kRadPts - 1) / (4 * kRadPts), GV_RAD_GRID));
  hipLaunchKernelGGL(k_radius_sorted, dim3(rblk), dim3(256), 0, s, sorted, pre, blk_off, n_buckets, hi_mask, r2f, min_pts, keep, acc, nb);
}

void launch_pca_rect(const CellNode *sorted, const uint32_t *n_sel, uint32_t n, const uint8_t *keep, long long *acc, unsigned *ext,
                     unsigned *ticket, int nb, const RansacState *st, bool use_plane, gv_lshape_pose *poses, uint8_t *valid,
                     RansacState *st_copy, const CallDone &done, hipStream_t s, gv_lshape_pose *poses_dev)
{
  if (nb <= 0) return;
  // few workgroups: every one of them ends with a flush of its table into the boxes' accumulators
  const uint32_t blk = (uint32_t)std::max<size_t>(1, std::min<size_t>(((size_t)n + 1023) / 1024, 256));
  hipLaunchKernelGGL(k_pca_cov, dim3(blk), dim3(256), 0, s, sorted, n_sel, keep, acc, nb);
  hipLaunchKernelGGL(k_pca_extent, dim3(blk), dim3(256), 0, s, sorted, n_sel, keep, acc, ext, ticket, nb, st, use_plane ? 1 : 0, n,
                     poses, valid, st_copy, done, poses_dev);
}

size_t pca_acc_words(int nb) { return (size_t)nb * kAccStride; }
size_t pca_ext_words(int nb) { return (size_t)nb * kExtStride; }

}  // namespace gv
