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
                                                                 const float4 *__restrict__ planes, int iters, float thr_f,
                                                                 unsigned *__restrict__ counts, int stride)
{
  extern __shared__ float4 s_pl[];   // iters planes, then iters counters
  unsigned *s_cnt = reinterpret_cast<unsigned *>(s_pl + iters);
  const int tid = threadIdx.x;
  for (int t = tid; t < iters; t += kCoThreads) { s_pl[t] = planes[t]; s_cnt[t] = 0u; }
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

// unit eigenvector of the smallest eigenvalue of a symmetric 3x3 (cyclic Jacobi, fp64): oracle/ransac.c, operation
// for operation.  Every index is a compile-time constant (loops unrolled, the final column picked by selects): with
// run-time indices the two matrices live in scratch memory and every access of this single-lane tail is a trip to
// the L2 (14 us of the kernel's 29).
template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&a)[3][3], double (&e)[3][3])
{
  if (a[P][Q] == 0.0) return;
  const double theta = (a[Q][Q] - a[P][P]) / (2.0 * a[P][Q]);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double akp = a[k][P], akq = a[k][Q];
    a[k][P] = c * akp - s * akq;
    a[k][Q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double apk = a[P][k], aqk = a[Q][k];
    a[P][k] = c * apk - s * aqk;
    a[Q][k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double ekp = e[k][P], ekq = e[k][Q];
    e[k][P] = c * ekp - s * ekq;
    e[k][Q] = s * ekp + c * ekq;
  }
}

__device__ __forceinline__ void smallest_eigenvector3_dev(const double cov[6], double v[3])
{
  double a[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
  double e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    jacobi_rotate<0, 1>(a, e);
    jacobi_rotate<0, 2>(a, e);
    jacobi_rotate<1, 2>(a, e);
  }
  // column of the smallest diagonal entry (the first of equals), by selects
  const bool m1 = a[1][1] < a[0][0];
  const double d01 = m1 ? a[1][1] : a[0][0];
  const bool m2 = a[2][2] < d01;
  double nn[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) nn[k] = m2 ? e[k][2] : (m1 ? e[k][1] : e[k][0]);
  const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
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
                                                               const float4 *__restrict__ planes, unsigned *__restrict__ counts,
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
  const float4 pl = bestc ? planes[s_best] : make_float4(0.f, 0.f, 0.f, 0.f);
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
    if (lane == 0) part_a[(size_t)blockIdx.x * kMom + w] = t;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == gridDim.x - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    }
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
    if (st_copy) {   // the workgroup that finishes last hands the final state out (its count read past the L1)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
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
  hipLaunchKernelGGL(k_ransac_hypotheses, dim3((iters + 63) / 64), dim3(64), 0, s, x, y, z, n, m_cam, seed, iters, planes);
  for (int h0 = 0; h0 < iters; h0 += 2048) {   // planes + counters of one launch sit in LDS (20 bytes per hypothesis)
    const int hn = std::min(2048, iters - h0);
    hipLaunchKernelGGL(k_ransac_count_all, dim3(nblk), dim3(kCoThreads), (size_t)hn * (sizeof(float4) + sizeof(unsigned)), s, x, y, z,
                       n, m_cam, planes + h0, hn, thr_f, counts + h0, iters);
  }
  double *part_a = scratch, *part_b = scratch + (size_t)kMom * (nblk + 1);
  hipLaunchKernelGGL(k_ransac_moments, dim3(nblk), dim3(kCoThreads), 0, s, x, y, z, n, m_cam, planes, counts, iters, thr_f, part_a,
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
// is counted into its (cell, id) bucket; unselected points are marked dropped.  Also counts the ground points
// (st->n_inliers) when the plane is in use.
__global__ void __launch_bounds__(kCoThreads) k_pose_classify(const float *__restrict__ x, const float *__restrict__ y,
                                                              const float *__restrict__ z, uint32_t n, Mat34f m, CamK cam,
                                                              BBoxTest bt, int nb, int use_plane, float thr_f,
                                                              RansacState *__restrict__ st, int16_t *__restrict__ ids,
                                                              uint8_t *__restrict__ drop, uint32_t *__restrict__ cell_cnt,
                                                              uint32_t hi_mask)
{
  __shared__ unsigned s_n;
  float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
  bool have = false;
  if (use_plane) {
    pl = st->refined;
    have = st->best_count != 0;
  }
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    const size_t i = (size_t)blockIdx.x * kCoBlock + (size_t)j * kCoThreads + threadIdx.x;
    bool ground = false;
    if (i < n) {
      float cx, cy, cz;
      xform34(m, x[i], y[i], z[i], cx, cy, cz);
      ground = have && plane_inlier_dev(pl, cx, cy, cz, thr_f);
      int id = ground ? -1 : first_bbox(cam, bt, cx, cy, cz);
      if (id >= nb) id = -1;
      ids[i] = (int16_t)id;
      if (id >= 0) atomicAdd(&cell_cnt[bucket_of(cell_of(cx), cell_of(cy), cell_of(cz), id, hi_mask)], 1u);
      else drop[i] = 1;
    }
    c += (unsigned)__popcll(__ballot(ground));
  }
  if (use_plane) {
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&st->n_inliers, (unsigned long long)s_n);
  }
}

// Exclusive prefix of the bucket counts, one launch: every workgroup scans its 4096 buckets (pre[b] = points
// of its earlier buckets) and publishes its total; the workgroup whose ticket comes last turns the totals into
// block offsets (k_cell_starts then adds them: pre[b] = first slot of bucket b, pre[n_buckets] = the number of
// selected points).
constexpr int kScanBlock = 4096;
__global__ void __launch_bounds__(1024) k_cell_scan(const uint32_t *__restrict__ cell_cnt, uint32_t n_buckets,
                                                    uint32_t *__restrict__ pre, uint32_t *__restrict__ blk_off,
                                                    unsigned *__restrict__ ticket)
{
  __shared__ unsigned s_w[16];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t b0 = (size_t)blockIdx.x * kScanBlock + (size_t)tid * 4;
  const uint4 c = *reinterpret_cast<const uint4 *>(cell_cnt + b0);   // n_buckets is a multiple of 4096
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
    __hip_atomic_store(&blk_off[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = tk == gridDim.x - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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

// pre[b] += blk_off[b >> 12]: first slot of every bucket; entry n_buckets = the number of selected points
__global__ void __launch_bounds__(1024) k_cell_starts(uint32_t *__restrict__ pre, const uint32_t *__restrict__ blk_off, uint32_t n_buckets)
{
  const size_t b0 = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * 4;
  if (b0 < n_buckets) {
    const unsigned o = blk_off[blockIdx.x];
    uint4 v = *reinterpret_cast<uint4 *>(pre + b0);
    v.x += o; v.y += o; v.z += o; v.w += o;
    *reinterpret_cast<uint4 *>(pre + b0) = v;
  } else if (b0 == n_buckets) {
    pre[n_buckets] = blk_off[blockIdx.x];   // the extra workgroup: the total
  }
}

// selected points -> bucket order: slot = first slot of the bucket + a ticket out of the bucket's count, which
// is counted back down to zero (ready for the next call).  The order inside a bucket is arbitrary: the
// neighbour counts do not depend on it.
__global__ void __launch_bounds__(kCoThreads) k_cell_scatter(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, uint32_t n, Mat34f m,
                                                             const int16_t *__restrict__ ids, uint32_t *__restrict__ cell_cnt,
                                                             const uint32_t *__restrict__ start, uint32_t hi_mask,
                                                             CellNode *__restrict__ sorted)
{
#pragma unroll
  for (int j = 0; j < kCoPts; ++j) {
    const size_t i = (size_t)blockIdx.x * kCoBlock + (size_t)j * kCoThreads + threadIdx.x;
    if (i >= n) continue;
    const int id = ids[i];
    if (id < 0) continue;
    float cx, cy, cz;
    xform34(m, x[i], y[i], z[i], cx, cy, cz);
    const uint32_t b = bucket_of(cell_of(cx), cell_of(cy), cell_of(cz), id, hi_mask);
    const uint32_t k = atomicSub(&cell_cnt[b], 1u) - 1u;
    CellNode nd;
    nd.x = cx; nd.y = cy; nd.z = cz;
    nd.id = id;
    nd.orig = (int32_t)i;
    nd.pad[0] = nd.pad[1] = nd.pad[2] = 0;
    sorted[start[b] + k] = nd;
  }
}

// RadiusOutlierRemoval(0.4, 10) (:150-154) per bbox cloud: drop[i] = 0 iff at least min_pts + 1 points of the
// same bbox (the point itself included) lie within d2 <= r2f (FLANN L2_Simple in fp32: ((dx*dx) + dy*dy) + dz*dz),
// r2f = the largest float not above the fp64 radius^2 (PCL >= 1.11 dense path).  Points are taken IN BUCKET
// ORDER: neighbouring lanes sit in the same cell, probe the same buckets and read the same candidates, so the
// wavefront's loads collapse to a few lines.  Cells that are neighbours along x and share ix >> 3 are
// neighbouring BUCKETS (the low bucket bits are ix & 7), so a row of up to three cells is one contiguous run of
// the sorted array: nine rows, at most eighteen runs, whose bounds are requested together before any candidate
// is read.  The count stops at min_pts + 1 -- the outcome only asks whether it is reached.
__device__ __forceinline__ int quad_sum(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  return v;
}

template <int CTRL>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

__global__ void __launch_bounds__(256) k_radius_sorted(const CellNode *__restrict__ sorted, const uint32_t *__restrict__ start,
                                                       uint32_t n_buckets, uint32_t hi_mask, float r2f, int min_pts,
                                                       uint8_t *__restrict__ drop)
{
  // FOUR lanes per point: the selected points are few (1e5) and every query is a chain of dependent loads, so
  // one lane per point leaves the chip two wavefronts per SIMD and all of them waiting; the four lanes of a quad
  // take every fourth candidate of a run (one 128-byte read per quad and step) and share one count (DPP).
  const uint32_t n_sel = start[n_buckets];
  const uint32_t l = threadIdx.x & 3u;
  for (uint32_t t = (blockIdx.x * 256u + threadIdx.x) >> 2; t < n_sel; t += gridDim.x * 64u) {
    const CellNode p = sorted[t];
    const int ix = cell_of(p.x), iy = cell_of(p.y), iz = cell_of(p.z);
    const int myid = p.id;
    int x0, x1, y0, y1, z0, z1;
    cell_range(p.x, ix, x0, x1);
    cell_range(p.y, iy, y0, y1);
    cell_range(p.z, iz, z0, z1);
    int cnt = 0;   // the same in the four lanes
    auto walk = [&](uint32_t j, uint32_t e) {
      while (j < e && cnt <= min_pts) {
        CellNode c[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) c[q] = sorted[min(j + (uint32_t)q * 4u + l, e - 1u)];
        int hits = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float d = c[q].x - p.x;
          float r = __fmul_rn(d, d);
          d = c[q].y - p.y; r = __fadd_rn(r, __fmul_rn(d, d));
          d = c[q].z - p.z; r = __fadd_rn(r, __fmul_rn(d, d));
          hits += (j + (uint32_t)q * 4u + l < e && c[q].id == myid && r <= r2f) ? 1 : 0;
        }
        cnt += quad_sum(hits);
        j += 8u;
      }
    };
    if (x0 >= ix - 1 && x1 <= ix + 1 && y0 >= iy - 1 && y1 <= iy + 1 && z0 >= iz - 1 && z1 <= iz + 1) {
      // row (dy, dz): cells x0..x1 are one run, or two when they straddle a multiple of 8
      const int xs = ((x0 >> 3) != (x1 >> 3)) ? (x1 & ~7) : x1 + 1;   // first cell of the second run (none: x1 + 1)
      // The nine rows are dealt to the quad's four lanes (lane l: rows l, l + 4, l + 8): each lane hashes and
      // requests the bounds of at most three rows instead of all nine, and the walk gets a row's bounds from its
      // owner by a DPP quad broadcast.
      uint32_t sa[3][2], sb[3][2];
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        const int q = (int)l + 4 * sl;
        const int t9 = (q == 0) ? 4 : (q <= 4 ? q - 1 : q);   // the centre row first: it holds the point's own cell
        const int t3 = (t9 * 11) >> 5;                         // t9 / 3 for 0 .. 11
        const int cy = iy + (t9 - 3 * t3) - 1, cz = iz + t3 - 1;
        const bool in = q < 9 && cy >= y0 && cy <= y1 && cz >= z0 && cz <= z1;
        const uint32_t b_lo = bucket_of(x0, cy, cz, myid, hi_mask);
        const uint32_t b_hi = bucket_of(min(xs - 1, x1), cy, cz, myid, hi_mask);
        sa[sl][0] = in ? start[b_lo] : 0u;
        sb[sl][0] = in ? start[b_hi + 1u] : 0u;
        const bool two = in && xs <= x1;
        const uint32_t c_lo = bucket_of(xs, cy, cz, myid, hi_mask);
        const uint32_t c_hi = bucket_of(x1, cy, cz, myid, hi_mask);
        sa[sl][1] = two ? start[c_lo] : 0u;
        sb[sl][1] = two ? start[c_hi + 1u] : 0u;
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        constexpr int kBc[4] = {0x00, 0x55, 0xAA, 0xFF};   // quad_perm [k, k, k, k]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          uint32_t a, b;
          switch (q & 3) {
          case 0: a = quad_bcast<kBc[0]>(sa[q >> 2][u]); b = quad_bcast<kBc[0]>(sb[q >> 2][u]); break;
          case 1: a = quad_bcast<kBc[1]>(sa[q >> 2][u]); b = quad_bcast<kBc[1]>(sb[q >> 2][u]); break;
          case 2: a = quad_bcast<kBc[2]>(sa[q >> 2][u]); b = quad_bcast<kBc[2]>(sb[q >> 2][u]); break;
          default: a = quad_bcast<kBc[3]>(sa[q >> 2][u]); b = quad_bcast<kBc[3]>(sb[q >> 2][u]); break;
          }
          walk(a, b);
        }
      }
    } else {   // coordinates so large that fp32 spacing widens the range (capped at +-3 cells): cell by cell
      for (int cz = z0; cz <= z1 && cnt <= min_pts; ++cz)
        for (int cy = y0; cy <= y1 && cnt <= min_pts; ++cy)
          for (int cx = x0; cx <= x1 && cnt <= min_pts; ++cx) {
            const uint32_t b = bucket_of(cx, cy, cz, myid, hi_mask);
            walk(start[b], start[b + 1u]);
          }
    }
    if (l == 0) drop[p.orig] = (cnt >= min_pts + 1) ? 0 : 1;
  }
}

void launch_radius_filter(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, const CamK &cam,
                          const BBoxTest &bt, int nb, bool use_plane, float thr_f, RansacState *st, int16_t *ids, uint8_t *drop,
                          uint32_t *cell_cnt, uint32_t *pre, uint32_t *blk_off, unsigned *ticket, CellNode *sorted,
                          uint32_t n_buckets, float r2f, int min_pts, hipStream_t s)
{
  if (!n) return;
  const uint32_t nblk = (uint32_t)(((size_t)n + kCoBlock - 1) / kCoBlock);
  const uint32_t hi_mask = n_buckets / 512u - 1u;
  hipLaunchKernelGGL(k_pose_classify, dim3(nblk), dim3(kCoThreads), 0, s, x, y, z, n, m_cam, cam, bt, nb, use_plane ? 1 : 0, thr_f, st,
                     ids, drop, cell_cnt, hi_mask);
  hipLaunchKernelGGL(k_cell_scan, dim3(n_buckets / kScanBlock), dim3(1024), 0, s, cell_cnt, n_buckets, pre, blk_off, ticket);
  hipLaunchKernelGGL(k_cell_starts, dim3(n_buckets / kScanBlock + 1), dim3(1024), 0, s, pre, blk_off, n_buckets);
  hipLaunchKernelGGL(k_cell_scatter, dim3(nblk), dim3(kCoThreads), 0, s, x, y, z, n, m_cam, ids, cell_cnt, pre, hi_mask, sorted);
  // four lanes per selected point; their number is only known on the device: a fixed grid strides over them
  const uint32_t rblk = (uint32_t)std::min<size_t>(((size_t)n * 4 + 255) / 256, 8192);
  hipLaunchKernelGGL(k_radius_sorted, dim3(rblk), dim3(256), 0, s, sorted, pre, n_buckets, hi_mask, r2f, min_pts, drop);
}

// --------------------------------------------------- kept points by bbox + PCA --
// Stable split of the KEPT points by bbox id (the reference appends in cloud order, :286, and the filter keeps
// that order): blocks of 1024 points are counted, a column scan gives every (block, bbox) its first slot, and
// one wavefront per block places its points in order, writing their (recomputed) camera coordinates into the
// per-bbox arrays.  drop[i] != 0 removes the point.
constexpr int kSegBlock = 1024;

__global__ void __launch_bounds__(256) k_seg_count(const int16_t *__restrict__ ids, const uint8_t *__restrict__ drop,
                                                   uint32_t n, int nb, uint32_t *__restrict__ block_counts)
{
  extern __shared__ unsigned s_h[];
  for (int b = threadIdx.x; b < nb; b += 256) s_h[b] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSegBlock;
  for (int k = threadIdx.x; k < kSegBlock; k += 256) {
    const size_t i = base + k;
    if (i < n) {
      const int id = drop[i] ? -1 : (int)ids[i];
      if (id >= 0 && id < nb) atomicAdd(&s_h[id], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += 256) block_counts[(size_t)blockIdx.x * nb + b] = s_h[b];
}

// Exclusive prefix of every bbox's counts over the blocks (in place) and seg_start[0..nb].  One workgroup:
// 64 columns at a time (lane = bbox), the rows dealt to the 16 wavefronts in contiguous runs -- two short
// walks with independent loads instead of one dependent walk over all the blocks.
__global__ void __launch_bounds__(1024) k_seg_scan(uint32_t *__restrict__ block_counts, int nblocks, int nb,
                                                   int32_t *__restrict__ seg_start)
{
  __shared__ unsigned s_part[16][64];
  __shared__ unsigned s_tot[64];
  __shared__ unsigned s_base;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = (nblocks + 15) / 16;
  const int r0 = min(nblocks, w * per), r1 = min(nblocks, r0 + per);
  if (threadIdx.x == 0) s_base = 0u;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 64) {
    const int b = b0 + lane;
    unsigned run = 0;
    if (b < nb) {
#pragma unroll 8
      for (int k = r0; k < r1; ++k) run += block_counts[(size_t)k * nb + b];
    }
    s_part[w][lane] = run;
    __syncthreads();
    unsigned before = 0, total = 0;
    for (int q = 0; q < 16; ++q) {
      const unsigned v = s_part[q][lane];
      if (q < w) before += v;
      total += v;
    }
    if (b < nb) {
      unsigned acc = before;
#pragma unroll 8
      for (int k = r0; k < r1; ++k) {
        const unsigned c = block_counts[(size_t)k * nb + b];
        block_counts[(size_t)k * nb + b] = acc;
        acc += c;
      }
    }
    if (w == 0) {   // segment starts of these 64 bboxes: exclusive scan of their totals across the lanes
      unsigned t = (b < nb) ? total : 0u, inc = t;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
      }
      const unsigned base = s_base;
      if (b < nb) seg_start[b] = (int32_t)(base + inc - t);
      if (lane == 63) s_tot[0] = base + inc;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_tot[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) seg_start[nb] = (int32_t)s_base;
}

__global__ void __launch_bounds__(64) k_seg_scatter(const int16_t *__restrict__ ids, const uint8_t *__restrict__ drop,
                                                    const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, Mat34f m, uint32_t n, int nb,
                                                    const uint32_t *__restrict__ block_off, const int32_t *__restrict__ seg_start,
                                                    float *__restrict__ gx, float *__restrict__ gy, float *__restrict__ gz)
{
  extern __shared__ unsigned s_cur[];   // next slot of every bbox for this block (absolute position in the output)
  const int lane = threadIdx.x;
  for (int b = lane; b < nb; b += 64) s_cur[b] = (unsigned)seg_start[b] + block_off[(size_t)blockIdx.x * nb + b];
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSegBlock;
  int idb[kSegBlock / 64];   // this lane's ids of all 16 batches, requested together
#pragma unroll
  for (int q = 0; q < kSegBlock / 64; ++q) {
    const size_t i = base + (size_t)q * 64 + lane;
    int id = -1;
    if (i < n) {
      id = drop[i] ? -1 : (int)ids[i];
      if (id >= nb) id = -1;
    }
    idb[q] = id;
  }
  // One wavefront walks the block's 16 batches in order; per batch one round per distinct bbox id.  With at most
  // 64 boxes the cursors live in a REGISTER (lane b holds the next slot of bbox b): a round reads the leader's
  // cursor with a lane read and advances it in place -- no LDS round trip per round (measured the same as the LDS
  // form, 20 us on the lidar-like cloud, where a batch of 64 ring neighbours holds many ids).  More boxes: the cursors stay in LDS and are
  // advanced by the round's leader lane with a returning LDS add (LDS operations of one wavefront complete in
  // order).  Either way no barrier, so the stores of one round are still in flight while the next one runs.
  if (nb <= 64) {
    unsigned cur_reg = (lane < nb) ? s_cur[lane] : 0u;
#pragma unroll
    for (int q = 0; q < kSegBlock / 64; ++q) {
      const size_t i = base + (size_t)q * 64 + lane;
      const int id = idb[q];
      float cx = 0.f, cy = 0.f, cz = 0.f;
      if (id >= 0) xform34(m, x[i], y[i], z[i], cx, cy, cz);
      unsigned long long todo = __ballot(id >= 0);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int idl = __builtin_amdgcn_readlane(id, leader);
        const unsigned long long same = __ballot(id == idl);
        const unsigned cur = (unsigned)__builtin_amdgcn_readlane((int)cur_reg, idl);
        if (lane == idl) cur_reg += (unsigned)__popcll(same);
        if (id == idl) {
          const unsigned pos = cur + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
          gx[pos] = cx; gy[pos] = cy; gz[pos] = cz;
        }
        todo &= ~same;
      }
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < kSegBlock / 64; ++q) {
    const size_t i = base + (size_t)q * 64 + lane;
    const int id = idb[q];
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (id >= 0) xform34(m, x[i], y[i], z[i], cx, cy, cz);
    unsigned long long todo = __ballot(id >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int idl = __builtin_amdgcn_readlane(id, leader);
      const unsigned long long same = __ballot(id == idl);
      unsigned cur = 0;
      if (lane == leader) cur = atomicAdd(&s_cur[idl], (unsigned)__popcll(same));
      cur = (unsigned)__builtin_amdgcn_readlane((int)cur, leader);
      if (id == idl) {
        const unsigned pos = cur + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        gx[pos] = cx; gy[pos] = cy; gz[pos] = cz;
      }
      todo &= ~same;
    }
  }
}

// bboxPoseEstimation :156-181 + computePCABoundingBox :187-247 for one bbox cloud (the kept points, contiguous
// and in cloud order), one wavefront each.  The reference accumulates in cloud order -- pcl::compute3DCentroid
// and cv::PCA's mean in fp32, the covariance in fp64 -- so those sums are order dependent and are taken as
// sequential chains: lane 0 sums y (the centroid's y), lane 1 z and lane 2 x (the PCA rows (z, x), :167-172),
// each walking ITS array; in the covariance pass the three lanes hold c00, c01, c11.  The arrays are staged
// through LDS a tile ahead by the whole wavefront (coalesced loads; the chains read 16 bytes per LDS access),
// so a chain step costs one dependent add.  The projections' min / max are order free and run 64 wide.
constexpr int kPcaTile = 1024;   // points per LDS tile and array

__device__ __forceinline__ float lane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_f64(double v, int l)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

__global__ void __launch_bounds__(64) k_pca_bbox(const float *__restrict__ gx, const float *__restrict__ gy,
                                                 const float *__restrict__ gz, const int32_t *__restrict__ seg_start, int nb,
                                                 const RansacState *__restrict__ st, int use_plane, uint32_t n_cloud,
                                                 gv_lshape_pose *__restrict__ poses, uint8_t *__restrict__ valid,
                                                 RansacState *__restrict__ st_copy, CallDone done,
                                                 gv_lshape_pose *__restrict__ poses_dev)
{
  __shared__ __attribute__((aligned(16))) float s_t[2][3][kPcaTile];    // [buffer][y | z | x][point]
  __shared__ __attribute__((aligned(16))) double s_p[2][3][kPcaTile];   // [buffer][aa | ab | bb][point]
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= nb) return;
  if (b == 0 && lane == 0 && st_copy) *st_copy = *st;   // the state rides home in the same block as the poses
  int s0 = seg_start[b], s1 = seg_start[b + 1];
  // computeBBoxPose :307-309: an empty segmented cloud (no plane found, or everything is ground) -> no poses
  if (use_plane && (st->best_count == 0 || st->n_inliers == 0ull || st->n_inliers == (unsigned long long)n_cloud)) s1 = s0;
  const int cnt = s1 - s0;
  if (cnt <= 0) {   // :174-175 empty cloud: no pose
    if (lane == 0) {
      valid[b] = 0;
      poses[b] = gv_lshape_pose{};
      if (poses_dev) {
        gv_lshape_pose none{};
        none.length = -1.0;   // k_rects_from_poses skips it
        poses_dev[b] = none;
      }
      call_done(done, gridDim.x);
    }
    return;
  }
  const float *arr[3] = {gy + s0, gz + s0, gx + s0};
  const int ntiles = (cnt + kPcaTile - 1) / kPcaTile;
  constexpr int kPer = kPcaTile / 64;   // tile elements per lane and array
  float r[3][kPer];
  auto fetch = [&](int t) {   // tile t of the three arrays -> registers (zero padded), coalesced
    const int o = t * kPcaTile;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int q = 0; q < kPer; ++q) r[a][q] = (o + q * 64 + lane < cnt) ? arr[a][o + q * 64 + lane] : 0.0f;
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int q = 0; q < kPer; ++q) s_t[buf][a][q * 64 + lane] = r[a][q];
  };
  const int ch = lane < 3 ? lane : 0;   // the chain this lane runs (lanes >= 3 shadow lane 0)
  // pass 1: fp32 running sums in order.  The next tile's loads are in flight while this one is summed.
  float sum = 0.0f;
  fetch(0);
  put(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) fetch(t + 1);
    const float4 *src = reinterpret_cast<const float4 *>(&s_t[t & 1][ch][0]);
    const int m4 = (min(kPcaTile, cnt - t * kPcaTile) + 3) >> 2;   // the zero padding adds +0.0f: exact
#pragma unroll 8
    for (int q = 0; q < m4; ++q) {
      const float4 v = src[q];
      sum = sum + v.x; sum = sum + v.y; sum = sum + v.z; sum = sum + v.w;
    }
    if (t + 1 < ntiles) put((t + 1) & 1);
    __syncthreads();
  }
  float cy = lane_f32(sum, 0), m0 = lane_f32(sum, 1), m1 = lane_f32(sum, 2);
  cy = cy / (float)cnt;
  const float inv = (float)(1.0 / (double)cnt);
  m0 = m0 * inv;
  m1 = m1 * inv;
  // pass 2: fp64 covariance sums of the fp32-centred samples, in order: lane 0 sums a*a, lane 1 a*b, lane 2 b*b
  // (a = z - m0, b = x - m1).  The products (exact in fp64) are formed 64 wide and parked in LDS; a chain step
  // is one dependent fp64 add.  Padding contributes a product of +0.0.
  double cs = 0.0;
  auto put_products = [&](int t, int buf) {   // from the registers fetch(t) filled
    const int o = t * kPcaTile;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int e = q * 64 + lane;
      const bool in = o + e < cnt;
      const float a = r[1][q] - m0, bb = r[2][q] - m1;
      s_p[buf][0][e] = in ? (double)a * (double)a : 0.0;
      s_p[buf][1][e] = in ? (double)a * (double)bb : 0.0;
      s_p[buf][2][e] = in ? (double)bb * (double)bb : 0.0;
    }
  };
  fetch(0);
  put_products(0, 0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) fetch(t + 1);
    const double2 *src = reinterpret_cast<const double2 *>(&s_p[t & 1][ch][0]);
    const int m2 = (min(kPcaTile, cnt - t * kPcaTile) + 1) >> 1;
#pragma unroll 8
    for (int q = 0; q < m2; ++q) {
      const double2 v = src[q];
      cs = cs + v.x; cs = cs + v.y;
    }
    if (t + 1 < ntiles) put_products(t + 1, (t + 1) & 1);
    __syncthreads();
  }
  const double c00 = lane_f64(cs, 0), c01 = lane_f64(cs, 1), c11 = lane_f64(cs, 2);
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
    // :227 degrees, :236 passed to setRPY as if radians (as the reference does); atan2f evaluated in fp64, rounded once.
    // `std::atan2(float, float) * 180.0f / CV_PI`: float product, widened for the division by the DOUBLE CV_PI,
    // narrowed once on the assignment to `float angle`.
    const float a32 = (float)atan2((double)My, (double)Mx);
    const float angle = (float)((double)(a32 * 180.0f) / 3.1415926535897932384626433832795);
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
    if (poses_dev) poses_dev[b] = p;
    valid[b] = 1;
    call_done(done, gridDim.x);
  }
}

void launch_split_kept(const int16_t *ids, const uint8_t *drop, const float *x, const float *y, const float *z, const Mat34f &m_cam,
                       uint32_t n, int nb, uint32_t *block_counts,
                       int32_t *seg_start, float *gx, float *gy, float *gz, hipStream_t s)
{
  const int nblocks = (int)(((size_t)n + kSegBlock - 1) / kSegBlock);
  if (nblocks == 0 || nb <= 0) return;
  hipLaunchKernelGGL(k_seg_count, dim3(nblocks), dim3(256), (size_t)nb * sizeof(unsigned), s, ids, drop, n, nb, block_counts);
  hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, s, block_counts, nblocks, nb, seg_start);
  hipLaunchKernelGGL(k_seg_scatter, dim3(nblocks), dim3(64), (size_t)nb * sizeof(unsigned), s, ids, drop, x, y, z, m_cam, n, nb, block_counts,
                     seg_start, gx, gy, gz);
}

void launch_pca_bbox(const float *gx, const float *gy, const float *gz, const int32_t *seg_start, int nb, const RansacState *st,
                     bool use_plane, uint32_t n_cloud, gv_lshape_pose *poses, uint8_t *valid, RansacState *st_copy,
                     const CallDone &done, hipStream_t s, gv_lshape_pose *poses_dev)
{
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_pca_bbox, dim3(nb), dim3(64), 0, s, gx, gy, gz, seg_start, nb, st, use_plane ? 1 : 0, n_cloud, poses, valid,
                     st_copy, done, poses_dev);
}

}  // namespace gv
