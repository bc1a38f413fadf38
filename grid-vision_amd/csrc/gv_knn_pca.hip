// gv_knn_pca.hip -- exact brute-force kNN of each bbox centre over the projected cloud
// (buildKDTree + computeDepthForBoundingBoxes, src/cloud_detections.cpp:8-87).  The radius filter and the PCA
// rectangle of the same reference file live in gv_cloudops.hip.
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

// (d2, index) candidates ordered lexicographically: equal distances resolve to the lower point index (the
// oracle's stable insertion has the same rule; FLANN's own tie order is tree dependent).
// A candidate is ONE 64-bit key, (bits of d2) << 32 | index: d2 is a sum of squares, never negative, so the
// unsigned order of the keys is that lexicographic order and a wavefront arg-min is six 64-bit butterflies.
typedef unsigned long long KnnKey;
constexpr KnnKey kKnnNone = ~0ull;   // the d2 half is a NaN pattern: never a real candidate
__device__ __forceinline__ KnnKey knn_key(float d2, uint32_t idx) { return ((KnnKey)__float_as_uint(d2) << 32) | (KnnKey)idx; }
template <int CTRL>
__device__ __forceinline__ KnnKey dpp_key(KnnKey v)
{
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v & 0xffffffffull), CTRL, 0xF, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, false);
  return ((KnnKey)hi << 32) | (KnnKey)lo;
}
// minimum over the wavefront, in every lane: rotations by 8, 4, 2, 1 inside the rows of 16 (DPP row_ror: every lane
// ends with its row's minimum), then the two cross-row steps through the permute network
__device__ __forceinline__ KnnKey wave_min_key(KnnKey v)
{
  KnnKey o;
  o = dpp_key<0x128>(v); v = o < v ? o : v;   // row_ror:8
  o = dpp_key<0x124>(v); v = o < v ? o : v;   // row_ror:4
  o = dpp_key<0x122>(v); v = o < v ? o : v;   // row_ror:2
  o = dpp_key<0x121>(v); v = o < v ? o : v;   // row_ror:1
  o = __shfl_xor(v, 16); v = o < v ? o : v;
  o = __shfl_xor(v, 32); v = o < v ? o : v;
  return v;
}

// value of lane ^ J: quad permutes for J = 1, 2, ds_swizzle (bit-mask mode, xor inside 32 lanes) for 4, 8, 16, the
// permute network for 32
template <int J>
__device__ __forceinline__ KnnKey xor_lane_key(KnnKey v)
{
  if constexpr (J == 1) return dpp_key<0xB1>(v);        // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return dpp_key<0x4E>(v);   // quad_perm [2,3,0,1]
  else if constexpr (J == 32) return __shfl_xor(v, 32);
  else {
    constexpr int pat = (J << 10) | 0x1f;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(v & 0xffffffffull), pat);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(v >> 32), pat);
    return ((KnnKey)hi << 32) | (KnnKey)lo;
  }
}
// one compare-exchange step of a bitonic network over the 64 lanes: partner = lane ^ J, blocks of SIZE lanes
// alternate their direction (SIZE = 64: one block); DESC flips every direction
template <int SIZE, int J, bool DESC>
__device__ __forceinline__ KnnKey bitonic_step(KnnKey v, int lane)
{
  const KnnKey o = xor_lane_key<J>(v);
  const bool up = ((lane & SIZE) == 0) != DESC;   // (lane & 64 is always 0)
  const bool lower = (lane & J) == 0;
  const bool keep_min = lower == up;
  return ((o < v) == keep_min) ? o : v;
}
template <int SIZE, bool DESC>
__device__ __forceinline__ KnnKey bitonic_merge_steps(KnnKey v, int lane)   // the steps J = SIZE / 2 .. 1 of one stage
{
  if constexpr (SIZE >= 64) v = bitonic_step<SIZE, 32, DESC>(v, lane);
  if constexpr (SIZE >= 32) v = bitonic_step<SIZE, 16, DESC>(v, lane);
  if constexpr (SIZE >= 16) v = bitonic_step<SIZE, 8, DESC>(v, lane);
  if constexpr (SIZE >= 8) v = bitonic_step<SIZE, 4, DESC>(v, lane);
  if constexpr (SIZE >= 4) v = bitonic_step<SIZE, 2, DESC>(v, lane);
  v = bitonic_step<SIZE, 1, DESC>(v, lane);
  return v;
}
// the 64 lanes' keys sorted across the lanes (21 steps)
template <bool DESC>
__device__ __forceinline__ KnnKey bitonic_sort64(KnnKey v, int lane)
{
  v = bitonic_merge_steps<2, DESC>(v, lane);
  v = bitonic_merge_steps<4, DESC>(v, lane);
  v = bitonic_merge_steps<8, DESC>(v, lane);
  v = bitonic_merge_steps<16, DESC>(v, lane);
  v = bitonic_merge_steps<32, DESC>(v, lane);
  v = bitonic_merge_steps<64, DESC>(v, lane);
  return v;
}

constexpr int kKnnMaxK = 32;
constexpr int kKnnThreads = 256;
constexpr int kKnnWaves = kKnnThreads / 64;

// Stage 1: grid (chunks, nb), four wavefronts per workgroup, each on its own slice of the chunk and with its
// own running top-k (lane r holds the r-th best).  A lane's candidate is looked at only if it beats the
// wavefront's k-th best so far; survivors are appended -- ballot-compacted, no divergence -- to the wavefront's
// LDS buffer, and when 64 or more have gathered the buffer and the old top-k are merged by a bitonic network
// (below).  The threshold tightens with every merge (after 64, ~500, ~4000 points ...), so a slice of
// 8000 points sees three or four merges and a few hundred survivors.  (Round 2 kept a sorted list per THREAD:
// 122 points per thread make a loose threshold, some lane of the wavefront inserted on almost every iteration
// and the divergent insertion loops were the whole 0.4 ms.)  FLANN L2_Simple distance: ((du*du) + dv*dv) + dz*dz
// in fp32, query (cx, cy, 0).
constexpr int kKnnBuf = 128;
__global__ void __launch_bounds__(kKnnThreads) k_knn_stage1(const float *__restrict__ pu, const float *__restrict__ pv,
                                                            const float *__restrict__ pd, uint32_t n,
                                                            const gv_bbox *__restrict__ bboxes, int k,
                                                            KnnKey *__restrict__ partial)
{
  __shared__ KnnKey s_buf[kKnnWaves][kKnnBuf];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.y;
  const gv_bbox bb = bboxes[b];
  const float qx = (float)(bb.x_min + ((bb.x_max - bb.x_min) / 2.0f));   // :57
  const float qy = (float)(bb.y_min + ((bb.y_max - bb.y_min) / 2.0f));   // :58
  const float qz = 0.0f;                                                  // :59
  KnnKey *buf = s_buf[w];
  KnnKey mytop = kKnnNone;    // lane r: the r-th best so far (r < k)
  float thresh_f = INFINITY;  // squared distance of the k-th best so far once k candidates exist: the scan compares
                              // floats (<=; an equal distance with a higher index only adds a buffer entry the merge
                              // ranks behind the k-th), NaN never passes
  int nbuf = 0;               // wavefront-uniform
  // Merge of the buffer's first 64 entries into the top-k: the entries are sorted DESCENDING across the lanes by a
  // bitonic network (21 compare-exchange steps), lane-wise minimum with the ascending top-k (lanes >= k: "none")
  // leaves the 64 smallest of both as a bitonic sequence, six more steps sort it.  ~250 vector instructions
  // whatever k is (the k rounds of wavefront arg-min this replaces: 70 per round).  What the buffer holds beyond
  // 64 entries moves to its front and waits for the next merge.
  auto merge = [&]() {
    const int take = min(nbuf, 64);
    KnnKey a1 = (lane < take) ? buf[lane] : kKnnNone;
    if (nbuf > 64) {
      const KnnKey rest = (lane + 64 < nbuf) ? buf[lane + 64] : kKnnNone;
      buf[lane] = rest;
    }
    nbuf -= take;
    a1 = bitonic_sort64<true>(a1, lane);
    const KnnKey a0 = (lane < k) ? mytop : kKnnNone;
    KnnKey v = a0 < a1 ? a0 : a1;
    v = bitonic_merge_steps<64, false>(v, lane);
    mytop = (lane < k) ? v : kKnnNone;
    const unsigned tl = __builtin_amdgcn_readlane((unsigned)(mytop & 0xffffffffull), k - 1);
    const unsigned th = __builtin_amdgcn_readlane((unsigned)(mytop >> 32), k - 1);
    const KnnKey kth = ((KnnKey)th << 32) | (KnnKey)tl;   // still "none" while fewer than k candidates exist
    thresh_f = (kth == kKnnNone) ? INFINITY : __uint_as_float(th);
  };
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = min(n, lo + per);
  constexpr int kBatch = 4;   // points per lane requested before the first is looked at
  for (uint32_t base = lo; base < hi; base += kKnnThreads * kBatch) {   // the same trip count for every lane: the ballots need whole wavefronts
    const uint32_t i0 = base + (uint32_t)tid;
    float bu[kBatch], bv[kBatch], bd[kBatch];
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const uint32_t i = i0 + (uint32_t)q * kKnnThreads;
      const bool in = i < hi;
      bu[q] = in ? pu[i] : 0.0f;
      bv[q] = in ? pv[i] : 0.0f;
      bd[q] = in ? pd[i] : __uint_as_float(0x7fc00000u);   // past the end: NaN, never ranks
    }
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const uint32_t i = i0 + (uint32_t)q * kKnnThreads;
      float d = bu[q] - qx;
      float r = __fmul_rn(d, d);
      d = bv[q] - qy; r = __fadd_rn(r, __fmul_rn(d, d));
      d = bd[q] - qz; r = __fadd_rn(r, __fmul_rn(d, d));
      const bool ok = r <= thresh_f;
      const unsigned long long mk = __ballot(ok);
      if (mk) {
        if (ok) buf[nbuf + __popcll(mk & ((1ull << lane) - 1ull))] = knn_key(r, i);
        nbuf += (int)__popcll(mk);
        if (nbuf >= 64) merge();
      }
    }
  }
  while (nbuf) merge();
  KnnKey *out = partial + (((size_t)b * gridDim.x + blockIdx.x) * kKnnWaves + w) * k;
  if (lane < k) out[lane] = mytop;   // sorted ascending, "none" past the candidates found
}

// Stage 2: one wavefront per bbox merges the sorted lists of stage 1, writes the sorted squared distances and the
// upper-median depth (nth_element at size/2, :78-81).  The lists (at most 256 x 32 keys) are staged in LDS first --
// one coalesced sweep -- so that advancing a list's head inside the k rounds is an LDS read, not a dependent
// trip to the L2 per round (16 -> 11 us).  A lane owns every 64th list and keeps their heads in registers.
constexpr int kKnnListsPerLane = 4;
constexpr int kStageBatch = 20;   // 128 lists x k = 10 keys: one batch
__global__ void __launch_bounds__(64) k_knn_stage2(const KnnKey *__restrict__ partial, int nlists, int k,
                                                   const float *__restrict__ pd, float *__restrict__ depths,
                                                   float *__restrict__ knn_d2, CallDone done)
{
  extern __shared__ KnnKey s_lists[];   // nlists * k keys, then kKnnMaxK results
  KnnKey *s_best = s_lists + (size_t)nlists * k;
  const int lane = threadIdx.x, b = blockIdx.x;
  const KnnKey *p = partial + (size_t)b * nlists * k;
  const int total = nlists * k;
  for (int i0 = 0; i0 < total; i0 += 64 * kStageBatch) {   // batches of predicated loads, all in flight together
    KnnKey v[kStageBatch];
#pragma unroll
    for (int u = 0; u < kStageBatch; ++u) {
      const int i = i0 + u * 64 + lane;
      v[u] = (i < total) ? p[i] : kKnnNone;
    }
#pragma unroll
    for (int u = 0; u < kStageBatch; ++u) {
      const int i = i0 + u * 64 + lane;
      if (i < total) s_lists[i] = v[u];
    }
  }
  __syncthreads();
  int hp[kKnnListsPerLane];
  KnnKey cur[kKnnListsPerLane];
#pragma unroll
  for (int q = 0; q < kKnnListsPerLane; ++q) {
    const int l = lane + 64 * q;
    hp[q] = 0;
    cur[q] = (l < nlists) ? s_lists[l * k] : kKnnNone;
  }
  for (int round = 0; round < k; ++round) {
    KnnKey cand = cur[0];
#pragma unroll
    for (int q = 1; q < kKnnListsPerLane; ++q) cand = cur[q] < cand ? cur[q] : cand;
    const KnnKey best = wave_min_key(cand);
    if (best != kKnnNone) {
#pragma unroll
      for (int q = 0; q < kKnnListsPerLane; ++q)
        if (cur[q] == best) {   // the owner advances that list
          const int l = lane + 64 * q;
          ++hp[q];
          cur[q] = (hp[q] < k) ? s_lists[l * k + hp[q]] : kKnnNone;
        }
    }
    if (lane == 0) s_best[round] = best;
  }
  __syncthreads();
  // the depths of the neighbours found, requested together (lane j: neighbour j)
  const KnnKey mine = (lane < k) ? s_best[lane] : kKnnNone;
  const float dep = (mine != kKnnNone) ? pd[(uint32_t)(mine & 0xffffffffull)] : 0.0f;
  if (knn_d2 && lane < k) knn_d2[(size_t)b * k + lane] = (mine == kKnnNone) ? INFINITY : __uint_as_float((unsigned)(mine >> 32));
  const unsigned long long have = __ballot(mine != kKnnNone);   // candidates fill the rounds from the front
  const int cnt = (int)__popcll(have);
  // rank of this lane's depth among the cnt found (ties by position): the element of rank cnt / 2 is the
  // upper median -- what an insertion sort followed by [cnt / 2] yields
  int rank = 0;
  for (int j = 0; j < cnt; ++j) {
    const float o = __shfl(dep, j);
    rank += (o < dep || (o == dep && j < lane)) ? 1 : 0;
  }
  float out = -1.0f;   // :49
  const unsigned long long pick = __ballot(lane < cnt && rank == cnt / 2);
  if (cnt > 0) out = __shfl(dep, pick ? __ffsll((long long)pick) - 1 : 0);
  if (lane == 0) depths[b] = out;
  if (done.flag) __threadfence_system();   // every lane's result stores, then the ticket
  if (lane == 0) call_done(done, gridDim.x);
}

int knn_chunks() { return 64 * kKnnListsPerLane / kKnnWaves / 2; }   // 32 chunks x 4 wavefronts = 128 lists: two per lane
size_t knn_partial_entries(int nb, int k) { return (size_t)nb * knn_chunks() * kKnnWaves * k; }

void launch_knn(const float *pu, const float *pv, const float *pd, uint32_t n, const gv_bbox *bboxes, int nb, int k,
                Cand2 *partial, float *depths, float *knn_d2, const CallDone &done, hipStream_t s)
{
  if (nb <= 0) return;
  static_assert(sizeof(Cand2) == sizeof(KnnKey), "candidate layout");
  const int nchunks = knn_chunks();
  hipLaunchKernelGGL(k_knn_stage1, dim3(nchunks, nb), dim3(kKnnThreads), 0, s, pu, pv, pd, n, bboxes, k,
                     reinterpret_cast<KnnKey *>(partial));
  const size_t lds2 = ((size_t)nchunks * kKnnWaves * k + kKnnMaxK) * sizeof(KnnKey);   // 128 lists x k <= 32 keys: 32 KB
  hipLaunchKernelGGL(k_knn_stage2, dim3(nb), dim3(64), lds2, s, reinterpret_cast<const KnnKey *>(partial), nchunks * kKnnWaves, k,
                     pd, depths, knn_d2, done);
}

}  // namespace gv
