// gv_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the grid-vision hot path.
//
// Built with -ffp-contract=off: the reference arithmetic keeps separate
// multiply/add roundings (PCL's SSE transform, grid_map's getIndex), and cell
// indices must be bit-exact, so no FMA contraction anywhere in this file.
// fp64 and fp32 divisions are the compiler's correctly rounded sequences.
//
// Reference lines are cited as file:line relative to the reference root.
#include "gv_kernels.hpp"
#include "gv_device.hpp"

#include <algorithm>
#include <cstdlib>

namespace gv {

// ------------------------------------------------------------ points pass --
// Generic points pass (any grid shape; also the bbox-only calls of the reference surface).
// The tile path of the production frame bins through gv_binning.hip instead.
// One pass over the resident SoA cloud (12 B/point read):
//   X1  base<-lidar transform, getIndex, hits[cell] += 1
//   X2  ray end of out-of-map points (clip_end[cell] = 1)
//   A1+A5 camera<-lidar transform, pinhole projection, first-match bbox id
//       (src/cloud_detections.cpp:250-298)
// Hit counting: points of one workgroup that fall into the same cell (the cells next to the
// sensor collect hundreds of hits per frame) are combined in an LDS direct-mapped cache
// (cell -> count) and flushed with ONE global atomicAdd per cached cell; a cell that loses its
// slot to another cell goes straight to the global atomic.  Integer adds: the grid is the same
// whatever the interleaving.
constexpr int kHitSlots = 4096;
constexpr unsigned kHitEmpty = 0xFFFFFFFFu;

template <bool BIN, bool RAY, bool BBOX, bool KEEPCELL>
__global__ void __launch_bounds__(BIN ? 1024 : 256) k_points(PointsArgs a)
{
  __shared__ unsigned s_key[BIN ? kHitSlots : 1];
  __shared__ unsigned s_cnt[BIN ? kHitSlots : 1];
  if (BIN) {
    for (int k = threadIdx.x; k < kHitSlots; k += blockDim.x) { s_key[k] = kHitEmpty; s_cnt[k] = 0; }
    __syncthreads();
  }
  // Out-of-map points (a minority) need the fp64 slab clip, ~10x the work of an in-map point: they
  // are compacted through LDS so that the clip runs on dense lanes of as few wavefronts as
  // possible instead of on a few lanes of every wavefront.
  constexpr bool CLIP = BIN && RAY;
  constexpr int kMaxThreads = BIN ? 1024 : 256;
  __shared__ float2 s_out[CLIP ? kMaxThreads : 1];
  __shared__ unsigned s_nout;
  const BBoxTest bt = a.bt;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t base = blockIdx.x * blockDim.x; base < a.n; base += stride) {   // uniform per workgroup
    const uint32_t i = base + threadIdx.x;
    const bool live = i < a.n;
    if (CLIP) {
      if (threadIdx.x == 0) s_nout = 0;
      __syncthreads();
    }
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) { px = a.x[i]; py = a.y[i]; pz = a.z[i]; }
    if (BIN) {
      float bx, by, bz;
      xform34(a.m_base, px, py, pz, bx, by, bz);
      int cell = -1;
      bool outside = false;
      if (live && isfinite(bx) && isfinite(by) && isfinite(bz)) {
        int ix, iy;
        if (get_index_fast(a.g, (double)bx, (double)by, ix, iy)) {
          cell = iy * a.g.nx + ix;
          const unsigned slot = ((unsigned)cell * 2654435761u) >> 20;   // 12 bits
          const unsigned old = atomicCAS(&s_key[slot], kHitEmpty, (unsigned)cell);
          if (old == kHitEmpty || old == (unsigned)cell) atomicAdd(&s_cnt[slot], 1u);
          else atomicAdd(&a.hits[cell], 1);   // no-return global_atomic_add
        } else if (RAY && a.org.valid) {
          outside = true;
        }
      }
      if (CLIP) {
        const unsigned long long bm = __ballot(outside);
        if (bm) {
          const int lane = threadIdx.x & 63;
          unsigned wbase = 0;
          if (lane == 0) wbase = atomicAdd(&s_nout, (unsigned)__popcll(bm));
          wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)wbase);
          if (outside) s_out[wbase + (unsigned)__popcll(bm & ((1ull << lane) - 1ull))] = make_float2(bx, by);
        }
      }
      if (KEEPCELL && live) a.cell_idx[i] = cell;
    }
    if (BBOX) {
      float cx, cy, cz;
      xform34(a.m_cam, px, py, pz, cx, cy, cz);
      if (live) a.bbox_id[i] = (int16_t)first_bbox(a.cam, bt, cx, cy, cz);
    }
    if (CLIP) {
      __syncthreads();
      const unsigned nout = s_nout;
      for (unsigned k = threadIdx.x; k < nout; k += blockDim.x) {
        const float2 p = s_out[k];
        int ex, ey;
        clip_ray_end(a.g, a.org, (double)p.x, (double)p.y, ex, ey);
        a.clip_end[ey * a.g.nx + ex] = 1;   // idempotent byte store
      }
      __syncthreads();   // the list is reused by the next chunk
    }
  }
  if (BIN) {
    __syncthreads();
    for (int k = threadIdx.x; k < kHitSlots; k += blockDim.x) {
      const unsigned key = s_key[k];
      if (key != kHitEmpty) atomicAdd(&a.hits[key], (int)s_cnt[k]);
    }
  }
}

void launch_points(const PointsArgs &a, hipStream_t s)
{
  if (a.n == 0) return;
  // counting: 8192 points per 1024-thread workgroup (measured best of 2k..16k) so that the LDS hit cache
  // sees the duplicates of the hot cells; bbox-only: plain streaming configuration
  const uint32_t threads = a.do_bin ? 1024u : 256u;
  const uint32_t blocks = a.do_bin ? (uint32_t)std::min<uint64_t>(((uint64_t)a.n + 8191) / 8192, (uint64_t)2048)
                                   : (uint32_t)std::min<uint64_t>(((uint64_t)a.n + 255) / 256, (uint64_t)1 << 20);
  const bool keep = a.cell_idx != nullptr;
#define GV_LP(B, R, X, K) hipLaunchKernelGGL((k_points<B, R, X, K>), dim3(blocks), dim3(threads), 0, s, a)
  if (a.do_bin && a.do_ray && a.do_bbox && keep) GV_LP(true, true, true, true);
  else if (a.do_bin && a.do_ray && a.do_bbox) GV_LP(true, true, true, false);
  else if (a.do_bin && a.do_ray && keep) GV_LP(true, true, false, true);
  else if (a.do_bin && a.do_ray) GV_LP(true, true, false, false);
  else if (a.do_bin && a.do_bbox && keep) GV_LP(true, false, true, true);
  else if (a.do_bin && a.do_bbox) GV_LP(true, false, true, false);
  else if (a.do_bin && keep) GV_LP(true, false, false, true);
  else if (a.do_bin) GV_LP(true, false, false, false);
  else if (a.do_bbox) GV_LP(false, false, true, false);
#undef GV_LP
}

// Exact fp32 form of the first-match bbox test, built on the device from the uploaded gv_bbox
// array: (double)u >= x_min <=> u >= (smallest float >= x_min), (double)u <= x_max <=> u <= (largest
// float <= x_max) for every float u (NaN bounds stay NaN: always false), and per 16x16-pixel tile the
// mask of boxes whose float range can contain a pixel of that tile (a superset is enough: the mask
// only prunes).  One thread per (tile, 64-box word); every workgroup first converts the boxes of its
// words once into LDS (the first workgroup also stores them), so a thread's loop reads LDS only.
constexpr int kPrepBoxes = 256;   // boxes staged per round
__global__ void __launch_bounds__(256) k_bbox_prepare(const gv_bbox *__restrict__ bb, int32_t nb, int32_t tiles_x,
                                                      int32_t tiles_y, int32_t mask_words, float4 *__restrict__ bbox_f,
                                                      unsigned long long *__restrict__ tile_mask,
                                                      const uint4 *__restrict__ copy_src, uint4 *__restrict__ copy_dst,
                                                      uint32_t copy_words)
{
  __shared__ int4 s_r[kPrepBoxes];   // tile range {tx0, ty0, tx1, ty1} of a box (empty range: tx0 > tx1)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  // optional rider (gv_tick): the detection block goes from its pinned staging to the device in this kernel instead of
  // a copy command in front of it (`bb` then points INTO the staging: the boxes are read over PCIe by every workgroup)
  for (uint32_t i = (uint32_t)gid; i < copy_words; i += gridDim.x * blockDim.x) copy_dst[i] = copy_src[i];
  const int nwords = tiles_x * tiles_y * mask_words;
  const int wd = (gid < nwords) ? gid % mask_words : 0, tile = (gid < nwords) ? gid / mask_words : 0;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  unsigned long long m = 0ull;
  for (int b0 = 0; b0 < nb; b0 += kPrepBoxes) {
    __syncthreads();
    const int i = b0 + (int)threadIdx.x;
    if (i < nb) {
      const gv_bbox b = bb[i];
      const float4 f = make_float4(__double2float_ru(b.x_min), __double2float_ru(b.y_min), __double2float_rd(b.x_max),
                                   __double2float_rd(b.y_max));
      if (blockIdx.x == 0) bbox_f[i] = f;
      // the box's tile range, once per workgroup (every thread used to redo this for every box): tiles whose
      // pixel range [16t, 16t+16) can contain a u in [f.x, f.z]
      int4 r = make_int4(1, 1, 0, 0);   // empty or NaN box never matches
      if (f.x <= f.z && f.y <= f.w) {
        int tx0 = (int)floorf(fmaxf(f.x, 0.0f) / 16.0f), tx1 = (int)floorf(fminf(f.z, 16.0f * tiles_x - 1.0f) / 16.0f);
        int ty0 = (int)floorf(fmaxf(f.y, 0.0f) / 16.0f), ty1 = (int)floorf(fminf(f.w, 16.0f * tiles_y - 1.0f) / 16.0f);
        r = make_int4(max(tx0, 0), max(ty0, 0), min(tx1, tiles_x - 1), min(ty1, tiles_y - 1));
      }
      s_r[threadIdx.x] = r;
    }
    __syncthreads();
    const int lo = max(b0, 64 * wd), hi = min(min(nb, b0 + kPrepBoxes), 64 * wd + 64);
    for (int q0 = lo; q0 < hi; q0 += 8) {   // eight LDS reads in flight, predicated (no rolled remainder loop)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u;
        const int4 r = s_r[min(q, hi - 1) - b0];
        if (q < hi && tx >= r.x && tx <= r.z && ty >= r.y && ty <= r.w) m |= 1ull << (q & 63);
      }
    }
  }
  if (gid < nwords) tile_mask[gid] = m;
}

void launch_bbox_prepare(const gv_bbox *bboxes, int32_t nb, int32_t tiles_x, int32_t tiles_y, int32_t mask_words,
                         float4 *bbox_f, unsigned long long *tile_mask, hipStream_t s, const void *copy_src, void *copy_dst,
                         size_t copy_bytes)
{
  const int n = std::max(tiles_x * tiles_y * mask_words, 1);   // (>= 1 workgroup: it also stores the thresholds)
  hipLaunchKernelGGL(k_bbox_prepare, dim3((n + 255) / 256), dim3(256), 0, s, bboxes, nb, tiles_x, tiles_y, mask_words,
                     bbox_f, tile_mask, static_cast<const uint4 *>(copy_src), static_cast<uint4 *>(copy_dst),
                     (uint32_t)((copy_bytes + 15) / 16));
}

// The packed grid to pinned host memory by a kernel of its own (gv_publish_grid_async): 16-byte stores of consecutive
// lanes, i.e. posted PCIe writes of whole lines, from a few workgroups -- the copy engines stay with the cloud uploads.
__global__ void __launch_bounds__(256) k_publish_grid(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}

void launch_publish_grid(const int8_t *src, int8_t *dst_host, size_t bytes, int blocks, hipStream_t s)
{
  const uint32_t n16 = (uint32_t)(bytes / 16);
  if (n16) hipLaunchKernelGGL(k_publish_grid, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4 *>(src),
                              reinterpret_cast<uint4 *>(dst_host), n16);
}

// A1 standalone: camera-frame copy of the cloud (transformLidarToCamera)
__global__ void __launch_bounds__(256) k_transform(const float *__restrict__ x, const float *__restrict__ y,
                                                   const float *__restrict__ z, uint32_t n, Mat34f m,
                                                   float *__restrict__ ox, float *__restrict__ oy,
                                                   float *__restrict__ oz)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a, b, c;
    xform34(m, x[i], y[i], z[i], a, b, c);
    ox[i] = a;
    oy[i] = b;
    oz[i] = c;
  }
}

void launch_transform_cloud(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m,
                            float *ox, float *oy, float *oz, hipStream_t s)
{
  if (!n) return;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 255) / 256, (uint64_t)4096);
  hipLaunchKernelGGL(k_transform, dim3(blocks), dim3(256), 0, s, x, y, z, n, m, ox, oy, oz);
}

// PointCloud2 bytes -> SoA (pcl::fromROSMsg, src/grid_vision_node.cpp:105).  One
// thread per point; fields may be unaligned inside point_step, so assemble from bytes
// when the offsets are not 4-byte aligned.
__global__ void __launch_bounds__(256) k_deinterleave(const uint8_t *__restrict__ d, uint32_t n,
                                                      uint32_t step, uint32_t offx, uint32_t offy,
                                                      uint32_t offz, float *__restrict__ x,
                                                      float *__restrict__ y, float *__restrict__ z,
                                                      bool aligned)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint8_t *p = d + (size_t)i * step;
    if (aligned) {
      x[i] = *reinterpret_cast<const float *>(p + offx);
      y[i] = *reinterpret_cast<const float *>(p + offy);
      z[i] = *reinterpret_cast<const float *>(p + offz);
    } else {
      uint32_t w[3];
      const uint32_t off[3] = {offx, offy, offz};
      for (int k = 0; k < 3; ++k)
        w[k] = (uint32_t)p[off[k]] | ((uint32_t)p[off[k] + 1] << 8) | ((uint32_t)p[off[k] + 2] << 16)
               | ((uint32_t)p[off[k] + 3] << 24);
      x[i] = __uint_as_float(w[0]);
      y[i] = __uint_as_float(w[1]);
      z[i] = __uint_as_float(w[2]);
    }
  }
}

void launch_deinterleave(const uint8_t *data, uint32_t n, uint32_t point_step, uint32_t off_x,
                         uint32_t off_y, uint32_t off_z, float *x, float *y, float *z, hipStream_t s)
{
  if (!n) return;
  const bool aligned = ((point_step | off_x | off_y | off_z) & 3u) == 0 && (((uintptr_t)data) & 3u) == 0;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 255) / 256, (uint64_t)4096);
  hipLaunchKernelGGL(k_deinterleave, dim3(blocks), dim3(256), 0, s, data, n, point_step, off_x, off_y,
                     off_z, x, y, z, aligned);
}

// ------------------------------------------------------------- rectangles --
__global__ void k_rects_from_poses(const gv_lshape_pose *__restrict__ poses, int32_t n, GridParams g,
                                   bool from_cam, Xform64 bc, Rect *__restrict__ rects)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  gv_lshape_pose p = poses[i];
  if (from_cam) {
    // tf2::doTransform(Pose): position = basis * v + origin  (grid_vision_node.cpp:370-374)
    const double vx = p.px, vy = p.py, vz = p.pz;
    p.px = ((bc.b[0] * vx + bc.b[1] * vy) + bc.b[2] * vz) + bc.o[0];
    p.py = ((bc.b[3] * vx + bc.b[4] * vy) + bc.b[5] * vz) + bc.o[1];
  }
  rects[i] = rect_from_pose(g, p);
}

void launch_rects_from_poses(const gv_lshape_pose *poses, int32_t n, const GridParams &g, bool from_cam,
                             const Xform64 &bc, Rect *rects, hipStream_t s)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rects_from_poses, dim3((n + 63) / 64), dim3(64), 0, s, poses, n, g, from_cam, bc, rects);
}

// getEstimatedDepth  src/occupancy_grid.cpp:185-196
__device__ __forceinline__ float estimated_depth(int label)
{
  switch (label) {
  case 9: return 3.5f;   // VEHICLE
  case 2: return 0.6f;   // PERSON
  case 0: return 2.5f;   // BIKE
  case 1: return 2.5f;   // MOTORBIKE
  default: return -1.0f;
  }
}

// computeBoundingBox3D  src/occupancy_grid.cpp:107-138 (dead code in the reference)
__global__ void k_rects_from_points(const double *__restrict__ pts, const gv_bbox *__restrict__ bb, int32_t n,
                                    GridParams g, Rect *__restrict__ rects)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = estimated_depth(bb[i].label);
  const double x = pts[3 * i], y = pts[3 * i + 1];
  const double c[8] = {x + d, y + (d / 2), x + d, y - (d / 2), x, y - (d / 2), x, y + (d / 2)};
  rects[i] = rect_from_corners(g, c);
}

void launch_rects_from_points(const double *pts_xyz, const gv_bbox *bboxes, int32_t n, const GridParams &g,
                              Rect *rects, hipStream_t s)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rects_from_points, dim3((n + 63) / 64), dim3(64), 0, s, pts_xyz, bboxes, n, g, rects);
}

// ------------------------------------------------- vision orientation (A13/A14) --
// Eigen::ColPivHouseholderQR<Matrix<float,4,3>>::solve, restated; one lane each.
__device__ void qr_solve_4x3(const float Ain[12], const float bin[4], float x[3])
{
  float a[4][3], c[4], hc[3], ncu[3], ncd[3];
  int transp[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) a[i][j] = Ain[i * 3 + j];
    c[i] = bin[i];
  }
  float maxn = 0.0f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += a[i][j] * a[i][j];
    ncu[j] = ncd[j] = sqrtf(s);
    if (ncu[j] > maxn) maxn = ncu[j];
  }
  const float eps = FLT_EPSILON;
  const float th = maxn * eps / 4.0f;
  const float threshold_helper = th * th;
  const float downdate_thr = sqrtf(eps);
  int nonzero = 3;
  float maxpivot = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int big = k;
    float bigv = ncu[k];
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
      if (ncu[j] > bigv) { bigv = ncu[j]; big = j; }
    const float big_sq = bigv * bigv;
    if (nonzero == 3 && big_sq < threshold_helper * (float)(4 - k)) nonzero = k;
    transp[k] = big;
    if (big != k) {
#pragma unroll
      for (int j = k + 1; j < 3; ++j) {
        if (j == big) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float t = a[i][k]; a[i][k] = a[i][j]; a[i][j] = t; }
          float t = ncu[k]; ncu[k] = ncu[j]; ncu[j] = t;
          t = ncd[k]; ncd[k] = ncd[j]; ncd[j] = t;
        }
      }
    }
    float tail = 0.0f;
#pragma unroll
    for (int i = k + 1; i < 4; ++i) tail += a[i][k] * a[i][k];
    const float c0 = a[k][k];
    float tau, beta;
    if (tail <= FLT_MIN) {
      tau = 0.0f;
      beta = c0;
#pragma unroll
      for (int i = k + 1; i < 4; ++i) a[i][k] = 0.0f;
    } else {
      beta = sqrtf(c0 * c0 + tail);
      if (c0 >= 0.0f) beta = -beta;
#pragma unroll
      for (int i = k + 1; i < 4; ++i) a[i][k] = a[i][k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hc[k] = tau;
    a[k][k] = beta;
    if (fabsf(beta) > maxpivot) maxpivot = fabsf(beta);
    if (tau != 0.0f) {
#pragma unroll
      for (int j = k + 1; j < 3; ++j) {
        float tmp = 0.0f;
#pragma unroll
        for (int i = k + 1; i < 4; ++i) tmp += a[i][k] * a[i][j];
        tmp += a[k][j];
        a[k][j] -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < 4; ++i) a[i][j] -= tau * a[i][k] * tmp;
      }
    }
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
      if (ncu[j] != 0.0f) {
        float t = fabsf(a[k][j]) / ncu[j];
        t = (1.0f + t) * (1.0f - t);
        t = t < 0.0f ? 0.0f : t;
        const float r = ncu[j] / ncd[j];
        const float t2 = t * r * r;
        if (t2 <= downdate_thr) {
          float s = 0.0f;
#pragma unroll
          for (int i = k + 1; i < 4; ++i) s += a[i][j] * a[i][j];
          ncd[j] = sqrtf(s);
          ncu[j] = ncd[j];
        } else
          ncu[j] *= sqrtf(t);
      }
    }
  }
  int perm[3] = {0, 1, 2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // perm[k] <-> perm[transp[k]] with register-resident selects
    const int tk = transp[k];
    const int pk = perm[k];
    const int pt = (tk == 0) ? perm[0] : (tk == 1) ? perm[1] : perm[2];
    perm[k] = pt;
    if (tk == 0) perm[0] = (k == 0) ? pt : pk;
    if (tk == 1) perm[1] = (k == 1) ? pt : pk;
    if (tk == 2) perm[2] = (k == 2) ? pt : pk;
  }
  const float prethr = fabsf(maxpivot) * (eps * 3.0f);
  int rank = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (i < nonzero && fabsf(a[i][i]) > prethr) ++rank;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (hc[k] != 0.0f) {
      float tmp = 0.0f;
#pragma unroll
      for (int i = k + 1; i < 4; ++i) tmp += a[i][k] * c[i];
      tmp += c[k];
      c[k] -= hc[k] * tmp;
#pragma unroll
      for (int i = k + 1; i < 4; ++i) c[i] -= hc[k] * a[i][k] * tmp;
    }
  }
  float y[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    if (i < rank) {
      float s = c[i];
#pragma unroll
      for (int j = i + 1; j < 3; ++j)
        if (j < rank) s -= a[i][j] * y[j];
      y[i] = s / a[i][i];
    }
  }
  x[0] = x[1] = x[2] = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i < rank) {
      const int p = perm[i];
      if (p == 0) x[0] = y[i];
      if (p == 1) x[1] = y[i];
      if (p == 2) x[2] = y[i];
    }
  }
}

// One wavefront per bbox: the 2 x 4 x 2 x 4 = 64 constraint combinations of
// calcLocation (src/vision_orientation.cpp:359-374) are exactly the 64 lanes.
__global__ void __launch_bounds__(64) k_vision(const float *__restrict__ orient, const float *__restrict__ conf,
                                               const float *__restrict__ dims, const gv_bbox *__restrict__ bboxes,
                                               int32_t nb, gv_cam_params cam, VisionOut *__restrict__ out,
                                               gv_lshape_pose *__restrict__ poses_cam)
{
  const int bi = blockIdx.x;
  if (bi >= nb) return;
  const int lane = threadIdx.x;
  const gv_bbox bb = bboxes[bi];
  const float *cs = conf + bi * 2, *os = orient + bi * 4, *ds = dims + bi * 3;
  // postProcessOutputs :466-470
  const int argmax = (cs[1] > cs[0]) ? 1 : 0;
  // generateBins(2) :241-258
  const float interval = (float)(2.0f * 3.14159265358979323846 / 2);
  const float bin = (argmax ? interval : 0.0f) + interval / 2.0f;
  // computeAlpha :260-275
  // The float trig of the device library differs from glibc's by ulps; each one is evaluated in fp64 and
  // rounded once instead, which is the correctly rounded float result (what glibc returns except for its
  // own rare last-bit misses), so alpha / theta_ray / R agree with the host and the 64 residuals with them.
  float alpha = (float)atan2((double)os[argmax * 2 + 1], (double)os[argmax * 2 + 0]);
  alpha += bin;
  alpha -= (float)3.14159265358979323846;
  // computeThetaRay :277-292
  const float fovx = 2.0f * (float)atan((double)(cam.orig_w / (2.0f * cam.fx)));
  const float box_center_x = (float)((bb.x_min + bb.x_max) / 2.0f);
  float ddx = box_center_x - (cam.orig_w / 2.0f);
  const float sign = (ddx < 0) ? -1.0f : 1.0f;
  ddx = fabsf(ddx);
  float theta_ray = (float)atan((double)((2.0f * ddx * (float)tan((double)(fovx / 2.0f))) / cam.orig_w));
  theta_ray *= sign;
  // class averages  include/grid_vision/vision_orientation.hpp:58-69, dims :472-495
  float al = 0, aw = 0, ah = 0;
  int valid = 1;
  switch (bb.label) {
  case 9: al = 3.884f; aw = 1.629f; ah = 1.526f; break;
  case 0: al = 1.763f; aw = 0.597f; ah = 1.737f; break;
  case 1: al = 2.2f; aw = 0.8f; ah = 1.5f; break;
  case 2: al = 0.842f; aw = 0.660f; ah = 1.761f; break;
  default: valid = 0; break;
  }
  const float len = ds[2] + al, wid = ds[0] + aw, hgt = ds[1] + ah;
  // calcLocation :294-447
  const float orient_f = alpha + theta_ray;
  const float c = (float)cos((double)orient_f), s = (float)sin((double)orient_f);
  const float Rm[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  const float box[4] = {(float)bb.x_min, (float)bb.y_min, (float)bb.x_max, (float)bb.y_max};
  const float hx = (float)((double)len / 2.0f);
  const float hy = (float)((double)wid / 2.0f);
  const float hz = (float)((double)hgt / 2.0f);
  int left_mult = 1, right_mult = -1;
  const float deg88 = (float)(88 * 3.14159265358979323846 / 180.0f);
  const float deg90 = (float)(90 * 3.14159265358979323846 / 180.0f);
  const float deg92 = (float)(92 * 3.14159265358979323846 / 180.0f);
  if (alpha < deg92 && alpha > deg88) { left_mult = 1; right_mult = 1; }
  else if (alpha < -deg88 && alpha > -deg92) { left_mult = -1; right_mult = -1; }
  else if (alpha < deg90 && alpha > -deg90) { left_mult = -1; right_mult = 1; }
  const int switch_mult = (alpha > 0) ? 1 : -1;
  // lane -> (left, top, right, bottom) in the reference's loop order :363-374
  const int l = lane >> 5, t = (lane >> 3) & 3, r = (lane >> 2) & 1, b = lane & 3;
  const float li = l ? 1.0f : -1.0f, ri = r ? 1.0f : -1.0f;
  const float ti = (t >> 1) ? 1.0f : -1.0f, tj = (t & 1) ? 1.0f : -1.0f;
  const float bi_ = (b >> 1) ? 1.0f : -1.0f, bj = (b & 1) ? 1.0f : -1.0f;
  const float X[4][3] = {{left_mult * hx, li * hy, -switch_mult * hz},
                         {ti * hx, -hy, tj * hz},
                         {right_mult * hx, ri * hy, switch_mult * hz},
                         {bi_ * hx, hy, bj * hz}};
  const float P[3][4] = {{cam.fx, 0.0f, cam.cx, 0.0f}, {0.0f, cam.fy, cam.cy, 0.0f}, {0.0f, 0.0f, 1.0f, 0.0f}};
  float A[12], bv[4];
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    float RX[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) RX[q] = (Rm[q * 3] * X[row][0] + Rm[q * 3 + 1] * X[row][1]) + Rm[q * 3 + 2] * X[row][2];
    float pM3[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) pM3[q] = ((P[q][0] * RX[0] + P[q][1] * RX[1]) + P[q][2] * RX[2]) + P[q][3] * 1.0f;
    const int idx = row & 1;   // indices = {0,1,0,1} :379
    const float v = box[row];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) A[row * 3 + cc] = P[idx][cc] - v * P[2][cc];   // :412 (pM(:, :3) == P(:, :3))
    bv[row] = v * pM3[2] - pM3[idx];                                             // :415
  }
  float loc[3];
  qr_solve_4x3(A, bv, loc);
  float err = 0.0f;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const float rr = ((A[row * 3] * loc[0] + A[row * 3 + 1] * loc[1]) + A[row * 3 + 2] * loc[2]) - bv[row];
    err += rr * rr;
  }
  // sequential "if (error < best_error)" from FLT_MAX (:382,:424) == min with lowest index on ties
  float best = (err < FLT_MAX) ? err : FLT_MAX;
  int who = (err < FLT_MAX) ? lane : 64;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int ow = __shfl_xor(who, off);
    if (ob < best || (ob == best && ow < who)) { best = ob; who = ow; }
  }
  const int src = (who < 64) ? who : 0;
  const float l0 = __shfl(loc[0], src), l1 = __shfl(loc[1], src), l2 = __shfl(loc[2], src);
  if (lane == 0) {
    VisionOut o;
    o.loc[0] = (who < 64) ? l0 : 0.0f;
    o.loc[1] = (who < 64) ? l1 : 0.0f;
    o.loc[2] = (who < 64) ? l2 : 0.0f;
    o.orient = orient_f;
    o.err = best;
    o.dims[0] = len; o.dims[1] = wid; o.dims[2] = hgt;
    o.valid = valid;
    out[bi] = o;
    gv_lshape_pose p;
    p.px = o.loc[0]; p.py = o.loc[1]; p.pz = o.loc[2];
    p.qx = 0; p.qy = 0; p.qz = 0; p.qw = 1;
    p.length = valid ? (double)len : -1.0;   // length < 0 marks "skipped" for k_rects_from_poses
    p.width = wid;
    p.height = hgt;
    poses_cam[bi] = p;
  }
}

void launch_vision(const float *orient, const float *conf, const float *dims, const gv_bbox *bboxes, int32_t nb,
                   const gv_cam_params &cam, VisionOut *out, gv_lshape_pose *poses_cam, hipStream_t s)
{
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_vision, dim3(nb), dim3(64), 0, s, orient, conf, dims, bboxes, nb, cam, out, poses_cam);
}

// --------------------------------------------------------- ray march (X2) --
// Pass 1: compact the ray ends.  A cell is an end if it has hits (end exclusive)
// and/or a clipped out-of-map ray ends there (end inclusive); both at once is the
// inclusive ray (same path).  entry = cell | include_end << 31.
__global__ void __launch_bounds__(256) k_ray_compact(const int32_t *__restrict__ hits,
                                                     const uint8_t *__restrict__ clip_end, int32_t G,
                                                     uint32_t *__restrict__ list, uint32_t *__restrict__ count)
{
  const int lane = threadIdx.x & 63;
  const int32_t stride = gridDim.x * blockDim.x;
  for (int32_t base = blockIdx.x * blockDim.x; base < G; base += stride) {
    const int32_t c = base + threadIdx.x;
    bool is_end = false;
    uint32_t e = 0;
    if (c < G) {
      const int h = hits[c];
      const uint8_t k = clip_end[c];
      is_end = (h > 0) || k;
      e = (uint32_t)c | (k ? 0x80000000u : 0u);
    }
    const unsigned long long m = __ballot(is_end);
    if (m) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(count, (uint32_t)__popcll(m));
      b = __shfl(b, 0);
      if (is_end) list[b + __popcll(m & ((1ull << lane) - 1ull))] = e;
    }
  }
}

void launch_ray_compact(const int32_t *hits, const uint8_t *clip_end, const GridParams &g, uint32_t *list,
                        uint32_t *count, hipStream_t s)
{
  const uint32_t blocks = (uint32_t)std::min<int64_t>(((int64_t)g.G + 255) / 256, (int64_t)256 * 8);
  hipLaunchKernelGGL(k_ray_compact, dim3(blocks), dim3(256), 0, s, hits, clip_end, g.G, list, count);
}

// Pass 2: one wavefront per ray, lanes over the steps.  grid_map::LineIterator
// stepping in closed form: after i steps along the major axis the minor offset is
// (major/2 + i*minor) / major (integer division).
__global__ void __launch_bounds__(256) k_ray_march(const uint32_t *__restrict__ list,
                                                   const uint32_t *__restrict__ count, GridParams g,
                                                   RayOrigin o, uint8_t *__restrict__ miss,
                                                   unsigned long long *__restrict__ stats)
{
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t n = *count;
  unsigned long long visits = 0;
  for (uint32_t r = wave; r < n; r += nwaves) {
    const uint32_t e = list[r];
    const int inc = (int)(e >> 31);
    const int cell = (int)(e & 0x7fffffffu);
    const int ey = cell / g.nx, ex = cell - ey * g.nx;
    const int dx = ex - o.cx, dy = ey - o.cy;
    const int adx = abs(dx), ady = abs(dy);
    const int sx = (dx >= 0) ? 1 : -1, sy = (dy >= 0) ? 1 : -1;
    const bool xmaj = adx >= ady;
    const int major = xmaj ? adx : ady, minor = xmaj ? ady : adx;
    const int half = major / 2;
    const int nmark = inc ? major + 1 : major;
    const double rcp = major ? 1.0 / (double)major : 0.0;
    for (int i = lane; i < nmark; i += 64) {
      const int num = half + i * minor;
      int q = (int)((double)num * rcp);
      const int rem = num - q * major;        // exact integer quotient after one correction
      if (rem >= major) ++q;
      else if (rem < 0) --q;
      const int cx = o.cx + sx * (xmaj ? i : q);
      const int cy = o.cy + sy * (xmaj ? q : i);
      miss[(size_t)cy * g.nx + cx] = 1;
    }
    visits += (unsigned long long)nmark;
  }
  if (lane == 0 && stats) {
    if (visits) atomicAdd(&stats[1], visits);
    if (wave == 0) atomicAdd(&stats[0], (unsigned long long)n);
  }
}

void launch_ray_march(const uint32_t *list, const uint32_t *count, const GridParams &g, const RayOrigin &org,
                      uint8_t *miss, unsigned long long *stats, hipStream_t s)
{
  if (!org.valid) return;
  hipLaunchKernelGGL(k_ray_march, dim3(256 * 8), dim3(256), 0, s, list, count, g, org, miss, stats);
}

// ---------------------------------------------------------- grid finalise --
// One pass over the cells (A7/A8/A9/A18 + the X2 rule):
//   l += decay (:19/:69); l += 0.85f per covering rectangle, in object order (:175-182);
//   hits>0 -> l += 1.2f, else miss>0 -> l += -0.4f  [EXTENSION];
//   clamp (:21-22); occupancy = 1/(1+exp(-l)) (:25-30); int8 = (int8)(clamp01(p)*100)
//   stored at data[G-1-cell] (toOccupancyGrid).
__device__ __forceinline__ float sigmoid_ref(float l)
{
  // expf evaluated in fp64 and rounded once: agrees with a correctly rounded expf
  const float e = (float)exp((double)(-l));
  return 1.0f / (1.0f + e);
}

__device__ __forceinline__ int8_t pack_i8(float p)
{
  float v = (p - 0.0f) / (1.0f - 0.0f);
  if (isnan(v)) v = -1.0f;
  else {
    float c = v < 0.0f ? 0.0f : v;
    c = c > 1.0f ? 1.0f : c;
    v = 0.0f + c * 100.0f;
  }
  return (int8_t)v;
}

__device__ __forceinline__ float cell_update(float l, int nrect_hits, bool counts, int h, int m)
{
  l = l + kLogOddsDecay;
  for (int k = 0; k < nrect_hits; ++k) l = l + kRectIncrement;
  if (counts) {
    if (h > 0) l = l + kLogOddsOccupied;
    else if (m > 0) l = l + kLogOddsFree;
  }
  l = (l < kMinLogOdds) ? kMinLogOdds : l;
  l = (l > kMaxLogOdds) ? kMaxLogOdds : l;
  return l;
}

template <bool COUNTS>
__global__ void __launch_bounds__(256) k_finalize_vec4(FinalizeArgs a)
{
  // requires nx % 4 == 0 and band edges % 4 == 0: the 4 cells of a thread share a row
  const int64_t q0 = a.cell_begin >> 2, q1 = a.cell_end >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = q0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < q1; q += stride) {
    const int64_t c = q << 2;
    const int iy = (int)(c / a.g.nx), ix = (int)(c - (int64_t)iy * a.g.nx);
    float4 l4 = *reinterpret_cast<const float4 *>(a.log_odds + c);
    int4 h4 = make_int4(0, 0, 0, 0);
    uint32_t m4 = 0;
    if (COUNTS) {
      h4 = *reinterpret_cast<const int4 *>(a.hits + c);
      m4 = *reinterpret_cast<const uint32_t *>(a.miss + c);
    }
    int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    for (int r = 0; r < a.n_rects; ++r) {
      const Rect R = a.rects[r];
      if (R.valid && iy >= R.y0 && iy <= R.y1 && ix + 3 >= R.x0 && ix <= R.x1) {
        k0 += (ix >= R.x0 && ix <= R.x1);
        k1 += (ix + 1 >= R.x0 && ix + 1 <= R.x1);
        k2 += (ix + 2 >= R.x0 && ix + 2 <= R.x1);
        k3 += (ix + 3 >= R.x0 && ix + 3 <= R.x1);
      }
    }
    l4.x = cell_update(l4.x, k0, COUNTS, h4.x, (int)(m4 & 0xffu));
    l4.y = cell_update(l4.y, k1, COUNTS, h4.y, (int)((m4 >> 8) & 0xffu));
    l4.z = cell_update(l4.z, k2, COUNTS, h4.z, (int)((m4 >> 16) & 0xffu));
    l4.w = cell_update(l4.w, k3, COUNTS, h4.w, (int)(m4 >> 24));
    float4 p4;
    p4.x = sigmoid_ref(l4.x); p4.y = sigmoid_ref(l4.y); p4.z = sigmoid_ref(l4.z); p4.w = sigmoid_ref(l4.w);
    *reinterpret_cast<float4 *>(a.log_odds + c) = l4;
    *reinterpret_cast<float4 *>(a.occupancy + c) = p4;
    // data[G-1-cell]: cells c..c+3 land at G-4-c .. G-1-c in reverse order
    const uint32_t packed = ((uint32_t)(uint8_t)pack_i8(p4.w)) | ((uint32_t)(uint8_t)pack_i8(p4.z) << 8)
                            | ((uint32_t)(uint8_t)pack_i8(p4.y) << 16) | ((uint32_t)(uint8_t)pack_i8(p4.x) << 24);
    *reinterpret_cast<uint32_t *>(a.occ_i8 + ((int64_t)a.g.G - 4 - c)) = packed;
    if (COUNTS && a.zero_counts) {
      *reinterpret_cast<int4 *>(a.hits + c) = make_int4(0, 0, 0, 0);
      *reinterpret_cast<uint32_t *>(a.miss + c) = 0u;
      *reinterpret_cast<uint32_t *>(a.clip_end + c) = 0u;
    }
  }
}

template <bool COUNTS>
__global__ void __launch_bounds__(256) k_finalize_scalar(FinalizeArgs a)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = a.cell_begin + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < a.cell_end; c += stride) {
    const int iy = (int)(c / a.g.nx), ix = (int)(c - (int64_t)iy * a.g.nx);
    int k = 0;
    for (int r = 0; r < a.n_rects; ++r) {
      const Rect R = a.rects[r];
      k += (R.valid && iy >= R.y0 && iy <= R.y1 && ix >= R.x0 && ix <= R.x1);
    }
    const int h = COUNTS ? a.hits[c] : 0;
    const int m = COUNTS ? (int)a.miss[c] : 0;
    const float l = cell_update(a.log_odds[c], k, COUNTS, h, m);
    const float p = sigmoid_ref(l);
    a.log_odds[c] = l;
    a.occupancy[c] = p;
    a.occ_i8[(int64_t)a.g.G - 1 - c] = pack_i8(p);
    if (COUNTS && a.zero_counts) {
      a.hits[c] = 0;
      a.miss[c] = 0;
      a.clip_end[c] = 0;
    }
  }
}

void launch_finalize(const FinalizeArgs &a, hipStream_t s)
{
  const int64_t ncell = a.cell_end - a.cell_begin;
  if (ncell <= 0) return;
  const bool vec = (a.g.nx % 4 == 0) && (a.cell_begin % 4 == 0) && (a.cell_end % 4 == 0);
  const bool counts = a.hits != nullptr;
  if (vec) {
    const uint32_t blocks = (uint32_t)std::min<int64_t>((ncell / 4 + 255) / 256, (int64_t)256 * 16);
    if (counts) hipLaunchKernelGGL(k_finalize_vec4<true>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_finalize_vec4<false>, dim3(blocks), dim3(256), 0, s, a);
  } else {
    const uint32_t blocks = (uint32_t)std::min<int64_t>((ncell + 255) / 256, (int64_t)256 * 16);
    if (counts) hipLaunchKernelGGL(k_finalize_scalar<true>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_finalize_scalar<false>, dim3(blocks), dim3(256), 0, s, a);
  }
}

// ------------------------------------------------------------------ misc --
__global__ void k_fill_f32(float *p, float v, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

void launch_fill_f32(float *p, float v, size_t n, hipStream_t s)
{
  if (!n) return;
  const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, (size_t)4096);
  hipLaunchKernelGGL(k_fill_f32, dim3(blocks), dim3(256), 0, s, p, v, n);
}

// one wavefront that stays on the device for `ticks` of the constant-rate clock (100 MHz): the queue probe of
// gv_create (is the upload stream's hardware queue shared with a busy stream?)
__global__ void k_hold(unsigned long long ticks, unsigned *sink)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned n = 0;
  while ((unsigned long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    ++n;
  }
  if (sink && n == 0xFFFFFFFFu) *sink = n;
}

void launch_hold(unsigned long long ticks, hipStream_t s)
{
  hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, s, ticks, (unsigned *)nullptr);
}

__global__ void k_i16_to_i32(const int16_t *in, int32_t *out, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (int32_t)in[i];
}

void launch_i16_to_i32(const int16_t *in, int32_t *out, size_t n, hipStream_t s)
{
  if (!n) return;
  const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, (size_t)4096);
  hipLaunchKernelGGL(k_i16_to_i32, dim3(blocks), dim3(256), 0, s, in, out, n);
}

__global__ void k_u8_to_i32(const uint8_t *in, int32_t *out, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i] ? 1 : 0;
}

void launch_u8_to_i32(const uint8_t *in, int32_t *out, size_t n, hipStream_t s)
{
  if (!n) return;
  const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, (size_t)4096);
  hipLaunchKernelGGL(k_u8_to_i32, dim3(blocks), dim3(256), 0, s, in, out, n);
}

}  // namespace gv
