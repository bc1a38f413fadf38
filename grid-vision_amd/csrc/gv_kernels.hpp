// gv_kernels.hpp -- launchers of the gfx950 kernels (defined in gv_kernels.hip).
// Every launcher only enqueues work on `s`; none allocates or synchronises.
#pragma once

#include <hip/hip_runtime.h>

#include "gv_types.hpp"

namespace gv {

// exact fp32 form of the bbox test + per-16x16-pixel-tile candidate masks (built on the device by
// launch_bbox_prepare from the uploaded gv_bbox array)
// Diagnostic build only (-DGV_DIAG, GV_TIMELINE=1): every workgroup of a launch reports the constant-rate clock when
// it starts and when it ends; min / max over the launch = the kernel's residence on the device, taken without any
// packet in the queues (tools/native_timeline.py).  Nothing in the shipped kernels.
#ifdef GV_DIAG
// (begin: the first workgroup of the launch only; end: every eighth workgroup -- one atomic per workgroup on one
//  address slowed the 1024-workgroup grid pass by a third)
#define GV_TL_LINEAR_ID (blockIdx.x + blockIdx.y * gridDim.x)
#define GV_TL_BEGIN(tl) do { if ((tl) && threadIdx.x == 0 && GV_TL_LINEAR_ID == 0) atomicMin(&(tl)[0], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define GV_TL_END(tl) do { if ((tl) && threadIdx.x == 0 && (GV_TL_LINEAR_ID & 7u) == 0) atomicMax(&(tl)[1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define GV_TL_BEGIN(tl) do { } while (0)
#define GV_TL_END(tl) do { } while (0)
#endif

struct BBoxTest {
  const float4 *bbox_f;                  // (x_min, y_min, x_max, y_max) as float thresholds
  const unsigned long long *tile_mask;   // [tiles_y][tiles_x][mask_words]
  int32_t tiles_x, tiles_y, mask_words;
};

struct PointsArgs {
  const float *x, *y, *z;
  uint32_t n;
  GridParams g;
  Mat34f m_base;     // base <- lidar   (X1/X2)
  Mat34f m_cam;      // camera <- lidar (A1/A5)
  CamK cam;
  RayOrigin org;
  BBoxTest bt;
  int32_t *hits;       // G   hits[cell] += 1
  uint8_t *clip_end;   // G
  int32_t *cell_idx;   // N or null
  int16_t *bbox_id;    // N or null
  bool do_bin, do_ray, do_bbox;
};
void launch_points(const PointsArgs &a, hipStream_t s);
// float thresholds + 16x16-pixel tile candidate masks of the bbox test, from the device copy of the bboxes
// copy_src / copy_dst / copy_bytes (optional): the kernel also copies that many bytes (rounded up to 16) first -- the
// detection block from its pinned, device-visible staging, `bboxes` then pointing into the staging
void launch_bbox_prepare(const gv_bbox *bboxes, int32_t nb, int32_t tiles_x, int32_t tiles_y, int32_t mask_words,
                         float4 *bbox_f, unsigned long long *tile_mask, hipStream_t s, const void *copy_src = nullptr,
                         void *copy_dst = nullptr, size_t copy_bytes = 0);

void launch_transform_cloud(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m,
                            float *ox, float *oy, float *oz, hipStream_t s);
void launch_deinterleave(const uint8_t *data, uint32_t n, uint32_t point_step, uint32_t off_x,
                         uint32_t off_y, uint32_t off_z, float *x, float *y, float *z, hipStream_t s);

// poses -> index rectangles (updateMap corners + updateGridCellsFast index part).
// from_cam: poses are camera-frame, apply `bc` (base <- camera) to the position first.
void launch_rects_from_poses(const gv_lshape_pose *poses, int32_t n, const GridParams &g, bool from_cam,
                             const Xform64 &bc, Rect *rects, hipStream_t s);
// dead-code overload: centre points + class depth (computeBoundingBox3D)
void launch_rects_from_points(const double *pts_xyz, const gv_bbox *bboxes, int32_t n,
                              const GridParams &g, Rect *rects, hipStream_t s);

// vision-orientation geometry, one wavefront per bbox; also emits camera-frame
// gv_lshape_pose (position + dims only; the quaternion is filled by the host)
void launch_vision(const float *orient, const float *conf, const float *dims, const gv_bbox *bboxes,
                   int32_t nb, const gv_cam_params &cam, VisionOut *out, gv_lshape_pose *poses_cam,
                   hipStream_t s);

void launch_ray_compact(const int32_t *hits, const uint8_t *clip_end, const GridParams &g,
                        uint32_t *list, uint32_t *count, hipStream_t s);
void launch_ray_march(const uint32_t *list, const uint32_t *count, const GridParams &g,
                      const RayOrigin &org, uint8_t *miss, unsigned long long *stats, hipStream_t s);

struct FinalizeArgs {
  GridParams g;
  float *log_odds, *occupancy;
  int8_t *occ_i8;          // OccupancyGrid.data order (reversed linear)
  const Rect *rects;
  int32_t n_rects;
  int32_t *hits;           // null => no hit/miss rule (plain updateMap)
  uint8_t *miss;
  uint8_t *clip_end;
  bool zero_counts;        // clear hits/miss/clip_end for the next frame
  int64_t cell_begin, cell_end;   // band of cells to finalise ([0,G) on one GPU)
};
void launch_finalize(const FinalizeArgs &a, hipStream_t s);

void launch_fill_f32(float *p, float v, size_t n, hipStream_t s);
// bytes (a multiple of 16) from device memory to pinned, device-visible host memory by `blocks` workgroups
void launch_publish_grid(const int8_t *src, int8_t *dst_host, size_t bytes, int blocks, hipStream_t s);
void launch_hold(unsigned long long ticks_100mhz, hipStream_t s);   // one idle wavefront for that long (queue probe)
void launch_u8_to_i32(const uint8_t *in, int32_t *out, size_t n, hipStream_t s);
void launch_i16_to_i32(const int16_t *in, int32_t *out, size_t n, hipStream_t s);

// ---- tile-path binning: partition by 128x128-cell tile + per-tile LDS histogram (gv_binning.hip) ----
constexpr int kBinTileLog = 7;
constexpr int kBinTile = 1 << kBinTileLog;             // cells per tile side
constexpr int kBinTileCells = kBinTile * kBinTile;     // 16384: an int32 tile is 64 KB of LDS
constexpr int kBinSplitMax = 8;                        // workgroups that may share one crowded tile
constexpr uint32_t kBinSplitKeys = 32768;              // keys per share of a crowded tile

struct BinArgs {
  const float *x, *y, *z;
  uint32_t n;
  GridParams g;
  Mat34f m_base, m_cam;
  CamK cam;
  RayOrigin org;
  BBoxTest bt;
  int32_t nb, nb_pad;      // bboxes of the test (do_bbox); nb rounded up to a multiple of 4
  int16_t *bbox_id;        // N (do_bbox)
  int32_t *cell_idx;       // N or null
  bool do_ray, do_bbox;    // do_bbox needs bin_bbox_fits(): the test's tables are staged in LDS
  uint32_t chunk;          // points per partition workgroup (multiple of 256, <= 32768)
  uint32_t n_wg;           // ceil(n / chunk)
  int32_t tiles_x, tiles_y, n_tiles;
  uint16_t *keys;          // [n_wg][chunk]  keys of a chunk grouped by tile
  uint16_t *tab;           // [n_wg][n_tiles + 1]  start of every tile's run inside the chunk; [n_tiles] = count
  uint32_t *tile_total;    // [n_tiles]  keys per tile over all chunks (zero on entry)
  const gv_lshape_pose *rect_poses;   // optional rider: n_rect_poses base-frame poses -> rects_out (one extra workgroup)
  int32_t n_rect_poses;
  Rect *rects_out;
  unsigned long long *dbg; // diagnostic build: 16 clock stamps per workgroup (null in production)
  unsigned long long *tl;  // diagnostic build: {first workgroup in, last workgroup out} of this launch (GV_TIMELINE)
};
constexpr size_t kBinBBoxLdsMax = 24 * 1024;   // LDS the partition kernel may spend on the bbox-test tables
uint32_t bin_chunk_for(size_t n);
bool bin_bbox_fits(int nb, const BBoxTest &bt);
void launch_bin_partition(const BinArgs &a, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, bool any_order = false);

struct BinTileArgs {
  int32_t nx, ny, tiles_x, tiles_y, n_tiles;
  uint32_t n_wg, chunk;
  const uint16_t *keys;
  const uint16_t *tab;
  const uint32_t *tile_total;   // this frame's totals
  uint32_t *tile_total_next;    // cleared here for the next frame's partition
  uint32_t *done;               // [n_tiles] arrival tickets of shared tiles (zero between frames)
  uint32_t *scratch;            // [max_slots][kBinSplitMax][kBinTileCells + 512] partial tiles
  uint32_t split_keys, max_slots;
  int32_t *hits;                // G int32 (every cell written) or null
  uint32_t *hitN, *clipN, *hitT, *clipT;   // end bitmaps (layout: gv_raysector.hip), every word written
  uint32_t *freeN, *freeT;                 // free-cell bitmaps of the same buffer set: zeroed here (or null)
  int32_t nxw, nyw, nx_pad, ny_pad;
  unsigned long long *dbg;                 // diagnostic build: 16 clock stamps per workgroup (null in production)
  unsigned long long *tl;                  // diagnostic build: launch begin / end (GV_TIMELINE)
};
// n_helpers >= n / split_keys extra workgroups serve the shares 1.. of crowded tiles
void launch_bin_tiles(const BinTileArgs &a, uint32_t n_helpers, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// ---- sector/gather ray stage + tile grid pass (gv_raysector.hip) ----
struct SectorArgs {
  GridParams g;
  RayOrigin org;
  uint8_t log2s_oct[8];   // sectors of octant o = 1 << log2s_oct[o]: short wedges get fewer, fatter sectors
  uint16_t wg_base[9];    // first workgroup of the k-th octant in dispatch order (oct_perm); [8] = grid size
  uint16_t rev_oct[8];    // octant o, bit c: the threads hold row c of the wedge's columns (512 per row) in descending order
  int32_t cap;            // ends per LDS chunk (>= 2048)
  int32_t log2m;          // slope buckets per sector = 1 << log2m (<= 9)
  int32_t marks_words;    // >= max(nx, ny) + 1
  uint32_t oct_perm;      // dispatch order of the octants, 3 bits each (longest wedge first); 0 = identity off
  int32_t reorder;        // 1: longest octants and the rational-gap sectors (0, S/2-1, S/2, S-1) first
  const uint32_t *hitN, *clipN, *hitT, *clipT;
  int32_t nxw, nyw, nx_pad, ny_pad;
  uint32_t *freeN;        // free-cell bitmap, bits along x (written for y-major octants); same layout as hitN
  uint32_t *freeT;        // free-cell bitmap, bits along y (written for x-major octants); same layout as hitT
  unsigned long long *stats;
  int32_t wg_first, wg_stride;   // this launch runs workgroups wg_first, wg_first + wg_stride, ... of the dispatch order
  int32_t n_helpers;      // 0 or 16: second workgroups for the first / last sector of every octant, first in the dispatch order
  uint32_t march_limit;   // most ray cells beyond the threshold column a sector will march (else: exact per cell)
  int32_t flat_k;         // cost ratio exact-cell evaluation : marched cell for the choice beyond T (0: always march when possible)
  uint32_t flat_direct;   // a tail of at most this many cells is evaluated exactly without looking at the long rays at all
  int32_t ablate;         // timing experiments only: 2 skip gather, 4 skip flush, 8/16/32 early exits
  unsigned long long *dbg; // diagnostic phase stamps, 16 per workgroup (null in production)
  unsigned long long *tl;  // diagnostic build: launch begin / end (GV_TIMELINE)
};
size_t sector_lds_bytes(int cap, int marks_words, int log2m);
bool launch_ray_sectors(const SectorArgs &a, hipStream_t s, hipEvent_t done = nullptr, hipEvent_t t0 = nullptr);

struct FinalizeTileArgs {
  GridParams g;
  float *log_odds, *occupancy;
  int8_t *occ_i8;
  const Rect *rects;
  int32_t n_rects;
  const uint32_t *hitN;            // hit bitmap (bits along x)
  const uint32_t *freeN, *freeT;   // free-cell bitmaps of the ray stage (N | T)
  int32_t nx_pad, ny_pad;
  bool counts;            // apply the hit/miss rule
  int32_t y_begin, y_end; // rows to finalise ([0, ny) on one GPU)
  unsigned long long *tl; // diagnostic build: launch begin / end (GV_TIMELINE)
};
bool launch_finalize_tiles(const FinalizeTileArgs &a, hipStream_t s, hipEvent_t done = nullptr, hipEvent_t t0 = nullptr);
void launch_miss_to_i32(const uint32_t *freeN, const uint32_t *freeT, int nx, int ny, int nx_pad, int ny_pad,
                        int32_t *out, hipStream_t s);

// ---- frame sharded by points over several GPUs: OR-exchange helpers (gv_shard.hip) ----
void launch_or_slices(const uint32_t *src, uint32_t *dst, size_t count_words, int world, hipStream_t s);
size_t free_band_chunk_words(int nxw, int nx_pad, int ny_pad, int world);
void launch_pack_free_bands(const uint32_t *fN, const uint32_t *fT, int nxw, int nx_pad, int ny_pad, int world,
                            size_t chunk, uint32_t *out, hipStream_t s);
void launch_unpack_free_band(const uint32_t *in, int world, size_t chunk, int rank, int nxw, int nx_pad, int ny_pad,
                             uint32_t *fN, uint32_t *fT, hipStream_t s);
// rows [y0, y1) rank `rank` of `world` finalises: whole 64-row blocks of the padded grid, clipped to ny
void shard_band_rows(int rank, int world, int ny, int ny_pad, int32_t &y0, int32_t &y1);

// ---- kNN depth + radius outlier counts (gv_knn_pca.hip) ----
struct Cand2 {
  float d2;
  uint32_t idx;
};
void launch_project_uvd(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m,
                        const CamK &cam, float *pu, float *pv, float *pd, hipStream_t s);
size_t knn_partial_entries(int nb, int k);   // Cand2 entries of the stage-1 lists

// Completion of a synchronous call without a copy command or a runtime wait: the call's last kernel stores its
// (small) results straight into pinned, device-mapped host memory, every workgroup then takes a ticket, and the
// one whose ticket comes last publishes the call's sequence number in the block's first word -- the host spins on
// that word (tools/micro/call_latency.hip: 8 us instead of 31 for copy + hipStreamSynchronize).  flag == nullptr:
// plain device outputs, nothing published.
struct CallDone {
  unsigned *ticket = nullptr;   // device memory, zero between calls
  unsigned *flag = nullptr;     // host-mapped
  unsigned seq = 0;
};
#if defined(__HIPCC__)
// by ONE lane of every workgroup of the grid, after that lane's result stores (other lanes that stored: a system-scope
// fence of their own, then a barrier, first).  The system-scope fences stay: the results go to the HOST, and stores
// from different CUs reach it on different paths -- with only "my stores are acknowledged" (s_waitcnt vmcnt(0)) before
// the ticket the host saw the flag ahead of another workgroup's payload in 4 calls of 10 (tools/result_block_soak.py,
// profiles/r04/ticket_fence_ab.txt).  Tickets BETWEEN WORKGROUPS OF ONE KERNEL over device memory need no fence when
// the data goes through agent-scope atomics (k_ransac_moments, k_pca_extent, k_cell_scan).
__device__ __forceinline__ void call_done(const CallDone &d, unsigned n_wg)
{
  if (!d.flag) return;
  __threadfence_system();
  const unsigned t = __hip_atomic_fetch_add(d.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (t == n_wg - 1u) {
    __hip_atomic_store(d.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#endif

// depths / knn_d2 may be host-mapped (then done.flag is set)
void launch_knn(const float *pu, const float *pv, const float *pd, uint32_t n, const gv_bbox *bboxes, int nb, int k,
                Cand2 *partial, float *depths, float *knn_d2, const CallDone &done, hipStream_t s);

// ---- device-resident RANSAC ground plane + per-bbox clouds / radius filter / PCA (gv_cloudops.hip) ----
struct RansacState {
  float4 plane;             // best sampled plane
  float4 refined;           // least-squares plane of its inliers (= plane when fewer than 3)
  unsigned long long m;         // inliers of `plane`
  unsigned long long n_inliers; // inliers of `refined` (the mask / the ground points the classify pass dropped)
  unsigned best_count;          // 0: could not estimate a planar model
  unsigned ticket;              // arrival counter of the moments pass (zero between calls)
};
size_t ransac_scratch_doubles(size_t n);
// The inlier counters of the hypotheses are kept in kRansacCountSlices copies (workgroup w adds to copy w % slices,
// the selection sums them): every workgroup ends with one atomic per hypothesis, and 245 workgroups on the same
// 50 words queue up in two L2 channels otherwise.  counts holds kRansacCountSlices * iters words.
constexpr int kRansacCountSlices = 16;
// hypotheses, inlier counts of all of them, selection + refinement on stream s; *st stays on the device.
// counts[kRansacCountSlices * iters] must be zero on entry (the pass leaves it zero); thr_f = smallest float >= the fp64 threshold
void launch_ransac_plane(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, float thr_f, int iters,
                         unsigned long long seed, float4 *planes, unsigned *counts, double *scratch, RansacState *st,
                         hipStream_t s);
// mask[n] of the refined plane's inliers + st->n_inliers
// st_copy (optional, may be host-mapped): the final *st, stored by the workgroup that finishes last
void launch_ransac_mask(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, float thr_f,
                        RansacState *st, uint8_t *mask, RansacState *st_copy, const CallDone &done, hipStream_t s);
// one selected point of the radius filter, in bucket order
struct CellNode {
  float x, y, z;    // camera frame
  int32_t id;       // bbox
};
static_assert(sizeof(CellNode) == 16, "one node = one 16-byte access");
// extractCloudPerBBox + RadiusOutlierRemoval: ids[n] (ground points of st->refined dropped when use_plane), the
// selected points in bucket order (sorted, their number at pre[n_buckets]), keep[t] = 1 where sorted point t survives
// the filter, and the kept points' coordinate sums in acc (pca_acc_words(nb) 64-bit words, zero on entry).
// cell_cnt[n_buckets] must be zero on entry (left zero), pre has n_buckets + 1 entries, blk_off n_buckets / 4096 + 1,
// *ticket zero on entry; sorted holds n nodes, keep n bytes, ticket_of n words (a selected point's slot inside its
// bucket).  n_buckets: a power of two >= 4096
void launch_radius_filter(const float *x, const float *y, const float *z, uint32_t n, const Mat34f &m_cam, const CamK &cam,
                          const BBoxTest &bt, int nb, bool use_plane, float thr_f, RansacState *st, int16_t *ids,
                          uint32_t *cell_cnt, uint32_t *pre, uint32_t *blk_off, unsigned *ticket, CellNode *sorted, uint8_t *keep,
                          uint32_t *ticket_of, long long *acc, uint32_t n_buckets, float r2f, int min_pts, hipStream_t s);
size_t pca_acc_words(int nb);   // 64-bit words of acc
size_t pca_ext_words(int nb);   // 32-bit words of ext
// centroid + PCA rectangle of every bbox's kept points (bboxPoseEstimation :156-181, computePCABoundingBox :187-247)
// from order-independent integer sums: covariance pass, extents pass, poses by the last workgroup.  acc / ext
// (pca_acc_words / pca_ext_words) / *ticket are zero on entry and left zero.  n_sel: device address of the number of
// selected points (blk_off[n_buckets / 4096] of the bucket scan).  st_copy (optional): *st is copied there (one read-back
// block for poses, flags and state); poses_dev (optional, device memory): a second copy of the camera-frame poses with
// length = -1 where valid[b] == 0
void launch_pca_rect(const CellNode *sorted, const uint32_t *n_sel, uint32_t n, const uint8_t *keep, long long *acc, unsigned *ext,
                     unsigned *ticket, int nb, const RansacState *st, bool use_plane, gv_lshape_pose *poses, uint8_t *valid,
                     RansacState *st_copy, const CallDone &done, hipStream_t s, gv_lshape_pose *poses_dev = nullptr);

}  // namespace gv
