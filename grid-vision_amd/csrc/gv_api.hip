// gv_api.hip -- C ABI (include/gridvision_hip.h) over the gfx950 kernels.
// One gv_context = one device + one stream + one resident grid.  No exception
// leaves this file; every entry point returns a gv_status.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "gv_host_math.hpp"
#include "gv_kernels.hpp"

using namespace gv;

struct gv_context {
  static constexpr int kSetsMax = 4;
  int device = 0;
  hipStream_t stream = nullptr;
  // frame pipelining: points/bitmaps of frame f+1 (stream) overlap sectors/grid pass of frame f (stream2)
  hipStream_t stream2 = nullptr, stream3 = nullptr;   // B: sector ray stage, C: grid pass
  hipStream_t stream4 = nullptr;                      // D: bbox test of the points pass when split off (GV_SPLIT_POINTS=1)
  bool split_points = false;
  bool rects_on_c = false;                            // GV_RECTS_ON_C=1: rectangle kernels on the grid-pass stream (lidar-like cloud -6 %, uniform +10 %)
  hipStream_t stream2b = nullptr;                     // B': sector kernels of odd frames, so that one frame's sector
                                                      // kernel fills the CUs its predecessor's tail leaves idle
  int sector_streams = 1;                             // GV_SECTOR_STREAMS=2 alternates two sector streams: measured slower (75.9 vs 69.0 us)
  unsigned long long *x_stats[kSetsMax]{};           // per-set (rays, visits) slots of the sector kernel
  int last_stats_set = 0;
  static constexpr int kSets = kSetsMax;                     // buffer sets the pipelined frames rotate through (n_sets in use)
  hipEvent_t ev_build[kSets]{}, ev_sec[kSets]{}, ev_fin[kSets]{};
  std::vector<hipEvent_t> *trace = nullptr;         // diagnostic: timing events around every pipelined kernel (gv_debug_pipeline_trace)
  int n_sets = 3;                                     // GV_PIPE_SETS (2..4): how far stream A may run ahead
  // sets 1..: bitmaps, rectangles and miss grids of the frames in flight (set 0 = the primary buffers)
  uint32_t *x_hitN[kSets]{}, *x_clipN[kSets]{}, *x_hitT[kSets]{}, *x_clipT[kSets]{};
  Rect *x_rects[kSets]{};
  uint8_t *x_miss[kSets]{}, *x_missT[kSets]{};
  bool three_streams = true;                         // GV_PIPELINE=2: grid pass on stream B (two streams)
  uint64_t frame_no = 0;
  int since_drain = 0;       // pipelined frames enqueued since both streams were last idle
  bool pipe_busy = false;
  bool no_pipeline = false;  // GV_PIPELINE=0
  GridParams g{};
  gv_cam_params cam{};
  CamK camk{};
  double K[9]{}, Kinv[9]{};

  bool has_cl = false, has_bc = false, has_bl = false;
  gv_transform tf_cl{}, tf_bc{}, tf_bl{};
  Mat34f m_cam{}, m_base{};
  Xform64 x_bc{};
  RayOrigin org{};

  // grid state (resident across frames)
  float *log_odds = nullptr, *occupancy = nullptr;
  int8_t *occ_i8 = nullptr;
  // per-frame count grids
  int32_t *hits = nullptr;
  uint8_t *miss = nullptr, *clip_end = nullptr, *hit8 = nullptr;
  uint32_t *ray_list = nullptr;
  uint32_t *ray_count = nullptr;            // [0] = number of list entries
  unsigned long long *ray_stats = nullptr;  // [0] rays, [1] visits
  int32_t *scratch_i32 = nullptr;           // G ints (miss read-back)
  // sector/gather ray stage
  uint8_t *missT = nullptr;                 // G bytes, [x][y]
  uint32_t *hitN = nullptr, *clipN = nullptr, *hitT = nullptr, *clipT = nullptr;
  int32_t nxw = 0, nyw = 0, nx_pad = 0, ny_pad = 0;
  bool tile_path = false;                   // nx % 4 == 0 and the grid fits the packed (a,b) fields
  bool force_simple = false;                // GV_RAY_IMPL=simple
  int env_reorder = 1;                      // GV_SECTOR_REORDER=0: workgroups in natural (octant, sector) order
  int32_t last_log2s = 0, last_cap = 0;
  unsigned long long *d_dbg = nullptr;      // GV_SECTOR_DBG=1: phase stamps of the sector kernel
  size_t stat_slots = 1;                    // ray_stats slots written by the last frame
  int32_t env_log2s_oct[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // GV_LOG2S_OCT="a,b,..." per octant index (experiments)
  int32_t env_flat_k = 8;                   // GV_FLAT_K: exact-cell : marched-cell cost ratio (tools/flatk.sh: 8 is best on both clouds; 0 = always march)
  int32_t env_log2s = 0, env_cap = 0, env_ablate = 0, env_log2m = 0;   // GV_LOG2S / GV_CAP / GV_ABLATE (experiments)

  // resident cloud
  float *cx = nullptr, *cy = nullptr, *cz = nullptr;
  size_t n = 0, cap = 0;
  float *tx = nullptr, *ty = nullptr, *tz = nullptr;   // transformed copy (A1 read-back)
  size_t tcap = 0;
  uint8_t *raw = nullptr;
  size_t raw_cap = 0;
  int32_t *cell_idx = nullptr, *bbox_id = nullptr;
  size_t idx_cap = 0;

  // detections of the current frame
  gv_bbox *d_bboxes = nullptr;
  gv_lshape_pose *d_poses = nullptr;
  Rect *d_rects = nullptr;
  float *d_orient = nullptr, *d_conf = nullptr, *d_dims = nullptr;
  VisionOut *d_vout = nullptr;
  double *d_pts = nullptr;
  // kNN depth / PCA pose scratch
  Cand2 *knn_partial = nullptr; size_t knn_partial_cap = 0;
  float *d_depths = nullptr, *d_knn_d2 = nullptr; size_t knn_out_cap = 0;
  int32_t *d_idx = nullptr, *d_segof = nullptr, *d_segstart = nullptr; size_t seg_cap = 0, segstart_cap = 0;
  float *gx = nullptr, *gy = nullptr, *gz = nullptr; uint8_t *d_keep = nullptr; size_t gcap = 0;
  float4 *d_planes = nullptr; unsigned *d_plane_counts = nullptr; size_t planes_cap = 0;
  uint8_t *d_ground = nullptr; size_t ground_cap = 0;
  std::vector<uint8_t> ground_mask;   // last gv_segment_ground_plane result (host copy)
  float4 *d_bbox_f = nullptr;                // float thresholds of the bbox test
  unsigned long long *d_tile_mask = nullptr; // candidate masks per 16x16-pixel tile
  size_t tile_mask_cap = 0;
  int32_t tiles_x = 0, tiles_y = 0, mask_words = 1;
  int32_t det_cap = 0;
  uint32_t frame_flags = 0;
  int32_t nb = 0, n_poses = 0;

  bool counts_dirty = false;   // hits/miss/clip_end hold a kept frame
  bool frame_counts = false;   // the frame in flight used int32 hit counts (else byte flags)
  bool force_counts = false;   // GV_HIT_COUNTS=1: always count (A/B measurement of the atomics path)
  bool have_counts = false, have_cell_idx = false, have_bbox_id = false;

  // multi-GPU (one large frame sharded by points)
  ncclComm_t comm = nullptr;
  int32_t rank = 0, world = 1;

  hipEvent_t ev[kNumStages + 1]{};
  std::string err;
};

namespace {

constexpr size_t kMaxStatSlots = 8u << 12;   // one (rays, visits) slot per sector workgroup

#define GV_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      char buf_[256];                                                                         \
      std::snprintf(buf_, sizeof(buf_), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      h->err = buf_;                                                                          \
      return GV_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define GV_TRY try {
#define GV_CATCH                               \
  }                                            \
  catch (const std::bad_alloc &) {             \
    if (h) h->err = "host allocation failed";  \
    return GV_ERR_HIP;                         \
  }                                            \
  catch (...) {                                \
    if (h) h->err = "unexpected exception";    \
    return GV_ERR_HIP;                         \
  }

template <typename T>
int grow(gv_context *h, T *&p, size_t &cap, size_t need)
{
  if (need <= cap) return GV_OK;
  if (p) GV_HIP(hipFree(p));
  p = nullptr;
  cap = 0;
  GV_HIP(hipMalloc(reinterpret_cast<void **>(&p), need * sizeof(T)));
  cap = need;
  return GV_OK;
}

int ensure_cloud(gv_context *h, size_t n)
{
  if (n > h->cap) {
    const size_t want = n + n / 8 + 1024;
    for (float **p : {&h->cx, &h->cy, &h->cz}) {
      if (*p) GV_HIP(hipFree(*p));
      *p = nullptr;
    }
    h->cap = 0;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->cx), want * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->cy), want * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->cz), want * sizeof(float)));
    h->cap = want;
  }
  if (n > h->idx_cap) {
    const size_t want = n + n / 8 + 1024;
    for (int32_t **p : {&h->cell_idx, &h->bbox_id}) {
      if (*p) GV_HIP(hipFree(*p));
      *p = nullptr;
    }
    h->idx_cap = 0;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->cell_idx), want * sizeof(int32_t)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->bbox_id), want * sizeof(int32_t)));
    h->idx_cap = want;
  }
  return GV_OK;
}

int ensure_det(gv_context *h, int32_t n)
{
  if (n <= h->det_cap) return GV_OK;
  const int32_t want = std::max(n, 64);
  auto re = [&](auto *&p, size_t bytes) -> int {
    if (p) GV_HIP(hipFree(p));
    p = nullptr;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&p), bytes));
    return GV_OK;
  };
  int rc;
  if ((rc = re(h->d_bboxes, (size_t)want * sizeof(gv_bbox)))) return rc;
  if ((rc = re(h->d_poses, (size_t)want * sizeof(gv_lshape_pose)))) return rc;
  if ((rc = re(h->d_rects, (size_t)want * sizeof(Rect)))) return rc;
  for (int k = 1; k < gv_context::kSets; ++k)
    if ((rc = re(h->x_rects[k], (size_t)want * sizeof(Rect)))) return rc;
  h->x_rects[0] = h->d_rects;
  if ((rc = re(h->d_orient, (size_t)want * 4 * sizeof(float)))) return rc;
  if ((rc = re(h->d_conf, (size_t)want * 2 * sizeof(float)))) return rc;
  if ((rc = re(h->d_dims, (size_t)want * 3 * sizeof(float)))) return rc;
  if ((rc = re(h->d_vout, (size_t)want * sizeof(VisionOut)))) return rc;
  if ((rc = re(h->d_pts, (size_t)want * 3 * sizeof(double)))) return rc;
  if ((rc = re(h->d_bbox_f, (size_t)want * sizeof(float4)))) return rc;
  h->det_cap = want;
  return GV_OK;
}

int set_device_only(gv_context *h)
{
  GV_HIP(hipSetDevice(h->device));
  return GV_OK;
}

// Every entry point except the pipelined gv_frame_enqueue starts from two idle streams.
int use_device(gv_context *h)
{
  GV_HIP(hipSetDevice(h->device));
  if (h->pipe_busy) {
    GV_HIP(hipStreamSynchronize(h->stream));
    GV_HIP(hipStreamSynchronize(h->stream2));
    GV_HIP(hipStreamSynchronize(h->stream2b));
    GV_HIP(hipStreamSynchronize(h->stream3));
    GV_HIP(hipStreamSynchronize(h->stream4));
    h->pipe_busy = false;
    h->since_drain = 0;
  }
  return GV_OK;
}

void refresh_origin(gv_context *h)
{
  // [EXTENSION] sensor origin = image of (0,0,0) under base<-lidar = fp32 translation column
  h->org.ox = (double)h->m_base.m[3];
  h->org.oy = (double)h->m_base.m[7];
  int ix = 0, iy = 0;
  h->org.valid = host::get_index(h->g, h->org.ox, h->org.oy, ix, iy) ? 1 : 0;
  h->org.cx = ix;
  h->org.cy = iy;
}

int clear_counts(gv_context *h)
{
  const size_t G = (size_t)h->g.G;
  GV_HIP(hipMemsetAsync(h->hits, 0, G * sizeof(int32_t), h->stream));
  GV_HIP(hipMemsetAsync(h->miss, 0, G, h->stream));
  GV_HIP(hipMemsetAsync(h->clip_end, 0, G, h->stream));
  GV_HIP(hipMemsetAsync(h->hit8, 0, G, h->stream));
  GV_HIP(hipMemsetAsync(h->missT, 0, G, h->stream));
  h->counts_dirty = false;
  return GV_OK;
}

// plain grid update (A7 / A8 / A10): rectangles already in d_rects
int enqueue_plain_update(gv_context *h, int32_t n_rects)
{
  if (h->tile_path) {
    FinalizeTileArgs t{};
    t.g = h->g;
    t.log_odds = h->log_odds;
    t.occupancy = h->occupancy;
    t.occ_i8 = h->occ_i8;
    t.rects = h->d_rects;
    t.n_rects = n_rects;
    t.hitN = h->hitN;
    t.nxw = h->nxw;
    t.ny_pad = h->ny_pad;
    t.missN = h->miss;
    t.missT = h->missT;
    t.counts = false;
    t.zero = false;
    t.use_missT = false;
    t.y_begin = 0;
    t.y_end = h->g.ny;
    launch_finalize_tiles(t, h->stream);
    GV_HIP(hipGetLastError());
    return GV_OK;
  }
  FinalizeArgs f{};
  f.g = h->g;
  f.log_odds = h->log_odds;
  f.occupancy = h->occupancy;
  f.occ_i8 = h->occ_i8;
  f.rects = h->d_rects;
  f.n_rects = n_rects;
  f.hits = nullptr;
  f.miss = nullptr;
  f.clip_end = nullptr;
  f.zero_counts = false;
  f.cell_begin = 0;
  f.cell_end = h->g.G;
  launch_finalize(f, h->stream);
  GV_HIP(hipGetLastError());
  return GV_OK;
}

int sharded_tail(gv_context *h, int32_t n_rects);

// sector-kernel launch parameters for the resident cloud and grid (set-0 bitmaps by default)
int fill_sector_args(gv_context *h, SectorArgs &sa)
{
  sa.g = h->g;
  sa.org = h->org;
  const int imax = std::max(std::max(h->org.cx, h->g.nx - 1 - h->org.cx), std::max(h->org.cy, h->g.ny - 1 - h->org.cy));
  // octant o: xmaj = bit 2, smaj = bit 1; wedge length = distance to the map edge along the major axis
  int len[8], ord[8];
  for (int o = 0; o < 8; ++o) {
    const bool xmaj = (o >> 2) & 1, pos = (o >> 1) & 1;
    len[o] = xmaj ? (pos ? h->g.nx - 1 - h->org.cx : h->org.cx) : (pos ? h->g.ny - 1 - h->org.cy : h->org.cy);
    ord[o] = o;
  }
  // Sectors per octant: the far end of a wedge about 16 cells wide (len <= 16*S; the kernel needs
  // <= 32) and an estimated <= 12000 ends per sector (the estimate runs ~2x high; above one LDS chunk
  // of 4096 ends a wedge is processed in row groups, which measured better on config 5 -- 10 M points,
  // 160 vs 390 us -- than four times as many, thinner wedges).  Measured on
  // config 3 (tools/sweep_oct.sh, tools/sweep_sectors.sh): the kernel is bound by per-workgroup
  // latency chains, so fewer, fatter wedges win as long as those two hold, and an octant whose wedge
  // is short (origin near that map edge) gets proportionally fewer sectors: 128/64/32 sectors for
  // wedges of 1660/1000/340 columns instead of 128 everywhere does the same frame in 576 instead of
  // 1024 workgroups, 84 -> 76 us pipelined.  Wider wedges (S = 16 for 340 columns) lose again.
  const double dens = std::min((double)h->n, (double)h->g.G) / (double)h->g.G;
  double est_max = 0.0;
  int log2s_max = 0;
  for (int o = 0; o < 8; ++o) {
    int l2 = 3;   // the gap-sector logic wants S >= 8
    while ((16 << l2) < len[o]) ++l2;
    double est = 1.5 * dens * (double)len[o] * (double)len[o] / (double)(2 << l2);
    while (est > 12000.0 && l2 < 12) {
      ++l2;
      est *= 0.5;
    }
    if (h->env_log2s > 0) { l2 = h->env_log2s; est = 1.5 * dens * (double)len[o] * (double)len[o] / (double)(2 << l2); }
    if (h->env_log2s_oct[o] > 0) { l2 = h->env_log2s_oct[o]; est = 1.5 * dens * (double)len[o] * (double)len[o] / (double)(2 << l2); }
    sa.log2s_oct[o] = (uint8_t)l2;
    est_max = std::max(est_max, est);
    log2s_max = std::max(log2s_max, l2);
  }
  sa.cap = h->env_cap > 0 ? std::max(2048, h->env_cap) : ((est_max <= 1700.0 && h->env_log2s <= 0) ? 2048 : 4096);
  sa.ablate = h->env_ablate;
  sa.flat_k = h->env_flat_k;
  sa.dbg = h->d_dbg;
  sa.log2m = h->env_log2m > 0 ? h->env_log2m : 9;
  sa.marks_words = (imax + 3) & ~1;   // one word per wedge column, 0..imax
  std::stable_sort(ord, ord + 8, [&](int a, int b) { return len[a] > len[b]; });
  sa.oct_perm = 0;
  sa.reorder = h->env_reorder;
  uint32_t base = 0;
  for (int k = 0; k < 8; ++k) {
    sa.oct_perm |= (uint32_t)ord[k] << (3 * k);
    sa.wg_base[k] = (uint16_t)base;
    base += 1u << sa.log2s_oct[sa.reorder ? ord[k] : k];
  }
  if (base > kMaxStatSlots || base > 65535u) { h->err = "too many sector workgroups"; return GV_ERR_BAD_ARG; }
  sa.wg_base[8] = (uint16_t)base;
  sa.hitN = h->hitN; sa.clipN = h->clipN; sa.hitT = h->hitT; sa.clipT = h->clipT;
  sa.nxw = h->nxw; sa.nyw = h->nyw; sa.nx_pad = h->nx_pad; sa.ny_pad = h->ny_pad;
  sa.missN = h->miss;
  sa.missT = h->missT;
  sa.stats = h->ray_stats;
  h->last_stats_set = 0;
  h->last_log2s = log2s_max;
  h->last_cap = sa.cap;
  h->stat_slots = (size_t)sa.wg_base[8];
  return GV_OK;
}


int enqueue_frame(gv_context *h, bool stage_events, bool sharded = false)
{
  const uint32_t fl = h->frame_flags;
  const bool do_bin = fl & GV_FRAME_BIN, do_ray = fl & GV_FRAME_RAYMARCH, do_bbox = fl & GV_FRAME_BBOX_TEST;
  const bool keep_cell = fl & GV_FRAME_KEEP_CELL_IDX, keep_counts = fl & GV_FRAME_KEEP_COUNTS;
  const bool vision = fl & GV_FRAME_VISION_ORIENT;
  if (do_ray && !do_bin) return GV_ERR_BAD_ARG;
  if (sharded && (!do_bin || !h->tile_path || h->force_simple || !h->comm)) return GV_ERR_STATE;
  if (do_bin && !h->has_bl) return GV_ERR_TF;
  if (do_bbox && !h->has_cl) return GV_ERR_TF;
  if (vision && !h->has_bc) return GV_ERR_TF;
  if (h->counts_dirty) { int rc = clear_counts(h); if (rc) return rc; }
  if (stage_events) GV_HIP(hipEventRecord(h->ev[0], h->stream));
  // int32 hit counts only where someone reads them (KEEP_COUNTS getter, generic ray path);
  // otherwise points mark byte flags (no atomics)
  const bool sectors = h->tile_path && !h->force_simple;
  const bool counts = keep_counts || !sectors || h->force_counts;
  h->frame_counts = counts;

  // --- detections -> rectangles
  int32_t n_rects = 0;
  if (vision && h->nb > 0) {
    launch_vision(h->d_orient, h->d_conf, h->d_dims, h->d_bboxes, h->nb, h->cam, h->d_vout, h->d_poses, h->stream);
    launch_rects_from_poses(h->d_poses, h->nb, h->g, true, h->x_bc, h->d_rects, h->stream);
    n_rects = h->nb;
  } else if (h->n_poses > 0) {
    launch_rects_from_poses(h->d_poses, h->n_poses, h->g, false, h->x_bc, h->d_rects, h->stream);
    n_rects = h->n_poses;
  }
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageDetections + 1], h->stream));

  // --- points pass
  if (do_bin || do_bbox) {
    PointsArgs a{};
    a.x = h->cx; a.y = h->cy; a.z = h->cz;
    a.n = (uint32_t)h->n;
    a.g = h->g;
    a.m_base = h->m_base;
    a.m_cam = h->m_cam;
    a.cam = h->camk;
    a.org = h->org;
    a.bboxes = h->d_bboxes;
    a.nb = h->nb;
    a.bbox_f = h->d_bbox_f;
    a.tile_mask = h->d_tile_mask;
    a.tiles_x = h->tiles_x; a.tiles_y = h->tiles_y; a.mask_words = h->mask_words;
    a.hits = h->hits;
    a.hit8 = h->hit8;
    a.clip_end = h->clip_end;
    a.cell_idx = keep_cell ? h->cell_idx : nullptr;
    a.bbox_id = h->bbox_id;
    a.do_bin = do_bin; a.do_ray = do_ray; a.do_bbox = do_bbox;
    a.counts = counts;
    launch_points(a, h->stream);
  }
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStagePoints + 1], h->stream));

  if (sectors) {
    // --- end bitmaps (both orientations), sector gather, tile grid pass
    if (do_bin) {
      BitmapArgs b{};
      b.nx = h->g.nx; b.ny = h->g.ny;
      b.hits = h->hits; b.clip_end = h->clip_end;
      b.hit8 = counts ? nullptr : h->hit8;
      b.hitN = h->hitN; b.clipN = h->clipN; b.hitT = h->hitT; b.clipT = h->clipT;
      b.nxw = h->nxw; b.nyw = h->nyw; b.nx_pad = h->nx_pad; b.ny_pad = h->ny_pad;
      b.zero_hits = !keep_counts && !sharded;   // the sharded path reduces the counts first
      launch_build_bitmaps(b, h->stream);
    }
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayCompact + 1], h->stream));
    if (do_ray && h->org.valid) {
      SectorArgs sa{};
      { int rc2 = fill_sector_args(h, sa); if (rc2) return rc2; }
      launch_ray_sectors(sa, h->stream);
    }
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayMarch + 1], h->stream));
    if (sharded) return sharded_tail(h, n_rects);
    FinalizeTileArgs t{};
    t.g = h->g;
    t.log_odds = h->log_odds;
    t.occupancy = h->occupancy;
    t.occ_i8 = h->occ_i8;
    t.rects = h->d_rects;
    t.n_rects = n_rects;
    t.hitN = h->hitN;
    t.nxw = h->nxw;
    t.ny_pad = h->ny_pad;
    t.missN = h->miss;
    t.missT = h->missT;
    t.counts = do_bin;
    t.zero = do_bin && !keep_counts;
    t.use_missT = true;
    t.y_begin = 0;
    t.y_end = h->g.ny;
    launch_finalize_tiles(t, h->stream);
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageFinalize + 1], h->stream));
    GV_HIP(hipGetLastError());
  } else {
  // --- ray march (generic path: any grid shape)
  if (do_ray && h->org.valid) {
    GV_HIP(hipMemsetAsync(h->ray_count, 0, sizeof(uint32_t), h->stream));
    GV_HIP(hipMemsetAsync(h->ray_stats, 0, 2 * sizeof(unsigned long long), h->stream));
    h->stat_slots = 1;
    h->last_stats_set = 0;
    launch_ray_compact(h->hits, h->clip_end, h->g, h->ray_list, h->ray_count, h->stream);
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayCompact + 1], h->stream));
    launch_ray_march(h->ray_list, h->ray_count, h->g, h->org, h->miss, h->ray_stats, h->stream);
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayMarch + 1], h->stream));
  } else if (stage_events) {
    GV_HIP(hipEventRecord(h->ev[kStageRayCompact + 1], h->stream));
    GV_HIP(hipEventRecord(h->ev[kStageRayMarch + 1], h->stream));
  }

  // --- one grid pass
  FinalizeArgs f{};
  f.g = h->g;
  f.log_odds = h->log_odds;
  f.occupancy = h->occupancy;
  f.occ_i8 = h->occ_i8;
  f.rects = h->d_rects;
  f.n_rects = n_rects;
  f.hits = do_bin ? h->hits : nullptr;
  f.miss = h->miss;
  f.clip_end = h->clip_end;
  f.zero_counts = do_bin && !keep_counts;
  f.cell_begin = 0;
  f.cell_end = h->g.G;
  launch_finalize(f, h->stream);
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageFinalize + 1], h->stream));
  GV_HIP(hipGetLastError());
  }

  h->counts_dirty = do_bin && keep_counts;
  h->have_counts = do_bin && keep_counts;
  h->have_cell_idx = do_bin && keep_cell;
  h->have_bbox_id = do_bbox;
  return GV_OK;
}

// Pipelined frame (production path): stream A = rectangles, points pass, end bitmaps of frame f;
// stream B = sector ray stage + grid pass of frame f.  A may run up to two frames ahead of B:
// the end bitmaps and rectangles are double buffered (set f & 1), everything else is either
// private to one stream (hits/clip_end: A; miss grids, grid layers: B) or ordered by events.
int enqueue_frame_pipelined(gv_context *h)
{
  auto mark = [&](hipStream_t st) {   // diagnostic build-up of a device timeline; null in production
    if (!h->trace) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    h->trace->push_back(e);
  };

  const uint32_t fl = h->frame_flags;
  const bool do_bin = fl & GV_FRAME_BIN, do_ray = fl & GV_FRAME_RAYMARCH, do_bbox = fl & GV_FRAME_BBOX_TEST;
  const bool vision = fl & GV_FRAME_VISION_ORIENT;
  if (do_ray && !do_bin) return GV_ERR_BAD_ARG;
  if (do_bin && !h->has_bl) return GV_ERR_TF;
  if (do_bbox && !h->has_cl) return GV_ERR_TF;
  if (vision && !h->has_bc) return GV_ERR_TF;
  if (h->counts_dirty) { int rc = clear_counts(h); if (rc) return rc; }
  const int p = (int)(h->frame_no % (unsigned)h->n_sets);
  hipStream_t sA = h->stream, sB = (h->sector_streams > 1 && (h->frame_no & 1)) ? h->stream2b : h->stream2;
  uint32_t *hitN = h->x_hitN[p], *clipN = h->x_clipN[p], *hitT = h->x_hitT[p], *clipT = h->x_clipT[p];
  Rect *rects = h->x_rects[p];
  if (h->since_drain >= h->n_sets) GV_HIP(hipStreamWaitEvent(sA, h->ev_fin[p], 0));   // set p is free again

  // the detection -> rectangle kernels are tiny and only the grid pass reads their output: they run on
  // the grid-pass stream right before it (same-stream order, no event), off the stream the sector
  // kernel waits for
  const bool rects_on_c = h->three_streams && h->rects_on_c;
  hipStream_t sR = rects_on_c ? h->stream3 : sA;
  int32_t n_rects = (vision && h->nb > 0) ? h->nb : ((!vision || h->nb <= 0) && h->n_poses > 0 ? h->n_poses : 0);
  auto launch_rects = [&]() {
    if (vision && h->nb > 0) {
      launch_vision(h->d_orient, h->d_conf, h->d_dims, h->d_bboxes, h->nb, h->cam, h->d_vout, h->d_poses, sR);
      launch_rects_from_poses(h->d_poses, h->nb, h->g, true, h->x_bc, rects, sR);
    } else if (h->n_poses > 0) {
      mark(sR);
      launch_rects_from_poses(h->d_poses, h->n_poses, h->g, false, h->x_bc, rects, sR);
      mark(sR);
    }
  };
  if (!rects_on_c) launch_rects();
  if (do_bin || do_bbox) {
    PointsArgs a{};
    a.x = h->cx; a.y = h->cy; a.z = h->cz;
    a.n = (uint32_t)h->n;
    a.g = h->g;
    a.m_base = h->m_base;
    a.m_cam = h->m_cam;
    a.cam = h->camk;
    a.org = h->org;
    a.bboxes = h->d_bboxes;
    a.nb = h->nb;
    a.bbox_f = h->d_bbox_f;
    a.tile_mask = h->d_tile_mask;
    a.tiles_x = h->tiles_x; a.tiles_y = h->tiles_y; a.mask_words = h->mask_words;
    a.hits = h->hits;
    a.hit8 = h->hit8;
    a.clip_end = h->clip_end;
    a.cell_idx = nullptr;
    a.bbox_id = h->bbox_id;
    a.do_bin = do_bin; a.do_ray = do_ray; a.do_bbox = do_bbox;
    a.counts = h->force_counts;
    mark(sA);
    if (h->split_points && do_bin && do_bbox) {
      // binning (what the bitmaps wait for) on stream A, the bbox test on its own stream: two lighter
      // kernels (46 and 30 VGPRs, no SGPR spills) instead of one fused pass over the cloud
      PointsArgs ab = a, ax = a;
      ab.do_bbox = false;
      ax.do_bin = false; ax.do_ray = false; ax.counts = false;
      launch_points(ab, sA);
      launch_points(ax, h->stream4);
    } else {
      launch_points(a, sA);
    }
    mark(sA);
  }
  if (do_bin) {
    BitmapArgs b{};
    b.nx = h->g.nx; b.ny = h->g.ny;
    b.hits = h->hits; b.clip_end = h->clip_end;
    b.hit8 = h->force_counts ? nullptr : h->hit8;
    b.hitN = hitN; b.clipN = clipN; b.hitT = hitT; b.clipT = clipT;
    b.nxw = h->nxw; b.nyw = h->nyw; b.nx_pad = h->nx_pad; b.ny_pad = h->ny_pad;
    b.zero_hits = true;
    mark(sA);
    launch_build_bitmaps(b, sA);
    mark(sA);
  }
  GV_HIP(hipEventRecord(h->ev_build[p], sA));
  GV_HIP(hipStreamWaitEvent(sB, h->ev_build[p], 0));
  // the grid pass runs on its own stream: HBM-bound, it overlaps the issue-bound sector kernel of the
  // next frame; the miss grids alternate with the frame parity like the bitmaps do
  hipStream_t sC = h->three_streams ? h->stream3 : sB;   // (two-stream mode runs with one sector stream)
  uint8_t *missN = h->x_miss[p], *missT = h->x_missT[p];
  if ((h->three_streams || h->sector_streams > 1) && h->since_drain >= h->n_sets) GV_HIP(hipStreamWaitEvent(sB, h->ev_fin[p], 0));   // miss set p cleared by its last reader
  if (do_ray && h->org.valid) {
    SectorArgs sa{};
    int rc = fill_sector_args(h, sa);
    if (rc) return rc;
    sa.hitN = hitN; sa.clipN = clipN; sa.hitT = hitT; sa.clipT = clipT;
    sa.missN = missN; sa.missT = missT;
    mark(sB);
    launch_ray_sectors(sa, sB);
    mark(sB);
  }
  if (rects_on_c) launch_rects();   // before the wait: they do not depend on the sector kernel
  if (h->three_streams) {
    GV_HIP(hipEventRecord(h->ev_sec[p], sB));
    GV_HIP(hipStreamWaitEvent(sC, h->ev_sec[p], 0));
  }
  FinalizeTileArgs t{};
  t.g = h->g;
  t.log_odds = h->log_odds;
  t.occupancy = h->occupancy;
  t.occ_i8 = h->occ_i8;
  t.rects = rects;
  t.n_rects = n_rects;
  t.hitN = hitN;
  t.nxw = h->nxw;
  t.ny_pad = h->ny_pad;
  t.missN = missN;
  t.missT = missT;
  t.counts = do_bin;
  t.zero = do_bin;
  t.use_missT = true;
  t.y_begin = 0;
  t.y_end = h->g.ny;
  mark(sC);
  launch_finalize_tiles(t, sC);
  mark(sC);
  GV_HIP(hipEventRecord(h->ev_fin[p], sC));
  GV_HIP(hipGetLastError());
  h->frame_no++;
  if (h->since_drain < h->n_sets) h->since_drain++;
  h->pipe_busy = true;
  h->counts_dirty = false;
  h->have_counts = false;
  h->have_cell_idx = false;
  h->have_bbox_id = do_bbox;
  return GV_OK;
}

#define GV_NCCL(call)                                                                          \
  do {                                                                                         \
    ncclResult_t r_ = (call);                                                                  \
    if (r_ != ncclSuccess) {                                                                   \
      char buf_[256];                                                                          \
      std::snprintf(buf_, sizeof(buf_), "%s:%d %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
      h->err = buf_;                                                                           \
      return GV_ERR_RCCL;                                                                      \
    }                                                                                          \
  } while (0)

inline void band_rows(const gv_context *h, int r, int32_t &y0, int32_t &y1)
{
  y0 = (int32_t)((int64_t)h->g.ny * r / h->world);
  y1 = (int32_t)((int64_t)h->g.ny * (r + 1) / h->world);
}

// [EXTENSION] SURVEY 8(e)-2: the one exchange step of the sharded frame.  Every rank has
// binned + ray-marched ITS slice of the points into private full-size count grids; integer
// sum / byte max make the result independent of the reduce order, hence bit-identical to
// one GPU.  Rank r then finalises row band r and the packed int8 bands are exchanged.
int sharded_tail(gv_context *h, int32_t n_rects)
{
  const int nx = h->g.nx, ny = h->g.ny;
  const size_t G = (size_t)h->g.G;
  launch_merge_miss(h->miss, h->missT, nx, ny, h->stream);   // miss = N | T^T, missT cleared
  GV_HIP(hipGetLastError());
  int32_t y0, y1;
  band_rows(h, h->rank, y0, y1);
  const bool counts = h->frame_counts;   // int32 sums only when the caller keeps the counts; else byte flags, max
  if (ny % h->world == 0) {
    const size_t cnt = G / (size_t)h->world;
    if (counts) GV_NCCL(ncclReduceScatter(h->hits, h->hits + (size_t)h->rank * cnt, cnt, ncclInt32, ncclSum, h->comm, h->stream));
    else GV_NCCL(ncclReduceScatter(h->hit8, h->hit8 + (size_t)h->rank * cnt, cnt, ncclUint8, ncclMax, h->comm, h->stream));
    GV_NCCL(ncclReduceScatter(h->miss, h->miss + (size_t)h->rank * cnt, cnt, ncclUint8, ncclMax, h->comm, h->stream));
  } else {   // bands are not equal sized: reduce everything everywhere
    if (counts) GV_NCCL(ncclAllReduce(h->hits, h->hits, G, ncclInt32, ncclSum, h->comm, h->stream));
    else GV_NCCL(ncclAllReduce(h->hit8, h->hit8, G, ncclUint8, ncclMax, h->comm, h->stream));
    GV_NCCL(ncclAllReduce(h->miss, h->miss, G, ncclUint8, ncclMax, h->comm, h->stream));
  }
  if (counts) launch_band_hit_bitmap(h->hits, nx, h->ny_pad, y0, y1, h->hitN, h->stream);
  else launch_band_hit_bitmap8(h->hit8, nx, h->ny_pad, y0, y1, h->hitN, h->stream);
  FinalizeTileArgs t{};
  t.g = h->g;
  t.log_odds = h->log_odds;
  t.occupancy = h->occupancy;
  t.occ_i8 = h->occ_i8;
  t.rects = h->d_rects;
  t.n_rects = n_rects;
  t.hitN = h->hitN;
  t.nxw = h->nxw;
  t.ny_pad = h->ny_pad;
  t.missN = h->miss;
  t.missT = h->missT;
  t.counts = true;
  t.zero = false;
  t.use_missT = false;
  t.y_begin = y0;
  t.y_end = y1;
  launch_finalize_tiles(t, h->stream);
  GV_HIP(hipGetLastError());
  if (counts) GV_HIP(hipMemsetAsync(h->hits, 0, G * sizeof(int32_t), h->stream));
  else GV_HIP(hipMemsetAsync(h->hit8, 0, G, h->stream));
  GV_HIP(hipMemsetAsync(h->miss, 0, G, h->stream));
  // packed bands to everyone: band r sits at data[G - e_r, G - b_r) (toOccupancyGrid order)
  GV_NCCL(ncclGroupStart());
  for (int r = 0; r < h->world; ++r) {
    int32_t r0, r1;
    band_rows(h, r, r0, r1);
    const size_t b = (size_t)r0 * nx, e = (size_t)r1 * nx;
    if (e > b) GV_NCCL(ncclBroadcast(h->occ_i8 + (G - e), h->occ_i8 + (G - e), e - b, ncclInt8, r, h->comm, h->stream));
  }
  GV_NCCL(ncclGroupEnd());
  h->counts_dirty = false;
  h->have_counts = false;
  h->have_cell_idx = false;
  h->have_bbox_id = (h->frame_flags & GV_FRAME_BBOX_TEST) != 0;
  return GV_OK;
}

int upload_bboxes(gv_context *h, const gv_bbox *b, int32_t nb)
{
  int rc = ensure_det(h, nb);
  if (rc) return rc;
  // float thresholds + tile candidate masks for the first-match test (extractCloudPerBBox)
  h->tiles_x = (h->cam.orig_w + 15) / 16;
  h->tiles_y = (h->cam.orig_h + 15) / 16;
  if (h->tiles_x < 1) h->tiles_x = 1;
  if (h->tiles_y < 1) h->tiles_y = 1;
  h->mask_words = std::max(1, (nb + 63) / 64);
  const size_t nmask = (size_t)h->tiles_x * h->tiles_y * h->mask_words;
  if ((rc = grow(h, h->d_tile_mask, h->tile_mask_cap, nmask))) return rc;
  std::vector<float4> bf((size_t)std::max(nb, 1));
  std::vector<unsigned long long> masks(nmask, 0ull);
  for (int32_t i = 0; i < nb; ++i) {
    float4 f;
    f.x = host::ceil_to_float(b[i].x_min);
    f.y = host::ceil_to_float(b[i].y_min);
    f.z = host::floor_to_float(b[i].x_max);
    f.w = host::floor_to_float(b[i].y_max);
    bf[i] = f;
    if (!(f.x <= f.z && f.y <= f.w)) continue;   // empty or NaN box never matches
    // tiles whose pixel range [16t, 16t+16) can contain a u in [f.x, f.z]
    int tx0 = (int)std::floor(std::max(f.x, 0.0f) / 16.0f), tx1 = (int)std::floor(std::min(f.z, 16.0f * h->tiles_x - 1.0f) / 16.0f);
    int ty0 = (int)std::floor(std::max(f.y, 0.0f) / 16.0f), ty1 = (int)std::floor(std::min(f.w, 16.0f * h->tiles_y - 1.0f) / 16.0f);
    tx0 = std::max(tx0, 0); ty0 = std::max(ty0, 0);
    tx1 = std::min(tx1, h->tiles_x - 1); ty1 = std::min(ty1, h->tiles_y - 1);
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx)
        masks[((size_t)ty * h->tiles_x + tx) * h->mask_words + (i >> 6)] |= 1ull << (i & 63);
  }
  if (nb > 0) {
    GV_HIP(hipMemcpyAsync(h->d_bboxes, b, (size_t)nb * sizeof(gv_bbox), hipMemcpyHostToDevice, h->stream));
    GV_HIP(hipMemcpyAsync(h->d_bbox_f, bf.data(), (size_t)nb * sizeof(float4), hipMemcpyHostToDevice, h->stream));
  }
  GV_HIP(hipMemcpyAsync(h->d_tile_mask, masks.data(), nmask * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));   // bf / masks are stack-owned
  return GV_OK;
}

}  // namespace

extern "C" {

int gv_abi_version(void) { return 1; }

int gv_create(gv_handle *out, uint8_t grid_x, uint8_t grid_y, double resolution, const gv_cam_params *cam,
              int device_id)
{
  if (!out) return GV_ERR_BAD_ARG;
  *out = nullptr;
  if (!cam || grid_x == 0 || grid_y == 0 || !(resolution > 0.0)) return GV_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GV_ERR_NO_DEVICE;
  gv_context *h = new (std::nothrow) gv_context();
  if (!h) return GV_ERR_HIP;
  GV_TRY
  if (device_id < 0) {
    if (hipGetDevice(&device_id) != hipSuccess) device_id = 0;
  }
  if (device_id >= ndev) { delete h; return GV_ERR_NO_DEVICE; }
  h->device = device_id;
  // grid_map::GridMap::setGeometry + setPosition  (src/occupancy_grid.cpp:10-11)
  GridParams &g = h->g;
  g.res = resolution;
  g.inv_res = 1.0 / resolution;
  const double sx = std::round((double)grid_x / resolution), sy = std::round((double)grid_y / resolution);
  if (!(sx >= 1.0 && sy >= 1.0) || sx * sy > (double)(1 << 30)) { delete h; return GV_ERR_BAD_ARG; }
  g.nx = (int32_t)sx;
  g.ny = (int32_t)sy;
  g.G = g.nx * g.ny;
  g.len_x = (double)g.nx * resolution;
  g.len_y = (double)g.ny * resolution;
  g.pos_x = (double)(grid_x / 3);   // uint8_t / int: integer division (:11)
  g.pos_y = 0.0;
  g.off_x = 0.5 * g.len_x;
  g.off_y = 0.5 * g.len_y;
  h->cam = *cam;
  host::intrinsics((double)cam->fx, (double)cam->fy, (double)cam->cx, (double)cam->cy, h->K, h->Kinv);
  for (int i = 0; i < 9; ++i) h->camk.k[i] = h->K[i];
  h->camk.W = cam->orig_w;
  h->camk.H = cam->orig_h;

  auto fail = [&](int code) { gv_destroy(h); return code; };
#define GV_C(call)                                           \
  do {                                                       \
    if ((call) != hipSuccess) return fail(GV_ERR_HIP);       \
  } while (0)
  GV_C(hipSetDevice(h->device));
  {
    // GV_PRIO (experiment): 1 = stream A (points/bitmaps) high priority, 2 = stream B high priority
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char *pe = std::getenv("GV_PRIO");
    const int mode = pe ? std::atoi(pe) : 0;
    GV_C(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, mode == 1 ? hi : (mode == 2 ? lo : 0)));
    GV_C(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, mode == 2 ? hi : (mode == 1 ? lo : 0)));
    GV_C(hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, mode == 3 ? hi : 0));
    GV_C(hipStreamCreateWithPriority(&h->stream4, hipStreamNonBlocking, 0));
    GV_C(hipStreamCreateWithPriority(&h->stream2b, hipStreamNonBlocking, mode == 2 ? hi : (mode == 1 ? lo : 0)));
  }
  for (int i = 0; i < gv_context::kSets; ++i) {
    GV_C(hipEventCreateWithFlags(&h->ev_build[i], hipEventDisableTiming));
    GV_C(hipEventCreateWithFlags(&h->ev_fin[i], hipEventDisableTiming));
    GV_C(hipEventCreateWithFlags(&h->ev_sec[i], hipEventDisableTiming));
  }
  const size_t G = (size_t)g.G;
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->log_odds), G * sizeof(float)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->occupancy), G * sizeof(float)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->occ_i8), G));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->hits), G * sizeof(int32_t)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->miss), G + 16));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->clip_end), G + 16));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->hit8), G + 16));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->ray_list), G * sizeof(uint32_t)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->ray_count), 4 * sizeof(uint32_t)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->ray_stats), kMaxStatSlots * 2 * sizeof(unsigned long long)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->scratch_i32), G * sizeof(int32_t)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->missT), G + 16));
  h->x_miss[0] = h->miss;
  h->x_missT[0] = h->missT;
  for (int k = 1; k < gv_context::kSets; ++k) {
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_miss[k]), G + 16));
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_missT[k]), G + 16));
    GV_C(hipMemsetAsync(h->x_miss[k], 0, G + 16, h->stream));
    GV_C(hipMemsetAsync(h->x_missT[k], 0, G + 16, h->stream));
  }
  h->nxw = 2 * ((g.nx + 63) / 64);
  h->nyw = 2 * ((g.ny + 63) / 64);
  h->nx_pad = 64 * ((g.nx + 63) / 64);
  h->ny_pad = 64 * ((g.ny + 63) / 64);
  {
    const size_t nN = (size_t)h->ny_pad * h->nxw + 4, nT = (size_t)h->nx_pad * h->nyw + 4;
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->hitN), nN * sizeof(uint32_t)));
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->clipN), nN * sizeof(uint32_t)));
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->hitT), nT * sizeof(uint32_t)));
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->clipT), nT * sizeof(uint32_t)));
    GV_C(hipMemsetAsync(h->hitN, 0, nN * sizeof(uint32_t), h->stream));
    GV_C(hipMemsetAsync(h->clipN, 0, nN * sizeof(uint32_t), h->stream));
    GV_C(hipMemsetAsync(h->hitT, 0, nT * sizeof(uint32_t), h->stream));
    GV_C(hipMemsetAsync(h->clipT, 0, nT * sizeof(uint32_t), h->stream));
    h->x_hitN[0] = h->hitN; h->x_clipN[0] = h->clipN; h->x_hitT[0] = h->hitT; h->x_clipT[0] = h->clipT;
    for (int k = 1; k < gv_context::kSets; ++k) {
      GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_hitN[k]), nN * sizeof(uint32_t)));
      GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_clipN[k]), nN * sizeof(uint32_t)));
      GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_hitT[k]), nT * sizeof(uint32_t)));
      GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_clipT[k]), nT * sizeof(uint32_t)));
      GV_C(hipMemsetAsync(h->x_hitN[k], 0, nN * sizeof(uint32_t), h->stream));
      GV_C(hipMemsetAsync(h->x_clipN[k], 0, nN * sizeof(uint32_t), h->stream));
      GV_C(hipMemsetAsync(h->x_hitT[k], 0, nT * sizeof(uint32_t), h->stream));
      GV_C(hipMemsetAsync(h->x_clipT[k], 0, nT * sizeof(uint32_t), h->stream));
    }
  }
  // packed (a,b) fields hold 13 bits each; vector stores need nx % 4 == 0
  h->tile_path = (g.nx % 4 == 0) && g.nx <= 8000 && g.ny <= 8000;
  {
    const char *impl = std::getenv("GV_RAY_IMPL");
    h->force_simple = impl && std::strcmp(impl, "simple") == 0;
    if (const char *e = std::getenv("GV_PIPELINE")) { h->no_pipeline = std::atoi(e) == 0; h->three_streams = std::atoi(e) != 2; }
    if (const char *e = std::getenv("GV_HIT_COUNTS")) h->force_counts = std::atoi(e) != 0;
    if (const char *e = std::getenv("GV_RECTS_ON_C")) h->rects_on_c = std::atoi(e) != 0;
    if (const char *e = std::getenv("GV_SPLIT_POINTS")) h->split_points = std::atoi(e) != 0;
    if (const char *e = std::getenv("GV_SECTOR_STREAMS")) h->sector_streams = std::max(1, std::min(2, std::atoi(e)));
    if (!h->three_streams) h->sector_streams = 1;
    if (const char *e = std::getenv("GV_PIPE_SETS")) h->n_sets = std::min(gv_context::kSets, std::max(2, std::atoi(e)));
    if (const char *e = std::getenv("GV_LOG2S")) h->env_log2s = std::atoi(e);
    if (const char *e = std::getenv("GV_LOG2S_OCT")) {
      int k = 0;
      for (const char *q = e; *q && k < 8; ++k) {
        h->env_log2s_oct[k] = std::atoi(q);
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
    }
    if (const char *e = std::getenv("GV_SECTOR_REORDER")) h->env_reorder = std::atoi(e) != 0;
    if (const char *e = std::getenv("GV_CAP")) h->env_cap = std::atoi(e);
    if (const char *e = std::getenv("GV_ABLATE")) h->env_ablate = std::atoi(e);
    if (const char *e = std::getenv("GV_FLAT_K")) h->env_flat_k = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("GV_SECTOR_DBG")) {
      if (std::atoi(e) > 0) {
        GV_C(hipMalloc(reinterpret_cast<void **>(&h->d_dbg), kMaxStatSlots * 16 * sizeof(unsigned long long)));
        GV_C(hipMemsetAsync(h->d_dbg, 0, kMaxStatSlots * 16 * sizeof(unsigned long long), h->stream));
      }
    }
    if (const char *e = std::getenv("GV_LOG2M")) h->env_log2m = std::min(9, std::max(4, std::atoi(e)));
  }
  for (auto &e : h->ev) GV_C(hipEventCreate(&e));
  GV_C(hipMemsetAsync(h->ray_count, 0, 4 * sizeof(uint32_t), h->stream));
  GV_C(hipMemsetAsync(h->ray_stats, 0, kMaxStatSlots * 2 * sizeof(unsigned long long), h->stream));
  h->x_stats[0] = h->ray_stats;
  for (int k = 1; k < gv_context::kSets; ++k) {
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_stats[k]), kMaxStatSlots * 2 * sizeof(unsigned long long)));
    GV_C(hipMemsetAsync(h->x_stats[k], 0, kMaxStatSlots * 2 * sizeof(unsigned long long), h->stream));
  }
#undef GV_C
  if (ensure_det(h, 64) != GV_OK) return fail(GV_ERR_HIP);
  if (clear_counts(h) != GV_OK) return fail(GV_ERR_HIP);
  *out = h;
  int rc = gv_reset(h);
  if (rc != GV_OK) { *out = nullptr; return fail(rc); }
  return GV_OK;
  GV_CATCH
}

int gv_destroy(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  if (h->stream3) (void)hipStreamSynchronize(h->stream3);
  if (h->stream2b) (void)hipStreamSynchronize(h->stream2b);
  if (h->stream4) (void)hipStreamSynchronize(h->stream4);
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  void *bufs[] = {h->log_odds, h->occupancy, h->occ_i8, h->hits, h->miss, h->clip_end, h->hit8, h->ray_list, h->ray_count,
                  h->ray_stats, h->scratch_i32, h->d_dbg, h->missT, h->hitN, h->clipN, h->hitT, h->clipT, h->cx, h->cy, h->cz, h->tx, h->ty, h->tz, h->raw, h->cell_idx,
                  h->bbox_id, h->d_bboxes, h->d_poses, h->d_rects, h->d_orient, h->d_conf, h->d_dims, h->d_vout,
                  h->d_pts, h->d_bbox_f, h->d_tile_mask, h->knn_partial, h->d_depths, h->d_knn_d2, h->d_idx, h->d_segof,
                  h->d_segstart, h->gx, h->gy, h->gz, h->d_keep, h->d_planes, h->d_plane_counts, h->d_ground};
  for (void *p : bufs)
    if (p) (void)hipFree(p);
  for (int k = 1; k < gv_context::kSets; ++k) {
    void *xs[] = {h->x_hitN[k], h->x_clipN[k], h->x_hitT[k], h->x_clipT[k], h->x_rects[k], h->x_miss[k], h->x_missT[k], h->x_stats[k]};
    for (void *p : xs)
      if (p) (void)hipFree(p);
  }
  for (auto &e : h->ev)
    if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < gv_context::kSets; ++i) {
    if (h->ev_build[i]) (void)hipEventDestroy(h->ev_build[i]);
    if (h->ev_fin[i]) (void)hipEventDestroy(h->ev_fin[i]);
    if (h->ev_sec[i]) (void)hipEventDestroy(h->ev_sec[i]);
  }
  if (h->stream3) (void)hipStreamDestroy(h->stream3);
  if (h->stream2b) (void)hipStreamDestroy(h->stream2b);
  if (h->stream4) (void)hipStreamDestroy(h->stream4);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return GV_OK;
}

const char *gv_last_error(gv_handle h) { return h ? h->err.c_str() : "null handle"; }

int gv_grid_geometry(gv_handle h, int32_t *nx, int32_t *ny, double *pos_x, double *pos_y)
{
  if (!h) return GV_ERR_BAD_ARG;
  if (nx) *nx = h->g.nx;
  if (ny) *ny = h->g.ny;
  if (pos_x) *pos_x = h->g.pos_x;
  if (pos_y) *pos_y = h->g.pos_y;
  return GV_OK;
}

int gv_reset(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  const size_t G = (size_t)h->g.G;
  launch_fill_f32(h->log_odds, kLogOddsPrior, G, h->stream);      // :12
  launch_fill_f32(h->occupancy, kInitProbability, G, h->stream);  // :13
  // toOccupancyGrid of the initial layer: (int8)(0.5f*100) = 50
  GV_HIP(hipMemsetAsync(h->occ_i8, 50, G, h->stream));
  GV_HIP(hipGetLastError());
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

int gv_set_transforms(gv_handle h, const gv_transform *cl, const gv_transform *bc, const gv_transform *bl)
{
  if (!h) return GV_ERR_BAD_ARG;
  GV_TRY
  if (cl) { h->tf_cl = *cl; h->m_cam = host::pcl_matrix_from_tf(*cl); h->has_cl = true; }
  if (bc) { h->tf_bc = *bc; h->x_bc = host::xform_from_tf(*bc); h->has_bc = true; }
  if (bl) { h->tf_bl = *bl; h->m_base = host::pcl_matrix_from_tf(*bl); h->has_bl = true; refresh_origin(h); }
  return GV_OK;
  GV_CATCH
}

int gv_cloud_upload_xyz(gv_handle h, const float *x, const float *y, const float *z, size_t n)
{
  if (!h || (n && (!x || !y || !z)) || n > 0x7fffffffu) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = ensure_cloud(h, n))) return rc;
  if (n) {
    GV_HIP(hipMemcpyAsync(h->cx, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    GV_HIP(hipMemcpyAsync(h->cy, y, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    GV_HIP(hipMemcpyAsync(h->cz, z, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  }
  GV_HIP(hipStreamSynchronize(h->stream));
  h->n = n;
  h->have_cell_idx = h->have_bbox_id = false;
  return GV_OK;
  GV_CATCH
}

int gv_cloud_upload_pointcloud2(gv_handle h, const uint8_t *data, size_t n, uint32_t point_step, uint32_t off_x,
                                uint32_t off_y, uint32_t off_z)
{
  if (!h || (n && !data) || n > 0x7fffffffu) return GV_ERR_BAD_ARG;
  if (point_step < 4 || off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step)
    return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = ensure_cloud(h, n))) return rc;
  const size_t bytes = n * (size_t)point_step;
  if ((rc = grow(h, h->raw, h->raw_cap, bytes + 16))) return rc;
  if (n) {
    GV_HIP(hipMemcpyAsync(h->raw, data, bytes, hipMemcpyHostToDevice, h->stream));
    launch_deinterleave(h->raw, (uint32_t)n, point_step, off_x, off_y, off_z, h->cx, h->cy, h->cz, h->stream);
    GV_HIP(hipGetLastError());
  }
  GV_HIP(hipStreamSynchronize(h->stream));
  h->n = n;
  h->have_cell_idx = h->have_bbox_id = false;
  return GV_OK;
  GV_CATCH
}

int gv_transform_lidar_to_camera(gv_handle h, float *x_cam, float *y_cam, float *z_cam)
{
  if (!h || !x_cam || !y_cam || !z_cam) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;   // the reference returns nullptr (:292-297)
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  const size_t n = h->n;
  if (n > h->tcap) {
    for (float **p : {&h->tx, &h->ty, &h->tz}) {
      if (*p) GV_HIP(hipFree(*p));
      *p = nullptr;
    }
    h->tcap = 0;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->tx), n * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->ty), n * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->tz), n * sizeof(float)));
    h->tcap = n;
  }
  if (n) {
    launch_transform_cloud(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, h->tx, h->ty, h->tz, h->stream);
    GV_HIP(hipGetLastError());
    GV_HIP(hipMemcpyAsync(x_cam, h->tx, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    GV_HIP(hipMemcpyAsync(y_cam, h->ty, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    GV_HIP(hipMemcpyAsync(z_cam, h->tz, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  }
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

int gv_extract_cloud_per_bbox(gv_handle h, const gv_bbox *bboxes, int32_t nb, int32_t *bbox_id, int32_t *counts)
{
  if (!h || nb < 0 || (nb && !bboxes) || !bbox_id) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = upload_bboxes(h, bboxes, nb))) return rc;
  PointsArgs a{};
  a.x = h->cx; a.y = h->cy; a.z = h->cz;
  a.n = (uint32_t)h->n;
  a.g = h->g;
  a.m_cam = h->m_cam;
  a.cam = h->camk;
  a.bboxes = h->d_bboxes;
  a.nb = nb;
  a.bbox_f = h->d_bbox_f;
  a.tile_mask = h->d_tile_mask;
  a.tiles_x = h->tiles_x; a.tiles_y = h->tiles_y; a.mask_words = h->mask_words;
  a.bbox_id = h->bbox_id;
  a.do_bbox = true;
  launch_points(a, h->stream);
  GV_HIP(hipGetLastError());
  if (h->n) GV_HIP(hipMemcpyAsync(bbox_id, h->bbox_id, h->n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  h->have_bbox_id = true;
  if (counts) {
    for (int32_t b = 0; b < nb; ++b) counts[b] = 0;
    for (size_t i = 0; i < h->n; ++i)
      if (bbox_id[i] >= 0) counts[bbox_id[i]]++;
  }
  return GV_OK;
  GV_CATCH
}

int gv_convert_pixels_to_3d(gv_handle h, const gv_bbox *bboxes, const float *depths, int32_t nb,
                            double *base_points_xyz)
{
  if (!h || nb < 0 || (nb && (!bboxes || !depths || !base_points_xyz))) return GV_ERR_BAD_ARG;
  if (!h->has_bc) return GV_ERR_TF;
  GV_TRY
  for (int32_t i = 0; i < nb; ++i) {
    // grid_vision_node.cpp:320-322 pixel centre (cv::Point2f), :325 pixelTo3D, :328-329 to base
    const float pcx = (float)(bboxes[i].x_min + ((bboxes[i].x_max - bboxes[i].x_min) / 2.0f));
    const float pcy = (float)(bboxes[i].y_min + ((bboxes[i].y_max - bboxes[i].y_min) / 2.0f));
    const double hx = pcx, hy = pcy, hz = 1.0;
    const double d = depths[i];
    double cam[3];
    for (int r = 0; r < 3; ++r)
      cam[r] = d * ((h->Kinv[r * 3] * hx + h->Kinv[r * 3 + 1] * hy) + h->Kinv[r * 3 + 2] * hz);   // cloud_detections.cpp:95
    host::apply(h->x_bc, cam, &base_points_xyz[3 * i]);
  }
  return GV_OK;
  GV_CATCH
}

int gv_vision_post_process(gv_handle h, const float *orient, const float *conf, const float *dims,
                           const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out, int32_t *n_out)
{
  if (!h || nb < 0 || !n_out || (nb && (!orient || !conf || !dims || !bboxes || !poses_out))) return GV_ERR_BAD_ARG;
  GV_TRY
  *n_out = 0;
  if (nb == 0) return GV_OK;
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = upload_bboxes(h, bboxes, nb))) return rc;
  GV_HIP(hipMemcpyAsync(h->d_orient, orient, (size_t)nb * 4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  GV_HIP(hipMemcpyAsync(h->d_conf, conf, (size_t)nb * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  GV_HIP(hipMemcpyAsync(h->d_dims, dims, (size_t)nb * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  launch_vision(h->d_orient, h->d_conf, h->d_dims, h->d_bboxes, nb, h->cam, h->d_vout, h->d_poses, h->stream);
  GV_HIP(hipGetLastError());
  std::vector<VisionOut> vo((size_t)nb);
  GV_HIP(hipMemcpyAsync(vo.data(), h->d_vout, (size_t)nb * sizeof(VisionOut), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  int32_t m = 0;
  for (int32_t i = 0; i < nb; ++i) {
    if (!vo[i].valid) continue;   // vision_orientation.cpp:496-499
    gv_lshape_pose p;
    p.px = vo[i].loc[0]; p.py = vo[i].loc[1]; p.pz = vo[i].loc[2];     // :434-436
    const host::Quat q = host::quat_from_rpy(0, -vo[i].orient, 0);       // :440
    p.qx = q.x; p.qy = q.y; p.qz = q.z; p.qw = q.w;
    p.length = vo[i].dims[0]; p.width = vo[i].dims[1]; p.height = vo[i].dims[2];
    poses_out[m++] = p;
  }
  *n_out = m;
  return GV_OK;
  GV_CATCH
}

int gv_transform_lshape_objects(gv_handle h, gv_lshape_pose *poses, int32_t n)
{
  if (!h || n < 0 || (n && !poses)) return GV_ERR_BAD_ARG;
  if (!h->has_bc) return GV_ERR_TF;
  GV_TRY
  for (int32_t i = 0; i < n; ++i) host::transform_pose(h->tf_bc, poses[i]);
  return GV_OK;
  GV_CATCH
}

int gv_extract_bboxes(const float *boxes, const float *scores, int32_t n, int32_t c, double conf_threshold,
                      double iou_threshold, int32_t orig_w, int32_t orig_h, int32_t resize, gv_bbox *out,
                      int32_t *n_out)
{
  gv_context *h = nullptr;
  if (!n_out || n < 0 || c <= 0 || resize <= 0 || (n && (!boxes || !scores || !out))) return GV_ERR_BAD_ARG;
  GV_TRY
  std::vector<gv_bbox> cand;
  for (int32_t i = 0; i < n; ++i) {
    int32_t best = 0;
    float mx = scores[(size_t)i * c];
    for (int32_t k = 1; k < c; ++k)
      if (scores[(size_t)i * c + k] > mx) { mx = scores[(size_t)i * c + k]; best = k; }   // :121-122
    if (mx >= conf_threshold) {                                                            // :125
      gv_bbox b;
      b.confidence = mx;
      b.label = host::object_class(best);
      b.x_min = boxes[i * 4 + 0]; b.y_min = boxes[i * 4 + 1];
      b.x_max = boxes[i * 4 + 2]; b.y_max = boxes[i * 4 + 3];
      cand.push_back(b);
    }
  }
  std::vector<gv_bbox> kept = host::nms(std::move(cand), (float)iou_threshold);   // :142
  host::denormalize(kept, orig_w, orig_h, resize);                                // :143
  for (size_t i = 0; i < kept.size(); ++i) out[i] = kept[i];
  *n_out = (int32_t)kept.size();
  return GV_OK;
  GV_CATCH
}

int gv_filter_bboxes(const gv_bbox *in, int32_t n, gv_bbox *static_out, int32_t *n_static, gv_bbox *dynamic_out,
                     int32_t *n_dynamic)
{
  if (n < 0 || !n_static || !n_dynamic || (n && (!in || !static_out || !dynamic_out))) return GV_ERR_BAD_ARG;
  int32_t ns = 0, nd = 0;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t l = in[i].label;
    if (l == 9 || l == 0 || l == 1 || l == 2) dynamic_out[nd++] = in[i];   // VEHICLE, BIKE, MOTORBIKE, PERSON
    else static_out[ns++] = in[i];
  }
  *n_static = ns;
  *n_dynamic = nd;
  return GV_OK;
}

int gv_get_intrinsics(gv_handle h, double K[9], double K_inv[9])
{
  if (!h) return GV_ERR_BAD_ARG;
  if (K) std::memcpy(K, h->K, sizeof(h->K));
  if (K_inv) std::memcpy(K_inv, h->Kinv, sizeof(h->Kinv));
  return GV_OK;
}

int gv_update_map(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = enqueue_plain_update(h, 0))) return rc;
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

int gv_update_map_poses(gv_handle h, const gv_lshape_pose *poses, int32_t n)
{
  if (!h || n < 0 || (n && !poses)) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = ensure_det(h, n))) return rc;
  if (n) {
    GV_HIP(hipMemcpyAsync(h->d_poses, poses, (size_t)n * sizeof(gv_lshape_pose), hipMemcpyHostToDevice, h->stream));
    launch_rects_from_poses(h->d_poses, n, h->g, false, h->x_bc, h->d_rects, h->stream);
  }
  if ((rc = enqueue_plain_update(h, n))) return rc;
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

int gv_update_map_points(gv_handle h, const double *pts, const gv_bbox *bboxes, int32_t n)
{
  if (!h || n < 0 || (n && (!pts || !bboxes))) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = upload_bboxes(h, bboxes, n))) return rc;
  if (n) {
    GV_HIP(hipMemcpyAsync(h->d_pts, pts, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_rects_from_points(h->d_pts, h->d_bboxes, n, h->g, h->d_rects, h->stream);
  }
  if ((rc = enqueue_plain_update(h, n))) return rc;
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

int gv_to_occupancy_grid(gv_handle h, int8_t *data, gv_grid_info *info)
{
  if (!h || !data) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  GV_HIP(hipMemcpyAsync(data, h->occ_i8, (size_t)h->g.G, hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  if (info) {
    info->width = (uint32_t)h->g.nx;
    info->height = (uint32_t)h->g.ny;
    info->resolution = h->g.res;
    info->origin_x = h->g.pos_x - 0.5 * h->g.len_x;
    info->origin_y = h->g.pos_y - 0.5 * h->g.len_y;
  }
  return GV_OK;
  GV_CATCH
}

static int copy_out(gv_context *h, void *dst, const void *src, size_t bytes)
{
  int rc = use_device(h);
  if (rc) return rc;
  GV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
}

int gv_get_log_odds(gv_handle h, float *out)
{
  if (!h || !out) return GV_ERR_BAD_ARG;
  return copy_out(h, out, h->log_odds, (size_t)h->g.G * sizeof(float));
}

int gv_get_occupancy(gv_handle h, float *out)
{
  if (!h || !out) return GV_ERR_BAD_ARG;
  return copy_out(h, out, h->occupancy, (size_t)h->g.G * sizeof(float));
}

int gv_set_log_odds(gv_handle h, const float *in)
{
  if (!h || !in) return GV_ERR_BAD_ARG;
  int rc = use_device(h);
  if (rc) return rc;
  GV_HIP(hipMemcpyAsync(h->log_odds, in, (size_t)h->g.G * sizeof(float), hipMemcpyHostToDevice, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
}

int gv_frame_set_detections(gv_handle h, const gv_frame_desc *d)
{
  if (!h || !d) return GV_ERR_BAD_ARG;
  if (d->n_bboxes < 0 || d->n_poses < 0) return GV_ERR_BAD_ARG;
  if (d->n_bboxes && !d->bboxes) return GV_ERR_BAD_ARG;
  const bool vision = d->flags & GV_FRAME_VISION_ORIENT;
  if (vision && d->n_bboxes && (!d->orient || !d->conf || !d->dims)) return GV_ERR_BAD_ARG;
  if (!vision && d->n_poses && !d->poses) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  const int32_t need = std::max(d->n_bboxes, d->n_poses);
  if ((rc = ensure_det(h, need))) return rc;
  if ((rc = upload_bboxes(h, d->bboxes, d->n_bboxes))) return rc;
  if (vision && d->n_bboxes) {
    const size_t nb = (size_t)d->n_bboxes;
    GV_HIP(hipMemcpyAsync(h->d_orient, d->orient, nb * 4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    GV_HIP(hipMemcpyAsync(h->d_conf, d->conf, nb * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    GV_HIP(hipMemcpyAsync(h->d_dims, d->dims, nb * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  } else if (!vision && d->n_poses) {
    GV_HIP(hipMemcpyAsync(h->d_poses, d->poses, (size_t)d->n_poses * sizeof(gv_lshape_pose), hipMemcpyHostToDevice,
                          h->stream));
  }
  GV_HIP(hipStreamSynchronize(h->stream));   // host buffers are free to reuse after return
  h->frame_flags = d->flags;
  h->nb = d->n_bboxes;
  h->n_poses = vision ? 0 : d->n_poses;
  return GV_OK;
  GV_CATCH
}

int gv_frame_enqueue(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  GV_TRY
  const uint32_t keep = GV_FRAME_KEEP_CELL_IDX | GV_FRAME_KEEP_COUNTS;
  const bool pipelined = h->tile_path && !h->force_simple && !h->no_pipeline && !(h->frame_flags & keep);
  if (pipelined) {
    int rc = set_device_only(h);
    if (rc) return rc;
    return enqueue_frame_pipelined(h);
  }
  int rc = use_device(h);
  if (rc) return rc;
  return enqueue_frame(h, false);
  GV_CATCH
}

int gv_synchronize(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  int rc = use_device(h);
  if (rc) return rc;
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
}

int gv_process_frame(gv_handle h, const gv_frame_desc *desc)
{
  int rc = gv_frame_set_detections(h, desc);
  if (rc) return rc;
  if ((rc = gv_frame_enqueue(h))) return rc;
  return gv_synchronize(h);
}

int gv_get_hits(gv_handle h, int32_t *out)
{
  if (!h || !out) return GV_ERR_BAD_ARG;
  if (!h->have_counts) return GV_ERR_STATE;
  return copy_out(h, out, h->hits, (size_t)h->g.G * sizeof(int32_t));
}

int gv_get_miss(gv_handle h, int32_t *out)
{
  if (!h || !out) return GV_ERR_BAD_ARG;
  if (!h->have_counts) return GV_ERR_STATE;
  int rc = use_device(h);
  if (rc) return rc;
  if (h->tile_path && !h->force_simple)
    launch_miss_to_i32(h->miss, h->missT, h->g.nx, h->g.ny, h->scratch_i32, h->stream);
  else
    launch_u8_to_i32(h->miss, h->scratch_i32, (size_t)h->g.G, h->stream);
  GV_HIP(hipGetLastError());
  return copy_out(h, out, h->scratch_i32, (size_t)h->g.G * sizeof(int32_t));
}

int gv_get_cell_idx(gv_handle h, int32_t *out)
{
  if (!h || !out) return GV_ERR_BAD_ARG;
  if (!h->have_cell_idx) return GV_ERR_STATE;
  return copy_out(h, out, h->cell_idx, h->n * sizeof(int32_t));
}

int gv_get_bbox_id(gv_handle h, int32_t *out)
{
  if (!h || !out) return GV_ERR_BAD_ARG;
  if (!h->have_bbox_id) return GV_ERR_STATE;
  return copy_out(h, out, h->bbox_id, h->n * sizeof(int32_t));
}

int gv_get_ray_stats(gv_handle h, uint64_t *n_rays, uint64_t *n_visits)
{
  if (!h) return GV_ERR_BAD_ARG;
  GV_TRY
  std::vector<unsigned long long> st(2 * h->stat_slots, 0ull);
  int rc = copy_out(h, st.data(), h->x_stats[h->last_stats_set], st.size() * sizeof(unsigned long long));
  if (rc) return rc;
  unsigned long long rays = 0, visits = 0;
  for (size_t i = 0; i < h->stat_slots; ++i) { rays += st[2 * i]; visits += st[2 * i + 1]; }
  if (n_rays) *n_rays = rays;
  if (n_visits) *n_visits = visits;
  return GV_OK;
  GV_CATCH
}

// diagnostic only (tools/sector_phases.py): copies the phase stamps of the last sector launch
int gv_debug_sector_stamps(gv_handle h, unsigned long long *out, size_t n_wg)
{
  if (!h || !out || !h->d_dbg) return GV_ERR_STATE;
  return copy_out(h, out, h->d_dbg, n_wg * 16 * sizeof(unsigned long long));
}

// diagnostic (not part of the ABI): enqueue `frames` pipelined frames with timing events around every
// kernel; out[frame*10 + 2*k + {0,1}] = start/end in us of kernel k (rects, points, bitmaps, sectors, grid pass)
extern "C" int gv_debug_pipeline_trace(gv_handle h, int32_t frames, float *out)
{
  if (!h || frames <= 0 || !out) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  std::vector<hipEvent_t> ev;
  hipEvent_t e0;
  GV_HIP(hipEventCreate(&e0));
  GV_HIP(hipEventRecord(e0, h->stream));
  h->trace = &ev;
  for (int32_t i = 0; i < frames && rc == GV_OK; ++i) rc = gv_frame_enqueue(h);
  h->trace = nullptr;
  int rc2 = use_device(h);
  for (size_t k = 0; k < ev.size(); ++k) {
    float ms = 0.f;
    if (k < (size_t)frames * 10 && hipEventElapsedTime(&ms, e0, ev[k]) == hipSuccess) out[k] = ms * 1000.f;
    (void)hipEventDestroy(ev[k]);
  }
  (void)hipEventDestroy(e0);
  return rc ? rc : rc2;
  GV_CATCH
}

void *gv_stream(gv_handle h) { return h ? (void *)h->stream : nullptr; }

int gv_time_frames(gv_handle h, int32_t frames, float *ms_total)
{
  if (!h || frames <= 0 || !ms_total) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  hipEvent_t e0 = h->ev[0], e1 = h->ev[kNumStages];
  GV_HIP(hipEventRecord(e0, h->stream));
  for (int32_t i = 0; i < frames; ++i)
    if ((rc = gv_frame_enqueue(h))) return rc;
  if (h->pipe_busy) {   // join streams B and C into stream A before the closing event
    GV_HIP(hipEventRecord(h->ev[1], h->stream2));
    GV_HIP(hipStreamWaitEvent(h->stream, h->ev[1], 0));
    GV_HIP(hipEventRecord(h->ev[2], h->stream3));
    GV_HIP(hipStreamWaitEvent(h->stream, h->ev[2], 0));
    GV_HIP(hipEventRecord(h->ev[3], h->stream2b));
    GV_HIP(hipStreamWaitEvent(h->stream, h->ev[3], 0));
    GV_HIP(hipEventRecord(h->ev[4], h->stream4));
    GV_HIP(hipStreamWaitEvent(h->stream, h->ev[4], 0));
  }
  GV_HIP(hipEventRecord(e1, h->stream));
  GV_HIP(hipEventSynchronize(e1));
  GV_HIP(hipEventElapsedTime(ms_total, e0, e1));
  return GV_OK;
  GV_CATCH
}

int gv_time_frame_stages(gv_handle h, int32_t frames, float *stage_ms)
{
  if (!h || frames <= 0 || !stage_ms) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  for (int s = 0; s < kNumStages; ++s) stage_ms[s] = 0.0f;
  for (int32_t i = 0; i < frames; ++i) {
    if ((rc = enqueue_frame(h, true))) return rc;
    GV_HIP(hipEventSynchronize(h->ev[kNumStages]));
    for (int s = 0; s < kNumStages; ++s) {
      float ms = 0.0f;
      GV_HIP(hipEventElapsedTime(&ms, h->ev[s], h->ev[s + 1]));
      stage_ms[s] += ms;
    }
  }
  for (int s = 0; s < kNumStages; ++s) stage_ms[s] /= (float)frames;
  return GV_OK;
  GV_CATCH
}

static int ensure_tbuf(gv_context *h, size_t n)
{
  if (n <= h->tcap) return GV_OK;
  for (float **p : {&h->tx, &h->ty, &h->tz}) {
    if (*p) GV_HIP(hipFree(*p));
    *p = nullptr;
  }
  h->tcap = 0;
  GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->tx), n * sizeof(float)));
  GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->ty), n * sizeof(float)));
  GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->tz), n * sizeof(float)));
  h->tcap = n;
  return GV_OK;
}

int gv_compute_depth_for_bboxes(gv_handle h, const gv_bbox *bboxes, int32_t nb, int32_t k, float *depths,
                                float *knn_d2)
{
  if (!h || nb < 0 || (nb && (!bboxes || !depths)) || k < 1 || k > 32) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  GV_TRY
  if (nb == 0) return GV_OK;
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = upload_bboxes(h, bboxes, nb))) return rc;
  if ((rc = ensure_tbuf(h, std::max<size_t>(h->n, 1)))) return rc;
  const int nchunks = 64;
  if ((rc = grow(h, h->knn_partial, h->knn_partial_cap, (size_t)nb * nchunks * k))) return rc;
  if ((size_t)nb * k > h->knn_out_cap) {
    if (h->d_depths) GV_HIP(hipFree(h->d_depths));
    if (h->d_knn_d2) GV_HIP(hipFree(h->d_knn_d2));
    h->d_depths = h->d_knn_d2 = nullptr;
    h->knn_out_cap = 0;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_depths), (size_t)nb * 32 * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_knn_d2), (size_t)nb * 32 * sizeof(float)));
    h->knn_out_cap = (size_t)nb * 32;
  }
  // buildKDTree projection (cloud_detections.cpp:8-33) then the exact k nearest (:43-87)
  launch_project_uvd(h->cx, h->cy, h->cz, (uint32_t)h->n, h->m_cam, h->camk, h->tx, h->ty, h->tz, h->stream);
  launch_knn(h->tx, h->ty, h->tz, (uint32_t)h->n, h->d_bboxes, nb, k, nchunks, h->knn_partial, h->d_depths,
             h->d_knn_d2, h->stream);
  GV_HIP(hipGetLastError());
  GV_HIP(hipMemcpyAsync(depths, h->d_depths, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (knn_d2)
    GV_HIP(hipMemcpyAsync(knn_d2, h->d_knn_d2, (size_t)nb * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

static int compute_bbox_pose_impl(gv_handle h, const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out,
                                  uint8_t *valid, const uint8_t *skip)
{
  if (!h || nb < 0 || (nb && (!bboxes || !poses_out || !valid))) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  GV_TRY
  if (nb == 0) return GV_OK;
  int rc = use_device(h);
  if (rc) return rc;
  const size_t n = h->n;
  for (int32_t b = 0; b < nb; ++b) { valid[b] = 0; poses_out[b] = gv_lshape_pose{}; }
  if (n == 0) return GV_OK;
  if ((rc = upload_bboxes(h, bboxes, nb))) return rc;
  if ((rc = ensure_tbuf(h, n))) return rc;
  // extractCloudPerBBox (cloud_detections.cpp:250-298): first-match bbox id per point,
  // and the camera-frame cloud the per-bbox clouds are cut from
  {
    PointsArgs a{};
    a.x = h->cx; a.y = h->cy; a.z = h->cz;
    a.n = (uint32_t)n;
    a.g = h->g;
    a.m_cam = h->m_cam;
    a.cam = h->camk;
    a.bboxes = h->d_bboxes;
    a.nb = nb;
    a.bbox_f = h->d_bbox_f;
    a.tile_mask = h->d_tile_mask;
    a.tiles_x = h->tiles_x; a.tiles_y = h->tiles_y; a.mask_words = h->mask_words;
    a.bbox_id = h->bbox_id;
    a.do_bbox = true;
    launch_points(a, h->stream);
    launch_transform_cloud(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, h->tx, h->ty, h->tz, h->stream);
    GV_HIP(hipGetLastError());
  }
  std::vector<int32_t> ids(n);
  GV_HIP(hipMemcpyAsync(ids.data(), h->bbox_id, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  h->have_bbox_id = true;
  // per-bbox point lists in cloud order (the reference appends in cloud order, :286)
  std::vector<int32_t> seg_start((size_t)nb + 1, 0);
  if (skip)   // ground points were removed before extractCloudPerBBox (cloud_detections.cpp:306-314)
    for (size_t i = 0; i < n; ++i)
      if (skip[i]) ids[i] = -1;
  for (size_t i = 0; i < n; ++i)
    if (ids[i] >= 0) seg_start[(size_t)ids[i] + 1]++;
  for (int32_t b = 0; b < nb; ++b) seg_start[b + 1] += seg_start[b];
  const int32_t m = seg_start[nb];
  if (m == 0) return GV_OK;
  std::vector<int32_t> idx((size_t)m), seg_of((size_t)m), cur(seg_start.begin(), seg_start.end() - 1);
  for (size_t i = 0; i < n; ++i)
    if (ids[i] >= 0) {
      const int32_t p = cur[ids[i]]++;
      idx[p] = (int32_t)i;
      seg_of[p] = ids[i];
    }
  if ((size_t)m > h->gcap) {
    for (float **p : {&h->gx, &h->gy, &h->gz})
      if (*p) { GV_HIP(hipFree(*p)); *p = nullptr; }
    if (h->d_keep) { GV_HIP(hipFree(h->d_keep)); h->d_keep = nullptr; }
    h->gcap = 0;
    const size_t want = (size_t)m + (size_t)m / 4 + 1024;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->gx), want * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->gy), want * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->gz), want * sizeof(float)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_keep), want));
    h->gcap = want;
  }
  if ((size_t)m > h->seg_cap) {
    if (h->d_idx) { GV_HIP(hipFree(h->d_idx)); h->d_idx = nullptr; }
    if (h->d_segof) { GV_HIP(hipFree(h->d_segof)); h->d_segof = nullptr; }
    h->seg_cap = 0;
    const size_t want = (size_t)m + (size_t)m / 4 + 1024;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_idx), want * sizeof(int32_t)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_segof), want * sizeof(int32_t)));
    h->seg_cap = want;
  }
  if ((rc = grow(h, h->d_segstart, h->segstart_cap, (size_t)nb + 1))) return rc;
  GV_HIP(hipMemcpyAsync(h->d_idx, idx.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  GV_HIP(hipMemcpyAsync(h->d_segof, seg_of.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  GV_HIP(hipMemcpyAsync(h->d_segstart, seg_start.data(), ((size_t)nb + 1) * sizeof(int32_t), hipMemcpyHostToDevice,
                        h->stream));
  launch_gather_xyz(h->tx, h->ty, h->tz, h->d_idx, m, h->gx, h->gy, h->gz, h->stream);
  // RadiusOutlierRemoval(0.4, 10)  (cloud_detections.cpp:150-154)
  const double radius = 0.4;
  launch_radius_count(h->gx, h->gy, h->gz, h->d_segof, h->d_segstart, m, host::floor_to_float(radius * radius), 10,
                      h->d_keep, h->stream);
  GV_HIP(hipGetLastError());
  std::vector<float> px((size_t)m), py((size_t)m), pz((size_t)m);
  std::vector<uint8_t> keep((size_t)m);
  GV_HIP(hipMemcpyAsync(px.data(), h->gx, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipMemcpyAsync(py.data(), h->gy, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipMemcpyAsync(pz.data(), h->gz, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipMemcpyAsync(keep.data(), h->d_keep, (size_t)m, hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  // centroid + PCA rectangle per bbox on the filtered points, reference order (:156-247)
  std::vector<float> fx, fy, fz;
  for (int32_t b = 0; b < nb; ++b) {
    fx.clear(); fy.clear(); fz.clear();
    for (int32_t p = seg_start[b]; p < seg_start[b + 1]; ++p)
      if (keep[p]) { fx.push_back(px[p]); fy.push_back(py[p]); fz.push_back(pz[p]); }
    valid[b] = host::pca_bbox(fx.data(), fy.data(), fz.data(), fx.size(), poses_out[b]) ? 1 : 0;
  }
  return GV_OK;
  GV_CATCH
}

int gv_compute_bbox_pose(gv_handle h, const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out, uint8_t *valid)
{
  return compute_bbox_pose_impl(h, bboxes, nb, poses_out, valid, nullptr);
}

int gv_segment_ground_plane(gv_handle h, double threshold, int32_t iterations, uint64_t seed, uint8_t *is_ground,
                            float coeff[4], int64_t *n_inliers)
{
  if (!h || !(threshold > 0.0) || iterations < 1 || iterations > 4096) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  const size_t n = h->n;
  h->ground_mask.assign(n, 0);
  if (coeff) coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.0f;
  if (n_inliers) *n_inliers = 0;
  if (n < 3) return GV_OK;
  if ((rc = ensure_tbuf(h, n))) return rc;
  // camera-frame cloud (the reference segments transformed_cloud, grid_vision_node.cpp:215-216)
  launch_transform_cloud(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, h->tx, h->ty, h->tz, h->stream);
  GV_HIP(hipGetLastError());
  std::vector<float> px(n), py(n), pz(n);
  GV_HIP(hipMemcpyAsync(px.data(), h->tx, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipMemcpyAsync(py.data(), h->ty, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipMemcpyAsync(pz.data(), h->tz, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  // hypotheses: three counter-based draws each (specified by outcome, not PCL's RNG)
  std::vector<float4> planes;
  std::vector<int32_t> hyp_of;
  for (int32_t t = 0; t < iterations; ++t) {
    size_t id[3];
    for (int k = 0; k < 3; ++k) id[k] = (size_t)(host::splitmix64(seed + 3ull * (uint64_t)t + (uint64_t)k) % (uint64_t)n);
    const float p0[3] = {px[id[0]], py[id[0]], pz[id[0]]}, p1[3] = {px[id[1]], py[id[1]], pz[id[1]]},
                p2[3] = {px[id[2]], py[id[2]], pz[id[2]]};
    float c[4];
    if (!host::plane_from_sample(p0, p1, p2, c)) continue;
    planes.push_back(make_float4(c[0], c[1], c[2], c[3]));
    hyp_of.push_back(t);
  }
  if (planes.empty()) return GV_OK;
  const size_t nh = planes.size();
  if (nh > h->planes_cap) {
    if (h->d_planes) { GV_HIP(hipFree(h->d_planes)); h->d_planes = nullptr; }
    if (h->d_plane_counts) { GV_HIP(hipFree(h->d_plane_counts)); h->d_plane_counts = nullptr; }
    h->planes_cap = 0;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_planes), nh * sizeof(float4)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_plane_counts), nh * sizeof(unsigned)));
    h->planes_cap = nh;
  }
  if ((rc = grow(h, h->d_ground, h->ground_cap, n))) return rc;
  GV_HIP(hipMemcpyAsync(h->d_planes, planes.data(), nh * sizeof(float4), hipMemcpyHostToDevice, h->stream));
  GV_HIP(hipMemsetAsync(h->d_plane_counts, 0, nh * sizeof(unsigned), h->stream));
  launch_plane_count(h->tx, h->ty, h->tz, (uint32_t)n, h->d_planes, (int)nh, threshold, h->d_plane_counts, h->stream);
  GV_HIP(hipGetLastError());
  std::vector<unsigned> counts(nh);
  GV_HIP(hipMemcpyAsync(counts.data(), h->d_plane_counts, nh * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  size_t best = 0;
  unsigned bestc = 0;
  for (size_t k = 0; k < nh; ++k)
    if (counts[k] > bestc) { bestc = counts[k]; best = k; }   // first wins ties
  if (bestc == 0) return GV_OK;   // "Could not estimate a planar model" (:122-126)
  const float c0[4] = {planes[best].x, planes[best].y, planes[best].z, planes[best].w};
  float refined[4];
  host::refine_plane(px.data(), py.data(), pz.data(), n, c0, threshold, refined);   // optimizeCoefficients
  launch_plane_mask(h->tx, h->ty, h->tz, (uint32_t)n, make_float4(refined[0], refined[1], refined[2], refined[3]),
                    threshold, h->d_ground, h->stream);
  GV_HIP(hipGetLastError());
  GV_HIP(hipMemcpyAsync(h->ground_mask.data(), h->d_ground, n, hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  int64_t m = 0;
  for (size_t i = 0; i < n; ++i) m += h->ground_mask[i];
  if (is_ground) std::memcpy(is_ground, h->ground_mask.data(), n);
  if (coeff) std::memcpy(coeff, refined, sizeof(refined));
  if (n_inliers) *n_inliers = m;
  return GV_OK;
  GV_CATCH
}

int gv_compute_bbox_pose_ground_removed(gv_handle h, const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out,
                                        uint8_t *valid, int32_t *n_poses_or_fail)
{
  if (!h) return GV_ERR_BAD_ARG;
  // computeBBoxPose (cloud_detections.cpp:300-321): segmentGroundPlane -> extractCloudPerBBox -> PCA
  int64_t m = 0;
  int rc = gv_segment_ground_plane(h, 0.04, 50, 12345ull, nullptr, nullptr, &m);
  if (rc) return rc;
  if (n_poses_or_fail) *n_poses_or_fail = 0;
  if (m == 0 || (size_t)m == h->n) {   // empty segmented cloud -> the reference returns {} (:307-309)
    for (int32_t b = 0; b < nb; ++b) valid[b] = 0;
    if (n_poses_or_fail) *n_poses_or_fail = -1;
    return GV_OK;
  }
  rc = compute_bbox_pose_impl(h, bboxes, nb, poses_out, valid, h->ground_mask.data());
  if (rc) return rc;
  if (n_poses_or_fail)
    for (int32_t b = 0; b < nb; ++b) *n_poses_or_fail += valid[b];
  return GV_OK;
}

int gv_comm_unique_id(uint8_t id_out[128])
{
  if (!id_out) return GV_ERR_BAD_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return GV_ERR_RCCL;
  std::memcpy(id_out, &id, sizeof(id));
  return GV_OK;
}

int gv_comm_init(gv_handle h, const uint8_t id[128], int32_t rank, int32_t world)
{
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return GV_ERR_BAD_ARG;
  if (h->comm) return GV_ERR_STATE;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  GV_NCCL(ncclCommInitRank(&h->comm, world, uid, rank));
  h->rank = rank;
  h->world = world;
  return GV_OK;
  GV_CATCH
}

int gv_comm_destroy(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  if (!h->comm) return GV_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  ncclCommDestroy(h->comm);
  h->comm = nullptr;
  h->rank = 0;
  h->world = 1;
  return GV_OK;
}

int gv_process_frame_sharded(gv_handle h, const gv_frame_desc *desc)
{
  if (!h || !desc) return GV_ERR_BAD_ARG;
  if (!h->comm) return GV_ERR_STATE;
  int rc = gv_frame_set_detections(h, desc);
  if (rc) return rc;
  GV_TRY
  if ((rc = use_device(h))) return rc;
  if ((rc = enqueue_frame(h, false, true))) return rc;
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

int gv_comm_band(gv_handle h, int64_t *begin, int64_t *end)
{
  if (!h) return GV_ERR_BAD_ARG;
  int32_t y0, y1;
  band_rows(h, h->rank, y0, y1);
  if (begin) *begin = (int64_t)y0 * h->g.nx;
  if (end) *end = (int64_t)y1 * h->g.nx;
  return GV_OK;
}

}  // extern "C"
