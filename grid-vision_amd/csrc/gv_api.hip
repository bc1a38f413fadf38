// gv_api.hip -- C ABI (include/gridvision_hip.h) over the gfx950 kernels.
// One gv_context = one device + one resident grid + its HIP streams.  No exception
// leaves this file; every entry point returns a gv_status.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "gv_host_math.hpp"
#include "gv_test_hooks.h"
#include "gv_kernels.hpp"

using namespace gv;

namespace {

// One of the three resident clouds: frames read the current one while the copy stream fills the next
// (cloudCallback / timerCallback overlap, src/grid_vision_node.cpp:103-106,108-244).  Three, so that the
// set being filled was last read two uploads ago: its readers have long finished in a streaming run.
struct CloudSet {
  float *base = nullptr;        // one allocation of 3 * cap floats
  float *x = nullptr, *y = nullptr, *z = nullptr;   // base, base + n, base + 2n of the cloud it holds (SoA, back to back)
  size_t cap = 0;
  uint8_t *raw = nullptr;       // PointCloud2 bytes before the de-interleave
  size_t raw_cap = 0;
  hipEvent_t ready = nullptr;   // copy stream: upload complete
  int release_slot = -1;        // ev_fin[release_slot]: the last frame that reads this set (-1: none since it was filled)
  uint32_t seen = ~0u;          // bit k: stream k has waited for `ready` (or the upload is known complete)
};

// Per-frame detection inputs (bboxes, poses / network outputs) and what the device derives from them.
// Sets 0/1 alternate between "read by the frames in flight" and "being uploaded"; set 2 belongs to the
// standalone entry points of the reference surface, which therefore never disturb the frame's inputs.
struct DetSet {
  uint8_t *block = nullptr;                  // ONE device allocation = one H2D copy per frame; the arrays below point into it
  gv_bbox *bboxes = nullptr;
  gv_lshape_pose *poses = nullptr;
  float *orient = nullptr, *conf = nullptr, *dims = nullptr;
  float4 *bbox_f = nullptr;                  // float thresholds of the bbox test
  unsigned long long *tile_mask = nullptr;   // candidate masks per 16x16-pixel tile
  size_t tile_mask_cap = 0;
  int32_t cap = 0;
  int32_t mask_words = 1;
  int32_t nb = 0, n_poses = 0;
  uint32_t flags = 0;
  bool valid = false;           // a gv_frame_set_detections* call has filled this set
  uint8_t *stage = nullptr;     // pinned host copy of the caller's arrays (free to reuse on return)
  size_t stage_cap = 0;
  hipEvent_t ready = nullptr;   // the set's last upload is complete (and has left its staging block)
  uint32_t seen = ~0u;          // bit k: stream k (0 public, 1 / 2 the lanes) is ordered after the upload
  int release_slot = -1;        // last frame that reads this set: ev_fin[release_slot] (a finished grid pass => every earlier frame finished)
  uint32_t readers = 0;         // bit k: a frame on stream k has read this set since its last upload
};

}  // namespace

struct gv_context {
  // Frames in flight run on LANES (three; GV_LANES=2: two): frame f does partition, tile pass and sector stage back
  // to back on the in-order stream of lane f % lanes, then its grid pass on the PUBLIC stream behind one event.
  // No event sits between the stages on a lane (a barrier packet between two kernels costs ~6 us of queue time,
  // back to back kernels of one queue follow each other with a gap of a few us that the other lanes fill), the
  // grid passes are one in-order sequence by construction (the log-odds grid is one sequence of updates), and
  // everything a frame produced is visible on the public stream right behind it.  Buffer sets 1..2*lanes rotate
  // with the frames (set 0: serial frames and standalone calls); a set is handed to frame f + 2*lanes once the
  // HOST has seen frame f finish -- back-pressure on the caller instead of a barrier on a lane.
  // Measured on config 3: one in-order stream 12.0 k frames/s; one stream per STAGE with three events per frame
  // (round 1) 14.4 k; two lanes with the grid pass on the lane behind a cross-lane wait 16.5 k; two lanes as above
  // 18.6 k in round 2, 23.5 k at the end of round 3; three lanes 24.7 k (the in-kernel timeline of the two-lane
  // form shows the lanes in step, all of them between kernels at the same moments: profiles/r03/native_timeline.txt).
  // Three lanes + public + copy are five streams on the four hardware queues a process gets by default; with
  // GPU_MAX_HW_QUEUES=8 the same five streams run slower (57 us per frame against 40).  Independent HANDLES side by
  // side (three or four grids, round 2: 15.4 / 14.3 k, tools/multi_handle.py) are a different thing: every grid
  // pays its own grid pass.
  static constexpr int kLanesMax = 3;              // three lanes by default, GV_LANES=2: two
  static constexpr int kStreams = 1 + kLanesMax;   // public + lanes
  static constexpr int kSets = 1 + 2 * kLanesMax;  // set 0: the serial frame; two sets per lane
  static constexpr int kRing = 8;   // event rings: one slot per frame, reused every 8 frames
  int n_lanes = 3;
  int upload_stream_retries = 0;    // gv_create: upload streams replaced because they shared a hardware queue
  double upload_probe_us = 0.0;     // the last probe's wait
  // The third lane runs on the UPLOAD stream (public + two lanes + uploads are the four hardware queues a process
  // gets; a fifth stream shares one of them with whatever the runtime picks, and when that is the upload stream the
  // streamed frame drops to 0.8 of the copy rate).  It is used only while the upload stream is quiet: no cloud
  // upload for kQuietFrames frames.  With a cloud per frame the library runs on two lanes, as in round 2.
  static constexpr uint32_t kQuietFrames = 8;
  uint32_t quiet_frames = 0;        // frames enqueued since the last cloud upload
  int lanes_now() const { return (n_lanes == 3 && quiet_frames >= kQuietFrames) ? 3 : 2; }
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr, stream4 = nullptr, stream_copy = nullptr;
  hipStream_t streams[kStreams]{};  // = {stream (public), stream2 (lane 0), stream3 (lane 1), stream4 (lane 2, GV_LANES=3)}
  hipEvent_t ev_sec[kRing]{};       // lane: partition, tile pass, sector stage of frame (slot) done
  hipEvent_t ev_fin[kRing]{};       // public stream: grid pass of frame (slot) done => that frame and every earlier one are done
  hipEvent_t ev_join = nullptr;     // copy stream -> public stream (gv_frame_fence)
  uint64_t lane_frames = 0;         // lane frames enqueued so far: lane = n % lanes, buffer set = 1 + n % (2 * lanes)
  int set_fin_slot[kSets]{-1, -1, -1, -1, -1, -1, -1};   // ev_fin slot of the last frame that used the set
  int last_fin_slot = -1;           // ev_fin slot of the most recently enqueued frame (-1: idle)
  // per-set buffers of the frames in flight: end bitmaps, rectangles, free-cell bitmaps, ray statistics
  uint32_t *x_ends[kSets]{};      // one allocation per set: [hitN | clipN | hitT | clipT], ends_words in all
  uint32_t *x_hitN[kSets]{}, *x_clipN[kSets]{}, *x_hitT[kSets]{}, *x_clipT[kSets]{};
  uint32_t *x_free[kSets]{};      // [freeN | freeT]: free-cell bitmaps of the ray stage
  uint32_t *x_freeN[kSets]{}, *x_freeT[kSets]{};
  size_t ends_words = 0, bmN_words = 0, bmT_words = 0;
  Rect *x_rects[kSets]{};
  uint8_t *miss8 = nullptr;       // generic path only: byte miss grid of the literal march
  unsigned long long *x_stats[kSets]{};
  int last_set = 0;
  uint64_t frame_no = 0;
  bool pipe_busy = false;         // lane frames enqueued since the streams were last drained
  bool no_pipeline = false;       // GV_PIPELINE=0
  int32_t env_sector_rev = -1;    // GV_SECTOR_REV (sweeps)
#ifdef GV_DIAG
  std::vector<hipEvent_t> *trace = nullptr;   // timing events around every pipelined kernel (gv_debug_pipeline_trace)
  unsigned long long *d_dbg = nullptr;        // GV_SECTOR_DBG=1: phase stamps of the sector kernel
  unsigned long long *d_bin_dbg[2] = {nullptr, nullptr};   // GV_BIN_DBG=1: phase stamps of the partition / tile kernels
  int32_t env_ablate = 0;                     // GV_ABLATE
  unsigned long long *d_tl = nullptr;         // GV_TIMELINE=1: {begin, end} of the four kernels of the last kTlFrames frames
  static constexpr uint64_t kTlFrames = 4096;
  unsigned long long *tl_slot(int kernel) const
  {
    return d_tl ? d_tl + ((frame_no % kTlFrames) * 4 + (uint64_t)kernel) * 2 : nullptr;
  }
#endif
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
  int32_t *hits_s[kStreams]{};                     // per stream (public, lane 0, lane 1); tile path: every cell written by every BIN frame
  int32_t *hits = nullptr;                  // = hits_s[stream of the last frame]
  uint8_t *clip_end = nullptr;              // generic path only
  uint32_t *ray_list = nullptr;
  uint32_t *ray_count = nullptr;            // [0] = number of list entries
  int32_t *scratch_i32 = nullptr;           // G ints (miss read-back), also max(N) ints for id read-back
  size_t scratch_cap = 0;
  int32_t nxw = 0, nyw = 0, nx_pad = 0, ny_pad = 0;
  bool tile_path = false;                   // nx % 4 == 0 and the grid fits the packed (a,b) fields
  bool force_simple = false;                // GV_RAY_IMPL=simple
  int env_reorder = 1;                      // GV_SECTOR_REORDER=0: workgroups in natural (octant, sector) order
  int env_helpers = -1;                     // GV_SECTOR_HELPERS: -1 automatic, 0 off, 1 on
  // A lane's partition pass does not depend on the sector kernel queued in front of it (the previous frame of that
  // lane: other buffers), only the in-order queue says so.  When nothing else was put on the lane since that
  // sector kernel -- no wait, no upload, no table kernel -- the partition pass is launched without the barrier
  // bit (hipExtAnyOrderLaunch) and starts while the sector kernel's last workgroups still run.  lane_clean[k]:
  // the last packet on lane k is a sector kernel.  GV_ANYORDER=0 switches it off.
  bool lane_clean[kStreams] = {false, false, false, false};
  bool env_anyorder = true;
  size_t stat_slots = 1;                    // ray statistics slots written by the last frame
  int32_t env_log2s_oct[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // GV_LOG2S_OCT="a,b,..." per octant index (sweeps)
  uint32_t env_march_limit = 64u * 512u;     // GV_MARCH_LIMIT
  uint32_t env_flat_direct = 2048;           // GV_FLAT_DIRECT
  int32_t env_flat_k = 8;                   // GV_FLAT_K: exact-cell : marched-cell cost ratio (0 = always march)
  int32_t env_log2s = 0, env_cap = 0, env_log2m = 0;     // GV_LOG2S / GV_CAP / GV_LOG2M (sweeps)

  // tile-path binning (gv_binning.hip)
  int32_t tiles_x = 0, tiles_y = 0, n_tiles = 0;
  uint16_t *bin_keys[kStreams]{}, *bin_tab[kStreams]{};   // per stream: partition(f+1) of one lane runs beside tiles(f) of the other
  size_t bin_keys_cap = 0, bin_tab_cap = 0;
  uint32_t *bin_total[kStreams][2]{};
  uint32_t *bin_done[kStreams]{}, *bin_scratch[kStreams]{};
  size_t bin_slots = 0;
  int bin_parity[kStreams]{};

  // resident clouds
  CloudSet cloud[3];
  int cloud_cur = 0;
  bool cloud_wait = false;                  // an asynchronous upload may still be in flight
  float *cx = nullptr, *cy = nullptr, *cz = nullptr;   // = cloud[cloud_cur]
  size_t n = 0;
  float *tx = nullptr, *ty = nullptr, *tz = nullptr;   // transformed copy (A1 read-back)
  size_t tcap = 0;
  int32_t *cell_idx_s[kStreams]{};                 // per-point outputs, per stream (two frames in flight write them)
  int16_t *bbox_id_s[kStreams]{};
  int32_t *cell_idx = nullptr;              // = *_s[stream of the last frame]
  int16_t *bbox_id = nullptr;
  size_t idx_cap = 0;

  // detections
  DetSet det[3];
  int det_cur = 0;
  int32_t bt_tiles_x = 1, bt_tiles_y = 1;   // 16x16-pixel tiles of the image
  VisionOut *d_vout_s[kStreams]{};
  VisionOut *d_vout = nullptr;              // = d_vout_s[0]
  int32_t vout_cap = 0;
  double *d_pts = nullptr;
  int32_t pts_cap = 0;
  // kNN depth / PCA pose scratch
  Cand2 *knn_partial = nullptr; size_t knn_partial_cap = 0;
  CellNode *d_nodes = nullptr; uint8_t *d_keep = nullptr; size_t pc_cap = 0;   // selected points in bucket order; 1 = survives the radius filter
  uint32_t *d_ticket_of = nullptr;   // per cloud point: its slot inside its bucket (selected points only)
  long long *d_pca_acc = nullptr; unsigned *d_pca_ext = nullptr; size_t pca_cap = 0;   // per bbox: integer sums / extent keys of the PCA rectangle (zero between calls)
  unsigned *d_pca_ticket = nullptr;
  uint32_t *d_cellcnt = nullptr, *d_cellpre = nullptr, *d_celloff = nullptr; size_t head_cap = 0;   // cell buckets: counts, prefix, block offsets (+ ticket)
  float4 *d_planes = nullptr; unsigned *d_plane_counts = nullptr; size_t planes_cap = 0;
  uint8_t *d_ground = nullptr; size_t ground_cap = 0;   // last ground mask (device resident)
  size_t ground_n = 0;
  double *d_rscratch = nullptr; size_t rscratch_cap = 0;   // tree-sum partials of the plane refinement
  RansacState *d_rstate = nullptr;
  // result block of the synchronous kNN / RANSAC / PCA calls: pinned and device-mapped, written by the call's last
  // kernel; [0] = the sequence number of the last finished call (CallDone, gv_kernels.hpp), payload from byte 64
  uint8_t *res_host = nullptr; size_t res_cap = 0;
  unsigned *d_res_ticket = nullptr;
  unsigned res_seq = 0;

  // the node's tick (gv_tick_enqueue / gv_tick_wait): what the pending tick put where in the result block
  struct Tick {
    bool pending = false;
    uint32_t flags = 0;
    int32_t n_all = 0, n_static = 0, n_dynamic = 0;
    bool pca_ran = false, vision_ran = false, knn_ran = false;
    size_t off_depth = 0, off_pose = 0, off_vout = 0;
    std::vector<gv_bbox> st_boxes;   // the static boxes (host copy: convertPixelsTo3D after the wait)
    hipEvent_t done = nullptr;       // public stream: everything the tick enqueued has finished
    hipEvent_t fork = nullptr, join = nullptr;   // the kNN depth on a lane beside the pose branch
  } tick;
  bool env_tick_knn_lane = true;   // GV_TICK_KNN_LANE=0: the static boxes' kNN in line on the public stream

  bool counts_dirty = false;   // generic path: hits/miss/clip_end hold a kept frame
  bool have_hits = false, have_miss = false, have_cell_idx = false, have_bbox_id = false;

  // multi-GPU (one large frame sharded by points)
  ncclComm_t comm = nullptr;
  int32_t rank = 0, world = 1;
  uint32_t *sh_xchg = nullptr;    // exchange scratch: `world` received slices / packed bands
  size_t sh_xchg_cap = 0;
  hipStream_t stream_x = nullptr; // the exchanges of the sharded frame (created by gv_comm_init)
  hipEvent_t ev_sh[kRing][5]{};   // per frame slot: binning, exchange 1, sectors + packing, exchange 2, grid pass done
  int sh_counts_slot[kStreams]{-1, -1, -1, -1};   // ev_fin slot of the lane's last sharded KEEP_COUNTS frame (its x3 reduces hits_s[lane] in place)
  hipEvent_t sh_t[7]{};           // stage timing of the sharded frame (gv_time_frame_sharded_stages)

  hipEvent_t ev[kNumStages + 1]{};
  // stage timing (gv_time_frame_stages): start / end of the partition, tile-pass, sector and grid-pass kernels,
  // taken from their own dispatch packets; kt_used: the kernel was launched in the frame just timed
  hipEvent_t kt[4][2]{};
  bool kt_used[4]{};
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

int drain(gv_context *h)
{
  GV_HIP(hipStreamSynchronize(h->stream_copy));
  GV_HIP(hipStreamSynchronize(h->stream));
  GV_HIP(hipStreamSynchronize(h->stream2));
  GV_HIP(hipStreamSynchronize(h->stream3));
  if (h->stream4) GV_HIP(hipStreamSynchronize(h->stream4));
  if (h->stream_x) GV_HIP(hipStreamSynchronize(h->stream_x));
  h->pipe_busy = false;
  h->last_fin_slot = -1;
  for (int &q : h->set_fin_slot) q = -1;
  h->cloud_wait = false;
  for (auto &c : h->cloud) { c.seen = ~0u; c.release_slot = -1; }   // every upload landed, every reader finished
  for (auto &d : h->det) { d.seen = ~0u; d.release_slot = -1; d.readers = 0; }
  for (int &q : h->sh_counts_slot) q = -1;
  return GV_OK;
}

bool sector_path(const gv_context *h) { return h->tile_path && !h->force_simple; }

int set_device_only(gv_context *h)
{
  GV_HIP(hipSetDevice(h->device));
  return GV_OK;
}

// Every entry point except the streaming ones (gv_frame_enqueue, gv_*_async, the uploads) starts from
// idle streams: its work on the public stream then sees every earlier frame and upload completed.
int use_device(gv_context *h)
{
  GV_HIP(hipSetDevice(h->device));
  if (h->pipe_busy || h->cloud_wait) return drain(h);
  return GV_OK;
}

// buffers whose size follows the cloud: per-point outputs and the binning scratch, one of each per
// buffer set.  Growing them needs the frames in flight to finish first (rare: the cloud grew).
// keys / table entries one binning launch over n points needs (the chunk size follows n)
void bin_needs(const gv_context *h, size_t n, size_t &keys_need, size_t &tab_need)
{
  const uint32_t chunk = bin_chunk_for(n);
  const size_t n_wg = (n + chunk - 1) / chunk;
  keys_need = n_wg * chunk + 64;   // + slack: the tile pass reads whole 16-byte windows
  tab_need = n_wg * ((size_t)h->n_tiles + 1) + 2;
}

// n_slice > 0: binning launches over slices of n_slice points will run as well (the one-device emulation of
// the sharded frame): a slice may pick a smaller chunk than the whole cloud and then needs MORE table rows
int ensure_point_buffers(gv_context *h, size_t n, size_t n_slice = 0)
{
  const int nsets = sector_path(h) ? 1 + h->n_lanes : 1;   // per-stream copies
  const bool need_idx = n > h->idx_cap || !h->cell_idx;
  size_t keys_need, tab_need;
  bin_needs(h, n, keys_need, tab_need);
  for (size_t m : {n_slice, n_slice ? n_slice - 1 : (size_t)0}) {   // slices are floor or ceil of n / world
    if (!m) continue;
    size_t k2, t2;
    bin_needs(h, m, k2, t2);
    keys_need = std::max(keys_need, k2);
    tab_need = std::max(tab_need, t2);
  }
  const size_t slots_need = n / kBinSplitKeys + 1;
  const bool need_bin = sector_path(h) && (keys_need > h->bin_keys_cap || tab_need > h->bin_tab_cap || slots_need > h->bin_slots);
  if (!need_idx && !need_bin) return GV_OK;
  int rc = drain(h);
  if (rc) return rc;
  if (need_idx) {
    const size_t want = n + n / 8 + 1024;
    h->idx_cap = 0;
    h->cell_idx = nullptr;
    h->bbox_id = nullptr;
    for (int k = 0; k < nsets; ++k) {
      if (h->cell_idx_s[k]) GV_HIP(hipFree(h->cell_idx_s[k]));
      if (h->bbox_id_s[k]) GV_HIP(hipFree(h->bbox_id_s[k]));
      h->cell_idx_s[k] = nullptr;
      h->bbox_id_s[k] = nullptr;
      GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->cell_idx_s[k]), want * sizeof(int32_t)));
      GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->bbox_id_s[k]), want * sizeof(int16_t)));
    }
    h->cell_idx = h->cell_idx_s[0];
    h->bbox_id = h->bbox_id_s[0];
    h->have_cell_idx = h->have_bbox_id = false;
    h->idx_cap = want;
  }
  if (need_bin) {
    const size_t keys_want = std::max(h->bin_keys_cap, keys_need + keys_need / 8);
    const size_t tab_want = std::max(h->bin_tab_cap, tab_need + tab_need / 8);
    const size_t slots_want = slots_need > h->bin_slots ? slots_need + slots_need / 8 : h->bin_slots;
    const size_t keys_had = h->bin_keys_cap, tab_had = h->bin_tab_cap, slots_had = h->bin_slots;
    h->bin_keys_cap = h->bin_tab_cap = h->bin_slots = 0;
    for (int k = 0; k < nsets; ++k) {
      size_t cap = keys_had;
      if ((rc = grow(h, h->bin_keys[k], cap, keys_want))) return rc;
      cap = tab_had;
      if ((rc = grow(h, h->bin_tab[k], cap, tab_want))) return rc;
      if (slots_want > slots_had) {
        if (h->bin_scratch[k]) GV_HIP(hipFree(h->bin_scratch[k]));
        h->bin_scratch[k] = nullptr;
        GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->bin_scratch[k]),
                         slots_want * kBinSplitMax * ((size_t)kBinTileCells + 512) * sizeof(uint32_t)));
      }
    }
    h->bin_keys_cap = keys_want;
    h->bin_tab_cap = tab_want;
    h->bin_slots = slots_want;
  }
  return GV_OK;
}

int ensure_scratch_i32(gv_context *h, size_t n)
{
  if (n <= h->scratch_cap) return GV_OK;
  return grow(h, h->scratch_i32, h->scratch_cap, n + n / 8);
}

// layout of a detection block for `cap` entries (host staging and device copy share it)
struct DetLayout {
  size_t bboxes, poses, orient, conf, dims, total;
};
DetLayout det_layout(int32_t cap)
{
  DetLayout L;
  size_t o = 0;
  L.bboxes = o; o += (size_t)cap * sizeof(gv_bbox);
  L.poses = o;  o += (size_t)cap * sizeof(gv_lshape_pose);
  L.orient = o; o += (size_t)cap * 4 * sizeof(float);
  L.conf = o;   o += (size_t)cap * 2 * sizeof(float);
  L.dims = o;   o += (size_t)cap * 3 * sizeof(float);
  L.total = (o + 15) & ~(size_t)15;
  return L;
}

int ensure_det(gv_context *h, DetSet &d, int32_t n)
{
  if (n <= d.cap) return GV_OK;
  if (d.cap) {   // frames that read this set must be past it (rare: the count grew)
    int rc0 = drain(h);
    if (rc0) return rc0;
  }
  const int32_t want = std::max(n + n / 4, 64);
  const DetLayout L = det_layout(want);
  d.cap = 0;
  if (d.block) GV_HIP(hipFree(d.block));
  d.block = nullptr;
  GV_HIP(hipMalloc(reinterpret_cast<void **>(&d.block), L.total));
  d.bboxes = reinterpret_cast<gv_bbox *>(d.block + L.bboxes);
  d.poses = reinterpret_cast<gv_lshape_pose *>(d.block + L.poses);
  d.orient = reinterpret_cast<float *>(d.block + L.orient);
  d.conf = reinterpret_cast<float *>(d.block + L.conf);
  d.dims = reinterpret_cast<float *>(d.block + L.dims);
  if (d.bbox_f) GV_HIP(hipFree(d.bbox_f));
  d.bbox_f = nullptr;
  GV_HIP(hipMalloc(reinterpret_cast<void **>(&d.bbox_f), (size_t)want * sizeof(float4)));
  const size_t nmask = (size_t)h->bt_tiles_x * h->bt_tiles_y * (size_t)((want + 63) / 64);
  int rc = grow(h, d.tile_mask, d.tile_mask_cap, nmask);
  if (rc) return rc;
  if (d.stage) GV_HIP(hipHostFree(d.stage));
  d.stage = nullptr;
  d.stage_cap = 0;
  GV_HIP(hipHostMalloc(reinterpret_cast<void **>(&d.stage), L.total, hipHostMallocDefault));
  d.stage_cap = L.total;
  d.cap = want;
  return GV_OK;
}

// rectangles and vision outputs (all buffer sets) and centre points follow the detection count; they
// are written by the frames in flight, hence the drain
int ensure_det_shared(gv_context *h, int32_t n)
{
  if (n <= h->vout_cap) return GV_OK;
  int rc = drain(h);
  if (rc) return rc;
  const int32_t want = std::max(n + n / 4, 64);
  auto re = [&](auto *&p, size_t bytes) -> int {
    if (p) GV_HIP(hipFree(p));
    p = nullptr;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&p), bytes));
    return GV_OK;
  };
  h->vout_cap = 0;
  for (int k = 0; k < gv_context::kSets; ++k)
    if ((rc = re(h->x_rects[k], (size_t)want * sizeof(Rect)))) return rc;
  for (int k = 0; k < gv_context::kStreams; ++k)
    if ((rc = re(h->d_vout_s[k], (size_t)want * sizeof(VisionOut)))) return rc;
  h->d_vout = h->d_vout_s[0];
  if ((rc = re(h->d_pts, (size_t)want * 3 * sizeof(double)))) return rc;
  h->vout_cap = want;
  h->pts_cap = want;
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

// generic path: the atomics-based count grids start every frame from zero
int clear_counts(gv_context *h)
{
  const size_t G = (size_t)h->g.G;
  GV_HIP(hipMemsetAsync(h->hits, 0, G * sizeof(int32_t), h->stream));
  GV_HIP(hipMemsetAsync(h->miss8, 0, G, h->stream));
  GV_HIP(hipMemsetAsync(h->clip_end, 0, G, h->stream));
  h->counts_dirty = false;
  return GV_OK;
}

BBoxTest bbox_test_of(const gv_context *h, const DetSet &d)
{
  BBoxTest t;
  t.bbox_f = d.bbox_f;
  t.tile_mask = d.tile_mask;
  t.tiles_x = h->bt_tiles_x;
  t.tiles_y = h->bt_tiles_y;
  t.mask_words = d.mask_words;
  return t;
}

// Upload the small per-frame arrays into detection set `d` on stream `s` and derive the bbox-test
// tables there.  The caller's arrays are copied into the set's pinned staging block first (they are
// free on return) and the block goes to the device in ONE asynchronous copy.
// masks = false: the caller's kernels read the raw boxes / poses only (kNN depth, vision orientation, plain pose
// update) -- the thresholds and tile masks of the bbox test are not rebuilt, and whoever tests points against this
// set uploads it again first (every such call does).
// n_net >= 0: the network outputs cover n_net boxes (default: nb).  nb_test >= 0: the bbox test -- thresholds, tile
// masks, d.nb -- covers the first nb_test boxes only; what follows them in the block is read by other kernels (the
// tick keeps [all | static | dynamic] boxes in one block: one copy).
int upload_det(gv_context *h, DetSet &d, const gv_bbox *bboxes, int32_t nb, const gv_lshape_pose *poses,
               int32_t n_poses, const float *orient, const float *conf, const float *dims, hipStream_t s, bool masks = true,
               int32_t n_net = -1, int32_t nb_test = -1, bool fused = false)
{
  if (n_net < 0) n_net = nb;
  if (nb_test < 0) nb_test = nb;
  int rc = ensure_det(h, d, std::max(nb, n_poses));
  if (rc) return rc;
  if ((rc = ensure_det_shared(h, std::max(nb, n_poses)))) return rc;
  if (d.ready) GV_HIP(hipEventSynchronize(d.ready));   // the staging's previous copy has left it
  const DetLayout L = det_layout(d.cap);
  size_t used = 0;   // the block is copied up to the end of the last array in use
  auto put = [&](size_t off, const void *src, size_t bytes) {
    if (!bytes) return;
    std::memcpy(d.stage + off, src, bytes);
    used = std::max(used, off + bytes);
  };
  put(L.bboxes, bboxes, (size_t)nb * sizeof(gv_bbox));
  put(L.poses, poses, (size_t)n_poses * sizeof(gv_lshape_pose));
  if (orient) put(L.orient, orient, (size_t)n_net * 4 * sizeof(float));
  if (conf) put(L.conf, conf, (size_t)n_net * 2 * sizeof(float));
  if (dims) put(L.dims, dims, (size_t)n_net * 3 * sizeof(float));
  d.mask_words = std::max(1, (nb_test + 63) / 64);
  if (fused && used) {
    // fused: ONE kernel reads the pinned staging (device visible) -- copies the block and builds the tables from the
    // staged boxes -- instead of a copy command (7 us as a blit kernel) + the table kernel behind it
    launch_bbox_prepare(reinterpret_cast<const gv_bbox *>(d.stage + L.bboxes), masks ? nb_test : 0, h->bt_tiles_x, h->bt_tiles_y,
                        d.mask_words, d.bbox_f, d.tile_mask, s, d.stage, d.block, used);
  } else {
    if (used) GV_HIP(hipMemcpyAsync(d.block, d.stage, used, hipMemcpyHostToDevice, s));
    if (masks) launch_bbox_prepare(d.bboxes, nb_test, h->bt_tiles_x, h->bt_tiles_y, d.mask_words, d.bbox_f, d.tile_mask, s);
  }
  GV_HIP(hipGetLastError());
  d.nb = nb_test;
  d.n_poses = n_poses;
  d.valid = true;
  return GV_OK;
}

// bboxes only, synchronously, into the standalone set (extractCloudPerBBox and friends)
int upload_scratch_bboxes(gv_context *h, const gv_bbox *b, int32_t nb, bool masks = true)
{
  if (h->tick.pending) { h->err = "a tick is pending: call gv_tick_wait first"; return GV_ERR_STATE; }
  DetSet &d = h->det[2];
  // (fused: the table kernel reads the pinned staging itself -- one launch instead of a copy command + a kernel: 6 us)
  int rc = upload_det(h, d, b, nb, nullptr, 0, nullptr, nullptr, nullptr, h->stream, masks, -1, -1, true);
  if (rc) return rc;
  GV_HIP(hipEventRecord(d.ready, h->stream));
  return GV_OK;
}

// plain grid update (A7 / A8 / A10): rectangles already in x_rects[0]
int enqueue_plain_update(gv_context *h, int32_t n_rects)
{
  if (sector_path(h)) {
    FinalizeTileArgs t{};
    t.g = h->g;
    t.log_odds = h->log_odds;
    t.occupancy = h->occupancy;
    t.occ_i8 = h->occ_i8;
    t.rects = h->x_rects[0];
    t.n_rects = n_rects;
    t.hitN = h->x_hitN[0];
    t.freeN = h->x_freeN[0];
    t.freeT = h->x_freeT[0];
    t.nx_pad = h->nx_pad;
    t.ny_pad = h->ny_pad;
    t.counts = false;
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
  f.rects = h->x_rects[0];
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

// sector-kernel launch parameters for the resident cloud and grid, buffer set p
int fill_sector_args(gv_context *h, SectorArgs &sa, int p)
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
  }
  // Rows of 512 columns, 8 blocks of 64 each, one block per wavefront: a row goes to the wavefronts in ascending or in
  // descending order, whichever keeps the fullest wavefront lightest (far columns are wider: weight ~ column number),
  // rows taken from the heaviest (last) one down.  GV_SECTOR_REV=0 / 1: never / always alternate (experiments).
  for (int o = 0; o < 8; ++o) {
    const int rows = (len[o] + 511) / 512;
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint16_t mask = 0;
    const double wbase = (double)(1 << sa.log2s_oct[o]);   // a column's cost: its cells (a / S + 1), times S
    for (int r = std::min(rows, 16) - 1; r >= 0; --r) {
      double w[8];
      for (int b = 0; b < 8; ++b) {
        const int a0 = 512 * r + 64 * b + 1, a1 = std::min(a0 + 63, len[o]);
        w[b] = a1 >= a0 ? (double)(a1 - a0 + 1) * (0.5 * (double)(a0 + a1) + wbase) : 0.0;
      }
      double up = 0.0, down = 0.0;
      for (int b = 0; b < 8; ++b) { up = std::max(up, load[b] + w[b]); down = std::max(down, load[b] + w[7 - b]); }
      bool rev = down < up;
      if (h->env_sector_rev == 0) rev = false;
      if (h->env_sector_rev == 1) rev = (r & 1) != 0;
      if (rev) mask |= (uint16_t)(1u << r);
      for (int b = 0; b < 8; ++b) load[b] += rev ? w[7 - b] : w[b];
    }
    sa.rev_oct[o] = mask;
  }
  sa.cap = h->env_cap > 0 ? std::max(2048, h->env_cap) : ((est_max <= 1700.0 && h->env_log2s <= 0) ? 2048 : 4096);
  sa.ablate = 0;
  sa.dbg = nullptr;
#ifdef GV_DIAG
  sa.ablate = h->env_ablate;
  sa.dbg = h->d_dbg;
  sa.tl = h->tl_slot(2);
#endif
  sa.flat_k = h->env_flat_k;
  sa.march_limit = h->env_march_limit;
  sa.flat_direct = h->env_flat_direct;
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
  // second workgroups for the axis / diagonal sectors of every octant once the wedges are long enough to have
  // heavy tails (GV_SECTOR_HELPERS=0 / 1 forces them off / on)
  sa.n_helpers = (h->env_helpers >= 0) ? (h->env_helpers ? 16 : 0) : (imax >= 512 ? 16 : 0);
  if (base + 16 > kMaxStatSlots || base > 65535u) { h->err = "too many sector workgroups"; return GV_ERR_BAD_ARG; }
  sa.wg_base[8] = (uint16_t)base;
  sa.hitN = h->x_hitN[p]; sa.clipN = h->x_clipN[p]; sa.hitT = h->x_hitT[p]; sa.clipT = h->x_clipT[p];
  sa.nxw = h->nxw; sa.nyw = h->nyw; sa.nx_pad = h->nx_pad; sa.ny_pad = h->ny_pad;
  sa.freeN = h->x_freeN[p];
  sa.freeT = h->x_freeT[p];
  sa.stats = h->x_stats[p];
  sa.wg_first = 0;
  sa.wg_stride = 1;
  h->stat_slots = (size_t)sa.wg_base[8] + (size_t)sa.n_helpers;
  return GV_OK;
}

// poses / network outputs of detection set D -> index rectangles on stream s
int32_t enqueue_rects(gv_context *h, const DetSet &D, Rect *rects, VisionOut *vout, hipStream_t s)
{
  const bool vision = D.flags & GV_FRAME_VISION_ORIENT;
  if (vision && D.nb > 0) {
    launch_vision(D.orient, D.conf, D.dims, D.bboxes, D.nb, h->cam, vout, D.poses, s);
    launch_rects_from_poses(D.poses, D.nb, h->g, true, h->x_bc, rects, s);
    return D.nb;
  }
  if (!vision && D.n_poses > 0) {
    launch_rects_from_poses(D.poses, D.n_poses, h->g, false, h->x_bc, rects, s);
    return D.n_poses;
  }
  return 0;
}

int check_frame_flags(const gv_context *h, uint32_t fl)
{
  const bool do_bin = fl & GV_FRAME_BIN, do_ray = fl & GV_FRAME_RAYMARCH, do_bbox = fl & GV_FRAME_BBOX_TEST;
  if (do_ray && !do_bin) return GV_ERR_BAD_ARG;
  if (do_bin && !h->has_bl) return GV_ERR_TF;
  if (do_bbox && !h->has_cl) return GV_ERR_TF;
  if ((fl & GV_FRAME_VISION_ORIENT) && !h->has_bc) return GV_ERR_TF;
  return GV_OK;
}

// --- building blocks of the tile-path frame (shared by the one-GPU frame, the sharded frame and its
// one-device emulation) ---

// partition + tile histogram of points [lo, lo + n) of the current cloud on stream k: that stream's hits[]
// (or not), per-point outputs and binning scratch; the end bitmaps of buffer set p; zeroes the set's
// free-cell bitmaps.  ev_* are stage-timing events or null.
int enqueue_binning(gv_context *h, const DetSet &D, int p, int k, size_t lo, size_t n, bool keep_cell, bool do_ray,
                    bool do_bbox, bool write_hits, hipEvent_t ev_points, Rect *fold_rects = nullptr, bool timed = false,
                    bool any_order = false)
{
  hipStream_t s = h->streams[k];
  const uint32_t chunk = bin_chunk_for(n);
  const uint32_t n_wg = (uint32_t)((n + chunk - 1) / chunk);
  {   // the partition pass writes n_wg table rows and n_wg * chunk keys: never past what was allocated
    size_t keys_need, tab_need;
    bin_needs(h, n, keys_need, tab_need);
    if (keys_need > h->bin_keys_cap || tab_need > h->bin_tab_cap || lo + n > h->idx_cap) {
      h->err = "binning scratch too small for this launch";
      return GV_ERR_STATE;
    }
  }
  BinArgs a{};
  a.x = h->cx + lo; a.y = h->cy + lo; a.z = h->cz + lo;
  a.n = (uint32_t)n;
  a.g = h->g;
  a.m_base = h->m_base;
  a.m_cam = h->m_cam;
  a.cam = h->camk;
  a.org = h->org;
  a.bt = bbox_test_of(h, D);
  a.nb = D.nb;
  a.nb_pad = (D.nb + 3) & ~3;
  a.bbox_id = h->bbox_id_s[k] + lo;
  a.cell_idx = keep_cell ? h->cell_idx_s[k] + lo : nullptr;
  a.do_ray = do_ray;
  // the fused bbox test keeps its tables in LDS; a detection set too large for that (hundreds of boxes, or a
  // large image: one mask word per 16x16-pixel tile) runs the test as a pass of its own over the cloud
  const bool bbox_fused = do_bbox && bin_bbox_fits(D.nb, a.bt);
  a.do_bbox = bbox_fused;
  a.chunk = chunk;
  a.n_wg = n_wg;
  a.tiles_x = h->tiles_x; a.tiles_y = h->tiles_y; a.n_tiles = h->n_tiles;
  a.keys = h->bin_keys[k];
  a.tab = h->bin_tab[k];
  a.tile_total = h->bin_total[k][h->bin_parity[k]];
  if (fold_rects) {   // the frame's rectangles ride the partition launch
    a.rect_poses = D.poses;
    a.n_rect_poses = D.n_poses;
    a.rects_out = fold_rects;
  }
#ifdef GV_DIAG
  a.dbg = h->d_bin_dbg[0];
  a.tl = h->tl_slot(0);
#endif
  launch_bin_partition(a, s, timed ? h->kt[0][0] : nullptr, timed ? h->kt[0][1] : nullptr, any_order);
  h->lane_clean[k] = false;
  if (timed) h->kt_used[0] = n > 0 || fold_rects;
  if (do_bbox && !bbox_fused) {
    PointsArgs pa{};
    pa.x = a.x; pa.y = a.y; pa.z = a.z;
    pa.n = a.n;
    pa.g = h->g;
    pa.m_cam = h->m_cam;
    pa.cam = h->camk;
    pa.bt = a.bt;
    pa.bbox_id = a.bbox_id;
    pa.do_bbox = true;
    launch_points(pa, s);
  }
  if (ev_points) GV_HIP(hipEventRecord(ev_points, s));
  BinTileArgs t{};
  t.nx = h->g.nx; t.ny = h->g.ny;
  t.tiles_x = h->tiles_x; t.tiles_y = h->tiles_y; t.n_tiles = h->n_tiles;
  t.n_wg = n_wg;
  t.chunk = chunk;
  t.keys = h->bin_keys[k];
  t.tab = h->bin_tab[k];
  t.tile_total = h->bin_total[k][h->bin_parity[k]];
  t.tile_total_next = h->bin_total[k][h->bin_parity[k] ^ 1];
  t.done = h->bin_done[k];
  t.scratch = h->bin_scratch[k];
  t.split_keys = kBinSplitKeys;
  t.max_slots = (uint32_t)h->bin_slots;
  t.hits = write_hits ? h->hits_s[k] : nullptr;
  t.hitN = h->x_hitN[p]; t.clipN = h->x_clipN[p]; t.hitT = h->x_hitT[p]; t.clipT = h->x_clipT[p];
  t.freeN = h->x_freeN[p]; t.freeT = h->x_freeT[p];
  t.nxw = h->nxw; t.nyw = h->nyw; t.nx_pad = h->nx_pad; t.ny_pad = h->ny_pad;
#ifdef GV_DIAG
  t.dbg = h->d_bin_dbg[1];
  t.tl = h->tl_slot(1);
#endif
  launch_bin_tiles(t, (uint32_t)(n / kBinSplitKeys), s, timed ? h->kt[1][0] : nullptr, timed ? h->kt[1][1] : nullptr);
  if (timed) h->kt_used[1] = true;
  h->bin_parity[k] ^= 1;
  GV_HIP(hipGetLastError());
  return GV_OK;
}

// sector ray stage over the end bitmaps of set p into its free-cell bitmaps; workgroups first,
// first + stride, ... of the dispatch order (one GPU: 0, 1)
int enqueue_sectors(gv_context *h, int p, int first, int stride, hipStream_t s, hipEvent_t done = nullptr,
                    bool *done_attached = nullptr, hipEvent_t t0 = nullptr)
{
  if (done_attached) *done_attached = false;
  if (!h->org.valid) return GV_OK;
  SectorArgs sa{};
  int rc = fill_sector_args(h, sa, p);
  if (rc) return rc;
  sa.wg_first = first;
  sa.wg_stride = stride;
  const bool launched = launch_ray_sectors(sa, s, done, t0);
  if (done_attached) *done_attached = launched && done;
  GV_HIP(hipGetLastError());
  return GV_OK;
}

int enqueue_grid_pass(gv_context *h, int p, const Rect *rects, int32_t n_rects, bool counts, int32_t y0, int32_t y1,
                      hipStream_t s, hipEvent_t done = nullptr, hipEvent_t t0 = nullptr, bool *launched = nullptr)
{
  FinalizeTileArgs t{};
  t.g = h->g;
  t.log_odds = h->log_odds;
  t.occupancy = h->occupancy;
  t.occ_i8 = h->occ_i8;
  t.rects = rects;
  t.n_rects = n_rects;
  t.hitN = h->x_hitN[p];
  t.freeN = h->x_freeN[p];
  t.freeT = h->x_freeT[p];
  t.nx_pad = h->nx_pad;
  t.ny_pad = h->ny_pad;
  t.counts = counts;
  t.y_begin = y0;
  t.y_end = y1;
#ifdef GV_DIAG
  t.tl = h->tl_slot(3);
#endif
  const bool ran = launch_finalize_tiles(t, s, done, t0);
  if (launched) *launched = ran;
  if (!ran && done) GV_HIP(hipEventRecord(done, s));
  GV_HIP(hipGetLastError());
  return GV_OK;
}

// Stream k (0 public, 1 / 2 the lanes) reads cloud C / detection set D: ordered after their uploads
// (once per upload and stream)
int wait_inputs(gv_context *h, CloudSet &C, DetSet &D, int k)
{
  hipStream_t s = h->streams[k];
  if (!(C.seen >> k & 1u)) {
    GV_HIP(hipStreamWaitEvent(s, C.ready, 0));
    C.seen |= 1u << k;
    h->lane_clean[k] = false;
  }
  if (!(D.seen >> k & 1u)) {
    GV_HIP(hipStreamWaitEvent(s, D.ready, 0));
    D.seen |= 1u << k;
    h->lane_clean[k] = false;
  }
  return GV_OK;
}

// The tile-path frame: rectangles + partition, tile histogram + end bitmaps, sector ray stage back to
// back on one in-order stream, then the grid pass on the public stream.  pipelined: the stream of lane
// n % 2 and buffer set 1 + n % 4 (n = lane frames so far), the grid pass behind one event.  Serial
// (GV_PIPELINE=0, stage timing): everything on the public stream, buffer set 0.  The sharded frame has its own
// enqueue (enqueue_frame_sharded).
int enqueue_frame_tiles(gv_context *h, bool pipelined, bool stage_events)
{
  DetSet &D = h->det[h->det_cur];
  const uint32_t fl = D.flags;
  const bool do_bin = fl & GV_FRAME_BIN, do_ray = fl & GV_FRAME_RAYMARCH, do_bbox = fl & GV_FRAME_BBOX_TEST;
  const bool keep_cell = fl & GV_FRAME_KEEP_CELL_IDX;
  int rc = check_frame_flags(h, fl);
  if (rc) return rc;
  const int p = pipelined ? 1 + (int)(h->lane_frames % (uint64_t)(2 * h->n_lanes)) : 0;
  const int k = pipelined ? 1 + (int)(h->lane_frames % (uint64_t)h->lanes_now()) : 0;
  hipStream_t s = h->streams[k];
  // back-pressure: the frame that last used this buffer set (four frames ago) has finished
  if (pipelined && h->set_fin_slot[p] >= 0) GV_HIP(hipEventSynchronize(h->ev_fin[h->set_fin_slot[p]]));
  CloudSet &CS = h->cloud[h->cloud_cur];
#ifdef GV_DIAG
  auto mark = [&](hipStream_t st) {   // device timeline of the pipelined frame (gv_debug_pipeline_trace)
    if (!h->trace) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    h->trace->push_back(e);
  };
#else
  auto mark = [](hipStream_t) {};
#endif
  if ((rc = wait_inputs(h, CS, D, k))) return rc;
  if (stage_events) {
    GV_HIP(hipEventRecord(h->ev[0], s));
    for (bool &u : h->kt_used) u = false;
  }

  // --- detections -> rectangles.  Base-frame poses of a binning frame ride the partition launch (one
  // extra workgroup) instead of a launch of their own; network outputs go through the vision kernels.
  Rect *rects = h->x_rects[p];
  const bool fold_rects = do_bin && !(fl & GV_FRAME_VISION_ORIENT) && D.n_poses > 0;
  mark(s);
  if (!fold_rects) h->lane_clean[k] = false;   // (the rectangle / vision kernels go on the lane)
  const int32_t n_rects = fold_rects ? D.n_poses : enqueue_rects(h, D, rects, h->d_vout_s[k], s);
  mark(s);
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageDetections + 1], s));
  bool part_any_order = pipelined && !stage_events && h->env_anyorder && h->lane_clean[k];
#ifdef GV_DIAG
  if (h->trace) part_any_order = false;   // the trace markers are packets on the lane
#endif

  // --- points: partition by tile (+ ray ends, bbox test), then the tile histogram: hits[] + end bitmaps
  mark(s);
  if (do_bin) {
    if ((rc = enqueue_binning(h, D, p, k, 0, h->n, keep_cell, do_ray, do_bbox, true,
                              stage_events ? h->ev[kStagePoints + 1] : nullptr, fold_rects ? rects : nullptr, stage_events,
                              part_any_order)))
      return rc;
  } else {
    h->lane_clean[k] = false;
    if (do_bbox) {
      PointsArgs a{};
      a.x = h->cx; a.y = h->cy; a.z = h->cz;
      a.n = (uint32_t)h->n;
      a.g = h->g;
      a.m_cam = h->m_cam;
      a.cam = h->camk;
      a.bt = bbox_test_of(h, D);
      a.bbox_id = h->bbox_id_s[k];
      a.do_bbox = true;
      launch_points(a, s);
    }
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStagePoints + 1], s));
  }
  mark(s);
  mark(s); mark(s);   // (trace slot of the former bitmap kernel: the tile pass is part of the binning pair)
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayCompact + 1], s));
  h->last_set = p;
  h->hits = h->hits_s[k];
  h->bbox_id = h->bbox_id_s[k];
  h->cell_idx = h->cell_idx_s[k];
  h->have_cell_idx = do_bin && keep_cell;
  h->have_bbox_id = do_bbox;

  // --- free-space ray stage.  On a lane its completion event rides the kernel's own dispatch packet.
  const int slot = (int)(h->frame_no % (uint64_t)gv_context::kRing);
  bool sec_event = false;
  mark(s);
  if (do_ray && (rc = enqueue_sectors(h, p, 0, 1, s, pipelined ? h->ev_sec[slot] : (stage_events ? h->kt[2][1] : nullptr), &sec_event,
                                      stage_events ? h->kt[2][0] : nullptr)))
    return rc;
  if (stage_events) h->kt_used[2] = sec_event;
  // the lane now ends in a sector kernel that carries its own completion event: nothing behind it
  h->lane_clean[k] = pipelined && do_bin && do_ray && sec_event;
  mark(s);
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayMarch + 1], s));

  // --- grid pass, on the public stream: in order behind the previous frame's and behind whatever the
  // caller queued there (the download of the previous grid, a plain map update)
  if (pipelined) {
    if (!sec_event) GV_HIP(hipEventRecord(h->ev_sec[slot], s));
    GV_HIP(hipStreamWaitEvent(h->stream, h->ev_sec[slot], 0));
    s = h->stream;
  }

  mark(s);
  // ev_fin[slot] completes with the grid pass: this frame done => every earlier frame done
  if (stage_events) {   // stage timing: the kernel carries its own start / end events, ev_fin follows as a marker
    bool ran = false;
    if ((rc = enqueue_grid_pass(h, p, rects, n_rects, do_bin, 0, h->g.ny, s, h->kt[3][1], h->kt[3][0], &ran))) return rc;
    h->kt_used[3] = ran;
    GV_HIP(hipEventRecord(h->ev_fin[slot], s));
  } else if ((rc = enqueue_grid_pass(h, p, rects, n_rects, do_bin, 0, h->g.ny, s, h->ev_fin[slot]))) return rc;
  mark(s);
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageFinalize + 1], s));
  // cloud, detection set and buffer set remember their last user
  h->last_fin_slot = slot;
  h->set_fin_slot[p] = slot;
  CS.release_slot = slot;
  D.release_slot = slot;
  D.readers |= 1u << k;
  h->frame_no++;
  if (pipelined) {
    h->lane_frames++;
    h->pipe_busy = true;
    if (h->quiet_frames < 0x7fffffffu) h->quiet_frames++;
  }
  h->have_hits = do_bin;
  h->have_miss = do_bin;   // the free-cell bitmaps of set p stay until the set's next frame
  return GV_OK;
}

// Generic frame: any grid shape (nx % 4 != 0, more than 8000 cells per side, GV_RAY_IMPL=simple).
// One stream; atomics-based count grids; literal per-ray march.
int enqueue_frame_generic(gv_context *h, bool stage_events)
{
  DetSet &D = h->det[h->det_cur];
  const uint32_t fl = D.flags;
  const bool do_bin = fl & GV_FRAME_BIN, do_ray = fl & GV_FRAME_RAYMARCH, do_bbox = fl & GV_FRAME_BBOX_TEST;
  const bool keep_cell = fl & GV_FRAME_KEEP_CELL_IDX, keep_counts = fl & GV_FRAME_KEEP_COUNTS;
  int rc = check_frame_flags(h, fl);
  if (rc) return rc;
  if (h->counts_dirty && (rc = clear_counts(h))) return rc;
  hipStream_t s = h->stream;
  CloudSet &CS = h->cloud[h->cloud_cur];
  if ((rc = wait_inputs(h, CS, D, 0))) return rc;
  if (stage_events) GV_HIP(hipEventRecord(h->ev[0], s));
  const int32_t n_rects = enqueue_rects(h, D, h->x_rects[0], h->d_vout_s[0], s);
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageDetections + 1], s));
  if (do_bin || do_bbox) {
    PointsArgs a{};
    a.x = h->cx; a.y = h->cy; a.z = h->cz;
    a.n = (uint32_t)h->n;
    a.g = h->g;
    a.m_base = h->m_base;
    a.m_cam = h->m_cam;
    a.cam = h->camk;
    a.org = h->org;
    a.bt = bbox_test_of(h, D);
    a.hits = h->hits;
    a.clip_end = h->clip_end;
    a.cell_idx = keep_cell ? h->cell_idx : nullptr;
    a.bbox_id = h->bbox_id;
    a.do_bin = do_bin; a.do_ray = do_ray; a.do_bbox = do_bbox;
    launch_points(a, s);
  }
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStagePoints + 1], s));
  if (do_ray && h->org.valid) {
    GV_HIP(hipMemsetAsync(h->ray_count, 0, sizeof(uint32_t), s));
    GV_HIP(hipMemsetAsync(h->x_stats[0], 0, 2 * sizeof(unsigned long long), s));
    h->stat_slots = 1;
    launch_ray_compact(h->hits, h->clip_end, h->g, h->ray_list, h->ray_count, s);
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayCompact + 1], s));
    launch_ray_march(h->ray_list, h->ray_count, h->g, h->org, h->miss8, h->x_stats[0], s);
    if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageRayMarch + 1], s));
  } else if (stage_events) {
    GV_HIP(hipEventRecord(h->ev[kStageRayCompact + 1], s));
    GV_HIP(hipEventRecord(h->ev[kStageRayMarch + 1], s));
  }
  FinalizeArgs f{};
  f.g = h->g;
  f.log_odds = h->log_odds;
  f.occupancy = h->occupancy;
  f.occ_i8 = h->occ_i8;
  f.rects = h->x_rects[0];
  f.n_rects = n_rects;
  f.hits = do_bin ? h->hits : nullptr;
  f.miss = h->miss8;
  f.clip_end = h->clip_end;
  f.zero_counts = do_bin && !keep_counts;
  f.cell_begin = 0;
  f.cell_end = h->g.G;
  launch_finalize(f, s);
  if (stage_events) GV_HIP(hipEventRecord(h->ev[kStageFinalize + 1], s));
  GV_HIP(hipGetLastError());
  const int slot = (int)(h->frame_no % (uint64_t)gv_context::kRing);
  GV_HIP(hipEventRecord(h->ev_fin[slot], s));   // cloud and detection set remember their last reader
  h->last_fin_slot = slot;
  CS.release_slot = slot;
  D.release_slot = slot;
  D.readers |= 1u;
  h->frame_no++;
  h->last_set = 0;
  h->counts_dirty = do_bin && keep_counts;
  h->have_hits = h->have_miss = do_bin && keep_counts;
  h->have_cell_idx = do_bin && keep_cell;
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

// --- the frame sharded by points (BASELINE configs[4]) ---
// Every rank bins ITS slice of the cloud into private end bitmaps.  Two exchanges follow, both of the
// form "all-to-all of equal slices + local OR" (RCCL has no bitwise-OR reduction; the slices are bitmap
// words, 32 cells per word):
//   1. ray ends: the OR-ed slices are all-gathered, so every rank holds the complete end bitmaps and
//      runs only every world-th workgroup of the sector ray stage (the dispatch order is sorted by
//      expected cost, so the shares are balanced);
//   2. free cells: each rank's partial free-cell bitmaps are packed by row band and rank q receives
//      and ORs band q.
// Rank q then runs the grid pass on band q (whole 64-row blocks) and the packed int8 bands are
// broadcast.  OR and integer sums commute: the result is bit-identical to one GPU.
// The exchanges are expressed over `ShardLink`, which is RCCL in production and a set of device
// copies in the one-device emulation that the tests use to run every (rank, world).
struct ShardLink {
  gv_context *h;
  int rank, world;
  // emulation: the `world` per-rank source buffers of the current exchange (null with RCCL)
  uint32_t *const *emu_src = nullptr;
};

// recv[q'] (count words each) <- slice `rank` of peer q'; send holds `world` slices of count words
int shard_all_to_all(const ShardLink &L, const uint32_t *send, uint32_t *recv, size_t count, hipStream_t s)
{
  gv_context *h = L.h;
  if (L.emu_src) {
    for (int q = 0; q < L.world; ++q)
      GV_HIP(hipMemcpyAsync(recv + (size_t)q * count, L.emu_src[q] + (size_t)L.rank * count, count * sizeof(uint32_t),
                            hipMemcpyDeviceToDevice, s));
    return GV_OK;
  }
  ncclResult_t first_err = ncclGroupStart();
  for (int q = 0; q < L.world && first_err == ncclSuccess; ++q) {
    if (q == L.rank) continue;
    ncclResult_t r = ncclSend(send + (size_t)q * count, count, ncclUint32, q, h->comm, s);
    if (r == ncclSuccess) r = ncclRecv(recv + (size_t)q * count, count, ncclUint32, q, h->comm, s);
    if (r != ncclSuccess) first_err = r;
  }
  const ncclResult_t ge = ncclGroupEnd();   // always closed, also on the error path
  if (first_err == ncclSuccess) first_err = ge;
  if (first_err != ncclSuccess) {
    h->err = std::string("sharded all-to-all -> ") + ncclGetErrorString(first_err);
    return GV_ERR_RCCL;
  }
  GV_HIP(hipMemcpyAsync(recv + (size_t)L.rank * count, send + (size_t)L.rank * count, count * sizeof(uint32_t),
                        hipMemcpyDeviceToDevice, s));
  return GV_OK;
}

size_t shard_ends_slice(const gv_context *h, int world)
{
  return (size_t)gv_shard_slice_words((int64_t)h->ends_words, world);
}

int ensure_shard_scratch(gv_context *h, int world)
{
  const size_t chunk = free_band_chunk_words(h->nxw, h->nx_pad, h->ny_pad, world);
  const size_t need = std::max(shard_ends_slice(h, world) * (size_t)world, 2 * chunk * (size_t)world) + 16;
  return grow(h, h->sh_xchg, h->sh_xchg_cap, need);
}

// exchange 1 (this rank's part): OR of everyone's slice `rank` of the end bitmaps, written back in place
int shard_or_ends_slice(const ShardLink &L, uint32_t *ends, hipStream_t s)
{
  gv_context *h = L.h;
  const size_t slice = shard_ends_slice(h, L.world);
  int rc = shard_all_to_all(L, ends, h->sh_xchg, slice, s);
  if (rc) return rc;
  launch_or_slices(h->sh_xchg, ends + (size_t)L.rank * slice, slice, L.world, s);
  GV_HIP(hipGetLastError());
  return GV_OK;
}

// exchange 2 (this rank's part): band `rank` of everyone's free-cell bitmaps OR-ed into set p
int shard_or_free_band(const ShardLink &L, int p, const uint32_t *packed, hipStream_t s)
{
  gv_context *h = L.h;
  const size_t chunk = free_band_chunk_words(h->nxw, h->nx_pad, h->ny_pad, L.world);
  uint32_t *recv = h->sh_xchg + chunk * (size_t)L.world;
  int rc = shard_all_to_all(L, packed, recv, chunk, s);
  if (rc) return rc;
  launch_unpack_free_band(recv, L.world, chunk, L.rank, h->nxw, h->nx_pad, h->ny_pad, h->x_freeN[p], h->x_freeT[p], s);
  GV_HIP(hipGetLastError());
  return GV_OK;
}

// The asynchronous sharded frame.  Three queues work on it: the frame's lane (binning, this rank's share of the
// sector stage, band packing), the exchange stream X (RCCL: ends exchange, free-band exchange, band broadcast,
// count reduce) and the public stream (the band's grid pass -- grid passes stay one in-order sequence).  Events
// chain the steps of ONE frame; nothing orders frame f + 1's binning (the other lane) behind frame f's
// exchanges, so they overlap.  RCCL calls are issued on X in the same order on every rank (x1, x2, x3 of frame
// f, then of f + 1).  te (optional, 7 timing events): start, binning, x1, sectors, x2, grid pass, x3 done.
int enqueue_frame_sharded(gv_context *h, hipEvent_t *te)
{
  DetSet &D = h->det[h->det_cur];
  const uint32_t fl = D.flags;
  const bool do_bin = fl & GV_FRAME_BIN, do_ray = fl & GV_FRAME_RAYMARCH, do_bbox = fl & GV_FRAME_BBOX_TEST;
  const bool keep_cell = fl & GV_FRAME_KEEP_CELL_IDX, keep_counts = fl & GV_FRAME_KEEP_COUNTS;
  int rc = check_frame_flags(h, fl);
  if (rc) return rc;
  if (!do_bin || !h->comm || !h->stream_x) return GV_ERR_STATE;
  const int p = 1 + (int)(h->lane_frames % 4u);
  const int k = 1 + (int)(h->lane_frames % 2u);
  hipStream_t s = h->streams[k], X = h->stream_x;
  h->lane_clean[1] = h->lane_clean[2] = false;   // events between the steps: every kernel of this form keeps its barrier bit
  if (h->set_fin_slot[p] >= 0) GV_HIP(hipEventSynchronize(h->ev_fin[h->set_fin_slot[p]]));   // back-pressure: four frames in flight
  CloudSet &CS = h->cloud[h->cloud_cur];
  if ((rc = wait_inputs(h, CS, D, k))) return rc;
  if ((rc = ensure_shard_scratch(h, h->world))) return rc;
  const int slot = (int)(h->frame_no % (uint64_t)gv_context::kRing);
  hipEvent_t *ev = h->ev_sh[slot];
  // Step x3 of a KEEP_COUNTS frame reduces hits_s[k] IN PLACE on the exchange stream, and nothing else orders this
  // lane's next tile pass -- which rewrites every cell of hits_s[k] -- behind it (the buffer-set back-pressure is
  // four frames deep, the lane comes round every second frame).  The lane waits for that frame's last exchange
  // (round-3 advisor finding; test_sharded_keep_counts_frames_in_flight).
  if (h->sh_counts_slot[k] >= 0) {
    GV_HIP(hipStreamWaitEvent(s, h->ev_fin[h->sh_counts_slot[k]], 0));
    h->sh_counts_slot[k] = -1;
  }
  if (te) GV_HIP(hipEventRecord(te[0], s));
  // --- lane: rectangles + binning of this rank's points into private end bitmaps
  Rect *rects = h->x_rects[p];
  const bool fold_rects = !(fl & GV_FRAME_VISION_ORIENT) && D.n_poses > 0;
  const int32_t n_rects = fold_rects ? D.n_poses : enqueue_rects(h, D, rects, h->d_vout_s[k], s);
  if ((rc = enqueue_binning(h, D, p, k, 0, h->n, keep_cell, do_ray, do_bbox, keep_counts, nullptr, fold_rects ? rects : nullptr)))
    return rc;
  if (te) GV_HIP(hipEventRecord(te[1], s));
  GV_HIP(hipEventRecord(ev[0], s));
  // --- X: complete end bitmaps everywhere (slices all-to-all + OR, then all-gather)
  ShardLink L{h, h->rank, h->world};
  const size_t slice = shard_ends_slice(h, h->world);
  GV_HIP(hipStreamWaitEvent(X, ev[0], 0));
  if ((rc = shard_or_ends_slice(L, h->x_ends[p], X))) return rc;
  GV_NCCL(ncclAllGather(h->x_ends[p] + (size_t)h->rank * slice, h->x_ends[p], slice, ncclUint32, h->comm, X));
  if (te) GV_HIP(hipEventRecord(te[2], X));
  GV_HIP(hipEventRecord(ev[1], X));
  // --- lane: this rank's share of the ray stage, its free cells packed by band
  GV_HIP(hipStreamWaitEvent(s, ev[1], 0));
  if (do_ray && (rc = enqueue_sectors(h, p, h->rank, h->world, s))) return rc;
  const size_t chunk = free_band_chunk_words(h->nxw, h->nx_pad, h->ny_pad, h->world);
  launch_pack_free_bands(h->x_freeN[p], h->x_freeT[p], h->nxw, h->nx_pad, h->ny_pad, h->world, chunk, h->sh_xchg, s);
  GV_HIP(hipGetLastError());
  if (te) GV_HIP(hipEventRecord(te[3], s));
  GV_HIP(hipEventRecord(ev[2], s));
  // --- X: the free cells of MY band from everyone
  GV_HIP(hipStreamWaitEvent(X, ev[2], 0));
  if ((rc = shard_or_free_band(L, p, h->sh_xchg, X))) return rc;
  if (te) GV_HIP(hipEventRecord(te[4], X));
  GV_HIP(hipEventRecord(ev[3], X));
  // --- public stream: grid pass on the band (whole 64-row blocks)
  int32_t y0, y1;
  shard_band_rows(h->rank, h->world, h->g.ny, h->ny_pad, y0, y1);
  GV_HIP(hipStreamWaitEvent(h->stream, ev[3], 0));
  if ((rc = enqueue_grid_pass(h, p, rects, n_rects, true, y0, y1, h->stream))) return rc;
  if (te) GV_HIP(hipEventRecord(te[5], h->stream));
  GV_HIP(hipEventRecord(ev[4], h->stream));
  // --- X: packed bands to everyone (band r sits at data[G - e_r, G - b_r)); band totals of the hit counts
  GV_HIP(hipStreamWaitEvent(X, ev[4], 0));
  const size_t G = (size_t)h->g.G;
  ncclResult_t first_err = ncclGroupStart();
  for (int r = 0; r < h->world && first_err == ncclSuccess; ++r) {
    int32_t r0, r1;
    shard_band_rows(r, h->world, h->g.ny, h->ny_pad, r0, r1);
    const size_t b = (size_t)r0 * h->g.nx, e = (size_t)r1 * h->g.nx;
    if (e > b) {
      const ncclResult_t br = ncclBroadcast(h->occ_i8 + (G - e), h->occ_i8 + (G - e), e - b, ncclInt8, r, h->comm, X);
      if (br != ncclSuccess) first_err = br;
    }
  }
  ncclResult_t ge = ncclGroupEnd();   // always closed, also on the error path
  if (first_err == ncclSuccess) first_err = ge;
  if (first_err == ncclSuccess && keep_counts) {
    // reduce-scatter by band -- rank q ends with the summed counts of band q (in place, at the
    // band's rows of its hits[]; the other rows keep this rank's partial counts).  Bands are whole 64-row blocks
    // and may differ in length: equal bands are one ncclReduceScatter, otherwise one grouped ncclReduce per band.
    int32_t *hk = h->hits_s[k];
    bool equal = true;
    size_t cnt0 = 0;
    for (int r = 0; r < h->world; ++r) {
      int32_t r0, r1;
      shard_band_rows(r, h->world, h->g.ny, h->ny_pad, r0, r1);
      const size_t c = (size_t)(r1 - r0) * h->g.nx;
      if (r == 0) cnt0 = c;
      equal = equal && c == cnt0 && (size_t)r0 * h->g.nx == (size_t)r * cnt0;
    }
    if (equal && cnt0) {
      first_err = ncclReduceScatter(hk, hk + (size_t)h->rank * cnt0, cnt0, ncclInt32, ncclSum, h->comm, X);
    } else {
      first_err = ncclGroupStart();
      for (int r = 0; r < h->world && first_err == ncclSuccess; ++r) {
        int32_t r0, r1;
        shard_band_rows(r, h->world, h->g.ny, h->ny_pad, r0, r1);
        const size_t b = (size_t)r0 * h->g.nx, e = (size_t)r1 * h->g.nx;
        if (e > b) first_err = ncclReduce(hk + b, hk + b, e - b, ncclInt32, ncclSum, r, h->comm, X);
      }
      ge = ncclGroupEnd();
      if (first_err == ncclSuccess) first_err = ge;
    }
  }
  if (first_err != ncclSuccess) {
    h->err = std::string("sharded band exchange -> ") + ncclGetErrorString(first_err);
    return GV_ERR_RCCL;
  }
  if (te) GV_HIP(hipEventRecord(te[6], X));
  GV_HIP(hipEventRecord(h->ev_fin[slot], X));
  // what the frame produced (the gathered packed grid) is visible on the public stream right behind it
  GV_HIP(hipStreamWaitEvent(h->stream, h->ev_fin[slot], 0));
  h->last_fin_slot = slot;
  h->set_fin_slot[p] = slot;
  CS.release_slot = slot;
  D.release_slot = slot;
  D.readers |= (1u << k) | 1u;
  h->frame_no++;
  h->lane_frames++;
  h->pipe_busy = true;
  h->last_set = p;
  h->hits = h->hits_s[k];
  h->bbox_id = h->bbox_id_s[k];
  h->cell_idx = h->cell_idx_s[k];
  h->have_cell_idx = keep_cell;
  h->have_bbox_id = do_bbox;
  if (keep_counts) h->sh_counts_slot[k] = slot;
  h->have_hits = keep_counts;   // band totals at this rank's band rows (gv_comm_band)
  h->have_miss = false;         // free-cell bitmaps are complete for this rank's band only
  return GV_OK;
}

}  // namespace

extern "C" {

int gv_abi_version(void) { return 4; }

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
  h->bt_tiles_x = std::max(1, (cam->orig_w + 15) / 16);
  h->bt_tiles_y = std::max(1, (cam->orig_h + 15) / 16);

  auto fail = [&](int code) { gv_destroy(h); return code; };
#define GV_C(call)                                           \
  do {                                                       \
    if ((call) != hipSuccess) return fail(GV_ERR_HIP);       \
  } while (0)
  GV_C(hipSetDevice(h->device));
  GV_C(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  GV_C(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  GV_C(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
  GV_C(hipStreamCreateWithFlags(&h->stream_copy, hipStreamNonBlocking));
  if (const char *e = std::getenv("GV_LANES")) h->n_lanes = (std::atoi(e) == 2) ? 2 : 3;
  // (GV_LANE3_OWN_STREAM=1, experiment: the third lane on a fifth stream instead of the upload stream)
  if (h->n_lanes == 3 && std::getenv("GV_LANE3_OWN_STREAM")) GV_C(hipStreamCreateWithFlags(&h->stream4, hipStreamNonBlocking));
  // The upload stream must not share a hardware queue with the public stream or a lane (a process gets four
  // queues; a stream created when four exist joins the one with the fewest streams, ties by address -- e.g. a host
  // application or framework that owns a stream already pushes one of ours onto a shared queue, and when that is
  // the upload stream every cloud waits behind kernels: the 0.55-0.8-of-copy-rate regime of profiles/r03/h2d_notes.md).
  // Probe: hold the three compute streams busy for 150 us each, time a 4-byte memset on the upload stream; if it had
  // to wait, make another upload stream (before letting go of this one, so that it lands elsewhere) and try again.
  if (!(std::getenv("GV_QUEUE_PROBE") && std::atoi(std::getenv("GV_QUEUE_PROBE")) == 0)) {
    unsigned *probe = nullptr;
    GV_C(hipMalloc(reinterpret_cast<void **>(&probe), 256));
    std::vector<hipStream_t> rejected;
    launch_hold(1ull, h->stream);   // (the kernel's code object is loaded before anything is timed)
    // Only the handle's own streams are synchronised (a device-wide wait would stall on, and be perturbed by, every
    // other handle or application stream of the process); a stream's hardware queue is created on its first use, so
    // one untimed memset goes first.  Costs 0.3-1 ms per gv_create; GV_QUEUE_PROBE=0 skips it.
    auto sync_own = [&]() -> hipError_t {
      for (hipStream_t q : {h->stream, h->stream2, h->stream3, h->stream_copy}) {
        const hipError_t e = hipStreamSynchronize(q);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    };
    for (int attempt = 0; attempt < 6; ++attempt) {
      GV_C(hipMemsetAsync(probe, 0, 4, h->stream_copy));   // untimed: the queue exists afterwards
      GV_C(sync_own());
      for (hipStream_t q : {h->stream, h->stream2, h->stream3}) launch_hold(15000ull, q);   // 150 us at 100 MHz
      const auto t0 = std::chrono::steady_clock::now();
      GV_C(hipMemsetAsync(probe, 0, 4, h->stream_copy));
      GV_C(hipStreamSynchronize(h->stream_copy));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      GV_C(sync_own());
      h->upload_probe_us = us;
      if (us < 90.0) break;
      h->upload_stream_retries++;
      hipStream_t nw = nullptr;
      GV_C(hipStreamCreateWithFlags(&nw, hipStreamNonBlocking));
      rejected.push_back(h->stream_copy);
      h->stream_copy = nw;
    }
    for (hipStream_t q : rejected) (void)hipStreamDestroy(q);
    (void)hipFree(probe);
    if (std::getenv("GV_VERBOSE"))
      std::fprintf(stderr, "gridvision_hip: upload stream probe %.0f us, %d replacement(s)\n", h->upload_probe_us, h->upload_stream_retries);
  }
  h->streams[0] = h->stream;
  h->streams[1] = h->stream2;
  h->streams[2] = h->stream3;
  h->streams[3] = h->stream4 ? h->stream4 : h->stream_copy;
  // Ordering-only events between queues of this device (and a completion flag the host polls): nobody reads
  // memory on the strength of them -- results are read in stream order on the public stream or after a
  // stream synchronise -- so the kernels that carry them need no system-scope release at their end.
  for (auto &e : h->ev_fin) GV_C(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  for (auto &e : h->ev_sec) GV_C(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  GV_C(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  GV_C(hipEventCreateWithFlags(&h->tick.done, hipEventDisableTiming));
  GV_C(hipEventCreateWithFlags(&h->tick.fork, hipEventDisableTiming));
  GV_C(hipEventCreateWithFlags(&h->tick.join, hipEventDisableTiming));
  if (const char *e = std::getenv("GV_TICK_KNN_LANE")) h->env_tick_knn_lane = std::atoi(e) != 0;
  for (auto &c : h->cloud) GV_C(hipEventCreateWithFlags(&c.ready, hipEventDisableTiming));
  for (auto &d : h->det) GV_C(hipEventCreateWithFlags(&d.ready, hipEventDisableTiming));
  const size_t G = (size_t)g.G;
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->log_odds), G * sizeof(float)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->occupancy), G * sizeof(float)));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->occ_i8), G));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->ray_count), 4 * sizeof(uint32_t)));
  GV_C(hipMemsetAsync(h->ray_count, 0, 4 * sizeof(uint32_t), h->stream));
  GV_C(hipMalloc(reinterpret_cast<void **>(&h->scratch_i32), G * sizeof(int32_t)));
  h->scratch_cap = G;
  // packed (a,b) fields hold 13 bits each; vector stores need nx % 4 == 0
  h->tile_path = (g.nx % 4 == 0) && g.nx <= 8000 && g.ny <= 8000;
  {
    const char *impl = std::getenv("GV_RAY_IMPL");
    h->force_simple = impl && std::strcmp(impl, "simple") == 0;
    if (const char *e = std::getenv("GV_PIPELINE")) h->no_pipeline = std::atoi(e) == 0;
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
    if (const char *e = std::getenv("GV_SECTOR_HELPERS")) h->env_helpers = std::atoi(e) != 0;
    if (const char *e = std::getenv("GV_ANYORDER")) h->env_anyorder = std::atoi(e) != 0;
    if (const char *e = std::getenv("GV_CAP")) h->env_cap = std::atoi(e);
    if (const char *e = std::getenv("GV_FLAT_K")) h->env_flat_k = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("GV_FLAT_DIRECT")) h->env_flat_direct = (uint32_t)std::max(0, std::atoi(e));
    if (const char *e = std::getenv("GV_MARCH_LIMIT")) h->env_march_limit = (uint32_t)std::max(0, std::atoi(e));
    if (const char *e = std::getenv("GV_LOG2M")) h->env_log2m = std::min(9, std::max(4, std::atoi(e)));
    if (const char *e = std::getenv("GV_SECTOR_REV")) h->env_sector_rev = std::atoi(e);
#ifdef GV_DIAG
    if (const char *e = std::getenv("GV_ABLATE")) h->env_ablate = std::atoi(e);
    if (const char *e = std::getenv("GV_BIN_DBG")) {
      if (std::atoi(e) > 0)
        for (auto &q : h->d_bin_dbg) {
          GV_C(hipMalloc(reinterpret_cast<void **>(&q), 8192 * 16 * sizeof(unsigned long long)));
          GV_C(hipMemsetAsync(q, 0, 8192 * 16 * sizeof(unsigned long long), h->stream));
        }
    }
    if (const char *e = std::getenv("GV_TIMELINE")) {
      if (std::atoi(e) > 0) {
        const size_t nt = gv_context::kTlFrames * 8;
        GV_C(hipMalloc(reinterpret_cast<void **>(&h->d_tl), nt * sizeof(unsigned long long)));
        std::vector<unsigned long long> init(nt);
        for (size_t i = 0; i < nt; i += 2) { init[i] = ~0ull; init[i + 1] = 0ull; }
        GV_C(hipMemcpy(h->d_tl, init.data(), nt * sizeof(unsigned long long), hipMemcpyHostToDevice));
      }
    }
    if (const char *e = std::getenv("GV_SECTOR_DBG")) {
      if (std::atoi(e) > 0) {
        GV_C(hipMalloc(reinterpret_cast<void **>(&h->d_dbg), kMaxStatSlots * 16 * sizeof(unsigned long long)));
        GV_C(hipMemsetAsync(h->d_dbg, 0, kMaxStatSlots * 16 * sizeof(unsigned long long), h->stream));
      }
    }
#endif
  }
  const bool sectors = h->tile_path && !h->force_simple;
  const int nsets_alloc = sectors ? gv_context::kSets : 1;
  for (int k = 0; k < (sectors ? 1 + h->n_lanes : 1); ++k) {
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->hits_s[k]), G * sizeof(int32_t)));
    GV_C(hipMemsetAsync(h->hits_s[k], 0, G * sizeof(int32_t), h->stream));
  }
  h->hits = h->hits_s[0];
  for (int k = 0; k < nsets_alloc; ++k) {
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_stats[k]), kMaxStatSlots * 2 * sizeof(unsigned long long)));
    GV_C(hipMemsetAsync(h->x_stats[k], 0, kMaxStatSlots * 2 * sizeof(unsigned long long), h->stream));
  }
  // bitmaps: padded to whole binning tiles, so that every word belongs to exactly one tile
  h->nx_pad = kBinTile * ((g.nx + kBinTile - 1) / kBinTile);
  h->ny_pad = kBinTile * ((g.ny + kBinTile - 1) / kBinTile);
  h->nxw = h->nx_pad / 32;
  h->nyw = h->ny_pad / 32;
  h->tiles_x = h->nx_pad / kBinTile;
  h->tiles_y = h->ny_pad / kBinTile;
  h->n_tiles = h->tiles_x * h->tiles_y;
  if (sectors) {
    h->bmN_words = (size_t)h->ny_pad * h->nxw;   // multiples of 4 words (pads are multiples of 128)
    h->bmT_words = (size_t)h->nx_pad * h->nyw;
    h->ends_words = 2 * (h->bmN_words + h->bmT_words);
    for (int k = 0; k < gv_context::kSets; ++k) {
      // + slack: the sharded exchange pads the buffer to `world` equal slices
      GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_ends[k]), (h->ends_words + 1024) * sizeof(uint32_t)));
      GV_C(hipMemsetAsync(h->x_ends[k], 0, (h->ends_words + 1024) * sizeof(uint32_t), h->stream));
      h->x_hitN[k] = h->x_ends[k];
      h->x_clipN[k] = h->x_hitN[k] + h->bmN_words;
      h->x_hitT[k] = h->x_clipN[k] + h->bmN_words;
      h->x_clipT[k] = h->x_hitT[k] + h->bmT_words;
      GV_C(hipMalloc(reinterpret_cast<void **>(&h->x_free[k]), (h->bmN_words + h->bmT_words + 16) * sizeof(uint32_t)));
      GV_C(hipMemsetAsync(h->x_free[k], 0, (h->bmN_words + h->bmT_words + 16) * sizeof(uint32_t), h->stream));
      h->x_freeN[k] = h->x_free[k];
      h->x_freeT[k] = h->x_free[k] + h->bmN_words;
    }
    for (int q = 0; q < 1 + h->n_lanes; ++q) {
      for (int k = 0; k < 2; ++k) {
        GV_C(hipMalloc(reinterpret_cast<void **>(&h->bin_total[q][k]), (size_t)h->n_tiles * sizeof(uint32_t)));
        GV_C(hipMemsetAsync(h->bin_total[q][k], 0, (size_t)h->n_tiles * sizeof(uint32_t), h->stream));
      }
      GV_C(hipMalloc(reinterpret_cast<void **>(&h->bin_done[q]), (size_t)h->n_tiles * sizeof(uint32_t)));
      GV_C(hipMemsetAsync(h->bin_done[q], 0, (size_t)h->n_tiles * sizeof(uint32_t), h->stream));
    }
  } else {
    // generic path: byte flags of clipped ray ends and of free cells + the compacted ray list
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->clip_end), G + 16));
    GV_C(hipMemsetAsync(h->clip_end, 0, G + 16, h->stream));
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->miss8), G + 16));
    GV_C(hipMemsetAsync(h->miss8, 0, G + 16, h->stream));
    GV_C(hipMalloc(reinterpret_cast<void **>(&h->ray_list), G * sizeof(uint32_t)));
  }
  for (auto &e : h->ev) GV_C(hipEventCreate(&e));
  for (auto &pr : h->kt)
    for (auto &e : pr) GV_C(hipEventCreate(&e));
#undef GV_C
  if (ensure_det_shared(h, 64) != GV_OK) return fail(GV_ERR_HIP);
  for (auto &d : h->det)
    if (ensure_det(h, d, 64) != GV_OK) return fail(GV_ERR_HIP);
  if (ensure_point_buffers(h, 0) != GV_OK) return fail(GV_ERR_HIP);
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
  for (hipStream_t s : {h->stream_copy, h->stream, h->stream2, h->stream3, h->stream4, h->stream_x})
    if (s) (void)hipStreamSynchronize(s);
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  for (auto &row : h->ev_sh)
    for (auto &e : row)
      if (e) (void)hipEventDestroy(e);
  for (auto &e : h->sh_t)
    if (e) (void)hipEventDestroy(e);
  if (h->stream_x) (void)hipStreamDestroy(h->stream_x);
  void *bufs[] = {h->log_odds, h->occupancy, h->occ_i8, h->clip_end, h->miss8, h->sh_xchg, h->ray_list, h->ray_count, h->scratch_i32,
                  h->tx, h->ty, h->tz, h->d_pts, h->knn_partial,
                  h->d_nodes, h->d_keep, h->d_ticket_of, h->d_pca_acc, h->d_pca_ext, h->d_pca_ticket, h->d_cellcnt, h->d_cellpre, h->d_celloff, h->d_planes,
                  h->d_plane_counts, h->d_ground, h->d_rscratch, h->d_rstate, h->d_res_ticket};
  for (void *p : bufs)
    if (p) (void)hipFree(p);
  if (h->res_host) (void)hipHostFree(h->res_host);
#ifdef GV_DIAG
  if (h->d_dbg) (void)hipFree(h->d_dbg);
  if (h->d_tl) (void)hipFree(h->d_tl);
  for (auto q : h->d_bin_dbg) if (q) (void)hipFree(q);
#endif
  for (int k = 0; k < gv_context::kSets; ++k) {
    void *xs[] = {h->x_ends[k], h->x_free[k], h->x_rects[k], h->x_stats[k]};
    for (void *p : xs)
      if (p) (void)hipFree(p);
  }
  for (int k = 0; k < gv_context::kStreams; ++k) {
    void *xs[] = {h->hits_s[k], h->cell_idx_s[k], h->bbox_id_s[k], h->d_vout_s[k], h->bin_keys[k], h->bin_tab[k],
                  h->bin_total[k][0], h->bin_total[k][1], h->bin_done[k], h->bin_scratch[k]};
    for (void *p : xs)
      if (p) (void)hipFree(p);
  }
  for (auto &c : h->cloud) {
    for (void *p : {(void *)c.base, (void *)c.raw})
      if (p) (void)hipFree(p);
    if (c.ready) (void)hipEventDestroy(c.ready);
  }
  for (auto &d : h->det) {
    for (void *p : {(void *)d.block, (void *)d.bbox_f, (void *)d.tile_mask})
      if (p) (void)hipFree(p);
    if (d.stage) (void)hipHostFree(d.stage);
    if (d.ready) (void)hipEventDestroy(d.ready);
  }
  for (auto &e : h->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto &pr : h->kt)
    for (auto &e : pr)
      if (e) (void)hipEventDestroy(e);
  for (auto &e : h->ev_fin)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : h->ev_sec)
    if (e) (void)hipEventDestroy(e);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  for (hipEvent_t e : {h->tick.done, h->tick.fork, h->tick.join})
    if (e) (void)hipEventDestroy(e);
  for (hipStream_t s : {h->stream4, h->stream3, h->stream2, h->stream_copy, h->stream})
    if (s) (void)hipStreamDestroy(s);
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

/* ------------------------------------------------------------------ ingest -- */
int gv_host_alloc(void **ptr, size_t bytes)
{
  if (!ptr || !bytes) return GV_ERR_BAD_ARG;
  *ptr = nullptr;
  return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? GV_OK : GV_ERR_HIP;
}

int gv_host_free(void *ptr)
{
  if (!ptr) return GV_OK;
  return hipHostFree(ptr) == hipSuccess ? GV_OK : GV_ERR_HIP;
}

}  // extern "C"

namespace {

// The next cloud set in rotation (read two uploads ago at the latest), grown to n points, with the copy
// stream ordered after the last frame that read it.
int begin_cloud_upload(gv_context *h, size_t n, int &target)
{
  int rc = set_device_only(h);
  if (rc) return rc;
  if ((rc = ensure_point_buffers(h, n))) return rc;
  target = (h->cloud_cur + 1) % 3;
  h->quiet_frames = 0;   // the upload stream is in use: the frames stay off it for a while
  CloudSet &c = h->cloud[target];
  if (n > c.cap) {
    if (c.release_slot >= 0) GV_HIP(hipEventSynchronize(h->ev_fin[c.release_slot]));
    GV_HIP(hipEventSynchronize(c.ready));
    if (c.base) GV_HIP(hipFree(c.base));
    c.base = nullptr;
    c.cap = 0;
    const size_t want = (n + n / 8 + 1024 + 3) & ~(size_t)3;   // the arrays sit at a stride of (n + 3) & ~3 floats
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&c.base), 3 * want * sizeof(float)));
    c.cap = want;
  }
  const size_t n4 = (n + 3) & ~(size_t)3;   // 16-byte aligned arrays
  c.x = c.base;
  c.y = c.base + n4;
  c.z = c.base + 2 * n4;
  // ordered after the last frame that read this set (if the ring slot has been re-recorded since, that is
  // a later frame: it only waits longer)
  // (asked first on the host: in a streaming run that frame finished long ago, and a wait that is already
  // satisfied would still put a barrier packet -- ~6 us of queue time -- in front of every copy)
  if (c.release_slot >= 0 && hipEventQuery(h->ev_fin[c.release_slot]) != hipSuccess)
    GV_HIP(hipStreamWaitEvent(h->stream_copy, h->ev_fin[c.release_slot], 0));
  c.release_slot = -1;
  return GV_OK;
}

int end_cloud_upload(gv_context *h, int target, size_t n)
{
  CloudSet &c = h->cloud[target];
  GV_HIP(hipEventRecord(c.ready, h->stream_copy));
  c.seen = 0;   // every stream that reads it waits for `ready` once
  h->cloud_cur = target;
  h->cx = c.x; h->cy = c.y; h->cz = c.z;
  h->n = n;
  h->cloud_wait = true;
  h->have_cell_idx = h->have_bbox_id = false;
  return GV_OK;
}

int upload_xyz(gv_context *h, const float *x, const float *y, const float *z, size_t n, bool wait)
{
  int target = 0;
  int rc = begin_cloud_upload(h, n, target);
  if (rc) return rc;
  CloudSet &c = h->cloud[target];
  if (n) {
    // the copy engine does ~54 GB/s inside a copy and leaves ~10 us between copies: x, y, z laid out back to
    // back in one (pinned) block go up in a single copy
    if (y == x + n && z == y + n && (n & 3) == 0) {
      GV_HIP(hipMemcpyAsync(c.x, x, 3 * n * sizeof(float), hipMemcpyHostToDevice, h->stream_copy));
    } else {
      GV_HIP(hipMemcpyAsync(c.x, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream_copy));
      GV_HIP(hipMemcpyAsync(c.y, y, n * sizeof(float), hipMemcpyHostToDevice, h->stream_copy));
      GV_HIP(hipMemcpyAsync(c.z, z, n * sizeof(float), hipMemcpyHostToDevice, h->stream_copy));
    }
  }
  if ((rc = end_cloud_upload(h, target, n))) return rc;
  if (wait) GV_HIP(hipEventSynchronize(c.ready));
  return GV_OK;
}

int upload_pc2(gv_context *h, const uint8_t *data, size_t n, uint32_t point_step, uint32_t off_x, uint32_t off_y,
               uint32_t off_z, bool wait)
{
  int target = 0;
  int rc = begin_cloud_upload(h, n, target);
  if (rc) return rc;
  CloudSet &c = h->cloud[target];
  const size_t bytes = n * (size_t)point_step;
  if (bytes + 16 > c.raw_cap) {
    GV_HIP(hipEventSynchronize(c.ready));   // the previous de-interleave out of this buffer is done
    if ((rc = grow(h, c.raw, c.raw_cap, bytes + bytes / 8 + 16))) return rc;
  }
  if (n) {
    GV_HIP(hipMemcpyAsync(c.raw, data, bytes, hipMemcpyHostToDevice, h->stream_copy));
    launch_deinterleave(c.raw, (uint32_t)n, point_step, off_x, off_y, off_z, c.x, c.y, c.z, h->stream_copy);
    GV_HIP(hipGetLastError());
  }
  if ((rc = end_cloud_upload(h, target, n))) return rc;
  if (wait) GV_HIP(hipEventSynchronize(c.ready));
  return GV_OK;
}

int set_detections(gv_context *h, const gv_frame_desc *d)
{
  if (!h || !d) return GV_ERR_BAD_ARG;
  if (d->n_bboxes < 0 || d->n_poses < 0) return GV_ERR_BAD_ARG;
  if (d->n_bboxes && !d->bboxes) return GV_ERR_BAD_ARG;
  const bool vision = d->flags & GV_FRAME_VISION_ORIENT;
  if (vision && d->n_bboxes && (!d->orient || !d->conf || !d->dims)) return GV_ERR_BAD_ARG;
  if (!vision && d->n_poses && !d->poses) return GV_ERR_BAD_ARG;
  int rc = set_device_only(h);
  if (rc) return rc;
  // The other detection set (frames already enqueued read the current one), uploaded on the stream of the
  // frame that will read it first: in order before that frame.  The frames that read the set's previous
  // contents: when all of them ran on that same lane (every frame brings new detections: two sets, two
  // lanes) the upload is already in order behind them; a set that was read on another stream as well (a
  // detection set kept for several frames is read on both lanes) waits for the grid pass of its last
  // reader, which completes after every earlier frame.
  const int target = h->det_cur ^ 1;
  DetSet &D = h->det[target];
  const int k = (sector_path(h) && !h->no_pipeline) ? 1 + (int)(h->lane_frames % (uint64_t)h->lanes_now()) : 0;
  hipStream_t s = h->streams[k];
  h->lane_clean[k] = false;   // the upload and the table kernels go on this stream, in front of the frame's partition pass
  if (D.release_slot >= 0 && D.readers != (1u << k)) GV_HIP(hipStreamWaitEvent(s, h->ev_fin[D.release_slot], 0));
  const bool net = vision && d->n_bboxes;
  if ((rc = upload_det(h, D, d->bboxes, d->n_bboxes, vision ? nullptr : d->poses, vision ? 0 : d->n_poses,
                       net ? d->orient : nullptr, net ? d->conf : nullptr, net ? d->dims : nullptr, s)))
    return rc;
  D.flags = d->flags;
  GV_HIP(hipEventRecord(D.ready, s));
  D.seen = 1u << k;
  D.release_slot = -1;
  D.readers = 0;
  h->det_cur = target;
  return GV_OK;
}

}  // namespace

extern "C" {

int gv_cloud_upload_xyz(gv_handle h, const float *x, const float *y, const float *z, size_t n)
{
  if (!h || (n && (!x || !y || !z)) || n > 0x7fffffffu) return GV_ERR_BAD_ARG;
  GV_TRY
  return upload_xyz(h, x, y, z, n, true);
  GV_CATCH
}

int gv_cloud_upload_xyz_async(gv_handle h, const float *x, const float *y, const float *z, size_t n)
{
  if (!h || (n && (!x || !y || !z)) || n > 0x7fffffffu) return GV_ERR_BAD_ARG;
  GV_TRY
  return upload_xyz(h, x, y, z, n, false);
  GV_CATCH
}

int gv_cloud_upload_pointcloud2(gv_handle h, const uint8_t *data, size_t n, uint32_t point_step, uint32_t off_x,
                                uint32_t off_y, uint32_t off_z)
{
  if (!h || (n && !data) || n > 0x7fffffffu) return GV_ERR_BAD_ARG;
  if (point_step < 4 || off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step)
    return GV_ERR_BAD_ARG;
  GV_TRY
  return upload_pc2(h, data, n, point_step, off_x, off_y, off_z, true);
  GV_CATCH
}

int gv_cloud_upload_pointcloud2_async(gv_handle h, const uint8_t *data, size_t n, uint32_t point_step, uint32_t off_x,
                                      uint32_t off_y, uint32_t off_z)
{
  if (!h || (n && !data) || n > 0x7fffffffu) return GV_ERR_BAD_ARG;
  if (point_step < 4 || off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step)
    return GV_ERR_BAD_ARG;
  GV_TRY
  return upload_pc2(h, data, n, point_step, off_x, off_y, off_z, false);
  GV_CATCH
}

int gv_cloud_upload_wait(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  int rc = set_device_only(h);
  if (rc) return rc;
  // the clouds' own `ready` events, not the upload stream: the third lane's frames run on that stream too and are
  // none of this call's business (round-3 advisor finding)
  for (auto &c : h->cloud) GV_HIP(hipEventSynchronize(c.ready));
  return GV_OK;
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

static void bbox_points_args(gv_context *h, PointsArgs &a)
{
  a.x = h->cx; a.y = h->cy; a.z = h->cz;
  a.n = (uint32_t)h->n;
  a.g = h->g;
  a.m_cam = h->m_cam;
  a.cam = h->camk;
  a.bt = bbox_test_of(h, h->det[2]);
  a.bbox_id = h->bbox_id;
  a.do_bbox = true;
}

// device int16 ids -> caller's int32 array
static int read_back_ids(gv_context *h, int32_t *out)
{
  if (!h->n) return GV_OK;
  int rc = ensure_scratch_i32(h, h->n);
  if (rc) return rc;
  launch_i16_to_i32(h->bbox_id, h->scratch_i32, h->n, h->stream);
  GV_HIP(hipGetLastError());
  GV_HIP(hipMemcpyAsync(out, h->scratch_i32, h->n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
}

int gv_extract_cloud_per_bbox(gv_handle h, const gv_bbox *bboxes, int32_t nb, int32_t *bbox_id, int32_t *counts)
{
  if (!h || nb < 0 || nb > 32767 || (nb && !bboxes) || !bbox_id) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if ((rc = upload_scratch_bboxes(h, bboxes, nb))) return rc;
  PointsArgs a{};
  bbox_points_args(h, a);
  launch_points(a, h->stream);
  GV_HIP(hipGetLastError());
  if ((rc = read_back_ids(h, bbox_id))) return rc;
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

// convertPixelsTo3D (grid_vision_node.cpp:309-335): B points, fp64, on the host
static void convert_pixels_host(const gv_context *h, const gv_bbox *bboxes, const float *depths, int32_t nb, double *base_points_xyz)
{
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
}

// camera-frame pose of one VisionOut (vision_orientation.cpp:432-444)
static gv_lshape_pose pose_of_vision_out(const VisionOut &vo)
{
  gv_lshape_pose p;
  p.px = vo.loc[0]; p.py = vo.loc[1]; p.pz = vo.loc[2];     // :434-436
  const host::Quat q = host::quat_from_rpy(0, -vo.orient, 0);   // :440
  p.qx = q.x; p.qy = q.y; p.qz = q.z; p.qw = q.w;
  p.length = vo.dims[0]; p.width = vo.dims[1]; p.height = vo.dims[2];
  return p;
}

int gv_convert_pixels_to_3d(gv_handle h, const gv_bbox *bboxes, const float *depths, int32_t nb,
                            double *base_points_xyz)
{
  if (!h || nb < 0 || (nb && (!bboxes || !depths || !base_points_xyz))) return GV_ERR_BAD_ARG;
  if (!h->has_bc) return GV_ERR_TF;
  GV_TRY
  convert_pixels_host(h, bboxes, depths, nb, base_points_xyz);
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
  DetSet &d = h->det[2];
  if ((rc = upload_det(h, d, bboxes, nb, nullptr, 0, orient, conf, dims, h->stream, false))) return rc;
  GV_HIP(hipEventRecord(d.ready, h->stream));
  launch_vision(d.orient, d.conf, d.dims, d.bboxes, nb, h->cam, h->d_vout, d.poses, h->stream);
  GV_HIP(hipGetLastError());
  std::vector<VisionOut> vo((size_t)nb);
  GV_HIP(hipMemcpyAsync(vo.data(), h->d_vout, (size_t)nb * sizeof(VisionOut), hipMemcpyDeviceToHost, h->stream));
  GV_HIP(hipStreamSynchronize(h->stream));
  int32_t m = 0;
  for (int32_t i = 0; i < nb; ++i) {
    if (!vo[i].valid) continue;   // vision_orientation.cpp:496-499
    poses_out[m++] = pose_of_vision_out(vo[i]);
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
  DetSet &d = h->det[2];
  if ((rc = upload_det(h, d, nullptr, 0, poses, n, nullptr, nullptr, nullptr, h->stream, false))) return rc;
  GV_HIP(hipEventRecord(d.ready, h->stream));
  if (n) launch_rects_from_poses(d.poses, n, h->g, false, h->x_bc, h->x_rects[0], h->stream);
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
  if ((rc = upload_scratch_bboxes(h, bboxes, n, false))) return rc;
  if (n) {
    GV_HIP(hipMemcpyAsync(h->d_pts, pts, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    launch_rects_from_points(h->d_pts, h->det[2].bboxes, n, h->g, h->x_rects[0], h->stream);
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

int gv_to_occupancy_grid_async(gv_handle h, int8_t *data)
{
  if (!h || !data) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = set_device_only(h);   // the public stream runs the grid passes: this copy sits between two of them
  if (rc) return rc;
  GV_HIP(hipMemcpyAsync(data, h->occ_i8, (size_t)h->g.G, hipMemcpyDeviceToHost, h->stream));
  return GV_OK;
  GV_CATCH
}

// The packed grid to PINNED host memory by a small kernel on the public stream, right behind the grid pass.  Measured
// with a cloud streaming in per frame (tools/stream_run.py, profiles/r04/publish_variants.txt): hipMemcpyAsync on the
// public stream 292-349 us per frame and erratic (the download's copy engine interleaves with the upload's: one
// download in six took 300 us instead of 85); the download ordered on the upload stream between two uploads 381 us
// (steady, but every copy command costs ~25 us of engine turn-around); this kernel 262 us, within 0.3 % frame after
// frame -- the copy engines stay with the uploads, PCIe carries both directions at once.
static int publish_by_kernel(gv_context *h, int8_t *data, hipStream_t s, bool *done)
{
  *done = false;
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, data) != hipSuccess || at.type != hipMemoryTypeHost || !at.devicePointer ||
      (reinterpret_cast<uintptr_t>(at.devicePointer) & 15u) != 0) {
    (void)hipGetLastError();   // pageable memory, or not 16-byte aligned (the kernel stores 16 bytes per lane): the copy command instead
    return GV_OK;
  }
  const size_t G = (size_t)h->g.G, body = G & ~(size_t)15;
  launch_publish_grid(h->occ_i8, static_cast<int8_t *>(at.devicePointer), body, 32, s);
  GV_HIP(hipGetLastError());
  if (G > body) GV_HIP(hipMemcpyAsync(data + body, h->occ_i8 + body, G - body, hipMemcpyDeviceToHost, s));
  *done = true;
  return GV_OK;
}

int gv_publish_grid_async(gv_handle h, int8_t *data)
{
  if (!h || !data) return GV_ERR_BAD_ARG;
  GV_TRY
  int rc = set_device_only(h);
  if (rc) return rc;
  bool done = false;
  if ((rc = publish_by_kernel(h, data, h->stream, &done))) return rc;
  if (!done) GV_HIP(hipMemcpyAsync(data, h->occ_i8, (size_t)h->g.G, hipMemcpyDeviceToHost, h->stream));
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
  GV_TRY
  return set_detections(h, d);
  GV_CATCH
}

int gv_frame_set_detections_async(gv_handle h, const gv_frame_desc *d)
{
  GV_TRY
  return set_detections(h, d);
  GV_CATCH
}

int gv_frame_enqueue(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  GV_TRY
  if (!h->det[h->det_cur].valid) return GV_ERR_STATE;   // no gv_frame_set_detections yet
  if (sector_path(h) && !h->no_pipeline) {
    int rc = set_device_only(h);
    if (rc) return rc;
    return enqueue_frame_tiles(h, true, false);
  }
  int rc = use_device(h);
  if (rc) return rc;
  return sector_path(h) ? enqueue_frame_tiles(h, false, false) : enqueue_frame_generic(h, false);
  GV_CATCH
}

int gv_frame_fence(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  int rc = set_device_only(h);
  if (rc) return rc;
  // Every frame ends with its grid pass on the public stream, behind an event that follows its other
  // kernels: frames are already in order there.  What is left to join is the copy stream.
  if (h->cloud_wait) {
    GV_HIP(hipEventRecord(h->ev_join, h->stream_copy));
    GV_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  }
  return GV_OK;
}

int gv_synchronize(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  int rc = set_device_only(h);
  if (rc) return rc;
  return drain(h);
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
  if (!h->have_hits) return GV_ERR_STATE;
  return copy_out(h, out, h->hits, (size_t)h->g.G * sizeof(int32_t));
}

int gv_get_miss(gv_handle h, int32_t *out)
{
  if (!h || !out) return GV_ERR_BAD_ARG;
  if (!h->have_miss) return GV_ERR_STATE;
  int rc = use_device(h);
  if (rc) return rc;
  if (sector_path(h))
    launch_miss_to_i32(h->x_freeN[h->last_set], h->x_freeT[h->last_set], h->g.nx, h->g.ny, h->nx_pad, h->ny_pad,
                       h->scratch_i32, h->stream);
  else
    launch_u8_to_i32(h->miss8, h->scratch_i32, (size_t)h->g.G, h->stream);
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
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  return read_back_ids(h, out);
  GV_CATCH
}

int gv_get_ray_stats(gv_handle h, uint64_t *n_rays, uint64_t *n_visits)
{
  if (!h) return GV_ERR_BAD_ARG;
  GV_TRY
  std::vector<unsigned long long> st(2 * h->stat_slots, 0ull);
  int rc = copy_out(h, st.data(), h->x_stats[h->last_set], st.size() * sizeof(unsigned long long));
  if (rc) return rc;
  unsigned long long rays = 0, visits = 0;
  for (size_t i = 0; i < h->stat_slots; ++i) { rays += st[2 * i]; visits += st[2 * i + 1]; }
  if (n_rays) *n_rays = rays;
  if (n_visits) *n_visits = visits;
  return GV_OK;
  GV_CATCH
}

#ifdef GV_DIAG
// diagnostic build only (tools/native_timeline.py, GV_TIMELINE=1): reset (out == nullptr) or copy out the
// {begin, end} clock pairs of the four kernels of the last `frames` <= 4096 frames, slot = frame number % 4096
int gv_debug_timeline(gv_handle h, unsigned long long *out, size_t frames)
{
  if (!h || !h->d_tl || frames > gv_context::kTlFrames) return GV_ERR_STATE;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  GV_HIP(hipDeviceSynchronize());
  const size_t n = gv_context::kTlFrames * 8;
  if (!out) {
    std::vector<unsigned long long> init(n);
    for (size_t i = 0; i < n; i += 2) { init[i] = ~0ull; init[i + 1] = 0ull; }
    GV_HIP(hipMemcpy(h->d_tl, init.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice));
    return GV_OK;
  }
  GV_HIP(hipMemcpy(out, h->d_tl, frames * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return GV_OK;
  GV_CATCH
}
uint64_t gv_debug_frame_no(gv_handle h) { return h ? h->frame_no : 0; }

// diagnostic build only (tools/sector_phases.py): copies the phase stamps of the last sector launch
int gv_debug_sector_stamps(gv_handle h, unsigned long long *out, size_t n_wg)
{
  if (!h || !out || !h->d_dbg) return GV_ERR_STATE;
  return copy_out(h, out, h->d_dbg, n_wg * 16 * sizeof(unsigned long long));
}

// diagnostic build only (tools/bin_phases.py): phase stamps of the last partition (which = 0) / tile (1) launch
int gv_debug_bin_stamps(gv_handle h, int which, unsigned long long *out, size_t n_wg)
{
  if (!h || !out || which < 0 || which > 1 || !h->d_bin_dbg[which] || n_wg > 8192) return GV_ERR_STATE;
  return copy_out(h, out, h->d_bin_dbg[which], n_wg * 16 * sizeof(unsigned long long));
}

// diagnostic build only: enqueue `frames` pipelined frames with timing events around every kernel;
// out[frame*10 + 2*k + {0,1}] = start/end in us of kernel k (rects, partition, tiles, sectors, grid pass)
int gv_debug_pipeline_trace(gv_handle h, int32_t frames, float *out)
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
#endif

void *gv_stream(gv_handle h) { return h ? (void *)h->stream : nullptr; }

int gv_device_layers(gv_handle h, int8_t **occ_i8, float **log_odds, float **occupancy)
{
  if (!h) return GV_ERR_BAD_ARG;
  if (occ_i8) *occ_i8 = h->occ_i8;
  if (log_odds) *log_odds = h->log_odds;
  if (occupancy) *occupancy = h->occupancy;
  return GV_OK;
}

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
  if ((rc = gv_frame_fence(h))) return rc;
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
  if (!h->det[h->det_cur].valid) return GV_ERR_STATE;
  int rc = use_device(h);
  if (rc) return rc;
  for (int s = 0; s < kNumStages; ++s) stage_ms[s] = 0.0f;
  for (int32_t i = 0; i < frames; ++i) {
    if ((rc = sector_path(h) ? enqueue_frame_tiles(h, false, true) : enqueue_frame_generic(h, true))) return rc;
    GV_HIP(hipEventSynchronize(h->ev[kNumStages]));
    for (int s = 0; s < kNumStages; ++s) {
      float ms = 0.0f;
      // tile path: the four kernels report their own start / end (dispatch-packet timestamps, the figure
      // rocprofv3 shows); everything else is the interval between two event records on the stream
      const int kq = s - kStagePoints;   // points, tile pass ("ray ends"), sectors, grid pass
      if (sector_path(h) && kq >= 0 && kq < 4) {
        if (h->kt_used[kq]) GV_HIP(hipEventElapsedTime(&ms, h->kt[kq][0], h->kt[kq][1]));
      } else {
        GV_HIP(hipEventElapsedTime(&ms, h->ev[s], h->ev[s + 1]));
      }
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

// ---- the result block (see gv_context::res_host) ----
constexpr size_t kResHeader = 64;

// a block with room for `bytes` of payload; the CallDone of the call about to be enqueued
static int begin_result(gv_context *h, size_t bytes, CallDone &done)
{
  // a tick between gv_tick_enqueue and gv_tick_wait owns the result block (and the standalone detection set): the
  // calls that would reuse them are refused until the tick has been waited for
  if (h->tick.pending) { h->err = "a tick is pending: call gv_tick_wait first"; return GV_ERR_STATE; }
  if (bytes + kResHeader > h->res_cap) {
    GV_HIP(hipStreamSynchronize(h->stream));   // nothing in flight writes the old block
    if (h->res_host) { GV_HIP(hipHostFree(h->res_host)); h->res_host = nullptr; }
    h->res_cap = 0;
    const size_t want = std::max<size_t>(2 * (bytes + kResHeader), 16384);
    // coherent (fine-grained) explicitly: the host must see the payload and the flag while the kernel that stores them
    // is still running, whatever HIP_HOST_COHERENT says
    GV_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->res_host), want, hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(h->res_host, 0, want);
    h->res_cap = want;
  }
  if (!h->d_res_ticket) {
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_res_ticket), 64));
    GV_HIP(hipMemsetAsync(h->d_res_ticket, 0, 64, h->stream));
  }
  if (++h->res_seq == 0u) h->res_seq = 1u;   // 0 = "nothing published yet"
  done.ticket = h->d_res_ticket;
  done.flag = reinterpret_cast<unsigned *>(h->res_host);
  done.seq = h->res_seq;
  return GV_OK;
}

static inline void cpu_relax()
{
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
  __builtin_ia32_pause();
#endif
}

// Host side of CallDone: spin on the block's first word.  The stream is looked at now and then so that a call
// whose kernels failed ends in an error instead of a hang.
static int wait_result(gv_context *h)
{
  volatile unsigned *flag = reinterpret_cast<volatile unsigned *>(h->res_host);
  const unsigned seq = h->res_seq;
  for (unsigned spins = 1;; ++spins) {
    if (*flag == seq) break;
    cpu_relax();   // the calls take 80-400 us: leave the core's other thread its issue slots
    if ((spins & 0xfffu) == 0u) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipErrorNotReady) continue;
      if (q == hipSuccess && *flag == seq) break;
      h->err = q == hipSuccess ? "result block never published" : hipGetErrorString(q);
      return GV_ERR_HIP;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
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
  if ((rc = upload_scratch_bboxes(h, bboxes, nb, false))) return rc;   // the kNN reads the boxes' centres only
  if ((rc = ensure_tbuf(h, std::max<size_t>(h->n, 1)))) return rc;
  if ((rc = grow(h, h->knn_partial, h->knn_partial_cap, knn_partial_entries(nb, k)))) return rc;
  // depths | sorted squared distances, stored by the merge kernel straight into the result block
  CallDone done;
  if ((rc = begin_result(h, (size_t)nb * (1 + (size_t)k) * sizeof(float), done))) return rc;
  float *r_depths = reinterpret_cast<float *>(h->res_host + kResHeader), *r_d2 = r_depths + nb;
  // buildKDTree projection (cloud_detections.cpp:8-33) then the exact k nearest (:43-87)
  launch_project_uvd(h->cx, h->cy, h->cz, (uint32_t)h->n, h->m_cam, h->camk, h->tx, h->ty, h->tz, h->stream);
  launch_knn(h->tx, h->ty, h->tz, (uint32_t)h->n, h->det[2].bboxes, nb, k, h->knn_partial, r_depths, knn_d2 ? r_d2 : nullptr, done,
             h->stream);
  GV_HIP(hipGetLastError());
  if ((rc = wait_result(h))) return rc;
  std::memcpy(depths, r_depths, (size_t)nb * sizeof(float));
  if (knn_d2) std::memcpy(knn_d2, r_d2, (size_t)nb * k * sizeof(float));
  return GV_OK;
  GV_CATCH
}

// smallest float >= the fp64 threshold: for a float f, f < thr_f <=> (double)f < thr
static float ceil_to_float(double v)
{
  float f = (float)v;
  if ((double)f < v) f = std::nextafterf(f, INFINITY);
  return f;
}

static int ensure_ransac_buffers(gv_context *h, size_t n, int32_t iterations)
{
  int rc;
  if ((size_t)iterations > h->planes_cap) {
    if (h->d_planes) { GV_HIP(hipFree(h->d_planes)); h->d_planes = nullptr; }
    if (h->d_plane_counts) { GV_HIP(hipFree(h->d_plane_counts)); h->d_plane_counts = nullptr; }
    h->planes_cap = 0;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_planes), (size_t)iterations * sizeof(float4)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_plane_counts), (size_t)iterations * kRansacCountSlices * sizeof(unsigned)));
    GV_HIP(hipMemsetAsync(h->d_plane_counts, 0, (size_t)iterations * kRansacCountSlices * sizeof(unsigned), h->stream));   // every pass leaves them zero
    h->planes_cap = (size_t)iterations;
  }
  if ((rc = grow(h, h->d_rscratch, h->rscratch_cap, ransac_scratch_doubles(n)))) return rc;
  if (!h->d_rstate) {
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_rstate), sizeof(RansacState)));
    GV_HIP(hipMemsetAsync(h->d_rstate, 0, sizeof(RansacState), h->stream));
  }
  return GV_OK;
}

static size_t pose_block_valid_off(int32_t nb) { return (size_t)nb * sizeof(gv_lshape_pose) + sizeof(RansacState); }
static size_t pose_block_bytes(int32_t nb) { return pose_block_valid_off(nb) + (size_t)nb; }

// extractCloudPerBBox -> RadiusOutlierRemoval -> centroid + PCA rectangle, all on the device and all enqueued
// without a host wait in between; only the nb poses come back.  with_ground: the points of the refined RANSAC
// plane in *d_rstate are dropped first (computeBBoxPose, cloud_detections.cpp:300-321), and the "empty segmented
// cloud" outcomes (:307-309) are decided on the device.
// poses_dev (optional): the camera-frame poses also go to device memory (length < 0 marks "no pose": what
// k_rects_from_poses skips), for a map update enqueued right behind this without a trip to the host.
static int enqueue_bbox_pose(gv_context *h, int32_t nb, bool with_ground, float thr_f, uint8_t *out, const CallDone &done,
                             gv_lshape_pose *poses_dev = nullptr)
{
  const size_t n = h->n;
  if (n > h->pc_cap) {
    if (h->d_nodes) { GV_HIP(hipFree(h->d_nodes)); h->d_nodes = nullptr; }
    if (h->d_keep) { GV_HIP(hipFree(h->d_keep)); h->d_keep = nullptr; }
    if (h->d_ticket_of) { GV_HIP(hipFree(h->d_ticket_of)); h->d_ticket_of = nullptr; }
    h->pc_cap = 0;
    const size_t want = n + n / 8 + 1024;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_nodes), want * sizeof(CellNode)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_keep), want));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_ticket_of), want * sizeof(uint32_t)));
    h->pc_cap = want;
  }
  if ((size_t)nb > h->pca_cap) {
    if (h->d_pca_acc) { GV_HIP(hipFree(h->d_pca_acc)); h->d_pca_acc = nullptr; }
    if (h->d_pca_ext) { GV_HIP(hipFree(h->d_pca_ext)); h->d_pca_ext = nullptr; }
    h->pca_cap = 0;
    const size_t want = (size_t)nb + (size_t)nb / 4 + 64;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_pca_acc), pca_acc_words((int)want) * sizeof(long long)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_pca_ext), pca_ext_words((int)want) * sizeof(unsigned)));
    GV_HIP(hipMemsetAsync(h->d_pca_acc, 0, pca_acc_words((int)want) * sizeof(long long), h->stream));   // every call leaves them zero
    GV_HIP(hipMemsetAsync(h->d_pca_ext, 0, pca_ext_words((int)want) * sizeof(unsigned), h->stream));
    h->pca_cap = want;
  }
  if (!h->d_pca_ticket) {
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_pca_ticket), 64));
    GV_HIP(hipMemsetAsync(h->d_pca_ticket, 0, 64, h->stream));
  }
  // cell buckets: a power of two, about one per two points (the three arrays stay L2 resident at config-3 size;
  // cells that share a bucket only add candidates that fail the id or distance test)
  size_t n_buckets = 4096;
  while (n_buckets < n / 2 && n_buckets < ((size_t)1 << 25)) n_buckets <<= 1;
  if (n_buckets > h->head_cap) {
    for (uint32_t **p : {&h->d_cellcnt, &h->d_cellpre, &h->d_celloff})
      if (*p) { GV_HIP(hipFree(*p)); *p = nullptr; }
    h->head_cap = 0;
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_cellcnt), n_buckets * sizeof(uint32_t)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_cellpre), (n_buckets + 4) * sizeof(uint32_t)));
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_celloff), (n_buckets / 4096 + 4) * sizeof(uint32_t)));
    GV_HIP(hipMemsetAsync(h->d_cellcnt, 0, n_buckets * sizeof(uint32_t), h->stream));   // every call counts them back to zero
    GV_HIP(hipMemsetAsync(h->d_celloff, 0, (n_buckets / 4096 + 4) * sizeof(uint32_t), h->stream));   // [n_buckets / 4096 + 2] = the scan's ticket
    h->head_cap = n_buckets;
  }
  n_buckets = h->head_cap;   // the table only grows
  if (!h->d_rstate) {
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_rstate), sizeof(RansacState)));
    GV_HIP(hipMemsetAsync(h->d_rstate, 0, sizeof(RansacState), h->stream));
  }
  hipStream_t s = h->stream;
  // extractCloudPerBBox + RadiusOutlierRemoval(0.4, 10)  (cloud_detections.cpp:250-298, 150-154)
  const double radius = 0.4;
  launch_radius_filter(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, h->camk, bbox_test_of(h, h->det[2]), nb, with_ground, thr_f,
                       h->d_rstate, h->bbox_id, h->d_cellcnt, h->d_cellpre, h->d_celloff,
                       h->d_celloff + n_buckets / 4096 + 2, h->d_nodes, h->d_keep, h->d_ticket_of, h->d_pca_acc, (uint32_t)n_buckets,
                       host::floor_to_float(radius * radius), 10, s);
  h->have_bbox_id = true;
  // centroid + PCA rectangle per bbox from order-independent integer sums over the kept points (:156-247)
  launch_pca_rect(h->d_nodes, h->d_celloff + n_buckets / 4096, (uint32_t)n, h->d_keep, h->d_pca_acc, h->d_pca_ext, h->d_pca_ticket, nb,
                  h->d_rstate, with_ground, reinterpret_cast<gv_lshape_pose *>(out), out + pose_block_valid_off(nb),
                  reinterpret_cast<RansacState *>(out + (size_t)nb * sizeof(gv_lshape_pose)), done, s, poses_dev);
  GV_HIP(hipGetLastError());
  return GV_OK;
}

static int compute_bbox_pose_impl(gv_handle h, const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out,
                                  uint8_t *valid, bool with_ground, RansacState *st_out)
{
  if (!h || nb < 0 || nb > 32767 || (nb && (!bboxes || !poses_out || !valid))) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  const size_t n = h->n;
  for (int32_t b = 0; b < nb; ++b) { valid[b] = 0; poses_out[b] = gv_lshape_pose{}; }
  if (st_out) *st_out = RansacState{};
  if (n == 0 || (with_ground && n < 3)) return GV_OK;
  if (nb && (rc = upload_scratch_bboxes(h, bboxes, nb))) return rc;
  const float thr_f = ceil_to_float(0.04);
  if (with_ground) {   // segmentGroundPlane(0.04, 50 hypotheses) on the camera-frame cloud (grid_vision_node.cpp:215-216)
    if ((rc = ensure_ransac_buffers(h, n, 50))) return rc;
    launch_ransac_plane(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, thr_f, 50, 12345ull, h->d_planes, h->d_plane_counts,
                        h->d_rscratch, h->d_rstate, h->stream);
    GV_HIP(hipGetLastError());
    h->ground_n = 0;   // the mask itself is not materialised on this path
  }
  if (nb) {
    // poses | state | flags: stored by the PCA kernel straight into the result block, no copy, no runtime wait
    CallDone done;
    if ((rc = begin_result(h, pose_block_bytes(nb), done))) return rc;
    const uint8_t *blk = h->res_host + kResHeader;
    if ((rc = enqueue_bbox_pose(h, nb, with_ground, thr_f, h->res_host + kResHeader, done))) return rc;
    if ((rc = wait_result(h))) return rc;
    std::memcpy(poses_out, blk, (size_t)nb * sizeof(gv_lshape_pose));
    std::memcpy(valid, blk + pose_block_valid_off(nb), (size_t)nb);
    if (st_out) std::memcpy(st_out, blk + (size_t)nb * sizeof(gv_lshape_pose), sizeof(RansacState));
    return GV_OK;
  }
  if (with_ground) {   // no boxes: the ground count still decides the return value
    if ((rc = grow(h, h->d_ground, h->ground_cap, n))) return rc;
    CallDone done;
    if ((rc = begin_result(h, sizeof(RansacState), done))) return rc;
    launch_ransac_mask(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, thr_f, h->d_rstate, h->d_ground,
                       reinterpret_cast<RansacState *>(h->res_host + kResHeader), done, h->stream);
    GV_HIP(hipGetLastError());
    if ((rc = wait_result(h))) return rc;
    if (st_out) std::memcpy(st_out, h->res_host + kResHeader, sizeof(RansacState));
    return GV_OK;
  }
  GV_HIP(hipStreamSynchronize(h->stream));
  return GV_OK;
  GV_CATCH
}

int gv_compute_bbox_pose(gv_handle h, const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out, uint8_t *valid)
{
  return compute_bbox_pose_impl(h, bboxes, nb, poses_out, valid, false, nullptr);
}

// segmentGroundPlane on the device; state (plane, inlier count) comes back, the mask stays resident
static int segment_ground_device(gv_context *h, double threshold, int32_t iterations, uint64_t seed, RansacState &st)
{
  const size_t n = h->n;
  st = RansacState{};
  h->ground_n = 0;
  if (n < 3) return GV_OK;
  int rc;
  if ((rc = ensure_ransac_buffers(h, n, iterations))) return rc;
  if ((rc = grow(h, h->d_ground, h->ground_cap, n))) return rc;
  const float thr_f = ceil_to_float(threshold);
  // camera-frame cloud (the reference segments transformed_cloud, grid_vision_node.cpp:215-216): transformed on the fly
  launch_ransac_plane(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, thr_f, iterations, seed, h->d_planes, h->d_plane_counts,
                      h->d_rscratch, h->d_rstate, h->stream);
  CallDone done;
  if ((rc = begin_result(h, sizeof(RansacState), done))) return rc;
  launch_ransac_mask(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, thr_f, h->d_rstate, h->d_ground,
                     reinterpret_cast<RansacState *>(h->res_host + kResHeader), done, h->stream);
  GV_HIP(hipGetLastError());
  if ((rc = wait_result(h))) return rc;
  std::memcpy(&st, h->res_host + kResHeader, sizeof(RansacState));
  h->ground_n = n;
  return GV_OK;
}

int gv_segment_ground_plane(gv_handle h, double threshold, int32_t iterations, uint64_t seed, uint8_t *is_ground,
                            float coeff[4], int64_t *n_inliers)
{
  if (!h || !(threshold > 0.0) || iterations < 1 || iterations > 4096) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  GV_TRY
  int rc = use_device(h);
  if (rc) return rc;
  if (coeff) coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.0f;
  if (n_inliers) *n_inliers = 0;
  if (is_ground && h->n) std::memset(is_ground, 0, h->n);
  RansacState st;
  if ((rc = segment_ground_device(h, threshold, iterations, seed, st))) return rc;
  if (!st.best_count) return GV_OK;   // "Could not estimate a planar model" (:122-126)
  if (is_ground) {   // the caller asked for the per-point mask: the only O(N) transfer of this call
    GV_HIP(hipMemcpyAsync(is_ground, h->d_ground, h->n, hipMemcpyDeviceToHost, h->stream));
    GV_HIP(hipStreamSynchronize(h->stream));
  }
  if (coeff) { coeff[0] = st.refined.x; coeff[1] = st.refined.y; coeff[2] = st.refined.z; coeff[3] = st.refined.w; }
  if (n_inliers) *n_inliers = (int64_t)st.n_inliers;
  return GV_OK;
  GV_CATCH
}

int gv_compute_bbox_pose_ground_removed(gv_handle h, const gv_bbox *bboxes, int32_t nb, gv_lshape_pose *poses_out,
                                        uint8_t *valid, int32_t *n_poses_or_fail)
{
  if (!h || nb < 0 || (nb && (!bboxes || !poses_out || !valid))) return GV_ERR_BAD_ARG;
  if (!h->has_cl) return GV_ERR_TF;
  // computeBBoxPose (cloud_detections.cpp:300-321): segmentGroundPlane -> extractCloudPerBBox -> PCA, enqueued as
  // one batch: the device decides the "empty segmented cloud" cases, the host reads 56 bytes of state + the poses
  if (n_poses_or_fail) *n_poses_or_fail = 0;
  RansacState st;
  int rc = compute_bbox_pose_impl(h, bboxes, nb, poses_out, valid, true, &st);
  if (rc) return rc;
  const uint64_t m = st.best_count ? st.n_inliers : 0;
  if (m == 0 || (size_t)m == h->n) {   // empty segmented cloud -> the reference returns {} (:307-309)
    for (int32_t b = 0; b < nb; ++b) valid[b] = 0;
    if (n_poses_or_fail) *n_poses_or_fail = -1;
    return GV_OK;
  }
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
  // the exchange stream of the sharded frame and the events that chain its steps (ordering only)
  if (!h->stream_x) GV_HIP(hipStreamCreateWithFlags(&h->stream_x, hipStreamNonBlocking));
  for (auto &row : h->ev_sh)
    for (auto &e : row)
      if (!e) GV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : h->sh_t)
    if (!e) GV_HIP(hipEventCreate(&e));
  return GV_OK;
  GV_CATCH
}

int gv_comm_info(gv_handle h, int32_t *n_ranks, int32_t *rank, int32_t *device)
{
  if (!h) return GV_ERR_BAD_ARG;
  if (!h->comm) return GV_ERR_STATE;
  int nr = 0, rk = 0, dev = 0;
  GV_NCCL(ncclCommCount(h->comm, &nr));
  GV_NCCL(ncclCommUserRank(h->comm, &rk));
  GV_NCCL(ncclCommCuDevice(h->comm, &dev));
  if (n_ranks) *n_ranks = nr;
  if (rank) *rank = rk;
  if (device) *device = dev;
  return GV_OK;
}

int gv_comm_destroy(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  if (!h->comm) return GV_OK;
  (void)hipSetDevice(h->device);
  (void)drain(h);
  ncclCommDestroy(h->comm);
  h->comm = nullptr;
  h->rank = 0;
  h->world = 1;
  return GV_OK;
}

int gv_frame_enqueue_sharded(gv_handle h)
{
  if (!h) return GV_ERR_BAD_ARG;
  if (!h->comm || !sector_path(h)) return GV_ERR_STATE;
  GV_TRY
  if (!h->det[h->det_cur].valid) return GV_ERR_STATE;   // no gv_frame_set_detections yet
  int rc = set_device_only(h);
  if (rc) return rc;
  return enqueue_frame_sharded(h, nullptr);
  GV_CATCH
}

int gv_process_frame_sharded(gv_handle h, const gv_frame_desc *desc)
{
  if (!h || !desc) return GV_ERR_BAD_ARG;
  if (!h->comm) return GV_ERR_STATE;
  if (!sector_path(h)) return GV_ERR_STATE;
  int rc = gv_frame_set_detections(h, desc);
  if (rc) return rc;
  if ((rc = gv_frame_enqueue_sharded(h))) return rc;
  return gv_synchronize(h);
}

int gv_time_frame_sharded_stages(gv_handle h, int32_t frames, float stage_ms[6])
{
  if (!h || frames <= 0 || !stage_ms) return GV_ERR_BAD_ARG;
  if (!h->comm || !sector_path(h)) return GV_ERR_STATE;
  GV_TRY
  if (!h->det[h->det_cur].valid) return GV_ERR_STATE;
  int rc = use_device(h);
  if (rc) return rc;
  for (int s = 0; s < 6; ++s) stage_ms[s] = 0.0f;
  for (int32_t i = 0; i < frames; ++i) {   // one frame at a time: every step alone on the device
    if ((rc = enqueue_frame_sharded(h, h->sh_t))) return rc;
    if ((rc = drain(h))) return rc;
    for (int s = 0; s < 6; ++s) {
      float ms = 0.0f;
      GV_HIP(hipEventElapsedTime(&ms, h->sh_t[s], h->sh_t[s + 1]));
      stage_ms[s] += ms;
    }
  }
  for (int s = 0; s < 6; ++s) stage_ms[s] /= (float)frames;
  return GV_OK;
  GV_CATCH
}

int gv_shard_band_rows(int32_t rank, int32_t world, int32_t ny, int32_t *y0, int32_t *y1)
{
  if (world < 1 || rank < 0 || rank >= world || ny < 1 || !y0 || !y1) return GV_ERR_BAD_ARG;
  const int ny_pad = kBinTile * ((ny + kBinTile - 1) / kBinTile);
  shard_band_rows(rank, world, ny, ny_pad, *y0, *y1);
  return GV_OK;
}

int64_t gv_shard_slice_words(int64_t words, int32_t world)
{
  if (words < 0 || world < 1) return -1;
  return (int64_t)(((((size_t)words + (size_t)world - 1) / (size_t)world) + 3) & ~(size_t)3);
}

// Test hook: the sharded frame for every rank of a `world`-GPU job, run on THIS device with the RCCL
// exchanges replaced by device copies (ShardLink emulation).  The resident cloud is the whole cloud;
// rank r takes points [n*r/world, n*(r+1)/world).  Every piece the ranks would run -- binning of a
// slice, OR of the end-bitmap slices, every world-th sector workgroup, band packing, band OR, band grid
// pass -- runs with its real (rank, world); the bands land in the one resident grid.
int gv_test_frame_sharded_emulated(gv_handle h, const gv_frame_desc *desc, int32_t world)
{
  if (!h || !desc || world < 1 || world > 16) return GV_ERR_BAD_ARG;
  if (!sector_path(h)) return GV_ERR_STATE;
  int rc = gv_frame_set_detections(h, desc);
  if (rc) return rc;
  GV_TRY
  if ((rc = use_device(h))) return rc;
  DetSet &D = h->det[h->det_cur];
  const uint32_t fl = D.flags;
  const bool do_bin = fl & GV_FRAME_BIN, do_ray = fl & GV_FRAME_RAYMARCH, do_bbox = fl & GV_FRAME_BBOX_TEST;
  const bool keep_cell = fl & GV_FRAME_KEEP_CELL_IDX;
  if ((rc = check_frame_flags(h, fl))) return rc;
  if (!do_bin) return GV_ERR_STATE;
  if ((rc = ensure_shard_scratch(h, world))) return rc;
  if ((rc = ensure_point_buffers(h, h->n, (h->n + (size_t)world - 1) / (size_t)world))) return rc;
  hipStream_t s = h->stream;
  const size_t slice = shard_ends_slice(h, world), Ep = slice * (size_t)world;
  const size_t chunk = free_band_chunk_words(h->nxw, h->nx_pad, h->ny_pad, world);
  std::vector<uint32_t *> ends((size_t)world, nullptr), packs((size_t)world, nullptr);
  uint32_t *comb = nullptr;
  auto cleanup = [&]() {
    for (uint32_t *q : ends) if (q) (void)hipFree(q);
    for (uint32_t *q : packs) if (q) (void)hipFree(q);
    if (comb) (void)hipFree(comb);
  };
  auto body = [&]() -> int {
    GV_HIP(hipMalloc(reinterpret_cast<void **>(&comb), Ep * sizeof(uint32_t)));
    for (int r = 0; r < world; ++r) {
      GV_HIP(hipMalloc(reinterpret_cast<void **>(&ends[r]), Ep * sizeof(uint32_t)));
      GV_HIP(hipMalloc(reinterpret_cast<void **>(&packs[r]), chunk * (size_t)world * sizeof(uint32_t)));
    }
    Rect *rects = h->x_rects[0];
    int rc2;
    if ((rc2 = wait_inputs(h, h->cloud[h->cloud_cur], D, 0))) return rc2;
    const int32_t n_rects = enqueue_rects(h, D, rects, h->d_vout_s[0], s);
    for (int r = 0; r < world; ++r) {   // every rank bins its slice
      const size_t lo = h->n * (size_t)r / (size_t)world, hi = h->n * (size_t)(r + 1) / (size_t)world;
      if ((rc2 = enqueue_binning(h, D, 0, 0, lo, hi - lo, keep_cell, do_ray, do_bbox, false, nullptr))) return rc2;
      GV_HIP(hipMemcpyAsync(ends[r], h->x_ends[0], Ep * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    for (int q = 0; q < world; ++q) {   // exchange 1: rank q ORs slice q; the all-gather is the union of the slices
      ShardLink L{h, q, world, ends.data()};
      if ((rc2 = shard_or_ends_slice(L, comb, s))) return rc2;
    }
    GV_HIP(hipMemcpyAsync(h->x_ends[0], comb, Ep * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    for (int r = 0; r < world; ++r) {   // every rank's share of the ray stage, packed by band
      GV_HIP(hipMemsetAsync(h->x_free[0], 0, (h->bmN_words + h->bmT_words) * sizeof(uint32_t), s));
      if (do_ray && (rc2 = enqueue_sectors(h, 0, r, world, s))) return rc2;
      launch_pack_free_bands(h->x_freeN[0], h->x_freeT[0], h->nxw, h->nx_pad, h->ny_pad, world, chunk, packs[r], s);
      GV_HIP(hipGetLastError());
    }
    for (int q = 0; q < world; ++q) {   // exchange 2 + grid pass of band q
      ShardLink L{h, q, world, packs.data()};
      if ((rc2 = shard_or_free_band(L, 0, nullptr, s))) return rc2;
      int32_t y0, y1;
      shard_band_rows(q, world, h->g.ny, h->ny_pad, y0, y1);
      if ((rc2 = enqueue_grid_pass(h, 0, rects, n_rects, true, y0, y1, s))) return rc2;
    }
    GV_HIP(hipStreamSynchronize(s));
    return GV_OK;
  };
  rc = body();
  (void)hipStreamSynchronize(s);
  cleanup();
  h->last_set = 0;
  h->hits = h->hits_s[0];
  h->bbox_id = h->bbox_id_s[0];
  h->cell_idx = h->cell_idx_s[0];
  h->have_hits = false;
  h->have_miss = false;
  h->have_cell_idx = do_bin && keep_cell;
  h->have_bbox_id = do_bbox;
  return rc;
  GV_CATCH
}

/* ------------------------------------------------------------ the node's tick -- */
// GridVision::timerCallback from filterBBoxes on (grid_vision_node.cpp:153-244) as ONE batch of device work: the
// static boxes' kNN depth (:168-184), the dynamic boxes' poses -- orientation-network geometry (:190-209) or ground
// removal + per-box clouds + radius filter + PCA rectangle (:210-231) --, their rectangles in the base frame, the
// map update with the int8 pack, and the packed grid's way home.  The poses never leave the device on their way
// into the grid (k_pca_bbox / k_vision -> k_rects_from_poses(from_cam) -> grid pass); what the markers need comes
// back through the pinned result block.  gv_tick_wait is the tick's only host wait.
int gv_tick_enqueue(gv_handle h, const gv_tick_desc *d)
{
  if (!h || !d || d->n_bboxes < 0 || d->n_bboxes > 16383 || (d->n_bboxes && !d->bboxes)) return GV_ERR_BAD_ARG;
  if (d->n_net < 0 || (d->n_net && (!d->orient || !d->conf || !d->dims))) return GV_ERR_BAD_ARG;
  const bool vision = d->flags & GV_TICK_VISION_ORIENT;
  const bool lidar = d->flags & GV_TICK_LIDAR_BIN, lidar_ray = d->flags & GV_TICK_LIDAR_RAYMARCH;
  if (lidar_ray && !lidar) return GV_ERR_BAD_ARG;
  GV_TRY
  gv_context::Tick &T = h->tick;
  if (T.pending) return GV_ERR_STATE;   // one tick at a time (the node's timer is single threaded, grid_vision_node.cpp:49-50)
  const int32_t n_all = d->n_bboxes;
  // filterBBoxes (:384-403), order preserving
  std::vector<gv_bbox> cat((size_t)2 * n_all + 1);
  int32_t ns = 0, nd = 0;
  if (n_all) {
    std::memcpy(cat.data(), d->bboxes, (size_t)n_all * sizeof(gv_bbox));
    std::vector<gv_bbox> dy((size_t)n_all);
    int rcf = gv_filter_bboxes(d->bboxes, n_all, cat.data() + n_all, &ns, dy.data(), &nd);
    if (rcf) return rcf;
    std::memcpy(cat.data() + n_all + ns, dy.data(), (size_t)nd * sizeof(gv_bbox));
  }
  const int32_t k = d->k_near;
  if (ns && (k < 1 || k > 32)) return GV_ERR_BAD_ARG;
  if (vision && d->n_net && d->n_net != nd) return GV_ERR_BAD_ARG;
  if (n_all && (!h->has_cl || !h->has_bc)) return GV_ERR_TF;   // transformLidarToCamera / transformPoseToBaseFrame
  if (lidar && !h->has_bl) return GV_ERR_TF;
  if (lidar && !sector_path(h)) { h->err = "the lidar extension inside the tick needs the tile path (nx % 4 == 0)"; return GV_ERR_STATE; }
  int rc = use_device(h);   // frames in flight finish first: the tick's work is one sequence on the public stream
  if (rc) return rc;
  hipStream_t s = h->stream;
  const size_t n = h->n;
  const bool pca = !vision && nd > 0 && n >= 3;              // computeBBoxPose on ALL boxes (:215-216)
  const bool net = vision && nd > 0 && d->n_net == nd;       // poses only when the network ran for every dynamic box
  DetSet &D = h->det[2];
  if (n_all) {
    if ((rc = upload_det(h, D, cat.data(), 2 * n_all, nullptr, 0, net ? d->orient : nullptr, net ? d->conf : nullptr,
                         net ? d->dims : nullptr, s, pca, net ? nd : 0, n_all, true)))
      return rc;
    GV_HIP(hipEventRecord(D.ready, s));
  }
  // result block: depths | poses, state, valid (the PCA call's layout) | VisionOut
  T.off_depth = 0;
  T.off_pose = ((size_t)ns * sizeof(float) + 15) & ~(size_t)15;
  T.off_vout = (T.off_pose + pose_block_bytes(n_all) + 15) & ~(size_t)15;
  CallDone none;   // nothing published: the tick ends with an event on the public stream
  if ((rc = begin_result(h, T.off_vout + (size_t)nd * sizeof(VisionOut) + 16, none))) return rc;
  none = CallDone{};
  uint8_t *blk = h->res_host + kResHeader;
  // --- static boxes: buildKDTree + computeDepthForBoundingBoxes (:168-184).  Independent of the pose branch: it
  // runs on a lane beside it and joins the public stream before the tick's last event.
  T.knn_ran = ns > 0;
  bool knn_forked = false;
  if (ns > 0) {
    if ((rc = ensure_tbuf(h, std::max<size_t>(n, 1)))) return rc;
    if ((rc = grow(h, h->knn_partial, h->knn_partial_cap, knn_partial_entries(ns, k)))) return rc;
    hipStream_t sk = s;
    if (h->env_tick_knn_lane && nd > 0) {
      sk = h->streams[1];
      GV_HIP(hipEventRecord(T.fork, s));
      GV_HIP(hipStreamWaitEvent(sk, T.fork, 0));
      h->lane_clean[1] = false;
      knn_forked = true;
    }
    launch_project_uvd(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, h->camk, h->tx, h->ty, h->tz, sk);
    launch_knn(h->tx, h->ty, h->tz, (uint32_t)n, D.bboxes + n_all, ns, k, h->knn_partial,
               reinterpret_cast<float *>(blk + T.off_depth), nullptr, none, sk);
    GV_HIP(hipGetLastError());
    if (knn_forked) GV_HIP(hipEventRecord(T.join, sk));
  }
  // --- dynamic boxes -> camera-frame poses on the device -> rectangles
  int32_t n_rects = 0;
  Rect *rects = h->x_rects[0];
  T.pca_ran = T.vision_ran = false;
  if (net) {   // VisionOrientation::postProcessOutputs (:190-209)
    launch_vision(D.orient, D.conf, D.dims, D.bboxes + n_all + ns, nd, h->cam, reinterpret_cast<VisionOut *>(blk + T.off_vout),
                  D.poses, s);
    launch_rects_from_poses(D.poses, nd, h->g, true, h->x_bc, rects, s);
    n_rects = nd;
    T.vision_ran = true;
  } else if (pca) {   // cloud_detections::computeBBoxPose (:210-231)
    const float thr_f = ceil_to_float(0.04);
    if ((rc = ensure_ransac_buffers(h, n, 50))) return rc;
    launch_ransac_plane(h->cx, h->cy, h->cz, (uint32_t)n, h->m_cam, thr_f, 50, 12345ull, h->d_planes, h->d_plane_counts,
                        h->d_rscratch, h->d_rstate, s);
    h->ground_n = 0;
    if ((rc = enqueue_bbox_pose(h, n_all, true, thr_f, blk + T.off_pose, none, D.poses))) return rc;
    launch_rects_from_poses(D.poses, n_all, h->g, true, h->x_bc, rects, s);
    n_rects = n_all;
    T.pca_ran = true;
  }
  GV_HIP(hipGetLastError());
  // --- map update (:145, :206, :230, :235) + int8 pack (:265-278)
  if (lidar && n > 0) {   // [EXTENSION] the fused frame's kernels, serial on the public stream
    if ((rc = ensure_point_buffers(h, n))) return rc;
    if ((rc = enqueue_binning(h, D, 0, 0, 0, n, false, lidar_ray, false, true, nullptr))) return rc;
    if (lidar_ray && (rc = enqueue_sectors(h, 0, 0, 1, s))) return rc;
    if ((rc = enqueue_grid_pass(h, 0, rects, n_rects, true, 0, h->g.ny, s))) return rc;
    h->last_set = 0;
    h->hits = h->hits_s[0];
    h->have_hits = true;
    h->have_miss = true;
  } else if ((rc = enqueue_plain_update(h, n_rects)))
    return rc;
  // a copy command, not gv_publish_grid_async's kernel: no upload competes for the copy engines inside a tick, and the
  // kernel measured no faster here (0.326 vs 0.321 ms PCA tick, 0.167 vs 0.161 ms vision tick)
  if (d->grid_out) GV_HIP(hipMemcpyAsync(d->grid_out, h->occ_i8, (size_t)h->g.G, hipMemcpyDeviceToHost, s));
  if (knn_forked) GV_HIP(hipStreamWaitEvent(s, T.join, 0));
  GV_HIP(hipEventRecord(T.done, s));
  T.flags = d->flags;
  T.n_all = n_all; T.n_static = ns; T.n_dynamic = nd;
  T.st_boxes.assign(cat.begin() + n_all, cat.begin() + n_all + ns);
  T.pending = true;
  return GV_OK;
  GV_CATCH
}

int gv_tick_wait(gv_handle h, gv_tick_result *r)
{
  if (!h || !r) return GV_ERR_BAD_ARG;
  GV_TRY
  gv_context::Tick &T = h->tick;
  if (!T.pending) return GV_ERR_STATE;
  int rc = set_device_only(h);
  if (rc) return rc;
  GV_HIP(hipEventSynchronize(T.done));   // the tick's one host wait
  T.pending = false;
  const uint8_t *blk = h->res_host + kResHeader;
  r->n_static = T.n_static;
  r->n_dynamic = T.n_dynamic;
  r->n_poses = 0;
  r->pca_empty = 0;
  if (T.n_static) {
    const float *dep = reinterpret_cast<const float *>(blk + T.off_depth);
    if (r->static_bboxes) std::memcpy(r->static_bboxes, T.st_boxes.data(), (size_t)T.n_static * sizeof(gv_bbox));
    if (r->depths) std::memcpy(r->depths, dep, (size_t)T.n_static * sizeof(float));
    if (r->base_points_xyz) convert_pixels_host(h, T.st_boxes.data(), dep, T.n_static, r->base_points_xyz);   // :180
  }
  if (T.vision_ran) {
    const VisionOut *vo = reinterpret_cast<const VisionOut *>(blk + T.off_vout);
    for (int32_t i = 0; i < T.n_dynamic; ++i) {
      if (!vo[i].valid) continue;   // vision_orientation.cpp:496-499
      gv_lshape_pose p = pose_of_vision_out(vo[i]);
      host::transform_pose(h->tf_bc, p);   // transformLShapeObjects (:204)
      if (r->poses) r->poses[r->n_poses] = p;
      r->n_poses++;
    }
  } else if (T.pca_ran) {
    const gv_lshape_pose *ps = reinterpret_cast<const gv_lshape_pose *>(blk + T.off_pose);
    RansacState st;
    std::memcpy(&st, blk + T.off_pose + (size_t)T.n_all * sizeof(gv_lshape_pose), sizeof(st));
    const uint8_t *valid = blk + T.off_pose + pose_block_valid_off(T.n_all);
    const uint64_t m = st.best_count ? st.n_inliers : 0;
    if (m == 0 || (size_t)m == h->n) r->pca_empty = 1;   // empty segmented cloud: computeBBoxPose returns {} (:307-309)
    else
      for (int32_t b = 0; b < T.n_all; ++b) {
        if (!valid[b]) continue;   // :174-175
        gv_lshape_pose p = ps[b];
        host::transform_pose(h->tf_bc, p);   // transformLShapeObjects (:227)
        if (r->poses) r->poses[r->n_poses] = p;
        r->n_poses++;
      }
  } else if (!(T.flags & GV_TICK_VISION_ORIENT) && T.n_dynamic > 0)
    r->pca_empty = 1;   // fewer than three points: no plane, no poses
  return GV_OK;
  GV_CATCH
}

int gv_tick(gv_handle h, const gv_tick_desc *d, gv_tick_result *r)
{
  int rc = gv_tick_enqueue(h, d);
  if (rc) return rc;
  return gv_tick_wait(h, r);
}

int gv_comm_band(gv_handle h, int64_t *begin, int64_t *end)
{
  if (!h) return GV_ERR_BAD_ARG;
  int32_t y0, y1;
  shard_band_rows(h->rank, h->world, h->g.ny, h->ny_pad, y0, y1);
  if (begin) *begin = (int64_t)y0 * h->g.nx;
  if (end) *end = (int64_t)y1 * h->g.nx;
  return GV_OK;
}

}  // extern "C"
