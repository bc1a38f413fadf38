// gv_binning.hip -- [EXTENSION] X1 per-point binning with int32 hit counts for the tile path of
// the production frame, plus the X2 ray ends and the A5 bbox test that share the pass over the
// cloud.  gfx950, wave64, built with -ffp-contract=off (gv_device.hpp).
//
// Two kernels, no global atomics on the count grid, no cleared buffers:
//
//   k_bin_partition  one workgroup per chunk of the SoA cloud.  Per point: base<-lidar
//       transform, getIndex -> (128x128-cell tile, cell inside the tile); out-of-map points go
//       through the fp64 slab clip on dense lanes afterwards and become "clipped end" keys; the
//       camera transform + projection + first-match bbox id ride along.  The chunk's 16-bit keys
//       are counting-sorted by tile in LDS and written out as ONE contiguous run per workgroup
//       (fully coalesced whatever the cloud looks like) with a row of per-tile start offsets.
//   k_bin_tiles      one workgroup per tile: gathers the tile's key segments from every chunk,
//       histograms them in LDS (int32, 64 KB), writes the tile of hits[] with plain coalesced
//       stores and emits the four end bitmaps (hit/clip, both orientations) the sector ray stage
//       reads -- every cell and every bitmap word is written every frame, so nothing is cleared.
//       A tile that holds more than kSplitKeys keys (the cells next to the sensor of a real
//       lidar) is shared by up to kSplitMax workgroups, each histogramming every k-th chunk; the
//       partial tiles meet in a scratch slab and the workgroup whose ticket comes last sums them
//       (integer sums: the result does not depend on the arrival order).
//
// Keys: bits 0..13 = cell inside the tile (ly << 7 | lx), bit 14 = clipped ray end (X2: end of
// an out-of-map point's ray on the map border, counts as traversed), bit 15 unused.
#include "gv_kernels.hpp"

#include <cstdlib>

#include <hip/hip_ext.h>
#include "gv_device.hpp"

#include <algorithm>

namespace gv {

constexpr unsigned kKeyClip = 1u << 14;
constexpr unsigned kStagedNone = 0xFFFFFFFFu;      // dropped point (non-finite, or outside with no ray)
constexpr unsigned kStagedOutside = 0xFFFFFFFEu;   // out-of-map point waiting for the clip
#ifndef GV_PART_BATCH
#define GV_PART_BATCH 4                       // points per lane and step of the partition pass
#endif
#ifndef GV_PART_WPE
#define GV_PART_WPE 7                         // wavefronts per SIMD the partition pass is compiled for
#endif
constexpr int kPartThreads = 512;             // 8 wavefronts per chunk (fits beside a sector workgroup)
constexpr int kTileThreads = 1024;
constexpr uint32_t kInLaneKeys = 128;         // tile pass: a segment up to this long is histogrammed by its own lane

// Diagnostic build only (-DGV_DIAG): thread 0 of every workgroup stamps the shader clock at phase
// boundaries into a buffer nothing else reads (tools/bin_phases.py); absent from the shipped kernels.
#ifdef GV_DIAG
#define GV_STAMP(dbg, k)                                                                                  \
  do {                                                                                                    \
    if ((dbg) && threadIdx.x == 0) (dbg)[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();   \
  } while (0)
#else
#define GV_STAMP(dbg, k) do { } while (0)
#endif

// inclusive add-scan over the 64 lanes as DPP modifiers of six dependent VALU adds (row_shr 1, 2, 4, 8, then
// row_bcast:15 / :31 into the upper rows) instead of six ds_bpermute round trips
__device__ __forceinline__ unsigned wave_incl_scan_add(unsigned v)
{
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}

// ------------------------------------------------------------- partition -----
#define GV_KARG(field) load_karg<decltype(BinArgs::field)>(offsetof(BinArgs, field))
template <bool RAY, bool BBOX, bool KEEPCELL>
__global__ void __launch_bounds__(kPartThreads, GV_PART_WPE) k_bin_partition(BinArgs a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  if (blockIdx.x >= a.n_wg) {
    // one extra workgroup: the frame's base-frame poses -> index rectangles for the grid pass (A8/A9,
    // src/occupancy_grid.cpp:79-90,147-172).  Rides this launch instead of costing one of its own.
    for (int q = threadIdx.x; q < a.n_rect_poses; q += kPartThreads) a.rects_out[q] = rect_from_pose(a.g, a.rect_poses[q]);
    return;
  }
  const int T = a.n_tiles;
  unsigned *staged = reinterpret_cast<unsigned *>(smem);                      // [chunk] tile << 16 | key
  unsigned *hist = staged + a.chunk;                                          // [T] counts, then cursors
  unsigned short *sorted = reinterpret_cast<unsigned short *>(hist + T);      // [chunk] keys grouped by tile
  unsigned short *outl = sorted;                                              // outside-point list (dead before `sorted` is written)
  __shared__ unsigned s_wsum[kPartThreads / 64], s_nout;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t w = blockIdx.x;
  const uint32_t base = w * a.chunk;
  const uint32_t npts = min(a.chunk, a.n - base);
  GV_STAMP(a.dbg, 0);
  GV_TL_BEGIN(a.tl);
  for (int t = tid; t < T; t += kPartThreads) hist[t] = 0;
  if (tid == 0) s_nout = 0;
  // bbox test: thresholds and tile candidate masks staged in LDS -- per point they are a chain of dependent
  // look-ups (mask word -> one threshold quadruple per candidate), an LDS latency each instead of a global one
  BBoxTest lbt = a.bt;
  if (BBOX) {
    const size_t off = ((size_t)a.chunk * 4 + (size_t)T * 4 + (size_t)a.chunk * 2 + 15) & ~(size_t)15;
    float4 *l_bbox = reinterpret_cast<float4 *>(smem + off);
    unsigned long long *l_mask = reinterpret_cast<unsigned long long *>(smem + off + (size_t)a.nb_pad * 16);
    const int nmask = a.bt.tiles_x * a.bt.tiles_y * a.bt.mask_words;
    for (int q = tid; q < a.nb; q += kPartThreads) l_bbox[q] = a.bt.bbox_f[q];
    for (int q = tid; q < nmask; q += kPartThreads) l_mask[q] = a.bt.tile_mask[q];
    lbt.bbox_f = l_bbox;
    lbt.tile_mask = l_mask;
  }
  __syncthreads();

  // PP points per lane and step, handled phase by phase over all PP of them in straight-line code: the loads,
  // then the base transform and the cell of every point, then (bbox test) the camera transform and the
  // projection of every point, then one candidate loop over all of them.  Nothing per point sits behind a branch
  // except the side effects (LDS count, stores), so the PP dependency chains interleave, and the constants of a
  // phase are fetched once per step instead of once per point (chunk % (PP * kPartThreads) == 0).  The rare exact
  // divisions are taken by the whole wavefront when any of its PP * 64 quotients needs one.
  constexpr int PP = GV_PART_BATCH;
  for (uint32_t k0 = tid; k0 < a.chunk; k0 += PP * kPartThreads) {
    float px[PP], py[PP], pz[PP];
    bool live[PP];
#pragma unroll
    for (int u = 0; u < PP; ++u) {
      const uint32_t k = k0 + (uint32_t)u * kPartThreads;
      live[u] = k < npts;
      const uint32_t i = base + (live[u] ? k : 0u);   // base < n: a valid address for the idle lanes of the last chunk
      px[u] = a.x[i]; py[u] = a.y[i]; pz[u] = a.z[i];
    }
    // ---- base frame: cell, tile, key
    {
      const Mat34f mb = GV_KARG(m_base);   // constants are fetched where they are used (gv_device.hpp, load_karg)
      const GridParams g = GV_KARG(g);
      double dx[PP], dy[PP], qx[PP], qy[PP];
      bool fin[PP], in0[PP], rx[PP], ry[PP];
      bool any_risky = false;
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        float bx, by, bz;
        xform34(mb, px[u], py[u], pz[u], bx, by, bz);
        fin[u] = live[u] & isfinite(bx) & isfinite(by) & isfinite(bz);   // (bitwise: no short-circuit branches)
        // get_index_fast (gv_device.hpp), flattened
        const double x = (double)bx, y = (double)by;
        const double tx = -((x - g.pos_x) - g.off_x);
        const double ty = -((y - g.pos_y) - g.off_y);
        in0[u] = fin[u] & (tx >= 0.0) & (ty >= 0.0) & (tx < g.len_x) & (ty < g.len_y);
        dx[u] = (x - g.off_x) - g.pos_x;
        dy[u] = (y - g.off_y) - g.pos_y;
        qx[u] = -dx[u] * g.inv_res;
        qy[u] = -dy[u] * g.inv_res;
        rx[u] = in0[u] & (fabs(qx[u] - rint(qx[u])) < 1e-6);
        ry[u] = in0[u] & (fabs(qy[u] - rint(qy[u])) < 1e-6);
        any_risky = any_risky | rx[u] | ry[u];
      }
      if (__ballot(any_risky) != 0ull) {
#pragma unroll
        for (int u = 0; u < PP; ++u) {
          if (rx[u]) qx[u] = -(dx[u] / g.res);
          if (ry[u]) qy[u] = -(dy[u] / g.res);
        }
      }
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        const uint32_t k = k0 + (uint32_t)u * kPartThreads;
        const int jx = (int)(in0[u] ? qx[u] : 0.0);
        const int jy = (int)(in0[u] ? qy[u] : 0.0);
        const bool inside = in0[u] & (jx >= 0) & (jy >= 0) & (jx < g.nx) & (jy < g.ny);
        const unsigned tile = (unsigned)((jy >> kBinTileLog) * a.tiles_x + (jx >> kBinTileLog));
        const unsigned key = (tile << 16) | (unsigned)(((jy & (kBinTile - 1)) << kBinTileLog) | (jx & (kBinTile - 1)));
        const unsigned st = inside ? key : ((RAY & fin[u] & (a.org.valid != 0)) ? kStagedOutside : kStagedNone);
        if (inside) atomicAdd(&hist[tile], 1u);
        staged[k] = st;
        if (RAY) {
          // out-of-map points (a minority) need the fp64 slab clip, ~10x the work of an in-map point: their
          // positions in the chunk are collected here so that the clip below runs on dense lanes
          const bool o = st == kStagedOutside;
          const unsigned long long bm = __ballot(o);
          if (bm) {
            unsigned wbase = 0;
            if (lane == 0) wbase = atomicAdd(&s_nout, (unsigned)__popcll(bm));
            wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)wbase);
            if (o) outl[wbase + (unsigned)__popcll(bm & ((1ull << lane) - 1ull))] = (unsigned short)k;
          }
        }
        if (KEEPCELL && live[u]) a.cell_idx[base + k] = inside ? jy * g.nx + jx : -1;
      }
    }
    // ---- camera frame: index of the first bbox containing the projection (first_bbox, gv_device.hpp, flattened)
    if (BBOX) {
      const Mat34f mc = GV_KARG(m_cam);
      const CamK ck = GV_KARG(cam);
      float uu[PP], vv[PP];
      double n0[PP], n1[PP], zz[PP];
      bool ok[PP], k0r[PP], k1r[PP];
      bool any_risky = false;
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        float cx, cy, cz;
        xform34(mc, px[u], py[u], pz[u], cx, cy, cz);
        ok[u] = live[u] & isfinite(cx) & isfinite(cy) & isfinite(cz) & !(cz <= 0.001f);   // :264
        const double X = (double)cx, Y = (double)cy, Z = (double)cz;
        zz[u] = ok[u] ? Z : 1.0;
        const double riz = rcp_newton(zz[u]);
        n0[u] = ck.k[0] * X + ck.k[2] * Z;   // K's zero and unit entries dropped: see first_bbox
        n1[u] = ck.k[4] * Y + ck.k[5] * Z;
        const double q0 = n0[u] * riz, q1 = n1[u] * riz;
        k0r[u] = ok[u] & div_risky(q0);
        k1r[u] = ok[u] & div_risky(q1);
        any_risky = any_risky | k0r[u] | k1r[u];
        uu[u] = (float)q0;
        vv[u] = (float)q1;
      }
      if (__ballot(any_risky) != 0ull) {
#pragma unroll
        for (int u = 0; u < PP; ++u) {
          if (k0r[u]) uu[u] = (float)(n0[u] / zz[u]);
          if (k1r[u]) vv[u] = (float)(n1[u] / zz[u]);
        }
      }
      int id[PP];
      unsigned moff[PP];
      bool img[PP];
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        img[u] = ok[u] & !((uu[u] < 0) | (uu[u] >= (float)ck.W) | (vv[u] < 0) | (vv[u] >= (float)ck.H));   // :276
        const int tx = (int)(img[u] ? uu[u] : 0.0f) >> 4, ty = (int)(img[u] ? vv[u] : 0.0f) >> 4;
        moff[u] = (unsigned)((ty * lbt.tiles_x + tx) * lbt.mask_words);
        id[u] = -1;
      }
      for (int wd = 0; wd < lbt.mask_words; ++wd) {   // :280-288 first match wins; one word = 64 boxes
        unsigned long long m[PP];
        bool more = false;
#pragma unroll
        for (int u = 0; u < PP; ++u) {
          const unsigned long long mw = lbt.tile_mask[moff[u] + (unsigned)wd];
          m[u] = (img[u] & (id[u] < 0)) ? mw : 0ull;
          more = more | (m[u] != 0ull);
        }
        while (__ballot(more) != 0ull) {
          more = false;
#pragma unroll
          for (int u = 0; u < PP; ++u) {
            const bool have = m[u] != 0ull;
            const int b = have ? wd * 64 + (__ffsll((long long)m[u]) - 1) : 0;
            const float4 f = lbt.bbox_f[b];
            const bool hit = have & (uu[u] >= f.x) & (uu[u] <= f.z) & (vv[u] >= f.y) & (vv[u] <= f.w);
            id[u] = hit ? b : id[u];
            m[u] = hit ? 0ull : (m[u] & (m[u] - 1ull));
            more = more | (m[u] != 0ull);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        const uint32_t k = k0 + (uint32_t)u * kPartThreads;
        if (live[u]) __builtin_nontemporal_store((int16_t)id[u], &a.bbox_id[base + k]);   // host-read output
      }
    }
  }
  __syncthreads();
  GV_STAMP(a.dbg, 1);   // points done

  if (RAY) {
    const unsigned nout = s_nout;
    for (unsigned j = tid; j < nout; j += kPartThreads) {
      const unsigned k = outl[j];
      const uint32_t i = base + k;
      float bx, by, bz;
      xform34(a.m_base, a.x[i], a.y[i], a.z[i], bx, by, bz);
      int ex, ey;
      clip_ray_end(a.g, a.org, (double)bx, (double)by, ex, ey);
      const unsigned tile = (unsigned)((ey >> kBinTileLog) * a.tiles_x + (ex >> kBinTileLog));
      staged[k] = (tile << 16) | kKeyClip | (unsigned)(((ey & (kBinTile - 1)) << kBinTileLog) | (ex & (kBinTile - 1)));
      atomicAdd(&hist[tile], 1u);
    }
    __syncthreads();
  }

  GV_STAMP(a.dbg, 2);   // clipped ends done
  // exclusive prefix over the tile counts: thread owns the tiles [tid*per, tid*per + per)
  const int per = (T + kPartThreads - 1) / kPartThreads;
  const int t0 = tid * per, t1 = min(T, t0 + per);
  unsigned mine = 0;
  for (int t = t0; t < t1; ++t) mine += hist[t];
  const unsigned incl = wave_incl_scan_add(mine);
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  unsigned run = incl - mine, total = 0;
#pragma unroll
  for (int wv = 0; wv < kPartThreads / 64; ++wv) {
    const unsigned s = s_wsum[wv];
    if (wv < wave) run += s;
    total += s;
  }
  unsigned short *row = a.tab + (size_t)w * (size_t)(T + 1);
  for (int t = t0; t < t1; ++t) {
    const unsigned c = hist[t];
    row[t] = (unsigned short)run;
    hist[t] = run;   // placement cursor
    if (c) atomicAdd(&a.tile_total[t], c);   // no-return; 256 B contiguous per wavefront when per == 1
    run += c;
  }
  if (tid == 0) row[T] = (unsigned short)total;
  __syncthreads();
  GV_STAMP(a.dbg, 3);   // scan + table row

  for (uint32_t k = tid; k < a.chunk; k += kPartThreads) {
    const unsigned st = staged[k];
    if (st < kStagedOutside) sorted[atomicAdd(&hist[st >> 16], 1u)] = (unsigned short)(st & 0xFFFFu);
  }
  __syncthreads();
  GV_STAMP(a.dbg, 4);   // sorted in LDS
  // one contiguous run per workgroup (base of the chunk's key region is 4-byte aligned: chunk is even)
  const unsigned *src = reinterpret_cast<const unsigned *>(sorted);
  unsigned *dst = reinterpret_cast<unsigned *>(a.keys + (size_t)w * a.chunk);
  for (unsigned j = tid; j < (total + 1) / 2; j += kPartThreads) dst[j] = src[j];
  GV_STAMP(a.dbg, 5);
  GV_TL_END(a.tl);
}

// ------------------------------------------------------------------ tiles -----
__device__ __forceinline__ unsigned bin_splits(unsigned n, unsigned split_keys)
{
  const unsigned k = (n + split_keys - 1) / split_keys;
  return k < 1u ? 1u : (k > (unsigned)kBinSplitMax ? (unsigned)kBinSplitMax : k);
}

template <bool WRITE_HITS>
__global__ void __launch_bounds__(kTileThreads) k_bin_tiles(BinTileArgs a)
{
  __shared__ unsigned hist[kBinTileCells];           // 64 KB
  __shared__ unsigned bits[2][kBinTile][4];          // [0] hit, [1] clipped end: bit x of row y
  __shared__ unsigned s_scanh[kTileThreads / 64], s_scane[kTileThreads / 64];
  __shared__ unsigned s_lbase[kTileThreads], s_llen[kTileThreads], s_nlong;   // long segments of one round
  __shared__ int s_t, s_sp, s_slot;
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.n_tiles;
  GV_STAMP(a.dbg, 0);
  GV_TL_BEGIN(a.tl);
  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx b and b + 8 share one, each with its own L2).
  // Consecutive tiles read neighbouring bytes of every chunk (one 128-byte line of a chunk's offset row covers
  // 64 tiles, one line of its sorted keys ~9 tiles), so each XCD takes a contiguous run of tiles: its L2 then
  // fetches those lines once instead of every XCD fetching all of them.
  const int per_xcd = (T + 7) >> 3;
  const int n_primary = per_xcd << 3;
  const bool primary = (int)blockIdx.x < n_primary;
  int t = primary ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : -1, sp = 0, slot = 0;
  if (primary && t >= T) return;   // T is not a multiple of 8: a few idle workgroups
  unsigned k = 1;
  if (primary) {
    k = bin_splits(a.tile_total[t], a.split_keys);
    if (tid == 0) a.tile_total_next[t] = 0;   // the next frame's partition adds into it
  }
  if (!primary || k > 1) {
    // Every workgroup of a shared tile derives the same (slot, share) from the tile totals: slot =
    // number of shared tiles before t, helper j serves the tile whose run of extra shares holds j.
    if (tid == 0) { s_t = -1; s_sp = 0; s_slot = 0; }
    const int per = (T + kTileThreads - 1) / kTileThreads;
    const int q0 = tid * per, q1 = min(T, q0 + per);
    unsigned hv = 0, ex = 0;
    for (int q = q0; q < q1; ++q) {
      const unsigned kq = bin_splits(a.tile_total[q], a.split_keys);
      hv += (kq > 1u);
      ex += kq - 1u;
    }
    const unsigned ih = wave_incl_scan_add(hv), ie = wave_incl_scan_add(ex);
    if (lane == 63) { s_scanh[wave] = ih; s_scane[wave] = ie; }
    __syncthreads();
    unsigned bh = ih - hv, be = ie - ex;
    for (int wv = 0; wv < wave; ++wv) { bh += s_scanh[wv]; be += s_scane[wv]; }
    const unsigned j = primary ? 0u : (unsigned)((int)blockIdx.x - n_primary);
    for (int q = q0; q < q1; ++q) {
      const unsigned kq = bin_splits(a.tile_total[q], a.split_keys);
      if (primary) {
        if (q == t) s_slot = (int)bh;
      } else if (kq > 1u && j >= be && j < be + (kq - 1u)) {
        s_t = q;
        s_sp = 1 + (int)(j - be);
        s_slot = (int)bh;
      }
      bh += (kq > 1u);
      be += kq - 1u;
    }
    __syncthreads();
    slot = s_slot;
    if (!primary) {
      t = s_t;
      sp = s_sp;
      if (t < 0) return;   // more helpers than extra shares this frame
      k = bin_splits(a.tile_total[t], a.split_keys);
    }
    if (slot >= (int)a.max_slots) {   // cannot happen while max_slots > n / split_keys; keeps the scratch in bounds
      if (!primary) return;
      k = 1;
    }
  }

  GV_STAMP(a.dbg, 1);   // role known
  // ---- gather: this share takes the chunks sp, sp + k, sp + 2k, ...  One lane per chunk segment: it fetches
  // the segment's (start, end) pair while the LDS is being zeroed, then the two aligned 16-byte windows (8 keys
  // each) that hold the segment's first keys -- at ~7 keys per segment that is the whole segment -- so the tile
  // costs two dependent global latencies.  What a crowded tile's segments hold beyond that goes to a list the
  // wavefronts then stream through with 16-byte loads, 512 keys per wave instruction.
  const uint32_t nshare = (a.n_wg > (uint32_t)sp) ? (a.n_wg - (uint32_t)sp + k - 1) / k : 0u;
  const size_t rowlen = (size_t)T + 1;
  auto fetch_desc = [&](uint32_t q, uint32_t &kb) -> unsigned {   // segment q of this share: start | end << 16
    kb = 0;
    if (q >= nshare) return 0u;
    const uint32_t w = (uint32_t)sp + q * k;
    kb = w * a.chunk;   // first key of the chunk (n_wg * chunk < 2^32)
    const unsigned short *row = a.tab + (size_t)w * rowlen;
    return (unsigned)row[t] | ((unsigned)row[t + 1] << 16);
  };
  uint32_t kb0 = 0;
  unsigned desc = fetch_desc((uint32_t)tid, kb0);
  for (int c = tid; c < kBinTileCells / 4; c += kTileThreads) reinterpret_cast<uint4 *>(hist)[c] = make_uint4(0, 0, 0, 0);
  if (tid < 2 * kBinTile * 4) (&bits[0][0][0])[tid] = 0;
  if (tid == 0) s_nlong = 0;
  __syncthreads();
  GV_STAMP(a.dbg, 2);   // LDS zeroed, first descriptors on their way
  auto add_key = [&](unsigned kk) {
    const unsigned local = kk & (kBinTileCells - 1);
    if (kk & kKeyClip) {
      // clipped ray ends pile up on the map border (every out-of-map point ends on one of ~8000 cells): thousands of
      // keys for the same few words.  Same-address LDS atomics serialise, plain reads broadcast: look first.
      unsigned *wp = &bits[1][local >> kBinTileLog][(local & (kBinTile - 1)) >> 5];
      const unsigned bit = 1u << (local & 31u);
      if (!(*reinterpret_cast<volatile unsigned *>(wp) & bit)) atomicOr(wp, bit);
    } else {
      atomicAdd(&hist[local], 1u);
    }
  };
  // keys [lo, hi) of the aligned window at key index wb (8 keys = one uint4)
  auto add_window = [&](const uint4 &v, uint32_t wb, uint32_t lo, uint32_t hi) {
    const unsigned wv[4] = {v.x, v.y, v.z, v.w};
    if (wb >= lo && wb + 8u <= hi && !((v.x | v.y | v.z | v.w) & (kKeyClip | (kKeyClip << 16)))) {
      // a whole window inside the segment and no clipped end in it (every window of a long segment but its first
      // and last): eight plain increments
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        atomicAdd(&hist[wv[u] & (kBinTileCells - 1)], 1u);
        atomicAdd(&hist[(wv[u] >> 16) & (kBinTileCells - 1)], 1u);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t pos = wb + (uint32_t)u;
      if (pos >= lo && pos < hi) add_key((wv[u >> 1] >> (16 * (u & 1))) & 0xFFFFu);
    }
  };
  const uint4 *keys4 = reinterpret_cast<const uint4 *>(a.keys);
  const uint32_t nrounds = (nshare + kTileThreads - 1) / kTileThreads;
  for (uint32_t round = 0; round < nrounds; ++round) {
    uint32_t kbn = 0;
    const unsigned next = (round + 1 < nrounds) ? fetch_desc((round + 1) * kTileThreads + (uint32_t)tid, kbn) : 0u;
    if (round) {
      __syncthreads();   // the long-segment list of the previous round is consumed
      if (tid == 0) s_nlong = 0;
      __syncthreads();
    }
    {
      const uint32_t lo = kb0 + (desc & 0xFFFFu), hi = kb0 + (desc >> 16);   // absolute key range of the segment
      if (hi > lo) {
        // up to four aligned windows (>= 25 keys) per lane, all loads issued before the first key is used:
        // covers every segment of an evenly filled tile and of the map-border tiles (clipped ray ends double
        // their segments); only a crowded tile's segments leave a remainder for the list
        const uint32_t wb = lo & ~7u;
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = make_uint4(0, 0, 0, 0);
          if (hi > wb + 8u * (uint32_t)q) v[q] = keys4[(wb >> 3) + (uint32_t)q];   // (the buffer has slack past the last chunk)
        }
        // what lies beyond: a long remainder (a crowded tile: hundreds of keys per segment) goes to the list and
        // gets a wavefront; a medium one (the ring of tiles around the sensor: 40..128 keys per segment) stays in
        // this lane, four windows at a time -- a wavefront per 5-window remainder would run 59 idle lanes through
        // a full load latency, 30 times in a row (measured: 80 k cycles for such a tile, the slowest of the launch)
        const bool listed = hi > wb + 32u && hi - wb > kInLaneKeys;
        if (listed) {
          const unsigned slot = atomicAdd(&s_nlong, 1u);
          s_lbase[slot] = wb + 32u;
          s_llen[slot] = hi - (wb + 32u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (hi > wb + 8u * (uint32_t)q) add_window(v[q], wb + 8u * (uint32_t)q, lo, hi);
        if (!listed) {
          for (uint32_t w2 = wb + 32u; w2 < hi; w2 += 32u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[q] = make_uint4(0, 0, 0, 0);
              if (hi > w2 + 8u * (uint32_t)q) v[q] = keys4[(w2 >> 3) + (uint32_t)q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (hi > w2 + 8u * (uint32_t)q) add_window(v[q], w2 + 8u * (uint32_t)q, lo, hi);
          }
        }
      }
    }
    desc = next;
    kb0 = kbn;
    __syncthreads();
    const unsigned nl = s_nlong;   // <= 1024 per round
    GV_STAMP(a.dbg, 6 + 2 * (round > 0));   // diagnostic: in-lane part of the round done
    // long remainders: wavefront per list entry, a lane per aligned 8-key window, four windows in flight per lane
    for (unsigned e = (unsigned)wave; e < nl; e += kTileThreads / 64) {
      const uint32_t lo = s_lbase[e], hi = lo + s_llen[e];   // lo is window aligned
      const uint32_t nwin = (hi - lo + 7u) >> 3;
      for (uint32_t w0 = (uint32_t)lane; w0 < nwin; w0 += 256u) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = make_uint4(0, 0, 0, 0);
          if (w0 + 64u * (uint32_t)q < nwin) v[q] = keys4[(lo >> 3) + w0 + 64u * (uint32_t)q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (w0 + 64u * (uint32_t)q < nwin) add_window(v[q], lo + 8u * (w0 + 64u * (uint32_t)q), lo, hi);
      }
    }
    GV_STAMP(a.dbg, 7 + 2 * (round > 0));   // diagnostic: wavefront 0 done with its list entries
  }
  __syncthreads();
  GV_STAMP(a.dbg, 3);   // keys histogrammed

  if (k > 1) {
    // partial tile -> scratch slab; the last of the k shares to arrive sums them
    const size_t stride = (size_t)kBinTileCells + 512;
    unsigned *slab = a.scratch + (size_t)slot * kBinSplitMax * stride;
    unsigned *dst = slab + (size_t)sp * stride;
    for (int c = tid; c < kBinTileCells / 4; c += kTileThreads)
      reinterpret_cast<uint4 *>(dst)[c] = reinterpret_cast<const uint4 *>(hist)[c];
    if (tid < 512) dst[kBinTileCells + tid] = (&bits[1][0][0])[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned ticket = __hip_atomic_fetch_add(&a.done[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == k - 1u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&a.done[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next frame
      }
      s_ticket = ticket;
    }
    __syncthreads();
    if (s_ticket != k - 1u) return;
    for (int c = tid; c < kBinTileCells / 4; c += kTileThreads) {
      uint4 p[kBinSplitMax];   // every share's vector requested before the first is added (k <= kBinSplitMax)
#pragma unroll
      for (unsigned q = 0; q < (unsigned)kBinSplitMax; ++q) {
        p[q] = make_uint4(0, 0, 0, 0);
        if (q < k) p[q] = reinterpret_cast<const uint4 *>(slab + (size_t)q * stride)[c];
      }
      uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (unsigned q = 0; q < (unsigned)kBinSplitMax; ++q) { v.x += p[q].x; v.y += p[q].y; v.z += p[q].z; v.w += p[q].w; }
      reinterpret_cast<uint4 *>(hist)[c] = v;
    }
    if (tid < 512) {
      unsigned v = 0;
      for (unsigned q = 0; q < k; ++q) v |= slab[(size_t)q * stride + kBinTileCells + tid];
      (&bits[1][0][0])[tid] = v;
    }
    __syncthreads();
  }

  // ---- write-out: the tile of hits[] (plain coalesced 16-byte stores) and the hit bits of every row.
  // A lane takes 4 consecutive cells; 8 lanes make one 32-bit word of the row's hit bits: the lane's
  // nibble is shifted into place and OR-reduced over the 8 lanes with three DPP row shifts.
  const int x0 = (t % a.tiles_x) << kBinTileLog, y0 = (t / a.tiles_x) << kBinTileLog;
#pragma unroll
  for (int it = 0; it < kBinTileCells / 4 / kTileThreads; ++it) {
    const int c4 = it * kTileThreads + tid;          // 4-cell group: row c4 >> 5, cells 4 * (c4 & 31) ..
    const int ly = c4 >> 5, lx = (c4 & 31) << 2;
    const uint4 v = reinterpret_cast<const uint4 *>(hist)[c4];
    if (WRITE_HITS) {
      const int x = x0 + lx, y = y0 + ly;
      if (x < a.nx && y < a.ny) {   // nx % 4 == 0 on the tile path
        // written once, never read again on the device: streaming stores keep 16 MB per frame out of the L2
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u nv; nv.x = v.x; nv.y = v.y; nv.z = v.z; nv.w = v.w;
        __builtin_nontemporal_store(nv, reinterpret_cast<v4u *>(a.hits + (size_t)y * a.nx + x));
      }
    }
    unsigned nib = (v.x ? 1u : 0u) | (v.y ? 2u : 0u) | (v.z ? 4u : 0u) | (v.w ? 8u : 0u);
    nib <<= 4 * (tid & 7);
    // OR over the 8 lanes of a word into its first lane: lane i takes lane i + 1, + 2, + 4 of its row (DPP row_shl,
    // zero past the row end -- the 8-lane groups are row aligned)
    nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x101, 0xF, 0xF, false);
    nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x102, 0xF, 0xF, false);
    nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x104, 0xF, 0xF, false);
    if ((tid & 7) == 0) bits[0][ly][lx >> 5] = nib;
  }
  __syncthreads();
  GV_STAMP(a.dbg, 4);   // hits[] written
  // end bitmaps, 32-bit words stored transposed (gv_raysector.hip):
  //   N: bits run along x, word(x>>5, y) at (x>>5)*ny_pad + y;  T: bits run along y, word(y>>5, x) at (y>>5)*nx_pad + x
  {
    const int which = tid >> 9, wq = (tid >> 7) & 3, l = tid & (kBinTile - 1);
    {
      const int gw = (x0 >> 5) + wq, gy = y0 + l;
      unsigned *dN = which ? a.clipN : a.hitN;
      if (gw < a.nxw && gy < a.ny_pad) {
        dN[(size_t)gw * a.ny_pad + gy] = bits[which][l][wq];
        if (which == 0 && a.freeN) a.freeN[(size_t)gw * a.ny_pad + gy] = 0u;   // this set's ray stage starts from no free cells
      }
    }
    {
      unsigned wv = 0;
#pragma unroll
      for (int r = 0; r < 32; ++r) wv |= ((bits[which][32 * wq + r][l >> 5] >> (l & 31)) & 1u) << r;
      const int gw = (y0 >> 5) + wq, gx = x0 + l;
      unsigned *dT = which ? a.clipT : a.hitT;
      if (gw < a.nyw && gx < a.nx_pad) {
        dT[(size_t)gw * a.nx_pad + gx] = wv;
        if (which == 0 && a.freeT) a.freeT[(size_t)gw * a.nx_pad + gx] = 0u;
      }
    }
  }
  GV_STAMP(a.dbg, 5);
  GV_TL_END(a.tl);
}

// ---------------------------------------------------------------- launch -----
uint32_t bin_chunk_for(size_t n)
{
  // ~500..1500 partition workgroups: enough of them to fill 256 CUs several times over, few enough
  // that a tile's gather (one segment descriptor per chunk) stays short
  uint32_t chunk = 2048;
  while ((n + chunk - 1) / chunk > 1536 && chunk < 8192) chunk *= 2;
#ifdef GV_DIAG
  static const int forced = [] { const char *e = std::getenv("GV_BIN_CHUNK"); return e ? std::atoi(e) : 0; }();
  if (forced >= GV_PART_BATCH * kPartThreads && forced % (GV_PART_BATCH * kPartThreads) == 0) chunk = (uint32_t)forced;
#endif
  return chunk;
}

size_t bin_partition_lds(uint32_t chunk, int n_tiles) { return (size_t)chunk * 4 + (size_t)n_tiles * 4 + (size_t)chunk * 2; }

// LDS bytes of the bbox-test tables (thresholds + tile candidate masks)
size_t bin_bbox_lds(int nb_pad, const BBoxTest &bt)
{
  return 16 + (size_t)nb_pad * 16 + (size_t)bt.tiles_x * bt.tiles_y * bt.mask_words * 8;
}

bool bin_bbox_fits(int nb, const BBoxTest &bt) { return bin_bbox_lds((nb + 3) & ~3, bt) <= kBinBBoxLdsMax; }

// t0 / t1 (optional, timing-enabled events): start and end of the kernel itself, taken from its dispatch packet
// (hipExtLaunchKernelGGL) -- the per-kernel times of gv_time_frame_stages, free of the event-record overhead
void launch_bin_partition(const BinArgs &a, hipStream_t s, hipEvent_t t0, hipEvent_t t1, bool any_order)
{
  const unsigned fl = any_order ? hipExtAnyOrderLaunch : 0u;
  const uint32_t grid = a.n_wg + (a.n_rect_poses > 0 ? 1u : 0u);
  if (grid == 0) return;
  const size_t lds = bin_partition_lds(a.chunk, a.n_tiles) + (a.do_bbox ? bin_bbox_lds(a.nb_pad, a.bt) : 0);
  const bool keep = a.cell_idx != nullptr;
#define GV_BP(R, X, K) hipExtLaunchKernelGGL((k_bin_partition<R, X, K>), dim3(grid), dim3(kPartThreads), (uint32_t)lds, s, t0, t1, fl, a)
  if (a.do_ray && a.do_bbox && keep) GV_BP(true, true, true);
  else if (a.do_ray && a.do_bbox) GV_BP(true, true, false);
  else if (a.do_ray && keep) GV_BP(true, false, true);
  else if (a.do_ray) GV_BP(true, false, false);
  else if (a.do_bbox && keep) GV_BP(false, true, true);
  else if (a.do_bbox) GV_BP(false, true, false);
  else if (keep) GV_BP(false, false, true);
  else GV_BP(false, false, false);
#undef GV_BP
}

void launch_bin_tiles(const BinTileArgs &a, uint32_t n_helpers, hipStream_t s, hipEvent_t t0, hipEvent_t t1)
{
  const uint32_t grid = (uint32_t)(((a.n_tiles + 7) >> 3) << 3) + n_helpers;   // primaries by XCD run, then the helpers
  if (a.hits) hipExtLaunchKernelGGL(k_bin_tiles<true>, dim3(grid), dim3(kTileThreads), 0, s, t0, t1, 0, a);
  else hipExtLaunchKernelGGL(k_bin_tiles<false>, dim3(grid), dim3(kTileThreads), 0, s, t0, t1, 0, a);
}

}  // namespace gv
