// gv_raysector.hip -- [EXTENSION] X2 free-space ray-march, sector/gather form,
// plus the tile-based grid pass that consumes it.  gfx950, wave64.
//
// Why not march every ray: 1M rays x ~860 steps is ~6.5e8 cell visits per frame.
// grid_map::LineIterator's integer stepping has a closed form: a ray from the
// origin cell O to an end at octant-local offset (a,b) (a = major, b = minor,
// 0 <= b <= a) passes, at major step i, through minor offset
//        j = (a/2 + i*b) / a            (integer divisions)
// which is equivalent to   (2j-1)*a <= 2*i*b < (2j+1)*a,   i.e. the traversed
// cell depends on the ray only through its slope b/a.  So cell (i,j) is free
//        iff  some end has slope in [(2j-1)/2i, (2j+1)/2i)  and  reach > i
// (reach = a for a hit end, a+1 for a clipped end whose own cell counts).
// That is a range-max query over the ends ordered by slope.  Each workgroup owns
// one angular sector of one octant: it collects its ends from end bitmaps,
// drops them into M slope buckets in LDS (exact integer bucket index; a row of
// kSlots per bucket plus one overflow list, no sorting passes),
// builds a sparse range-max table over the bucket maxima, and answers every cell
// of its wedge from the table plus exact cross-multiplication tests on the (at
// most two) buckets its slope interval only partly covers.  Everything is
// integer arithmetic: bit-exact against the oracle's literal march.
#include "gv_kernels.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

namespace gv {

// ------------------------------------------------------------ bitmaps ------
// End flags (hit / clipped ray end) and free-cell flags live in bitmaps of two orientations, with the
// 32-bit WORDS stored transposed so that a wavefront scanning 64 consecutive wedge columns reads (or
// ORs into) 64 adjacent words:
//   N: bits run along x, word(x>>5, y) at  (x>>5)*ny_pad + y   (y-major octants)
//   T: bits run along y, word(y>>5, x) at  (y>>5)*nx_pad + x   (x-major octants)
// nx_pad / ny_pad are multiples of 128.  The end bitmaps are written by the binning tile pass
// (gv_binning.hip), which also zeroes the free-cell bitmaps of the buffer set it is about to use.

// ------------------------------------------------ wavefront primitives (DPP) --
// Cross-lane steps as DPP modifiers of VALU ops (gfx9: row_shr, wave_shl:1, row_bcast:15/31)
// instead of ds_bpermute round trips through the LDS crossbar: a 64-lane scan is six dependent
// VALU ops (~tens of cycles) rather than six LDS latencies (~hundreds).
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned src)
{
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}
struct OpAdd { static constexpr unsigned id = 0u; __device__ static unsigned f(unsigned a, unsigned b) { return a + b; } };
struct OpMax { static constexpr unsigned id = 0u; __device__ static unsigned f(unsigned a, unsigned b) { return a > b ? a : b; } };
struct OpMin { static constexpr unsigned id = 0xFFFFFFFFu; __device__ static unsigned f(unsigned a, unsigned b) { return a < b ? a : b; } };
// inclusive scan over the 64 lanes (lane 63 ends up with the reduction)
template <class Op>
__device__ __forceinline__ unsigned wave_scan(unsigned v)
{
  v = Op::f(v, dpp_u32<0x111>(Op::id, v));        // row_shr:1
  v = Op::f(v, dpp_u32<0x112>(Op::id, v));        // row_shr:2
  v = Op::f(v, dpp_u32<0x114>(Op::id, v));        // row_shr:4
  v = Op::f(v, dpp_u32<0x118>(Op::id, v));        // row_shr:8
  v = Op::f(v, dpp_u32<0x142, 0xA>(Op::id, v));   // row_bcast:15 -> rows 1, 3
  v = Op::f(v, dpp_u32<0x143, 0xC>(Op::id, v));   // row_bcast:31 -> rows 2, 3
  return v;
}
template <class Op>
__device__ __forceinline__ unsigned wave_reduce(unsigned v)   // same value in every lane
{
  return (unsigned)__builtin_amdgcn_readlane((int)wave_scan<Op>(v), 63);
}
// value of lane + H (H = 1, 2, 4, 8) or `fill` past lane 63: H single-lane wave shifts
template <int H>
__device__ __forceinline__ unsigned wave_shift_down(unsigned v, unsigned fill)
{
#pragma unroll
  for (int k = 0; k < H; ++k) v = dpp_u32<0x130>(fill, v);   // wave_shl:1
  return v;
}

// Products of small non-negative / small signed integers (wedge offsets, sector numbers, slopes: all far below
// 2^23): v_mul_u32_u24 / v_mul_i32_i24 issue at the full vector rate, v_mul_lo_u32 at half of it (measured,
// tools/microbench/valu_rate.hip: 590 against 1010 G wavefront-instructions/s).
__device__ __forceinline__ int m24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ unsigned um24(unsigned a, unsigned b) { return __umul24(a, b); }

// ------------------------------------------------------- sector gather -----
struct Oct {
  int xmaj, smaj, smin;   // major axis is x?; signs of the major / minor step
  int imax, jmaxo;        // last in-map major / minor offset
  int bmin;               // minor offsets start at 1 on the negative side (delta >= 0 is "+")
};

__device__ __forceinline__ Oct make_octant(int o, const GridParams &g, const RayOrigin &org)
{
  Oct c;
  c.xmaj = (o >> 2) & 1;
  c.smaj = ((o >> 1) & 1) ? 1 : -1;
  c.smin = (o & 1) ? 1 : -1;
  if (c.xmaj) {
    c.imax = (c.smaj > 0) ? g.nx - 1 - org.cx : org.cx;
    c.jmaxo = (c.smin > 0) ? g.ny - 1 - org.cy : org.cy;
  } else {
    c.imax = (c.smaj > 0) ? g.ny - 1 - org.cy : org.cy;
    c.jmaxo = (c.smin > 0) ? g.nx - 1 - org.cx : org.cx;
  }
  c.bmin = (c.smin > 0) ? 0 : 1;
  return c;
}

// packed end: a(13) | b(13) | inclusive(1)
__device__ __forceinline__ unsigned pack_ab(int a, int b, int incl) { return ((unsigned)a << 14) | ((unsigned)b << 1) | (unsigned)incl; }
__device__ __forceinline__ int ab_a(unsigned p) { return (int)(p >> 14) & 0x1FFF; }
__device__ __forceinline__ int ab_b(unsigned p) { return (int)(p >> 1) & 0x1FFF; }

constexpr int kSecThreads = 512;   // 8 wavefronts per sector workgroup
// Timing experiments (phase ablation, in-kernel clock stamps) exist only in the diagnostic build
// (-DGV_DIAG, tools/): the production kernel carries none of their branches or stores.
#ifdef GV_DIAG
#define GV_ABL(bit) (((A.ablate) & (bit)) != 0)
#else
#define GV_ABL(bit) (false)
#endif
#ifndef GV_SECTOR_WPE
#define GV_SECTOR_WPE 4             // min waves per SIMD the register allocator must allow
#endif

// Ends of a slope bucket: the first kSlots in the bucket's own row of a table, in arrival order; what a crowded bucket
// holds beyond that (the buckets of the few rational slopes many ends share) goes to one overflow list.  No count /
// prefix / placement passes: an end is placed the moment its bucket is known.
constexpr int kSlotLog = 3, kSlots = 1 << kSlotLog;
#ifndef GV_SLOT_TOTAL
#define GV_SLOT_TOTAL 2048
#endif
constexpr int kSlotTotal = GV_SLOT_TOTAL;   // a wedge with more ends than this (4 per bucket: measured, config 3 and config 5) is placed by counting sort instead

// LDS layout of one sector workgroup (bytes), shared by the kernel and the launcher
struct SectorLds {
  size_t tab, marks, cnt, bstart, bmax32, lvl, pfx, sfx, raw, over, total;
};
__host__ __device__ inline SectorLds sector_lds_layout(int cap, int marks_words, int log2m)
{
  const size_t M = (size_t)1 << log2m;
  SectorLds L;
  size_t o = 0;
  L.tab = o;    o += (M * kSlots > (size_t)cap ? M * kSlots : (size_t)cap) * 4;   // [M][kSlots] packed ends (16-byte aligned rows); a crowded group: cap ends grouped by bucket
  L.raw = o;    o += (size_t)cap * 4;           // packed ends in scan order (staging list; the long-ray pass reads it too)
  L.over = o;   o += (size_t)(cap > 2 * kSlotTotal ? cap : 2 * kSlotTotal) * 4;   // overflow entries: bucket << 16 | index into raw -- two lists of
                                                // kSlotTotal (as placed / after the same-slope merge); a crowded group: cap bucket numbers (16 bit)
  L.marks = o;  o += (size_t)marks_words * 4;
  L.cnt = o;    o += M * 4;                     // ends per bucket (may exceed kSlots); a crowded group: the end of every bucket's run
  L.bstart = o; o += M * 4;                     // a crowded group: the start of every bucket's run
  L.bmax32 = o; o += 2 * M * 4;                 // bucket max reach; dead once the range-max tables are built: then the long-ray list (2M entries)
  L.lvl = o;
  L.pfx = L.lvl + 7 * M * 2;                    // in-block (64 buckets) sparse levels 0..6
  L.sfx = L.pfx + M * 2;
  o += 9 * M * 2;
  L.total = (o + 15) & ~(size_t)15;
  return L;
}

template <int CH>
__global__ void __launch_bounds__(kSecThreads, GV_SECTOR_WPE) k_ray_sectors(SectorArgs A)
{
  constexpr int NT = kSecThreads;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  GV_TL_BEGIN(A.tl);
  // position in the dispatch order: a multi-GPU rank runs every wg_stride-th workgroup
  const int bid_all = A.wg_first + (int)blockIdx.x * A.wg_stride;
  // Dispatch order (workgroups start roughly in blockIdx order and the launch is ~2 rounds deep):
  // helpers first (below), then octants with the longest wedges, and inside an octant the sectors next to
  // the slopes 0, 1/2, 1 first (they run longest).  Every octant has its own sector count: a short wedge
  // (origin close to that map edge) split into as many sectors as a long one would be all
  // fixed per-workgroup cost and no ends.
  // Helpers: the first and the last sector of an octant (next to the axis / the diagonal) are the heaviest by
  // far -- the ends there sit on few rational slopes, the "every bucket group holds a long ray" run ends early
  // and hundreds of columns are evaluated cell by cell (70 k cycles against a 36 k mean: they set the kernel's
  // makespan).  Each of them is run by TWO workgroups that build the same tables and take every second column
  // of the evaluation.  Helper 2q + e = octant q of the dispatch order, sector e ? S - 1 : 0, odd columns.
  const bool helper = bid_all < A.n_helpers;
  const int bid = helper ? (int)A.wg_base[bid_all >> 1] : bid_all - A.n_helpers;   // (helpers: only k_oct is taken from it)
  int k_oct = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k) k_oct += (bid >= (int)A.wg_base[k]) ? 1 : 0;
  const int o = A.reorder ? ((int)(A.oct_perm >> (3 * k_oct)) & 7) : k_oct;
  const int log2s = A.log2s_oct[o];
  const int S = 1 << log2s;
  int s = bid - (int)A.wg_base[k_oct];
  if (A.reorder && S >= 8) {
    const int r = s;
    if (r < 4) s = (r == 0) ? 0 : (r == 1) ? (S >> 1) - 1 : (r == 2) ? S - 1 : (S >> 1);
    else s = (r - 4 < (S >> 1) - 2) ? r - 3 : r - 1;
  }
  if (helper) {
    s = (bid_all & 1) ? S - 1 : 0;
    if (S < 2 && (bid_all & 1)) return;   // one sector only: it has one helper
  }
  // column share: workgroup `part` of `nparts` evaluates the columns i with i % nparts == part
  const int nparts = (A.n_helpers > 0 && (s == 0 || s == S - 1)) ? 2 : 1;
  const int part = helper ? 1 : 0;
  const int stat_slot = helper ? (int)A.wg_base[8] + bid_all : bid;
  const int wg = stat_slot;   // diagnostics slot
  // a clipped ray that ends in the origin cell itself (a == 0, inclusive end)
  if (bid == 0 && !helper && threadIdx.x == 0) {
    const size_t wo = (size_t)(A.org.cx >> 5) * A.ny_pad + A.org.cy;
    if ((A.clipN[wo] >> (A.org.cx & 31)) & 1u) atomicOr(&A.freeN[wo], 1u << (A.org.cx & 31));
  }
  const Oct oc = make_octant(o, A.g, A.org);
  if (oc.imax < 1) {
    if (threadIdx.x == 0 && A.stats) { A.stats[2 * stat_slot] = 0; A.stats[2 * stat_slot + 1] = 0; }
    return;
  }
  const int cap = A.cap;
  const int LM = A.log2m, M = 1 << LM, NB = (M + 63) >> 6;

  const SectorLds L = sector_lds_layout(cap, A.marks_words, LM);
  unsigned *tab = reinterpret_cast<unsigned *>(smem + L.tab);          // packed ends by slope bucket: kSlots per bucket
  unsigned *over = reinterpret_cast<unsigned *>(smem + L.over);        // what does not fit: bucket << 16 | index into raw
  unsigned *over2 = over + kSlotTotal;                                  // ... and what is left of it after the same-slope merge
  unsigned *marks = reinterpret_cast<unsigned *>(smem + L.marks);      // one word of cell bits per wedge column
  unsigned *cnt = reinterpret_cast<unsigned *>(smem + L.cnt);          // ends per bucket
  unsigned *bstart = reinterpret_cast<unsigned *>(smem + L.bstart);    // crowded group: bucket run starts
  unsigned *bmax32 = reinterpret_cast<unsigned *>(smem + L.bmax32);    // bucket max reach; later the long-ray list
  unsigned short *lvl = reinterpret_cast<unsigned short *>(smem + L.lvl);   // lvl[l*M+m] = max of buckets m..m+2^l-1 inside m's 64-block
  unsigned short *pfx = reinterpret_cast<unsigned short *>(smem + L.pfx);   // max from the block start to m
  unsigned short *sfx = reinterpret_cast<unsigned short *>(smem + L.sfx);   // max from m to the block end
  unsigned *raw = reinterpret_cast<unsigned *>(smem + L.raw);               // packed ends, scan order
  __shared__ unsigned s_wsum[NT / 64], s_wsum2[NT / 64], s_blkmax[8], s_blkpfx[9], s_blksfx[9], s_lvlmin[10 * 8], s_nlong, s_nover, s_nover2, s_T, s_maxreach;
  __shared__ unsigned long long s_wvis[NT / 64];
  __shared__ unsigned s_rowpart[CH][NT / 64];   // multi-group path: ends per (row, wavefront)

  // diagnostic build only (-DGV_DIAG with GV_SECTOR_DBG=1): thread 0 stamps the shader clock at
  // phase boundaries into a debug buffer nothing else reads
#ifdef GV_DIAG
  int stamp_n = 0;
  auto stamp = [&]() {
    if (A.dbg && tid == 0 && stamp_n < 16) A.dbg[(size_t)wg * 16 + stamp_n] = __builtin_amdgcn_s_memtime();
    ++stamp_n;
  };
  auto fstamp = [&](int n) {   // fine stamps: second half of the buffer
    if (A.dbg && tid == 0) A.dbg[((size_t)16384 + wg) * 16 + n] = __builtin_amdgcn_s_memtime();
  };
  auto fnote = [&](int n, unsigned long long v) {
    if (A.dbg && tid == 0) A.dbg[((size_t)16384 + wg) * 16 + n] = v;
  };
  stamp();
  if (A.dbg && tid == 0) A.dbg[(size_t)wg * 16 + 14] = ((unsigned long long)o << 32) | (unsigned)s | ((unsigned long long)log2s << 40);
#else
  auto stamp = []() {};
  auto fstamp = [](int) {};
  auto fnote = [](int, unsigned long long) {};
  (void)wg;
#endif
  for (int i = tid; i <= oc.imax; i += NT) marks[i] = 0;
  if (tid < NT / 64) s_wvis[tid] = 0;
  for (int m = tid; m < M; m += NT) { cnt[m] = 0; bmax32[m] = 0; }
  if (tid == 0) { s_nlong = 0; s_nover = 0; s_nover2 = 0; }
  // (no barrier here: nothing reads what was just cleared before the barrier behind the column scan below, and
  //  the scan's bitmap loads go out without waiting for the slowest wavefront's stores)
#ifdef GV_DIAG
  if (A.dbg || A.ablate) __syncthreads();
#endif
  stamp();   // 1: init done
  if GV_ABL(8) return;    // timing experiment: launch + init only

  const unsigned *bmH = oc.xmaj ? A.hitT : A.hitN;
  const unsigned *bmC = oc.xmaj ? A.clipT : A.clipN;
  const int roww = oc.xmaj ? A.nyw : A.nxw;                 // words along the minor axis
  const unsigned major_pad = (unsigned)(oc.xmaj ? A.nx_pad : A.ny_pad);   // word-row stride
  const int oc_major = oc.xmaj ? A.org.cx : A.org.cy;
  const int oc_minor = oc.xmaj ? A.org.cy : A.org.cx;

  // exact floor(N / Q) for 0 <= N < 2^24, 0 < Q: float estimate + one integer correction
  auto idiv = [](int N, int Q) -> int {
    int q = (int)((float)N * __builtin_amdgcn_rcpf((float)Q));
    const int r = N - m24(q, Q);
    if (r < 0) --q;
    else if (r >= Q) ++q;
    return q;
  };
  // wavefront sum / min (no same-address LDS atomics: those serialise)
  auto wave_sum = [](unsigned v) -> unsigned { return wave_reduce<OpAdd>(v); };
  auto wave_min = [](unsigned v) -> unsigned { return wave_reduce<OpMin>(v); };
  // slope bucket of an end: floor((b*S - a*s) * M / a); slope 1 (last sector) -> M-1
  auto bucket_of_end = [&](int a, int b) -> int {
    const int rel = m24(b, S) - m24(a, s);
    return (rel >= a) ? (M - 1) : idiv(rel << LM, a);
  };
  // bucket holding the boundary slope P/Q: -1 below the sector, M at/above its end
  auto bucket_of_boundary = [&](int P, int Q) -> int {
    const int relb = m24(P, S) - m24(s, Q);
    return (relb <= 0) ? -1 : (relb >= Q) ? M : idiv(relb << LM, Q);
  };
  // max reach over whole buckets [l, r] (l <= r)
  auto rmq = [&](int l, int r) -> unsigned {
    const int bl = l >> 6, br = r >> 6;
    if (bl == br) {
      const int lev = 31 - __clz(r - l + 1);
      return max((unsigned)lvl[(lev << LM) + l], (unsigned)lvl[(lev << LM) + r - (1 << lev) + 1]);
    }
    unsigned mx = max((unsigned)sfx[l], (unsigned)pfx[r]);
    for (int bk = bl + 1; bk < br; ++bk) mx = max(mx, s_blkmax[bk]);
    return mx;
  };

  // largest reach among the c ends of bucket m whose slope b/a lies in [Plo/Q, Phi/Q); lo_open / hi_open: no
  // bound on that side.  The bucket's row of the table is read whole (two 16-byte reads issued together); the
  // overflow list is only looked at for a bucket that holds more than its row.
  // `mode`: std::true_type = the group's buckets are contiguous runs of `tab` (counting-sort placement, crowded
  // wedges), std::false_type = rows of kSlots + overflow list (everything else)
  // the overflow list the walks scan: as placed (short lists) or after the same-slope merge
  const unsigned *ovl = over;
  unsigned novl = 0;
#ifndef GV_MERGE_MIN
#define GV_MERGE_MIN 16
#endif
  constexpr unsigned kMergeMin = GV_MERGE_MIN;   // a list shorter than this is scanned as it is: the merge pass is ~1.3 k cycles of
                                                 // the workgroup's chain (profiles/r04/sector_round4.txt)
  auto walk_bucket = [&](auto mode, int m, unsigned e0, unsigned c, int Q, bool lo_open, int Plo, bool hi_open, int Phi) -> unsigned {
    constexpr bool sorted_mode = decltype(mode)::value;
    unsigned mx = 0;
    auto take = [&](unsigned p, bool valid) {
      const int a = ab_a(p), bq = m24(ab_b(p), Q);
      const bool in = valid && (lo_open || bq >= m24(Plo, a)) && (hi_open || bq < m24(Phi, a));
      mx = in ? max(mx, (unsigned)(a + (int)(p & 1u))) : mx;
    };
    if constexpr (sorted_mode) {   // c = end of the run: four ends per step, the LDS reads of a step issued together
      const unsigned e1 = c;
      for (unsigned e = e0; e < e1; e += 4) {   // (e0 = bstart[m], read by the caller together with its other table reads)
        unsigned p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = tab[min(e + (unsigned)q, e1 - 1u)];
#pragma unroll
        for (int q = 0; q < 4; ++q) take(p[q], e + (unsigned)q < e1);
      }
      return mx;
    } else {
    const uint4 r0 = *reinterpret_cast<const uint4 *>(tab + ((unsigned)m << kSlotLog));
    const uint4 r1 = *reinterpret_cast<const uint4 *>(tab + ((unsigned)m << kSlotLog) + 4);
    take(r0.x, c > 0u); take(r0.y, c > 1u); take(r0.z, c > 2u); take(r0.w, c > 3u);
    if (c > 4u) { take(r1.x, true); take(r1.y, c > 5u); take(r1.z, c > 6u); take(r1.w, c > 7u); }
    if (c > (unsigned)kSlots && !GV_ABL(4096)) {
      const unsigned no = novl;
      for (unsigned e = 0; e < no; e += 4) {   // four entries per step, their reads issued together (the list's space is
        unsigned v[4];                          // readable past the end: stale words are masked by e + q < no)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ovl[e + (unsigned)q];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (e + (unsigned)q < no && (int)(v[q] >> 16) == m) take(raw[v[q] & 0xFFFFu], true);
      }
    }
    return mx;
    }
  };
  // one cell (column i, minor offset jc), exactly: whole buckets between its two boundary buckets by
  // range-max, the boundary buckets themselves end by end
  auto cell_exact = [&](auto mode, int i, int Q, int jc) -> bool {
    const int Plo = 2 * jc - 1, Phi = 2 * jc + 1;
    const int lo = bucket_of_boundary(Plo, Q), hi = bucket_of_boundary(Phi, Q);   // lo < hi unless both outside
    unsigned mx = 0;
    const int l = lo + 1, r = min(hi - 1, M - 1);
    // the boundary buckets' bounds and own maxima are requested together with the range-max reads
    const int loc = min(max(lo, 0), M - 1), hic = min(max(hi, 0), M - 1);
    const unsigned cl = cnt[loc], ch = cnt[hic];
    unsigned sl = 0, sh = 0;
    if constexpr (decltype(mode)::value) { sl = bstart[loc]; sh = bstart[hic]; }
    const unsigned lvlo = lvl[loc], lvh = lvl[hic];
    if (l <= r) mx = rmq(l, r);
    // a boundary bucket is only walked when its own max reach (level 0 of the range-max table) says that
    // one of its ends could decide the cell: beyond the threshold column most buckets hold short rays only
    if (mx <= (unsigned)i && lo >= 0 && lo < M && lvlo > (unsigned)i)
      mx = max(mx, walk_bucket(mode, loc, sl, cl, Q, false, Plo, hi != lo, Phi));
    if (mx <= (unsigned)i && hi >= 0 && hi < M && hi != lo && lvh > (unsigned)i)
      mx = max(mx, walk_bucket(mode, hic, sh, ch, Q, true, 0, false, Phi));
    return mx > (unsigned)i;
  };
  // minor-offset range [blo, bhi] of the wedge in column a (empty when blo > bhi)
  auto col_bounds = [&](int a, int &blo, int &bhi) {
    const int bmaxa = oc.xmaj ? a : a - 1;
    blo = (m24(a, s) + S - 1) >> log2s;
    bhi = ((m24(a, s + 1) + S - 1) >> log2s) - 1;
    if (s == S - 1) bhi = bmaxa;
    blo = max(blo, oc.bmin);
    bhi = min(min(bhi, bmaxa), oc.jmaxo);
  };
  // Column slot c of a thread: column 1 + NT*c + tid, or -- rows the host marks in rev_oct -- 1 + NT*c + (NT-1-tid).  A
  // wavefront still holds 64 consecutive columns of a row (coalesced bitmap words), but the row's 8 blocks go to the
  // wavefronts in descending order: far columns are wider (more cells, more ends), and with every row ascending the last
  // wavefront of a 1000-column wedge held 2.5 x the ends of the first and the other seven waited ~3.8 k cycles at the barrier
  // behind the append loop (profiles/r04/sector_phases_fine.txt).
  const unsigned rev = A.rev_oct[o];
  auto col_of = [&](int c) -> int { return 1 + NT * c + (((rev >> c) & 1u) ? NT - 1 - tid : tid); };
  // ---- scan: CH columns per thread (a = col_of(c)), bitmap loads issued back to back.
  // ends / vcs live only from here to the staging loop of the group that consumes them; the
  // (rare) multi-group path re-reads its rows instead of keeping CH columns in registers.
  unsigned ends[CH], vcs[CH];
  auto scan_columns = [&](unsigned rowmask) {
    unsigned h0[CH], h1[CH], c0[CH], c1[CH];
    int shs[CH], ws[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      h0[c] = h1[c] = c0[c] = c1[c] = 0u;
      shs[c] = ws[c] = 0;
      // a row of column slots that lies wholly beyond this octant's wedge (the kernel is instantiated for the
      // longest octant): nothing to load -- a wavefront-uniform skip
      if (NT * c >= oc.imax) continue;
      const int a = col_of(c);
      const bool in = (a <= oc.imax) && ((rowmask >> c) & 1u);
      const int ac = in ? a : 1;
      int blo, bhi;
      col_bounds(ac, blo, bhi);
      const bool ok = in && blo <= bhi;
      const int w = ok ? bhi - blo + 1 : 0;   // <= 32 (host guarantees imax <= 30*S)
      const int major_abs = oc_major + m24(oc.smaj, ac);
      const int m_lo = ok ? ((oc.smin > 0) ? oc_minor + blo : oc_minor - bhi) : 0;
      const int w0 = m_lo >> 5;
      const unsigned base = um24((unsigned)w0, major_pad) + (unsigned)major_abs;   // < 2^21 words
      const unsigned base1 = (w0 + 1 < roww) ? base + major_pad : base;
      h0[c] = bmH[base]; c0[c] = bmC[base];
      h1[c] = bmH[base1]; c1[c] = bmC[base1];
      shs[c] = m_lo & 31;
      ws[c] = w;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const unsigned long long h64 = ((unsigned long long)h1[c] << 32) | h0[c];
      const unsigned long long c64 = ((unsigned long long)c1[c] << 32) | c0[c];
      const unsigned wm = (ws[c] >= 32) ? 0xFFFFFFFFu : ((1u << ws[c]) - 1u);
      const unsigned vh = (unsigned)(h64 >> shs[c]) & wm;
      vcs[c] = (unsigned)(c64 >> shs[c]) & wm;
      ends[c] = vh | vcs[c];   // bit t <-> b = blo + t (positive minor side) or bhi - t
    }
  };
  scan_columns((1u << CH) - 1u);
  // one row only (multi-group path): the other rows read as empty
  auto scan_row = [&](int c) {
    const int a = col_of(c);
    unsigned e = 0, vc = 0;
    int blo, bhi;
    col_bounds(a, blo, bhi);
    if (a <= oc.imax && blo <= bhi) {
      const int w = bhi - blo + 1;
      const int major_abs = oc_major + m24(oc.smaj, a);
      const int m_lo = (oc.smin > 0) ? oc_minor + blo : oc_minor - bhi;
      const int w0 = m_lo >> 5;
      const unsigned base = um24((unsigned)w0, major_pad) + (unsigned)major_abs;
      const unsigned base1 = (w0 + 1 < roww) ? base + major_pad : base;
      const unsigned long long h64 = ((unsigned long long)bmH[base1] << 32) | bmH[base];
      const unsigned long long c64 = ((unsigned long long)bmC[base1] << 32) | bmC[base];
      const unsigned wm = (w >= 32) ? 0xFFFFFFFFu : ((1u << w) - 1u);
      vc = (unsigned)(c64 >> (m_lo & 31)) & wm;
      e = ((unsigned)(h64 >> (m_lo & 31)) & wm) | vc;
    }
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      ends[cc] = (cc == c) ? e : 0u;
      vcs[cc] = (cc == c) ? vc : 0u;
    }
  };
  // ends per thread -> inclusive prefix inside the wavefront, wavefront totals in LDS: gives
  // the total AND (fast path) every thread's slot range without another barrier
  unsigned scan_mine = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) scan_mine += (unsigned)__popc(ends[c]);
  const unsigned scan_incl = wave_scan<OpAdd>(scan_mine);
  if (lane == 63) s_wsum[wave] = scan_incl;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int wv = 0; wv < NT / 64; ++wv) total += (int)s_wsum[wv];
  stamp();   // 2: scan done
  if GV_ABL(16) return;   // timing experiment: + scan
  if (total == 0) {
    if (tid == 0 && A.stats) { A.stats[2 * stat_slot] = 0; A.stats[2 * stat_slot + 1] = 0; }
    return;
  }

  // One group = the ends of the selected (row c, wavefront) sets; at most cap of them.
  // rowmask: rows taking part; wsel: -1 = every wavefront, else only that one.
  // highest level of aligned bucket groups a column of this wedge can ask for
  const int tqmax = (2 * oc.imax + S - 1) >> log2s;
  const int lv_max = (tqmax <= 1) ? 0 : 32 - __clz(tqmax - 1);
  // Lattice gaps.  Ends are integer points (a, b), a <= imax, so no end has a slope strictly
  // within 1/(q*imax) of a rational p/q without being p/q itself.  A sector that starts at slope 0
  // or 1/2 (ends at 1/2 or 1) therefore has a run of slope buckets next to that boundary that is
  // empty whatever the cloud: Z*M buckets, Z = S/(q*imax).  The "every aligned bucket group holds a
  // long ray" test must not count groups inside such a run (they would pin T at 512 or 1024 for good,
  // and those 16 workgroups would run 2.5x the mean), and because a run is narrower than one cell
  // (1/(q*imax) <= 1/i) only the cell next to the edge cell can reach into it: that one is
  // evaluated exactly below (cell 1 resp. w-2), like the edge cells themselves.
  int xb0 = 0, xb1 = 0, xt0 = 0, xt1 = 0;   // excluded buckets [xb0, xb1) at the bottom, [xt0, xt1) at the top
  if (S >= 8) {
    const unsigned SM = (unsigned)S << LM;   // S <= 2^12 (the host), M <= 2^9: 32 bits (a 64-bit division here was ~4 k cycles
                                             // of these four sectors' -- the heaviest ones' -- chain)
    const int qlo = (s == 0) ? 1 : ((s == (S >> 1)) ? 2 : 0);
    const int qhi = (s == S - 1) ? 1 : ((s == (S >> 1) - 1) ? 2 : 0);
    if (qlo) {   // bucket k lies inside (0, Z) iff k + 1 <= Z*M; bucket 0 also holds the rational itself
      xb0 = (s == 0 && oc.bmin == 1) ? 0 : 1;
      xb1 = (int)min((unsigned)M, SM / (unsigned)(qlo * oc.imax));
    }
    if (qhi) {   // bucket k lies inside (1 - Z, 1) iff k > M - Z*M; the last bucket of sector S-1 holds slope 1,
                 // which only the x-major octants own (the diagonal is theirs: bmaxa)
      const unsigned den = (unsigned)(qhi * oc.imax);
      const int c = (int)((SM + den - 1u) / den);   // ceil(Z*M)
      xt0 = max(0, M - c + 1);
      xt1 = (s == S - 1 && oc.xmaj) ? M - 1 : M;
    }
  }
  auto gap_group = [&](int g0, int size) -> bool {   // aligned group [g0, g0+size) entirely inside a gap run
    return (g0 >= xb0 && g0 + size <= xb1) || (g0 >= xt0 && g0 + size <= xt1);
  };
  auto process_group = [&](auto mode, unsigned rowmask, int wsel, bool first) {
    constexpr bool sorted_mode = decltype(mode)::value;
    const bool mine_w = (wsel < 0) || (wave == wsel);
    unsigned mycnt, incl;
    if (first) {   // whole wedge in one group: the scan already produced the prefix and s_wsum
      mycnt = scan_mine;
      incl = scan_incl;
    } else {
      __syncthreads();   // previous group fully done with the tables
      for (int m = tid; m < M; m += NT) { cnt[m] = 0; bmax32[m] = 0; }
      if (tid == 0) { s_nlong = 0; s_nover = 0; s_nover2 = 0; }
      mycnt = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (((rowmask >> c) & 1u) && mine_w) mycnt += (unsigned)__popc(ends[c]);
      incl = wave_scan<OpAdd>(mycnt);
      if (lane == 63) s_wsum[wave] = incl;
      __syncthreads();
    }
    fstamp(0);
    // slots of the dense staging list from the wavefront prefix sums
    unsigned slot = incl - mycnt;
    unsigned n = 0;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) {
      if (wv < wave) slot += s_wsum[wv];
      n += s_wsum[wv];
    }
    // append the packed ends in scan order (a lane holds 0..32 ends of its columns: only the cheap packing runs
    // in this lane-unbalanced loop) ...
    fstamp(1);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!((rowmask >> c) & 1u) || !mine_w || NT * c >= oc.imax) continue;
      const int a = col_of(c);
      unsigned e = ends[c];
      int blo, bhi;
      col_bounds(a, blo, bhi);
      while (e) {
        const int t = __ffs(e) - 1;
        e &= e - 1;
        const int b = (oc.smin > 0) ? blo + t : bhi - t;
        raw[slot++] = pack_ab(a, b, (int)((vcs[c] >> t) & 1u));
      }
    }
    fstamp(2);
    __syncthreads();
    stamp();   // 3: ends appended
    // ... then, over the dense list with one end per lane and step: its slope bucket (an integer division), its place
    // in the bucket's row (or the overflow list), the bucket's max reach, the visit statistics
    {
      unsigned vis = 0;   // <= 9 ends x reach < 2^13 per thread: the wavefront sum fits 32 bits
      for (unsigned k = tid; k < n; k += NT) {
        const unsigned p = raw[k];
        const int bk = bucket_of_end(ab_a(p), ab_b(p));
        const unsigned rch = (unsigned)(ab_a(p) + (int)(p & 1u));
        if constexpr (sorted_mode) {
          reinterpret_cast<unsigned short *>(over)[k] = (unsigned short)bk;   // (the overflow list's space: unused in this mode)
          atomicAdd(&cnt[bk], 1u);
        } else {
          const unsigned slot_in = atomicAdd(&cnt[bk], 1u);
          if (slot_in < (unsigned)kSlots) tab[((unsigned)bk << kSlotLog) + slot_in] = p;
          else over[atomicAdd(&s_nover, 1u)] = ((unsigned)bk << 16) | k;
        }
        atomicMax(&bmax32[bk], rch);
        vis += rch;
      }
      vis = wave_sum(vis);
      if (lane == 0 && vis) s_wvis[wave] += vis;   // one owner per slot
    }
    __syncthreads();
    stamp();   // 4: placed
#ifdef GV_DIAG
    if (A.dbg) {   // largest bucket, buckets beyond their row
      unsigned c0 = 0, big = 0;
      for (int m = tid; m < M; m += NT) { c0 = max(c0, cnt[m]); big += cnt[m] > (unsigned)kSlots ? 1u : 0u; }
      c0 = wave_reduce<OpMax>(c0); big = wave_reduce<OpAdd>(big);
      if (lane == 0) { s_wsum2[wave] = c0 | (big << 16); }
      __syncthreads();
      if (tid == 0) {
        unsigned mc = 0, nb = 0;
        for (int wv = 0; wv < NT / 64; ++wv) { mc = max(mc, s_wsum2[wv] & 0xFFFFu); nb += s_wsum2[wv] >> 16; }
        A.dbg[(size_t)wg * 16 + 15] = ((unsigned long long)(sorted_mode ? 1 : 0) << 32) | s_nover | ((unsigned long long)mc << 40) | ((unsigned long long)nb << 52);
      }
      __syncthreads();
    }
#endif
    // A crowded wedge (thousands of ends per group: most buckets hold more than their row) would push half its ends
    // into the overflow list, which every walk of a full bucket scans: it is placed by counting sort instead --
    // prefix over the bucket counts, every end into its bucket's contiguous run.
    if constexpr (sorted_mode) {
      {
        const unsigned bc = (tid < M) ? cnt[tid] : 0u;
        const unsigned in2 = wave_scan<OpAdd>(bc);
        if (lane == 63) s_wsum2[wave] = in2;
        __syncthreads();
        unsigned base = 0;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv) {
          const unsigned t = s_wsum2[wv];
          if (wv < wave) base += t;
        }
        if (tid < M) {
          bstart[tid] = base + in2 - bc;
          cnt[tid] = base + in2 - bc;   // placement cursor; ends up as the end of the run
        }
      }
      __syncthreads();
      for (unsigned k = tid; k < n; k += NT) {
        tab[atomicAdd(&cnt[reinterpret_cast<const unsigned short *>(over)[k]], 1u)] = raw[k];
      }
      __syncthreads();
    }
    // Row mode: what crowds a bucket beyond its row are ends on ONE rational slope p/q with a small q (every k-th lattice
    // point of that line: imax / q candidates), and of ends with equal slope only the longest can decide a cell.  Every
    // overflow entry whose slope a row entry shares is merged into that row entry (atomicMax on the packed word: equal
    // slope, larger a = larger word, reach not smaller) and leaves the list; what stays -- ends of other slopes in a
    // bucket that is simply full -- is a few entries per wedge, and a walk of a full bucket scans those instead of
    // 40-180 (measured: that scan was 13-18 k of the 25-31 k cycles of the heaviest middle sectors' evaluation phase,
    // profiles/r04/sector_overflow_merge.txt).  No barrier of its own: rows, list and counts are next read behind the
    // barrier that follows the range-max build.
    if constexpr (!sorted_mode) {
      const unsigned no_all = s_nover;
      const unsigned no = (no_all < kMergeMin || GV_ABL(8192)) ? 0u : no_all;
      for (unsigned e0 = 0; e0 < no; e0 += NT) {
        const unsigned e = e0 + tid;
        bool live = false;
        unsigned v = 0;
        if (e < no) {
          v = over[e];
          const unsigned m = v >> 16, p = raw[v & 0xFFFFu];
          const int a = ab_a(p), b = ab_b(p);
          const uint4 r0 = *reinterpret_cast<const uint4 *>(tab + (m << kSlotLog));
          const uint4 r1 = *reinterpret_cast<const uint4 *>(tab + (m << kSlotLog) + 4);
          const unsigned row[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          int hit = -1;
#pragma unroll
          for (int q = kSlots - 1; q >= 0; --q)
            if (m24(b, ab_a(row[q])) == m24(ab_b(row[q]), a)) hit = q;
          if (hit >= 0) {
            atomicMax(&tab[(m << kSlotLog) + (unsigned)hit], p);
            atomicSub(&cnt[m], 1u);
          } else {
            live = true;
          }
        }
        const unsigned long long bm = __ballot(live);
        if (bm) {
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(&s_nover2, (unsigned)__popcll(bm));
          base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
          if (live) over2[base + (unsigned)__popcll(bm & ((1ull << lane) - 1ull))] = v;
        }
      }
    }
    stamp();   // 5: (crowded groups: prefix + placement; the others: same-slope merge of the overflow list)
    // range-max structure: wavefront `wave` owns the 64-bucket block `wave` (cross-lane ops only)
    if (wave < NB) {
      const int m = wave * 64 + lane;
      const unsigned v0 = (m < M) ? bmax32[m] : 0u;
      unsigned v = v0;
      if (m < M) lvl[m] = (unsigned short)v;
      // level minima: s_lvlmin[Lv * 8 + block] = min over the aligned groups of M>>Lv buckets inside
      // this block of the group's max reach (0 if a group is empty).  Groups of <= 64 buckets live
      // inside one block; coarser levels are read off s_blkmax at query time.
      // only levels Lv <= lv_max are ever queried (2^Lv >= 2i/S, i <= imax)
      if (LM <= lv_max) {
        const unsigned mn = wave_min((m < M && !gap_group(m, 1)) ? v : 0xFFFFFFFFu);
        if (lane == 0) s_lvlmin[LM * 8 + wave] = mn;
      }
#pragma unroll
      for (int l = 1; l <= 6; ++l) {
        constexpr unsigned kNone = 0u;   // max with 0 = no-op: lanes past the block end keep v
        unsigned up;
        if (l == 1) up = wave_shift_down<1>(v, kNone);
        else if (l == 2) up = wave_shift_down<2>(v, kNone);
        else if (l == 3) up = wave_shift_down<4>(v, kNone);
        else if (l == 4) up = wave_shift_down<8>(v, kNone);
        else {
          const int h = 1 << (l - 1);
          up = __shfl_down(v, h);
          if (lane + h >= 64) up = 0u;
        }
        v = max(v, up);
        if (m < M) lvl[(l << LM) + m] = (unsigned short)v;
        if (l <= LM && LM - l <= lv_max) {
          const unsigned mn = wave_min(((lane & ((1 << l) - 1)) == 0 && m < M && !gap_group(m, 1 << l)) ? v : 0xFFFFFFFFu);
          if (lane == 0) s_lvlmin[(LM - l) * 8 + wave] = mn;
        }
      }
      // in-block prefix maxima by a DPP scan; suffix maxima = the same scan on the mirrored lanes
      const unsigned p = wave_scan<OpMax>(v0);
      const unsigned q = __shfl(wave_scan<OpMax>(__shfl(v0, 63 - lane)), 63 - lane);
      if (m < M) { pfx[m] = (unsigned short)p; sfx[m] = (unsigned short)q; }
      if (lane == 63) s_blkmax[wave] = p;
    }
    __syncthreads();
    if constexpr (!sorted_mode) {
      const unsigned no_all = s_nover;
      const bool merged = no_all >= kMergeMin && !GV_ABL(8192);
      ovl = merged ? over2 : over;
      novl = merged ? s_nover2 : no_all;
    }
    stamp();   // 6: range-max built
    // T = last column up to which every interior cell is known free from the level minima
    // (the test is monotone in i).  Columns beyond T are crossed only by the few rays with
    // reach > T+1: when that is cheap, march exactly those rays over exactly those columns
    // instead of evaluating every cell there.
    // Wavefront 0 does the whole (tiny, mostly scalar) computation and publishes T and the largest
    // reach; the other seven go straight to the barrier.  One LDS read per lane fetches the
    // per-block level minima (lanes 0..55: level offset l = lane>>3, block lane&7) and the block
    // maxima (lanes 56..63); 8-lane minima by DPP, then everything is scalar (readlane).
    if (wave == 0) {
    unsigned vread = 0xFFFFFFFFu;
    {
      const int l = lane >> 3, bk = lane & 7;
      if (lane < 56) {
        const int Lv = LM - l;
        if (bk < NB && Lv >= 0 && Lv <= lv_max) vread = s_lvlmin[Lv * 8 + bk];
      } else {
        vread = (bk < NB) ? s_blkmax[bk] : 0u;
      }
    }
    unsigned vmin = vread;   // lane 8l+7 <- min over the blocks of level LM-l
    vmin = min(vmin, dpp_u32<0x111>(0xFFFFFFFFu, vmin));
    vmin = min(vmin, dpp_u32<0x112>(0xFFFFFFFFu, vmin));
    vmin = min(vmin, dpp_u32<0x114>(0xFFFFFFFFu, vmin));
    unsigned bm[8];
#pragma unroll
    for (int bk = 0; bk < 8; ++bk) bm[bk] = (unsigned)__builtin_amdgcn_readlane((int)vread, 56 + bk);
    unsigned maxreach0 = 0;
    {
      // exclusive prefix / suffix maxima over the blocks (edge-cell queries), one writer each
      unsigned run = 0;
#pragma unroll
      for (int bk = 0; bk < 8; ++bk) {
        if (lane == 0 && bk < NB) s_blkpfx[bk] = run;
        run = max(run, bm[bk]);
      }
      maxreach0 = run;
      run = 0;
#pragma unroll
      for (int bk = 7; bk >= 0; --bk) {
        if (lane == 1 && bk < NB) s_blksfx[bk] = run;
        run = max(run, bm[bk]);
      }
    }
    int T0 = 0;
    {
      // lane Lv holds level Lv: its minimum over (non-gap) groups and the last column hiL that asks
      // for it (ceil(2i/S) in (2^(Lv-1), 2^Lv]); the first level whose minimum does not clear hiL
      // ends the run of "every interior cell free" columns
      const int lv_top = min(lv_max, LM);
      const unsigned q01 = max(bm[0], bm[1]), q23 = max(bm[2], bm[3]), q45 = max(bm[4], bm[5]), q67 = max(bm[6], bm[7]);
      unsigned lm7 = gap_group(0, 128) ? 0xFFFFFFFFu : q01;   // groups of 2 whole blocks
      if (NB > 2 && !gap_group(128, 128)) lm7 = min(lm7, q23);
      if (NB > 4) {
        if (!gap_group(256, 128)) lm7 = min(lm7, q45);
        if (!gap_group(384, 128)) lm7 = min(lm7, q67);
      }
      unsigned lm8 = gap_group(0, 256) ? 0xFFFFFFFFu : max(q01, q23);   // groups of 4 whole blocks
      if (NB > 4 && !gap_group(256, 256)) lm8 = min(lm8, max(q45, q67));
      const unsigned lm9 = gap_group(0, 512) ? 0xFFFFFFFFu : maxreach0;   // one group: everything
      const int Lv = lane, l = LM - Lv;
      const unsigned fine = (unsigned)__shfl((int)vmin, (l >= 0 && l <= 6) ? l * 8 + 7 : 0);
      const unsigned lmv = (l <= 6) ? fine : (l == 7) ? lm7 : (l == 8) ? lm8 : lm9;
      const int hiL = min(oc.imax, (Lv == 0) ? (S >> 1) : (int)(((long long)S << min(Lv, 20)) >> 1));
      const bool valid = Lv <= lv_top;
      const unsigned long long failm = __ballot(valid && !(lmv > (unsigned)hiL));
      if (failm == 0ull) {
        T0 = __builtin_amdgcn_readlane(hiL, lv_top);
      } else {
        const int f = __ffsll((long long)failm) - 1;
        const int prev = (f > 0) ? __builtin_amdgcn_readlane(hiL, f - 1) : 0;
        const int hf = __builtin_amdgcn_readlane(hiL, f);
        const int lf = __builtin_amdgcn_readlane((int)lmv, f);
        T0 = max(prev, min(hf, lf - 1));
      }
      if GV_ABL(64) T0 = 0;
    }
    if (lane == 0) { s_T = (unsigned)T0; s_maxreach = maxreach0; }
    } else {
      // Meanwhile the other seven wavefronts take the NARROW columns (at most two cells wide: all of them lie
      // below column 2S): every cell of such a column is evaluated exactly whatever T turns out to be, and the
      // exact evaluation needs only the tables that are complete by now.  One (column, cell) pair per lane; in
      // the gather loop below these columns were what made wavefront 0 the last to finish.
      const int n_narrow = min(oc.imax, 2 * S - 1);
      for (int t = tid - 64; t < 2 * n_narrow; t += NT - 64) {
        const int i = 1 + (t >> 1), k = t & 1;
        if (nparts > 1 && (i & 1) != part) continue;
        const int jlo = (m24(2 * i, s) + S) >> (log2s + 1);
        const int jhi = (m24(2 * i, s + 1) + S) >> (log2s + 1);
        if (jhi - jlo > 1 || k > jhi - jlo) continue;
        if (cell_exact(mode, i, 2 * i, jlo + k)) atomicOr(&marks[i], 1u << k);
      }
    }
    __syncthreads();   // s_T, s_maxreach, s_blkpfx, s_blksfx visible
    const int T = (int)s_T;
    const unsigned maxreach = s_maxreach;
    stamp();   // 7: threshold
#ifdef GV_DIAG
    if (A.dbg && tid == 0) {
      const unsigned mr = maxreach;
      A.dbg[(size_t)wg * 16 + 12] = ((unsigned long long)T << 48) | ((unsigned long long)mr << 32) | ((unsigned long long)oc.imax << 16) | (unsigned long long)min(n, 65535u);
    }
#endif
    // long rays (reach > T+1) -> compact list in the (now free) bucket-maximum array
    bool march_tail = false;
    // A short tail (few columns beyond T, all of them narrow) is cheaper to evaluate cell by cell than to find
    // the long rays for: the compaction below is a pass over all ends plus a barrier.
    bool direct_flat = false;
    {
      const int first = T + 1, last = min(oc.imax, (int)maxreach - 1);
      if (last >= first) {
        const int wl = ((2 * last * (s + 1) + S) >> (log2s + 1)) - ((2 * last * s + S) >> (log2s + 1)) + 1;
        direct_flat = (unsigned)(last - first + 1) * (unsigned)wl <= A.flat_direct * (unsigned)nparts;
      } else {
        direct_flat = true;   // no ray reaches beyond T: nothing to do there either way
      }
    }
    if (T < oc.imax && !direct_flat && !GV_ABL(256)) {
      unsigned st = 0;
      for (unsigned k0 = 0; k0 < n; k0 += NT) {
        const unsigned k = k0 + tid;
        unsigned p = 0;
        bool lng = false;
        if (k < n) {
          p = raw[k];
          const int rch = ab_a(p) + (int)(p & 1u);
          lng = rch > T + 1;
          if (lng) st += (unsigned)(rch - (T + 1));
        }
        const unsigned long long bm = __ballot(lng);
        if (bm) {
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(&s_nlong, (unsigned)__popcll(bm));
          base = __shfl(base, 0);
          const unsigned pos = base + (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
          if (lng && pos < 2u * (unsigned)M) bmax32[pos] = p;   // (the bucket maxima are dead by now: 2M entries)
        }
      }
      const unsigned r = wave_sum(st);
      if (lane == 0) s_wsum[wave] = r;
      __syncthreads();
      unsigned tail_steps = 0;
#pragma unroll
      for (int wv = 0; wv < NT / 64; ++wv) tail_steps += s_wsum[wv];
      const unsigned nlong = s_nlong;
      march_tail = (nlong <= 2u * (unsigned)M) && (tail_steps <= A.march_limit);
      if (march_tail && A.flat_k > 0) {
        // marching costs ~ one wavefront step per 64 ray cells (+ a partial step per ray); evaluating every
        // cell beyond T exactly costs ~ flat_k times that per cell: take the cheaper one
        const int first = T + 1, last = min(oc.imax, (int)maxreach - 1);
        if (last >= first) {
          const int wl = ((2 * last * (s + 1) + S) >> (log2s + 1)) - ((2 * last * s + S) >> (log2s + 1)) + 1;
          const int wf = ((2 * first * (s + 1) + S) >> (log2s + 1)) - ((2 * first * s + S) >> (log2s + 1)) + 1;
          const unsigned ncells = (unsigned)(last - first + 1) * (unsigned)(wl + wf) / 2u;
          if ((unsigned)A.flat_k * ncells < tail_steps + 32u * nlong) march_tail = false;
        }
      }
#ifdef GV_DIAG
      if (A.dbg && tid == 0) A.dbg[(size_t)wg * 16 + 13] = ((unsigned long long)tail_steps << 32) | nlong;
#endif
      stamp();   // 8: long rays compacted
      if (march_tail) {
        // one ray per wavefront, lanes over consecutive columns: distinct LDS words
        for (unsigned r0 = wave; r0 < nlong; r0 += NT / 64) {
          const unsigned p = bmax32[r0];
          const int a = ab_a(p), b = ab_b(p);
          const int rch = a + (int)(p & 1u);
          const int half = a >> 1;
          // (a sector run by two workgroups: the lanes step over this workgroup's columns only)
          const int i0 = T + 1 + ((nparts > 1 && ((T + 1) & 1) != part) ? 1 : 0);
          for (int i = i0 + m24(lane, nparts); i < rch; i += 64 * nparts) {
            // LineIterator stepping in closed form: j = (a/2 + i*b) / a
            const int num = half + m24(i, b);
            int q = (int)((float)num * __builtin_amdgcn_rcpf((float)a));
            const int rem = num - m24(q, a);
            if (rem < 0) --q;
            else if (rem >= a) ++q;
            const int bit = q - ((m24(2 * i, s) + S) >> (log2s + 1));
            atomicOr(&marks[i], 1u << bit);
          }
        }
      }
    } else {
      stamp();   // 8: (no compaction for this group: the stamp slots stay aligned)
    }
#ifdef GV_DIAG
    if (A.dbg) __syncthreads();
#endif
    stamp();   // 9: tail marched (barrier only in the diagnostic build)
    fstamp(4);
    // gather: one lane per column of the wedge
    const bool flat_tail = !march_tail && T < oc.imax && !GV_ABL(512);   // columns beyond T, every cell exactly
    const int gather_hi = (march_tail || flat_tail) ? T : oc.imax;
    for (int i = tid * nparts + part; i <= (GV_ABL(2) ? -1 : gather_hi); i += NT * nparts) {
      if (i == 0) {
        marks[0] |= 1u;   // every ray (reach >= 1) starts in the origin cell
        continue;
      }
      if ((unsigned)i >= maxreach) continue;   // no ray of this group gets this far
      const int jlo = (m24(2 * i, s) + S) >> (log2s + 1);
      const int jhi = (m24(2 * i, s + 1) + S) >> (log2s + 1);
      const int w = jhi - jlo + 1;
      const int Q = 2 * i;
      // a cell interval (width 1/i in slope) fully contains an aligned group of level Lv
      // buckets when 2^Lv >= 2i/S; if every such group holds a ray longer than i, every
      // cell that lies inside the sector's slope range (k = 1..w-2) is traversed.
      // columns up to T have every interior cell free (monotone test, computed once above)
      if (w <= 2) continue;                          // narrow column: done beside the threshold computation above
      bool interior_free = i <= T;
      if GV_ABL(64) interior_free = false;           // timing experiment: always the full loop
      if (GV_ABL(128) && !interior_free) continue;   // timing experiment: skip the full loop
      unsigned mask = 0, todo = 0;   // todo: cells of this column to evaluate exactly
      if (interior_free) {
        mask = ((w >= 32) ? 0xFFFFFFFFu : ((1u << w) - 1u)) & ~1u & ~(1u << (w - 1));
        // the two edge cells: all table reads of both issued before anything is evaluated
        // edge cell 0: slopes below the first interior boundary -> buckets [0, hi-1] + part of hi
        const int Phi = 2 * jlo + 1;
        const int hi = bucket_of_boundary(Phi, Q);
        // edge cell w-1: slopes at or above the last interior boundary -> part of lo + [lo+1, M-1]
        const int Plo = 2 * jhi - 1;
        const int lo = bucket_of_boundary(Plo, Q);
        const int r = min(max(hi, 1), M) - 1;              // last whole bucket below the boundary
        const int l = min(max(lo + 1, 0), M - 1);          // first whole bucket above it
        const int hic = min(max(hi, 0), M - 1), loc = min(max(lo, 0), M - 1);
        const unsigned pf = pfx[r], bpf = s_blkpfx[r >> 6];
        const unsigned sf = sfx[l], bsf = s_blksfx[l >> 6];
        const unsigned ch = cnt[hic], cl = cnt[loc];
        unsigned sh = 0, sl = 0;
        if constexpr (sorted_mode) { sh = bstart[hic]; sl = bstart[loc]; }
        const unsigned lvh = lvl[hic], lvlo = lvl[loc];   // read with the rest: one LDS round trip less on the walk path
        unsigned mxh = (hi >= 1) ? max(pf, bpf) : 0u;
        unsigned mxl = (lo + 1 <= M - 1) ? max(sf, bsf) : 0u;
        if (!GV_ABL(1024) && mxh <= (unsigned)i && hi >= 0 && hi < M && lvh > (unsigned)i)
          mxh = max(mxh, walk_bucket(mode, hic, sh, ch, Q, true, 0, false, Phi));
        if (!GV_ABL(1024) && mxl <= (unsigned)i && lo >= 0 && lo < M && lvlo > (unsigned)i)
          mxl = max(mxl, walk_bucket(mode, loc, sl, cl, Q, false, Plo, true, 0));
        if (mxh > (unsigned)i) mask |= 1u;
        if (mxl > (unsigned)i) mask |= 1u << (w - 1);
        // sectors with a lattice-gap run: the cell next to the edge cell is not covered by the level test
        if (xb0 < xb1) todo |= 2u;
        if (xt0 < xt1) todo |= 1u << (w - 2);
        mask &= ~todo;
      } else {
        // narrow columns (w <= 2: edge cells only), or -- without the flattened pass -- any column
        // beyond T: every cell exactly
        todo = (w >= 32) ? 0xFFFFFFFFu : ((1u << w) - 1u);
      }
      if GV_ABL(2048) todo = 0;
      while (todo) {
        const int k = __ffs(todo) - 1;
        todo &= todo - 1;
        if (cell_exact(mode, i, Q, jlo + k)) mask |= 1u << k;
      }
      if (mask) atomicOr(&marks[i], mask);   // no-return LDS OR: the lane does not wait for a read of the old word
    }
    fstamp(5);
    if (flat_tail) {
      // Too many long rays to march them: every cell of the columns T+1 .. (largest reach - 1) is
      // evaluated exactly, one (column, cell) pair per lane -- balanced over the 512 lanes, where a
      // lane per column would leave 60 % of them idle and walk ~16 cells in sequence.
      const int first0 = T + 1;
      const int first = first0 + ((nparts > 1 && (first0 & 1) != part) ? 1 : 0);   // this workgroup's first column
      const int last = min(oc.imax, (int)maxreach - 1);
      if (last >= first) {
        const int wlast = ((2 * last * (s + 1) + S) >> (log2s + 1)) - ((2 * last * s + S) >> (log2s + 1)) + 1;
        const int wfirst = ((2 * first * (s + 1) + S) >> (log2s + 1)) - ((2 * first * s + S) >> (log2s + 1)) + 1;
        const int wmax = max(wlast, wfirst) + 1;                 // w(i) grows with i, +-1 by rounding
        const int ncol = (last - first) / nparts + 1;
        const int ntask = m24(ncol, wmax);                       // (rounding wmax up to a power of two left up to half the lanes idle)
        const float inv_w = __builtin_amdgcn_rcpf((float)wmax);
        fnote(11, ((unsigned long long)ntask << 32) | ((unsigned)wmax << 16) | (unsigned)ncol);
        for (int t = tid; t < ntask; t += NT) {
          int c = (int)((float)t * inv_w);                       // t / wmax: estimate + correction (t < 2^24)
          int k = t - m24(c, wmax);
          if (k < 0) { --c; k += wmax; }
          else if (k >= wmax) { ++c; k -= wmax; }
          const int i = first + m24(nparts, c);
          const int jlo = (m24(2 * i, s) + S) >> (log2s + 1);
          const int jhi = (m24(2 * i, s + 1) + S) >> (log2s + 1);
          if (k > jhi - jlo) continue;
          if (cell_exact(mode, i, 2 * i, jlo + k)) atomicOr(&marks[i], 1u << k);
        }
      }
    }
    fstamp(6);
    __syncthreads();
    stamp();   // 10: gather done
  };

  if GV_ABL(32) return;   // timing experiment
  // Normally the whole wedge is one group.  With more ends than one LDS group holds: one group
  // per row of columns, and per wavefront (64 columns x <= 32 ends <= 2048 <= cap) where a row
  // alone is too big.
  // Two instantiations of the group code: a wedge whose ends fit the bucket rows (at most kSlotTotal: the usual case)
  // is one group in row mode; a crowded wedge runs the counting-sort mode, one group or several (one call site).
  if (total <= kSlotTotal && total <= cap) {
    process_group(std::false_type{}, (1u << CH) - 1u, -1, true);
  } else {
    const bool single = total <= cap;
    if (!single) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const unsigned r = wave_sum((unsigned)__popc(ends[c]));
        if (lane == 0) s_rowpart[c][wave] = r;
      }
      __syncthreads();
    }
    int gc = 0, gw = 0, gnw = 0;   // multi-group enumeration: row gc, wavefront gw of gnw
    auto next_group = [&]() -> bool {
      while (gc < CH) {
        if (gnw == 0) {
          int rt = 0;
          for (int wv = 0; wv < NT / 64; ++wv) rt += (int)s_rowpart[gc][wv];
          if (rt == 0) { ++gc; continue; }
          gnw = (rt <= cap) ? 1 : NT / 64;
          gw = 0;
        }
        if (gw < gnw) return true;
        ++gc;
        gnw = 0;
      }
      return false;
    };
    bool more = single ? true : next_group();
    while (more) {
      unsigned rowmask = (1u << CH) - 1u;
      int wsel = -1;
      if (!single) {
        scan_row(gc);
        rowmask = 1u << gc;
        wsel = (gnw == 1) ? -1 : gw;
        ++gw;
      }
      process_group(std::true_type{}, rowmask, wsel, single);
      more = single ? false : next_group();
    }
  }

  // ---- flush the wedge's marks into the free-cell bitmaps.  marks[i] is a run of cell bits along the
  // minor axis of column i, i.e. a shifted (for the negative minor direction: mirrored) slice of at
  // most two 32-bit words of the bitmap whose bits run along that axis: T (bits along y) for x-major
  // octants, N (bits along x) for y-major ones.  Consecutive columns are consecutive words of the
  // transposed word layout, so a wavefront ORs 256 contiguous bytes.  Neighbouring sectors share words:
  // atomicOr (no return value).
  {
    unsigned *bm = oc.xmaj ? A.freeT : A.freeN;
    const unsigned pad = (unsigned)(oc.xmaj ? A.nx_pad : A.ny_pad);
    for (int i = tid; i <= (GV_ABL(4) ? -1 : oc.imax); i += NT) {
      unsigned w = marks[i];
      const int jlo = (m24(2 * i, s) + S) >> (log2s + 1);
      const int tmax = oc.jmaxo - jlo;                        // bits beyond it are outside the map
      if (tmax < 31) w &= (tmax < 0) ? 0u : ((2u << tmax) - 1u);
      if (!w) continue;
      const int mb = oc_minor + m24(oc.smin, jlo);                // minor coordinate of bit 0 of w
      const int b0 = (oc.smin > 0) ? mb : mb - 31;            // minor coordinate of bit 0 of v
      const unsigned v = (oc.smin > 0) ? w : __brev(w);
      const int wi = b0 >> 5, sh = b0 & 31;                   // arithmetic shift: floor for b0 < 0
      const unsigned lo = v << sh, hi = sh ? (v >> (32 - sh)) : 0u;
      const unsigned col = (unsigned)(oc_major + m24(oc.smaj, i));
      if (lo) atomicOr(&bm[um24((unsigned)wi, pad) + col], lo);    // set bits are in-map cells: wi >= 0 whenever lo != 0
      if (hi) atomicOr(&bm[um24((unsigned)(wi + 1), pad) + col], hi);
    }
  }
  __syncthreads();
  stamp();   // 11: flush done
  GV_TL_END(A.tl);
  if (tid == 0) {
    // per-workgroup slots, summed by the host on demand: a shared counter would
    // serialise 2 x 8*S atomics on one address (~12 ns each)
    if (A.stats) {
      unsigned long long vsum = 0;
      for (int wv = 0; wv < NT / 64; ++wv) vsum += s_wvis[wv];
      A.stats[2 * stat_slot] = helper ? 0ull : (unsigned long long)total;   // a helper's rays are its partner's
      A.stats[2 * stat_slot + 1] = helper ? 0ull : vsum;
    }
  }
}

size_t sector_lds_bytes(int cap, int marks_words, int log2m)
{
  return sector_lds_layout(cap, marks_words, log2m).total;
}

// `done` (optional): an event that completes with this kernel, carried by the kernel's own dispatch packet
// (hipExtLaunchKernelGGL) -- a hipEventRecord behind it would put a marker packet into the queue and ~7 us
// between this kernel and the next one of the stream.  Returns false when nothing was launched (the
// caller then records the event the ordinary way).
bool launch_ray_sectors(const SectorArgs &a, hipStream_t s, hipEvent_t done, hipEvent_t t0)
{
  if (!a.org.valid) return false;
  const size_t lds = sector_lds_bytes(a.cap, a.marks_words, a.log2m);
  const int imax = std::max(std::max(a.org.cx, a.g.nx - 1 - a.org.cx), std::max(a.org.cy, a.g.ny - 1 - a.org.cy));
  // every wedge column lives in a register slot of one thread: CH * 512 >= imax
  const int total = a.wg_base[8] + a.n_helpers;
  if (a.wg_first >= total) return false;
  const int grid = (total - a.wg_first + a.wg_stride - 1) / a.wg_stride;
  // more than 64 KB of dynamic LDS has to be announced per kernel (once per size increase)
  auto allow_lds = [&](const void *fn, size_t &granted) {
    if (lds > granted && lds > 64u * 1024u) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
      granted = lds;
    }
    return true;
  };
  static size_t granted[3] = {0, 0, 0};
  if (imax <= 4 * kSecThreads) {
    if (!allow_lds(reinterpret_cast<const void *>(&k_ray_sectors<4>), granted[0])) return false;
    hipExtLaunchKernelGGL(k_ray_sectors<4>, dim3(grid), dim3(kSecThreads), (uint32_t)lds, s, t0, done, 0, a);
  } else if (imax <= 8 * kSecThreads) {
    if (!allow_lds(reinterpret_cast<const void *>(&k_ray_sectors<8>), granted[1])) return false;
    hipExtLaunchKernelGGL(k_ray_sectors<8>, dim3(grid), dim3(kSecThreads), (uint32_t)lds, s, t0, done, 0, a);
  } else {
    if (!allow_lds(reinterpret_cast<const void *>(&k_ray_sectors<16>), granted[2])) return false;
    hipExtLaunchKernelGGL(k_ray_sectors<16>, dim3(grid), dim3(kSecThreads), (uint32_t)lds, s, t0, done, 0, a);
  }
  return true;
}

// ------------------------------------------------------ tile grid pass -----
__device__ __forceinline__ float sigmoid_ref_t(float l)
{
  const float e = (float)exp((double)(-l));   // fp64 exp rounded once: agrees with glibc's expf, which the oracle calls (an fp32 expf measured +2 % on the frame, round 3, and is not bit-equal)
  return 1.0f / (1.0f + e);
}

__device__ __forceinline__ unsigned pack_i8_t(float p)
{
  float v = (p - 0.0f) / (1.0f - 0.0f);
  if (isnan(v)) v = -1.0f;
  else {
    float c = v < 0.0f ? 0.0f : v;
    c = c > 1.0f ? 1.0f : c;
    v = 0.0f + c * 100.0f;
  }
  return (unsigned)(unsigned char)(signed char)v;
}

__device__ __forceinline__ float cell_update_t(float l, int k, bool counts, bool hit, bool miss)
{
  l = l + kLogOddsDecay;
  for (int r = 0; r < k; ++r) l = l + kRectIncrement;
  if (counts) {
    if (hit) l = l + kLogOddsOccupied;
    else if (miss) l = l + kLogOddsFree;
  }
  l = (l < kMinLogOdds) ? kMinLogOdds : l;
  l = (l > kMaxLogOdds) ? kMaxLogOdds : l;
  return l;
}

// One 64x64 tile per workgroup.  Requires nx % 4 == 0.  Per cell: 4 B log-odds in, 4 + 4 + 1 B out;
// hit and free-space flags come from the bitmaps (N: one word per 32 cells of a row; T: bits run
// along y, a thread's four cells are four consecutive words = one 16-byte load).  Nothing is cleared
// here: the binning tile pass rewrites / zeroes every bitmap word of the set it is about to use.
#ifndef GV_FIN_ROWS
#define GV_FIN_ROWS 64
#endif
template <bool COUNTS>
__global__ void __launch_bounds__(256) k_finalize_tiles(FinalizeTileArgs a)
{
  __shared__ Rect s_rects[64];
  __shared__ int s_nr;
  constexpr int kRows = GV_FIN_ROWS;   // rows of the grid per workgroup (64 cells wide): 16 per pass
  const int x0 = blockIdx.x * 64, y0 = a.y_begin + blockIdx.y * kRows;
  const int tid = threadIdx.x;
  GV_TL_BEGIN(a.tl);
  if (tid == 0) s_nr = 0;
  __syncthreads();
  // rectangles that touch this tile (object order must be kept only for equal cells:
  // all adds are the same constant, so the count is what matters)
  for (int r = tid; r < a.n_rects; r += 256) {
    const Rect R = a.rects[r];
    if (R.valid && R.y1 >= y0 && R.y0 <= y0 + kRows - 1 && R.x1 >= x0 && R.x0 <= x0 + 63) {
      const int k = atomicAdd(&s_nr, 1);
      if (k < 64) s_rects[k] = R;
    }
  }
  __syncthreads();
  const int nr = min(s_nr, 64);
  const bool overflow = s_nr > 64;
  const int xq = tid & 15;        // which float4 of the row
  const int x = x0 + xq * 4;
#pragma unroll
  for (int pass = 0; pass < kRows / 16; ++pass) {
    const int yl = pass * 16 + (tid >> 4);
    const int y = y0 + yl;
    if (x >= a.g.nx || y >= a.y_end) continue;
    const size_t c = (size_t)y * a.g.nx + x;
    float4 l4 = *reinterpret_cast<const float4 *>(a.log_odds + c);
    unsigned hb = 0, fb = 0;
    if (COUNTS) {
      const size_t wn = (size_t)(x >> 5) * a.ny_pad + y;
      hb = (a.hitN[wn] >> (x & 31)) & 0xFu;
      fb = (a.freeN[wn] >> (x & 31)) & 0xFu;
      const uint4 ft = *reinterpret_cast<const uint4 *>(a.freeT + (size_t)(y >> 5) * a.nx_pad + x);
      const int by = y & 31;
      fb |= ((ft.x >> by) & 1u) | (((ft.y >> by) & 1u) << 1) | (((ft.z >> by) & 1u) << 2) | (((ft.w >> by) & 1u) << 3);
    }
    int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    if (!overflow) {
      for (int r = 0; r < nr; ++r) {
        const Rect R = s_rects[r];
        if (y >= R.y0 && y <= R.y1) {
          k0 += (x >= R.x0 && x <= R.x1);
          k1 += (x + 1 >= R.x0 && x + 1 <= R.x1);
          k2 += (x + 2 >= R.x0 && x + 2 <= R.x1);
          k3 += (x + 3 >= R.x0 && x + 3 <= R.x1);
        }
      }
    } else {
      for (int r = 0; r < a.n_rects; ++r) {
        const Rect R = a.rects[r];
        if (R.valid && y >= R.y0 && y <= R.y1) {
          k0 += (x >= R.x0 && x <= R.x1);
          k1 += (x + 1 >= R.x0 && x + 1 <= R.x1);
          k2 += (x + 2 >= R.x0 && x + 2 <= R.x1);
          k3 += (x + 3 >= R.x0 && x + 3 <= R.x1);
        }
      }
    }
    l4.x = cell_update_t(l4.x, k0, COUNTS, hb & 1u, fb & 1u);
    l4.y = cell_update_t(l4.y, k1, COUNTS, hb & 2u, fb & 2u);
    l4.z = cell_update_t(l4.z, k2, COUNTS, hb & 4u, fb & 4u);
    l4.w = cell_update_t(l4.w, k3, COUNTS, hb & 8u, fb & 8u);
    float4 p4;
    p4.x = sigmoid_ref_t(l4.x); p4.y = sigmoid_ref_t(l4.y); p4.z = sigmoid_ref_t(l4.z); p4.w = sigmoid_ref_t(l4.w);
    *reinterpret_cast<float4 *>(a.log_odds + c) = l4;   // read again by the next frame
    // occupancy and the packed grid are outputs nobody reads on the device: streaming stores
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f np4; np4.x = p4.x; np4.y = p4.y; np4.z = p4.z; np4.w = p4.w;
    __builtin_nontemporal_store(np4, reinterpret_cast<v4f *>(a.occupancy + c));
    const unsigned packed = pack_i8_t(p4.w) | (pack_i8_t(p4.z) << 8) | (pack_i8_t(p4.y) << 16) | (pack_i8_t(p4.x) << 24);
    __builtin_nontemporal_store(packed, reinterpret_cast<unsigned *>(a.occ_i8 + ((size_t)a.g.G - 4 - c)));
  }
  GV_TL_END(a.tl);
}

// `done` (optional) completes with the kernel, on its own dispatch packet (see launch_ray_sectors);
// false: nothing launched
bool launch_finalize_tiles(const FinalizeTileArgs &a, hipStream_t s, hipEvent_t done, hipEvent_t t0)
{
  const int rows = a.y_end - a.y_begin;
  if (rows <= 0) return false;
  const dim3 grid((a.g.nx + 63) / 64, (rows + GV_FIN_ROWS - 1) / GV_FIN_ROWS);
  if (a.counts) hipExtLaunchKernelGGL(k_finalize_tiles<true>, grid, dim3(256), 0, s, t0, done, 0, a);
  else hipExtLaunchKernelGGL(k_finalize_tiles<false>, grid, dim3(256), 0, s, t0, done, 0, a);
  return true;
}

// miss read-back: free-cell bitmaps N | T as int32 0/1 per cell
__global__ void __launch_bounds__(256) k_miss_to_i32(const unsigned *__restrict__ fN, const unsigned *__restrict__ fT,
                                                     int nx, int ny, int nx_pad, int ny_pad,
                                                     int32_t *__restrict__ out)
{
  const size_t G = (size_t)nx * ny;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < G; c += stride) {
    const int y = (int)(c / nx), x = (int)(c - (size_t)y * nx);
    const unsigned n = fN[(size_t)(x >> 5) * ny_pad + y] >> (x & 31);
    const unsigned t = fT[(size_t)(y >> 5) * nx_pad + x] >> (y & 31);
    out[c] = (int32_t)((n | t) & 1u);
  }
}

void launch_miss_to_i32(const uint32_t *fN, const uint32_t *fT, int nx, int ny, int nx_pad, int ny_pad, int32_t *out,
                        hipStream_t s)
{
  hipLaunchKernelGGL(k_miss_to_i32, dim3(2048), dim3(256), 0, s, fN, fT, nx, ny, nx_pad, ny_pad, out);
}

}  // namespace gv
