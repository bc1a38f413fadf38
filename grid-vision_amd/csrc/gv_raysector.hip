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
// That is a range-max query over the ends sorted by slope.  Each workgroup owns
// one angular sector of one octant: it collects its ends from end bitmaps,
// sorts them by slope in LDS, builds a block-decomposed range-max structure,
// and answers every cell of its wedge with two binary searches.  Everything is
// integer arithmetic: bit-exact against the oracle's literal march.
#include "gv_kernels.hpp"

#include <algorithm>

namespace gv {

// ------------------------------------------------------------ bitmaps ------
// hit/clip end flags as bitmaps in both orientations, with the 32-bit WORDS stored
// transposed so that a wavefront scanning 64 consecutive wedge columns reads 64
// adjacent words:
//   N: bits run along x, word(x>>5, y) at  (x>>5)*ny_pad + y   (y-major octants)
//   T: bits run along y, word(y>>5, x) at  (y>>5)*nx_pad + x   (x-major octants)
// One 64x64 tile per workgroup; also clears the per-frame clip bytes (and the
// hit counts unless the caller keeps them).
__global__ void __launch_bounds__(256) k_build_bitmaps(BitmapArgs a)
{
  __shared__ unsigned long long rows[2][64];   // [0] hit, [1] clip
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = x0 + lane;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int yl = wv * 16 + r;
    const int y = y0 + yl;
    const bool valid = (x < a.nx) && (y < a.ny);
    int h = 0;
    unsigned c = 0;
    if (valid) {
      const size_t cell = (size_t)y * a.nx + x;
      h = a.hits[cell];
      c = a.clip_end[cell];
      if (c) a.clip_end[cell] = 0;
      if (a.zero_hits && h) a.hits[cell] = 0;
    }
    const unsigned long long mh = __ballot(h > 0);
    const unsigned long long mc = __ballot(c != 0);
    if (lane == 0) {
      rows[0][yl] = mh;
      rows[1][yl] = mc;
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  {
    // N words: thread (which, half, yl)
    const int which = t >> 7, half = (t >> 6) & 1, yl = t & 63;
    unsigned *dst = which ? a.clipN : a.hitN;
    dst[(size_t)(2 * blockIdx.x + half) * a.ny_pad + (y0 + yl)] = (unsigned)(rows[which][yl] >> (32 * half));
  }
  {
    // T words: thread (which, half, xl) gathers bit xl of rows 32*half .. 32*half+31
    const int which = t >> 7, half = (t >> 6) & 1, xl = t & 63;
    unsigned w = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) w |= (unsigned)((rows[which][half * 32 + r] >> xl) & 1ull) << r;
    unsigned *dst = which ? a.clipT : a.hitT;
    dst[(size_t)(2 * blockIdx.y + half) * a.nx_pad + (x0 + xl)] = w;
  }
}

void launch_build_bitmaps(const BitmapArgs &a, hipStream_t s)
{
  hipLaunchKernelGGL(k_build_bitmaps, dim3((a.nx + 63) / 64, (a.ny + 63) / 64), dim3(256), 0, s, a);
}

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

__global__ void __launch_bounds__(256) k_ray_sectors(SectorArgs A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int S = 1 << A.log2s;
  const int o = blockIdx.x >> A.log2s;
  const int s = blockIdx.x & (S - 1);
  const Oct oc = make_octant(o, A.g, A.org);
  if (oc.imax < 1) {
    if (threadIdx.x == 0 && A.stats) { A.stats[2 * blockIdx.x] = 0; A.stats[2 * blockIdx.x + 1] = 0; }
    return;
  }
  const int cap = A.cap, nblk = cap >> 4;
  int nlev = 0;
  while ((1 << nlev) < nblk) ++nlev;
  ++nlev;   // levels 0..log2(nblk)

  // LDS carve-up (10 bytes per end + marks): packed ends, then reach / prefix / suffix maxima
  unsigned *abv = reinterpret_cast<unsigned *>(smem);                                   // cap
  unsigned *marks = abv + cap;                                                          // marks_words
  unsigned short *reach = reinterpret_cast<unsigned short *>(marks + A.marks_words);    // cap
  unsigned short *pm = reach + cap;                                                     // cap
  unsigned short *sm = pm + cap;                                                        // cap
  unsigned short *bmx = sm + cap;                                                       // nlev * nblk
  __shared__ unsigned s_count, s_batch;
  __shared__ unsigned long long s_rays, s_visits;

  for (int i = tid; i <= oc.imax; i += 256) marks[i] = 0;
  if (tid == 0) { s_count = 0; s_rays = 0; s_visits = 0; }
  __syncthreads();

  const unsigned *bmH = oc.xmaj ? A.hitT : A.hitN;
  const unsigned *bmC = oc.xmaj ? A.clipT : A.clipN;
  const int roww = oc.xmaj ? A.nyw : A.nxw;                 // words along the minor axis
  const size_t major_pad = oc.xmaj ? A.nx_pad : A.ny_pad;   // word-row stride
  const int oc_major = oc.xmaj ? A.org.cx : A.org.cy;
  const int oc_minor = oc.xmaj ? A.org.cy : A.org.cx;

  // process the collected chunk [0, n): sort by slope, build range-max, gather
  auto process_chunk = [&](int n) {
    int npad = 64;
    while (npad < n) npad <<= 1;
    constexpr unsigned kSent = 0xFFFFFFFFu;   // sorts after every real end
    for (int k = n + tid; k < npad; k += 256) abv[k] = kSent;
    __syncthreads();
    // bitonic sort by slope b/a, ascending; exact order by cross-multiplication
    if (!(A.ablate & 1)) {
      // thread t owns compare-exchange pairs t, t+256, ...; a wavefront's 64 consecutive
      // pairs span 128 consecutive elements, so strides j <= 64 never cross wavefronts:
      // those steps need only wave-level ordering of the LDS traffic, no block barrier.
      for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = tid; t < (npad >> 1); t += 256) {
            const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int hi = lo | j;
            const bool up = ((lo & k) == 0);
            const unsigned x = abv[lo], y = abv[hi];
            bool sw;
            if (x == kSent || y == kSent) {
              // sentinel sorts last
              sw = up ? (x == kSent && y != kSent) : (y == kSent && x != kSent);
            } else {
              const unsigned l = (unsigned)ab_b(x) * (unsigned)ab_a(y), r = (unsigned)ab_b(y) * (unsigned)ab_a(x);
              sw = up ? (l > r) : (l < r);
            }
            if (sw) { abv[lo] = y; abv[hi] = x; }
          }
          if (j > 64) __syncthreads();
          else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
        __syncthreads();
      }
    }
    // payloads + per-block (16) prefix / suffix maxima + sparse table over block maxima
    unsigned long long vis = 0;
    for (int b = tid; b < nblk; b += 256) {
      unsigned short run = 0;
      unsigned short r16[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = b * 16 + e;
        unsigned short r = 0;
        if (k < n) {
          const unsigned p = abv[k];
          r = (unsigned short)(ab_a(p) + (int)(p & 1u));
          vis += r;
        }
        r16[e] = r;
        run = r > run ? r : run;
        if (k < cap) { reach[k] = r; pm[k] = run; }
      }
      bmx[b] = run;
      run = 0;
#pragma unroll
      for (int e = 15; e >= 0; --e) {
        run = r16[e] > run ? r16[e] : run;
        sm[b * 16 + e] = run;
      }
    }
    if (vis) atomicAdd(&s_visits, vis);
    __syncthreads();
    for (int l = 1; l < nlev; ++l) {
      const int half = 1 << (l - 1);
      for (int b = tid; b < nblk; b += 256) {
        const unsigned short x = bmx[(l - 1) * nblk + b];
        const unsigned short y = (b + half < nblk) ? bmx[(l - 1) * nblk + b + half] : (unsigned short)0;
        bmx[l * nblk + b] = x > y ? x : y;
      }
      __syncthreads();
    }
    // gather: one lane per column of the wedge
    for (int i = tid; i <= ((A.ablate & 2) ? -1 : oc.imax); i += 256) {
      if (i == 0) {
        marks[0] |= 1u;   // every ray (reach >= 1) starts in the origin cell
        continue;
      }
      const int jlo = (2 * i * s + S) >> (A.log2s + 1);
      const int jhi = (2 * i * (s + 1) + S) >> (A.log2s + 1);
      const unsigned Q = 2u * (unsigned)i;
      unsigned mask = 0;
      int prev = 0;
      for (int j = jlo; j <= jhi; ++j) {
        int nxt = n;
        if (j < jhi) {
          // first k with b_k/a_k >= (2j+1)/(2i):  b_k*Q >= P*a_k
          const unsigned P = 2u * (unsigned)j + 1u;
          int lo = prev, hi = n;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned p = abv[mid];
            if ((unsigned)ab_b(p) * Q >= P * (unsigned)ab_a(p)) hi = mid;
            else lo = mid + 1;
          }
          nxt = lo;
        }
        if (nxt > prev) {
          // max reach over [prev, nxt)
          const int bl = prev >> 4, bh = (nxt - 1) >> 4;
          unsigned m;
          if (bl == bh) {
            m = 0;
            for (int k = prev; k < nxt; ++k) m = max(m, (unsigned)reach[k]);
          } else {
            m = max((unsigned)sm[prev], (unsigned)pm[nxt - 1]);
            const int len = bh - bl - 1;
            if (len > 0 && m <= (unsigned)i) {
              const int l = 31 - __clz(len);
              m = max(m, (unsigned)bmx[l * nblk + bl + 1]);
              m = max(m, (unsigned)bmx[l * nblk + bh - (1 << l)]);
            }
          }
          if (m > (unsigned)i) mask |= 1u << (j - jlo);
        }
        prev = nxt;
        if (prev >= n) break;
      }
      if (mask) marks[i] |= mask;
    }
    __syncthreads();
  };

  // ---- scan the wedge column by column, collecting ends
  // ends of column a as a bit mask over b = blo..bhi (bit t <-> b = blo+t, or bhi-t on the
  // negative minor side); vc marks the clipped (inclusive) ones
  auto scan_col = [&](int a, unsigned &ends, unsigned &vc, int &blo, int &bhi) {
    ends = 0;
    vc = 0;
    const int bmaxa = oc.xmaj ? a : a - 1;
    blo = (a * s + S - 1) >> A.log2s;
    bhi = ((a * (s + 1) + S - 1) >> A.log2s) - 1;
    if (s == S - 1) bhi = bmaxa;
    blo = max(blo, oc.bmin);
    bhi = min(min(bhi, bmaxa), oc.jmaxo);
    if (blo > bhi) return;
    const int w = bhi - blo + 1;   // <= 32 (host guarantees imax <= 30*S)
    const int major_abs = oc_major + oc.smaj * a;
    const int m_lo = (oc.smin > 0) ? oc_minor + blo : oc_minor - bhi;
    const int w0 = m_lo >> 5, sh = m_lo & 31;
    const size_t base = (size_t)w0 * major_pad + major_abs;
    unsigned long long h64 = bmH[base], c64 = bmC[base];
    if (sh + w > 32 && w0 + 1 < roww) {
      h64 |= (unsigned long long)bmH[base + major_pad] << 32;
      c64 |= (unsigned long long)bmC[base + major_pad] << 32;
    }
    const unsigned wm = (w >= 32) ? 0xFFFFFFFFu : ((1u << w) - 1u);
    const unsigned vh = (unsigned)(h64 >> sh) & wm;
    vc = (unsigned)(c64 >> sh) & wm;
    ends = vh | vc;
  };
  auto append_col = [&](int a, unsigned ends, unsigned vc, int blo, int bhi, int slot) {
    while (ends) {
      const int t = __ffs(ends) - 1;
      ends &= ends - 1;
      const int b = (oc.smin > 0) ? blo + t : bhi - t;
      abv[slot++] = pack_ab(a, b, (int)((vc >> t) & 1u));
    }
  };

  // pass A: count this thread's ends over all its columns (independent loads, no barriers)
  int mine_total = 0;
  for (int a = 1 + tid; a <= oc.imax; a += 256) {
    unsigned ends, vc;
    int blo, bhi;
    scan_col(a, ends, vc, blo, bhi);
    mine_total += __popc(ends);
  }
  if (tid == 0) s_batch = 0;
  __syncthreads();
  if (mine_total) atomicAdd(&s_batch, (unsigned)mine_total);
  __syncthreads();
  const int total = (int)s_batch;
  __syncthreads();
  if (total == 0) {
    // nothing ends in this wedge
  } else if (total <= cap) {
    // common case: one chunk, one slot allocation per thread
    int slot = mine_total ? (int)atomicAdd(&s_count, (unsigned)mine_total) : 0;
    for (int a = 1 + tid; a <= oc.imax && mine_total; a += 256) {
      unsigned ends, vc;
      int blo, bhi;
      scan_col(a, ends, vc, blo, bhi);
      const int c = __popc(ends);
      append_col(a, ends, vc, blo, bhi, slot);
      slot += c;
    }
    __syncthreads();
    process_chunk(total);
    if (tid == 0) { s_rays += (unsigned long long)total; s_count = 0; }
    __syncthreads();
  } else {
    // rare: more ends than one LDS chunk holds -> batches of 256 columns, flushing chunks
    for (int a0 = 1; a0 <= oc.imax; a0 += 256) {
      const int a = a0 + tid;
      unsigned ends = 0, vc = 0;
      int blo = 0, bhi = -1;
      if (a <= oc.imax) scan_col(a, ends, vc, blo, bhi);
      const int cnt = __popc(ends);
      if (tid == 0) s_batch = 0;
      __syncthreads();
      if (cnt) atomicAdd(&s_batch, (unsigned)cnt);
      __syncthreads();
      const int batch = (int)s_batch;
      __syncthreads();   // everyone has read s_batch before it is reset again
      if (batch == 0) continue;
      // sub-batches of 64 columns bound one append to 64*32 = 2048 <= cap
      const int nsub = (batch > cap) ? 4 : 1;
      for (int sb = 0; sb < nsub; ++sb) {
        const bool mine = (nsub == 1) || ((tid >> 6) == sb);
        if (nsub > 1) {
          if (tid == 0) s_batch = 0;
          __syncthreads();
          if (mine && cnt) atomicAdd(&s_batch, (unsigned)cnt);
          __syncthreads();
        }
        const int need = (nsub > 1) ? (int)s_batch : batch;
        if ((int)s_count + need > cap) {
          const int n = (int)s_count;
          __syncthreads();
          process_chunk(n);
          if (tid == 0) { s_rays += (unsigned long long)n; s_count = 0; }
          __syncthreads();
        }
        if (mine && cnt) {
          const int slot = (int)atomicAdd(&s_count, (unsigned)cnt);
          append_col(a, ends, vc, blo, bhi, slot);
        }
        __syncthreads();
      }
    }
    const int n = (int)s_count;
    __syncthreads();
    if (n > 0) {
      process_chunk(n);
      if (tid == 0) s_rays += (unsigned long long)n;
      __syncthreads();
    }
  }

  // ---- flush the wedge's marks as bytes: N grid for x-major, T grid for y-major
  for (int i = tid; i <= ((A.ablate & 4) ? -1 : oc.imax); i += 256) {
    unsigned w = marks[i];
    if (!w) continue;
    const int jlo = (2 * i * s + S) >> (A.log2s + 1);
    const int major_abs = oc_major + oc.smaj * i;
    while (w) {
      const int t = __ffs(w) - 1;
      w &= w - 1;
      const int j = jlo + t;
      if (j > oc.jmaxo) continue;
      const int minor_abs = oc_minor + oc.smin * j;
      if (oc.xmaj) A.missN[(size_t)minor_abs * A.g.nx + major_abs] = 1;
      else A.missT[(size_t)minor_abs * A.g.ny + major_abs] = 1;
    }
  }
  if (tid == 0) {
    // per-workgroup slots, summed by the host on demand: a shared counter would
    // serialise 2 x 8*S atomics on one address (~12 ns each)
    if (A.stats) {
      A.stats[2 * blockIdx.x] = s_rays;
      A.stats[2 * blockIdx.x + 1] = s_visits;
    }
    // a clipped ray that ends in the origin cell itself (a == 0, inclusive)
    if (blockIdx.x == 0) {
      const unsigned wbit = A.clipN[(size_t)(A.org.cx >> 5) * A.ny_pad + A.org.cy] >> (A.org.cx & 31);
      if (wbit & 1u) A.missN[(size_t)A.org.cy * A.g.nx + A.org.cx] = 1;
    }
  }
}

size_t sector_lds_bytes(int cap, int marks_words)
{
  const int nblk = cap >> 4;
  int nlev = 0;
  while ((1 << nlev) < nblk) ++nlev;
  ++nlev;
  return (size_t)cap * 4 + (size_t)marks_words * 4 + (size_t)cap * 2 * 3 + (size_t)nlev * nblk * 2;
}

void launch_ray_sectors(const SectorArgs &a, hipStream_t s)
{
  if (!a.org.valid) return;
  const size_t lds = sector_lds_bytes(a.cap, a.marks_words);
  hipLaunchKernelGGL(k_ray_sectors, dim3(8u << a.log2s), dim3(256), lds, s, a);
}

// ------------------------------------------------------ tile grid pass -----
__device__ __forceinline__ float sigmoid_ref_t(float l)
{
  const float e = (float)exp((double)(-l));
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

// One 64x64 tile per workgroup.  Requires nx % 4 == 0.  Reads the hit bitmap (N),
// the miss bytes (N) and the transposed miss bytes (T, through an LDS transpose);
// clears both miss grids for the next frame.
template <bool COUNTS>
__global__ void __launch_bounds__(256) k_finalize_tiles(FinalizeTileArgs a)
{
  __shared__ unsigned char tileT[64][68];
  __shared__ Rect s_rects[64];
  __shared__ int s_nr;
  const int x0 = blockIdx.x * 64, y0 = a.y_begin + blockIdx.y * 64;
  const int tid = threadIdx.x;
  if (tid == 0) s_nr = 0;
  __syncthreads();
  // rectangles that touch this tile (object order must be kept only for equal cells:
  // all adds are the same constant, so the count is what matters)
  for (int r = tid; r < a.n_rects; r += 256) {
    const Rect R = a.rects[r];
    if (R.valid && R.y1 >= y0 && R.y0 <= y0 + 63 && R.x1 >= x0 && R.x0 <= x0 + 63) {
      const int k = atomicAdd(&s_nr, 1);
      if (k < 64) s_rects[k] = R;
    }
  }
  if (COUNTS) {
    // missT rows are x; each holds 64 contiguous y bytes of this tile
    for (int t = tid; t < 64 * 16; t += 256) {
      const int xr = t >> 4, wq = t & 15;
      const int x = x0 + xr, y = y0 + wq * 4;
      unsigned v = 0;
      if (x < a.g.nx && y < a.g.ny) {   // ny % 4 == 0 is not required: guard per byte below
        const size_t off = (size_t)x * a.g.ny + y;
        if (y + 3 < a.g.ny && (off & 3) == 0) {
          v = *reinterpret_cast<const unsigned *>(a.missT + off);
          if (v && a.zero) *reinterpret_cast<unsigned *>(a.missT + off) = 0u;
        } else {
          for (int k = 0; k < 4 && y + k < a.g.ny; ++k) {
            const unsigned b = a.missT[off + k];
            v |= b << (8 * k);
            if (b && a.zero) a.missT[off + k] = 0;
          }
        }
      }
      tileT[xr][wq * 4 + 0] = (unsigned char)(v & 0xff);
      tileT[xr][wq * 4 + 1] = (unsigned char)((v >> 8) & 0xff);
      tileT[xr][wq * 4 + 2] = (unsigned char)((v >> 16) & 0xff);
      tileT[xr][wq * 4 + 3] = (unsigned char)(v >> 24);
    }
  }
  __syncthreads();
  const int nr = min(s_nr, 64);
  const bool overflow = s_nr > 64;
  const int xq = tid & 15;        // which float4 of the row
  const int x = x0 + xq * 4;
#pragma unroll 2
  for (int pass = 0; pass < 4; ++pass) {
    const int yl = pass * 16 + (tid >> 4);
    const int y = y0 + yl;
    if (x >= a.g.nx || y >= a.y_end) continue;
    const size_t c = (size_t)y * a.g.nx + x;
    float4 l4 = *reinterpret_cast<const float4 *>(a.log_odds + c);
    unsigned hb = 0, mN = 0;
    if (COUNTS) {
      hb = (a.hitN[(size_t)(x >> 5) * a.ny_pad + y] >> (x & 31)) & 0xFu;
      mN = *reinterpret_cast<const unsigned *>(a.missN + c);
      if (mN && a.zero) *reinterpret_cast<unsigned *>(a.missN + c) = 0u;
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
    const int xr = xq * 4;
    const bool m0 = COUNTS && ((mN & 0xffu) | tileT[xr + 0][yl]);
    const bool m1 = COUNTS && (((mN >> 8) & 0xffu) | tileT[xr + 1][yl]);
    const bool m2 = COUNTS && (((mN >> 16) & 0xffu) | tileT[xr + 2][yl]);
    const bool m3 = COUNTS && ((mN >> 24) | tileT[xr + 3][yl]);
    l4.x = cell_update_t(l4.x, k0, COUNTS, hb & 1u, m0);
    l4.y = cell_update_t(l4.y, k1, COUNTS, hb & 2u, m1);
    l4.z = cell_update_t(l4.z, k2, COUNTS, hb & 4u, m2);
    l4.w = cell_update_t(l4.w, k3, COUNTS, hb & 8u, m3);
    float4 p4;
    p4.x = sigmoid_ref_t(l4.x); p4.y = sigmoid_ref_t(l4.y); p4.z = sigmoid_ref_t(l4.z); p4.w = sigmoid_ref_t(l4.w);
    *reinterpret_cast<float4 *>(a.log_odds + c) = l4;
    *reinterpret_cast<float4 *>(a.occupancy + c) = p4;
    const unsigned packed = pack_i8_t(p4.w) | (pack_i8_t(p4.z) << 8) | (pack_i8_t(p4.y) << 16) | (pack_i8_t(p4.x) << 24);
    *reinterpret_cast<unsigned *>(a.occ_i8 + ((size_t)a.g.G - 4 - c)) = packed;
  }
}

void launch_finalize_tiles(const FinalizeTileArgs &a, hipStream_t s)
{
  const int rows = a.y_end - a.y_begin;
  if (rows <= 0) return;
  const dim3 grid((a.g.nx + 63) / 64, (rows + 63) / 64);
  if (a.counts) hipLaunchKernelGGL(k_finalize_tiles<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_finalize_tiles<false>, grid, dim3(256), 0, s, a);
}

// miss read-back: N | T^T as int32 0/1
__global__ void __launch_bounds__(256) k_miss_to_i32(const unsigned char *__restrict__ mN,
                                                     const unsigned char *__restrict__ mT, int nx, int ny,
                                                     int32_t *__restrict__ out)
{
  const size_t G = (size_t)nx * ny;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < G; c += stride) {
    const int y = (int)(c / nx), x = (int)(c - (size_t)y * nx);
    out[c] = (mN[c] | mT[(size_t)x * ny + y]) ? 1 : 0;
  }
}

void launch_miss_to_i32(const uint8_t *mN, const uint8_t *mT, int nx, int ny, int32_t *out, hipStream_t s)
{
  hipLaunchKernelGGL(k_miss_to_i32, dim3(2048), dim3(256), 0, s, mN, mT, nx, ny, out);
}

}  // namespace gv
