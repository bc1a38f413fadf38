// gv_shard.hip -- [EXTENSION] SURVEY 8(e)-2 device-side pieces of the frame that is sharded by
// POINTS over `world` GPUs.  RCCL has no bitwise-OR reduction, so the two exchanges of the frame (ray
// end bitmaps before the ray stage, free-cell bitmaps after it) are an all-to-all of equal slices
// followed by a local OR of the `world` slices received -- these kernels are that OR, plus the
// packing of the free-cell bitmaps by row band.  gfx950, wave64.
#include "gv_kernels.hpp"

#include <algorithm>

namespace gv {

// dst[i] = OR over r of src[r * count + i]      (count words, 16-byte vectors)
__global__ void __launch_bounds__(256) k_or_slices(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                   size_t count4, int world)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count4; i += stride) {
    uint4 v = src[i];
    for (int r = 1; r < world; ++r) {
      const uint4 p = src[(size_t)r * count4 + i];
      v.x |= p.x; v.y |= p.y; v.z |= p.z; v.w |= p.w;
    }
    dst[i] = v;
  }
}

void launch_or_slices(const uint32_t *src, uint32_t *dst, size_t count_words, int world, hipStream_t s)
{
  const size_t c4 = count_words / 4;   // callers pad slices to multiples of 4 words
  if (!c4) return;
  const uint32_t blocks = (uint32_t)std::min<size_t>((c4 + 255) / 256, (size_t)2048);
  hipLaunchKernelGGL(k_or_slices, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4 *>(src),
                     reinterpret_cast<uint4 *>(dst), c4, world);
}

// Row band q of the free-cell bitmaps, packed for peer q.  Bands are whole 64-row blocks of the padded
// grid: rows [64*blk(q), 64*blk(q+1)), blk(q) = q * (ny_pad/64) / world.  Chunk q of `out` (chunk words
// each) holds   [wc][y - y0] words of freeN   then   [wr - y0/32][x] words of freeT.
__device__ __forceinline__ int band_block(int q, int nblk, int world) { return (int)((long long)q * nblk / world); }

__global__ void __launch_bounds__(256) k_pack_free_bands(const uint32_t *__restrict__ fN, const uint32_t *__restrict__ fT,
                                                         int nxw, int nx_pad, int ny_pad, int world, size_t chunk,
                                                         uint32_t *__restrict__ out)
{
  const int q = blockIdx.y;
  const int nblk = ny_pad / 64;
  const int y0 = 64 * band_block(q, nblk, world), y1 = 64 * band_block(q + 1, nblk, world);
  const int rows = y1 - y0;
  const size_t nN = (size_t)nxw * rows, nT = (size_t)(rows / 32) * nx_pad;
  uint32_t *dst = out + (size_t)q * chunk;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < chunk; i += stride) {
    uint32_t v = 0;
    if (i < nN) {
      const int wc = (int)(i / rows), yy = (int)(i - (size_t)wc * rows);
      v = fN[(size_t)wc * ny_pad + y0 + yy];
    } else if (i < nN + nT) {
      v = fT[(size_t)(y0 / 32) * nx_pad + (i - nN)];
    }
    dst[i] = v;
  }
}

void launch_pack_free_bands(const uint32_t *fN, const uint32_t *fT, int nxw, int nx_pad, int ny_pad, int world,
                            size_t chunk, uint32_t *out, hipStream_t s)
{
  const uint32_t bx = (uint32_t)std::min<size_t>((chunk + 255) / 256, (size_t)512);
  hipLaunchKernelGGL(k_pack_free_bands, dim3(bx, world), dim3(256), 0, s, fN, fT, nxw, nx_pad, ny_pad, world, chunk, out);
}

// OR of the `world` chunks received for MY band, written back into the bitmap layout at the band's rows
__global__ void __launch_bounds__(256) k_unpack_free_band(const uint32_t *__restrict__ in, int world, size_t chunk,
                                                          int rank, int nxw, int nx_pad, int ny_pad,
                                                          uint32_t *__restrict__ fN, uint32_t *__restrict__ fT)
{
  const int nblk = ny_pad / 64;
  const int y0 = 64 * band_block(rank, nblk, world), y1 = 64 * band_block(rank + 1, nblk, world);
  const int rows = y1 - y0;
  const size_t nN = (size_t)nxw * rows, nT = (size_t)(rows / 32) * nx_pad;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nN + nT; i += stride) {
    uint32_t v = 0;
    for (int r = 0; r < world; ++r) v |= in[(size_t)r * chunk + i];
    if (i < nN) {
      const int wc = (int)(i / rows), yy = (int)(i - (size_t)wc * rows);
      fN[(size_t)wc * ny_pad + y0 + yy] = v;
    } else {
      fT[(size_t)(y0 / 32) * nx_pad + (i - nN)] = v;
    }
  }
}

void launch_unpack_free_band(const uint32_t *in, int world, size_t chunk, int rank, int nxw, int nx_pad, int ny_pad,
                             uint32_t *fN, uint32_t *fT, hipStream_t s)
{
  const uint32_t bx = (uint32_t)std::min<size_t>((chunk + 255) / 256, (size_t)1024);
  hipLaunchKernelGGL(k_unpack_free_band, dim3(bx), dim3(256), 0, s, in, world, chunk, rank, nxw, nx_pad, ny_pad, fN, fT);
}

size_t free_band_chunk_words(int nxw, int nx_pad, int ny_pad, int world)
{
  const int nblk = ny_pad / 64;
  int rows_max = 0;
  for (int q = 0; q < world; ++q) {
    const int r = 64 * ((int)((long long)(q + 1) * nblk / world) - (int)((long long)q * nblk / world));
    rows_max = std::max(rows_max, r);
  }
  const size_t c = (size_t)nxw * rows_max + (size_t)(rows_max / 32) * nx_pad;
  return (c + 3) & ~(size_t)3;
}

void shard_band_rows(int rank, int world, int ny, int ny_pad, int32_t &y0, int32_t &y1)
{
  const int nblk = ny_pad / 64;
  y0 = std::min(ny, 64 * (int)((long long)rank * nblk / world));
  y1 = std::min(ny, 64 * (int)((long long)(rank + 1) * nblk / world));
}

}  // namespace gv
