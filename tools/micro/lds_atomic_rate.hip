// Microbenchmark: LDS atomic-add throughput of one CU (1024 threads, 64 KB int32 histogram, random cells),
// against plain LDS stores.  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 256;
template <int MODE>
__global__ void __launch_bounds__(1024) k(const unsigned *keys, unsigned long long *out, unsigned *sink)
{
  __shared__ unsigned hist[16384];
  const int tid = threadIdx.x;
  for (int c = tid; c < 16384; c += 1024) hist[c] = 0;
  unsigned kk[N / 8];   // a few registers of keys; the rest derived
  for (int i = 0; i < N / 8; ++i) kk[i] = keys[(blockIdx.x * 1024 + tid) * (N / 8) + i];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < N / 8; ++i) {
    unsigned x = kk[i];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned cell = (MODE == 2) ? ((x & 0xFFu) | 0x2000u) : (x & 16383u);   // MODE 2: 256 hot cells
      if (MODE == 1) hist[cell] = x; else atomicAdd(&hist[cell], 1u);
      x = x * 1664525u + 1013904223u;
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  unsigned s = 0;
  for (int c = tid; c < 16384; c += 1024) s += hist[c];
  if (s == 0xFFFFFFFFu) sink[0] = s;
}
int main()
{
  const int blocks = 256;
  std::vector<unsigned> h((size_t)blocks * 1024 * (N / 8));
  unsigned r = 12345;
  for (auto &v : h) { r = r * 1103515245u + 12345u; v = r >> 3; }
  unsigned *d, *sink; unsigned long long *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 8); hipMalloc(&sink, 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned long long> res(blocks);
  const char *names[3] = {"atomicAdd random cells", "plain store random cells", "atomicAdd 256 hot cells"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, d, o, sink);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, d, o, sink);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, d, o, sink);
      hipDeviceSynchronize();
    }
    hipMemcpy(res.data(), o, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : res) m += (double)v; m /= blocks;
    // s_memtime ticks at 100 MHz on this part; report ticks and ops per tick
    printf("%-28s %10.0f ticks per workgroup for %d lane-ops  -> %.1f lane-ops per tick\n", names[mode], m, 1024 * N, 1024.0 * N / m);
  }
  return 0;
}
