// Microbenchmark: what vector-instruction rate does a SIMD of gfx950 sustain with W waves resident?
// Every wave runs a long stream of INDEPENDENT plain VALU instructions (8 accumulators), integer add /
// fp32 fma / fp64 fma / mixed.  Prints wave-instructions per cycle and SIMD and the implied chip-wide rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(256) k_stream(unsigned *out, int iters, unsigned seed)
{
  unsigned a[8];
  float f[8];
  double d[8];
  for (int q = 0; q < 8; ++q) { a[q] = seed + q + threadIdx.x; f[q] = (float)(seed + q) * 1e-3f; d[q] = (double)(seed + q) * 1e-3; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (KIND == 0) a[q] = a[q] * 3u + 7u == 0u ? 1u : (a[q] + (unsigned)it);        // a few int ops
        if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(seed));
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(f[(q + 1) & 7]));
        if (KIND == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[q]) : "v"(d[(q + 1) & 7]));
        if (KIND == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[q]) : "v"(seed));
        if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[q]) : "v"(seed));
        if (KIND == 6) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(seed) : "vcc");
      }
    }
  }
  unsigned s = 0;
  for (int q = 0; q < 8; ++q) s += a[q] + (unsigned)f[q] + (unsigned)d[q];
  if (s == 0x12345u) out[threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int per_iter_instr)
{
  unsigned *out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k_stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, 200, 1u);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double instr = (double)blocks * 4.0 * iters * 64.0 * per_iter_instr / 64.0;   // wave-instructions
    printf("%-22s waves/SIMD %d: %.3f ms, %.1f G wave-instr/s\n", name, wps, best, instr / (best * 1e-3) / 1e9);
  }
  hipFree(out);
}

int main()
{
  run<1>("v_add_u32", 64);
  run<4>("v_and_b32", 64);
  run<5>("v_mul_lo_u32", 64);
  run<6>("v_cmp+v_cndmask", 128);
  run<2>("v_fma_f32", 64);
  run<3>("v_fma_f64", 64);
  return 0;
}
