// Micro-benchmark (GPU experiment): latency of a cross-stream dependency (hipEventRecord on one stream,
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/hop_latency.hip -o tools/micro/hop_latency
// hipStreamWaitEvent on another) compared with back-to-back launches on one stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_small(int *p, long long spin)
{
  const long long t0 = wall_clock64();   // 100 MHz
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main()
{
  int *d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  const int K = 1000;
  const long long spins[2] = {0, 2000};   // 0 and 20 us of device-side spinning per kernel
  hipEvent_t ev[2];
  hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
  hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, a, d, spins[rep]);
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    // ping-pong: a -> b -> a -> ...
    for (int i = 0; i < K; ++i) {
      hipStream_t s = (i & 1) ? b : a, o = (i & 1) ? a : b;
      hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, spins[rep]);
      hipEventRecord(ev[i & 1], s);
      hipStreamWaitEvent(o, ev[i & 1], 0);
    }
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    const double same = std::chrono::duration<double, std::micro>(t1 - t0).count() / K;
    const double hop = std::chrono::duration<double, std::micro>(t2 - t1).count() / K;
    printf("spin %lld0 ns, rep %d: same-stream launch %.2f us/kernel, cross-stream ping-pong %.2f us/kernel (hop ~ %.2f us)\n", spins[rep], rep, same, hop, hop - same);
  }
  return 0;
}
