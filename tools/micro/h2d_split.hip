// Microbenchmark: does splitting one 12 MB host-to-device copy over two streams (two DMA engines) beat one copy?
// hipcc --offload-arch=gfx950 -O2 -o h2d_split h2d_split.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
int main()
{
  const size_t B = 12u << 20;
  char *h; char *d[2];
  hipHostMalloc((void **)&h, B, hipHostMallocDefault);
  memset(h, 1, B);
  hipMalloc((void **)&d[0], B); hipMalloc((void **)&d[1], B);
  hipStream_t s[2];
  hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
  auto run = [&](int parts, int reps) {
    // keep 2 frames' worth of copies queued ahead, like the streaming loop
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      for (int p = 0; p < parts; ++p)
        hipMemcpyAsync(d[r & 1] + p * (B / parts), h + p * (B / parts), B / parts, hipMemcpyHostToDevice, s[p % 2]);
      if (r >= 2) { hipStreamSynchronize(s[0]); if (parts > 1) hipStreamSynchronize(s[1]); }
    }
    hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return (double)B * reps / dt / 1e9;
  };
  run(1, 100); run(2, 100);   // warm up (first DMA out of every pinned page)
  for (int k = 0; k < 3; ++k) {
    printf("one copy per frame, one stream : %.1f GB/s\n", run(1, 300));
    printf("two halves on two streams      : %.1f GB/s\n", run(2, 300));
    printf("four quarters on two streams   : %.1f GB/s\n", run(4, 300));
  }
  return 0;
}
