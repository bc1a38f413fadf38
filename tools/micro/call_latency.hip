// Micro-benchmark (GPU experiment): host-observed latency of a synchronous "call" shaped like the PCA / kNN
// entry points: small input in, a chain of kernels, a small result out, the host waits.  Variants of the in / out /
// wait mechanics, everything else equal.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/call_latency.hip -o tools/micro/call_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_spin(const int *in, int *out, long long ticks, int last, volatile unsigned *flag, unsigned seq)
{
  const long long t0 = wall_clock64();   // 100 MHz
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = in[0] + 1;
    if (last && flag) {
      __threadfence_system();
      *flag = seq;
    }
  }
}

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
  const size_t in_bytes = 800, out_bytes = 4096;
  int *d_in, *d_out;
  hipMalloc(&d_in, in_bytes);
  hipMalloc(&d_out, out_bytes);
  hipMemset(d_in, 0, in_bytes);
  hipMemset(d_out, 0, out_bytes);
  int *p_in, *p_out;
  unsigned *p_flag;
  hipHostMalloc(&p_in, in_bytes, hipHostMallocDefault);
  hipHostMalloc(&p_out, out_bytes, hipHostMallocDefault);
  hipHostMalloc(&p_flag, 64, hipHostMallocDefault);
  memset(p_in, 0, in_bytes);
  *p_flag = 0;
  std::vector<char> pg_in(in_bytes, 0), pg_out(out_bytes, 0);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int reps = 300;
  unsigned seq = 0;
  for (int nk : {1, 15}) {
    for (long long ticks : {0ll, 1000ll}) {   // 0 or 10 us per kernel
      for (int variant = 0; variant < 5; ++variant) {
        std::vector<double> ts;
        for (int r = 0; r < reps + 20; ++r) {
          const double t0 = now_us();
          ++seq;
          const int *kin = d_in;
          int *kout = d_out;
          if (variant == 0) hipMemcpyAsync(d_in, pg_in.data(), in_bytes, hipMemcpyHostToDevice, s);
          if (variant == 1) hipMemcpyAsync(d_in, p_in, in_bytes, hipMemcpyHostToDevice, s);
          if (variant >= 2) { kin = p_in; kout = p_out; }   // zero copy both ways
          for (int i = 0; i < nk; ++i)
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, kin, kout, ticks, i == nk - 1, variant >= 3 ? p_flag : nullptr, seq);
          if (variant == 0) hipMemcpyAsync(pg_out.data(), d_out, out_bytes, hipMemcpyDeviceToHost, s);
          if (variant == 1) hipMemcpyAsync(p_out, d_out, out_bytes, hipMemcpyDeviceToHost, s);
          if (variant <= 2) hipStreamSynchronize(s);
          if (variant == 3) {
            while (*(volatile unsigned *)p_flag != seq) {}
          }
          if (variant == 4) {   // spin on the flag, then let the runtime retire the commands too
            while (*(volatile unsigned *)p_flag != seq) {}
            const double t1 = now_us();
            hipStreamSynchronize(s);
            if (r == reps + 19) printf("      (variant 4: stream sync after the flag adds %.1f us)\n", now_us() - t1);
          }
          if (r >= 20) ts.push_back(now_us() - t0);
          if (variant == 3) hipStreamSynchronize(s);   // outside the timed part
        }
        std::sort(ts.begin(), ts.end());
        static const char *names[5] = {"pageable copies + sync", "pinned copies + sync", "zero copy + sync", "zero copy + flag spin",
                                       "zero copy + flag spin + sync"};
        printf("kernels %2d x %3lld us  %-30s p50 %7.1f us  p10 %7.1f  p90 %7.1f  (minus kernel time: %6.1f)\n", nk, ticks / 100,
               names[variant], ts[ts.size() / 2], ts[ts.size() / 10], ts[ts.size() * 9 / 10], ts[ts.size() / 2] - nk * ticks / 100.0);
      }
    }
  }
  return 0;
}
