// Developer microbenchmark: how many small kernels per second does ONE GPU dispatch, GPU-wide?
// T host threads, one stream each, launch a kernel of G workgroups that busy-waits ~W microseconds, N times; the aggregate
// rate says whether config 5 (16-20 solver threads x ~75 small launches per solve) is bound by the dispatch rate.
//   tools/bin/launch_rate [threads] [launches per thread] [workgroups] [busy us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void busy(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks == 0xFFFFFFFFull) *sink = t0;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 1, N = argc > 2 ? atoi(argv[2]) : 2000, G = argc > 3 ? atoi(argv[3]) : 1;
  const double us = argc > 4 ? atof(argv[4]) : 0.0;
  unsigned long long* sink;
  CK(hipMalloc(&sink, 8));
  std::vector<hipStream_t> st(T);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto run = [&](int t, int n) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(busy, dim3(G), dim3(256), 0, st[t], (unsigned long long)(us * 100.0), sink);
    CK(hipStreamSynchronize(st[t]));
  };
  for (int t = 0; t < T; ++t) run(t, 50);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back(run, t, N);
  for (auto& x : th) x.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("threads %2d x %d launches of %d workgroup(s), %.1f us busy: %.0f launches/s in all (%.2f us per launch per stream)\n", T, N, G, us,
         T * N / dt, dt / N * 1e6);
  return 0;
}
