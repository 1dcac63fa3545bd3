// Developer microbenchmark: issue cost of v_mfma_f64_16x16x4_f64 on gfx950 (dependent chain / independent chains,
// 1..4 waves per SIMD), in shader cycles (s_memtime) and wall-clock ns.  hipcc --offload-arch=gfx950 -O3 -o mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void bench(int n, double* out, unsigned long long* clk) {
  double4_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  __syncthreads();
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  const unsigned long long w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = w1 - w0;
  }
}

template <int CHAINS>
void run(int threads, int blocks, int n, double* out, unsigned long long* clk) {
  unsigned long long h[2];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(bench<CHAINS>, dim3(blocks), dim3(threads), 0, 0, n, out, clk);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double mf = (double)n * CHAINS;
  printf("chains=%d waves/WG=%2d blocks=%4d: %7.1f cycles/mfma/wave, %7.1f ns/mfma/wave (wall %llu ticks) -> per SIMD %.1f ns/mfma\n",
         CHAINS, threads / 64, blocks, (double)h[0] / mf, (double)h[1] * 10.0 / mf, h[1],
         (double)h[1] * 10.0 / (mf * ((threads / 64 + 3) / 4)));
}

int main() {
  double* out;
  unsigned long long* clk;
  hipMalloc(&out, 1024 * 1024 * sizeof(double));
  hipMalloc(&clk, 2 * sizeof(unsigned long long));
  const int n = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    run<1>(threads, 1, n, out, clk);
    run<4>(threads, 1, n, out, clk);
  }
  run<1>(1024, 128, n, out, clk);
  run<4>(1024, 128, n, out, clk);
  run<4>(1024, 256, n, out, clk);
  return 0;
}
