// Developer microbenchmark: streaming write/read bandwidth in the access pattern of the pairwise kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// pattern A: 3 planes, each thread double2 at linear index (grid covers everything exactly)
__global__ void wr_linear(double2* a, double2* b, double2* c, size_t n2) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n2) { double2 v = make_double2((double)t, 1.0); a[t] = v; b[t] = v; c[t] = v; }
}
// pattern B: workgroup owns 4096 rows: 8 steps, thread t writes rows (2t, 2t+1) + 512*s  (the pair kernel's pattern)
__global__ void wr_chunk(double* a, double* b, double* c, size_t n, int steps) {
  size_t c0 = (size_t)blockIdx.x * 256 * 2 * steps;
  for (int s = 0; s < steps; ++s) {
    size_t r = c0 + (size_t)s * 512 + 2 * threadIdx.x;
    if (r + 1 < n) {
      double2 v = make_double2((double)r, 1.0);
      *reinterpret_cast<double2*>(a + r) = v;
      *reinterpret_cast<double2*>(b + r) = v;
      *reinterpret_cast<double2*>(c + r) = v;
    }
  }
}
// pattern C: like B but 2D grid (x chunks, y = k) like the pair kernel
__global__ void wr_chunk2d(double* a, double* b, double* c, size_t nq, int steps) {
  size_t slice0 = (size_t)blockIdx.y * nq;
  size_t c0 = (size_t)blockIdx.x * 256 * 2 * steps;
  for (int s = 0; s < steps; ++s) {
    size_t o = c0 + (size_t)s * 512 + 2 * threadIdx.x;
    if (o + 1 < nq) {
      size_t r = slice0 + o;
      double2 v = make_double2((double)r, 1.0);
      *reinterpret_cast<double2*>(a + r) = v;
      *reinterpret_cast<double2*>(b + r) = v;
      *reinterpret_cast<double2*>(c + r) = v;
    }
  }
}
// pattern D: the same 2-D grid and the same 16 steps per workgroup, but the workgroups of a time step INTERLEAVE at 4 KB: step s of
// workgroup x writes rows x * 512 + s * (X * 512) -- at any moment the resident workgroups write a few compact windows
// instead of one 64 KB stream each
__global__ void wr_inter2d(double* a, double* b, double* c, size_t nq, int steps) {
  size_t slice0 = (size_t)blockIdx.y * nq;
  for (int s = 0; s < steps; ++s) {
    size_t o = ((size_t)s * gridDim.x + blockIdx.x) * 512 + 2 * threadIdx.x;
    if (o + 1 < nq) {
      size_t r = slice0 + o;
      double2 v = make_double2((double)r, 1.0);
      *reinterpret_cast<double2*>(a + r) = v;
      *reinterpret_cast<double2*>(b + r) = v;
      *reinterpret_cast<double2*>(c + r) = v;
    }
  }
}
// pattern E: gangs of G workgroups interleave (stride G * 512 rows between the steps of a workgroup)
__global__ void wr_gang2d(double* a, double* b, double* c, size_t nq, int steps, int G) {
  size_t slice0 = (size_t)blockIdx.y * nq;
  size_t gang = blockIdx.x / G, g = blockIdx.x % G;
  for (int s = 0; s < steps; ++s) {
    size_t o = gang * (size_t)G * 512 * steps + ((size_t)s * G + g) * 512 + 2 * threadIdx.x;
    if (o + 1 < nq) {
      size_t r = slice0 + o;
      double2 v = make_double2((double)r, 1.0);
      *reinterpret_cast<double2*>(a + r) = v;
      *reinterpret_cast<double2*>(b + r) = v;
      *reinterpret_cast<double2*>(c + r) = v;
    }
  }
}
// pattern F: gangs of G workgroups interleave PIECES of P consecutive steps (P * 512 rows)
__global__ void wr_gangp2d(double* a, double* b, double* c, size_t nq, int steps, int G, int P) {
  size_t slice0 = (size_t)blockIdx.y * nq;
  size_t gang = blockIdx.x / G, g = blockIdx.x % G;
  for (int s = 0; s < steps; ++s) {
    size_t piece = s / P, in = s % P;
    size_t o = gang * (size_t)G * 512 * steps + ((piece * G + g) * P + in) * 512 + 2 * threadIdx.x;
    if (o + 1 < nq) {
      size_t r = slice0 + o;
      double2 v = make_double2((double)r, 1.0);
      *reinterpret_cast<double2*>(a + r) = v;
      *reinterpret_cast<double2*>(b + r) = v;
      *reinterpret_cast<double2*>(c + r) = v;
    }
  }
}
typedef double d2_t __attribute__((ext_vector_type(2)));
// pattern C with non-temporal stores
__global__ void wr_chunk2d_nt(double* a, double* b, double* c, size_t nq, int steps) {
  size_t slice0 = (size_t)blockIdx.y * nq;
  size_t c0 = (size_t)blockIdx.x * 256 * 2 * steps;
  for (int s = 0; s < steps; ++s) {
    size_t o = c0 + (size_t)s * 512 + 2 * threadIdx.x;
    if (o + 1 < nq) {
      size_t r = slice0 + o;
      d2_t v = {(double)r, 1.0};
      __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(a + r));
      __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(b + r));
      __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(c + r));
    }
  }
}
__global__ void rd_linear(const double2* a, const double2* b, const double2* c, size_t n2, double* out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n2) { double2 x = a[t], y = b[t], z = c[t]; if (x.x + y.x + z.x == -1.0) out[0] = 1.0; }
}
int main() {
  const size_t nq = 523776, K = 50, n = nq * K;
  double *a, *b, *c, *out;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8)); CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-12s %8.1f us  %8.1f GB/s\n", name, ms * 1e3, 3.0 * n * 8 / ms / 1e6);
    }
  };
  time("wr_linear", [&] { hipLaunchKernelGGL(wr_linear, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (double2*)a, (double2*)b, (double2*)c, n / 2); });
  time("wr_chunk8", [&] { hipLaunchKernelGGL(wr_chunk, dim3((n + 4095) / 4096), dim3(256), 0, 0, a, b, c, n, 8); });
  time("wr_chunk2d", [&] { hipLaunchKernelGGL(wr_chunk2d, dim3((nq + 4095) / 4096, K), dim3(256), 0, 0, a, b, c, nq, 8); });
  time("wr_chunk2d16", [&] { hipLaunchKernelGGL(wr_chunk2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16); });
  time("wr_c2d16_nt", [&] { hipLaunchKernelGGL(wr_chunk2d_nt, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16); });
  time("wr_inter2d16", [&] { hipLaunchKernelGGL(wr_inter2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16); });
  time("wr_gang2_16", [&] { hipLaunchKernelGGL(wr_gang2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 2); });
  time("wr_gang4_16", [&] { hipLaunchKernelGGL(wr_gang2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 4); });
  time("wr_gang8_16", [&] { hipLaunchKernelGGL(wr_gang2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 8); });
  time("wr_g8_p2", [&] { hipLaunchKernelGGL(wr_gangp2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 8, 2); });
  time("wr_g8_p4", [&] { hipLaunchKernelGGL(wr_gangp2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 8, 4); });
  time("wr_g64_p2", [&] { hipLaunchKernelGGL(wr_gangp2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 64, 2); });
  time("wr_g64_p4", [&] { hipLaunchKernelGGL(wr_gangp2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 64, 4); });
  time("wr_g16_p1", [&] { hipLaunchKernelGGL(wr_gangp2d, dim3((nq + 8191) / 8192, K), dim3(256), 0, 0, a, b, c, nq, 16, 16, 1); });
  time("wr_chunk1", [&] { hipLaunchKernelGGL(wr_chunk, dim3((n + 511) / 512), dim3(256), 0, 0, a, b, c, n, 1); });
  time("rd_linear", [&] { hipLaunchKernelGGL(rd_linear, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (const double2*)a, (const double2*)b, (const double2*)c, n / 2, out); });
  return 0;
}
