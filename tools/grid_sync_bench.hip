// Developer microbenchmark: what one grid-wide exchange costs inside a persistent launch on MI355X, in the exact
// shape the persistent ADMM kernel uses (write-through payload, sharded arrival counters, relaxed agent-scope polls,
// no release / acquire fences: MI355X_MICROARCH.md, "Valid forms", table row 1).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/grid_sync_bench tools/grid_sync_bench.hip && tools/bin/grid_sync_bench
//
// modes: 0 = barrier only, 1 = barrier + all-reduce of one double per workgroup, 2 = the two-exchange ADMM step
// (publish an 800-double tile + a partial, barrier, gather 2 x 200 16-byte cells + read all partials, publish a second
// partial, barrier, read all partials).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

typedef unsigned long long u64;
constexpr int NSHARD = 8;
constexpr int SHARD_STRIDE = 16;  // u64 per shard: one 128-byte line each
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ inline void st_f64(double* p, double v) {
  __hip_atomic_store((u64*)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double ld_f64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// all workgroups arrive, then wait until every shard has reached `target` arrivals per shard
__device__ inline bool grid_barrier(u64* shards, int nwg, unsigned epoch, unsigned* give_up) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
  __syncthreads();
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(shards + (blockIdx.x % NSHARD) * SHARD_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok_s = 1;
  }
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    // shard s receives the arrivals of the workgroups b with b % NSHARD == s
    const u64 per = lane < NSHARD ? (u64)((nwg - lane + NSHARD - 1) / NSHARD) : 0ull;
    const u64 target = per * epoch;
    unsigned spins = 0;
    bool ok = true;
    for (;;) {
      u64 v = target;
      if (lane < NSHARD) v = __hip_atomic_load(shards + lane * SHARD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(v >= target)) break;
      if (++spins > SPIN_LIMIT || __hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok && lane == 0) {
      __hip_atomic_store(give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok_s = 0;
    }
  }
  __syncthreads();
  return ok_s != 0;
}

__global__ __launch_bounds__(1024) void sync_bench(int mode, int nit, int nwg, int tile, u64* shards, unsigned* give_up,
                                                    double* tiles, double* part, const int* cells, int ncell,
                                                    double* out) {
  __shared__ double red[16];
  const int T = blockDim.x;
  double acc = 0.0;
  unsigned epoch = 0;
  for (int it = 0; it < nit; ++it) {
    const int par = it & 1;
    if (mode == 2) {
      double* mine = tiles + (size_t)blockIdx.x * tile;
      for (int e = threadIdx.x; e < tile; e += T) st_f64(mine + e, (double)(it + e));
      if (threadIdx.x == 0) st_f64(part + (size_t)par * nwg + blockIdx.x, 1.0 + it);
    } else if (mode == 1) {
      if (threadIdx.x == 0) st_f64(part + (size_t)par * nwg + blockIdx.x, 1.0 + it);
    }
    if (!grid_barrier(shards, nwg, ++epoch, give_up)) break;
    if (mode >= 1) {
      double v = 0.0;
      for (int b = threadIdx.x; b < nwg; b += T) v += ld_f64(part + (size_t)par * nwg + b);
      if (mode == 2) {
        for (int e = threadIdx.x; e < ncell; e += T) {
          const int c = cells[(size_t)blockIdx.x * ncell + e];  // index of a 16-byte cell in the tile buffer
          v += ld_f64(tiles + 2 * (size_t)c) + ld_f64(tiles + 2 * (size_t)c + 1);
        }
      }
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      double t = 0.0;
      for (int w = 0; w < T / 64; ++w) t += red[w];
      acc += t;
      __syncthreads();
    }
    if (mode == 2) {
      if (threadIdx.x == 0) st_f64(part + (size_t)(2 + par) * nwg + blockIdx.x, acc);
      if (!grid_barrier(shards, nwg, ++epoch, give_up)) break;
      double v = 0.0;
      for (int b = threadIdx.x; b < nwg; b += T) v += ld_f64(part + (size_t)(2 + par) * nwg + b);
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      double t = 0.0;
      for (int w = 0; w < T / 64; ++w) t += red[w];
      acc = 1e-9 * t;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int nit = argc > 1 ? atoi(argv[1]) : 2000;
  const int tile = 800, ncell = 400;
  for (int nwg : {128, 256}) {
    for (int T : {512, 1024}) {
      for (int mode = 0; mode < 3; ++mode) {
        u64* shards;
        unsigned* give_up;
        double *tiles, *part, *out;
        int* cells;
        CK(hipMalloc(&shards, NSHARD * SHARD_STRIDE * sizeof(u64)));
        CK(hipMalloc(&give_up, 16));
        CK(hipMalloc(&tiles, (size_t)nwg * tile * 8));
        CK(hipMalloc(&part, (size_t)4 * nwg * 8));
        CK(hipMalloc(&out, (size_t)nwg * 8));
        CK(hipMalloc(&cells, (size_t)nwg * ncell * 4));
        std::vector<int> hc((size_t)nwg * ncell);
        for (size_t i = 0; i < hc.size(); ++i) hc[i] = rand() % (nwg * tile / 2);
        CK(hipMemcpy(cells, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(tiles, 0, (size_t)nwg * tile * 8));
        CK(hipMemset(part, 0, (size_t)4 * nwg * 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float best = 1e30f;
        unsigned gu = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(shards, 0, NSHARD * SHARD_STRIDE * sizeof(u64)));
          CK(hipMemset(give_up, 0, 16));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(sync_bench, dim3(nwg), dim3(T), 0, 0, mode, nit, nwg, tile, shards, give_up, tiles, part, cells,
                             ncell, out);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CK(hipMemcpy(&gu, give_up, 4, hipMemcpyDeviceToHost));
          if (gu) break;
        }
        printf("nwg %3d threads %4d mode %d: %.2f us per iteration%s\n", nwg, T, mode, best * 1e3 / nit,
               gu ? "  (GAVE UP: a barrier timed out)" : "");
        fflush(stdout);
        CK(hipFree(shards)); CK(hipFree(give_up)); CK(hipFree(tiles)); CK(hipFree(part)); CK(hipFree(out)); CK(hipFree(cells));
      }
    }
  }
  return 0;
}
