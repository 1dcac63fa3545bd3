// Assembly half of the SCP hot path on gfx950: context, kinematics (a4/a7), fixed bounds (a2), the O(N^2 K)
// pairwise passes (a5, a8, constraint generation) and the SCP relative step (a1).
// Reference: /root/reference/src/path_planning/solvers/scp.py (line numbers cited per kernel).
#include "scp_common.h"
#include "scp_pair_device.h"

#include <sys/prctl.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

// ----------------------------------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------------------------------
extern "C" int scp_abi_version(void) { return SCP_ABI_VERSION; }

extern "C" int scp_ctx_create(int device, void* hip_stream, scp_ctx** out) {
  if (!out) return SCP_ERR_INVALID;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) return SCP_ERR_HIP;
  scp_ctx* ctx = new scp_ctx();
  memset(ctx, 0, sizeof(*ctx));
  ctx->device = device;
  ctx->stream = (hipStream_t)hip_stream;
  ctx->timing = 1;
  ctx->small_pass = getenv("SCP_NO_SMALL_PASS") ? 0 : 1;  // (developer switch; scp_ctx_set_option at run time)
  if (hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->n_cu = 0;
  if (hipMalloc(&ctx->d_scratch, 72 * sizeof(double)) != hipSuccess ||
      hipMemset(ctx->d_scratch, 0, 72 * sizeof(double)) != hipSuccess ||  // ([64]: ticket counter of scp_rel_step)
      hipHostMalloc(&ctx->h_scratch, 72 * sizeof(double)) != hipSuccess ||
      memset(ctx->h_scratch, 0, 72 * sizeof(double)) == nullptr ||        // ([64]: its completion word)
      hipHostGetDevicePointer((void**)&ctx->h_scratch_dev, ctx->h_scratch, 0) != hipSuccess ||
      hipHostMalloc(&ctx->h_mirror, sizeof(scp_stats_mirror)) != hipSuccess ||
      hipHostGetDevicePointer((void**)&ctx->d_mirror, ctx->h_mirror, 0) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipEventCreate(&ctx->pair_ev0) != hipSuccess || hipEventCreate(&ctx->pair_ev1) != hipSuccess ||
      hipMalloc(&ctx->wg_part, 4 * SCP_SMALL_MAX_WG * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&ctx->d_ticket, 64) != hipSuccess || hipMemset(ctx->d_ticket, 0, 64) != hipSuccess) {
    delete ctx;
    return SCP_ERR_HIP;
  }
  *out = ctx;
  return SCP_OK;
}

// Per-context switches (include/scp_hip.h).  "kernel_timing": HIP events around every pairwise kernel and every QP solve
// (two queue packets each) are what `linearize_ms`, `violations_ms` and `solve_ms` are read from; 0: none are recorded --
// the pass times read 0, solve_ms becomes the host's wall clock around the solve (the host waits for its result anyway).
// "single_launch_passes": the one-launch form of the pairwise passes of small problems (pair_pass_kernel<.., SMALL>); 0:
// prep kernel + pass + compaction as for large problems (same results; tests compare the two).
extern "C" int scp_ctx_set_option(scp_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return SCP_ERR_INVALID;
  if (strcmp(key, "kernel_timing") == 0) {
    ctx->timing = value ? 1 : 0;
    if (!ctx->timing) ctx->pair_timed = false;
    return SCP_OK;
  }
  if (strcmp(key, "single_launch_passes") == 0) {
    ctx->small_pass = value ? 1 : 0;
    return SCP_OK;
  }
  return scp_fail(ctx, SCP_ERR_INVALID, "ctx_set_option: unknown key '%s'", key);
}

extern "C" void scp_ctx_destroy(scp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ctx->d_scratch);
  if (ctx->wg_part) (void)hipFree(ctx->wg_part);
  if (ctx->wg_rows) (void)hipFree(ctx->wg_rows);
  if (ctx->d_ticket) (void)hipFree(ctx->d_ticket);
  if (ctx->cmp_map) (void)hipFree(ctx->cmp_map);
  if (ctx->cmp_tot) (void)hipFree(ctx->cmp_tot);
  if (ctx->tm_scratch) (void)hipFree(ctx->tm_scratch);
  (void)hipHostFree(ctx->h_scratch);
  (void)hipHostFree(ctx->h_mirror);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  (void)hipEventDestroy(ctx->pair_ev0);
  (void)hipEventDestroy(ctx->pair_ev1);
  delete ctx;
}

hipError_t scp_raise_lds_limit(int device, const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> allowed;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = allowed[std::make_pair(device, kernel)];
  if (have >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

static std::atomic<int> g_host_wait_mode{0};

extern "C" void scp_set_host_wait(int mode) {
  g_host_wait_mode.store(mode == 1 || mode == 2 ? mode : 0, std::memory_order_relaxed);
}

bool scp_wait_host_word(volatile unsigned long long* word, unsigned long long seq, int timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  const int wait_mode = g_host_wait_mode.load(std::memory_order_relaxed);
  const bool sleepy = wait_mode != 0;
  const unsigned spin_first = wait_mode == 2 ? 0u : 2000u;  // mode 2: nap at once (more solver threads than cores)
  if (sleepy) {
    // the kernel's default timer slack (50 us) would stretch every 20 us nap to ~75 us -- a third of a 25-step persistent
    // launch; 1 us of slack for this thread
    static thread_local bool slack_set = false;
    if (!slack_set) {
      (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
      slack_set = true;
    }
  }
  unsigned spins = 0;
  while (*word != seq) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    ++spins;
    if (sleepy && spins > spin_first) {  // mode 1: ~20 us of spinning first: the kernel is a long one, give the core away
      struct timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
      if ((spins & 0x3FF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) break;
    } else if ((spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) {
      break;
    }
  }
  const bool ok = *word == seq;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return ok;
}

int scp_ctx_wait_stats(scp_ctx* ctx, scp_pair_stats* out) {
  if (ctx->mirror_seq == 0) return scp_fail(ctx, SCP_ERR_STATE, "no pass with a row list has run yet");
  if (!scp_wait_host_word(&ctx->h_mirror->seq, ctx->mirror_seq, 30)) {
    SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // surfaces a launch failure, if that is why nothing arrived
    if (ctx->h_mirror->seq != ctx->mirror_seq) return scp_fail(ctx, SCP_ERR_HIP, "pair pass: the stats mirror was not written");
  }
  *out = ctx->h_mirror->stats;
  return SCP_OK;
}

// scp_rel_step's result from the partial sums the latest small-problem violations pass left in the mirror (call after
// scp_ctx_wait_stats): the same sums in the same order as scp_rel_step's host side
void scp_ctx_mirror_rel(scp_ctx* ctx, int64_t n, double* out) {
  const int blocks = (int)((n + 256 * 8 - 1) / (256 * 8)) < 32 ? (int)((n + 256 * 8 - 1) / (256 * 8)) : 32;
  double d2 = 0.0, b2 = 0.0;
  for (int b = 0; b < blocks; ++b) {
    d2 += ctx->h_mirror->rel[2 * b];
    b2 += ctx->h_mirror->rel[2 * b + 1];
  }
  out[0] = std::sqrt(d2);
  out[1] = std::sqrt(b2);
  out[2] = out[0] / out[1];
}

extern "C" const char* scp_last_error(const scp_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int scp_ctx_synchronize(scp_ctx* ctx) {
  if (!ctx) return SCP_ERR_INVALID;
  SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// layout changes [N][K][D] <-> [K][N*D]
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_time_major_kernel(int N, int K, int D, const double* __restrict__ src,
                                                             double* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // destination index (coalesced writes)
  const int64_t C = (int64_t)N * D;
  if (t >= C * K) return;
  const int k = (int)(t / C);
  const int c = (int)(t % C);
  const int i = c / D, d = c % D;
  dst[t] = src[((int64_t)i * K + k) * D + d];
}

__global__ __launch_bounds__(256) void from_time_major_kernel(int N, int K, int D, const double* __restrict__ src,
                                                               double* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // source index (coalesced reads)
  const int64_t C = (int64_t)N * D;
  if (t >= C * K) return;
  const int k = (int)(t / C);
  const int c = (int)(t % C);
  const int i = c / D, d = c % D;
  dst[((int64_t)i * K + k) * D + d] = src[t];
}

int scp_launch_to_time_major(scp_ctx* ctx, int N, int K, int D, const double* src, double* dst) {
  const int64_t n = (int64_t)N * K * D;
  hipLaunchKernelGGL(to_time_major_kernel, dim3(scp_cdiv(n, 256)), dim3(256), 0, ctx->stream, N, K, D, src, dst);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

int scp_launch_from_time_major(scp_ctx* ctx, int N, int K, int D, const double* src, double* dst) {
  const int64_t n = (int64_t)N * K * D;
  hipLaunchKernelGGL(from_time_major_kernel, dim3(scp_cdiv(n, 256)), dim3(256), 0, ctx->stream, N, K, D, src, dst);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// a4 / a7 kinematics (scp.py:371-397, :559-595).  One thread per output sample; the inner sum runs in the
// reference's order with separately rounded multiply and add (no FMA) so the result is bitwise the
// reference's.
// ----------------------------------------------------------------------------------------------------
// one term of scp_rel_step's two sums (scp.py:157-159): ONE definition for rel_step_partial_kernel and for the tail of the
// small-problem violations pass, which emulates that kernel's blocks one after the other (same sums, same bits)
__device__ inline void rel_accum(double x, double y, double& d2, double& b2) {
  d2 += (x - y) * (x - y);
  b2 += y * y;
}
__host__ __device__ inline int rel_step_blocks(int64_t n) {
  const int b = (int)((n + 256 * 8 - 1) / (256 * 8));
  return b < 32 ? b : 32;
}

// position (and velocity) of one coordinate at step k from its acceleration samples a[0], a[stride], ...: ONE definition for
// the kinematics kernel and for the small-problem violations pass that derives its positions from the QP's time-major
// solution itself (pair_pass_kernel<.., SMALL>), so that both produce the same bits
__device__ inline void kin_point(const double* __restrict__ a, int64_t stride, int k, double h, double pi, double vi,
                                 double& p_out, double& v_out) {
#pragma clang fp contract(off)  // every product below is rounded before it is added, as numpy does
  double v = vi;
  const double hk = h * (double)k;
  const double hkv = hk * vi;
  double p = pi + hkv;  // p0 + (h*k)*v0   scp.py:393
  const double hh = h * h;
  constexpr int KIN_CHUNK = 32;  // loads in flight per pass; the sums stay in the reference's order
  for (int j0 = 0; j0 < k; j0 += KIN_CHUNK) {
    double av[KIN_CHUNK];
#pragma unroll
    for (int u = 0; u < KIN_CHUNK; ++u) av[u] = j0 + u < k ? a[(int64_t)(j0 + u) * stride] : 0.0;
#pragma unroll
    for (int u = 0; u < KIN_CHUNK; ++u) {
      if (j0 + u < k) {
        const int j = j0 + u;
        const double aj = av[u];
        const double hv = h * aj;
        v = v + hv;  // scp.py:390
        const double w = hh * ((double)(k - j) - 0.5);
        const double wa = w * aj;
        p = p + wa;  // scp.py:395
      }
    }
  }
  p_out = p;
  v_out = v;
}

__global__ __launch_bounds__(256) void kinematics_kernel(int N, int K, int D, double h,
                                                          const double* __restrict__ acc,
                                                          const double* __restrict__ p0,
                                                          const double* __restrict__ v0, double* __restrict__ pos,
                                                          double* __restrict__ vel, double* __restrict__ acc_copy) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)N * K * D) return;
  if (acc_copy) acc_copy[t] = acc[t];  // (the solver's final launch also hands the accelerations out: no copy launch)
  const int d = (int)(t % D);
  const int k = (int)((t / D) % K);
  const int i = (int)(t / ((int64_t)D * K));
  double p, v;
  kin_point(acc + (int64_t)i * K * D + d, D, k, h, p0[i * D + d], v0[i * D + d], p, v);
  pos[t] = p;
  if (vel) vel[t] = v;
}

extern "C" int scp_kinematics(scp_ctx* ctx, int N, int K, int D, double h, const double* acc, const double* p0,
                              const double* v0, double* pos_out, double* vel_out) {
  if (!ctx) return SCP_ERR_INVALID;
  SCP_REQUIRE(ctx, N > 0 && K > 0 && (D == 2 || D == 3), "kinematics: bad shape N=%d K=%d D=%d", N, K, D);
  SCP_REQUIRE(ctx, acc && p0 && v0 && pos_out, "kinematics: null pointer");
  const int64_t n = (int64_t)N * K * D;
  hipLaunchKernelGGL(kinematics_kernel, dim3(scp_cdiv(n, 256)), dim3(256), 0, ctx->stream, N, K, D, h, acc, p0,
                     v0, pos_out, vel_out, (double*)nullptr);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

// scp_kinematics + a copy of `acc` to acc_copy in the same launch (scp_common.h)
int scp_launch_kinematics_copy(scp_ctx* ctx, int N, int K, int D, double h, const double* acc, const double* p0,
                               const double* v0, double* pos_out, double* vel_out, double* acc_copy) {
  const int64_t n = (int64_t)N * K * D;
  hipLaunchKernelGGL(kinematics_kernel, dim3(scp_cdiv(n, 256)), dim3(256), 0, ctx->stream, N, K, D, h, acc, p0,
                     v0, pos_out, vel_out, acc_copy);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// a2 bounds (scp.py:189-190, :194-195, :206-224, :234-257)
// ----------------------------------------------------------------------------------------------------
struct BoundParams {
  double vel_min, vel_max, acc_min, acc_max, jerk_min, jerk_max;
  double pmin[3], pmax[3];
};

__device__ inline void bound_of(const BoundParams& bp, int block, int i, int k, int d, int K, int D, double h,
                                const double* p0, const double* v0, const double* pf, const double* vf,
                                double& lo, double& hi) {
#pragma clang fp contract(off)
  const int s = i * D + d;
  if (block == 0) {  // jerk
    lo = bp.jerk_min;
    hi = bp.jerk_max;
  } else if (block == 1) {  // acc
    lo = bp.acc_min;
    hi = bp.acc_max;
  } else if (block == 2) {  // vel: row k is the state k+1
    if (k < K - 1) {
      lo = bp.vel_min - v0[s];  // scp.py:218-221
      hi = bp.vel_max - v0[s];
    } else {
      lo = hi = vf[s] - v0[s];  // scp.py:223-224
    }
  } else {  // pos
    const double hk = h * (double)(k + 1);
    const double hkv = hk * v0[s];
    const double off = p0[s] + hkv;  // scp.py:246-247
    if (k < K - 1) {
      lo = bp.pmin[d] - off;  // scp.py:251-254
      hi = bp.pmax[d] - off;
    } else {
      lo = hi = pf[s] - off;  // scp.py:256-257
    }
  }
}

// reference stacking order: [jerk (N,K-1,D) | acc (N,K,D) | vel | pos]
__global__ __launch_bounds__(256) void bounds_ref_order_kernel(BoundParams bp, int N, int K, int D, double h,
                                                                const double* p0, const double* v0,
                                                                const double* pf, const double* vf,
                                                                double* __restrict__ l, double* __restrict__ u) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nj = (int64_t)N * (K - 1) * D, na = (int64_t)N * K * D;
  if (t >= nj + 3 * na) return;
  int block, i, k, d;
  if (t < nj) {
    block = 0;
    d = (int)(t % D);
    k = (int)((t / D) % (K - 1));
    i = (int)(t / ((int64_t)D * (K - 1)));
  } else {
    const int64_t r = t - nj;
    block = 1 + (int)(r / na);
    const int64_t e = r % na;
    d = (int)(e % D);
    k = (int)((e / D) % K);
    i = (int)(e / ((int64_t)D * K));
  }
  double lo, hi;
  bound_of(bp, block, i, k, d, K, D, h, p0, v0, pf, vf, lo, hi);
  l[t] = lo;
  u[t] = hi;
}

// time-major stacked layout used by the QP: row = block offset + k, column c = i*D + d
__global__ __launch_bounds__(256) void bounds_time_major_kernel(BoundParams bp, int N, int K, int D, double h,
                                                                 const double* p0, const double* v0,
                                                                 const double* pf, const double* vf,
                                                                 double* __restrict__ l, double* __restrict__ u,
                                                                 double* __restrict__ states_out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t C = (int64_t)N * D;
  const int rows = 4 * K - 1;
  if (states_out && t < 4 * C) {  // the QP's own copy of [p0 | v0 | pf | vf] (lean persistent kernels), no copy launches
    const int which = (int)(t / C);
    const int64_t c = t - which * C;
    states_out[t] = which == 0 ? p0[c] : (which == 1 ? v0[c] : (which == 2 ? pf[c] : vf[c]));
  }
  if (t >= C * rows) return;
  const int row = (int)(t / C);
  const int c = (int)(t % C);
  int block, k;
  if (row < K - 1) {
    block = 0;
    k = row;
  } else {
    block = 1 + (row - (K - 1)) / K;
    k = (row - (K - 1)) % K;
  }
  double lo, hi;
  bound_of(bp, block, c / D, k, c % D, K, D, h, p0, v0, pf, vf, lo, hi);
  l[t] = lo;
  u[t] = hi;
}

static void fill_bound_params(BoundParams& bp, int D, const double* limits, const double* space) {
  bp.vel_min = limits[0];
  bp.vel_max = limits[1];
  bp.acc_min = limits[2];
  bp.acc_max = limits[3];
  bp.jerk_min = limits[4];
  bp.jerk_max = limits[5];
  for (int d = 0; d < 3; ++d) {
    bp.pmin[d] = d < D ? space[d] : 0.0;
    bp.pmax[d] = d < D ? space[D + d] : 0.0;
  }
}

extern "C" int scp_fixed_bounds(scp_ctx* ctx, int N, int K, int D, double h, const double* limits,
                                const double* space, const double* p0, const double* v0, const double* pf,
                                const double* vf, double* l_out, double* u_out) {
  if (!ctx) return SCP_ERR_INVALID;
  SCP_REQUIRE(ctx, N > 0 && K > 1 && (D == 2 || D == 3), "fixed_bounds: bad shape N=%d K=%d D=%d", N, K, D);
  SCP_REQUIRE(ctx, limits && space && p0 && v0 && pf && vf && l_out && u_out, "fixed_bounds: null pointer");
  BoundParams bp;
  fill_bound_params(bp, D, limits, space);
  const int64_t m = (int64_t)N * D * (4 * K - 1);
  hipLaunchKernelGGL(bounds_ref_order_kernel, dim3(scp_cdiv(m, 256)), dim3(256), 0, ctx->stream, bp, N, K, D, h,
                     p0, v0, pf, vf, l_out, u_out);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

int scp_launch_bounds_time_major(scp_ctx* ctx, int N, int K, int D, double h, const double* limits,
                                 const double* space, const double* p0, const double* v0, const double* pf,
                                 const double* vf, double* l_tm, double* u_tm, double* states_out) {
  BoundParams bp;
  fill_bound_params(bp, D, limits, space);
  const int64_t m = (int64_t)N * D * (4 * K - 1);  // (4K - 1 >= 4 rows: the grid covers the 4 N D states too)
  hipLaunchKernelGGL(bounds_time_major_kernel, dim3(scp_cdiv(m, 256)), dim3(256), 0, ctx->stream, bp, N, K, D, h,
                     p0, v0, pf, vf, l_tm, u_tm, states_out);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// pairwise passes
// ----------------------------------------------------------------------------------------------------
// (called by thread 0 of the prep kernel that precedes every pairwise pass: one launch less per pass)
__device__ inline void pair_stats_init(scp_pair_stats* s) {
  s->min_dist = __longlong_as_double(0x7FF0000000000000LL);
  s->first_violation = 0xFFFFFFFFFFFFFFFFULL;
  s->n_selected = 0;
  s->max_violation = -__longlong_as_double(0x7FF0000000000000LL);
}

// Wavefront reductions on DPP moves (row_shr 1, 2, 4, 8, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows
// 2, 3): result in LANE 63.  Lanes without a source keep their own value (idempotent operators only).  The
// __shfl_xor butterfly goes through the LDS crossbar and costs about ten times as much.
template <int CTRL, int ROW_MASK>
__device__ inline unsigned long long dpp_self_u64(unsigned long long v) {
  const int lo = (int)(v & 0xFFFFFFFFu), hi = (int)(v >> 32);
  const unsigned int l2 = (unsigned int)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
  const unsigned int h2 = (unsigned int)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
  return ((unsigned long long)h2 << 32) | l2;
}
template <typename Op>
__device__ inline unsigned long long wave_reduce_u64(unsigned long long v, Op op) {
  v = op(v, dpp_self_u64<0x111, 0xF>(v));
  v = op(v, dpp_self_u64<0x112, 0xF>(v));
  v = op(v, dpp_self_u64<0x114, 0xF>(v));
  v = op(v, dpp_self_u64<0x118, 0xF>(v));
  v = op(v, dpp_self_u64<0x142, 0xA>(v));
  v = op(v, dpp_self_u64<0x143, 0xC>(v));
  return v;
}
__device__ inline double wave_min(double v) {
  return __longlong_as_double((long long)wave_reduce_u64((unsigned long long)__double_as_longlong(v),
      [](unsigned long long a, unsigned long long b) {
        return (unsigned long long)__double_as_longlong(fmin(__longlong_as_double((long long)a), __longlong_as_double((long long)b)));
      }));
}
__device__ inline double wave_max(double v) {
  return __longlong_as_double((long long)wave_reduce_u64((unsigned long long)__double_as_longlong(v),
      [](unsigned long long a, unsigned long long b) {
        return (unsigned long long)__double_as_longlong(fmax(__longlong_as_double((long long)a), __longlong_as_double((long long)b)));
      }));
}
__device__ inline unsigned long long wave_min_u64(unsigned long long v) {
  return wave_reduce_u64(v, [](unsigned long long a, unsigned long long b) { return b < a ? b : a; });
}

// positive doubles compare like their bit patterns
__device__ inline void atomic_min_pos_double(double* addr, double v) {
  atomicMin((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
}
__device__ inline void atomic_max_double(double* addr, double v) {
  // general sign: CAS loop (rare: once per wave)
  unsigned long long* a = (unsigned long long*)addr;
  unsigned long long old = *a;
  while (__longlong_as_double((long long)old) < v) {
    const unsigned long long assumed = old;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}

constexpr int PAIR_THREADS = 256;
constexpr int PAIR_STEPS = 16;                             // steps per thread, 2 adjacent rows per step
// independent steps in flight per thread: 4 when the slices come through L1/L2 (latency to cover; 179 VGPRs, 2 waves
// per SIMD) and in the violations pass (streaming reads want the loads in flight), 2 for the store-bound passes with
// LDS slices (104 VGPRs, 4 waves per SIMD: linearisation 110-118 us instead of 122 at 1024 x 50; with 4 steps the
// L1/L2 path runs 1.63 ms at 4096 x 50, with 2 steps 2.16 ms)
template <bool USE_LDS, int MODE>
struct PairUnroll {
  static constexpr int value = (USE_LDS && MODE != 2 /* MODE_VIOLATIONS: streaming reads, 4 in flight */) ? 2 : 4;
};
constexpr int PAIR_ROWS = PAIR_THREADS * PAIR_STEPS * 2;   // rows (= pairs at one k) per workgroup
constexpr int PAIR_LDS_SLICE_BYTES = 16 * 1024;            // largest slice of a two-slice pass staged in LDS (beyond: L1/L2 path)
constexpr int PAIR_LDS_Q_OFF = PAIR_LDS_SLICE_BYTES / 8;   // doubles between the P slice and the Q slice of a two-slice pass
constexpr int SMALL_STEPS = 4;                             // ... of a small-problem pass: 2048 rows per workgroup, so that a
constexpr int SMALL_ROWS = PAIR_THREADS * SMALL_STEPS * 2; // 128-agent problem still spreads over 200 compute units
constexpr int64_t CMP1_MAX_WORDS = 64 * 1024;             // bitmap words (2 M rows) one workgroup compacts (compact_small_body)

enum PairMode { MODE_LINEARIZE = 0, MODE_CHECK = 1, MODE_VIOLATIONS = 2, MODE_VIOL_RECOMPUTE = 3, MODE_SELECT = 4 };
// MODE_SELECT: the linearisation pass WITHOUT its row stream (scp_select_pairs): the same distances, the same selection test
// and the same a8 reductions as MODE_LINEARIZE, but eta / l are not written -- the consumer (scp_qp_add_rows_at) recomputes
// them for the few selected rows (0.05 % at 1024 x 50) from the linearisation point.
// MODE_VIOL_RECOMPUTE: the violations pass without the 8 (D + 1) bytes-per-row read-back: eta and R - dist are
// recomputed from the slice of the linearisation point (P_tm) exactly as the linearise pass computed them, and
//   l_r - (A x)_r = (R - dist) + eta.(Q_prev_i - Q_prev_j) - eta.(Q_new_i - Q_new_j) = (R - dist) - eta.(dP_i - dP_j),
// dP = P_new - P_prev (the free motion c cancels): Q_tm holds dP.  Same LDS footprint as the linearise pass, no HBM
// stream at all.

// Work items of the linearisation pass of a large problem (pair_pass_kernel, ITEMS): up to MAX_CLASSES classes of items;
// class c covers the time steps [k0[c], k0[c + 1]) in chunks of 2 * PAIR_THREADS * steps[c] rows, cpk[c] chunks per time
// step, its items are numbered from item0[c] (time step major).  Classes are ordered by decreasing chunk size.
struct PairItems {
  static constexpr int MAX_CLASSES = 4;
  int n_items, n_classes;
  int item0[MAX_CLASSES], k0[MAX_CLASSES], cpk[MAX_CLASSES], steps[MAX_CLASSES];
};

struct PairArgs {
  int N, K, D;
  double R, h;
  int64_t q_begin, q_end, pairs;
  const double* P_tm;    // [K][N][D] time-major positions (linearize / check)
  const double* Q_tm;    // [K][N][D] time-major acceleration displacement Q = P - (p0 + k h v0)
  double* eta;           // [D][eta_stride]
  int64_t eta_stride;    // scp_eta_stride(K, nq): K*nq rounded up to even (16-byte aligned planes)
  double* l;             // [K*nq]
  double margin;         // linearize: selection margin; violations: feas_tol
  const uint32_t* bitmap;  // working-set membership (read by the violations pass)
  uint32_t* mark;          // bits set by this pass: the bitmap itself (linearize) or a scratch map (violations)
  scp_pair_stats* stats;
  int ablate;            // developer switch (profiling build, env SCP_PAIR_ABLATE): 1 = skip the streaming stores, 2 = force no-LDS
  PairItems items;       // MODE_LINEARIZE of a large problem: the work list of the persistent grid
  // ---- small problems (SMALL instantiations: the whole pass in ONE launch) --------------------------------------------
  const double* pos_a;   // [N][K][D] the pass's positions (select / check) or the linearisation point (violations)
  const double* pos_b;   // [N][K][D] new positions (violations), or NULL: derived from x_tm
  const double* x_tm;    // [K][N D] the QP's solution (time-major); p0, v0, h: its kinematics
  const double *p0, *v0;
  double* x_out;         // [N][K][D] copy of x_tm, and
  double* pos_out;       // [N][K][D] its positions (written by the first workgroup of every time step)
  int64_t* spec_rows;    // violations from a solution: also the selection around the NEW positions (margin spec_margin) ->
  int64_t spec_cap;      //   this list and
  uint32_t* spec_bitmap; //   this bitmap (replaced); NULL: none
  double spec_margin;
  const double* rel_prev;  // [N][K][D] or NULL: the tail also leaves the partial sums of scp_rel_step(x_tm, rel_prev) in the mirror
  int rel_blocks;
  unsigned long long* wg_part;  // [workgroups][4]: per-workgroup (min distance | max violation, first violation, marked rows, -)
  uint32_t* wg_rows;     // [workgroups][SMALL_ROWS]: per-workgroup sorted sub-lists of the marked rows (offsets within the workgroup)
  unsigned* ticket;      // last-workgroup-done counter (self-resetting)
  int64_t* rows;         // tail: the sorted list of the marked rows, capacity `cap`
  int64_t cap;
  uint32_t* merge_into;  // tail: the working-set bitmap the marks are merged into (violations) / that they replace (select)
  int overwrite;
  int64_t words;
  scp_stats_mirror* mirror;
  unsigned long long seq;
};

// [N][K][D] -> time-major P and Q = P - (p0 + (k h) v0) (either output may be NULL); also clears the bitmap the pass
// is about to mark (linearize), so that no memset launch is needed
__global__ __launch_bounds__(256) void pair_prep_kernel(int N, int K, int D, double h, const double* __restrict__ pos,
                                                         const double* __restrict__ p0,
                                                         const double* __restrict__ v0, double* __restrict__ P_tm,
                                                         double* __restrict__ Q_tm, scp_pair_stats* __restrict__ stats,
                                                         uint32_t* __restrict__ clear_map, int64_t clear_words) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // destination index (coalesced writes)
  const int64_t C = (int64_t)N * D;
  if (t == 0) pair_stats_init(stats);
  for (int64_t w = t; w < clear_words; w += (int64_t)gridDim.x * 256) clear_map[w] = 0u;
  if (t >= C * K) return;
  const int k = (int)(t / C);
  const int c = (int)(t % C);
  const int i = c / D, d = c % D;
  const double p = pos[((int64_t)i * K + k) * D + d];
  if (P_tm) P_tm[t] = p;
  if (Q_tm) Q_tm[t] = p - free_motion(p0[c], v0[c], k, h);
}

// lexicographic pairs (i, j), i < j: row i + 1 of the triangle starts at j = i + 2.
// The two advances of the pairwise pass without wrap loops (a divergent while per row and step was a sixth of the
// kernel's instructions): +1 wraps at most once; +2*PAIR_THREADS wraps once in the long rows of the triangle and falls
// back to the closed form where the rows are shorter than the stride (i > N - 2*PAIR_THREADS: wave-uniform regions).
__device__ inline void pair_next(int i, int j, int N, int& i1, int& j1) {
  i1 = i;
  j1 = j + 1;
  if (j1 >= N) {
    ++i1;
    j1 = i1 + 1;
  }
}
__device__ inline void pair_advance_far(int& i, int& j, int N, int s, int64_t q_new, int64_t pairs) {
  j += s;
  if (j >= N) {
    j -= N - 2 - i;
    ++i;
    if (j >= N) decode_pair(q_new < pairs ? q_new : pairs - 1, N, i, j);
  }
}

// One workgroup = PAIR_ROWS consecutive local rows of one time step k.  The k-slices of P and Q are staged in
// LDS once per workgroup (16-byte loads): every row then costs four LDS reads (P_i, Q_i broadcast within the
// wave; P_j, Q_j: consecutive lanes -> consecutive 8*D-byte slots, conflict free), and the only HBM traffic is
// the fully coalesced, 16-byte-per-lane streaming write (linearize) or read (violations) of the compact rows:
// 8*(D+1) bytes per row.  PAIR_UNROLL independent steps per thread keep loads/stores and the fp64 chains of
// several rows in flight; the selection / first-violation bookkeeping is behind a wave-uniform ballot.
#ifdef SCP_PHASE_PROFILE
// developer build (make prof): shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) every workgroup of the latest
// pairwise kernel spent -> the clock the kernel actually ran at (MI355X_MICROARCH.md, in-kernel clock check)
constexpr int SCP_PAIR_CLK_WGS = 4096;
__device__ unsigned long long scp_pair_clk[2 * SCP_PAIR_CLK_WGS];
__device__ unsigned long long scp_pair_t0[SCP_PAIR_CLK_WGS];  // absolute 100 MHz stamp at which the workgroup started
// ... and the 100 MHz stamps of the LAST workgroup of a small-problem pass (thread 0), phase by phase (tools/small_pass_profile.py)
__device__ unsigned long long scp_small_clk[16];
#define SMALL_STAMP(i) do { if (threadIdx.x == 0) small_clk_local[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SMALL_STAMP(i) do { } while (0)
#endif

__device__ inline int block_exclusive_scan(int v, int* total);  // (256 threads; defined with the compaction kernels below)


// SMALL (problems whose bitmap one workgroup compacts: <= 2 M rows, e.g. 128 agents x 50 steps): the pass is ONE launch.
// Every workgroup stages its time step straight from the [N][K][D] arrays (no prep kernel) -- the violations pass can even
// derive the new positions from the QP's time-major solution (kin_point: no from_time_major, no kinematics launch) --,
// leaves its reduction as a per-workgroup partial (no atomics on the stats, nothing to initialise), and the LAST workgroup
// to finish (a ticket) reduces the partials, compacts the bitmap into the sorted row list (compact_small_body) and
// publishes the stats in the host mirror.  Same values, same list: the order of the partials does not matter for min / max.
template <int D, int MODE, bool USE_LDS, bool SMALL = false>
__global__ __launch_bounds__(PAIR_THREADS) void pair_pass_kernel(PairArgs a) {
  constexpr int PAIR_UNROLL = PairUnroll<USE_LDS, MODE>::value;
#ifdef SCP_PHASE_PROFILE
  const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) double lds[];
#ifdef SCP_PHASE_PROFILE
  unsigned long long small_clk_local[16] = {};
  SMALL_STAMP(0);
#endif
  const int N = a.N;
  const int64_t nq = a.q_end - a.q_begin;
  constexpr int STEPS = SMALL ? SMALL_STEPS : PAIR_STEPS;
  constexpr bool NEED_P = MODE != MODE_VIOLATIONS;
  constexpr bool NEED_Q = MODE != MODE_CHECK && MODE != MODE_SELECT;
  constexpr bool VIOL = MODE == MODE_VIOLATIONS || MODE == MODE_VIOL_RECOMPUTE;  // selects violated rows, reduces max violation
  // ITEMS (the linearisation of large problems): a 1-D grid over a list of work items (time step k, chunk of 2 * PAIR_THREADS
  // * steps rows) ordered LARGE -> SMALL (PairItems): the bulk in chunks of PAIR_STEPS steps, the last time steps in half-
  // and quarter-size chunks, so that the compute units do not drain one by one while the last full-size workgroups finish
  // (tools/pair_timeline.py: with equal chunks a quarter of the slot-time was idle, most of it in a 30 us tail).  A persistent
  // grid taking the same items from a counter was measured too: 139 registers (3 workgroups per CU) or spills, slower.
  constexpr bool ITEMS = MODE == MODE_LINEARIZE && !SMALL;

  __shared__ uint32_t small_map_store[SMALL ? SMALL_ROWS / 32 : 1];
  __shared__ uint32_t small_map2_store[SMALL && MODE == MODE_VIOL_RECOMPUTE ? SMALL_ROWS / 32 : 1];
  uint32_t* const small_map = small_map_store;
  uint32_t* const small_map2 = small_map2_store;  // the speculative selection around the NEW positions (a.spec_rows)
  const double thr = a.R - 0.01;  // scp.py:610
  const double INF = __longlong_as_double(0x7FF0000000000000LL);
  double my_min = INF, my_maxv = -INF;
  unsigned long long my_first = 0xFFFFFFFFFFFFFFFFULL;

  // one item: rows [chunk * rows_item - par, ... + rows_item) of time step k, rows_item = 2 * PAIR_THREADS * nsteps
  auto run_item = [&](const int k, const int64_t chunk, const int nsteps_rt) {
  const int nsteps = ITEMS ? nsteps_rt : STEPS;
  const int64_t slice0 = (int64_t)k * nq;       // first local row of this k
  const int64_t par = slice0 & 1;               // keep every thread's first row at an even local row id
  const int ROWS = PAIR_THREADS * nsteps * 2;
  const int64_t c0 = chunk * ROWS - par;  // first local pair offset of this item
  const double* P = NEED_P ? a.P_tm + (int64_t)k * N * D : nullptr;
  const double* Q = NEED_Q ? a.Q_tm + (int64_t)k * N * D : nullptr;
  const double* Pn = nullptr;                     // (SMALL: the new positions, third LDS slice)
  if (SMALL) {
    if (threadIdx.x < SMALL_ROWS / 32) {
      small_map[threadIdx.x] = 0u;
      if (MODE == MODE_VIOL_RECOMPUTE) small_map2[threadIdx.x] = 0u;
    }
    double* sP = lds;
    double* sQ = lds + (NEED_P ? (int64_t)N * D : 0);
    double* sN = lds + 2 * (int64_t)N * D;
    if (MODE == MODE_VIOL_RECOMPUTE && a.spec_rows) Pn = sN;
    const int C = N * D;
    for (int c = threadIdx.x; c < C; c += PAIR_THREADS) {
      const int i = c / D, d = c - i * D;
      const int64_t g = ((int64_t)i * a.K + k) * D + d;
      if (MODE == MODE_SELECT && a.x_tm) {  // the pass's positions are the kinematics of the QP's solution (after QP#0)
        double pn, vn;
        kin_point(a.x_tm + c, C, k, a.h, a.p0[c], a.v0[c], pn, vn);
        if (chunk == 0) {
          store_coherent(a.pos_out + g, pn);
          store_coherent(a.x_out + g, a.x_tm[(int64_t)k * C + c]);
        }
        sP[c] = pn;
        continue;
      }
      const double pa = a.pos_a[g];
      if (NEED_P) sP[c] = pa;
      if (MODE == MODE_VIOL_RECOMPUTE) {
        double pn;
        if (a.x_tm) {
          double vn;
          kin_point(a.x_tm + c, C, k, a.h, a.p0[c], a.v0[c], pn, vn);
          if (chunk == 0) {
            store_coherent(a.pos_out + g, pn);
            store_coherent(a.x_out + g, a.x_tm[(int64_t)k * C + c]);
          }
        } else {
          pn = a.pos_b[g];
        }
        sQ[c] = pn - pa;  // dP = P_new - P_prev (pair_prep_delta_kernel)
        if (a.spec_rows) sN[c] = pn;
      }
    }
    __syncthreads();
    SMALL_STAMP(1);
    P = sP;
    Q = sQ;
  } else if (USE_LDS) {
    // two slices: the second one at a COMPILE-TIME distance from the first (the LDS path takes slices of <= 16 KB), so that
    // P_i and Q_i are one address and an immediate offset -- one shift per index instead of a shift and two adds
    double* sP = lds;
    double* sQ = lds + (NEED_P ? PAIR_LDS_Q_OFF : 0);
    {
      const int n2 = (N * D) >> 1;  // slices are 16-byte aligned: N*D*8 bytes from an aligned base, even count or tail
      for (int t = threadIdx.x; t < n2; t += PAIR_THREADS) {
        if (NEED_P) reinterpret_cast<double2*>(sP)[t] = reinterpret_cast<const double2*>(P)[t];
        if (NEED_Q) reinterpret_cast<double2*>(sQ)[t] = reinterpret_cast<const double2*>(Q)[t];
      }
      if (((N * D) & 1) && threadIdx.x == 0) {
        if (NEED_P) sP[N * D - 1] = P[N * D - 1];
        if (NEED_Q) sQ[N * D - 1] = Q[N * D - 1];
      }
      __syncthreads();
    }
    P = sP;
    Q = sQ;
  }

  // first pair of this thread
  const int64_t off0 = c0 + 2 * threadIdx.x;
  int ci = 0, cj = 1;
  {
    int64_t qf = a.q_begin + (off0 < 0 ? 0 : off0);
    if (qf >= a.pairs) qf = a.pairs - 1;
    decode_pair(qf, N, ci, cj);
    if (off0 < 0) {  // thread 0 of the first workgroup of an odd slice: row -1 is masked, row 0 is (ci, cj)
      cj -= 1;       // so that advancing by one lands on the first real pair
    }
  }

  // Interior workgroups (all PAIR_ROWS rows inside the slice: all but the first and last of a time step) take the
  // FULL instantiation, which carries no per-row validity masks, index clamps or scalar-store fallbacks.
  const bool full = c0 >= 0 && c0 + ROWS <= nq;
  double* const eta_k = a.eta + slice0;  // wave-uniform bases + 32-bit lane offsets: saddr-form global accesses
  double* const l_k = a.l + slice0;
  const int o0 = (int)off0;
  auto body = [&](auto full_tag) {
  constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll 1
  for (int s0 = 0; s0 < nsteps; s0 += PAIR_UNROLL) {
    double eta_v[PAIR_UNROLL][2][D];
    double l_v[PAIR_UNROLL][2];
    bool valid[PAIR_UNROLL][2];
    bool sel[PAIR_UNROLL][2];
    int pi_[PAIR_UNROLL][2], pj_[PAIR_UNROLL][2];
    double ein[PAIR_UNROLL][2][D], lin_[PAIR_UNROLL][2];

    // indices of the 2*PAIR_UNROLL rows of this group
#pragma unroll
    for (int u = 0; u < PAIR_UNROLL; ++u) {
      const int64_t off = off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS);
      valid[u][0] = FULL || ((off >= 0) && (off < nq));
      valid[u][1] = FULL || ((off + 1 >= 0) && (off + 1 < nq));
      pi_[u][0] = ci;
      pj_[u][0] = cj;
      int i1, j1;
      pair_next(ci, cj, N, i1, j1);
      pi_[u][1] = i1;
      pj_[u][1] = j1;
      pair_advance_far(ci, cj, N, 2 * PAIR_THREADS, a.q_begin + off + 2 * PAIR_THREADS, a.pairs);
      if (!FULL) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (!valid[u][e] || pi_[u][e] >= N - 1 || pj_[u][e] >= N || pj_[u][e] <= pi_[u][e]) {
            valid[u][e] = false;
            pi_[u][e] = 0;
            pj_[u][e] = N > 1 ? 1 : 0;
          }
      }
    }
    uint32_t marked[PAIR_UNROLL];
    if (MODE == MODE_VIOL_RECOMPUTE && !(FULL && !SMALL)) {  // (interior workgroups of large problems: read only when a row is violated)
#pragma unroll
      for (int u = 0; u < PAIR_UNROLL; ++u) {
        const int64_t lrA = slice0 + off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS);
        marked[u] = (valid[u][0] || valid[u][1]) ? (a.bitmap[(lrA + (valid[u][0] ? 0 : 1)) >> 5] >> (lrA & 31)) & 3u : 0u;
      }
    }
    if (MODE == MODE_VIOLATIONS) {  // streaming reads of the compact rows, all issued before use
#pragma unroll
      for (int u = 0; u < PAIR_UNROLL; ++u) {
        const int64_t lrA = slice0 + off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS);
        const int o32 = o0 + (s0 + u) * (2 * PAIR_THREADS);
        // both rows of the step share one bitmap word (lrA is even); rows already in the working set stay out
        marked[u] = (valid[u][0] || valid[u][1]) ? (a.bitmap[(lrA + (valid[u][0] ? 0 : 1)) >> 5] >> (lrA & 31)) & 3u : 0u;
        if (FULL || (valid[u][0] && valid[u][1])) {
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const double2 t = *reinterpret_cast<const double2*>(eta_k + d * a.eta_stride + o32);
            ein[u][0][d] = t.x;
            ein[u][1][d] = t.y;
          }
          const double2 t = *reinterpret_cast<const double2*>(l_k + o32);
          lin_[u][0] = t.x;
          lin_[u][1] = t.y;
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int d = 0; d < D; ++d) ein[u][e][d] = valid[u][e] ? a.eta[d * a.eta_stride + lrA + e] : 0.0;
            lin_[u][e] = valid[u][e] ? a.l[lrA + e] : 0.0;
          }
        }
      }
    }

    bool any_sel = false;
    // STREAM (interior workgroups of the linearisation): the rare events of a row -- a degenerate pair (scp.py:503), a
    // distance below R - 0.01 (scp.py:610) -- are only FLAGGED in the row loop and handled behind one branch per group
    // afterwards; the loop itself runs pair_row_regular (no selects).  Same values: the repair calls pair_row.
    constexpr bool STREAM = (MODE == MODE_LINEARIZE || ((MODE == MODE_SELECT || MODE == MODE_CHECK) && !SMALL)) && FULL;
    // STREAMV: the same for the recomputing violations pass -- l - A x by the regular formulas, the group's largest value
    // decides whether any row can be violated (only then the membership bits are read), a possible degenerate pair sends
    // the whole group through the exact formulas again
    constexpr bool STREAMV = MODE == MODE_VIOL_RECOMPUTE && !SMALL && FULL;
    double viol_v[PAIR_UNROLL][2];
    double viol_max = -INF;
    double raw_v[PAIR_UNROLL][2];
    double raw_min = INF;  // over the group's rows: ONE comparison per group and event decides whether any row needs a closer look
    // UNIFORM i (slices read through L1 / L2, i.e. more than 1024 agents): a wave's 128 consecutive pairs of a step mostly
    // lie in ONE row of the triangle (rows are ~N / 2 long), so agent i's points are the same for all 64 lanes: they come
    // through the SCALAR cache into SGPRs (one s_load per point and step instead of 64 lanes x 16 B through the vector L1,
    // which is what bounds these passes: 48 of its 64 B / clk / CU in the violations pass at 4096 agents).  Pairs are
    // consecutive, so "lane 63's second row has lane 0's i" is the whole test; other waves take the per-lane loads.
    constexpr bool UNI_OK = !USE_LDS && !SMALL && MODE != MODE_VIOLATIONS;
    int iu[PAIR_UNROLL];
    bool uni = UNI_OK;
    if (UNI_OK) {
#pragma unroll
      for (int u = 0; u < PAIR_UNROLL; ++u) {
        iu[u] = __builtin_amdgcn_readfirstlane(pi_[u][0]);
        uni = uni && __builtin_amdgcn_readlane(pi_[u][1], 63) == iu[u];
      }
    }
    auto row_loop = [&](auto uni_tag) {
    constexpr bool UNI = decltype(uni_tag)::value;
    auto load_i = [&](const double* base, int i, int u) -> Pt<D> {
      if constexpr (UNI) return load_pt_uniform<D>(base, iu[u]);
      else return load_pt<D>(base, i);
    };
#pragma unroll
    for (int u = 0; u < PAIR_UNROLL; ++u) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = pi_[u][e], j = pj_[u][e];
        sel[u][e] = false;
        if (STREAM) {
          const Pt<D> Pi = load_i(P, i, u), Pj = load_pt<D>(P, j);
          const PairGeom<D> g = pair_geom<D>(Pi, Pj);
          raw_v[u][e] = g.raw;
          raw_min = fmin(raw_min, g.raw);
          if (MODE == MODE_LINEARIZE) {
            const Pt<D> Qi = load_i(Q, i, u), Qj = load_pt<D>(Q, j);
            pair_row_regular<D>(g, Qi, Qj, a.R, eta_v[u][e], l_v[u][e]);
          }
        } else if (STREAMV) {
          const Pt<D> Pi = load_i(P, i, u), Pj = load_pt<D>(P, j);
          const PairGeom<D> g = pair_geom<D>(Pi, Pj);
          raw_min = fmin(raw_min, g.raw);
          const Pt<D> Di = load_i(Q, i, u), Dj = load_pt<D>(Q, j);  // dP = P_new - P_prev
          double qd = 0.0;
#pragma unroll
          for (int d = 0; d < D; ++d) qd = fma(g.diff[d] * g.inv, Di.v[d] - Dj.v[d], qd);
          viol_v[u][e] = (a.R - g.raw) - qd;  // l_r - (A x)_r of a regular pair
          viol_max = fmax(viol_max, viol_v[u][e]);
        } else if (MODE != MODE_VIOLATIONS) {
          const Pt<D> Pi = load_i(P, i, u), Pj = load_pt<D>(P, j);
          const PairGeom<D> g = pair_geom<D>(Pi, Pj);
          const double* diff = g.diff;
          const bool deg = g.deg;
          const double inv = g.inv, raw = g.raw;
          if (MODE != MODE_VIOL_RECOMPUTE && valid[u][e]) {
            my_min = fmin(my_min, raw);
            if (raw < thr) {
              const int64_t off = off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS) + e;
              const unsigned long long g_ = (unsigned long long)((int64_t)k * a.pairs + a.q_begin + off);
              my_first = g_ < my_first ? g_ : my_first;
            }
          }
          if (MODE == MODE_LINEARIZE) {
            const Pt<D> Qi = load_i(Q, i, u), Qj = load_pt<D>(Q, j);
            double dist;
            pair_row<D>(g, Qi, Qj, a.R, eta_v[u][e], l_v[u][e], dist);
            sel[u][e] = valid[u][e] && ((dist - a.R) < a.margin);
          }
          if (MODE == MODE_SELECT) {  // the selection test of MODE_LINEARIZE on the same distance, nothing stored
            const double dist = deg ? 1.0 : raw;
            sel[u][e] = valid[u][e] && ((dist - a.R) < a.margin);
          }
          if (MODE == MODE_VIOL_RECOMPUTE) {
            const Pt<D> Di = load_i(Q, i, u), Dj = load_pt<D>(Q, j);  // dP = P_new - P_prev
            const double dist = deg ? 1.0 : raw;
            double qd = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
              const double e_d = deg ? (d == 0 ? 1.0 : 0.0) : diff[d] * inv;
              qd = fma(e_d, Di.v[d] - Dj.v[d], qd);
            }
            const double viol = (a.R - dist) - qd;  // l_r - (A x)_r
            if (valid[u][e]) my_maxv = fmax(my_maxv, viol);
            sel[u][e] = valid[u][e] && (viol > a.margin) && !((marked[u] >> e) & 1u);
            if (SMALL && Pn) {  // the selection test of MODE_SELECT around the new positions (the next linearisation point)
              const PairGeom<D> g2 = pair_geom<D>(load_pt<D>(Pn, i), load_pt<D>(Pn, j));
              const double dist2 = g2.deg ? 1.0 : g2.raw;
              if (valid[u][e] && ((dist2 - a.R) < a.spec_margin)) {
                const int o = 2 * (int)threadIdx.x + (s0 + u) * (2 * PAIR_THREADS) + e;
                atomicOr(small_map2 + (o >> 5), 1u << (o & 31));
              }
            }
          }
        } else {
          const Pt<D> Qi = load_pt<D>(Q, i), Qj = load_pt<D>(Q, j);
          double ax = 0.0;
#pragma unroll
          for (int d = 0; d < D; ++d) ax = fma(ein[u][e][d], Qi.v[d] - Qj.v[d], ax);
          const double viol = lin_[u][e] - ax;
          if (valid[u][e]) my_maxv = fmax(my_maxv, viol);
          sel[u][e] = valid[u][e] && (viol > a.margin) && !((marked[u] >> e) & 1u);
        }
        any_sel |= sel[u][e];
      }
    }
    };  // row_loop
    if (UNI_OK && uni) row_loop(std::true_type{});
    else row_loop(std::false_type{});
    if (STREAM) {
      my_min = fmin(my_min, raw_min);
      // (x -> x - R is monotone, so "some row passes the selection test" implies "the smallest distance passes it")
      if (MODE != MODE_CHECK && (raw_min - a.R) < a.margin) {
#pragma unroll
        for (int u = 0; u < PAIR_UNROLL; ++u)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            sel[u][e] = (raw_v[u][e] - a.R) < a.margin;
            any_sel |= sel[u][e];
          }
      }
      if (raw_min < thr) {
#pragma unroll
        for (int u = 0; u < PAIR_UNROLL; ++u)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            if (raw_v[u][e] < thr) {
              const int64_t off = off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS) + e;
              const unsigned long long g_ = (unsigned long long)((int64_t)k * a.pairs + a.q_begin + off);
              my_first = g_ < my_first ? g_ : my_first;
            }
      }
      if (MODE != MODE_CHECK && raw_min < 2e-6) {  // (a degenerate pair has dist^2 < 1e-12: the exact test is g.deg below)
        any_sel = false;
#pragma unroll
        for (int u = 0; u < PAIR_UNROLL; ++u)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = pi_[u][e], j = pj_[u][e];
            const PairGeom<D> g = pair_geom<D>(load_pt<D>(P, i), load_pt<D>(P, j));
            if (g.deg) {
              double dist = 1.0;  // scp.py:503-507
              if (MODE == MODE_LINEARIZE)
                pair_row<D>(g, load_pt<D>(Q, i), load_pt<D>(Q, j), a.R, eta_v[u][e], l_v[u][e], dist);
              sel[u][e] = (dist - a.R) < a.margin;
            }
            any_sel |= sel[u][e];
          }
      }
    }
    if (STREAMV) {
      if (raw_min < 2e-6) {  // a degenerate pair somewhere in the group: every row again, by the formulas that know about it
        viol_max = -INF;
#pragma unroll
        for (int u = 0; u < PAIR_UNROLL; ++u)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = pi_[u][e], j = pj_[u][e];
            const PairGeom<D> g = pair_geom<D>(load_pt<D>(P, i), load_pt<D>(P, j));
            const Pt<D> Di = load_pt<D>(Q, i), Dj = load_pt<D>(Q, j);
            const double dist = g.deg ? 1.0 : g.raw;
            double qd = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
              const double e_d = g.deg ? (d == 0 ? 1.0 : 0.0) : g.diff[d] * g.inv;
              qd = fma(e_d, Di.v[d] - Dj.v[d], qd);
            }
            viol_v[u][e] = (a.R - dist) - qd;
            viol_max = fmax(viol_max, viol_v[u][e]);
          }
      }
      my_maxv = fmax(my_maxv, viol_max);
      if (viol_max > a.margin) {  // some row of the group is violated: which ones, and are they in the working set already?
#pragma unroll
        for (int u = 0; u < PAIR_UNROLL; ++u) {
          const int64_t lrA = slice0 + off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS);
          const uint32_t mk = (a.bitmap[lrA >> 5] >> (lrA & 31)) & 3u;  // (both rows of the step share one word: lrA is even)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            sel[u][e] = (viol_v[u][e] > a.margin) && !((mk >> e) & 1u);
            any_sel |= sel[u][e];
          }
        }
      }
    }

    if (MODE == MODE_LINEARIZE && (a.ablate & 1)) {
#pragma unroll
      for (int u = 0; u < PAIR_UNROLL; ++u)
#pragma unroll
        for (int e = 0; e < 2; ++e) my_maxv = fmax(my_maxv, l_v[u][e] + eta_v[u][e][0] + eta_v[u][e][D - 1]);
    }
    if (MODE == MODE_LINEARIZE && !(a.ablate & 1)) {
#pragma unroll
      for (int u = 0; u < PAIR_UNROLL; ++u) {
        const int64_t lrA = slice0 + off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS);  // even by construction
        const int o32 = o0 + (s0 + u) * (2 * PAIR_THREADS);
        if (FULL) {
          // wave-uniform plane bases + an UNSIGNED 32-bit byte offset per lane (host: 8 nq < 2^32): the saddr form of the
          // store, one vector add per step instead of a 64-bit address per plane
          const uint32_t bo = (uint32_t)o32 * 8u;
#pragma unroll
          for (int d = 0; d < D; ++d)
            *reinterpret_cast<double2*>(reinterpret_cast<char*>(eta_k + d * a.eta_stride) + bo) =
                make_double2(eta_v[u][0][d], eta_v[u][1][d]);
          *reinterpret_cast<double2*>(reinterpret_cast<char*>(l_k) + bo) = make_double2(l_v[u][0], l_v[u][1]);
        } else if (valid[u][0] && valid[u][1]) {
#pragma unroll
          for (int d = 0; d < D; ++d)
            *reinterpret_cast<double2*>(eta_k + d * a.eta_stride + o32) = make_double2(eta_v[u][0][d], eta_v[u][1][d]);
          *reinterpret_cast<double2*>(l_k + o32) = make_double2(l_v[u][0], l_v[u][1]);
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            if (valid[u][e]) {
#pragma unroll
              for (int d = 0; d < D; ++d) a.eta[d * a.eta_stride + lrA + e] = eta_v[u][e][d];
              a.l[lrA + e] = l_v[u][e];
            }
        }
      }
    }
    if (MODE != MODE_CHECK && any_sel) {
      // Selected rows (a few in 10^4) only set their bit -- no-return atomics on distinct words, no shared
      // counter: a returning same-address atomic per wave cost 3x the whole streaming pass.  The row list is
      // produced afterwards from the bitmap by the compaction kernels below (sorted, deterministic).
#pragma unroll
      for (int u = 0; u < PAIR_UNROLL; ++u) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (sel[u][e]) {
            if (SMALL) {  // (this workgroup's own SMALL_ROWS-bit map in LDS: compacted below, no global bitmap)
              const int o = 2 * (int)threadIdx.x + (s0 + u) * (2 * PAIR_THREADS) + e;
              atomicOr(small_map + (o >> 5), 1u << (o & 31));
            } else {
              const int64_t lr = slice0 + off0 + (int64_t)(s0 + u) * (2 * PAIR_THREADS) + e;
              atomicOr(a.mark + (lr >> 5), 1u << (lr & 31));
            }
          }
        }
      }
    }
  }
  };
  if (full) body(std::true_type{});
  else body(std::false_type{});
  };  // run_item

  if constexpr (ITEMS) {  // one item per workgroup; the hardware dispatcher hands them out in order (large ones first)
    const int item = blockIdx.x;
    int c = 0;
#pragma unroll
    for (int t = 1; t < PairItems::MAX_CLASSES; ++t)
      if (t < a.items.n_classes && item >= a.items.item0[t]) c = t;
    const int j = item - a.items.item0[c];
    const int kk = j / a.items.cpk[c];
    run_item(a.items.k0[c] + kk, (int64_t)(j - kk * a.items.cpk[c]), a.items.steps[c]);
  } else {
    run_item((int)blockIdx.y, (int64_t)blockIdx.x, STEPS);
  }

  // wavefront reductions, then one candidate per WORKGROUP; the global atomic is issued only when a relaxed
  // (L1-bypassing) read says the candidate would improve the result: same-address atomics retire at < 100 per
  // microsecond chip-wide, 25 600 of them (one per wave) would cost more than the whole streaming pass.
  __shared__ double red_d[PAIR_THREADS / 64];
  __shared__ unsigned long long red_u[PAIR_THREADS / 64];
  if (!VIOL) {
    my_min = wave_min(my_min);
    my_first = wave_min_u64(my_first);
  } else {
    my_maxv = wave_max(my_maxv);
  }
  if ((threadIdx.x & 63) == 63) {
    red_d[threadIdx.x >> 6] = !VIOL ? my_min : my_maxv;
    red_u[threadIdx.x >> 6] = my_first;
  }
  __syncthreads();
  if constexpr (SMALL) {
    SMALL_STAMP(2);
    __shared__ int last_sh;
    __shared__ int woff[SCP_SMALL_MAX_WG + 1];
    const unsigned n_wg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
    constexpr bool SPEC = MODE == MODE_VIOL_RECOMPUTE;  // may carry a second, speculative selection (a.spec_rows)
    const bool spec = SPEC && a.spec_rows != nullptr;
    // (a) this workgroup's marks -> its sorted sub-list(s) (offsets within the workgroup's SMALL_ROWS rows)
    int cnt_wg = 0, cnt2_wg = 0;
    if (MODE != MODE_CHECK) {
      const uint32_t w = threadIdx.x < SMALL_ROWS / 32 ? small_map[threadIdx.x] : 0u;
      int ex = block_exclusive_scan(__popc(w), &cnt_wg);
      uint32_t* sub = a.wg_rows + (size_t)wg * SMALL_ROWS;
      for (uint32_t m_ = w; m_; m_ &= m_ - 1)
        __hip_atomic_store(sub + ex++, 32u * threadIdx.x + (uint32_t)(__ffs((int)m_) - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (spec) {
        const uint32_t w2 = threadIdx.x < SMALL_ROWS / 32 ? small_map2[threadIdx.x] : 0u;
        int ex2 = block_exclusive_scan(__popc(w2), &cnt2_wg);
        uint32_t* sub2 = a.wg_rows + ((size_t)n_wg + wg) * SMALL_ROWS;
        for (uint32_t m_ = w2; m_; m_ &= m_ - 1)
          __hip_atomic_store(sub2 + ex2++, 32u * threadIdx.x + (uint32_t)(__ffs((int)m_) - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (threadIdx.x == 0) {
      double m = red_d[0];
      unsigned long long f = red_u[0];
#pragma unroll
      for (int w = 1; w < PAIR_THREADS / 64; ++w) {
        m = VIOL ? fmax(m, red_d[w]) : fmin(m, red_d[w]);
        f = red_u[w] < f ? red_u[w] : f;
      }
      __hip_atomic_store(a.wg_part + 4 * wg, (unsigned long long)__double_as_longlong(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.wg_part + 4 * wg + 1, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.wg_part + 4 * wg + 2, (unsigned long long)cnt_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.wg_part + 4 * wg + 3, (unsigned long long)cnt2_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // a bitmap that is REPLACED (a new working set) is cleared by all workgroups together -- with atomics, the path of the ORs
    // that rebuild it in the tail, and performed before this workgroup's ticket
    if (MODE != MODE_CHECK && a.merge_into && a.overwrite)
      for (int64_t w = (int64_t)wg * PAIR_THREADS + threadIdx.x; w < a.words; w += (int64_t)n_wg * PAIR_THREADS) atomicAnd(a.merge_into + w, 0u);
    if (spec)
      for (int64_t w = (int64_t)wg * PAIR_THREADS + threadIdx.x; w < a.words; w += (int64_t)n_wg * PAIR_THREADS) atomicAnd(a.spec_bitmap + w, 0u);
    SMALL_STAMP(3);
    wait_stores_performed();  // this thread's sub-list entries, bitmap words, positions (all written through) are in place ...
    __syncthreads();
    if (threadIdx.x == 0) last_sh = atomicAdd(a.ticket, 1u) == n_wg - 1 ? 1 : 0;  // ... before the ticket is taken
    __syncthreads();
    if (!last_sh) return;
    SMALL_STAMP(4);
    SMALL_STAMP(5);  // (no acquire fence: everything below that other workgroups wrote is read with load_coherent / atomics)
    // ---- (b) tail: the last workgroup alone.  Work ~ workgroups + marked rows, not ~ bitmap words ----
    const int G = (int)((n_wg + PAIR_THREADS - 1) / PAIR_THREADS);  // consecutive workgroups per thread (<= 8)
    double m = VIOL ? -INF : INF;
    unsigned long long f = 0xFFFFFFFFFFFFFFFFULL;
    for (int e = 0; e < G; ++e) {
      const unsigned w = threadIdx.x * G + e;
      if (w < n_wg) {
        const double pm = __longlong_as_double((long long)__hip_atomic_load(a.wg_part + 4 * w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const unsigned long long pf = __hip_atomic_load(a.wg_part + 4 * w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m = VIOL ? fmax(m, pm) : fmin(m, pm);
        f = pf < f ? pf : f;
      }
    }
    m = VIOL ? wave_max(m) : wave_min(m);
    f = wave_min_u64(f);
    if ((threadIdx.x & 63) == 63) {  // (red_d / red_u of this workgroup's own reduction were read before the ticket barrier)
      red_d[threadIdx.x >> 6] = m;
      red_u[threadIdx.x >> 6] = f;
    }
    // one list: the sub-lists `which` of all workgroups, concatenated in workgroup (= row) order -> rows_out; their bits are
    // merged into (all or nothing: a list that is too short is repeated) or replace (overwrite) the bitmap bm
    auto emit = [&](int which, int64_t* rows_out, int64_t cap_out, uint32_t* bm, bool overwrite_) -> int {
      int cnt[SCP_SMALL_MAX_WG / PAIR_THREADS], mine = 0;
#pragma unroll
      for (int e = 0; e < SCP_SMALL_MAX_WG / PAIR_THREADS; ++e) {
        const unsigned w = threadIdx.x * G + e;
        cnt[e] = (e < G && w < n_wg) ? (int)__hip_atomic_load(a.wg_part + 4 * w + 2 + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        mine += cnt[e];
      }
      int total = 0;
      int run = block_exclusive_scan(mine, &total);  // (its barriers also order woff of an earlier list and red_d / red_u)
#pragma unroll
      for (int e = 0; e < SCP_SMALL_MAX_WG / PAIR_THREADS; ++e) {
        const unsigned w = threadIdx.x * G + e;
        if (e < G && w < n_wg) woff[w] = run;
        run += cnt[e];
      }
      if (threadIdx.x == 0) woff[n_wg] = total;
      __syncthreads();
      const bool overflow = bm != nullptr && !overwrite_ && (int64_t)total > cap_out;  // (a merging pass merges all or nothing)
      constexpr int EB = 8;  // slots per thread and pass: their sub-list loads (past the L2: ~1 us each) are in flight together
      for (int s0 = threadIdx.x; s0 < total; s0 += EB * PAIR_THREADS) {
        int lo_[EB];
        uint32_t o_[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const int sl = s0 + e * PAIR_THREADS;
          int lo = 0, hi = (int)n_wg;  // the workgroup whose sub-list holds slot sl: the last g with woff[g] <= sl
          if (sl < total) {
            while (hi - lo > 1) {
              const int mid = (lo + hi) >> 1;
              if (woff[mid] <= sl) lo = mid;
              else hi = mid;
            }
            o_[e] = __hip_atomic_load(a.wg_rows + ((size_t)which * n_wg + lo) * SMALL_ROWS + (sl - woff[lo]), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
          }
          lo_[e] = lo;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const int sl = s0 + e * PAIR_THREADS;
          if (sl >= total) continue;
          const int64_t kg = lo_[e] / (int)gridDim.x, xg = lo_[e] % (int)gridDim.x;
          const int64_t q_loc = xg * SMALL_ROWS - ((kg * nq) & 1) + (int64_t)o_[e];  // pair offset within [q_begin, q_end)
          if (!overflow && sl < cap_out) rows_out[sl] = kg * a.pairs + a.q_begin + q_loc;
          if (bm && !overflow) {
            const int64_t lr = kg * nq + q_loc;
            atomicOr(bm + (lr >> 5), 1u << (lr & 31));
          }
        }
      }
      __syncthreads();  // (woff is free again)
      return total;
    };
    int total = 0, total2 = 0;
    SMALL_STAMP(6);
    if (MODE != MODE_CHECK) total = emit(0, a.rows, a.cap, a.merge_into, a.overwrite != 0);
    else __syncthreads();
    SMALL_STAMP(7);
    if (spec) total2 = emit(1, a.spec_rows, a.spec_cap, a.spec_bitmap, true);
    SMALL_STAMP(8);
    if (SPEC && a.mirror && threadIdx.x == 0)
      __hip_atomic_store((unsigned long long*)&a.mirror->n_spec, (unsigned long long)(spec ? total2 : 0), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == MODE_VIOL_RECOMPUTE && a.rel_prev && a.mirror) {
      // rel_step_partial_kernel, block after block: thread t of "block" b sums the elements t + 256 (b + j blocks), the wave
      // butterfly and the (w0 + w1) + (w2 + w3) combination are that kernel's -- the same sums, bit for bit.  No barrier
      // between the blocks: their loads overlap.
      __shared__ double rs0[32][PAIR_THREADS / 64], rs1[32][PAIR_THREADS / 64];
      const int64_t n = (int64_t)N * a.K * D, C = (int64_t)N * D;
      constexpr int RB = 8;  // blocks whose loads are in flight together
      for (int b0 = 0; b0 < a.rel_blocks; b0 += RB) {
        double d2[RB], b2[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) d2[u] = b2[u] = 0.0;
        // x_out[t] = x_tm[k][c], read from the QP's own (time-major) array: written by an EARLIER kernel, so ordinary cached
        // loads do.  RB blocks x RQ grid-stride steps = 32 + 32 loads are issued before the first sum needs one (one load per
        // sum is a chain of ~50 memory latencies per thread: 18 us).
        constexpr int RQ = 4;
        const unsigned stride = (unsigned)a.rel_blocks * 256u, un = (unsigned)n;  // (n < 2^31 for a small problem)
        // (i, k, d) of every block's current element, advanced by the stride without divisions (a 32-bit division is ~40
        // instructions; two per element were 7 us of this tail)
        const unsigned s_d = stride % (unsigned)D, s_td = stride / (unsigned)D;
        const unsigned s_k = s_td % (unsigned)a.K, s_i = s_td / (unsigned)a.K;
        unsigned ei[RB], ek[RB], ed[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const unsigned t = (unsigned)(b0 + u) * 256u + threadIdx.x;
          const unsigned td = t / (unsigned)D;
          ed[u] = t - td * (unsigned)D;
          ei[u] = td / (unsigned)a.K;
          ek[u] = td - ei[u] * (unsigned)a.K;
        }
        for (unsigned j0 = 0; (unsigned)b0 * 256u + j0 * stride < un; j0 += RQ) {
          double xv[RB][RQ], yv[RB][RQ];
#pragma unroll
          for (int u = 0; u < RB; ++u) {
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
              const unsigned t = (unsigned)(b0 + u) * 256u + threadIdx.x + (j0 + q) * stride;
              const bool ok = b0 + u < a.rel_blocks && t < un;
              xv[u][q] = a.x_tm[ok ? (int64_t)ek[u] * C + (int64_t)ei[u] * D + ed[u] : 0];
              yv[u][q] = a.rel_prev[ok ? t : 0u];
              ed[u] += s_d;
              if (ed[u] >= (unsigned)D) { ed[u] -= (unsigned)D; ++ek[u]; }
              ek[u] += s_k;
              if (ek[u] >= (unsigned)a.K) { ek[u] -= (unsigned)a.K; ++ei[u]; }
              if (ek[u] >= (unsigned)a.K) { ek[u] -= (unsigned)a.K; ++ei[u]; }  // (the carry from d on top of s_k)
              ei[u] += s_i;
            }
          }
#pragma unroll
          for (int u = 0; u < RB; ++u) {
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
              const unsigned t = (unsigned)(b0 + u) * 256u + threadIdx.x + (j0 + q) * stride;
              if (b0 + u < a.rel_blocks && t < un) rel_accum(xv[u][q], yv[u][q], d2[u], b2[u]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            d2[u] += __shfl_xor(d2[u], o);
            b2[u] += __shfl_xor(b2[u], o);
          }
          if (b0 + u < a.rel_blocks && (threadIdx.x & 63) == 0) {
            rs0[b0 + u][threadIdx.x >> 6] = d2[u];
            rs1[b0 + u][threadIdx.x >> 6] = b2[u];
          }
        }
      }
      __syncthreads();
      if ((int)threadIdx.x < a.rel_blocks) {
        const int b = threadIdx.x;
        __hip_atomic_store((unsigned long long*)&a.mirror->rel[2 * b],
                           (unsigned long long)__double_as_longlong((rs0[b][0] + rs0[b][1]) + (rs0[b][2] + rs0[b][3])),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((unsigned long long*)&a.mirror->rel[2 * b + 1],
                           (unsigned long long)__double_as_longlong((rs1[b][0] + rs1[b][1]) + (rs1[b][2] + rs1[b][3])),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        wait_stores_performed();  // (written through to host memory before thread 0 stores the sequence number: a barrier follows)
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      m = red_d[0];
      f = red_u[0];
#pragma unroll
      for (int w = 1; w < PAIR_THREADS / 64; ++w) {
        m = VIOL ? fmax(m, red_d[w]) : fmin(m, red_d[w]);
        f = red_u[w] < f ? red_u[w] : f;
      }
#ifdef SCP_PHASE_PROFILE
      small_clk_local[9] = __builtin_amdgcn_s_memrealtime();
#endif
      scp_pair_stats st;
      st.min_dist = VIOL ? INF : m;
      st.first_violation = f;
      st.n_selected = (unsigned long long)total;
      st.max_violation = VIOL ? m : -INF;
      *a.stats = st;
      *a.ticket = 0u;  // (the next launch on this stream starts after this kernel has ended)
      if (a.mirror && MODE == MODE_SELECT && a.x_tm && f != 0xFFFFFFFFFFFFFFFFULL) {
        // the two positions of the first violating pair (scp.py:611-613 prints their distance): written by other workgroups
        // of this kernel, hence read past this XCD's L2
        const int64_t kf = (int64_t)(f / (unsigned long long)a.pairs);
        int fi, fj;
        decode_pair((int64_t)(f % (unsigned long long)a.pairs), N, fi, fj);
        for (int d = 0; d < D; ++d) {
          const unsigned long long vi = __hip_atomic_load((const unsigned long long*)(a.pos_out + ((int64_t)fi * a.K + kf) * D + d),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long vj = __hip_atomic_load((const unsigned long long*)(a.pos_out + ((int64_t)fj * a.K + kf) * D + d),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store((unsigned long long*)&a.mirror->pts[d], vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store((unsigned long long*)&a.mirror->pts[3 + d], vj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if (a.mirror) {
        __hip_atomic_store((unsigned long long*)&a.mirror->stats.min_dist, (unsigned long long)__double_as_longlong(st.min_dist),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((unsigned long long*)&a.mirror->stats.first_violation, st.first_violation, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((unsigned long long*)&a.mirror->stats.n_selected, st.n_selected, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((unsigned long long*)&a.mirror->stats.max_violation,
                           (unsigned long long)__double_as_longlong(st.max_violation), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // every field above was written through to host memory: waiting for the acknowledgements orders the sequence number
        // behind them without a system-scope release fence (an L2 write-back: ~10 us in this tail)
        wait_stores_performed();
        __hip_atomic_store((unsigned long long*)&a.mirror->seq, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef SCP_PHASE_PROFILE
        small_clk_local[10] = __builtin_amdgcn_s_memrealtime();
        for (int i_ = 0; i_ < 11; ++i_) scp_small_clk[i_] = small_clk_local[i_];
#endif
      }
    }
    return;
  }
  if (threadIdx.x == 0) {
    if (!VIOL) {
      double m = red_d[0];
      unsigned long long f = red_u[0];
#pragma unroll
      for (int w = 1; w < PAIR_THREADS / 64; ++w) {
        m = fmin(m, red_d[w]);
        f = red_u[w] < f ? red_u[w] : f;
      }
      const double cur = __hip_atomic_load(&a.stats->min_dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (m < cur) atomic_min_pos_double(&a.stats->min_dist, m);
      if ((a.ablate & 1) && my_maxv == 12345.678) a.stats->max_violation = my_maxv;  // keeps the ablated values live
      if (f != 0xFFFFFFFFFFFFFFFFULL) {
        const unsigned long long curf = __hip_atomic_load((unsigned long long*)&a.stats->first_violation,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f < curf) atomicMin((unsigned long long*)&a.stats->first_violation, f);
      }
    } else {
      double m = red_d[0];
#pragma unroll
      for (int w = 1; w < PAIR_THREADS / 64; ++w) m = fmax(m, red_d[w]);
      const double cur = __hip_atomic_load(&a.stats->max_violation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (m > cur) atomic_max_double(&a.stats->max_violation, m);
    }
  }
#ifdef SCP_PHASE_PROFILE
  if (threadIdx.x == 0) {
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < (unsigned)SCP_PAIR_CLK_WGS) {
      scp_pair_clk[2 * wg] = __builtin_amdgcn_s_memtime() - clk_c0;
      scp_pair_clk[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - clk_t0;
      scp_pair_t0[wg] = clk_t0;
    }
  }
#endif
}

// scratch for the time-major slices (grown on demand, owned by the ctx)
static int ensure_tm(scp_ctx* ctx, size_t bytes) {
  if (ctx->tm_bytes >= bytes) return SCP_OK;
  if (ctx->tm_scratch) {
    SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    SCP_HIP_CHECK(ctx, hipFree(ctx->tm_scratch));
    ctx->tm_scratch = nullptr;
    ctx->tm_bytes = 0;
  }
  SCP_HIP_CHECK(ctx, hipMalloc(&ctx->tm_scratch, bytes));
  ctx->tm_bytes = bytes;
  return SCP_OK;
}

// [N][K][D] -> time-major P_prev and dP = P_new - P_prev (MODE_VIOL_RECOMPUTE)
__global__ __launch_bounds__(256) void pair_prep_delta_kernel(int N, int K, int D, const double* __restrict__ pos_prev,
                                                               const double* __restrict__ pos_new,
                                                               double* __restrict__ P_tm, double* __restrict__ dP_tm,
                                                               scp_pair_stats* __restrict__ stats) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t C = (int64_t)N * D;
  if (t == 0) pair_stats_init(stats);
  if (t >= C * K) return;
  const int k = (int)(t / C);
  const int c = (int)(t % C);
  const int i = c / D, d = c % D;
  const int64_t g = ((int64_t)i * K + k) * D + d;
  const double pp = pos_prev[g];
  P_tm[t] = pp;
  dP_tm[t] = pos_new[g] - pp;
}

// What follows a pass: its marks become the sorted row list `rows` and are merged into (violations) / replace (select) the
// working-set bitmap `merge_into`.  Small problems run it as the tail of the pass kernel itself (*done = true), the others
// in launch_compaction afterwards.
struct PassTail {
  int64_t* rows;
  int64_t cap;
  uint32_t* merge_into;
  bool overwrite;
  int64_t words;
  bool done;
};

// a pass over nq pairs x K steps that runs as ONE launch (pair_pass_kernel<.., SMALL>): its bitmap fits the one-workgroup
// compaction, its time-step slices (n_slices of them) the LDS budget of the pass, its partials the ctx scratch
static bool small_pass_ok(const scp_ctx* ctx, int N, int K, int D, int64_t nq, int n_slices) {
  if (!ctx->small_pass || nq <= 0) return false;
  const int64_t words = (K * nq + 31) / 32;
  const int64_t n_wg = (int64_t)scp_cdiv(nq + 1, SMALL_ROWS) * K;
  return words <= CMP1_MAX_WORDS && (size_t)n_slices * N * D * sizeof(double) <= 32 * 1024 && n_wg <= SCP_SMALL_MAX_WG;
}

// MODE_VIOL_RECOMPUTE: pos_ref_layout = the linearisation point, p0 = the new positions (v0 unused)
template <int MODE>
static int launch_pair_pass(scp_ctx* ctx, PairArgs& a, const double* pos_ref_layout, const double* p0,
                            const double* v0, uint32_t* clear_map = nullptr, int64_t clear_words = 0,
                            PassTail* tail = nullptr) {
  const int N = a.N, K = a.K, D = a.D;
  const int64_t nq = a.q_end - a.q_begin;
  if (tail) tail->done = false;
  if constexpr (MODE == MODE_SELECT || MODE == MODE_VIOL_RECOMPUTE || MODE == MODE_CHECK)
  if (tail) {
    const int n_slices = MODE == MODE_VIOL_RECOMPUTE ? (a.spec_rows ? 3 : 2) : 1;
    const size_t lds_small = (size_t)n_slices * N * D * sizeof(double);
    if (small_pass_ok(ctx, N, K, D, nq, n_slices)) {
      a.pos_a = pos_ref_layout;
      if (MODE == MODE_VIOL_RECOMPUTE && !a.x_tm) a.pos_b = p0;
      if (MODE != MODE_CHECK) {  // per-workgroup sub-lists of the marked rows (grown on demand, owned by the ctx)
        const size_t need = (size_t)(a.spec_rows ? 2 : 1) * scp_cdiv(nq + 1, SMALL_ROWS) * K * SMALL_ROWS * sizeof(uint32_t);
        if (ctx->wg_rows_bytes < need) {
          if (ctx->wg_rows) {
            SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            SCP_HIP_CHECK(ctx, hipFree(ctx->wg_rows));
            ctx->wg_rows = nullptr;
            ctx->wg_rows_bytes = 0;
          }
          SCP_HIP_CHECK(ctx, hipMalloc(&ctx->wg_rows, need));
          ctx->wg_rows_bytes = need;
        }
      }
      a.wg_rows = ctx->wg_rows;
      a.wg_part = ctx->wg_part;
      a.ticket = ctx->d_ticket;
      a.rows = tail->rows; a.cap = tail->cap; a.merge_into = tail->merge_into; a.overwrite = tail->overwrite ? 1 : 0;
      a.words = tail->words;
      a.mirror = ctx->d_mirror;
      a.seq = ++ctx->mirror_seq;
#ifdef SCP_PHASE_PROFILE
      a.ablate = getenv("SCP_PAIR_ABLATE") ? atoi(getenv("SCP_PAIR_ABLATE")) : 0;
#else
      a.ablate = 0;
#endif
      dim3 grid(scp_cdiv(nq + 1, SMALL_ROWS), K);
      if (ctx->timing) SCP_HIP_CHECK(ctx, hipEventRecord(ctx->pair_ev0, ctx->stream));
      if (D == 2) hipLaunchKernelGGL((pair_pass_kernel<2, MODE, true, true>), grid, dim3(PAIR_THREADS), lds_small, ctx->stream, a);
      else hipLaunchKernelGGL((pair_pass_kernel<3, MODE, true, true>), grid, dim3(PAIR_THREADS), lds_small, ctx->stream, a);
      SCP_HIP_CHECK(ctx, hipGetLastError());
      if (ctx->timing) SCP_HIP_CHECK(ctx, hipEventRecord(ctx->pair_ev1, ctx->stream));
      ctx->pair_timed = ctx->timing != 0;
      ctx->pair_ran = true;
      tail->done = true;
      ctx->last_pass_small = true;
      return SCP_OK;
    }
  }
  ctx->last_pass_small = false;
  const size_t slice = ((size_t)N * K * D + 1) & ~(size_t)1;  // keep the second array 16-byte aligned
  int rc = ensure_tm(ctx, 2 * slice * sizeof(double));
  if (rc) return rc;
  double* P_tm = MODE != MODE_VIOLATIONS ? ctx->tm_scratch : nullptr;
  double* Q_tm = (MODE != MODE_CHECK && MODE != MODE_SELECT) ? ctx->tm_scratch + slice : nullptr;
  if (MODE == MODE_VIOL_RECOMPUTE)
    hipLaunchKernelGGL(pair_prep_delta_kernel, dim3(scp_cdiv((int64_t)N * K * D, 256)), dim3(256), 0, ctx->stream, N, K, D,
                       pos_ref_layout, p0, P_tm, Q_tm, a.stats);
  else
    hipLaunchKernelGGL(pair_prep_kernel, dim3(scp_cdiv((int64_t)N * K * D, 256)), dim3(256), 0, ctx->stream, N, K, D,
                       a.h, pos_ref_layout, p0, v0, P_tm, Q_tm, a.stats, clear_map, clear_words);
  a.P_tm = P_tm;
  a.Q_tm = Q_tm;
  if (nq <= 0) return SCP_OK;
  const size_t slice_bytes = (size_t)N * D * sizeof(double);
  const bool two_slices = MODE == MODE_LINEARIZE || MODE == MODE_VIOL_RECOMPUTE;  // (the second one PAIR_LDS_Q_OFF doubles in)
  const size_t lds_bytes = two_slices ? PAIR_LDS_SLICE_BYTES + slice_bytes : slice_bytes;
  // the k-slice must start 16-byte aligned in global memory for the double2 staging loads: N*D even
#ifdef SCP_PHASE_PROFILE  // developer build only (make prof): ablation switch of tools/pair_bench.py
  const char* abl = getenv("SCP_PAIR_ABLATE");
  a.ablate = abl ? atoi(abl) : 0;
#else
  a.ablate = 0;
#endif
  // slices beyond 32 KB cut the occupancy below 5 workgroups per CU and the L1/L2 path wins (measured at 2048 x 50:
  // 5.43 TB/s without LDS, 4.74 with; at 1024 x 50, 32 KB: 5.0 with, 4.6 without)
  const bool use_lds = slice_bytes <= (size_t)(two_slices ? 1 : 2) * PAIR_LDS_SLICE_BYTES && ((N * D) % 2 == 0) && !(a.ablate & 2);
  dim3 grid(scp_cdiv(nq + 1, PAIR_ROWS), K);
  dim3 block(PAIR_THREADS);
  if constexpr (MODE == MODE_LINEARIZE) {
    // work items ordered large -> small (PairItems): the bulk in chunks of PAIR_STEPS steps, then about one round of the
    // resident workgroups in half-size and one in quarter-size chunks (whole time steps per class; measured sweep at
    // 1024 x 50: profiles/r03_pair_tail_sweep.txt)
    int per_cu = 0;
    const void* kern = D == 2 ? (use_lds ? (const void*)pair_pass_kernel<2, MODE_LINEARIZE, true> : (const void*)pair_pass_kernel<2, MODE_LINEARIZE, false>)
                              : (use_lds ? (const void*)pair_pass_kernel<3, MODE_LINEARIZE, true> : (const void*)pair_pass_kernel<3, MODE_LINEARIZE, false>);
    SCP_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, PAIR_THREADS, use_lds ? lds_bytes : 0));
    const int64_t slots = (int64_t)std::max(per_cu, 1) * ctx->n_cu;
    PairItems& it = a.items;
    it = PairItems{};
    const int64_t big_per_k = scp_cdiv(nq + 1, PAIR_ROWS);
    int nk_quarter = 0, nk_half = 0;
    if (big_per_k * K > slots) {  // (otherwise every item is resident from the start)
      nk_quarter = (int)((slots * (PAIR_ROWS / 4) + nq / 2) / nq);
      nk_half = (int)((slots * (PAIR_ROWS / 2) + nq / 2) / nq);
      if (nk_quarter + nk_half > K / 2) nk_quarter = nk_half = 0;  // few, long time steps: the classes cannot be cut this way
    }
#ifdef SCP_PHASE_PROFILE
    if (const char* e = getenv("SCP_PAIR_TAIL")) {  // "half quarter": time steps cut in half- / quarter-size chunks
      int h_ = 0, q_ = 0;
      if (sscanf(e, "%d %d", &h_, &q_) == 2 && h_ >= 0 && q_ >= 0 && h_ + q_ < K) { nk_half = h_; nk_quarter = q_; }
    }
#endif
    const int nk[3] = {K - nk_half - nk_quarter, nk_half, nk_quarter};
    const int st[3] = {PAIR_STEPS, PAIR_STEPS / 2, PAIR_STEPS / 4};
    int k0 = 0, item0 = 0;
    for (int c = 0; c < 3; ++c) {
      if (nk[c] <= 0) continue;
      const int m = it.n_classes++;
      it.item0[m] = item0;
      it.k0[m] = k0;
      it.steps[m] = st[c];
      it.cpk[m] = (int)scp_cdiv(nq + 1, (int64_t)2 * PAIR_THREADS * st[c]);
      item0 += nk[c] * it.cpk[m];
      k0 += nk[c];
    }
    it.n_items = item0;
    grid = dim3((unsigned)it.n_items, 1);
  }
  if (ctx->timing) SCP_HIP_CHECK(ctx, hipEventRecord(ctx->pair_ev0, ctx->stream));
#define SCP_LAUNCH_PAIR(DD, LDS)                                                                        \
  hipLaunchKernelGGL((pair_pass_kernel<DD, MODE, LDS>), grid, block, (LDS) ? lds_bytes : 0, ctx->stream, a)
  if (D == 2) {
    if (use_lds) SCP_LAUNCH_PAIR(2, true);
    else SCP_LAUNCH_PAIR(2, false);
  } else {
    if (use_lds) SCP_LAUNCH_PAIR(3, true);
    else SCP_LAUNCH_PAIR(3, false);
  }
#undef SCP_LAUNCH_PAIR
  SCP_HIP_CHECK(ctx, hipGetLastError());
  if (ctx->timing) SCP_HIP_CHECK(ctx, hipEventRecord(ctx->pair_ev1, ctx->stream));
  ctx->pair_timed = ctx->timing != 0;
  ctx->pair_ran = true;
  return SCP_OK;
}

#ifdef SCP_PHASE_PROFILE
// developer hook of the profiling build only (not declared in include/scp_hip.h): the per-workgroup stamps above
extern "C" int scp_debug_small_clocks(unsigned long long* out, int n) {
  if (n > 16) n = 16;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scp_small_clk), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int scp_debug_pair_starts(unsigned long long* out, int n) {
  if (n > SCP_PAIR_CLK_WGS) n = SCP_PAIR_CLK_WGS;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scp_pair_t0), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int scp_debug_pair_clocks(unsigned long long* out, int n) {
  if (n > 2 * SCP_PAIR_CLK_WGS) n = 2 * SCP_PAIR_CLK_WGS;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scp_pair_clk), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

// Device time of the most recent pairwise kernel alone (HIP events on the ctx stream around that one launch).
extern "C" int scp_ctx_last_pair_ms(scp_ctx* ctx, float* ms) {
  if (!ctx || !ms) return SCP_ERR_INVALID;
  if (!ctx->pair_timed && ctx->pair_ran) {  // timing is off (scp_ctx_set_timing): no events were recorded
    *ms = 0.f;
    return SCP_OK;
  }
  if (!ctx->pair_timed) return scp_fail(ctx, SCP_ERR_STATE, "no pairwise pass has run yet");
  SCP_HIP_CHECK(ctx, hipEventSynchronize(ctx->pair_ev1));
  SCP_HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->pair_ev0, ctx->pair_ev1));
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// bitmap -> sorted row list (three tiny launches; the list order is the row order, hence deterministic)
// ----------------------------------------------------------------------------------------------------
constexpr int CMP_THREADS = 256;
constexpr int CMP_WPT = 4;                              // bitmap words per thread
constexpr int CMP_WORDS = CMP_THREADS * CMP_WPT;        // per workgroup

__device__ inline int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[CMP_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < CMP_THREADS / 64; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(CMP_THREADS) void compact_count_kernel(const uint32_t* __restrict__ map, int64_t words,
                                                                     uint32_t* __restrict__ block_tot) {
  const int64_t w0 = (int64_t)blockIdx.x * CMP_WORDS + (int64_t)threadIdx.x * CMP_WPT;
  int c = 0;
#pragma unroll
  for (int i = 0; i < CMP_WPT; ++i)
    if (w0 + i < words) c += __popc(map[w0 + i]);
  int tot;
  block_exclusive_scan(c, &tot);
  if (threadIdx.x == 0) block_tot[blockIdx.x] = (uint32_t)tot;
}

// The finished stats of a pass with a row list also go to the ctx's mapped host mirror (sequence number last): the
// native SCP loop reads them from there without a copy launch and without draining the stream.
__device__ inline void publish_stats(const scp_pair_stats* stats, unsigned long long n_selected, scp_stats_mirror* mirror,
                                     unsigned long long seq) {
  if (!mirror) return;
  const double mind = __hip_atomic_load(&stats->min_dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double maxv = __hip_atomic_load(&stats->max_violation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long fv =
      __hip_atomic_load((const unsigned long long*)&stats->first_violation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned long long*)&mirror->stats.min_dist, (unsigned long long)__double_as_longlong(mind),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store((unsigned long long*)&mirror->stats.first_violation, fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store((unsigned long long*)&mirror->stats.n_selected, n_selected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store((unsigned long long*)&mirror->stats.max_violation, (unsigned long long)__double_as_longlong(maxv),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store((unsigned long long*)&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// exclusive scan of the block totals in place (one workgroup; nblocks is a few hundred), total -> stats
__global__ __launch_bounds__(CMP_THREADS) void compact_scan_kernel(uint32_t* __restrict__ block_tot, int nblocks,
                                                                    scp_pair_stats* __restrict__ stats,
                                                                    scp_stats_mirror* __restrict__ mirror,
                                                                    unsigned long long seq) {
  int carry = 0;
  for (int b0 = 0; b0 < nblocks; b0 += CMP_THREADS) {
    const int b = b0 + threadIdx.x;
    const int v = b < nblocks ? (int)block_tot[b] : 0;
    int tot;
    const int ex = block_exclusive_scan(v, &tot);
    if (b < nblocks) block_tot[b] = (uint32_t)(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0) {
    stats->n_selected = (unsigned long long)carry;
    publish_stats(stats, (unsigned long long)carry, mirror, seq);
  }
}

// write the global row id of every set bit; optionally merge the map into `merge_into` and clear it.  When the list
// is too short for a merging pass (stats->n_selected > cap, written by compact_scan_kernel) nothing is merged: the
// scratch map is cleared and the caller repeats the pass with a longer list from an unchanged working-set bitmap.
__global__ __launch_bounds__(CMP_THREADS) void compact_write_kernel(uint32_t* __restrict__ map, int64_t words,
                                                                     const uint32_t* __restrict__ block_off, int64_t nq,
                                                                     int64_t q_begin, int64_t pairs,
                                                                     int64_t* __restrict__ rows, int64_t cap,
                                                                     uint32_t* __restrict__ merge_into,
                                                                     const scp_pair_stats* __restrict__ stats) {
  const bool overflow = merge_into != nullptr && stats->n_selected > (unsigned long long)cap;
  const int64_t w0 = (int64_t)blockIdx.x * CMP_WORDS + (int64_t)threadIdx.x * CMP_WPT;
  uint32_t wd[CMP_WPT];
  int c = 0;
#pragma unroll
  for (int i = 0; i < CMP_WPT; ++i) {
    wd[i] = (w0 + i < words) ? map[w0 + i] : 0u;
    c += __popc(wd[i]);
  }
  int tot;
  int64_t slot = (int64_t)block_off[blockIdx.x] + block_exclusive_scan(c, &tot);
  if (tot == 0) return;
#pragma unroll
  for (int i = 0; i < CMP_WPT; ++i) {
    uint32_t m = wd[i];
    if (m && merge_into) {
      if (!overflow) merge_into[w0 + i] |= m;
      map[w0 + i] = 0u;
    }
    if (m) {
      // local row lr = 32 (w0 + i) + bit -> global id (lr / nq) pairs + q_begin + lr % nq with ONE 64-bit division per
      // word (its bits belong to at most two time steps when nq >= 32; the inner loop covers tiny pair ranges)
      const int64_t base = (w0 + i) * 32;
      const int64_t kk = base / nq, rr = base - kk * nq;
      while (m) {
        const int bit = __ffs((int)m) - 1;
        m &= m - 1;
        int64_t k2 = kk, r2 = rr + bit;
        while (r2 >= nq) {
          r2 -= nq;
          ++k2;
        }
        if (!overflow && slot < cap) rows[slot] = k2 * pairs + q_begin + r2;
        ++slot;
      }
    }
  }
}

// Small maps (up to CMP1_MAX_WORDS words = 2 M rows, e.g. every map of a 128-agent problem): count, scan, write and the
// stats mirror in ONE workgroup -- the three-launch version costs more in launch boundaries than in work there.
constexpr int CMP1_THREADS = 1024;

// The body, for THREADS threads of ONE workgroup (all of them must call it): returns the number of set bits.
// OVERWRITE: merge_into := map (every word, also the empty ones: the working-set bitmap of a NEW linearisation, no clearing
// launch), map := 0.  COHERENT: the bits were set by other workgroups of the SAME kernel (the small-problem passes run this
// as their tail): the words are read past this XCD's L2.
template <int THREADS, bool COHERENT>
__device__ inline int compact_small_body(uint32_t* __restrict__ map, int64_t words, int64_t nq, int64_t q_begin,
                                         int64_t pairs, int64_t* __restrict__ rows, int64_t cap,
                                         uint32_t* __restrict__ merge_into, bool overwrite) {
  __shared__ int wsum[THREADS / 64];
  __shared__ int total_sh;
  constexpr int CHUNK = THREADS * CMP_WPT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto load = [&](int64_t w) -> uint32_t {
    return COHERENT ? __hip_atomic_load(map + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : map[w];
  };
  // pass 1: the total (decides whether a merging pass may merge at all)
  int c_all = 0;
  for (int64_t w = threadIdx.x; w < words; w += THREADS) c_all += __popc(load(w));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c_all += __shfl_xor(c_all, o);
  __syncthreads();  // (wsum / total_sh of an earlier call in the same kernel have been read)
  if (lane == 0) wsum[wave] = c_all;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < THREADS / 64; ++w) t += wsum[w];
    total_sh = t;
  }
  __syncthreads();
  const int total = total_sh;
  const bool overflow = merge_into != nullptr && !overwrite && (int64_t)total > cap;
  // pass 2: chunk by chunk in row order, block scan per chunk
  int64_t carry = 0;
  for (int64_t base = 0; base < words && (total > 0 || overwrite); base += CHUNK) {
    const int64_t w0 = base + (int64_t)threadIdx.x * CMP_WPT;
    uint32_t wd[CMP_WPT];
    int c = 0;
#pragma unroll
    for (int i = 0; i < CMP_WPT; ++i) {
      wd[i] = (w0 + i < words) ? load(w0 + i) : 0u;
      c += __popc(wd[i]);
    }
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    __syncthreads();  // wsum of the previous chunk (or of pass 1) has been read by everyone
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
      if (w < wave) before += wsum[w];
      tot += wsum[w];
    }
    int64_t slot = carry + before + incl - c;
    carry += tot;
    if (overwrite) {
#pragma unroll
      for (int i = 0; i < CMP_WPT; ++i)
        if (w0 + i < words) merge_into[w0 + i] = wd[i];
    }
    if (c == 0) continue;
#pragma unroll
    for (int i = 0; i < CMP_WPT; ++i) {
      uint32_t m = wd[i];
      if (m && merge_into) {
        if (!overflow && !overwrite) merge_into[w0 + i] |= m;
        map[w0 + i] = 0u;
      }
      if (m) {
        // local row lr = 32 (w0 + i) + bit -> global id (lr / nq) pairs + q_begin + lr % nq with ONE 64-bit division per
        // word (its bits belong to at most two time steps when nq >= 32; the inner loop covers tiny pair ranges)
        const int64_t base = (w0 + i) * 32;
        const int64_t kk = base / nq, rr = base - kk * nq;
        while (m) {
          const int bit = __ffs((int)m) - 1;
          m &= m - 1;
          int64_t k2 = kk, r2 = rr + bit;
          while (r2 >= nq) {
            r2 -= nq;
            ++k2;
          }
          if (!overflow && slot < cap) rows[slot] = k2 * pairs + q_begin + r2;
          ++slot;
        }
      }
    }
  }
  return total;
}

__global__ __launch_bounds__(CMP1_THREADS) void compact_small_kernel(uint32_t* __restrict__ map, int64_t words, int64_t nq,
                                                                      int64_t q_begin, int64_t pairs,
                                                                      int64_t* __restrict__ rows, int64_t cap,
                                                                      uint32_t* __restrict__ merge_into,
                                                                      scp_pair_stats* __restrict__ stats,
                                                                      scp_stats_mirror* __restrict__ mirror,
                                                                      unsigned long long seq) {
  const int total = compact_small_body<CMP1_THREADS, false>(map, words, nq, q_begin, pairs, rows, cap, merge_into, false);
  if (threadIdx.x == 0) {
    stats->n_selected = (unsigned long long)total;
    publish_stats(stats, (unsigned long long)total, mirror, seq);
  }
}

static int ensure_cmp(scp_ctx* ctx, int64_t words) {
  const size_t need_map = (size_t)words * sizeof(uint32_t);
  const size_t need_tot = (size_t)(scp_cdiv(words, CMP_WORDS) + 1) * sizeof(uint32_t);
  if (ctx->cmp_map_bytes < need_map) {
    if (ctx->cmp_map) {
      SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      SCP_HIP_CHECK(ctx, hipFree(ctx->cmp_map));
      ctx->cmp_map = nullptr;
      ctx->cmp_map_bytes = 0;
    }
    SCP_HIP_CHECK(ctx, hipMalloc(&ctx->cmp_map, need_map));
    SCP_HIP_CHECK(ctx, hipMemsetAsync(ctx->cmp_map, 0, need_map, ctx->stream));  // self-cleaning afterwards
    ctx->cmp_map_bytes = need_map;
  }
  if (ctx->cmp_tot_bytes < need_tot) {
    if (ctx->cmp_tot) {
      SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      SCP_HIP_CHECK(ctx, hipFree(ctx->cmp_tot));
      ctx->cmp_tot = nullptr;
      ctx->cmp_tot_bytes = 0;
    }
    SCP_HIP_CHECK(ctx, hipMalloc(&ctx->cmp_tot, need_tot));
    ctx->cmp_tot_bytes = need_tot;
  }
  return SCP_OK;
}

static int launch_compaction(scp_ctx* ctx, uint32_t* map, int64_t words, int64_t nq, int64_t q_begin, int64_t pairs,
                             int64_t* rows, int64_t cap, uint32_t* merge_into, scp_pair_stats* stats) {
  if (words <= 0) return SCP_OK;
  const unsigned long long seq = ++ctx->mirror_seq;
  if (words <= CMP1_MAX_WORDS) {
    hipLaunchKernelGGL(compact_small_kernel, dim3(1), dim3(CMP1_THREADS), 0, ctx->stream, map, words, nq, q_begin, pairs,
                       rows, cap, merge_into, stats, ctx->d_mirror, seq);
    SCP_HIP_CHECK(ctx, hipGetLastError());
    return SCP_OK;
  }
  const int nblocks = scp_cdiv(words, CMP_WORDS);
  hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(CMP_THREADS), 0, ctx->stream, map, words, ctx->cmp_tot);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(CMP_THREADS), 0, ctx->stream, ctx->cmp_tot, nblocks, stats,
                     ctx->d_mirror, seq);
  hipLaunchKernelGGL(compact_write_kernel, dim3(nblocks), dim3(CMP_THREADS), 0, ctx->stream, map, words, ctx->cmp_tot,
                     nq, q_begin, pairs, rows, cap, merge_into, stats);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

static int check_pair_range(scp_ctx* ctx, int N, int K, int D, int64_t q_begin, int64_t q_end) {
  SCP_REQUIRE(ctx, N >= 1 && K >= 1 && (D == 2 || D == 3), "pair pass: bad shape N=%d K=%d D=%d", N, K, D);
  SCP_REQUIRE(ctx, q_begin >= 0 && q_end >= q_begin && q_end <= scp_pairs(N),
              "pair pass: bad pair range [%lld, %lld) of %lld", (long long)q_begin, (long long)q_end,
              (long long)scp_pairs(N));
  SCP_REQUIRE(ctx, K <= 65535, "pair pass: K=%d exceeds grid.y", K);
  SCP_REQUIRE(ctx, q_end - q_begin < ((int64_t)1 << 31) - 2 * PAIR_ROWS,
              "pair pass: %lld pairs per call exceed the 32-bit lane offsets; shard the pair range",
              (long long)(q_end - q_begin));
  return SCP_OK;
}

extern "C" int scp_linearize_pairs(scp_ctx* ctx, int N, int K, int D, double R, double h, int64_t q_begin,
                                   int64_t q_end, const double* pos_prev, const double* p0, const double* v0,
                                   double* eta_out, double* l_out, double margin, int64_t* sel_rows,
                                   int64_t sel_cap, uint32_t* sel_bitmap, scp_pair_stats* stats) {
  if (!ctx) return SCP_ERR_INVALID;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  SCP_REQUIRE(ctx, pos_prev && p0 && v0 && eta_out && l_out && sel_bitmap && stats && (sel_rows || sel_cap == 0),
              "linearize_pairs: null pointer");
  SCP_REQUIRE(ctx, ((uintptr_t)eta_out % 16 == 0) && ((uintptr_t)l_out % 16 == 0),
              "linearize_pairs: eta/l must be 16-byte aligned");
  const int64_t nq = q_end - q_begin;
  PairArgs a{};
  a.N = N; a.K = K; a.D = D; a.R = R; a.h = h;
  a.q_begin = q_begin; a.q_end = q_end; a.pairs = scp_pairs(N);
  a.eta = eta_out; a.l = l_out; a.margin = margin;
  a.bitmap = sel_bitmap; a.mark = sel_bitmap; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, nq);
  const int64_t words = (K * nq + 31) / 32;
  rc = ensure_cmp(ctx, words);
  if (rc) return rc;
  rc = launch_pair_pass<MODE_LINEARIZE>(ctx, a, pos_prev, p0, v0, sel_bitmap, words);  // (its prep kernel clears the map)
  if (rc) return rc;
  return launch_compaction(ctx, sel_bitmap, words, nq, q_begin, a.pairs, sel_rows, sel_cap, nullptr, stats);
}

extern "C" int scp_select_pairs(scp_ctx* ctx, int N, int K, int D, double R, int64_t q_begin, int64_t q_end,
                                const double* pos_prev, double margin, int64_t* sel_rows, int64_t sel_cap,
                                uint32_t* sel_bitmap, scp_pair_stats* stats) {
  if (!ctx) return SCP_ERR_INVALID;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  SCP_REQUIRE(ctx, pos_prev && sel_bitmap && stats && (sel_rows || sel_cap == 0), "select_pairs: null pointer");
  const int64_t nq = q_end - q_begin;
  PairArgs a{};
  a.N = N; a.K = K; a.D = D; a.R = R; a.h = 0.0;
  a.q_begin = q_begin; a.q_end = q_end; a.pairs = scp_pairs(N);
  a.margin = margin;
  a.bitmap = sel_bitmap; a.mark = sel_bitmap; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, nq);
  const int64_t words = (K * nq + 31) / 32;
  rc = ensure_cmp(ctx, words);
  if (rc) return rc;
  PassTail tail{sel_rows, sel_cap, sel_bitmap, true, words, false};
  rc = launch_pair_pass<MODE_SELECT>(ctx, a, pos_prev, nullptr, nullptr, sel_bitmap, words, &tail);  // (its prep kernel clears the map)
  if (rc || tail.done) return rc;
  return launch_compaction(ctx, sel_bitmap, words, nq, q_begin, a.pairs, sel_rows, sel_cap, nullptr, stats);
}

// Working rows appended with eta / l RECOMPUTED from the linearisation point (the row-free loop: scp_select_pairs wrote no
// rows): decode (k, i, j), the two positions of the pair at step k from pos_prev ([N][K][D]), then pair_geom / pair_row --
// the very functions of the linearisation kernel, on the same operands (Q = P - free_motion as its prep kernel forms it):
// bit-identical eta and l.  z = max(A x, l), y = 0 as add_rows_kernel.
template <int D>
__global__ __launch_bounds__(256) void add_rows_at_kernel(int N, int K, int64_t C, int64_t pairs, int64_t base, int64_t n,
                                                           const int64_t* __restrict__ rows,
                                                           const double* __restrict__ pos_prev,
                                                           const double* __restrict__ p0, const double* __restrict__ v0,
                                                           double R, double h, const double* __restrict__ Qx,
                                                           int64_t* __restrict__ w_row, int* __restrict__ wk,
                                                           int* __restrict__ wi, int* __restrict__ wj,
                                                           double* __restrict__ weta, double* __restrict__ wl,
                                                           double* __restrict__ zc, double* __restrict__ yc) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  add_row_at<D>(t, N, K, C, pairs, base, rows, pos_prev, p0, v0, R, h, Qx, w_row, wk, wi, wj, weta, wl, zc, yc);
}

int scp_launch_add_rows_at(scp_ctx* ctx, int N, int K, int D, int64_t base, int64_t n, const int64_t* rows,
                           const double* pos_prev, const double* p0, const double* v0, double R, double h, const double* Qx,
                           int64_t* w_row, int* wk, int* wi, int* wj, double* weta, double* wl, double* zc, double* yc) {
  const int64_t C = (int64_t)N * D;
  if (D == 2)
    hipLaunchKernelGGL(add_rows_at_kernel<2>, dim3(scp_cdiv(n, 256)), dim3(256), 0, ctx->stream, N, K, C, scp_pairs(N), base, n,
                       rows, pos_prev, p0, v0, R, h, Qx, w_row, wk, wi, wj, weta, wl, zc, yc);
  else
    hipLaunchKernelGGL(add_rows_at_kernel<3>, dim3(scp_cdiv(n, 256)), dim3(256), 0, ctx->stream, N, K, C, scp_pairs(N), base, n,
                       rows, pos_prev, p0, v0, R, h, Qx, w_row, wk, wi, wj, weta, wl, zc, yc);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

extern "C" int scp_check_avoidance(scp_ctx* ctx, int N, int K, int D, double R, int64_t q_begin, int64_t q_end,
                                   const double* pos, scp_pair_stats* stats) {
  if (!ctx) return SCP_ERR_INVALID;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  SCP_REQUIRE(ctx, pos && stats, "check_avoidance: null pointer");
  PairArgs a{};
  a.N = N; a.K = K; a.D = D; a.R = R; a.h = 0.0;
  a.q_begin = q_begin; a.q_end = q_end; a.pairs = scp_pairs(N);
  a.stats = stats;
  PassTail tail{nullptr, 0, nullptr, false, (K * (q_end - q_begin) + 31) / 32, false};
  return launch_pair_pass<MODE_CHECK>(ctx, a, pos, nullptr, nullptr, nullptr, 0, &tail);
}

extern "C" int scp_collision_violations(scp_ctx* ctx, int N, int K, int D, double h, int64_t q_begin,
                                        int64_t q_end, const double* eta, const double* l_col, const double* pos,
                                        const double* p0, const double* v0, double feas_tol, int64_t* new_rows,
                                        int64_t new_cap, uint32_t* sel_bitmap, scp_pair_stats* stats) {
  if (!ctx) return SCP_ERR_INVALID;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  SCP_REQUIRE(ctx, eta && l_col && pos && p0 && v0 && sel_bitmap && stats && (new_rows || new_cap == 0),
              "collision_violations: null pointer");
  PairArgs a{};
  a.N = N; a.K = K; a.D = D; a.R = 0.0; a.h = h;
  a.q_begin = q_begin; a.q_end = q_end; a.pairs = scp_pairs(N);
  a.eta = const_cast<double*>(eta); a.l = const_cast<double*>(l_col); a.margin = feas_tol;
  const int64_t nq = q_end - q_begin;
  const int64_t words = (K * nq + 31) / 32;
  rc = ensure_cmp(ctx, words);
  if (rc) return rc;
  a.bitmap = sel_bitmap; a.mark = ctx->cmp_map; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, nq);
  rc = launch_pair_pass<MODE_VIOLATIONS>(ctx, a, pos, p0, v0);
  if (rc) return rc;
  // new rows = bits of the scratch map; merging them into the working-set bitmap also clears the scratch map
  return launch_compaction(ctx, ctx->cmp_map, words, nq, q_begin, a.pairs, new_rows, new_cap, sel_bitmap, stats);
}

extern "C" int scp_collision_violations_at(scp_ctx* ctx, int N, int K, int D, double R, int64_t q_begin, int64_t q_end,
                                           const double* pos_prev, const double* pos_new, double feas_tol,
                                           int64_t* new_rows, int64_t new_cap, uint32_t* sel_bitmap,
                                           scp_pair_stats* stats) {
  if (!ctx) return SCP_ERR_INVALID;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  SCP_REQUIRE(ctx, pos_prev && pos_new && sel_bitmap && stats && (new_rows || new_cap == 0),
              "collision_violations_at: null pointer");
  PairArgs a{};
  a.N = N; a.K = K; a.D = D; a.R = R; a.h = 0.0;
  a.q_begin = q_begin; a.q_end = q_end; a.pairs = scp_pairs(N);
  a.margin = feas_tol;
  const int64_t nq = q_end - q_begin;
  const int64_t words = (K * nq + 31) / 32;
  rc = ensure_cmp(ctx, words);
  if (rc) return rc;
  a.bitmap = sel_bitmap; a.mark = ctx->cmp_map; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, nq);
  PassTail tail{new_rows, new_cap, sel_bitmap, false, words, false};
  rc = launch_pair_pass<MODE_VIOL_RECOMPUTE>(ctx, a, pos_prev, pos_new, nullptr, nullptr, 0, &tail);
  if (rc || tail.done) return rc;
  return launch_compaction(ctx, ctx->cmp_map, words, nq, q_begin, a.pairs, new_rows, new_cap, sel_bitmap, stats);
}

// internal (scp_common.h): scp_qp_get_solution + scp_kinematics + scp_collision_violations_at of a SMALL problem in ONE
// launch -- the pass derives the new positions from the QP's time-major solution x_tm itself and leaves them (pos_out) and
// the solution in the reference layout (x_out) behind; bit-identical to the three calls.  *fused = false: not a small
// problem, nothing was launched.
int scp_violations_from_solution(scp_ctx* ctx, int N, int K, int D, double R, double h, int64_t q_begin, int64_t q_end,
                                 const double* pos_prev, const double* x_tm, const double* p0, const double* v0, double* x_out,
                                 double* pos_out, double feas_tol, int64_t* new_rows, int64_t new_cap, uint32_t* sel_bitmap,
                                 scp_pair_stats* stats, const double* rel_prev, int64_t* spec_rows, int64_t spec_cap,
                                 uint32_t* spec_bitmap, double spec_margin, bool* fused) {
  *fused = false;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  const int64_t nq = q_end - q_begin;
  if (!small_pass_ok(ctx, N, K, D, nq, 2)) return SCP_OK;
  if (!spec_bitmap || !small_pass_ok(ctx, N, K, D, nq, 3)) spec_rows = nullptr;  // (no LDS for the third slice: no speculation)
  PairArgs a{};
  a.N = N; a.K = K; a.D = D; a.R = R; a.h = h;
  a.q_begin = q_begin; a.q_end = q_end; a.pairs = scp_pairs(N);
  a.margin = feas_tol;
  const int64_t words = (K * nq + 31) / 32;
  rc = ensure_cmp(ctx, words);
  if (rc) return rc;
  a.bitmap = sel_bitmap; a.mark = ctx->cmp_map; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, nq);
  a.x_tm = x_tm; a.p0 = p0; a.v0 = v0; a.x_out = x_out; a.pos_out = pos_out;
  a.rel_prev = rel_prev;
  a.rel_blocks = rel_step_blocks((int64_t)N * K * D);
  a.spec_rows = spec_rows; a.spec_cap = spec_cap; a.spec_bitmap = spec_bitmap; a.spec_margin = spec_margin;
  PassTail tail{new_rows, new_cap, sel_bitmap, false, words, false};
  rc = launch_pair_pass<MODE_VIOL_RECOMPUTE>(ctx, a, pos_prev, nullptr, nullptr, nullptr, 0, &tail);
  if (rc) return rc;
  if (!tail.done) return scp_fail(ctx, SCP_ERR_STATE, "violations_from_solution: the small-problem pass did not run");
  *fused = true;
  return SCP_OK;
}

// internal (scp_common.h): scp_qp_get_solution + scp_kinematics + scp_check_avoidance + scp_select_pairs of a SMALL problem
// in ONE launch (after QP#0): the select pass stages the kinematics of the QP's time-major solution, leaves positions and
// solution in the reference layout behind, reduces the a8 statistics (the same ones the check pass reduces) and, for the
// reference's print, leaves the two positions of the first violating pair in the mirror.
int scp_select_from_solution(scp_ctx* ctx, int N, int K, int D, double R, double h, int64_t q_begin, int64_t q_end,
                             const double* x_tm, const double* p0, const double* v0, double* x_out, double* pos_out,
                             double margin, int64_t* sel_rows, int64_t sel_cap, uint32_t* sel_bitmap, scp_pair_stats* stats,
                             bool* fused) {
  *fused = false;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  const int64_t nq = q_end - q_begin;
  if (!small_pass_ok(ctx, N, K, D, nq, 1)) return SCP_OK;
  PairArgs a{};
  a.N = N; a.K = K; a.D = D; a.R = R; a.h = h;
  a.q_begin = q_begin; a.q_end = q_end; a.pairs = scp_pairs(N);
  a.margin = margin;
  a.bitmap = sel_bitmap; a.mark = sel_bitmap; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, nq);
  a.x_tm = x_tm; a.p0 = p0; a.v0 = v0; a.x_out = x_out; a.pos_out = pos_out;
  PassTail tail{sel_rows, sel_cap, sel_bitmap, true, (K * nq + 31) / 32, false};
  rc = launch_pair_pass<MODE_SELECT>(ctx, a, nullptr, nullptr, nullptr, nullptr, 0, &tail);
  if (rc) return rc;
  if (!tail.done) return scp_fail(ctx, SCP_ERR_STATE, "select_from_solution: the small-problem pass did not run");
  *fused = true;
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// gather compact rows
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(int64_t eta_stride, int D, int64_t pairs, int64_t q_begin, int64_t nq,
                                                           const double* __restrict__ eta,
                                                           const double* __restrict__ l,
                                                           const int64_t* __restrict__ rows, int64_t n,
                                                           double* __restrict__ w_eta, double* __restrict__ w_l) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int64_t r = rows[t];
  const int64_t k = r / pairs, q = r % pairs;
  const int64_t lr = k * nq + (q - q_begin);
  for (int d = 0; d < D; ++d) w_eta[t * D + d] = eta[(int64_t)d * eta_stride + lr];
  w_l[t] = l[lr];
}

extern "C" int scp_gather_rows(scp_ctx* ctx, int N, int K, int D, int64_t q_begin, int64_t q_end,
                               const double* eta, const double* l_col, const int64_t* rows, int64_t n,
                               double* w_eta, double* w_l) {
  if (!ctx) return SCP_ERR_INVALID;
  int rc = check_pair_range(ctx, N, K, D, q_begin, q_end);
  if (rc) return rc;
  if (n <= 0) return SCP_OK;
  SCP_REQUIRE(ctx, eta && l_col && rows && w_eta && w_l, "gather_rows: null pointer");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(scp_cdiv(n, 256)), dim3(256), 0, ctx->stream,
                     scp_eta_stride(K, q_end - q_begin), D, scp_pairs(N), q_begin, q_end - q_begin, eta, l_col, rows,
                     n, w_eta, w_l);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// a1: relative step (scp.py:157-159)
// ----------------------------------------------------------------------------------------------------
// partial[2 b], partial[2 b + 1] = block b's sums; the partials live in mapped host memory and the LAST block to finish
// (a ticket counter in device memory, at most 32 tickets) raises the completion word, so the host needs neither a copy
// launch nor a stream drain to read them.
__global__ __launch_bounds__(256) void rel_step_partial_kernel(int64_t n, const double* __restrict__ a,
                                                                const double* __restrict__ b,
                                                                double* __restrict__ partial,
                                                                unsigned* __restrict__ ticket,
                                                                unsigned long long* __restrict__ done,
                                                                unsigned long long seq) {
  __shared__ double s0[4], s1[4];
  double d2 = 0.0, b2 = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    rel_accum(a[t], b[t], d2, b2);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d2 += __shfl_xor(d2, o);
    b2 += __shfl_xor(b2, o);
  }
  if ((threadIdx.x & 63) == 0) {
    s0[threadIdx.x >> 6] = d2;
    s1[threadIdx.x >> 6] = b2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store((unsigned long long*)&partial[2 * blockIdx.x],
                       (unsigned long long)__double_as_longlong((s0[0] + s0[1]) + (s0[2] + s0[3])), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((unsigned long long*)&partial[2 * blockIdx.x + 1],
                       (unsigned long long)__double_as_longlong((s1[0] + s1[1]) + (s1[2] + s1[3])), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      *ticket = 0u;  // (the next launch on this stream starts after this kernel has ended)
      __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

extern "C" int scp_rel_step(scp_ctx* ctx, int64_t n, const double* a_new, const double* a_prev, double* out) {
  if (!ctx) return SCP_ERR_INVALID;
  SCP_REQUIRE(ctx, n > 0 && a_new && a_prev && out, "rel_step: bad arguments");
  const int blocks = rel_step_blocks(n);
  // the (at most 64) partial sums go straight to the mapped host scratch: no copy launch
  const unsigned long long seq = ++ctx->rel_seq;
  hipLaunchKernelGGL(rel_step_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, a_new, a_prev,
                     ctx->h_scratch_dev, (unsigned*)(ctx->d_scratch + 64), (unsigned long long*)(ctx->h_scratch_dev + 64), seq);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  {
    volatile unsigned long long* flag = (volatile unsigned long long*)(ctx->h_scratch + 64);
    if (!scp_wait_host_word(flag, seq, 30)) SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // a fault surfaces here
    if (*flag != seq) return scp_fail(ctx, SCP_ERR_HIP, "rel_step: completion word not written");
  }
  double d2 = 0.0, b2 = 0.0;
  for (int b = 0; b < blocks; ++b) {
    d2 += ctx->h_scratch[2 * b];
    b2 += ctx->h_scratch[2 * b + 1];
  }
  out[0] = std::sqrt(d2);
  out[1] = std::sqrt(b2);
  out[2] = out[0] / out[1];
  return SCP_OK;
}
