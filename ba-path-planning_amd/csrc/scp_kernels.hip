// Assembly half of the SCP hot path on gfx950: context, kinematics (a4/a7), fixed bounds (a2), the O(N^2 K)
// pairwise passes (a5, a8, constraint generation) and the SCP relative step (a1).
// Reference: /root/reference/src/path_planning/solvers/scp.py (line numbers cited per kernel).
#include "scp_common.h"

#include <cmath>

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
  if (hipMalloc(&ctx->d_scratch, 64 * sizeof(double)) != hipSuccess ||
      hipHostMalloc(&ctx->h_scratch, 64 * sizeof(double)) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipEventCreate(&ctx->pair_ev0) != hipSuccess || hipEventCreate(&ctx->pair_ev1) != hipSuccess) {
    delete ctx;
    return SCP_ERR_HIP;
  }
  *out = ctx;
  return SCP_OK;
}

extern "C" void scp_ctx_destroy(scp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ctx->d_scratch);
  if (ctx->tm_scratch) (void)hipFree(ctx->tm_scratch);
  (void)hipHostFree(ctx->h_scratch);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  (void)hipEventDestroy(ctx->pair_ev0);
  (void)hipEventDestroy(ctx->pair_ev1);
  delete ctx;
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
__global__ __launch_bounds__(256) void kinematics_kernel(int N, int K, int D, double h,
                                                          const double* __restrict__ acc,
                                                          const double* __restrict__ p0,
                                                          const double* __restrict__ v0, double* __restrict__ pos,
                                                          double* __restrict__ vel) {
#pragma clang fp contract(off)  // every product below is rounded before it is added, as numpy does
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)N * K * D) return;
  const int d = (int)(t % D);
  const int k = (int)((t / D) % K);
  const int i = (int)(t / ((int64_t)D * K));
  const double pi = p0[i * D + d], vi = v0[i * D + d];
  double v = vi;
  const double hk = h * (double)k;
  const double hkv = hk * vi;
  double p = pi + hkv;  // p0 + (h*k)*v0   scp.py:393
  const double hh = h * h;
  const double* a = acc + (int64_t)i * K * D + d;
  for (int j = 0; j < k; ++j) {
    const double aj = a[(int64_t)j * D];
    const double hv = h * aj;
    v = v + hv;  // scp.py:390
    const double w = hh * ((double)(k - j) - 0.5);
    const double wa = w * aj;
    p = p + wa;  // scp.py:395
  }
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
                     v0, pos_out, vel_out);
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
                                                                 double* __restrict__ l, double* __restrict__ u) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t C = (int64_t)N * D;
  const int rows = 4 * K - 1;
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
                                 const double* vf, double* l_tm, double* u_tm) {
  BoundParams bp;
  fill_bound_params(bp, D, limits, space);
  const int64_t m = (int64_t)N * D * (4 * K - 1);
  hipLaunchKernelGGL(bounds_time_major_kernel, dim3(scp_cdiv(m, 256)), dim3(256), 0, ctx->stream, bp, N, K, D, h,
                     p0, v0, pf, vf, l_tm, u_tm);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

// ----------------------------------------------------------------------------------------------------
// pairwise passes
// ----------------------------------------------------------------------------------------------------
// Lexicographic pair index q -> (i, j), i < j.  Row i of the triangle starts at off(i) = i (2N - i - 1) / 2.
__device__ __host__ inline int64_t tri_off(int64_t i, int64_t N) { return i * (2 * N - i - 1) / 2; }

__device__ inline void decode_pair(int64_t q, int N, int& i, int& j) {
  const double b = 2.0 * N - 1.0;
  int64_t ii = (int64_t)((b - sqrt(b * b - 8.0 * (double)q)) * 0.5);
  if (ii < 0) ii = 0;
  if (ii > N - 2) ii = N - 2;
  while (tri_off(ii, N) > q) --ii;
  while (ii < N - 2 && tri_off(ii + 1, N) <= q) ++ii;
  i = (int)ii;
  j = (int)(q - tri_off(ii, N) + ii + 1);
}

__global__ void pair_stats_init_kernel(scp_pair_stats* s) {
  s->min_dist = __longlong_as_double(0x7FF0000000000000LL);
  s->first_violation = 0xFFFFFFFFFFFFFFFFULL;
  s->n_selected = 0;
  s->max_violation = -__longlong_as_double(0x7FF0000000000000LL);
}

// wave-aggregated append of the lanes with `sel` to list[] (arrival order), returns nothing; rows beyond
// `cap` are counted but not stored.
__device__ inline void wave_append(bool sel, int64_t value, int64_t* list, int64_t cap,
                                   unsigned long long* counter) {
  const unsigned long long mask = __ballot(sel);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(mask));
  base = __shfl(base, leader);
  if (sel) {
    const unsigned long long below = mask & ((1ULL << lane) - 1ULL);
    const int64_t idx = (int64_t)(base + __popcll(below));
    if (idx < cap) list[idx] = value;
  }
}

__device__ inline double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ inline unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
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
constexpr int PAIR_RPT = 16;                         // rows per thread (as 8 steps of 2 adjacent rows)
constexpr int PAIR_ROWS = PAIR_THREADS * PAIR_RPT;   // rows (= pairs at one k) per workgroup

enum PairMode { MODE_LINEARIZE = 0, MODE_CHECK = 1, MODE_VIOLATIONS = 2 };

struct PairArgs {
  int N, K, D;
  double R, h;
  int64_t q_begin, q_end, pairs;
  const double* pos_tm;  // [K][N][D] time-major positions (prev for linearize, new for violations)
  const double* p0;      // [N][D]
  const double* v0;      // [N][D]
  double* eta;           // [D][eta_stride]
  int64_t eta_stride;    // scp_eta_stride(K, nq): K*nq rounded up to even (16-byte aligned planes)
  double* l;             // [K*nq]
  double margin;         // linearize: selection margin; violations: feas_tol
  int64_t* sel_rows;
  int64_t sel_cap;
  uint32_t* bitmap;
  scp_pair_stats* stats;
};

// One workgroup = PAIR_ROWS consecutive local rows of one time step k.  The k-slice of the trajectory array
// (and of p0, v0) is staged in LDS once per workgroup: every row then costs two LDS reads per array
// (P_i broadcast within the wave, P_j consecutive lanes -> consecutive 8*D-byte slots), and the only HBM
// traffic is the fully coalesced, 16-byte-per-lane streaming write (linearize) or read (violations) of the
// compact rows: 8*(D+1) bytes per row.
template <int D, int MODE, bool USE_LDS>
__global__ __launch_bounds__(PAIR_THREADS) void pair_pass_kernel(PairArgs a) {
  extern __shared__ double lds[];
  const int N = a.N;
  const int k = blockIdx.y;
  const int64_t nq = a.q_end - a.q_begin;
  const int64_t slice0 = (int64_t)k * nq;       // first local row of this k
  const int64_t par = slice0 & 1;               // keep every thread's first row at an even local row id
  const int64_t c0 = (int64_t)blockIdx.x * PAIR_ROWS - par;  // first local pair offset of this workgroup

  const double* Pg = a.pos_tm + (int64_t)k * N * D;
  const double* P;
  const double* P0;
  const double* V0;
  if (USE_LDS) {
    double* sP = lds;
    double* sP0 = lds + (int64_t)N * D;
    double* sV0 = lds + 2 * (int64_t)N * D;
    for (int t = threadIdx.x; t < N * D; t += PAIR_THREADS) {
      sP[t] = Pg[t];
      if (MODE != MODE_CHECK) {
        sP0[t] = a.p0[t];
        sV0[t] = a.v0[t];
      }
    }
    __syncthreads();
    P = sP;
    P0 = sP0;
    V0 = sV0;
  } else {
    P = Pg;
    P0 = a.p0;
    V0 = a.v0;
  }

  const double kh = (double)k * a.h;
  const double thr = a.R - 0.01;  // scp.py:610
  double my_min = __longlong_as_double(0x7FF0000000000000LL);
  double my_maxv = -my_min;
  unsigned long long my_first = 0xFFFFFFFFFFFFFFFFULL;

#pragma unroll 1
  for (int s = 0; s < PAIR_RPT / 2; ++s) {
    const int64_t off = c0 + (int64_t)s * (2 * PAIR_THREADS) + 2 * threadIdx.x;  // local pair offset of row A
    double eta_v[2][D];
    double l_v[2];
    bool valid[2], sel[2];
    int64_t grow[2];
    int i = 0, j = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t o = off + e;
      valid[e] = (o >= 0) && (o < nq);
      sel[e] = false;
      grow[e] = 0;
      l_v[e] = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) eta_v[e][d] = 0.0;
      if (!valid[e]) continue;
      const int64_t q = a.q_begin + o;
      if (e == 0 || !valid[0]) {
        decode_pair(q, N, i, j);
      } else {  // next pair in lexicographic order
        if (++j >= N) {
          ++i;
          j = i + 1;
        }
      }
      grow[e] = (int64_t)k * a.pairs + q;
      double diff[D];
      double ss = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        diff[d] = P[i * D + d] - P[j * D + d];
        ss += diff[d] * diff[d];
      }
      const double raw = sqrt(ss);
      my_min = fmin(my_min, raw);
      if (MODE != MODE_VIOLATIONS && raw < thr) {
        const unsigned long long g = (unsigned long long)grow[e];
        my_first = g < my_first ? g : my_first;
      }
      if (MODE == MODE_LINEARIZE) {
        double dist = raw;
        double eta[D];
        if (raw < 1e-6) {  // scp.py:503-507 with a fixed direction instead of a random one
          dist = 1.0;
#pragma unroll
          for (int d = 0; d < D; ++d) eta[d] = d == 0 ? 1.0 : 0.0;
        } else {
          const double inv = 1.0 / raw;
#pragma unroll
          for (int d = 0; d < D; ++d) eta[d] = diff[d] * inv;  // scp.py:509
        }
        double ip = 0.0, iv = 0.0, lin = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          ip += eta[d] * (P0[i * D + d] - P0[j * D + d]);  // scp.py:543
          iv += eta[d] * (V0[i * D + d] - V0[j * D + d]);  // scp.py:544
          lin += eta[d] * diff[d];                          // scp.py:547
          eta_v[e][d] = eta[d];
        }
        lin -= dist;
        l_v[e] = a.R + lin - (ip + iv * kh);  // scp.py:549
        sel[e] = (dist - a.R) < a.margin;
      } else if (MODE == MODE_VIOLATIONS) {
        // (A x)_r = eta . ((P_i - c_i) - (P_j - c_j)),  c = p0 + (k h) v0
        const int64_t lr = slice0 + o;
        double ax = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double qi = P[i * D + d] - (P0[i * D + d] + kh * V0[i * D + d]);
          const double qj = P[j * D + d] - (P0[j * D + d] + kh * V0[j * D + d]);
          ax += a.eta[(int64_t)d * a.eta_stride + lr] * (qi - qj);
        }
        const double viol = a.l[lr] - ax;
        my_maxv = fmax(my_maxv, viol);
        sel[e] = viol > a.margin;
      }
    }

    if (MODE == MODE_LINEARIZE) {
      const int64_t lrA = slice0 + off;  // even by construction
      const int64_t stride = a.eta_stride;
      if (valid[0] && valid[1]) {
#pragma unroll
        for (int d = 0; d < D; ++d)
          *reinterpret_cast<double2*>(a.eta + d * stride + lrA) = make_double2(eta_v[0][d], eta_v[1][d]);
        *reinterpret_cast<double2*>(a.l + lrA) = make_double2(l_v[0], l_v[1]);
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (valid[e]) {
#pragma unroll
            for (int d = 0; d < D; ++d) a.eta[d * stride + lrA + e] = eta_v[e][d];
            a.l[lrA + e] = l_v[e];
          }
      }
    }
    if (MODE != MODE_CHECK) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        bool take = sel[e];
        if (take) {
          const int64_t lr = slice0 + off + e;
          const uint32_t bit = 1u << (lr & 31);
          const uint32_t old = atomicOr(a.bitmap + (lr >> 5), bit);
          if (MODE == MODE_VIOLATIONS) take = (old & bit) == 0;  // rows already in the working set stay out
        }
        wave_append(take, grow[e], a.sel_rows, a.sel_cap, (unsigned long long*)&a.stats->n_selected);
      }
    }
  }

  // wavefront reductions -> one atomic per wave
  my_min = wave_min(my_min);
  my_first = wave_min_u64(my_first);
  if (MODE == MODE_VIOLATIONS) my_maxv = wave_max(my_maxv);
  if ((threadIdx.x & 63) == 0) {
    if (my_min < __longlong_as_double(0x7FF0000000000000LL)) atomic_min_pos_double(&a.stats->min_dist, my_min);
    if (my_first != 0xFFFFFFFFFFFFFFFFULL) atomicMin((unsigned long long*)&a.stats->first_violation, my_first);
    if (MODE == MODE_VIOLATIONS && my_maxv > -__longlong_as_double(0x7FF0000000000000LL))
      atomic_max_double(&a.stats->max_violation, my_maxv);
  }
}

// scratch for the time-major copy of the trajectory array (grown on demand, owned by the ctx)
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

template <int MODE>
static int launch_pair_pass(scp_ctx* ctx, PairArgs& a, const double* pos_ref_layout) {
  const int N = a.N, K = a.K, D = a.D;
  const int64_t nq = a.q_end - a.q_begin;
  int rc = ensure_tm(ctx, (size_t)N * K * D * sizeof(double));
  if (rc) return rc;
  rc = scp_launch_to_time_major(ctx, N, K, D, pos_ref_layout, ctx->tm_scratch);
  if (rc) return rc;
  a.pos_tm = ctx->tm_scratch;
  hipLaunchKernelGGL(pair_stats_init_kernel, dim3(1), dim3(1), 0, ctx->stream, a.stats);
  if (nq <= 0) return SCP_OK;
  const size_t lds_bytes = (size_t)(MODE == MODE_CHECK ? 1 : 3) * N * D * sizeof(double);
  const bool use_lds = lds_bytes <= 64 * 1024;
  dim3 grid(scp_cdiv(nq + 1, PAIR_ROWS), K);
  dim3 block(PAIR_THREADS);
  SCP_HIP_CHECK(ctx, hipEventRecord(ctx->pair_ev0, ctx->stream));
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
  SCP_HIP_CHECK(ctx, hipEventRecord(ctx->pair_ev1, ctx->stream));
  ctx->pair_timed = true;
  return SCP_OK;
}

// Device time of the most recent pairwise kernel alone (HIP events on the ctx stream around that one launch).
extern "C" int scp_ctx_last_pair_ms(scp_ctx* ctx, float* ms) {
  if (!ctx || !ms) return SCP_ERR_INVALID;
  if (!ctx->pair_timed) return scp_fail(ctx, SCP_ERR_STATE, "no pairwise pass has run yet");
  SCP_HIP_CHECK(ctx, hipEventSynchronize(ctx->pair_ev1));
  SCP_HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->pair_ev0, ctx->pair_ev1));
  return SCP_OK;
}

static int check_pair_range(scp_ctx* ctx, int N, int K, int D, int64_t q_begin, int64_t q_end) {
  SCP_REQUIRE(ctx, N >= 1 && K >= 1 && (D == 2 || D == 3), "pair pass: bad shape N=%d K=%d D=%d", N, K, D);
  SCP_REQUIRE(ctx, q_begin >= 0 && q_end >= q_begin && q_end <= scp_pairs(N),
              "pair pass: bad pair range [%lld, %lld) of %lld", (long long)q_begin, (long long)q_end,
              (long long)scp_pairs(N));
  SCP_REQUIRE(ctx, K <= 65535, "pair pass: K=%d exceeds grid.y", K);
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
  a.p0 = p0; a.v0 = v0; a.eta = eta_out; a.l = l_out; a.margin = margin;
  a.sel_rows = sel_rows; a.sel_cap = sel_cap; a.bitmap = sel_bitmap; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, nq);
  const size_t words = (size_t)((K * nq + 31) / 32);
  if (words) SCP_HIP_CHECK(ctx, hipMemsetAsync(sel_bitmap, 0, words * sizeof(uint32_t), ctx->stream));
  return launch_pair_pass<MODE_LINEARIZE>(ctx, a, pos_prev);
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
  return launch_pair_pass<MODE_CHECK>(ctx, a, pos);
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
  a.p0 = p0; a.v0 = v0; a.eta = const_cast<double*>(eta); a.l = const_cast<double*>(l_col); a.margin = feas_tol;
  a.sel_rows = new_rows; a.sel_cap = new_cap; a.bitmap = sel_bitmap; a.stats = stats;
  a.eta_stride = scp_eta_stride(K, q_end - q_begin);
  return launch_pair_pass<MODE_VIOLATIONS>(ctx, a, pos);
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
__global__ __launch_bounds__(256) void rel_step_partial_kernel(int64_t n, const double* __restrict__ a,
                                                                const double* __restrict__ b,
                                                                double* __restrict__ partial) {
  __shared__ double s0[4], s1[4];
  double d2 = 0.0, b2 = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const double x = a[t], y = b[t];
    d2 += (x - y) * (x - y);
    b2 += y * y;
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
    partial[2 * blockIdx.x] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
    partial[2 * blockIdx.x + 1] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
  }
}

extern "C" int scp_rel_step(scp_ctx* ctx, int64_t n, const double* a_new, const double* a_prev, double* out) {
  if (!ctx) return SCP_ERR_INVALID;
  SCP_REQUIRE(ctx, n > 0 && a_new && a_prev && out, "rel_step: bad arguments");
  const int blocks = (int)((n + 256 * 8 - 1) / (256 * 8)) < 32 ? (int)((n + 256 * 8 - 1) / (256 * 8)) : 32;
  hipLaunchKernelGGL(rel_step_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, a_new, a_prev,
                     ctx->d_scratch);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  SCP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch, 2 * blocks * sizeof(double),
                                    hipMemcpyDeviceToHost, ctx->stream));
  SCP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
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
