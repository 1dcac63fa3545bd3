// Internal declarations shared by the translation units of libscp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "scp_hip.h"

// mapped host copy of the stats of the latest pass that produced a row list (written by its last kernel, seq last)
struct scp_stats_mirror {
  scp_pair_stats stats;
  unsigned long long n_spec;  // small-problem violations pass from a solution: rows of its speculative selection
  double pts[6];   // small-problem select pass from a QP solution: the two positions of the first violating pair
  double rel[64];  // small-problem violations pass: the partial sums of scp_rel_step(x_new, x_prev), block by block
  volatile unsigned long long seq;
};

constexpr int SCP_SMALL_MAX_WG = 2048;  // workgroups of a small-problem pairwise pass (the whole pass in one launch)

struct scp_ctx {
  int device;
  int n_cu;  // compute units of the device (resident-workgroup limit of the persistent kernels)
  hipStream_t stream;
  char err[512];
  // small device scratch for reductions / host read-back
  double* d_scratch;      // 72 doubles
  double* h_scratch;      // pinned, 72 doubles
  double* h_scratch_dev;  // its device address (kernels that hand a few numbers to the host write there directly)
  hipEvent_t ev0, ev1;
  hipEvent_t pair_ev0, pair_ev1;  // bracket the most recent pairwise kernel (scp_ctx_last_pair_ms)
  bool pair_timed;
  bool pair_ran;
  bool last_pass_small;  // the latest pairwise pass ran as ONE launch and published its stats in the host mirror (also a check pass)
  double* tm_scratch;     // time-major copy of a trajectory array for the pairwise passes (grown on demand)
  size_t tm_bytes;
  uint32_t* cmp_map;      // scratch bitmap of the violations pass (self-cleaning), grown on demand
  size_t cmp_map_bytes;
  uint32_t* cmp_tot;      // per-block totals / offsets of the bitmap compaction
  size_t cmp_tot_bytes;
  scp_stats_mirror* h_mirror;  // mapped host memory and its device address
  scp_stats_mirror* d_mirror;
  unsigned long long mirror_seq;  // sequence number of the latest compaction launch
  unsigned long long* wg_part;    // [SCP_SMALL_MAX_WG][4] per-workgroup partials of a small-problem pairwise pass
  uint32_t* wg_rows;              // [workgroups][8192] per-workgroup sub-lists of its marked rows (grown on demand)
  size_t wg_rows_bytes;
  unsigned* d_ticket;             // its last-workgroup-done counter (zero between launches)
  int timing;                     // HIP events around the pairwise kernels and the QP solves (scp_ctx_set_option; default on)
  int small_pass;                 // one-launch pairwise passes for small problems (scp_ctx_set_option; default on)
  unsigned long long rel_seq;     // of the latest scp_rel_step (completion word: h_scratch[64]; partials: h_scratch[0..64))
};

static inline int scp_fail(scp_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define SCP_HIP_CHECK(ctx, call)                                                                     \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return scp_fail((ctx), SCP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                      __FILE__, __LINE__);                                                           \
  } while (0)

#define SCP_REQUIRE(ctx, cond, ...)                                   \
  do {                                                                \
    if (!(cond)) return scp_fail((ctx), SCP_ERR_INVALID, __VA_ARGS__); \
  } while (0)

static inline int64_t scp_pairs(int N) { return (int64_t)N * (N - 1) / 2; }
static inline int scp_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Raise a kernel's dynamic-LDS limit to at least `bytes` (gfx950: up to 160 KiB per workgroup).  The attribute belongs to
// the (device, kernel) pair and is only ever raised, so concurrent solves of different sizes cannot lower each other's
// limit.  Thread safe.
hipError_t scp_raise_lds_limit(int device, const void* kernel, size_t bytes);

// Wait until a mapped host word takes the value `seq` (written by the last kernel of a check / by the persistent kernel).
// mode 0 (default): spin -- lowest latency, one host core per waiting thread; mode 1 (scp_set_host_wait): spin for
// ~20 us, then poll every ~20 us from a sleep, for many solver threads on few cores (compute-trajectories-batch).
// Returns false after `timeout_s` seconds.
bool scp_wait_host_word(volatile unsigned long long* word, unsigned long long seq, int timeout_s);

// Stats of the latest scp_linearize_pairs / scp_collision_violations[_at] call of this ctx, from the host mirror: waits for
// that pass's last kernel only (no stream drain, no copy launch).
int scp_ctx_wait_stats(scp_ctx* ctx, scp_pair_stats* out);

// scp_qp_get_solution + scp_kinematics + scp_collision_violations_at of a small problem in one launch (scp_kernels.hip)
int scp_violations_from_solution(scp_ctx* ctx, int N, int K, int D, double R, double h, int64_t q_begin, int64_t q_end,
                                 const double* pos_prev, const double* x_tm, const double* p0, const double* v0, double* x_out,
                                 double* pos_out, double feas_tol, int64_t* new_rows, int64_t new_cap, uint32_t* sel_bitmap,
                                 scp_pair_stats* stats, const double* rel_prev /* [N][K][D] or NULL */,
                                 int64_t* spec_rows /* or NULL: + the selection around the new positions */, int64_t spec_cap,
                                 uint32_t* spec_bitmap, double spec_margin, bool* fused);
// scp_rel_step's numbers from the sums that pass left in the mirror (after its stats have arrived)
void scp_ctx_mirror_rel(scp_ctx* ctx, int64_t n, double* out);
// scp_qp_get_solution + scp_kinematics + scp_check_avoidance + scp_select_pairs of a small problem in one launch
int scp_select_from_solution(scp_ctx* ctx, int N, int K, int D, double R, double h, int64_t q_begin, int64_t q_end,
                             const double* x_tm, const double* p0, const double* v0, double* x_out, double* pos_out,
                             double margin, int64_t* sel_rows, int64_t sel_cap, uint32_t* sel_bitmap, scp_pair_stats* stats,
                             bool* fused);
int scp_launch_kinematics_copy(scp_ctx* ctx, int N, int K, int D, double h, const double* acc, const double* p0,
                               const double* v0, double* pos_out, double* vel_out, double* acc_copy);
// the QP's current iterate in its own layout ([K][N D], device pointer)
const double* scp_qp_solution_tm(const scp_qp* qp);

// scp_qp_reset(x0) + scp_qp_add_rows_at(rows): one launch when the problem is small, the two calls otherwise (same bits)
int scp_qp_reset_add_rows_at(scp_qp* qp, const double* x0, int64_t n, const int64_t* rows, const double* pos_prev,
                             const double* p0, const double* v0, double R);
// scp_gather_rows + scp_qp_add_rows in one launch: eta / l_col are the outputs of scp_linearize_pairs over [q_begin, q_end)
int scp_qp_add_rows_from_pass(scp_qp* qp, int64_t n, const int64_t* rows, const double* eta, const double* l_col,
                              int64_t q_begin, int64_t q_end);

// working rows [base, base + n) of a QP with eta / l recomputed from the linearisation point (scp_qp_add_rows_at)
int scp_launch_add_rows_at(scp_ctx* ctx, int N, int K, int D, int64_t base, int64_t n, const int64_t* rows,
                           const double* pos_prev, const double* p0, const double* v0, double R, double h, const double* Qx,
                           int64_t* w_row, int* wk, int* wi, int* wj, double* weta, double* wl, double* zc, double* yc);

// ---- internal launchers (time-major device layout [K][C], C = N*D) --------------------------------
// Y[R][C] = alpha * A[R][M] X[M][C] + beta * Y   (row-major; A small and L2 resident)
int scp_launch_gemm(scp_ctx* ctx, int use_mfma, int R, int M, int C, double alpha, const double* A,
                    const double* X, double beta, double* Y);
// [N][K][D] <-> [K][N*D]
int scp_launch_to_time_major(scp_ctx* ctx, int N, int K, int D, const double* src, double* dst);
int scp_launch_from_time_major(scp_ctx* ctx, int N, int K, int D, const double* src, double* dst);
// fixed-row bounds in time-major stacked layout: rows [0,K-1) jerk, [K-1,2K-1) acc, [2K-1,3K-1) vel,
// [3K-1,4K-1) pos, each row C = N*D wide.
int scp_launch_bounds_time_major(scp_ctx* ctx, int N, int K, int D, double h, const double* limits_host,
                                 const double* space_host, const double* p0, const double* v0,
                                 const double* pf, const double* vf, double* l_tm, double* u_tm,
                                 double* states_out /* [4][N][D] = p0, v0, pf, vf, or NULL */);
