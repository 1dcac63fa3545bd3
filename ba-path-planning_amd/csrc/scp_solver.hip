// scp_solver: the whole of SCP.generate_trajectories (scp.py:131-180) behind ONE C call.
//
// The loop itself -- QP#0, the avoidance check, per iteration: kinematics, pairwise linearisation, the joint QP with exact
// constraint generation, the relative-step test (scp.py:152-166) -- is control flow around the entry points of
// scp_hip.h; path_planning/solvers/scp.py drives it from Python (about a hundred ctypes / torch calls and half a dozen
// blocking reads per SCP iteration, ~1 ms of host time) and this file drives the SAME calls in the SAME order natively, so
// that a 100-agent solve is bounded by its ~2 ms of GPU time and many solves can run side by side (compute-trajectories-
// batch, config 5).  Results are bit-identical to the Python-driven loop (tests/test_native_gpu.py).
//
// The solver object owns its device memory (compact rows, bitmap, row list, QP workspace, trajectories), allocated with
// hipMalloc and grown on demand like path_planning/_hip.py does for the Python-driven path.
#include "scp_common.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <vector>

struct scp_solver {
  scp_ctx* ctx;
  int N, K, D;
  double h, R;
  scp_qp_settings st;
  int64_t pairs, rows, stride;
  // pairwise pass buffers
  double *eta, *l;
  uint32_t* bitmap;
  int64_t* sel;
  int64_t sel_cap;
  scp_pair_stats* stats;    // device
  scp_pair_stats* h_stats;  // pinned
  // QP
  void* ws;
  scp_qp* qp;
  int64_t row_cap;
  // trajectories [N][K][D]
  double *acc, *x, *pos_a, *pos_b, *vel;
  double* pair_pts;  // pinned, 2 D doubles: the two positions of the first violation
};

namespace {

#define SV_HIP(call) SCP_HIP_CHECK(s->ctx, (call))
#define SV_CHECK(call)             \
  do {                             \
    int rc_ = (call);              \
    if (rc_ != SCP_OK) return rc_; \
  } while (0)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// stats of the pass just enqueued.  Passes that end in a row list (linearise, violations) publish them in the ctx's mapped
// host mirror; the check pass has no trailing kernel, its stats are copied.
int read_stats(scp_solver* s, bool from_mirror) {
  if (from_mirror && s->pairs > 0) return scp_ctx_wait_stats(s->ctx, s->h_stats);
  SV_HIP(hipMemcpyAsync(s->h_stats, s->stats, sizeof(scp_pair_stats), hipMemcpyDeviceToHost, s->ctx->stream));
  SV_HIP(hipStreamSynchronize(s->ctx->stream));
  return SCP_OK;
}

int grow_sel(scp_solver* s, int64_t need) {
  int64_t cap = std::max<int64_t>(need, 2 * s->sel_cap);
  cap = std::min<int64_t>(std::max<int64_t>(s->rows, 1), cap);
  SV_HIP(hipStreamSynchronize(s->ctx->stream));
  SV_HIP(hipFree(s->sel));
  s->sel = nullptr;
  SV_HIP(hipMalloc(&s->sel, (size_t)cap * sizeof(int64_t)));
  s->sel_cap = cap;
  return SCP_OK;
}

int make_qp(scp_solver* s, int64_t cap, void** ws_out, scp_qp** qp_out) {
  const size_t bytes = scp_qp_workspace_bytes(s->N, s->K, s->D, cap);
  if (bytes == 0) return scp_fail(s->ctx, SCP_ERR_INVALID, "solver: bad QP shape");
  void* ws = nullptr;
  SV_HIP(hipMalloc(&ws, bytes));
  scp_qp* qp = nullptr;
  int rc = scp_qp_create(s->ctx, s->N, s->K, s->D, s->h, &s->st, ws, bytes, cap, &qp);
  if (rc != SCP_OK) {
    (void)hipFree(ws);
    return rc;
  }
  *ws_out = ws;
  *qp_out = qp;
  return SCP_OK;
}

// working set outgrew the capacity: continue in a larger workspace (keep_state: iterate, duals, rows, rho travel along)
int grow_qp(scp_solver* s, int64_t need, bool keep_state, const double* limits, const double* space, const double* p0,
            const double* v0, const double* pf, const double* vf) {
  int64_t cap = std::max<int64_t>(need, 2 * s->row_cap);
  cap = std::min<int64_t>((int64_t)s->K * s->pairs, cap);
  void* ws = nullptr;
  scp_qp* qp = nullptr;
  SV_CHECK(make_qp(s, cap, &ws, &qp));
  int rc = keep_state ? scp_qp_clone_state(qp, s->qp) : scp_qp_set_problem(qp, limits, space, p0, v0, pf, vf);
  if (rc != SCP_OK) {
    scp_qp_destroy(qp);
    (void)hipFree(ws);
    return rc;
  }
  scp_qp_destroy(s->qp);
  SV_HIP(hipFree(s->ws));
  s->qp = qp;
  s->ws = ws;
  s->row_cap = cap;
  return SCP_OK;
}

int add_rows_growing(scp_solver* s, int64_t n, int64_t have, bool keep_state, const double* x0, const double* limits,
                     const double* space, const double* p0, const double* v0, const double* pf, const double* vf) {
  if (n <= 0) return SCP_OK;
  int rc = scp_qp_add_rows_from_pass(s->qp, n, s->sel, s->eta, s->l, 0, s->pairs);  // (gathers the rows itself)
  if (rc != SCP_ERR_CAPACITY) return rc;
  SV_CHECK(grow_qp(s, have + n, keep_state, limits, space, p0, v0, pf, vf));
  if (!keep_state) SV_CHECK(scp_qp_reset(s->qp, x0));
  return scp_qp_add_rows_from_pass(s->qp, n, s->sel, s->eta, s->l, 0, s->pairs);
}

// _solve_with_avoidance_constraints (scp.py:399-451): linearise around `acc_in` (whose positions the caller holds in
// s->pos_a), joint QP with exact constraint generation; result in s->x, its positions in s->pos_b.  eps > 0: termination
// tolerances of this one QP (the polish step).
int solve_joint_qp(scp_solver* s, const double* acc_in, const double* limits, const double* space, const double* p0,
                   const double* v0, const double* pf, const double* vf, const scp_solve_options* o, double eps,
                   scp_qp_record* rec) {
  scp_ctx* ctx = s->ctx;
  const int N = s->N, K = s->K, D = s->D;
  const size_t nbytes = (size_t)N * K * D * sizeof(double);
  int max_iter = o->max_iter;
  // the per-QP tolerances (polish) and the per-round iteration budget go into s->st: put the caller's settings back on EVERY
  // exit path, or an error return would leave a pooled solver object with polish tolerances for all later scenarios
  struct RestoreSettings {
    scp_solver* s;
    scp_qp_settings saved;
    ~RestoreSettings() {
      s->st = saved;
      if (s->qp) (void)scp_qp_update_settings(s->qp, &saved);
    }
  } restore{s, s->st};
  if (eps > 0.0) {
    s->st.eps_abs = s->st.eps_rel = eps;
    max_iter = std::max(max_iter, 40000);
  }
  double* prev_pos = s->pos_a;
  for (;;) {
    SV_CHECK(scp_linearize_pairs(ctx, N, K, D, s->R, s->h, 0, s->pairs, prev_pos, p0, v0, s->eta, s->l, o->working_set_margin,
                                 s->sel, s->sel_cap, s->bitmap, s->stats));
    SV_CHECK(read_stats(s, true));
    if ((int64_t)s->h_stats->n_selected <= s->sel_cap) break;
    SV_CHECK(grow_sel(s, (int64_t)s->h_stats->n_selected));
  }
  float lin_ms = 0.f, viol_ms = 0.f;
  if (s->pairs > 0) SV_CHECK(scp_ctx_last_pair_ms(ctx, &lin_ms));  // (that kernel has finished: its stats were read)
  int64_t n = (int64_t)s->h_stats->n_selected;
  SV_CHECK(scp_qp_update_settings(s->qp, &s->st));
  SV_CHECK(scp_qp_reset(s->qp, acc_in));
  SV_CHECK(add_rows_growing(s, n, 0, false, acc_in, limits, space, p0, v0, pf, vf));
  int64_t nW = n;

  int used = 0;
  scp_qp_info info{};
  memset(rec, 0, sizeof(*rec));
  if (o->max_rounds < 1) {  // no round runs: the "solution" is the linearisation point
    SV_HIP(hipMemcpyAsync(s->x, acc_in, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    SV_HIP(hipMemcpyAsync(s->pos_b, s->pos_a, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
  }
  double max_v = 0.0;
  int rounds = 0;
  for (int rnd = 0; rnd < o->max_rounds; ++rnd) {
    s->st.max_iter = std::max(max_iter - used, 1);
    SV_CHECK(scp_qp_update_settings(s->qp, &s->st));
    SV_CHECK(scp_qp_solve(s->qp, &info));
    used += info.iter;
    rec->iter += info.iter;
    rec->cg_iters_total += info.cg_iters_total;
    rec->rho_updates += info.rho_updates;
    rec->solve_ms += info.solve_ms;
    rec->pipeline |= info.pipeline;
    rec->persist_launches += info.persist_launches;
    rec->persist_gave_up += info.persist_gave_up;
    rec->rho_switches_in_kernel += info.rho_switches_in_kernel;
    SV_CHECK(scp_qp_get_solution(s->qp, s->x));
    SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, s->x, p0, v0, s->pos_b, nullptr));
    for (;;) {
      SV_CHECK(scp_collision_violations_at(ctx, N, K, D, s->R, 0, s->pairs, prev_pos, s->pos_b, o->feasibility_tol, s->sel,
                                           s->sel_cap, s->bitmap, s->stats));
      SV_CHECK(read_stats(s, true));
      if ((int64_t)s->h_stats->n_selected <= s->sel_cap) break;
      SV_CHECK(grow_sel(s, (int64_t)s->h_stats->n_selected));
    }
    if (s->pairs > 0) SV_CHECK(scp_ctx_last_pair_ms(ctx, &viol_ms));
    n = (int64_t)s->h_stats->n_selected;
    max_v = s->h_stats->max_violation;
    if (rounds < SCP_MAX_ROUNDS_RECORDED) rec->added[rounds] = n;
    ++rounds;
    if (n == 0 || used >= max_iter) break;
    SV_CHECK(add_rows_growing(s, n, nW, true, nullptr, limits, space, p0, v0, pf, vf));
    nW += n;
  }
  rec->status_val = info.status_val;
  rec->working_rows = info.working_rows;
  rec->r_prim = info.r_prim;
  rec->r_dual = info.r_dual;
  rec->rho = info.rho;
  rec->rounds = rounds;
  rec->unresolved_rows = n;
  rec->max_violation = max_v;
  rec->linearize_ms = lin_ms;
  rec->violations_ms = viol_ms;
  return SCP_OK;
}

}  // namespace

extern "C" void scp_solve_default_options(scp_solve_options* o) {
  if (!o) return;
  o->max_iterations = 15;       // compute_trajectories.py:75, compute_trajectories_batch.py:21
  o->max_rounds = 20;
  o->max_iter0 = 4000;          // OSQP default (scp.py:360)
  o->max_iter = 10000;          // scp.py:442
  o->refresh_feasibility = 0;
  o->polish = 0;
  o->working_set_margin = 0.5;
  o->feasibility_tol = 1e-6;
  o->polish_eps = 1e-8;
  o->convergence_tolerance = 1.5e-2;  // scp.py:52
}

extern "C" void scp_solver_destroy(scp_solver* s) {
  if (!s) return;
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->qp) scp_qp_destroy(s->qp);
  void* dev[] = {s->eta, s->l, s->bitmap, s->sel, s->stats, s->ws, s->acc, s->x, s->pos_a, s->pos_b, s->vel};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (s->h_stats) (void)hipHostFree(s->h_stats);
  if (s->pair_pts) (void)hipHostFree(s->pair_pts);
  delete s;
}

extern "C" int scp_solver_create(scp_ctx* ctx, int N, int K, int D, double h, double R, const scp_qp_settings* st,
                                 int64_t qp_row_capacity, scp_solver** out) {
  if (!ctx) return SCP_ERR_INVALID;
  SCP_REQUIRE(ctx, out && st, "solver_create: null pointer");
  SCP_REQUIRE(ctx, N > 0 && K > 1 && (D == 2 || D == 3) && h > 0, "solver_create: bad shape");
  if (hipSetDevice(ctx->device) != hipSuccess) return scp_fail(ctx, SCP_ERR_HIP, "solver_create: hipSetDevice failed");
  scp_solver* s = new scp_solver();
  memset(s, 0, sizeof(*s));
  s->ctx = ctx; s->N = N; s->K = K; s->D = D; s->h = h; s->R = R; s->st = *st;
  s->pairs = scp_pairs(N);
  s->rows = (int64_t)K * s->pairs;
  s->stride = scp_eta_stride(K, s->pairs);
  s->sel_cap = std::min<int64_t>(std::max<int64_t>(s->rows, 1), std::max<int64_t>(65536, (int64_t)64 * N * K));
  s->row_cap = qp_row_capacity > 0 ? qp_row_capacity
                                   : std::min<int64_t>(s->rows, std::max<int64_t>(8192, (int64_t)32 * N * K));
  if (s->row_cap < 1) s->row_cap = 1;
  const size_t traj = (size_t)N * K * D * sizeof(double);
  bool ok = hipMalloc(&s->eta, (size_t)std::max<int64_t>(D * s->stride, 2) * sizeof(double)) == hipSuccess &&
            hipMalloc(&s->l, (size_t)std::max<int64_t>(s->rows + (s->rows & 1), 2) * sizeof(double)) == hipSuccess &&
            hipMalloc(&s->bitmap, (size_t)std::max<int64_t>((s->rows + 31) / 32, 1) * sizeof(uint32_t)) == hipSuccess &&
            hipMalloc(&s->sel, (size_t)s->sel_cap * sizeof(int64_t)) == hipSuccess &&
            hipMalloc(&s->stats, sizeof(scp_pair_stats)) == hipSuccess &&
            hipHostMalloc(&s->h_stats, sizeof(scp_pair_stats)) == hipSuccess &&
            hipHostMalloc(&s->pair_pts, 2 * 3 * sizeof(double)) == hipSuccess &&
            hipMalloc(&s->acc, traj) == hipSuccess && hipMalloc(&s->x, traj) == hipSuccess &&
            hipMalloc(&s->pos_a, traj) == hipSuccess && hipMalloc(&s->pos_b, traj) == hipSuccess &&
            hipMalloc(&s->vel, traj) == hipSuccess &&
            hipMemsetAsync(s->bitmap, 0, (size_t)std::max<int64_t>((s->rows + 31) / 32, 1) * sizeof(uint32_t), ctx->stream) == hipSuccess;
  if (!ok) {
    scp_solver_destroy(s);
    return scp_fail(ctx, SCP_ERR_HIP, "solver_create: device allocation failed (%lld collision rows)", (long long)s->rows);
  }
  int rc = make_qp(s, s->row_cap, &s->ws, &s->qp);
  if (rc != SCP_OK) {
    scp_solver_destroy(s);
    return rc;
  }
  *out = s;
  return SCP_OK;
}

extern "C" int scp_solver_update_settings(scp_solver* s, const scp_qp_settings* st) {
  if (!s || !st) return SCP_ERR_INVALID;
  int rc = scp_qp_update_settings(s->qp, st);
  if (rc == SCP_OK) s->st = *st;
  return rc;
}

extern "C" int scp_solver_solve(scp_solver* s, const double* limits, const double* space, const double* p0,
                                const double* v0, const double* pf, const double* vf, const scp_solve_options* o,
                                double* acc_out, double* pos_out, double* vel_out, scp_solve_result* res,
                                scp_qp_record* records, int record_capacity) {
  if (!s) return SCP_ERR_INVALID;
  scp_ctx* ctx = s->ctx;
  SCP_REQUIRE(ctx, limits && space && p0 && v0 && pf && vf && o && acc_out && pos_out && vel_out && res && records,
              "solver_solve: null pointer");
  SCP_REQUIRE(ctx, record_capacity >= o->max_iterations + 2, "solver_solve: %d records needed", o->max_iterations + 2);
  const int N = s->N, K = s->K, D = s->D;
  const size_t nbytes = (size_t)N * K * D * sizeof(double);
  memset(res, 0, sizeof(*res));
  res->first_violation = UINT64_MAX;
  const double t_start = now_s();

  // a2 + a3: bounds, QP#0 (scp.py:137-138, :323-369)
  SV_CHECK(scp_qp_set_problem(s->qp, limits, space, p0, v0, pf, vf));
  scp_qp_settings st0 = s->st;
  st0.max_iter = o->max_iter0;
  SV_CHECK(scp_qp_update_settings(s->qp, &st0));
  SV_CHECK(scp_qp_reset(s->qp, nullptr));
  scp_qp_info i0{};
  SV_CHECK(scp_qp_solve(s->qp, &i0));
  scp_qp_record* r0 = &records[0];
  memset(r0, 0, sizeof(*r0));
  r0->status_val = i0.status_val; r0->iter = i0.iter; r0->rho_updates = i0.rho_updates; r0->cg_iters_total = i0.cg_iters_total;
  r0->working_rows = i0.working_rows; r0->r_prim = i0.r_prim; r0->r_dual = i0.r_dual; r0->rho = i0.rho; r0->solve_ms = i0.solve_ms;
  r0->rounds = 1;
  r0->pipeline = i0.pipeline;
  res->n_records = 1;
  if (i0.status_val != 1 && i0.status_val != 2) {  // scp.py:363-365: the caller raises "OSQP failed: <status>"
    res->qp0_status = i0.status_val;
    res->time_sec = now_s() - t_start;
    return SCP_OK;
  }
  res->qp0_status = i0.status_val;
  SV_CHECK(scp_qp_get_solution(s->qp, s->acc));

  // a4 + a8: initial guess, avoidance check evaluated ONCE (scp.py:140-144, never refreshed inside the loop :152)
  SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, s->acc, p0, v0, s->pos_a, nullptr));
  SV_CHECK(scp_check_avoidance(ctx, N, K, D, s->R, 0, s->pairs, s->pos_a, s->stats));
  SV_CHECK(read_stats(s, false));
  bool is_feasible = s->h_stats->first_violation == UINT64_MAX;
  res->first_violation = s->h_stats->first_violation;
  if (!is_feasible && s->pairs > 0) {  // distance of the first violating pair, for the reference's print (scp.py:611-613)
    const int64_t fv = (int64_t)s->h_stats->first_violation;
    const int64_t k = fv / s->pairs, q = fv % s->pairs;
    const double b = 2.0 * N - 1.0;
    int64_t i = (int64_t)((b - std::sqrt(b * b - 8.0 * (double)q)) * 0.5);
    i = std::max<int64_t>(0, std::min<int64_t>(i, N - 2));
    while (i * (2LL * N - i - 1) / 2 > q) --i;
    while (i < N - 2 && (i + 1) * (2LL * N - i - 2) / 2 <= q) ++i;
    const int64_t j = q - i * (2LL * N - i - 1) / 2 + i + 1;
    SV_HIP(hipMemcpyAsync(s->pair_pts, s->pos_a + ((size_t)i * K + k) * D, D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SV_HIP(hipMemcpyAsync(s->pair_pts + 3, s->pos_a + ((size_t)j * K + k) * D, D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SV_HIP(hipStreamSynchronize(ctx->stream));
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) d2 += (s->pair_pts[d] - s->pair_pts[3 + d]) * (s->pair_pts[d] - s->pair_pts[3 + d]);
    res->first_violation_distance = std::sqrt(d2);
    res->first_violation_k = (int32_t)k; res->first_violation_i = (int32_t)i; res->first_violation_j = (int32_t)j;
  }
  res->initially_feasible = is_feasible ? 1 : 0;

  // a1: the SCP loop (scp.py:152-166)
  int iteration = 0;
  bool converged = false;
  double* acc = s->acc;
  while (iteration < o->max_iterations && !converged && !is_feasible) {
    const double t_it = now_s();
    scp_qp_record* rec = &records[res->n_records];
    SV_CHECK(solve_joint_qp(s, acc, limits, space, p0, v0, pf, vf, o, 0.0, rec));
    double rel[3];
    SV_CHECK(scp_rel_step(ctx, (int64_t)N * K * D, s->x, acc, rel));  // scp.py:157-159 (no zero guard)
    rec->rel_step = rel[2];
    rec->time_sec = now_s() - t_it;
    ++res->n_records;
    if (rel[2] <= o->convergence_tolerance) converged = true;
    std::swap(s->acc, s->x);  // the new iterate and its positions become the next linearisation point
    std::swap(s->pos_a, s->pos_b);
    acc = s->acc;
    ++iteration;
    if (o->refresh_feasibility && !converged) {  // opt-in (the reference leaves a TODO, scp.py:150)
      SV_CHECK(scp_check_avoidance(ctx, N, K, D, s->R, 0, s->pairs, s->pos_a, s->stats));
      SV_CHECK(read_stats(s, false));
      is_feasible = s->h_stats->first_violation == UINT64_MAX;
    }
  }
  res->n_iterations = iteration;
  res->converged = converged ? 1 : 0;
  res->feasible_at_exit = is_feasible ? 1 : 0;
  if (o->polish) {
    const double t_p = now_s();
    scp_qp_record* rec = &records[res->n_records];
    SV_CHECK(solve_joint_qp(s, s->acc, limits, space, p0, v0, pf, vf, o, o->polish_eps, rec));
    rec->time_sec = now_s() - t_p;
    rec->rel_step = -1.0;
    ++res->n_records;
    res->polished = 1;
    std::swap(s->acc, s->x);
    std::swap(s->pos_a, s->pos_b);
  }
  // final kinematics (scp.py:169)
  SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, s->acc, p0, v0, pos_out, vel_out));
  SV_HIP(hipMemcpyAsync(acc_out, s->acc, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  res->time_sec = now_s() - t_start;
  return SCP_OK;
}

extern "C" int scp_solver_step(scp_solver* s, const double* limits, const double* space, const double* p0, const double* v0,
                               const double* pf, const double* vf, const scp_solve_options* o, const double* acc_in,
                               double* acc_out, scp_qp_record* rec) {
  if (!s) return SCP_ERR_INVALID;
  scp_ctx* ctx = s->ctx;
  SCP_REQUIRE(ctx, limits && space && p0 && v0 && pf && vf && o && acc_in && acc_out && rec, "solver_step: null pointer");
  const int N = s->N, K = s->K, D = s->D;
  const size_t nbytes = (size_t)N * K * D * sizeof(double);
  const double t0 = now_s();
  SV_CHECK(scp_qp_set_problem(s->qp, limits, space, p0, v0, pf, vf));
  SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, acc_in, p0, v0, s->pos_a, nullptr));
  SV_CHECK(solve_joint_qp(s, acc_in, limits, space, p0, v0, pf, vf, o, 0.0, rec));
  double rel[3];
  SV_CHECK(scp_rel_step(ctx, (int64_t)N * K * D, s->x, acc_in, rel));  // scp.py:157-159
  rec->rel_step = rel[2];
  SV_HIP(hipMemcpyAsync(acc_out, s->x, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  rec->time_sec = now_s() - t0;
  return SCP_OK;
}
