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

// The problem of the step in flight: one SCP iteration is a sequence of phases (linearise / select, joint-QP round,
// violations pass, ... , relative step) with an exchange point between any two of them.  scp_solver_step and the SCP loop run
// the phases back to back over the full pair range; the sharded entry points (scp_solver_shard_*) run the same phases over
// one rank's pair range and leave the exchanges (an allgather of row ids) to the caller.
struct StepState {
  const double *limits, *space, *p0, *v0, *pf, *vf;
  const double* acc_in;
  scp_solve_options o;
  double eps;
  int64_t q_begin, q_end;
  bool row_free;
  int max_iter, used, rounds;
  int64_t nW, last_added;
  double max_v;
  float lin_ms, viol_ms;
  scp_qp_info info;
  scp_qp_settings saved;
  bool active;
  bool fuse_ok;           // all phases run inside one call (solve_joint_qp): a round's solution may stay in the QP's layout
  bool solution_pending;  // ... until the violations pass derives x and its positions itself (small problems)
  bool rel_ready;         // that pass also left scp_rel_step(x, acc_in) behind: rel[]
  bool spec_ok;           // ... and the selection around the new positions in s->sel2 / s->bitmap2 (spec_n rows)
  int64_t spec_n;
  double rel[3];
  double t0;            // wall clock at the start of the step
  double lim_copy[6], space_copy[6];  // sharded steps: the host arrays outlive the call that passed them
};

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
  // the selection of the NEXT linearisation is already in sel / bitmap (small problems: produced by the pass that computed
  // the positions it linearises around); consumed by step_linearize
  bool spec_valid;
  int64_t spec_n;
  double spec_margin;
  bool want_spec;       // scp_solver_solve is running: a next linearisation may follow the current one
  int64_t* sel2;        // the speculative selection's list and bitmap (allocated on first use, swapped in when accepted)
  int64_t sel2_cap;
  uint32_t* bitmap2;
  double rho_start;  // > 0: the joint QP of the next step starts at this rho (options.carry_rho), else at settings.rho
  struct StepState* step;  // the SCP iteration in flight (phases of solve_joint_qp / of the sharded entry points)
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
  if ((from_mirror || s->ctx->last_pass_small) && s->pairs > 0) return scp_ctx_wait_stats(s->ctx, s->h_stats);
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

// the eta / l planes of the row-writing linearisation (24 B per row: 10 GB at 4096 x 50) exist only once a solve asks for
// them (options.row_free == 0)
int ensure_row_planes(scp_solver* s) {
  if (s->eta && s->l) return SCP_OK;
  // ONE allocation for the D eta planes and l: the kernel's three write streams then run 5-6 % faster than into two
  // allocations (tools/pair_align.py: 124 against 131-134 us at 1024 x 50 on the same box)
  const size_t n_eta = (size_t)std::max<int64_t>(s->D * s->stride, 2);
  const size_t n_l = (size_t)std::max<int64_t>(s->rows + (s->rows & 1), 2);
  SV_HIP(hipMalloc(&s->eta, (n_eta + n_l) * sizeof(double)));
  s->l = s->eta + n_eta;  // (D * stride is even: l stays 16-byte aligned)
  return SCP_OK;
}

// rows[0, n) into the QP: gathered from the stored planes (rows = s->sel of a row-writing pass), or (row_free) recomputed
// from the linearisation point
int add_rows_once(scp_solver* s, const int64_t* rows, int64_t n, bool row_free, const double* prev_pos, const double* p0,
                  const double* v0) {
  if (row_free) return scp_qp_add_rows_at(s->qp, n, rows, prev_pos, p0, v0, s->R);
  return scp_qp_add_rows_from_pass(s->qp, n, rows, s->eta, s->l, 0, s->pairs);  // (gathers the rows itself)
}

int add_rows_growing(scp_solver* s, StepState& t, const int64_t* rows, int64_t n, bool keep_state) {
  if (n <= 0) return SCP_OK;
  int rc = add_rows_once(s, rows, n, t.row_free, s->pos_a, t.p0, t.v0);
  if (rc != SCP_ERR_CAPACITY) return rc;
  SV_CHECK(grow_qp(s, t.nW + n, keep_state, t.limits, t.space, t.p0, t.v0, t.pf, t.vf));
  if (!keep_state) SV_CHECK(scp_qp_reset(s->qp, t.acc_in));
  return add_rows_once(s, rows, n, t.row_free, s->pos_a, t.p0, t.v0);
}

// the per-QP tolerances (polish) and the per-round iteration budget go into s->st: the caller's settings come back when the
// step ends -- on EVERY exit path, or an error return would leave a pooled solver object with polish tolerances for all
// later scenarios
void step_restore(scp_solver* s, StepState& t) {
  if (!t.active) return;
  s->st = t.saved;
  if (s->qp) (void)scp_qp_update_settings(s->qp, &t.saved);
  t.active = false;
}
struct StepGuard {  // (for the callers that run all phases inside one function)
  scp_solver* s;
  StepState& t;
  ~StepGuard() { step_restore(s, t); }
};

// phase 1 -- _add_collision_constraints (scp.py:453-557) over the pair range [q_begin, q_end): linearise around `acc_in`,
// whose positions the caller holds in s->pos_a; the selected rows are s->sel[0, *n_sel) (global ids, ascending)
int step_linearize(scp_solver* s, StepState& t, const double* acc_in, const double* limits, const double* space,
                   const double* p0, const double* v0, const double* pf, const double* vf, const scp_solve_options* o,
                   double eps, int64_t q_begin, int64_t q_end, scp_qp_record* rec, int64_t* n_sel) {
  scp_ctx* ctx = s->ctx;
  const int N = s->N, K = s->K, D = s->D;
  step_restore(s, t);  // (a step that was abandoned between two phases)
  t = StepState{};
  t.limits = limits; t.space = space; t.p0 = p0; t.v0 = v0; t.pf = pf; t.vf = vf;
  t.acc_in = acc_in; t.o = *o; t.eps = eps; t.q_begin = q_begin; t.q_end = q_end;
  t.row_free = o->row_free != 0;
  t.max_iter = o->max_iter;
  t.saved = s->st;
  t.active = true;
  t.t0 = now_s();
  if (eps > 0.0) {
    s->st.eps_abs = s->st.eps_rel = eps;
    t.max_iter = std::max(t.max_iter, 40000);
  }
  memset(rec, 0, sizeof(*rec));
  if (!t.row_free) SV_CHECK(ensure_row_planes(s));
  const bool spec = s->spec_valid && t.row_free && q_begin == 0 && q_end == s->pairs && s->spec_margin == o->working_set_margin;
  s->spec_valid = false;
  if (spec) {  // (the pass that produced s->pos_a also selected around it: same list, same bitmap)
    *n_sel = s->spec_n;
    return SCP_OK;
  }
  for (;;) {
    if (t.row_free)
      SV_CHECK(scp_select_pairs(ctx, N, K, D, s->R, q_begin, q_end, s->pos_a, o->working_set_margin, s->sel, s->sel_cap,
                                s->bitmap, s->stats));
    else
      SV_CHECK(scp_linearize_pairs(ctx, N, K, D, s->R, s->h, q_begin, q_end, s->pos_a, p0, v0, s->eta, s->l,
                                   o->working_set_margin, s->sel, s->sel_cap, s->bitmap, s->stats));
    if (q_end <= q_begin) {  // an empty shard: nothing ran, nothing was published
      *n_sel = 0;
      return SCP_OK;
    }
    SV_CHECK(read_stats(s, true));
    if ((int64_t)s->h_stats->n_selected <= s->sel_cap) break;
    SV_CHECK(grow_sel(s, (int64_t)s->h_stats->n_selected));
  }
  SV_CHECK(scp_ctx_last_pair_ms(ctx, &t.lin_ms));  // (that kernel has finished: its stats were read)
  *n_sel = (int64_t)s->h_stats->n_selected;
  return SCP_OK;
}

// phase 2 -- one constraint-generation round of _solve_with_avoidance_constraints (scp.py:399-451): `rows[0, n)` join the
// working set (first round: the QP is reset to the warm start first), ADMM, solution -> s->x, its positions -> s->pos_b
int step_qp_round(scp_solver* s, StepState& t, const int64_t* rows, int64_t n, scp_qp_record* rec) {
  scp_ctx* ctx = s->ctx;
  const int N = s->N, K = s->K, D = s->D;
  const bool first = t.rounds == 0;
  bool rows_in = false;
  if (first) {
    SV_CHECK(scp_qp_update_settings(s->qp, &s->st));
    if (t.row_free && s->rho_start <= 0.0 && n > 0 && n <= s->row_cap) {
      // (reset and the QP's first rows in one launch when the problem is small; the two calls otherwise)
      SV_CHECK(scp_qp_reset_add_rows_at(s->qp, t.acc_in, n, rows, s->pos_a, t.p0, t.v0, s->R));
      rows_in = true;
    } else {
      SV_CHECK(scp_qp_reset(s->qp, t.acc_in));
      if (s->rho_start > 0.0) SV_CHECK(scp_qp_set_rho(s->qp, s->rho_start));
    }
  }
  if (!rows_in) SV_CHECK(add_rows_growing(s, t, rows, n, !first));
  t.nW += n;
  s->st.max_iter = std::max(t.max_iter - t.used, 1);
  SV_CHECK(scp_qp_update_settings(s->qp, &s->st));
  SV_CHECK(scp_qp_solve(s->qp, &t.info));
  t.used += t.info.iter;
  rec->iter += t.info.iter;
  rec->cg_iters_total += t.info.cg_iters_total;
  rec->rho_updates += t.info.rho_updates;
  rec->solve_ms += t.info.solve_ms;
  rec->pipeline |= t.info.pipeline;
  rec->persist_launches += t.info.persist_launches;
  rec->persist_gave_up += t.info.persist_gave_up;
  rec->rho_switches_in_kernel += t.info.rho_switches_in_kernel;
  if (t.fuse_ok) {  // (step_violations materialises s->x and s->pos_b, inside its pass when the problem is small)
    t.solution_pending = true;
    return SCP_OK;
  }
  SV_CHECK(scp_qp_get_solution(s->qp, s->x));
  SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, s->x, t.p0, t.v0, s->pos_b, nullptr));
  return SCP_OK;
}

int materialise_solution(scp_solver* s, StepState& t) {
  t.solution_pending = false;
  SV_CHECK(scp_qp_get_solution(s->qp, s->x));
  return scp_kinematics(s->ctx, s->N, s->K, s->D, s->h, s->x, t.p0, t.v0, s->pos_b, nullptr);
}

// phase 3 -- every row of [q_begin, q_end) checked at the round's solution; the violated rows outside the working set are
// s->sel[0, *n_new)
int step_violations(scp_solver* s, StepState& t, int64_t* n_new) {
  scp_ctx* ctx = s->ctx;
  *n_new = 0;
  if (t.q_end <= t.q_begin) {  // no pairs (a single agent, an empty shard): the pass's neutral element
    if (t.solution_pending) SV_CHECK(materialise_solution(s, t));
    t.max_v = -INFINITY;
    return SCP_OK;
  }
  for (;;) {
    if (t.solution_pending) {
      bool fused = false;
      const bool spec = s->want_spec && t.row_free && t.q_begin == 0 && t.q_end == s->pairs;
      if (spec && !s->sel2) {  // (first use: the second list / bitmap of the speculative selection)
        const size_t words = (size_t)std::max<int64_t>((s->rows + 31) / 32, 1);
        SV_HIP(hipMalloc(&s->sel2, (size_t)s->sel_cap * sizeof(int64_t)));
        s->sel2_cap = s->sel_cap;
        SV_HIP(hipMalloc(&s->bitmap2, words * sizeof(uint32_t)));
      }
      SV_CHECK(scp_violations_from_solution(ctx, s->N, s->K, s->D, s->R, s->h, t.q_begin, t.q_end, s->pos_a,
                                            scp_qp_solution_tm(s->qp), t.p0, t.v0, s->x, s->pos_b, t.o.feasibility_tol,
                                            s->sel, s->sel_cap, s->bitmap, s->stats, t.acc_in, spec ? s->sel2 : nullptr,
                                            s->sel2_cap, s->bitmap2, t.o.working_set_margin, &fused));
      t.solution_pending = false;  // (s->x and s->pos_b exist from here on, also for a repeat with a longer list)
      if (fused) {
        SV_CHECK(read_stats(s, true));
        scp_ctx_mirror_rel(ctx, (int64_t)s->N * s->K * s->D, t.rel);
        t.rel_ready = true;
        t.spec_n = (int64_t)ctx->h_mirror->n_spec;
        t.spec_ok = spec && t.spec_n > 0 && t.spec_n <= s->sel2_cap;
        if ((int64_t)s->h_stats->n_selected <= s->sel_cap) break;
        SV_CHECK(grow_sel(s, (int64_t)s->h_stats->n_selected));
        continue;
      }
      SV_CHECK(scp_qp_get_solution(s->qp, s->x));
      SV_CHECK(scp_kinematics(ctx, s->N, s->K, s->D, s->h, s->x, t.p0, t.v0, s->pos_b, nullptr));
    }
    SV_CHECK(scp_collision_violations_at(ctx, s->N, s->K, s->D, s->R, t.q_begin, t.q_end, s->pos_a, s->pos_b,
                                         t.o.feasibility_tol, s->sel, s->sel_cap, s->bitmap, s->stats));
    SV_CHECK(read_stats(s, true));
    if ((int64_t)s->h_stats->n_selected <= s->sel_cap) break;
    SV_CHECK(grow_sel(s, (int64_t)s->h_stats->n_selected));
  }
  SV_CHECK(scp_ctx_last_pair_ms(ctx, &t.viol_ms));
  *n_new = (int64_t)s->h_stats->n_selected;
  t.max_v = s->h_stats->max_violation;
  return SCP_OK;
}

// bookkeeping after a violations pass that found n_all rows (over all ranks); returns whether another round follows
bool step_round_done(StepState& t, int64_t n_all, scp_qp_record* rec) {
  if (t.rounds < SCP_MAX_ROUNDS_RECORDED) rec->added[t.rounds] = n_all;
  ++t.rounds;
  t.last_added = n_all;
  return !(n_all == 0 || t.used >= t.max_iter || t.rounds >= t.o.max_rounds);
}

void step_finish(scp_solver* s, StepState& t, scp_qp_record* rec) {
  rec->status_val = t.info.status_val;
  rec->working_rows = t.info.working_rows;
  rec->r_prim = t.info.r_prim;
  rec->r_dual = t.info.r_dual;
  rec->rho = t.info.rho;
  rec->rounds = t.rounds;
  rec->unresolved_rows = t.last_added;
  rec->max_violation = t.max_v;
  rec->linearize_ms = t.lin_ms;
  rec->violations_ms = t.viol_ms;
  step_restore(s, t);
}

// _solve_with_avoidance_constraints (scp.py:399-451) on ONE rank: all phases back to back over the full pair range.
// Linearises around `acc_in` (positions in s->pos_a); result in s->x, its positions in s->pos_b.  eps > 0: termination
// tolerances of this one QP (the polish step).
int solve_joint_qp(scp_solver* s, const double* acc_in, const double* limits, const double* space, const double* p0,
                   const double* v0, const double* pf, const double* vf, const scp_solve_options* o, double eps,
                   scp_qp_record* rec) {
  StepState& t = *s->step;
  StepGuard guard{s, t};
  const size_t nbytes = (size_t)s->N * s->K * s->D * sizeof(double);
  int64_t n = 0;
  SV_CHECK(step_linearize(s, t, acc_in, limits, space, p0, v0, pf, vf, o, eps, 0, s->pairs, rec, &n));
  t.fuse_ok = true;
  if (o->max_rounds < 1) {  // no round runs: the "solution" is the linearisation point
    SV_CHECK(scp_qp_update_settings(s->qp, &s->st));
    SV_CHECK(scp_qp_reset(s->qp, acc_in));
    SV_CHECK(add_rows_growing(s, t, s->sel, n, false));
    SV_HIP(hipMemcpyAsync(s->x, acc_in, nbytes, hipMemcpyDeviceToDevice, s->ctx->stream));
    SV_HIP(hipMemcpyAsync(s->pos_b, s->pos_a, nbytes, hipMemcpyDeviceToDevice, s->ctx->stream));
    t.last_added = n;
  }
  for (bool more = o->max_rounds >= 1; more;) {
    SV_CHECK(step_qp_round(s, t, s->sel, n, rec));
    SV_CHECK(step_violations(s, t, &n));
    more = step_round_done(t, n, rec);
  }
  step_finish(s, t, rec);
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
  o->row_free = 1;
  o->carry_rho = 0;
}

extern "C" void scp_solver_destroy(scp_solver* s) {
  if (!s) return;
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->qp) scp_qp_destroy(s->qp);
  void* dev[] = {s->eta, s->bitmap, s->sel, s->bitmap2, s->sel2, s->stats, s->ws, s->acc, s->x, s->pos_a, s->pos_b, s->vel};  // (l lives in eta's allocation)
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (s->h_stats) (void)hipHostFree(s->h_stats);
  if (s->pair_pts) (void)hipHostFree(s->pair_pts);
  delete s->step;
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
  s->step = new StepState();
  s->pairs = scp_pairs(N);
  s->rows = (int64_t)K * s->pairs;
  s->stride = scp_eta_stride(K, s->pairs);
  s->sel_cap = std::min<int64_t>(std::max<int64_t>(s->rows, 1), std::max<int64_t>(65536, (int64_t)64 * N * K));
  s->row_cap = qp_row_capacity > 0 ? qp_row_capacity
                                   : std::min<int64_t>(s->rows, std::max<int64_t>(8192, (int64_t)32 * N * K));
  if (s->row_cap < 1) s->row_cap = 1;
  const size_t traj = (size_t)N * K * D * sizeof(double);
  bool ok = hipMalloc(&s->bitmap, (size_t)std::max<int64_t>((s->rows + 31) / 32, 1) * sizeof(uint32_t)) == hipSuccess &&
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
  memset(res, 0, sizeof(*res));
  res->first_violation = UINT64_MAX;
  const double t_start = now_s();
  s->spec_valid = false;
  s->want_spec = true;

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

  // a4 + a8: initial guess, avoidance check evaluated ONCE (scp.py:140-144, never refreshed inside the loop :152)
  bool from_solution = false;
  if (o->row_free)  // small problems: solution -> acc, positions, the a8 statistics AND the first selection in one launch
    SV_CHECK(scp_select_from_solution(ctx, N, K, D, s->R, s->h, 0, s->pairs, scp_qp_solution_tm(s->qp), p0, v0, s->acc, s->pos_a,
                                      o->working_set_margin, s->sel, s->sel_cap, s->bitmap, s->stats, &from_solution));
  if (from_solution) {
    SV_CHECK(read_stats(s, true));
    s->spec_valid = (int64_t)s->h_stats->n_selected <= s->sel_cap;
    s->spec_n = (int64_t)s->h_stats->n_selected;
    s->spec_margin = o->working_set_margin;
  } else {
    SV_CHECK(scp_qp_get_solution(s->qp, s->acc));
    SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, s->acc, p0, v0, s->pos_a, nullptr));
    SV_CHECK(scp_check_avoidance(ctx, N, K, D, s->R, 0, s->pairs, s->pos_a, s->stats));
    SV_CHECK(read_stats(s, false));
  }
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
    if (from_solution) {  // (the pass left the two points in the mirror)
      memcpy(s->pair_pts, ctx->h_mirror->pts, 6 * sizeof(double));
    } else {
      SV_HIP(hipMemcpyAsync(s->pair_pts, s->pos_a + ((size_t)i * K + k) * D, D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      SV_HIP(hipMemcpyAsync(s->pair_pts + 3, s->pos_a + ((size_t)j * K + k) * D, D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      SV_HIP(hipStreamSynchronize(ctx->stream));
    }
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
  s->rho_start = 0.0;
  while (iteration < o->max_iterations && !converged && !is_feasible) {
    const double t_it = now_s();
    scp_qp_record* rec = &records[res->n_records];
    SV_CHECK(solve_joint_qp(s, acc, limits, space, p0, v0, pf, vf, o, 0.0, rec));
    if (o->carry_rho) s->rho_start = rec->rho;  // the next linearisation's QP starts where this one ended
    double rel[3];
    if (s->step->rel_ready) memcpy(rel, s->step->rel, sizeof(rel));  // (left behind by the last violations pass: small problems)
    else SV_CHECK(scp_rel_step(ctx, (int64_t)N * K * D, s->x, acc, rel));  // scp.py:157-159 (no zero guard)
    rec->rel_step = rel[2];
    rec->time_sec = now_s() - t_it;
    ++res->n_records;
    if (rel[2] <= o->convergence_tolerance) converged = true;
    std::swap(s->acc, s->x);  // the new iterate and its positions become the next linearisation point
    std::swap(s->pos_a, s->pos_b);
    acc = s->acc;
    ++iteration;
    if (s->step->spec_ok) {  // the round's last pass also selected around these positions: the next linearisation is done
      std::swap(s->sel, s->sel2);
      std::swap(s->sel_cap, s->sel2_cap);
      std::swap(s->bitmap, s->bitmap2);
      s->spec_valid = true;
      s->spec_n = s->step->spec_n;
      s->spec_margin = o->working_set_margin;
    }
    if (o->refresh_feasibility && !converged) {  // opt-in (the reference leaves a TODO, scp.py:150)
      SV_CHECK(scp_check_avoidance(ctx, N, K, D, s->R, 0, s->pairs, s->pos_a, s->stats));
      SV_CHECK(read_stats(s, false));
      is_feasible = s->h_stats->first_violation == UINT64_MAX;
    }
  }
  s->rho_start = 0.0;  // (the polish QP and later solves start at settings.rho)
  s->want_spec = false;  // (no linearisation follows the polish QP's)
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
  SV_CHECK(scp_launch_kinematics_copy(ctx, N, K, D, s->h, s->acc, p0, v0, pos_out, vel_out, acc_out));
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
  s->rho_start = 0.0;
  s->spec_valid = false;
  s->want_spec = false;
  SV_CHECK(scp_qp_set_problem(s->qp, limits, space, p0, v0, pf, vf));
  SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, acc_in, p0, v0, s->pos_a, nullptr));
  SV_CHECK(solve_joint_qp(s, acc_in, limits, space, p0, v0, pf, vf, o, 0.0, rec));
  double rel[3];
  if (s->step->rel_ready) memcpy(rel, s->step->rel, sizeof(rel));
  else SV_CHECK(scp_rel_step(ctx, (int64_t)N * K * D, s->x, acc_in, rel));  // scp.py:157-159
  rec->rel_step = rel[2];
  SV_HIP(hipMemcpyAsync(acc_out, s->x, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  rec->time_sec = now_s() - t0;
  return SCP_OK;
}

// ---- one SCP iteration split at its exchange points (one process per GPU, a pair range per rank) --------------------------
// the row ids the latest selection / violations phase of this rank found (ascending), as many as fit into rows_out
extern "C" int scp_solver_shard_rows(scp_solver* s, int64_t* rows_out, int64_t rows_cap) {
  if (!s) return SCP_ERR_INVALID;
  const int64_t n = s->step->q_end > s->step->q_begin ? std::min<int64_t>((int64_t)s->h_stats->n_selected, rows_cap) : 0;
  if (n > 0)
    SV_HIP(hipMemcpyAsync(rows_out, s->sel, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, s->ctx->stream));
  return SCP_OK;
}

// begin: bounds, positions of the linearisation point (pos_in: the allgathered per-shard trajectories; NULL: computed here
// from acc_in), row-free selection over [q_begin, q_end) -> *rows_local (device, ascending global ids), *n_local
extern "C" int scp_solver_shard_begin(scp_solver* s, const double* limits, const double* space, const double* p0,
                                      const double* v0, const double* pf, const double* vf, const scp_solve_options* o,
                                      const double* acc_in, const double* pos_in, int64_t q_begin, int64_t q_end,
                                      scp_qp_record* rec, int64_t* rows_out, int64_t rows_cap, int64_t* n_local) {
  if (!s) return SCP_ERR_INVALID;
  scp_ctx* ctx = s->ctx;
  SCP_REQUIRE(ctx, limits && space && p0 && v0 && pf && vf && o && acc_in && rec && n_local && (rows_out || rows_cap == 0),
              "solver_shard_begin: null pointer");
  SCP_REQUIRE(ctx, o->row_free != 0, "solver_shard_begin: sharded steps are row-free (a rank holds no other rank's rows)");
  SCP_REQUIRE(ctx, q_begin >= 0 && q_end >= q_begin && q_end <= s->pairs, "solver_shard_begin: bad pair range");
  const int N = s->N, K = s->K, D = s->D;
  StepState& t = *s->step;
  step_restore(s, t);
  s->rho_start = 0.0;
  s->spec_valid = false;
  s->want_spec = false;
  double lim[6], spc[6];
  memcpy(lim, limits, sizeof(lim));
  memcpy(spc, space, 2 * D * sizeof(double));
  SV_CHECK(scp_qp_set_problem(s->qp, limits, space, p0, v0, pf, vf));
  if (pos_in)
    SV_HIP(hipMemcpyAsync(s->pos_a, pos_in, (size_t)N * K * D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  else
    SV_CHECK(scp_kinematics(ctx, N, K, D, s->h, acc_in, p0, v0, s->pos_a, nullptr));
  int64_t n = 0;
  int rc = step_linearize(s, t, acc_in, limits, space, p0, v0, pf, vf, o, 0.0, q_begin, q_end, rec, &n);
  memcpy(t.lim_copy, lim, sizeof(lim));  // (step_linearize re-initialised the state)
  memcpy(t.space_copy, spc, sizeof(spc));
  t.limits = t.lim_copy;
  t.space = t.space_copy;
  if (rc != SCP_OK) {
    step_restore(s, t);
    return rc;
  }
  *n_local = n;
  return scp_solver_shard_rows(s, rows_out, rows_cap);
}

// one constraint-generation round: rows[0, n) = the rows ALL ranks found (ascending), into the replicated working set; ADMM
extern "C" int scp_solver_shard_qp(scp_solver* s, const int64_t* rows, int64_t n, scp_qp_record* rec) {
  if (!s) return SCP_ERR_INVALID;
  StepState& t = *s->step;
  if (!t.active) return scp_fail(s->ctx, SCP_ERR_STATE, "solver_shard_qp: call scp_solver_shard_begin first");
  SCP_REQUIRE(s->ctx, rec && (rows || n == 0), "solver_shard_qp: null pointer");
  int rc = step_qp_round(s, t, rows, n, rec);
  if (rc != SCP_OK) step_restore(s, t);
  return rc;
}

// this rank's rows of the violations pass at the round's solution -> *rows_local, *n_local, *max_violation
extern "C" int scp_solver_shard_violations(scp_solver* s, int64_t* rows_out, int64_t rows_cap, int64_t* n_local,
                                           double* max_violation) {
  if (!s) return SCP_ERR_INVALID;
  StepState& t = *s->step;
  if (!t.active) return scp_fail(s->ctx, SCP_ERR_STATE, "solver_shard_violations: call scp_solver_shard_begin first");
  SCP_REQUIRE(s->ctx, n_local && max_violation && (rows_out || rows_cap == 0), "solver_shard_violations: null pointer");
  int64_t n = 0;
  t.max_v = -INFINITY;
  int rc = step_violations(s, t, &n);
  if (rc != SCP_OK) {
    step_restore(s, t);
    return rc;
  }
  *n_local = n;
  *max_violation = t.max_v;
  return scp_solver_shard_rows(s, rows_out, rows_cap);
}

// after the exchange: n_all rows were found by all ranks together; *more = another round follows (same decision on every
// rank: the QP is replicated and deterministic)
extern "C" int scp_solver_shard_round_done(scp_solver* s, int64_t n_all, double max_violation_all, scp_qp_record* rec,
                                           int* more) {
  if (!s) return SCP_ERR_INVALID;
  StepState& t = *s->step;
  if (!t.active) return scp_fail(s->ctx, SCP_ERR_STATE, "solver_shard_round_done: no step in flight");
  SCP_REQUIRE(s->ctx, rec && more, "solver_shard_round_done: null pointer");
  t.max_v = max_violation_all;
  *more = step_round_done(t, n_all, rec) ? 1 : 0;
  return SCP_OK;
}

// relative step, new accelerations -> acc_out, record completed, the caller's settings restored
extern "C" int scp_solver_shard_end(scp_solver* s, double* acc_out, scp_qp_record* rec) {
  if (!s) return SCP_ERR_INVALID;
  scp_ctx* ctx = s->ctx;
  StepState& t = *s->step;
  if (!t.active) return scp_fail(ctx, SCP_ERR_STATE, "solver_shard_end: no step in flight");
  SCP_REQUIRE(ctx, acc_out && rec, "solver_shard_end: null pointer");
  StepGuard guard{s, t};
  const size_t nbytes = (size_t)s->N * s->K * s->D * sizeof(double);
  double rel[3];
  SV_CHECK(scp_rel_step(ctx, (int64_t)s->N * s->K * s->D, s->x, t.acc_in, rel));  // scp.py:157-159
  const double t0 = t.t0;
  step_finish(s, t, rec);
  rec->rel_step = rel[2];
  SV_HIP(hipMemcpyAsync(acc_out, s->x, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  rec->time_sec = now_s() - t0;
  return SCP_OK;
}
