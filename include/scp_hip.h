/*
 * scp_hip.h -- C-ABI of the MI355X (gfx950) SCP hot path: libscp_hip.so
 *
 * Drop-in boundary for /root/reference/src/path_planning/solvers/scp.py (class SCP).  The reference is
 * pure Python with no FFI of its own; each entry point below replaces the body of
 * one reference method and is what a ctypes binding inside that method would call (INTEGRATION.md shows the
 * stub).  All `double*` / `int64_t*` arguments are DEVICE pointers unless marked [host]; the caller owns
 * them (torch-ROCm tensors in our host code).  No torch types cross this boundary.
 *
 * Conventions
 *   - boundary arrays use the reference layout: accelerations/positions/velocities are [N][K][D]
 *     (flat index (i*K + k)*D + d, scp.py:15-26, :168, :581-582); states are [N][D].
 *   - D is 2 (the reference) or 3 (extension).
 *   - collision rows are numbered as the reference orders them: r = k*pairs + q, q = lexicographic index of
 *     (i, j), i < j  (scp.py:487-496); pairs = N(N-1)/2.
 *   - every function returns 0 on success or a negative scp_status; scp_last_error(ctx) gives the text.
 *   - all work is enqueued on the ctx's HIP stream; functions that return host values synchronise it.
 *   - a ctx and the objects created from it must be used from one thread at a time (one ctx per rank).
 */
#ifndef SCP_HIP_H
#define SCP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCP_ABI_VERSION 5

typedef enum scp_status {
  SCP_OK = 0,
  SCP_ERR_INVALID = -1,   /* bad argument (shape, NULL pointer, D not in {2,3}) */
  SCP_ERR_HIP = -2,       /* a HIP runtime call failed */
  SCP_ERR_CAPACITY = -3,  /* a caller-provided buffer is too small (row capacity, workspace) */
  SCP_ERR_STATE = -4      /* call order violated (e.g. solve before set_problem) */
} scp_status;

typedef struct scp_ctx scp_ctx;
typedef struct scp_qp scp_qp;

/* Result of a pairwise pass; lives in DEVICE memory (32 bytes), written by scp_linearize_pairs,
 * scp_check_avoidance and scp_collision_violations. */
typedef struct scp_pair_stats {
  double min_dist;            /* min over processed rows of ||p_i[k] - p_j[k]|| (before the dist:=1 rule) */
  uint64_t first_violation;   /* smallest row id with dist < R - 0.01 (scp.py:610), UINT64_MAX if none */
  uint64_t n_selected;        /* rows selected (may exceed the capacity of the output list: then only the first
                                 `capacity` ids were stored, the call still returns SCP_OK and the caller repeats it
                                 with a larger list) */
  double max_violation;       /* scp_collision_violations: max over rows of l_r - (A x)_r */
} scp_pair_stats;

/* Settings of the QP solver.  Defaults (scp_qp_default_settings) are OSQP's, because the reference calls
 * osqp with its defaults (scp.py:360) resp. max_iter=10000 (scp.py:442). */
typedef struct scp_qp_settings {
  double rho;                    /* 0.1 */
  double sigma;                  /* 1e-6 */
  double alpha;                  /* 1.6 */
  double rho_eq_scale;           /* 1e3: rho multiplier on equality rows */
  double eps_abs;                /* 1e-3 */
  double eps_rel;                /* 1e-3 */
  int32_t max_iter;              /* 4000 (QP#0, scp.py:360) / 10000 (scp.py:442) */
  int32_t check_termination;     /* 25 */
  int32_t adaptive_rho;          /* 1 */
  int32_t adaptive_rho_interval; /* 50 (iterations; a multiple of check_termination.  OSQP's default is a wall-clock rule;
                                    50 is the measured choice of profiles/r03_rho_interval_sweep.txt) */
  double adaptive_rho_tolerance; /* 5 */
  int32_t cg_iters;              /* 1: PCG steps per ADMM step (fixed count >= 1, warm started at x) */
  int32_t use_mfma;              /* 1: fused column-block kernels, every K-dimension product on
                                    v_mfma_f64_16x16x4_f64 (K <= 120; larger K falls back to 2);
                                    2: one MFMA product per launch (generic path); 0: VALU products */
  double rho_col_scale;          /* 10: rho of the collision rows = rho * rho_col_scale (like OSQP's per-row rho,
                                    which the reference gets x 1e3 on equality rows only).  Measured at 1024 x 50:
                                    the same SCP iterates with 3.5 x fewer ADMM iterations than scale 1 */
  double eps_prim_inf;           /* 1e-4: OSQP's primal infeasibility tolerance; the certificate (delta-y test) is
                                    evaluated at every termination check; <= 0 disables it */
  int32_t persistent;            /* 1: with cg_iters == 1, K <= 64 and at most one block of agents per compute unit, all ADMM
                                    steps between two termination checks run in ONE persistent launch (solver state on chip,
                                    two grid-wide exchanges per step).  Which kernel: 2-D up to 2048 agents the lean kernel
                                    with 8 agents per workgroup, up to 4096 with 16; 3-D up to 1024 agents the 4-agent kernel
                                    of round 2, up to 2048 the lean one with 8.  2 / 3 / 4: force the lean 16-agent / lean
                                    8-agent / round-2 kernel where it fits (tests, measurements); 0: three launches per step.
                                    Same arithmetic up to the association of sums */
  int32_t check_fine;            /* 5: adaptive check cadence.  After a termination check that finds both residuals within
                                    check_fine_ratio x their tolerances, or that changes rho, the next check comes after
                                    check_fine steps (a divisor of check_termination) instead of check_termination: a QP
                                    no longer pays up to 24 surplus steps for every check interval it does not need, and
                                    pays the fine checks (about one step each inside the persistent kernels) only near the
                                    end (profiles/r03_check_cadence.txt).  Applies to QPs with collision rows (QP#0 keeps
                                    the fixed cadence: a better converged start saves the first joint QP more) of up to
                                    4096 columns (beyond, it measured slower: profiles/r03_check_cadence.txt).  0, or a value that does not
                                    divide check_termination: fixed cadence */
  double check_fine_ratio;       /* 2 */
} scp_qp_settings;

/* [host] result of scp_qp_solve */
typedef struct scp_qp_info {
  int32_t status_val;   /* OSQP codes: 1 solved, 2 solved inaccurate (max_iter reached, 10 x eps met),
                           -2 maximum iterations reached, -3 primal infeasible */
  int32_t iter;         /* ADMM iterations of this call */
  int32_t rho_updates;
  int32_t cg_iters_total;
  int64_t working_rows;
  double r_prim, r_dual;
  double rho;
  double solve_ms;      /* device time of this call (HIP events on the ctx stream) */
  int32_t pipeline;     /* which pipelines ran the ADMM iterations of this call: OR of (1 << scp_qp_pipeline) */
  int32_t persist_launches;       /* persistent launches of this call that ran (their steps are in `iter`) */
  int32_t persist_gave_up;        /* persistent launches that timed out on a workgroup and left without writing state
                                     back; the iterations were repeated on the three-launch pipeline */
  int32_t rho_switches_in_kernel; /* adaptive-rho updates made inside a persistent launch (included in rho_updates) */
} scp_qp_info;

/* the pipelines of scp_qp_solve (bit numbers of scp_qp_info.pipeline / scp_qp_record.pipeline) */
typedef enum scp_qp_pipeline {
  SCP_PIPE_QP0 = 0,        /* fixed rows only: all steps up to a check in one column-local launch */
  SCP_PIPE_PERSIST = 1,    /* persistent single-step kernel, one wave per agent, 16/D agents per workgroup */
  SCP_PIPE_PERSIST16 = 2,  /* its lean form: 16 agents per workgroup (2-D, 2048 < N <= 4096) */
  SCP_PIPE_CG1 = 3,        /* single-step pipeline, three launches per ADMM step */
  SCP_PIPE_CG1_BIGK = 4,   /* the same with one workgroup per column (K > 120) */
  SCP_PIPE_FUSED = 5,      /* cg_iters > 1: fused column-block chains */
  SCP_PIPE_GENERIC = 6,    /* one product per launch */
  SCP_PIPE_PERSIST8L = 7   /* the lean kernel's state diet with 8 agents per workgroup (3-D, 1024 < N <= 2048) */
} scp_qp_pipeline;

int scp_abi_version(void);
/* How host threads wait for a kernel's completion word (process-wide): 0 = spin (default: lowest latency, one core per
 * waiting thread), 1 = spin ~20 us, then poll from 20 us sleeps (many solver threads on few cores), 2 = sleep between polls
 * from the first miss (more solver threads than cores). */
void scp_set_host_wait(int mode);

/* Plane stride (in doubles) of the SoA eta array of scp_linearize_pairs: K*nq rounded up to an even count so
 * that every plane starts 16-byte aligned. */
static inline int64_t scp_eta_stride(int64_t K, int64_t nq) { return (K * nq + 1) & ~(int64_t)1; }

/* ---- context -------------------------------------------------------------------------------------- */
int scp_ctx_create(int device, void* hip_stream /* hipStream_t or NULL for the default stream */, scp_ctx** out);
void scp_ctx_destroy(scp_ctx* ctx);
const char* scp_last_error(const scp_ctx* ctx);
int scp_ctx_synchronize(scp_ctx* ctx);
/* device time (ms) of the most recent pairwise kernel launch alone (scp_linearize_pairs, scp_check_avoidance,
 * scp_collision_violations), from HIP events recorded on the ctx stream around that launch; synchronises. */
int scp_ctx_last_pair_ms(scp_ctx* ctx, float* ms);
/* Per-context switches; results never depend on them.
 *   "kernel_timing" (default 1): HIP events around the pairwise kernels and the QP solves -- two queue packets each, ~25 per
 *     complete solve.  0: none are recorded; scp_ctx_last_pair_ms and the records' linearize_ms / violations_ms then read 0
 *     and solve_ms is the host's wall clock around the solve.  For many concurrent solver streams on one GPU
 *     (compute-trajectories-batch), where every packet of a stream costs dispatch latency.
 *   "single_launch_passes" (default 1): pairwise passes of small problems (<= 2 M collision rows) run as ONE launch -- staging
 *     from the [N][K][D] arrays, reduction, sorted row list and host-visible stats in the pass kernel's last workgroup.
 *     0: prep kernel + pass + compaction launches as for large problems. */
int scp_ctx_set_option(scp_ctx* ctx, const char* key, int value);

/* ---- a4 / a7: SCP._compute_positions_velocities (scp.py:371-397),
 *               SCP._accelerations_to_positions_velocities (scp.py:559-595) ---------------------------
 * pos[i][k] = p0_i + (h*k) v0_i + sum_{j<k} (h*h*(k-j-0.5)) a_i[j],  vel[i][k] = v0_i + sum_{j<k} h a_i[j],
 * summed in the reference's order without FMA contraction (bitwise equal to the reference). */
int scp_kinematics(scp_ctx* ctx, int N, int K, int D, double h, const double* acc, const double* p0,
                   const double* v0, double* pos_out, double* vel_out);

/* ---- a2: bounds of SCP._precompute_constraint_matrices (scp.py:182-257) -------------------------------
 * l_out/u_out have N*D*(4K-1) entries in the reference's stacking order jerk, acc, vel, pos (scp.py:342-358).
 * limits = {vel_min, vel_max, acc_min, acc_max, jerk_min, jerk_max} [host]; space = {min_0..min_{D-1},
 * max_0..max_{D-1}} [host]. */
int scp_fixed_bounds(scp_ctx* ctx, int N, int K, int D, double h, const double* limits, const double* space,
                     const double* p0, const double* v0, const double* pf, const double* vf, double* l_out,
                     double* u_out);

/* ---- a5 (+a8 fused): SCP._add_collision_constraints (scp.py:453-557) ---------------------------------
 * Linearises the pairs q in [q_begin, q_end) at every k around pos_prev and writes the COMPACT form of
 * A_collision / l_collision: eta_out[d*scp_eta_stride(K,nq) + k*nq + (q-q_begin)] (SoA planes, nq = q_end-q_begin,
 * D*scp_eta_stride doubles, 16-byte aligned) and l_out[k*nq + (q-q_begin)].
 * Row r of the reference's matrix is  +eta_r[d] h^2 (k-m-.5) on a_i[m], -(same) on a_j[m], m < k; u = +inf.
 * Degenerate pairs (dist < 1e-6): eta = e_0, dist := 1 (the reference draws a random direction, scp.py:505).
 * Fused reductions: stats->min_dist, stats->first_violation (dist < R-0.01), and the rows with
 * dist - R < margin are appended (global row ids r = k*pairs + q) to sel_rows (capacity sel_cap) and marked
 * in sel_bitmap (one bit per LOCAL row k*nq + (q-q_begin), ceil(K*nq/32) uint32 words, zeroed by this call). */
int scp_linearize_pairs(scp_ctx* ctx, int N, int K, int D, double R, double h, int64_t q_begin, int64_t q_end,
                        const double* pos_prev, const double* p0, const double* v0, double* eta_out,
                        double* l_out, double margin, int64_t* sel_rows, int64_t sel_cap, uint32_t* sel_bitmap,
                        scp_pair_stats* stats);

/* The same pass WITHOUT the row stream ("row-free" linearisation): identical distances, selection test (dist - R < margin),
 * sel_rows / sel_bitmap / stats as scp_linearize_pairs, but eta / l are not written (24 B per row: 630 MB at 1024 x 50,
 * 10 GB at 4096 x 50, of which the loop consumes the ~0.05 % selected rows).  The consumer recomputes the selected rows
 * with scp_qp_add_rows_at; scp_collision_violations_at never needed the stored rows. */
int scp_select_pairs(scp_ctx* ctx, int N, int K, int D, double R, int64_t q_begin, int64_t q_end, const double* pos_prev,
                     double margin, int64_t* sel_rows, int64_t sel_cap, uint32_t* sel_bitmap, scp_pair_stats* stats);

/* ---- a8: SCP._fast_check_avoidance_constraints (scp.py:597-615) --------------------------------------
 * stats->first_violation = first row (k -> i -> j order) with ||p_i - p_j|| < R - 0.01, stats->min_dist. */
int scp_check_avoidance(scp_ctx* ctx, int N, int K, int D, double R, int64_t q_begin, int64_t q_end,
                        const double* pos, scp_pair_stats* stats);

/* ---- constraint generation for the joint QP (a6): full pass over the linearised rows ------------------
 * For every local row not yet marked in sel_bitmap: if (A_col x)_r < l_r - feas_tol, mark it and append its
 * global id to new_rows.  (A_col x)_r = eta_r . ((pos_i - c_i) - (pos_j - c_j))[k], c = p0 + k h v0,
 * pos = kinematics(x).  stats->n_selected = rows found, stats->max_violation.  If n_selected exceeds new_cap NOTHING is
 * merged into sel_bitmap and new_rows is left untouched: repeat the call with a longer list. */
int scp_collision_violations(scp_ctx* ctx, int N, int K, int D, double h, int64_t q_begin, int64_t q_end,
                             const double* eta, const double* l_col, const double* pos, const double* p0,
                             const double* v0, double feas_tol, int64_t* new_rows, int64_t new_cap,
                             uint32_t* sel_bitmap, scp_pair_stats* stats);

/* The same pass without reading the stored rows back: eta_r and R - dist_r are recomputed from the linearisation
 * point pos_prev (the arithmetic of scp_linearize_pairs) and
 *   l_r - (A_col x)_r = (R - dist_r) - eta_r . ((pos_new - pos_prev)_i - (pos_new - pos_prev)_j)[k]
 * (the free motion c cancels), so the pass streams nothing from HBM (the stored rows cost 8 (D + 1) bytes per row to
 * read: 630 MB at 1024 x 50).  Differs from scp_collision_violations by rounding only (one fused sum instead of two). */
int scp_collision_violations_at(scp_ctx* ctx, int N, int K, int D, double R, int64_t q_begin, int64_t q_end,
                                const double* pos_prev, const double* pos_new, double feas_tol, int64_t* new_rows,
                                int64_t new_cap, uint32_t* sel_bitmap, scp_pair_stats* stats);

/* Gather the compact rows `rows` (global ids, all inside [q_begin,q_end) x K) out of (eta, l_col):
 * w_eta[n][D] (AoS) and w_l[n]. */
int scp_gather_rows(scp_ctx* ctx, int N, int K, int D, int64_t q_begin, int64_t q_end, const double* eta,
                    const double* l_col, const int64_t* rows, int64_t n, double* w_eta, double* w_l);

/* ---- a1: relative step of the SCP loop (scp.py:157-159) ----------------------------------------------
 * out[0] = ||a_new - a_prev||_2, out[1] = ||a_prev||_2, out[2] = out[0]/out[1] (no zero guard, as the
 * reference).  out is [host]. */
int scp_rel_step(scp_ctx* ctx, int64_t n, const double* a_new, const double* a_prev, double* out);

/* ---- a3 / a6 / a9: the joint QP  min ||x||^2  s.t. fixed rows, collision rows -------------------------
 * Replaces osqp.OSQP().setup/warm_start/solve at scp.py:326-367 and :441-449.  ADMM in OSQP's form with a
 * matrix-free x-update (PCG preconditioned by the exact inverse of the block-diagonal fixed part), on the
 * fixed rows plus a working set of collision rows supplied by the caller (exact constraint generation is
 * driven through scp_collision_violations). */
void scp_qp_default_settings(scp_qp_settings* s);
size_t scp_qp_workspace_bytes(int N, int K, int D, int64_t row_capacity);
int scp_qp_create(scp_ctx* ctx, int N, int K, int D, double h, const scp_qp_settings* s, void* workspace,
                  size_t workspace_bytes, int64_t row_capacity, scp_qp** out);
void scp_qp_destroy(scp_qp* qp);
int scp_qp_update_settings(scp_qp* qp, const scp_qp_settings* s);
/* bounds of the fixed rows from the problem data (same arithmetic as scp_fixed_bounds) */
int scp_qp_set_problem(scp_qp* qp, const double* limits /*[host]*/, const double* space /*[host]*/,
                       const double* p0, const double* v0, const double* pf, const double* vf);
/* start a new QP: x = x0 ([N][K][D], NULL -> 0), z = A x, y = 0 (primal warm start only, scp.py:443),
 * empty working set, rho = settings.rho */
int scp_qp_reset(scp_qp* qp, const double* x0);
/* Start (or continue) from another step size than settings.rho -- e.g. the value the previous SCP iteration's QP ended with
 * (scp_solve_options.carry_rho): the rho-dependent blocks are taken from the cache or rebuilt.  After scp_qp_reset. */
int scp_qp_set_rho(scp_qp* qp, double rho);
/* append working rows (global ids; eta AoS [n][D]; lower bounds); z = max(A x, l), y = 0 */
int scp_qp_add_rows(scp_qp* qp, int64_t n, const int64_t* rows, const double* w_eta, const double* w_l);
/* append working rows with eta / l recomputed from the linearisation point pos_prev ([N][K][D]) by the arithmetic of
 * scp_linearize_pairs (bit-identical to gathering its stored rows); p0, v0 [N][D]; R = min_distance */
int scp_qp_add_rows_at(scp_qp* qp, int64_t n, const int64_t* rows, const double* pos_prev, const double* p0,
                       const double* v0, double R);
int scp_qp_solve(scp_qp* qp, scp_qp_info* info /*[host]*/);
/* Move a QP to a larger workspace: dst (same N, K, D, h; row_capacity >= src's working rows) takes over the
 * problem bounds, iterate, duals, working set and rho of src, so a solve can continue after its working set outgrew
 * the capacity it was created with. */
int scp_qp_clone_state(scp_qp* dst, const scp_qp* src);
int scp_qp_get_solution(scp_qp* qp, double* x_out /*[N][K][D]*/);
/* duals: y_fixed in the reference stacking order (N*D*(4K-1)), y_col per working row (may be NULL) */
int scp_qp_get_duals(scp_qp* qp, double* y_fixed, double* y_col);

/* ---- a1 as ONE call: SCP.generate_trajectories (scp.py:131-180) driven natively ------------------------------
 * The SCP loop is control flow around the entry points above; path_planning/solvers/scp.py drives them from Python
 * (about a hundred calls and half a dozen blocking reads per SCP iteration).  scp_solver_solve makes the SAME calls in
 * the SAME order from C++: QP#0, the avoidance check evaluated once (scp.py:144), then per iteration kinematics,
 * scp_linearize_pairs, the joint QP with exact constraint generation (scp_qp_* + scp_collision_violations_at) and the
 * relative-step test, optionally the polish QP, and the final kinematics.  Bit-identical to the Python-driven loop.
 * The solver object owns its device memory (hipMalloc: compact rows, bitmap, row lists, QP workspace, trajectories). */
typedef struct scp_solver scp_solver;

typedef struct scp_solve_options {
  int32_t max_iterations;       /* 15: SCP iterations (compute_trajectories.py:75) */
  int32_t max_rounds;           /* 20: constraint-generation rounds per joint QP */
  int32_t max_iter0;            /* 4000: ADMM iterations of QP#0 (OSQP default, scp.py:360) */
  int32_t max_iter;             /* 10000: ADMM iterations per joint QP (scp.py:442) */
  int32_t refresh_feasibility;  /* 0: re-evaluate is_feasible inside the loop (the reference's TODO, scp.py:150) */
  int32_t polish;               /* 0: one more joint QP at polish_eps after the loop */
  double working_set_margin;    /* 0.5 m */
  double feasibility_tol;       /* 1e-6 */
  double polish_eps;            /* 1e-8 */
  double convergence_tolerance; /* 1.5e-2 (scp.py:52) */
  int32_t row_free;             /* 1: linearise with scp_select_pairs + scp_qp_add_rows_at (no eta / l planes are written
                                   or allocated); 0: scp_linearize_pairs writes all rows and the working rows are gathered
                                   from them.  Same working rows, same bits either way */
  int32_t carry_rho;            /* 0 (OSQP: every new solver object starts at settings.rho, scp.py:441); 1: the joint QP of SCP
                                   iteration n + 1 starts at the rho iteration n ended with (its blocks are cached), which saves
                                   the adaptive-rho transient: fewer ADMM steps, the same minimiser within eps */
} scp_solve_options;

#define SCP_MAX_ROUNDS_RECORDED 24
/* [host] one per QP of a solve: records[0] = QP#0, then one per SCP iteration, then the polish QP if requested */
typedef struct scp_qp_record {
  int32_t status_val;       /* of the last constraint-generation round */
  int32_t iter;             /* ADMM iterations, all rounds */
  int32_t rho_updates, cg_iters_total, rounds;
  int32_t pipeline;         /* OR over the rounds of scp_qp_info.pipeline */
  int64_t working_rows, unresolved_rows;
  int64_t added[SCP_MAX_ROUNDS_RECORDED]; /* rows found by the violations pass after each round */
  double r_prim, r_dual, rho, solve_ms, max_violation;
  double rel_step;          /* ||a_new - a_prev|| / ||a_prev|| (scp.py:157-159); -1 for QP#0 / polish */
  double time_sec;          /* wall time of the SCP iteration */
  double linearize_ms;      /* device time of the linearisation kernel alone (HIP events around that launch) */
  double violations_ms;     /* device time of the last violations kernel */
  int32_t persist_launches, persist_gave_up, rho_switches_in_kernel, reserved;  /* sums over the rounds (scp_qp_info) */
} scp_qp_record;

typedef struct scp_solve_result {
  int32_t n_iterations, converged, initially_feasible, feasible_at_exit, polished;
  int32_t qp0_status;       /* not in {1, 2}: the caller raises RuntimeError("OSQP failed: ...") like scp.py:363-365 */
  int32_t n_records;
  int32_t first_violation_k, first_violation_i, first_violation_j;  /* of the initial check (scp.py:611-613) */
  uint64_t first_violation; /* row id, UINT64_MAX if none */
  double first_violation_distance;
  double time_sec;
} scp_solve_result;

void scp_solve_default_options(scp_solve_options* o);
/* qp_row_capacity <= 0: the default min(K pairs, max(8192, 32 N K)) (grows on demand) */
int scp_solver_create(scp_ctx* ctx, int N, int K, int D, double h, double R, const scp_qp_settings* st,
                      int64_t qp_row_capacity, scp_solver** out);
void scp_solver_destroy(scp_solver* s);
int scp_solver_update_settings(scp_solver* s, const scp_qp_settings* st);
/* limits / space [host] as in scp_fixed_bounds; p0, v0, pf, vf [N][D] device; acc_out, pos_out, vel_out [N][K][D] device;
 * res, records [host], record_capacity >= max_iterations + 2.  Synchronises the stream before returning. */
int scp_solver_solve(scp_solver* s, const double* limits, const double* space, const double* p0, const double* v0,
                     const double* pf, const double* vf, const scp_solve_options* o, double* acc_out, double* pos_out,
                     double* vel_out, scp_solve_result* res, scp_qp_record* records, int record_capacity);

/* ONE SCP iteration (the loop body scp.py:152-166 without the convergence decision): linearise around acc_in, joint QP with
 * exact constraint generation, relative step -> *rec [host] (rel_step, time_sec, linearize_ms ... filled), acc_out = the new
 * accelerations ([N][K][D] device, may alias nothing).  Same calls in the same order as one pass of scp_solver_solve's loop;
 * bench.py times this call.  Synchronises the stream before returning. */
int scp_solver_step(scp_solver* s, const double* limits, const double* space, const double* p0, const double* v0,
                    const double* pf, const double* vf, const scp_solve_options* o, const double* acc_in, double* acc_out,
                    scp_qp_record* rec);

/* ---- the same iteration split at its exchange points: agents / pairs sharded over the GPUs of a node, one process per GPU ---
 * Every rank holds the whole (N, K, D) problem and calls, per SCP iteration,
 *   scp_solver_shard_begin       bounds, positions of the linearisation point (pos_in = the allgathered per-shard trajectories,
 *                                or NULL: computed from acc_in), row-free selection over ITS pair range -> its row ids
 *   [exchange: allgather of the ids, sorted ascending -- the only data that crosses ranks]
 *   scp_solver_shard_qp          the gathered rows join the REPLICATED working set (eta / l recomputed locally: every rank
 *                                has the linearisation point), ADMM on the joint QP: deterministic, the same bits on every rank
 *   scp_solver_shard_violations  every row of its pair range checked at the solution -> its new row ids, max violation
 *   [exchange: allgather of the ids, max of the violations]
 *   scp_solver_shard_round_done  *more = another round (-> shard_qp with the new rows) or not
 *   scp_solver_shard_end         relative step, acc_out, the record
 * which are the phases scp_solver_step runs back to back over the full pair range: a world of one rank that skips the
 * exchanges reproduces scp_solver_step bit for bit, and so does any world size (sorted ids = the single-rank row order).
 * rows_out (device, capacity rows_cap) receives this rank's ids, *n_local [host] their number; if n_local > rows_cap only the
 * first rows_cap were copied: fetch them again with scp_solver_shard_rows into a longer list.  rec [host] accumulates. */
int scp_solver_shard_begin(scp_solver* s, const double* limits, const double* space, const double* p0, const double* v0,
                           const double* pf, const double* vf, const scp_solve_options* o, const double* acc_in,
                           const double* pos_in, int64_t q_begin, int64_t q_end, scp_qp_record* rec,
                           int64_t* rows_out, int64_t rows_cap, int64_t* n_local);
int scp_solver_shard_rows(scp_solver* s, int64_t* rows_out, int64_t rows_cap);
int scp_solver_shard_qp(scp_solver* s, const int64_t* rows, int64_t n, scp_qp_record* rec);
int scp_solver_shard_violations(scp_solver* s, int64_t* rows_out, int64_t rows_cap, int64_t* n_local,
                                double* max_violation);
int scp_solver_shard_round_done(scp_solver* s, int64_t n_all, double max_violation_all, scp_qp_record* rec, int* more);
int scp_solver_shard_end(scp_solver* s, double* acc_out, scp_qp_record* rec);

/* ---- test hooks (dense K-dimension products used by the QP; exercised by tests/test_kernels_gpu.py::test_gemm_f64) ------
 * Y[R][C] = alpha * A[R][M] X[M][C] + beta * Y, row-major, device pointers. */
int scp_gemm_f64(scp_ctx* ctx, int use_mfma, int R, int M, int C, double alpha, const double* A,
                 const double* X, double beta, double* Y);
/* Copy one internal array of the solver (time-major device layout) to `out` (device, capacity `cap` doubles);
 * *n_out [host] = its length.  name: "x" [K][C], "zf" / "yf" / "fx" [4K-1][C] (fx = carried F x), "qx" [K][C] (carried S0 x),
 * "zc" / "yc" per working row, "gval" per incidence-list entry.  tests/test_qp_gpu.py compares the state the persistent
 * and the three-launch pipelines leave behind. */
int scp_qp_peek(scp_qp* qp, const char* name, double* out, int64_t cap, int64_t* n_out);
/* "persist_fault" = n: the next n persistent launches wait for a workgroup that does not exist, so their bounded spins time
 * out (the give-up path: nothing written back, the solve continues on the three-launch pipeline); "persist_off": read
 * (value < 0) or set whether the solver has fallen back.  Returns the value in effect. */
int scp_qp_debug_set(scp_qp* qp, const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* SCP_HIP_H */
