// Declarations shared by scp_qp.hip (ADMM driver, generic kernels) and scp_qp_fused.hip (column-block kernels).
#pragma once
#include "scp_common.h"

constexpr int SCP_KKT_SLOTS = 6;       // cached rho values per solver object: at least this many ...
constexpr int SCP_KKT_SLOTS_MAX = 32;  // ... and up to this many while the pool stays below SCP_KKT_POOL_BYTES
constexpr size_t SCP_KKT_POOL_BYTES = (size_t)48 << 20;
constexpr int NPART = 128;  // partial sums of a dot product (fixed -> deterministic summation order)

enum Slot {  // device scalar slots (doubles)
  SL_RZ0 = 0, SL_RZ1 = 1,
  SL_RP = 8, SL_NAX = 9, SL_NZ = 10, SL_RD = 11, SL_NPX = 12, SL_NATY = 13,
  SL_NDY = 14, SL_SUPP = 15, SL_NATDY = 16,
  SL_COUNT = 32
};

struct QpDev {
  // constant blocks
  double *F, *Ft, *S0, *S0t, *HS, *Hf, *Minv, *aug, *wrow;
  double* gj_tmp;  // [3K]: pivot row and column of the per-pivot Gauss-Jordan launches (K > SCP_INV_LDS_MAX_K)
  double* G0;  // [K][K]: F^T w F (constant; H_f = (2 + sigma) I + rho G0)
  // the same blocks in MFMA A-operand order for the column-block kernels (scp_qp_pack_operands):
  // [row tile][k step][lane] = A[16 tile + (lane & 15)][4 step + (lane >> 4)], zero beyond the matrix, so that one
  // wave-wide operand load is 512 contiguous bytes
  double *pF, *pFt, *pS0, *pS0t, *pHS, *pMinv;
  double *T, *pT;  // T = S0 H_f^{-1} (K x K, rebuilt with H_f^{-1}) and its packed form: S0 p = T r without waiting for p
  // fixed rows
  double *lf, *uf, *zf, *yf, *wf, *tf;
  double* states;  // [4][N][D]: p0, v0, pf, vf of the latest scp_qp_set_problem (the lean persistent kernel re-derives the bounds)
  // x-space vectors [K][C]
  double *x, *xt, *rhs, *r, *p, *zz, *G;
  double* HQ;  // [2K][C]: rows [0,K) = H v, rows [K,2K) = S0 v
  // working rows
  int64_t* w_row;
  int *w_k, *w_i, *w_j;
  double *w_eta, *w_l, *zc, *yc;
  // scalars
  double* scal;   // SL_COUNT, followed by SCP_RESID_CAP partial results of a termination check
  double* part;   // 2 * SCP_PART_CAP
  double* hpf;    // [K][C]: H_f p of the fused PCG
  double* fx;     // [Rf][C]: F x carried by the single-step pipeline (F p goes to tf)
  double* dyf;    // [Rf][C]: snapshot of y_f, then delta-y (primal infeasibility certificate)
  double* dyc;    // [cap]  : same for the working rows
  // deterministic row -> column transfer (single-step pipeline): incidence lists per (time step, agent) cell
  int* cell_ptr;  // [N*K + 1] exclusive offsets into the entry arrays
  int* cell_cur;  // [N*K]     fill cursors (build time)
  int* scan_tot;  // [N*K / 4096 + 2] per-workgroup totals of the offset scan
  int* ent_code;  // [2 cap]   2 n + side, sorted inside every cell (side 0: agent i, +eta; side 1: agent j, -eta)
  double* coef;   // [2 cap][D] signed eta of the entry
  double* gval;   // [2 cap]   per-entry row value, written by the row kernels (no atomics)
  double* gval2;  // [2 cap]   row vectors of a termination check: yc ...
  double* gval3;  // [2 cap]   ... and delta-yc (gval keeps the pipeline's values across a check)
  int *pos_i, *pos_j;  // [cap] entry positions of row n
  double* kkt_pool;  // scp_kkt_slots(K) cache slots of the rho-dependent blocks
  unsigned long long* sync_words;  // SCP_SYNC_WORDS: give-up word of the persistent kernel, scratch
  unsigned long long* cells;       // [K][N][D][2] tagged granules: S0 p cells published by the persistent kernel
  unsigned long long* gpart;       // SCP_GPART_WORDS tagged granules: line-search partials, two alternating buffers
  unsigned long long* gcheck;      // SCP_GCHECK_WORDS tagged granules: termination-check partials
};

struct scp_qp {
  scp_ctx* ctx;
  int N, K, D, Rf;
  int64_t C;
  double h;
  scp_qp_settings st;
  int64_t row_cap, nW;
  bool problem_set, reset_done;
  bool cg1_ready;  // carried state (Qx, gval) of the single-step pipeline matches (x, zc, yc, rho)
  bool gval_valid;   // the incidence lists and row values were built together with the latest rows (scp_qp_install_rows_small)
  double gval_rho_c; // ... at this column rho: scp_qp_cg1_prepare has nothing left to launch
  bool csr_valid;  // incidence lists match the working set
  bool qx_fresh;   // the current S0 x buffer and the F x slab are exact for x (written by the fused residual kernel)
  int qx_sel;      // which half of HQ holds S0 x (the single-step pipeline ping-pongs: 0 -> rows [K, 2K), 1 -> [0, K))
  double rho;
  QpDev d;
  // rho-dependent blocks (H_f, [H_f; S0], H_f^{-1}, T and their packed forms) are cached per rho: adaptive rho is snapped
  // to a geometric grid and every solve starts from settings.rho, so the same few values recur from QP to QP and the
  // 48 us single-workgroup inverse is paid once per value.  d.Hf / HS / Minv / T / pHS / pMinv / pT point into the active slot.
  struct KktSlot {
    double rho, sigma;
    unsigned long long used;  // LRU stamp, 0 = empty
    double *Hf, *HS, *Minv, *T, *pHS, *pMinv, *pT;
  } kkt[SCP_KKT_SLOTS_MAX];
  int n_kkt;  // slots in use for this K (scp_kkt_slots)
  unsigned long long kkt_clock;
  bool consts_packed;  // F, F^T, S0, S0^T are packed once
  double* h_scal;  // pinned, SL_COUNT + SCP_RESID_CAP doubles + the completion flag of a fused check
  double* h_scal_dev;  // the same memory as the device sees it: the check kernels write their partials straight to it
  unsigned long long check_seq;  // value the flag takes when the current check has finished
  // persistent single-step kernel (scp_qp_persist.hip)
  double space[6];                   // {min_0.., max_0..} of the latest scp_qp_set_problem
  double lim[6];                     // {vel, acc, jerk} x {min, max} of the latest scp_qp_set_problem (the lean persistent kernel
                                     // takes the jerk / acceleration bounds as scalars)
  int64_t steps_since_reset;         // ADMM steps since scp_qp_reset: the lean persistent kernels keep one double per fixed row
                                     // (v = z~ + y / rho), which presumes z = Pi(v) -- true after any ADMM update, not for the
                                     // unprojected z = A x0 of a reset: their first step of a QP takes z = v (y = 0 then)
  int persist_variant;               // of the latest persistent launch: 0 = 8 (4 in 3-D) agents per workgroup, 1 = lean, 16
  int persist_fault;                 // test hook: the next n persistent launches wait for a workgroup that does not exist
  bool persist_off;                  // a launch gave up (workgroups not co-resident): three-launch pipeline until the next
                                     // scp_qp_reset / scp_qp_set_problem re-arms the persistent path
  bool persist_skip_solve;           // the CUs for a persistent launch were not free: this scp_qp_solve call stays on the
                                     // three-launch pipeline (cleared at the start of every call)
  int persist_gave_up_total;         // give-ups over the life of the object (scp_qp_debug_set "persist_gave_up_total")
  int64_t persist_cap_nW;            // working-set size that overflowed the LDS entry tables (-1: none): not tried again
  int persist_cap;                   // LDS entry capacity per workgroup (all the LDS that is left)
  unsigned long long persist_epoch;  // ADMM steps run by persistent launches so far: the step tags of the granules never repeat
  unsigned* h_persist;               // mapped host words written by the kernel: [0] exit code, [1] iterations done, [2] rho switches
  int persist_rho_switches;          // of the latest persistent launch: adaptive-rho updates the kernel made by itself ...
  double persist_rho;                // ... and the rho it ended with
  unsigned* h_persist_dev;
};


// scp_qp_fused.hip: one ADMM iteration with the column-local chains fused into column-block kernels
// (K <= SCP_FUSED_MAX_K).  Same arithmetic as admm_iteration() in scp_qp.hip.
constexpr int SCP_RESID_STRIDE = 12;  // doubles per workgroup in the partial results of a fused termination check
constexpr int SCP_RESID_CAP = (4096 / 2 + 128) * SCP_RESID_STRIDE;  // (SCP_PART_CAP / 2 column blocks + row blocks)
constexpr int SCP_INV_LDS_MAX_K = 96;  // [H_f | I] (K x 2K doubles) resident in LDS for the Gauss-Jordan inverse
constexpr int SCP_BIGK_MAX_K = 1024;  // single-step pipeline with one workgroup per column and one thread per time step
constexpr int SCP_FUSED_MAX_K = 120;  // (6K + 4K-1) * 128 B of LDS tiles <= 160 KiB (limit raised above 64 KiB)
constexpr int SCP_PART_CAP = 4096;  // capacity of each partial-sum array (column blocks of the fused path)
int scp_qp_fused_iteration(scp_qp* qp, int* cg_count);
// single-PCG-step pipeline (cg_iters == 1 and a non-empty working set): 3 launches per ADMM step
int scp_qp_cg1_iteration(scp_qp* qp, int* cg_count, bool emit_dy);
// nW == 0: `nit` complete ADMM iterations in one launch (everything is column-local)
int scp_qp_qp0_iterations(scp_qp* qp, int nit, double* dy_out);
// Deterministic A_W^T g into the G slab (valid incidence lists required): mode 0: g = rho_c zc - yc, 1: g = yc,
// 2: g = vec[n].  Two launches, no atomics.
int scp_qp_csr_scatter(scp_qp* qp, int mode, const double* vec);
int scp_qp_csr_build(scp_qp* qp);
// G = A_W^T g, g = (rho zc - yc) - rho A_W v (init) or rho A_W v with Q = S0 v: deterministic (gather over the incidence lists)
int scp_qp_rows_gather(scp_qp* qp, bool init, const double* Q);
int scp_qp_cg1_prepare(scp_qp* qp);
// small problems: working rows [nW, nW + n) recomputed from the linearisation point (scp_qp_add_rows_at) AND the incidence
// lists + row values of all nW + n rows in ONE launch; *done = false: not eligible, nothing was launched
int scp_qp_install_rows_small(scp_qp* qp, int64_t n, const int64_t* rows, const double* pos_prev, const double* p0,
                              const double* v0, double R, const double* Qx, bool* done);
// small problems: scp_qp_reset(x0) AND the installation of the QP's first n rows in ONE launch (launch only: reset_impl in
// scp_qp.hip keeps the host-side state); *done = false: not eligible, nothing was launched
int scp_qp_reset_install_small(scp_qp* qp, const double* x0, int64_t n, const int64_t* rows, const double* pos_prev,
                               const double* p0, const double* v0, double R, double* Qx, bool* done);
constexpr int SCP_SYNC_WORDS = 16;  // u64: give-up word | scratch
// Workgroups of a persistent launch: at most one per CU (all resident), and the exchange buffers below are sized for
// exactly this many (+1: the fault-injection hook announces one workgroup more than it launches).
constexpr int SCP_PERSIST_MAX_WG = 256;
constexpr int SCP_GPART_WORDS = 2 * (SCP_PERSIST_MAX_WG + 1) * 4;  // two buffers x workgroups x two doubles as granule pairs
constexpr int SCP_GCHECK_WORDS = (SCP_PERSIST_MAX_WG + 1) * 9 * 2;  // nine check results per workgroup as granule pairs
bool scp_qp_persist_eligible(const scp_qp* qp);
// Compute units claimed by the persistent launches in flight on one device of THIS process (solver threads on several
// streams, compute-trajectories-batch): a launch needs all its workgroups resident at once, so it first claims one CU per
// workgroup.  scp_persist_claim waits up to `wait_ms` for the claim to fit (returns false otherwise: the caller runs this
// batch of iterations on the three-launch pipeline instead of discovering the shortage through a spin timeout).
bool scp_persist_claim(int device, int n_cu_total, int n_wg, int wait_ms);
void scp_persist_release(int device, int n_wg);
constexpr int SCP_PERSIST_GAVE_UP = 2;  // exit code of the persistent kernel: a spin timed out, nothing was written back
int scp_qp_cg1_persist(scp_qp* qp, int it0, int cad0, int* ran, int* code, int* it_done);  // cad0: steps to the next check
// (re)pack F, Ft, S0, S0t, HS, Minv into the MFMA operand order; called at the end of build_kkt
int scp_qp_pack_operands(scp_qp* qp, bool constants);  // constants: F, F^T, S0, S0^T; else the active slot's HS, Minv, T
static inline size_t scp_packed_count(int R, int M) { return (size_t)((R + 15) / 16) * ((M + 3) / 4) * 64; }
// Termination-check quantities of the single-step pipeline in 3 launches (row values, column blocks, rows):
// fills qp->h_scal[SL_RP .. SL_SUPP] like residuals() in scp_qp.hip and leaves S0 x and F x in their slabs.  Synchronises.
// with_dy: dyf / dyc hold delta-y of the last iteration (cg1_update_kernel); also fills h_scal[SL_NATDY].
int scp_qp_fused_residuals(scp_qp* qp, bool with_dy);
