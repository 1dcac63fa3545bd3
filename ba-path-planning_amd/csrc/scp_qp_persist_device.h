// Shared by the persistent ADMM kernels (scp_qp_persist.hip: one wave per agent, 8 / 4 agents per workgroup;
// scp_qp_persist16.hip: the lean 16-agent form): launch arguments, exit codes, the tagged-granule primitives of the two
// cross-workgroup exchanges.  gfx950 only.
#pragma once
#include "scp_qp_device.h"

namespace scp_persist {
using namespace scpdev;

typedef unsigned long long u64;
constexpr unsigned SPIN_LIMIT = 1u << 20;

struct PersistArgs {
  int K, N, nblk, ent_cap;
  int it0, max_iter, check_every, rho_interval;  // iterations done so far in this solve; limits; termination / rho periods
  int cad0, check_fine;    // adaptive check cadence (scp_qp_settings.check_fine): steps between checks when this launch starts
  double fine_ratio;       // (the host's decision after the previous check), the fine period, and "close" = residuals < ratio x tolerance
  int64_t C;
  double rho, rho_c, rho_eq, alpha, h;
  double eps_abs, eps_rel, eps_prim_inf, rho_tol;  // termination tests (eps_prim_inf <= 0: no certificate; rho_tol <= 0: fixed rho)
  // lean kernel: the bounds of the jerk / acceleration rows (the same for every row), and what the velocity / position
  // bounds are made of (scp_qp_set_problem keeps a copy of the four state arrays: [4][N][D] = p0, v0, pf, vf)
  double jerk_lo, jerk_hi, acc_lo, acc_hi, vel_lo, vel_hi, pmin[3], pmax[3];
  const double* states;
  int first_step;  // lean kernels: this launch starts with the first ADMM step after scp_qp_reset (z = A x0 unprojected, y = 0)
  int spin_sleep;  // naps (64 clocks each) between two polls of a granule that has not arrived (1; more when many persistent
                   // launches share the chip: their polls load the fabric the hand-offs travel on)
  const double* pMinv;
  const double* pT;      // packed T = S0 H_f^{-1}
  const double *lf, *uf;
  double *zf, *yf, *fx, *x, *Qx, *dyf;
  u64* cells;       // [K][N][D][2] granules: S0 p of (time step, agent), low / high word, each tagged with the step
  u64* gpart;       // [2 parities][nblk][4] granules: r.p and sum (eta . d S0 p)^2 of one workgroup
  u64* gcheck;      // [nblk][18] granules: the nine partial results of a termination check of one workgroup
  unsigned* give_up;
  const int *cell_ptr, *ent_code, *w_k, *w_i, *w_j;
  const double *w_eta, *w_l;
  double *zc, *yc, *dyc, *gval;
  unsigned* host_status;  // mapped host words: [0] exit code (EXIT_*), [1] ADMM iterations done when the kernel left
  double* host_scal;      // mapped host array: the nine check results in the SL_* slots of scp_qp::h_scal
  u64* host_flag;         // mapped completion word, set to `seq` last
  u64 seq;
  unsigned epoch0;        // steps completed by earlier launches (tags never repeat; the buffers start zeroed)
  // adaptive rho inside the kernel: the rho values whose blocks the host has cached (scp_qp::kkt).  When a check asks for a
  // new rho that is in this table the kernel switches by itself (operands reloaded, row values recomputed) and goes on;
  // otherwise it returns EXIT_RHO and the host builds the blocks.  host_status[2] = switches made, *host_rho = rho at exit.
  double rho_col_scale;
  double* host_rho;
  int n_tab;
  struct RhoSlot {
    double rho;
    const double* pMinv;
    const double* pT;
  } tab[SCP_KKT_SLOTS_MAX];
};

// why the kernel returned (host_status[0]); the host re-derives every decision from the nine check results
enum { EXIT_SOLVED = 1, EXIT_GAVE_UP = 2, EXIT_MAX_ITER = 3, EXIT_INFEASIBLE = 4, EXIT_RHO = 5, EXIT_OVERFLOW = 6 };
constexpr int NCHK = 9;  // rp, |Ax|, |z|, rd, |Px|, |A^T y|, |dy|, supp (a sum), |A^T dy|  (maxima of non-negative values)
constexpr int CK_RP = 0, CK_NAX = 1, CK_NZ = 2, CK_RD = 3, CK_NPX = 4, CK_NATY = 5, CK_NDY = 6, CK_SUPP = 7, CK_NATDY = 8;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// One double = one 16-byte pair of granules {low word, tag, high word, tag}: ONE write-through store, ONE load (a scalar
// sc1 store is one fabric write whatever its width: 8-byte stores doubled the hand-off's fabric traffic).  Each 8-byte
// half carries its own tag, so a torn pair is detected like a late one.  Inline asm because the builtins offer no 16-byte
// agent-scope access; the asm loads wait for their own data (the compiler does not count them).
__device__ inline void st_granules(u64* g, unsigned tag, double v) {
  const u32x4 w = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g), "v"(w) : "memory");
}
__device__ inline u32x4 ld_pair(const u64* g) {
  u32x4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(g) : "memory");
  return w;
}
template <int D>
__device__ inline void ld_cell(const u64* g, u32x4 (&w)[D]) {  // the D doubles of one cell: D loads in flight, one wait
  if (D == 2) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1])
                 : "v"(g)
                 : "memory");
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = ld_pair(g + 2 * d);
  }
}
__device__ inline void spin_nap(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
__device__ inline bool pair_ok(const u32x4& w, unsigned tag) { return w[1] == tag && w[3] == tag; }
__device__ inline double pair_value(const u32x4& w) { return __hiloint2double((int)w[2], (int)w[0]); }


}  // namespace scp_persist

// host side of the lean kernel (scp_qp_persist16.hip), called by scp_qp_cg1_persist
size_t scp_persist16_lds_bytes(int K, int cap, int nblk, int D, int apb);
int scp_persist16_launch(scp_ctx* ctx, const scp_persist::PersistArgs& a, int nblk, size_t lds, int D, int apb);
