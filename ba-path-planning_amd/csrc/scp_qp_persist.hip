// Persistent form of the single-PCG-step ADMM pipeline (settings.cg_iters == 1, K <= 64): ONE launch runs all ADMM
// steps up to the next termination check with the whole solver state on chip.
//
// Same arithmetic as the three-launch pipeline of scp_qp_fused.hip (cg1_col_kernel / cg1_rows_sq_kernel /
// cg1_update_kernel; CPU statement: oracle/qp_oracle.py:admm_structured), replaces osqp's solve loop at
// /root/reference/src/path_planning/solvers/scp.py:441-445.  What changes is where the state lives and how workgroups
// talk:
//   * one workgroup per block of 16/D agents, ONE WAVE PER AGENT, lane = time step: the 4K-1 fixed rows of a column
//     (z, y, l, u and the carried F x) and x, S0 x sit in the registers of the lane that owns the time step, so the
//     integrator blocks are wave scans on registers with no layout change in between (20 D doubles per lane);
//   * p = H_f^{-1} r stays on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, operands resident in registers);
//   * every working row (k, i, j) is REPLICATED in the workgroups that own agent i and agent j (entries of the
//     agent-major incidence lists, kept in LDS): both copies run the same instructions on the same values, so z_c, y_c
//     and the row value g evolve bit-identically on both sides and A_W^T g needs no exchange at all;
//   * what crosses workgroups per step is (1) the S0 p cells of the partner agents -- a neighbour-only hand-off -- and
//     (2) the two scalars of the exact line search  a = r.p / (r.p + rho_c sum (eta . d S0 p)^2), an all-gather of two
//     doubles per workgroup.  Both travel as DATA-TAGGED GRANULES (MI355X_MICROARCH.md, hand-off form R2): every 8-byte
//     word is {step tag, 32 payload bits}, written by one write-through (agent-scope relaxed atomic) store and polled by
//     agent-scope loads until its tag is the current step.  The data is the flag: no counters, no fences, no grid
//     barrier (measured with counters + barrier: 4.4 + 2.8 us of a 13 us step; profiles/r02_phase_profile_*.txt).
//     Reuse of a buffer is safe because the all-gather is a full dependency of every step: a workgroup republishes its
//     S0 p cells only after it has every partial of the previous step, which every workgroup publishes after reading
//     its partners' cells; the partials alternate between two buffers.
// Every spin is bounded: a workgroup that times out raises a give-up word, every workgroup then leaves WITHOUT writing
// state back, and the host repeats the iterations on the three-launch pipeline.
#include "scp_qp_persist_device.h"

#include <algorithm>
#include <cstdlib>
#include <cerrno>
#include <chrono>


namespace {
using namespace scpdev;
using namespace scp_persist;

// Developer build (make prof): 100 MHz wall-clock ticks spent per phase, summed over the steps of one launch, middle
// workgroup, thread 0.
#ifdef SCP_PHASE_PROFILE
__device__ unsigned long long scp_persist_clk[16];
#define PSTAMP(slot)                                  \
  do {                                                \
    const unsigned long long now_ = wall_clock64();   \
    pacc[slot] += now_ - plast;                       \
    plast = now_;                                     \
  } while (0)
#else
#define PSTAMP(slot) ((void)0)
#endif

// Agents (= waves) per workgroup.  2-D: 8 waves, two per SIMD, 256 registers each (waves 4-7 double as the T r matrix-core
// waves).  3-D: FOUR agents (12 of the 16 tile columns), not five -- a wave's 30 columns of fixed-row state need more than 256
// registers, and with one wave per SIMD the whole 512-entry file is its own (five waves spilled 160 registers to scratch
// inside the step loop: 12.9 us per step against 9.4 in 2-D).
constexpr int persist_apb(int D) { return D == 3 ? 4 : CB / D; }

template <int D>
__global__ __launch_bounds__(64 * persist_apb(D)) void cg1_persist_kernel(PersistArgs A) {
  constexpr int APB = persist_apb(D);    // agents = waves per workgroup
  constexpr int NT = 64 * APB;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red[NCHK][APB];
  __shared__ int fail_s;
  const int K = A.K, N = A.N;
  const int64_t C = A.C;
  const int RSK = pad_col(K);
  const int cap = A.ent_cap;
  double* Rt = lds;                        // [16][RSK]   r, MFMA B operand
  double* Pt = Rt + CB * RSK;              // [16][RSK]   p
  double* Tt = Pt + CB * RSK;              // [16][RSK]   S0 p = T r (when the workgroup has waves to spare for it)
  double* Qt = Tt + CB * RSK;              // [APB][64][D] own S0 p cells
  double* gp = Qt + APB * 64 * D;          // [nblk][9]   all-gathered partials (line search: 2 per workgroup; check: 9)
  double* e_c = gp + NCHK * A.nblk;        // [cap][D] signed eta (+ on agent i's side, - on agent j's)
  double* e_l = e_c + (size_t)cap * D;     // [cap] lower bound
  double* e_z = e_l + cap;                 // [cap]
  double* e_y = e_z + cap;                 // [cap]
  double* e_g = e_y + cap;                 // [cap] row value of the next right-hand side
  double* e_qo = e_g + cap;                // [cap][D] S0 x cell of the own agent
  double* e_qp = e_qo + (size_t)cap * D;   // [cap][D] S0 x cell of the partner agent
  double* e_pp = e_qp + (size_t)cap * D;   // [cap][D] S0 p cell of the partner agent (this step)
  int* e_code = (int*)(e_pp + (size_t)cap * D);  // [cap] k | local agent << 8 | side << 16
  int* e_pad = e_code + cap;               // [cap] granule index of the partner's cell
  int* e_row = e_pad + cap;                // [cap] working row n
  int* cptr = e_row + cap;                 // [APB K + 1] cell offsets relative to this workgroup's first entry

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a0 = blockIdx.x * APB;
  const int agent = a0 + wave;
  const bool aok = agent < N;
  const int k = lane;
  const bool live = aok && k < K;
  const double h = A.h, hh = h * h, alpha = A.alpha;
  double rho = A.rho, rho_c = A.rho_c;  // (an in-kernel rho switch changes them)
  unsigned n_rho = 0;
  const int tK = (K + 15) >> 4, nks = (K + 3) >> 2;
  const int nblk = A.nblk;

  // ---- entries of this block of agents (contiguous in the agent-major incidence lists) --------------------------
  const int a1 = min(a0 + APB, N);
  const int ebase = A.cell_ptr[cell_of(0, a0, K)];
  const int ne = A.cell_ptr[cell_of(0, a1, K)] - ebase;
  // More incident rows around some block of agents than the LDS tables hold: EVERY workgroup finds that out by itself
  // (the largest block's count, a few loads) and leaves before anything is published -- nobody spins on anybody.
  {
    int worst = 0;
    for (int b = threadIdx.x; b < (N + APB - 1) / APB; b += NT) {
      const int b0 = b * APB, b1 = min(b0 + APB, N);
      worst = max(worst, A.cell_ptr[cell_of(0, b1, K)] - A.cell_ptr[cell_of(0, b0, K)]);
    }
    if (__syncthreads_or(worst > cap)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(A.host_status, (unsigned)EXIT_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.host_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
  if (threadIdx.x == 0) fail_s = 0;
  for (int i = threadIdx.x; i <= (a1 - a0) * K; i += NT) cptr[i] = A.cell_ptr[cell_of(0, a0, K) + i] - ebase;
  for (int e = threadIdx.x; e < ne; e += NT) {
    const int code = A.ent_code[ebase + e];
    const int n = code >> 1, side = code & 1;
    const int wi = A.w_i[n], wj = A.w_j[n], wk = A.w_k[n];
    const int own = side ? wj : wi, par = side ? wi : wj;
    e_code[e] = wk | ((own - a0) << 8) | (side << 16);
    e_pad[e] = ((wk * N + par) * D) * 2;
    e_row[e] = n;
    const int64_t bo = (int64_t)wk * C + (int64_t)own * D, bp = (int64_t)wk * C + (int64_t)par * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double eta = A.w_eta[(size_t)n * D + d];
      e_c[(size_t)e * D + d] = side ? -eta : eta;
      e_qo[(size_t)e * D + d] = A.Qx[bo + d];
      e_qp[(size_t)e * D + d] = A.Qx[bp + d];
    }
    e_l[e] = A.w_l[n];
    e_z[e] = A.zc[n];
    e_y[e] = A.yc[n];
    e_g[e] = A.gval[ebase + e];
  }
  for (int i = threadIdx.x; i < CB * RSK; i += NT) Rt[i] = 0.0;  // columns beyond the block stay zero

  // ---- column state: lane k of the agent's wave holds the rows of time step k --------------------------------------
  // row types t = 0 jerk (k < K - 1), 1 acc, 2 vel, 3 pos;  slab row of (t, k): t = 0: k, else t K - 1 + k
  double z[D][4], y[D][4], lo[D][4], hi[D][4], fx[D][4], rr[4], x[D], qx[D];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool rok = live && (t > 0 || k < K - 1);
    const int row = t == 0 ? k : t * K - 1 + k;
    rr[t] = (t >= 2 && k == K - 1) ? rho * A.rho_eq : rho;  // final velocity / position equalities (scp.py:223-224, :256-257)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      z[d][t] = y[d][t] = lo[d][t] = hi[d][t] = fx[d][t] = 0.0;
      if (rok) {
        const int64_t g = (int64_t)row * C + (int64_t)agent * D + d;
        z[d][t] = A.zf[g]; y[d][t] = A.yf[g]; lo[d][t] = A.lf[g]; hi[d][t] = A.uf[g]; fx[d][t] = A.fx[g];
      }
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    x[d] = qx[d] = 0.0;
    if (live) {
      const int64_t g = (int64_t)k * C + (int64_t)agent * D + d;
      x[d] = A.x[g];
      qx[d] = A.Qx[g];
    }
  }
  // waves [0, tK): tiles of p = H_f^{-1} r; waves [tK, 2 tK), when the workgroup has them: tiles of S0 p = T r, so the cells
  // can be published one scan phase earlier (the hand-off then overlaps the F p scans)
  const bool use_T = APB >= 2 * tK;
  double aM[CHB];
  tile_prefetch<CHB>(use_T && wave >= tK ? A.pT : A.pMinv, nks, wave < tK ? wave : (use_T && wave < 2 * tK ? wave - tK : 0), 0, nks, aM);
  __syncthreads();
  // incidence-list range of this lane's cell (time step k of this wave's agent); cells without entries publish nothing
  const int c0 = aok ? cptr[wave * K + min(k, K - 1)] : 0;
  const int c1 = (aok && k < K) ? cptr[wave * K + k + 1] : c0;
  u64* my_cell = A.cells + ((size_t)((int64_t)min(k, K - 1) * N + (aok ? agent : 0)) * D) * 2;

#ifdef SCP_PHASE_PROFILE
  unsigned long long pacc[16] = {0}, plast = wall_clock64();
#endif
  PSTAMP(0);
  bool ok = true;
  double dy[D][4];
  int it_done = A.it0;      // ADMM iterations of this solve completed so far
  unsigned steps = 0;       // steps run by this launch
  unsigned exit_code = 0;
  double chk[NCHK];
  const bool with_dy = A.eps_prim_inf > 0.0;
  int cad = A.cad0;  // steps between two checks: check_every, or check_fine once the residuals are close / rho has changed
  for (;;) {  // one batch of steps up to the next termination check, then the check and the decision to go on
  int nit = cad - it_done % cad;
  if (it_done + nit > A.max_iter) nit = A.max_iter - it_done;
  for (int it = 0; it < nit; ++it, ++steps) {
    const unsigned tag = A.epoch0 + steps + 1u;
    u64* gpart = A.gpart + (size_t)(tag & 1u) * nblk * 4;
    // ---- r = -2 x + F^T W' + S0^T G: reverse cumulative sums as suffix scans over the lanes ------------------------
    double r[D];
    {
      double g[D];
#pragma unroll
      for (int d = 0; d < D; ++d) g[d] = 0.0;
      for (int e = c0; e < c1; ++e) {
        const double ge = e_g[e];
#pragma unroll
        for (int d = 0; d < D; ++d) g[d] += e_c[(size_t)e * D + d] * ge;
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const double wj = rr[0] * (z[d][0] - fx[d][0]) - y[d][0];
        const double wa = rr[1] * (z[d][1] - fx[d][1]) - y[d][1];
        const double wv = rr[2] * (z[d][2] - fx[d][2]) - y[d][2];
        const double wp = rr[3] * (z[d][3] - fx[d][3]) - y[d][3];
        const double u1 = h * wv + 0.5 * hh * (wp - g[d]);
        const double u2 = wp + g[d];
        const double d1 = wave_incl_rsum(u1);
        const double s1 = wave_incl_rsum(u2);
        const double d2 = lane_above(wave_incl_rsum(s1));  // exclusive suffix sum
        const double wjm = lane_below(wj);                  // w_j[k - 1]
        r[d] = (((wjm - wj) / h + wa) + (d1 + 0.5 * hh * g[d]) + hh * d2) - 2.0 * x[d];
        if (live) Rt[(wave * D + d) * RSK + k] = r[d];
      }
    }
    __syncthreads();
    PSTAMP(1);
    // ---- p = H_f^{-1} r on the matrix cores: one 16-row tile per wave --------------------------------------------
    if (wave < tK || (use_T && wave < 2 * tK)) {
      const int li = lane & 15, lk = lane >> 4;
      const int tile = wave < tK ? wave : wave - tK;
      double* Ot = wave < tK ? Pt : Tt;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < CHB; ++s) {  // K <= 64: the whole operand row block is resident in aM
        if (s < nks) {  // wave-uniform
          const int kk = 4 * s + lk;
          const double b = kk < K ? Rt[li * RSK + kk] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[s], b, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = tile * 16 + lk + 4 * q;
        if (row < K) Ot[li * RSK + row] = acc[q];
      }
    }
    __syncthreads();
    PSTAMP(2);
    // ---- S0 p, F p (forward scans), r.p; publish the S0 p cells that have rows -----------------------------------------
    double p[D], qp[D], fp[D][4];
    double rz = 0.0;
    if (use_T) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        qp[d] = live ? Tt[(wave * D + d) * RSK + k] : 0.0;
        if (c1 > c0) {
          st_granules(my_cell + 2 * d, tag, qp[d]);
          Qt[(wave * 64 + k) * D + d] = qp[d];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      p[d] = live ? Pt[(wave * D + d) * RSK + k] : 0.0;
      rz += (live ? r[d] : 0.0) * p[d];
      const double cs1 = wave_incl_sum(p[d]);
      const double cs2 = lane_below(wave_incl_sum(cs1));
      const double cs1p = lane_below(cs1);
      const double pn = lane_above(p[d]);
      if (!use_T) qp[d] = hh * (cs2 - 0.5 * cs1p);
      fp[d][0] = (live && k < K - 1) ? (pn - p[d]) / h : 0.0;
      fp[d][1] = p[d];
      fp[d][2] = live ? h * cs1 : 0.0;  // lanes beyond the horizon hold the running totals: keep their rows at zero,
      fp[d][3] = live ? hh * (cs2 + 0.5 * cs1) : 0.0;  // or the next step's suffix sums would pick them up
      if (!use_T && c1 > c0) {
        st_granules(my_cell + 2 * d, tag, qp[d]);
        Qt[(wave * 64 + k) * D + d] = qp[d];
      }
    }
    rz = wave_incl_sum(rz);
    if (lane == 63) red[0][wave] = rz;
    __syncthreads();
    PSTAMP(3);
    // ---- working rows: partner cells (polled until they carry this step's tag), eta . d(S0 p) ---------------------------
    {
      double sq = 0.0;
      unsigned spins = 0;
      bool bad = false;
      for (int e = threadIdx.x; e < ne; e += NT) {
        const int code = e_code[e];
        const int ek = code & 0xFF, al = (code >> 8) & 0xFF, side = (code >> 16) & 1;
        const u64* pc = A.cells + e_pad[e];
        u32x4 w[D];
        for (;;) {
          ld_cell<D>(pc, w);
          bool valid = true;
#pragma unroll
          for (int d = 0; d < D; ++d) valid = valid && pair_ok(w[d], tag);
          if (valid) break;
          if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u &&
                                       __hip_atomic_load(A.give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            bad = true;
            break;
          }
          spin_nap(A.spin_sleep);
        }
        if (bad) break;
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double pp = pair_value(w[d]);
          e_pp[(size_t)e * D + d] = pp;
          s += e_c[(size_t)e * D + d] * (Qt[(al * 64 + ek) * D + d] - pp);
        }
        if (!side) sq += s * s;  // every row once
      }
      if (bad) {
        __hip_atomic_store(A.give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fail_s = 1;
      }
      sq = wave_incl_sum(sq);
      if (lane == 63) red[1][wave] = sq;
    }
    __syncthreads();
    if (fail_s) { ok = false; break; }
    PSTAMP(4);
    // ---- all-gather of the two partials of every workgroup -----------------------------------------------------------------
    if (threadIdx.x < 2) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < APB; ++w) t += red[threadIdx.x][w];
      st_granules(gpart + (size_t)blockIdx.x * 4 + 2 * threadIdx.x, tag, t);
    }
    {
      unsigned spins = 0;
      bool bad = false;
      for (int q = threadIdx.x; q < 2 * nblk; q += NT) {  // one double (two granules) per thread and pass
        u32x4 w;
        for (;;) {
          w = ld_pair(gpart + 2 * q);
          if (pair_ok(w, tag)) break;
          if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u &&
                                       __hip_atomic_load(A.give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            bad = true;
            break;
          }
          spin_nap(A.spin_sleep);
        }
        if (bad) break;
        gp[q] = pair_value(w);
      }
      if (bad) {
        __hip_atomic_store(A.give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fail_s = 1;
      }
    }
    __syncthreads();
    if (fail_s) { ok = false; break; }
    PSTAMP(5);
    double a;
    {  // every wave sums the partials in the same order: the same bits everywhere, no further barrier
      double vr = 0.0, vs = 0.0;
      for (int b = lane; b < nblk; b += 64) {
        vr += gp[2 * b];
        vs += gp[2 * b + 1];
      }
      const double rzt = read_lane(wave_incl_sum(vr), 63);
      const double sqt = read_lane(wave_incl_sum(vs), 63);
      const double pHp = rzt + rho_c * sqt;
      a = (pHp > 0.0 && rzt != 0.0) ? rzt / pHp : 0.0;
    }
    const double aa = alpha * a;
    PSTAMP(6);
    // ---- everything after the step length is elementwise -------------------------------------------------------------------
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double zh = alpha * fma(a, fp[d][t], fx[d][t]) + (1.0 - alpha) * z[d][t];
        const double yo = y[d][t];
        const double zn = fmin(fmax(zh + yo / rr[t], lo[d][t]), hi[d][t]);
        const double yn = yo + rr[t] * (zh - zn);
        dy[d][t] = yn - yo;
        y[d][t] = yn;
        z[d][t] = zn;
        fx[d][t] = fma(aa, fp[d][t], fx[d][t]);
      }
      x[d] = fma(aa, p[d], x[d]);
      qx[d] = fma(aa, qp[d], qx[d]);
    }
    const bool last = it == nit - 1;
    for (int e = threadIdx.x; e < ne; e += NT) {
      const int code = e_code[e];
      const int ek = code & 0xFF, al = (code >> 8) & 0xFF;
      double tc = 0.0, ax = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const double c = e_c[(size_t)e * D + d];
        const double po = Qt[(al * 64 + ek) * D + d], pp = e_pp[(size_t)e * D + d];
        const double qo = e_qo[(size_t)e * D + d], qq = e_qp[(size_t)e * D + d];
        tc += c * (fma(a, po, qo) - fma(a, pp, qq));
        const double qon = fma(aa, po, qo), qqn = fma(aa, pp, qq);
        ax += c * (qon - qqn);
        e_qo[(size_t)e * D + d] = qon;
        e_qp[(size_t)e * D + d] = qqn;
      }
      const double zo = e_z[e], yo = e_y[e];
      const double zh = alpha * tc + (1.0 - alpha) * zo;
      const double zn = fmax(zh + yo / rho_c, e_l[e]);
      const double yn = yo + rho_c * (zh - zn);
      e_z[e] = zn;
      e_y[e] = yn;
      e_g[e] = (rho_c * zn - yn) - rho_c * ax;
      if (last && with_dy) e_pp[(size_t)e * D] = fmin(yn - yo, 0.0);  // delta-y of the batch's last step (u = +inf: polar of
                                                                      // the recession cone), parked until the check
    }
    __syncthreads();
    PSTAMP(7);
  }

  if (!ok) break;
  it_done += nit;
  // ==== termination check (replaces 4 launches + a host round trip per check) ==========================================
  // Same quantities as scp_qp_fused_residuals: F x and S0 x are rebuilt exactly from x (the carried slabs are refreshed,
  // as the three-launch pipeline does at every check), A^T y = F^T y_f + S0^T G(y_c) by suffix scans, the primal
  // infeasibility certificate from delta-y of the last step.  One more neighbour hand-off (exact S0 x cells, tagged with
  // bit 31 set) and one all-gather of nine values per workgroup.
  {
    const unsigned ctag = 0x80000000u | (A.epoch0 + steps);
    double m[NCHK];
#pragma unroll
    for (int j = 0; j < NCHK; ++j) m[j] = 0.0;
    {
      double gy[D], gd[D];
#pragma unroll
      for (int d = 0; d < D; ++d) gy[d] = gd[d] = 0.0;
      for (int e = c0; e < c1; ++e) {
        const double ye = e_y[e], de = with_dy ? e_pp[(size_t)e * D] : 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          gy[d] += e_c[(size_t)e * D + d] * ye;
          gd[d] += e_c[(size_t)e * D + d] * de;
        }
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        // exact F x, S0 x from x (forward scans)
        const double cs1 = wave_incl_sum(x[d]);
        const double cs2 = lane_below(wave_incl_sum(cs1));
        const double cs1p = lane_below(cs1);
        const double xn = lane_above(x[d]);
        fx[d][0] = (live && k < K - 1) ? (xn - x[d]) / h : 0.0;
        fx[d][1] = x[d];
        fx[d][2] = live ? h * cs1 : 0.0;
        fx[d][3] = live ? hh * (cs2 + 0.5 * cs1) : 0.0;
        qx[d] = hh * (cs2 - 0.5 * cs1p);
        if (c1 > c0) {
          st_granules(my_cell + 2 * d, ctag, qx[d]);
          Qt[(wave * 64 + k) * D + d] = qx[d];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (live && (t > 0 || k < K - 1)) {
            m[CK_RP] = fmax(m[CK_RP], fabs(fx[d][t] - z[d][t]));
            m[CK_NAX] = fmax(m[CK_NAX], fabs(fx[d][t]));
            m[CK_NZ] = fmax(m[CK_NZ], fabs(z[d][t]));
            if (with_dy) {
              m[CK_NDY] = fmax(m[CK_NDY], fabs(dy[d][t]));
              m[CK_SUPP] += hi[d][t] * fmax(dy[d][t], 0.0) + lo[d][t] * fmin(dy[d][t], 0.0);
            }
          }
        }
        // A^T y (and A^T delta-y): the r chain with W' -> y
        for (int pass = 0; pass < (with_dy ? 2 : 1); ++pass) {
          const double vj = pass ? dy[d][0] : y[d][0], va = pass ? dy[d][1] : y[d][1];
          const double vv = pass ? dy[d][2] : y[d][2], vp = pass ? dy[d][3] : y[d][3];
          const double g = pass ? gd[d] : gy[d];
          const double u1 = h * vv + 0.5 * hh * (vp - g);
          const double u2 = vp + g;
          const double d1 = wave_incl_rsum(u1);
          const double s1 = wave_incl_rsum(u2);
          const double d2 = lane_above(wave_incl_rsum(s1));
          const double vjm = lane_below(vj);
          const double at = ((vjm - vj) / h + va) + (d1 + 0.5 * hh * g) + hh * d2;
          if (live) {
            if (pass) {
              m[CK_NATDY] = fmax(m[CK_NATDY], fabs(at));
            } else {
              const double px = 2.0 * x[d];
              m[CK_RD] = fmax(m[CK_RD], fabs(px + at));
              m[CK_NPX] = fmax(m[CK_NPX], fabs(px));
              m[CK_NATY] = fmax(m[CK_NATY], fabs(at));
            }
          }
        }
      }
    }
    __syncthreads();
    {  // collision rows: exact S0 x cells of both agents (the carried copies are refreshed), residuals, delta-y
      unsigned spins = 0;
      bool bad = false;
      for (int e = threadIdx.x; e < ne; e += NT) {
        const int code = e_code[e];
        const int ek = code & 0xFF, al = (code >> 8) & 0xFF, side = (code >> 16) & 1;
        const u64* pc = A.cells + e_pad[e];
        u32x4 w[D];
        for (;;) {
          ld_cell<D>(pc, w);
          bool valid = true;
#pragma unroll
          for (int d = 0; d < D; ++d) valid = valid && pair_ok(w[d], ctag);
          if (valid) break;
          if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u &&
                                       __hip_atomic_load(A.give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            bad = true;
            break;
          }
          spin_nap(A.spin_sleep);
        }
        if (bad) break;
        double ax = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double qq = pair_value(w[d]);
          const double qo = Qt[(al * 64 + ek) * D + d];
          e_qo[(size_t)e * D + d] = qo;
          e_qp[(size_t)e * D + d] = qq;
          ax += e_c[(size_t)e * D + d] * (qo - qq);
        }
        if (!side) {
          const double zc_ = e_z[e];
          m[CK_RP] = fmax(m[CK_RP], fabs(ax - zc_));
          m[CK_NAX] = fmax(m[CK_NAX], fabs(ax));
          m[CK_NZ] = fmax(m[CK_NZ], fabs(zc_));
          if (with_dy) {
            const double dd = e_pp[(size_t)e * D];
            m[CK_NDY] = fmax(m[CK_NDY], fabs(dd));
            m[CK_SUPP] += e_l[e] * dd;
          }
        }
      }
      if (bad) {
        __hip_atomic_store(A.give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fail_s = 1;
      }
    }
#pragma unroll
    for (int j = 0; j < NCHK; ++j) {
      const double v = j == CK_SUPP ? wave_incl_sum(m[j]) : wave_max_nn(m[j]);
      if (lane == 63) red[j][wave] = v;
    }
    __syncthreads();
    if (fail_s) { ok = false; break; }
    if (threadIdx.x < NCHK) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < APB; ++w) t = threadIdx.x == CK_SUPP ? t + red[threadIdx.x][w] : fmax(t, red[threadIdx.x][w]);
      st_granules(A.gcheck + ((size_t)blockIdx.x * NCHK + threadIdx.x) * 2, ctag, t);
    }
    {
      unsigned spins = 0;
      bool bad = false;
      for (int q = threadIdx.x; q < NCHK * nblk; q += NT) {
        u32x4 w;
        for (;;) {
          w = ld_pair(A.gcheck + 2 * q);
          if (pair_ok(w, ctag)) break;
          if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u &&
                                       __hip_atomic_load(A.give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            bad = true;
            break;
          }
          spin_nap(A.spin_sleep);
        }
        if (bad) break;
        gp[q] = pair_value(w);
      }
      if (bad) {
        __hip_atomic_store(A.give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fail_s = 1;
      }
    }
    __syncthreads();
    if (fail_s) { ok = false; break; }
#pragma unroll
    for (int j = 0; j < NCHK; ++j) {  // the same reduction order in every wave of every workgroup: identical decisions
      double v = 0.0;
      for (int b = lane; b < nblk; b += 64) v = j == CK_SUPP ? v + gp[b * NCHK + j] : fmax(v, gp[b * NCHK + j]);
      chk[j] = read_lane(j == CK_SUPP ? wave_incl_sum(v) : wave_max_nn(v), 63);
    }
    __syncthreads();  // gp is reused by the next step's all-gather
    // ---- decide (the host repeats these tests on the same nine numbers, scp_qp_solve) ------------------------------------
    const double np_ = fmax(chk[CK_NAX], chk[CK_NZ]), nd_ = fmax(chk[CK_NPX], chk[CK_NATY]);
    const double tol_p = A.eps_abs + A.eps_rel * np_, tol_d = A.eps_abs + A.eps_rel * nd_;
    if (chk[CK_RP] <= tol_p && chk[CK_RD] <= tol_d) { exit_code = EXIT_SOLVED; break; }
    if (A.check_fine > 0)  // (the same decision as scp_qp_solve's, from the same nine numbers)
      cad = (chk[CK_RP] < A.fine_ratio * tol_p && chk[CK_RD] < A.fine_ratio * tol_d) ? A.check_fine : A.check_every;
    if (it_done >= A.max_iter) { exit_code = EXIT_MAX_ITER; break; }
    if (with_dy && chk[CK_NDY] > A.eps_prim_inf && chk[CK_SUPP] < -A.eps_prim_inf * chk[CK_NDY] &&
        chk[CK_NATDY] < A.eps_prim_inf * chk[CK_NDY]) { exit_code = EXIT_INFEASIBLE; break; }
    if (A.rho_tol > 0.0 && it_done % A.rho_interval == 0) {
      // OSQP's rho estimate, snapped to the 2^(1/4) grid as the host does it (scp_qp_solve).  Device log2 / exp2 may differ
      // from the host's in the last bit, so the candidate only SELECTS: the value that counts is the host-computed double
      // in the table of cached rho, and the threshold test is repeated on it exactly as the host would.
      const double prim = chk[CK_RP] / fmax(np_, 1e-10), dual = chk[CK_RD] / fmax(nd_, 1e-10);
      const double nr = fmin(fmax(rho * sqrt(prim / fmax(dual, 1e-10)), 1e-6), 1e6);
      const double cand = exp2(round(4.0 * log2(nr)) * 0.25);
      if (cand > rho * A.rho_tol * (1.0 - 1e-9) || cand < rho / A.rho_tol * (1.0 + 1e-9)) {  // (else: clearly no update)
        int slot = -1;
        for (int i = 0; i < A.n_tab; ++i)
          if (fabs(A.tab[i].rho - cand) <= 1e-12 * cand) slot = i;
        if (slot < 0) { exit_code = EXIT_RHO; break; }  // not cached yet: the host builds the blocks and relaunches
        const double nrs = A.tab[slot].rho;
        if (nrs > rho * A.rho_tol || nrs < rho / A.rho_tol) {
          // ---- switch rho in place (what the host does between two launches: build_kkt hit + rows_value_kernel) -------------
          rho = nrs;
          rho_c = rho * A.rho_col_scale;
#pragma unroll
          for (int t = 0; t < 4; ++t) rr[t] = (t >= 2 && k == K - 1) ? rho * A.rho_eq : rho;
          tile_prefetch<CHB>(use_T && wave >= tK ? A.tab[slot].pT : A.tab[slot].pMinv, nks,
                             wave < tK ? wave : (use_T && wave < 2 * tK ? wave - tK : 0), 0, nks, aM);
          for (int e = threadIdx.x; e < ne; e += NT) {  // row values of the next right-hand side from the exact S0 x cells
            double ax = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) ax += e_c[(size_t)e * D + d] * (e_qo[(size_t)e * D + d] - e_qp[(size_t)e * D + d]);
            e_g[e] = (rho_c * e_z[e] - e_y[e]) - rho_c * ax;
          }
          __syncthreads();
          ++n_rho;
          if (A.check_fine > 0) cad = A.check_fine;
        }
      }
    }
  }
  }  // batches

  // The exit decision is collective: a workgroup that timed out has raised the give-up word BEFORE the cell or partial it
  // was waiting for appeared, so every workgroup that got past that exchange afterwards sees the word here and leaves
  // without writing back as well (the host additionally drops its carried-state flags on a give-up).
  if (ok && __syncthreads_or(threadIdx.x == 0 &&
                             __hip_atomic_load(A.give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))
    ok = false;
  if (!ok) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(A.host_status, (unsigned)EXIT_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(A.host_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;  // nothing was written back: the state in global memory is the state before this launch
  }
  // ---- write the state back ---------------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool rok = live && (t > 0 || k < K - 1);
    const int row = t == 0 ? k : t * K - 1 + k;
    if (rok) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int64_t g = (int64_t)row * C + (int64_t)agent * D + d;
        A.zf[g] = z[d][t];
        A.yf[g] = y[d][t];
        A.fx[g] = fx[d][t];
        if (with_dy) A.dyf[g] = dy[d][t];
      }
    }
  }
  if (live) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int64_t g = (int64_t)k * C + (int64_t)agent * D + d;
      A.x[g] = x[d];
      A.Qx[g] = qx[d];
    }
  }
  for (int e = threadIdx.x; e < ne; e += NT) {
    A.gval[ebase + e] = e_g[e];
    if (!((e_code[e] >> 16) & 1)) {
      A.zc[e_row[e]] = e_z[e];
      A.yc[e_row[e]] = e_y[e];
      if (with_dy) A.dyc[e_row[e]] = e_pp[(size_t)e * D];
    }
  }
  PSTAMP(8);
#ifdef SCP_PHASE_PROFILE
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) scp_persist_clk[i] = i == 15 ? (unsigned long long)steps : pacc[i];
#endif
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the nine check results in the slots the host reads (scp_qp::h_scal), then the exit code and the completion word
    const int slot[NCHK] = {SL_RP, SL_NAX, SL_NZ, SL_RD, SL_NPX, SL_NATY, SL_NDY, SL_SUPP, SL_NATDY};
#pragma unroll
    for (int j = 0; j < NCHK; ++j)
      __hip_atomic_store((u64*)(A.host_scal + slot[j]), (u64)__double_as_longlong(chk[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_status + 1, (unsigned)it_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_status + 2, n_rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((u64*)A.host_rho, (u64)__double_as_longlong(rho), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_status, exit_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

size_t persist_lds_bytes(int K, int D, int cap, int nblk) {
  const int apb = persist_apb(D);
  size_t dbl = (size_t)3 * CB * pad_col(K) + (size_t)apb * 64 * D + (size_t)NCHK * nblk + (size_t)cap * (4 * D + 4);
  size_t ints = (size_t)3 * cap + (size_t)apb * K + 1;
  return dbl * sizeof(double) + ((ints + 1) / 2 * 2) * sizeof(int);
}

}  // namespace

// Can the persistent kernel run this QP?  (shape limits; the entry capacity is checked per working set)
// which kernel runs this shape: 0 = the round-2 kernel, one wave per agent with 8 (3-D: 4) agents per workgroup and the full
// z / y / F x state in registers; 1 = the lean kernel, 16 agents per workgroup (2-D, 2048 < N <= 4096); 2 = the lean kernel's
// state diet with 8 agents per workgroup (2-D up to 2048 agents: 5-10 % faster per step than the round-2 kernel, measured --
// profiles/r03_step_time_variants.txt; 3-D beyond 1024 agents); -1 = none.  settings.persistent: 1 = this choice, 2 / 3 / 4
// force variant 1 / 2 / 0 where it fits.
static int persist_variant_for(const scp_qp* qp) {
  // one workgroup per CU, all resident (grid-wide rendezvous); gpart / gcheck hold SCP_PERSIST_MAX_WG (+1) workgroups
  const int max_wg = std::min(qp->ctx->n_cu, SCP_PERSIST_MAX_WG);
  const int apb = persist_apb(qp->D);
  const bool fits_old = (qp->N + apb - 1) / apb <= max_wg;
  const bool fits16 = qp->D == 2 && (qp->N + 15) / 16 <= max_wg;
  const bool fits8 = (qp->N + 7) / 8 <= max_wg;
  if (qp->st.persistent == 2 && fits16) return 1;
  if (qp->st.persistent == 3 && fits8) return 2;
  if (qp->st.persistent == 4 && fits_old) return 0;
  if (qp->D == 2) return fits8 ? 2 : (fits16 ? 1 : -1);
  return fits_old ? 0 : (fits8 ? 2 : -1);  // 3-D: the 4-agent kernel is the faster one up to 1024 agents
}
static int variant_apb(int variant, int D) { return variant == 1 ? 16 : (variant == 2 ? 8 : persist_apb(D)); }

bool scp_qp_persist_eligible(const scp_qp* qp) {
  if (!qp->st.persistent || qp->st.cg_iters != 1 || qp->st.use_mfma != 1) return false;
  if (qp->K > 64 || qp->nW <= 0 || qp->persist_off) return false;
  return persist_variant_for(qp) >= 0;
}

// Run ADMM iterations from iteration count `it0` of the current solve in ONE launch, termination checks included, until
// the kernel has something for the host to decide.  *ran = 0: this working set does not fit, nothing was enqueued.
// Otherwise the call waits for the kernel's completion word (mapped host memory; the host spins instead of sleeping in
// hipStreamSynchronize) and returns its exit code (*code; SCP_PERSIST_GAVE_UP: nothing was written back, the caller
// repeats the iterations on the three-launch pipeline), the iteration count reached (*it_done) and, in qp->h_scal, the
// nine results of the last check.
int scp_qp_cg1_persist(scp_qp* qp, int it0, int cad0, int* ran, int* code, int* it_done) {
  const QpDev& d = qp->d;
  scp_ctx* ctx = qp->ctx;
  hipStream_t s = ctx->stream;
  const scp_qp_settings& st = qp->st;
  const int K = qp->K, D = qp->D;
  const int64_t C = qp->C, nx = (int64_t)K * C;
  const int variant = persist_variant_for(qp);
  if (variant < 0) { *ran = 0; return SCP_OK; }
  const bool lean = variant >= 1;
  const int apb = variant_apb(variant, D);
  const int nblk = (qp->N + apb - 1) / apb;
  *ran = 0;
  if (!qp->cg1_ready) {
    int rc = scp_qp_cg1_prepare(qp);
    if (rc) return rc;
  }
  // Entry tables: the workgroup is alone on its CU anyway (228 VGPRs x 8 waves), so it simply takes all the LDS there is
  // -- no read-back of the largest block's entry count.  A block of agents with more incident rows than fit (> ~1300 at
  // K = 50) makes the kernel leave at once with EXIT_OVERFLOW; that working set then runs on the three-launch pipeline.
  if (qp->persist_cap_nW == qp->nW) return SCP_OK;  // this working set overflowed before
  {
    const int nb = nblk + 1;  // (+1: the fault-injection hook below may announce one more workgroup)
    const size_t fixed = lean ? scp_persist16_lds_bytes(K, 0, nb, D, apb) : persist_lds_bytes(K, D, 0, nb);
    const size_t per_entry = (size_t)(4 * D + 4) * sizeof(double) + (lean ? 1 : 3) * sizeof(int);
    const size_t budget_lds = 160 * 1024 - (lean ? 2048 : 1024);  // minus the static __shared__ of the kernel (580 B / 1.2 KB)
    if (fixed + 64 * per_entry > budget_lds) return SCP_OK;
    qp->persist_cap = (int)((budget_lds - fixed) / per_entry / 64 * 64);
  }
  // test hook (scp_qp_debug_set "persist_fault"): expect one workgroup more than is launched, so that the all-gather can
  // never complete -- the bounded spins must time out, every workgroup must leave without writing state back, and the
  // host must carry on with the three-launch pipeline
  const int nblk_expected = nblk + (qp->persist_fault > 0 ? 1 : 0);
  const size_t lds = lean ? scp_persist16_lds_bytes(K, qp->persist_cap, nblk_expected, D, apb)
                          : persist_lds_bytes(K, D, qp->persist_cap, nblk_expected);
  if (lds > 160 * 1024) return SCP_OK;  // too many rows around one block of agents: three-launch pipeline
  const int budget = st.max_iter - it0;  // at most this many steps in this launch
  if (budget <= 0) return SCP_OK;  // nothing to run (the kernel's check would read tags and delta-y of an earlier launch)
  // Residency: every workgroup must be on the chip at once.  Claim one CU per workgroup among the persistent launches of
  // this device (all processes of this user, scp_persist_claim); a claim that does not fit within 200 ms sends this solve
  // to the three-launch pipeline instead of letting 2^20 polls find out.
  if (!scp_persist_claim(ctx->device, ctx->n_cu, nblk, 200)) {
    qp->persist_skip_solve = true;
    return SCP_OK;
  }
  if (qp->persist_fault > 0) --qp->persist_fault;
  struct ClaimGuard {
    int device, n;
    ~ClaimGuard() { scp_persist_release(device, n); }
  } claim_guard{ctx->device, nblk};
  PersistArgs a;
  a.K = K; a.N = qp->N; a.nblk = nblk_expected; a.ent_cap = qp->persist_cap;
  a.it0 = it0; a.max_iter = st.max_iter; a.check_every = st.check_termination;
  a.cad0 = cad0 > 0 ? cad0 : st.check_termination; a.fine_ratio = st.check_fine_ratio;
  a.check_fine = (st.check_fine > 0 && st.check_fine < st.check_termination && st.check_termination % st.check_fine == 0 && C <= 4096)
                     ? st.check_fine : 0;  // (the rule of scp_qp_solve; a persistent launch always has collision rows)
  a.rho_interval = st.adaptive_rho_interval > 0 ? st.adaptive_rho_interval : 1;
  a.C = C;
  a.rho = qp->rho; a.rho_c = qp->rho * st.rho_col_scale; a.rho_eq = st.rho_eq_scale; a.alpha = st.alpha; a.h = qp->h;
  a.eps_abs = st.eps_abs; a.eps_rel = st.eps_rel; a.eps_prim_inf = st.eps_prim_inf;
  a.vel_lo = qp->lim[0]; a.vel_hi = qp->lim[1];
  a.acc_lo = qp->lim[2]; a.acc_hi = qp->lim[3]; a.jerk_lo = qp->lim[4]; a.jerk_hi = qp->lim[5];
  for (int dd = 0; dd < 3; ++dd) { a.pmin[dd] = qp->space[dd]; a.pmax[dd] = qp->space[3 + dd]; }
  a.states = d.states;
  a.first_step = qp->steps_since_reset == 0 ? 1 : 0;  // (y = 0 and z = A x0 then: scp_qp_reset)
  {
    static const int spin_sleep = [] {  // developer knob (tools/batch_rate.sh experiments)
      const char* e = getenv("SCP_PERSIST_SPIN_SLEEP");
      const int v = e ? atoi(e) : 1;
      return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    a.spin_sleep = spin_sleep;
  }
  a.rho_tol = (st.adaptive_rho && st.adaptive_rho_interval > 0) ? st.adaptive_rho_tolerance : 0.0;
  a.pMinv = d.pMinv;
  a.pT = d.pT;
  a.lf = d.lf; a.uf = d.uf; a.zf = d.zf; a.yf = d.yf; a.fx = d.fx; a.x = d.x;
  a.Qx = qp->qx_sel ? d.HQ : d.HQ + nx;
  a.dyf = d.dyf;
  a.cells = d.cells;
  a.gpart = d.gpart;
  a.gcheck = d.gcheck;
  a.give_up = (unsigned*)d.sync_words;
  a.cell_ptr = d.cell_ptr; a.ent_code = d.ent_code; a.w_k = d.w_k; a.w_i = d.w_i; a.w_j = d.w_j;
  a.w_eta = d.w_eta; a.w_l = d.w_l; a.zc = d.zc; a.yc = d.yc; a.dyc = d.dyc; a.gval = d.gval;
  a.host_status = qp->h_persist_dev;
  a.host_scal = qp->h_scal_dev;
  a.host_flag = (u64*)(qp->h_scal_dev + SL_COUNT + SCP_RESID_CAP);
  a.host_rho = qp->h_scal_dev + SL_COUNT + SCP_RESID_CAP + 3;
  a.rho_col_scale = st.rho_col_scale;
  a.n_tab = 0;
  for (int i = 0; i < qp->n_kkt; ++i)  // every rho whose blocks are resident (same sigma)
    if (qp->kkt[i].used && qp->kkt[i].sigma == st.sigma) {
      a.tab[a.n_tab].rho = qp->kkt[i].rho;
      a.tab[a.n_tab].pMinv = qp->kkt[i].pMinv;
      a.tab[a.n_tab].pT = qp->kkt[i].pT;
      ++a.n_tab;
    }
  a.seq = ++qp->check_seq;
  qp->h_persist[0] = 0u;
  qp->h_persist[1] = (unsigned)it0;
  qp->h_persist[2] = 0u;
  qp->persist_rho_switches = 0;
  if (qp->persist_epoch == 0 || qp->persist_epoch + (u64)budget >= 0x7FFFFFF0ull) {
    // first launch, the one after a give-up, or the step tags would reach bit 31 (reserved for the checks): every polled
    // word starts from zero
    SCP_HIP_CHECK(ctx, hipMemsetAsync(d.sync_words, 0, 16, s));
    SCP_HIP_CHECK(ctx, hipMemsetAsync(d.cells, 0, (size_t)K * C * 2 * sizeof(u64), s));
    SCP_HIP_CHECK(ctx, hipMemsetAsync(d.gpart, 0, (size_t)SCP_GPART_WORDS * sizeof(u64), s));
    SCP_HIP_CHECK(ctx, hipMemsetAsync(d.gcheck, 0, (size_t)SCP_GCHECK_WORDS * sizeof(u64), s));
    qp->persist_epoch = 0;
  }
  a.epoch0 = (unsigned)qp->persist_epoch;
  qp->persist_variant = variant;
  if (lean) {
    int rc = scp_persist16_launch(ctx, a, nblk, lds, D, apb);
    if (rc) return rc;
  } else if (D == 2) {
    if (lds > 64 * 1024)
      SCP_HIP_CHECK(ctx, scp_raise_lds_limit(ctx->device, reinterpret_cast<const void*>(cg1_persist_kernel<2>), lds));
    hipLaunchKernelGGL(cg1_persist_kernel<2>, dim3(nblk), dim3(64 * persist_apb(2)), lds, s, a);
  } else {
    if (lds > 64 * 1024)
      SCP_HIP_CHECK(ctx, scp_raise_lds_limit(ctx->device, reinterpret_cast<const void*>(cg1_persist_kernel<3>), lds));
    hipLaunchKernelGGL(cg1_persist_kernel<3>, dim3(nblk), dim3(64 * persist_apb(3)), lds, s, a);
  }
  SCP_HIP_CHECK(ctx, hipGetLastError());
  *ran = 1;
  {
    volatile u64* flag = (volatile u64*)(qp->h_scal + SL_COUNT + SCP_RESID_CAP);
    if (!scp_wait_host_word(flag, a.seq, 30)) SCP_HIP_CHECK(ctx, hipStreamSynchronize(s));  // a fault surfaces here
    if (*flag != a.seq) return scp_fail(ctx, SCP_ERR_HIP, "persistent kernel: completion word not written");
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  *code = (int)((volatile unsigned*)qp->h_persist)[0];
  *it_done = (int)((volatile unsigned*)qp->h_persist)[1];
  if (*code == EXIT_OVERFLOW) {  // nothing ran, nothing was published: this working set stays on the three-launch pipeline
    qp->persist_cap_nW = qp->nW;
    *ran = 0;
    return SCP_OK;
  }
  if (*code == SCP_PERSIST_GAVE_UP) {
    // Nothing should have been written back; if a workgroup got through after another had timed out (the kernel's exit
    // decision closes that window but cannot exclude it), the carried slabs no longer match x / z / y: rebuild them.
    qp->cg1_ready = false;
    qp->qx_fresh = false;
    ++qp->persist_gave_up_total;
  }
  if (*code != SCP_PERSIST_GAVE_UP) {
    qp->persist_epoch += (u64)(*it_done - it0);  // one tag per ADMM step
    // rho switches the kernel made by itself (scp_qp_solve adopts the value and points d.* at that slot)
    qp->persist_rho_switches = (int)((volatile unsigned*)qp->h_persist)[2];
    qp->persist_rho = ((volatile double*)qp->h_scal)[SL_COUNT + SCP_RESID_CAP + 3];
  }
  return SCP_OK;
}

// ---- residency bookkeeping of the persistent launches ---------------------------------------------------------------------
// One table per (user, GPU) in POSIX shared memory, so that solver threads of SEVERAL processes on one GPU (compute-
// trajectories-batch: processes x streams) account for each other: `total` = CUs claimed by launches in flight; every
// process owns a slot {pid, claimed} so that the claims of a process that died can be taken back (kill(pid, 0) == ESRCH).
// Without shared memory (no /dev/shm) the table is process-local.
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <mutex>

namespace {
constexpr int RES_SLOTS = 128, RES_DEVICES = 16;
struct ResTable {
  std::atomic<int> total;
  struct Slot {
    std::atomic<int> pid, claimed;
  } slot[RES_SLOTS];
};
ResTable g_res_local[RES_DEVICES];
ResTable* g_res[RES_DEVICES];
int g_res_slot[RES_DEVICES];
std::mutex g_res_mu;

ResTable* res_table(int device, int* my_slot) {
  const int dv = device >= 0 && device < RES_DEVICES ? device : 0;
  std::lock_guard<std::mutex> lk(g_res_mu);
  if (!g_res[dv]) {
    ResTable* t = nullptr;
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) snprintf(bus, sizeof(bus), "dev%d", device);
    for (char* c = bus; *c; ++c)
      if (*c == ':' || *c == '.' || *c == '/') *c = '_';
    char name[128];
    snprintf(name, sizeof(name), "/scp_hip_persist_%u_%s", (unsigned)getuid(), bus);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd >= 0) {
      if (ftruncate(fd, sizeof(ResTable)) == 0) {
        void* m = mmap(nullptr, sizeof(ResTable), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) t = static_cast<ResTable*>(m);  // (a fresh segment is zero-filled: empty table)
      }
      close(fd);
    }
    if (!t) t = &g_res_local[dv];
    // this process's slot: a free one, or one whose owner is gone
    const int me = (int)getpid();
    int mine = -1;
    for (int i = 0; i < RES_SLOTS && mine < 0; ++i) {
      int owner = t->slot[i].pid.load();
      if (owner == me) {  // a slot left by an earlier process with this pid: its claims are stale
        const int lost = t->slot[i].claimed.exchange(0);
        if (lost > 0) t->total.fetch_sub(lost);
        mine = i;
        break;
      }
      if (owner != 0 && !(kill(owner, 0) != 0 && errno == ESRCH)) continue;
      if (owner != 0) {  // dead owner: its claims go back first
        const int lost = t->slot[i].claimed.exchange(0);
        if (lost > 0) t->total.fetch_sub(lost);
      }
      if (t->slot[i].pid.compare_exchange_strong(owner, me)) mine = i;
    }
    g_res[dv] = t;
    g_res_slot[dv] = mine;  // (-1: table full -- claims still count in `total`, only the crash recovery is lost)
  }
  *my_slot = g_res_slot[dv];
  return g_res[dv];
}

// give back the claims of processes that no longer exist
void res_reap(ResTable* t) {
  for (int i = 0; i < RES_SLOTS; ++i) {
    int owner = t->slot[i].pid.load();
    if (owner == 0 || owner == (int)getpid()) continue;
    if (kill(owner, 0) != 0 && errno == ESRCH && t->slot[i].pid.compare_exchange_strong(owner, 0)) {
      const int lost = t->slot[i].claimed.exchange(0);
      if (lost > 0) t->total.fetch_sub(lost);
    }
  }
}
}  // namespace

bool scp_persist_claim(int device, int n_cu_total, int n_wg, int wait_ms) {
  int slot = -1;
  ResTable* t = res_table(device, &slot);
  const auto t0 = std::chrono::steady_clock::now();
  bool reaped = false;
  for (;;) {
    const int before = t->total.fetch_add(n_wg);
    if (before + n_wg <= n_cu_total) {
      if (slot >= 0) t->slot[slot].claimed.fetch_add(n_wg);
      return true;
    }
    t->total.fetch_sub(n_wg);
    if (!reaped) {  // claims of a crashed process would block everybody for ever
      res_reap(t);
      reaped = true;
      continue;
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(wait_ms)) return false;
    struct timespec ts = {0, 20000};
    nanosleep(&ts, nullptr);
  }
}

void scp_persist_release(int device, int n_wg) {
  int slot = -1;
  ResTable* t = res_table(device, &slot);
  if (slot >= 0) t->slot[slot].claimed.fetch_sub(n_wg);
  t->total.fetch_sub(n_wg);
}

#ifdef SCP_PHASE_PROFILE
// developer hook of the profiling build only (not declared in include/scp_hip.h)
extern "C" int scp_debug_persist_clocks(unsigned long long* out, int n) {
  if (n > 16) n = 16;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scp_persist_clk), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
