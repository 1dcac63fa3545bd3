// Lean form of the persistent single-step ADMM kernel: SIXTEEN agents (= waves) per workgroup, so that 2-D problems of up to
// 16 x 256 = 4096 agents keep one workgroup per compute unit and stay on the persistent path (config 4 of BASELINE.json on
// one GPU; the 8-agent kernel of scp_qp_persist.hip stops at 2048 agents and the three-launch pipeline behind it costs
// 51 us per ADMM step at 4096 x 50 against ~10 us here).  Replaces osqp's solve loop at
// /root/reference/src/path_planning/solvers/scp.py:441-445 like its sibling; same algorithm, same exchanges, same exit
// protocol (scp_qp_persist_device.h) -- what changes is the register diet that four waves per SIMD (128 registers each)
// demand:
//   * ONE double per fixed row instead of two: v = z~ + y / rho, the point the projection is applied to.  ADMM's update
//     z' = Pi(v'), y' = rho (v' - Pi(v')) with v' = alpha F x~ + (1 - alpha) z + y / rho keeps z = Pi(v), y = rho (v - Pi(v)) as an
//     invariant, so v' = v + alpha (F x~ - Pi(v)) and z, y are two clamps away wherever they are needed ("one
//     state double per row").  The invariant holds after any ADMM update but not after scp_qp_reset (z = A x0 unprojected,
//     y = 0): the first step of a QP takes z = v itself (PersistArgs::first_step).  Rounding-level differences only;
//   * x and its first / second prefix sums c1 = cumsum(x), c2 = cumsum(c1) (exclusive) are carried, and F x, S0 x re-derived
//     from them where the 8-agent kernel carries the F x slab (16 registers) and S0 x; F (x + a p) likewise comes from
//     (c1 + a cumsum p, c2 + a cumsum^2 p), so F p is never formed.  Same quantities, sums associated differently;
//   * the jerk / acceleration limits are scalars (scp.py:188-195: the same for every row), only the velocity / position
//     bounds are per row;
//   * H_f^{-1} (packed MFMA operands, 26 KB at K = 50) lives in LDS and is streamed into the matrix cores, not held in 32
//     registers per lane; eight waves run the 4 row tiles x 2 column tiles of p = H_f^{-1} r; S0 p comes from two more
//     prefix scans (no T = S0 H_f^{-1} tiles: the matrix pipes are the busiest unit with four waves per SIMD);
//   * delta-y of a batch's last step (primal infeasibility certificate) is reduced on the spot -- |dy|, the support value and
//     |A^T dy| per lane -- instead of being kept in 16 registers until the check;
//   * y / rho and (.) / h are multiplications by reciprocals (fp64 division is ~14 VALU instructions a piece, eight of them
//     per lane and step).
// Entry tables: 100 B per incident row in the LDS that is left (~900 rows around a block of 16 agents at K = 50; more ->
// EXIT_OVERFLOW -> three-launch pipeline for that working set, as before).
#include "scp_qp_persist_device.h"

namespace {
using namespace scpdev;
using namespace scp_persist;

// Developer build (make prof): 100 MHz wall-clock ticks per phase, summed over the steps of one launch, middle workgroup,
// thread 0; the accumulators live in LDS (the kernel has no registers to spare).
#ifdef SCP_PHASE_PROFILE
__device__ unsigned long long scp_persist16_clk[16];
#define PSTAMP(slot)                                           \
  do {                                                         \
    if (prof_t) {                                              \
      const unsigned long long now_ = wall_clock64();          \
      pacc_s[slot] += now_ - pacc_s[15];                       \
      pacc_s[15] = now_;                                       \
    }                                                          \
  } while (0)
#else
#define PSTAMP(slot) ((void)0)
#endif

// Instantiations: <2, 16> the lean kernel proper (2-D, 2048 < N <= 4096: four waves per SIMD, 128 registers);
// <3, 8> 3-D with eight agents per workgroup (1024 < N <= 2048: the 4-agent 3-D kernel of scp_qp_persist.hip stops at 1024);
// <2, 8> the same state diet at two waves per SIMD (settings.persistent = 3: measurements against the 8-agent kernel).
constexpr int nct_of(int D, int APB) { return (D * APB + 15) / 16; }  // 16-column MFMA tiles of a workgroup
constexpr int AB_STRIDE = 32;  // per-agent bound table: 8 groups of 4 doubles

struct Lds16 {  // carve-up shared by the kernel and the host's size computation (doubles, then ints)
  int RSK, tK, nks;
  size_t rt, pt, ml, gp, ab, gchk, ent, n_dbl;
  __host__ __device__ Lds16(int K, int cap, int nblk, int D, int APB) {
    const int NC16 = 16 * nct_of(D, APB), APB16 = APB;
    RSK = pad_col(K);
    tK = (K + 15) >> 4;
    nks = (K + 3) >> 2;
    size_t o = 0;
    rt = o; o += (size_t)NC16 * RSK;
    pt = o; o += (size_t)NC16 * RSK;
    ml = o; o += (size_t)tK * nks * 64;
    gp = o; o += (size_t)2 * nblk;
    ab = o; o += (size_t)APB16 * AB_STRIDE;
    // the nine-value all-gather of a termination check reuses the two tiles when they are large enough
    if ((size_t)2 * NC16 * RSK >= (size_t)NCHK * nblk) gchk = rt;
    else { gchk = o; o += (size_t)NCHK * nblk; }
    ent = o; o += (size_t)cap * (4 * D + 4);
    n_dbl = o;
  }
};

template <int D, int APB16>
__global__ __launch_bounds__(64 * APB16) void cg1_persist16_kernel(PersistArgs A) {
  constexpr int NT16 = 64 * APB16;             // threads: one wave per agent
  constexpr int NCT = nct_of(D, APB16);        // column tiles of the MFMA phase
  constexpr int NC16 = 16 * NCT;               // tile columns (D APB16 of them in use)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red[NCHK][APB16];
  __shared__ double cert_s[3][APB16];  // per wave: |dy|, support value, |A^T dy| of the batch's last step (fixed rows)
  __shared__ double chk_s[NCHK];       // the nine results of the latest check (read again only at the exit)
  __shared__ int fail_s;
#ifdef SCP_PHASE_PROFILE
  __shared__ unsigned long long pacc_s[16];
  const bool prof_t = blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
  if (prof_t) {
    for (int i = 0; i < 15; ++i) pacc_s[i] = 0;
    pacc_s[15] = wall_clock64();
  }
#endif
  const int K = A.K, N = A.N;
  const int64_t C = A.C;
  const int cap = A.ent_cap, nblk = A.nblk;
  const Lds16 L(K, cap, nblk, D, APB16);
  const int RSK = L.RSK, tK = L.tK, nks = L.nks;
  double* Rt = lds + L.rt;            // [32][RSK] r, MFMA B operand
  double* Pt = lds + L.pt;            // [32][RSK] p, overwritten in place by the lane's S0 p cell (what the row loops read)
  double* Ml = lds + L.ml;            // [tK][nks][64] packed H_f^{-1}
  double* gp = lds + L.gp;            // [nblk][2] all-gathered line-search partials
  double* ab = lds + L.ab;            // [APB][8 groups of 4] per agent, per axis: l_vel, u_vel, vf - v0, pos_min, pos_max, pf, -1e300, +1e300
  double* gck = lds + L.gchk;         // [nblk][9] all-gathered check results
  double* e_c = lds + L.ent;          // [cap][2] signed eta
  double* e_l = e_c + (size_t)cap * D;
  double* e_z = e_l + cap;
  double* e_y = e_z + cap;
  double* e_g = e_y + cap;            // row value of the next right-hand side
  double* e_qo = e_g + cap;           // [cap][2] S0 x cell of the own agent
  double* e_qp = e_qo + (size_t)cap * D;  // [cap][2] ... of the partner agent
  double* e_pp = e_qp + (size_t)cap * D;  // [cap][2] S0 p cell of the partner (this step); [e][0]: delta-y parked for the check
  int* e_code = (int*)(lds + L.n_dbl);    // [cap] k | local agent << 6 | side << 10 | partner agent << 11
  int* cptr = e_code + cap;               // [16 K + 1] cell offsets relative to this workgroup's first entry

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a0 = blockIdx.x * APB16;
  const int agent = a0 + wave;
  const bool aok = agent < N;
  const int k = lane;
  const bool live = aok && k < K;
  const bool jok = live && k < K - 1;   // the jerk row of this lane exists
  const bool lastk = k == K - 1;        // the velocity / position rows of this lane are the final-state equalities
  const double h = A.h, hh = h * h, ih = 1.0 / h, alpha = A.alpha;
  double rho = A.rho, rho_c = A.rho_c;  // (an in-kernel rho switch changes them)
  unsigned n_rho = 0;

  // ---- entries of this block of agents (contiguous in the agent-major incidence lists) --------------------------
  const int a1 = min(a0 + APB16, N);
  const int ebase = A.cell_ptr[cell_of(0, a0, K)];
  const int ne = A.cell_ptr[cell_of(0, a1, K)] - ebase;
  {  // more incident rows around some block than the tables hold: every workgroup finds out by itself and leaves at once
    int worst = 0;
    for (int b = threadIdx.x; b < (N + APB16 - 1) / APB16; b += NT16) {
      const int b0 = b * APB16, b1 = min(b0 + APB16, N);
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
  for (int i = threadIdx.x; i <= (a1 - a0) * K; i += NT16) cptr[i] = A.cell_ptr[cell_of(0, a0, K) + i] - ebase;
  for (int e = threadIdx.x; e < ne; e += NT16) {
    const int code = A.ent_code[ebase + e];
    const int n = code >> 1, side = code & 1;
    const int wi = A.w_i[n], wj = A.w_j[n], wk = A.w_k[n];
    const int own = side ? wj : wi, par = side ? wi : wj;
    e_code[e] = wk | ((own - a0) << 6) | (side << 10) | (par << 11);
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
  for (int i = threadIdx.x; i < NC16 * RSK; i += NT16) Rt[i] = 0.0;  // columns beyond the block stay zero
  for (int i = threadIdx.x; i < tK * nks * 64; i += NT16) Ml[i] = A.pMinv[i];

  // ---- column state: lane k of the agent's wave holds the rows of time step k --------------------------------------
  // row types t = 0 jerk (k < K - 1), 1 acc, 2 vel, 3 pos;  slab row of (t, k): t = 0: k, else t K - 1 + k
  double v[D][4], off[D], x[D], c1[D], c2[D];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool rok = t == 0 ? jok : live;
    const int row = t == 0 ? k : t * K - 1 + k;
    const double rr = (t >= 2 && lastk) ? rho * A.rho_eq : rho;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      v[d][t] = 0.0;
      if (rok) {
        const int64_t g = (int64_t)row * C + (int64_t)agent * D + d;
        v[d][t] = A.zf[g] + A.yf[g] / rr;
      }
    }
  }
  // The velocity / position bounds of this lane's rows are re-derived where they are needed -- bound_of() of scp_kernels.hip,
  // the same arithmetic on the same operands (scp.py:212-224, :242-257), so the same bits as the lf / uf slabs -- from 4
  // registers (the free position of the row's state, k + 1) and eight per-agent scalars in LDS, instead of being held in 16.
  if (threadIdx.x < APB16 * D) {
    const int al = threadIdx.x / D, d = threadIdx.x % D, ag = a0 + al;
    double* o = ab + al * AB_STRIDE;
    const bool there = ag < N;
    const double v0 = there ? A.states[((size_t)N + ag) * D + d] : 0.0;
    o[d] = A.vel_lo - v0;                                                        // l_vel, k < K - 1
    o[4 + d] = A.vel_hi - v0;                                                    // u_vel
    o[8 + d] = (there ? A.states[((size_t)3 * N + ag) * D + d] : 0.0) - v0;      // final velocity equality: vf - v0
    o[12 + d] = A.pmin[d];
    o[16 + d] = A.pmax[d];
    o[20 + d] = there ? A.states[((size_t)2 * N + ag) * D + d] : 0.0;            // pf
    o[24 + d] = -1e300;                                                          // rows beyond the horizon: no bound at all,
    o[28 + d] = 1e300;                                                           // so that v = 0 stays 0 there
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma clang fp contract(off)
    const double p0 = aok ? A.states[(size_t)agent * D + d] : 0.0, v0 = aok ? A.states[((size_t)N + agent) * D + d] : 0.0;
    const double hk = h * (double)(k + 1);
    const double hkv = hk * v0;
    off[d] = p0 + hkv;  // scp.py:246-247
  }
  // which table entries bound this lane's velocity / position rows: the inequalities, the final-state equalities
  // (k = K - 1: lower = upper), or nothing (beyond the horizon) -- an address per lane instead of selects per use
  const double* const tb = ab + wave * AB_STRIDE;
  const double* const b_vl = tb + (live ? (lastk ? 8 : 0) : 24);
  const double* const b_vh = tb + (live ? (lastk ? 8 : 4) : 28);
  const double* const b_pl = tb + (live ? (lastk ? 20 : 12) : 24);
  const double* const b_ph = tb + (live ? (lastk ? 20 : 16) : 28);
  auto bounds = [&](int d, double (&lo)[2], double (&hi)[2]) {
    lo[0] = b_vl[d];
    hi[0] = b_vh[d];
    lo[1] = b_pl[d] - off[d];
    hi[1] = b_ph[d] - off[d];
  };
#pragma unroll
  for (int d = 0; d < D; ++d) {
    x[d] = live ? A.x[(int64_t)k * C + (int64_t)agent * D + d] : 0.0;
    const double s1 = wave_incl_sum(x[d]);            // exact prefix sums of x (refreshed at every check)
    const double s2 = lane_below(wave_incl_sum(s1));
    c1[d] = live ? s1 : 0.0;                           // lanes beyond the horizon would hold the running totals: zero, so
    c2[d] = live ? s2 : 0.0;                           // that their (non-existent) rows contribute nothing to the suffix sums
  }
  const double jlo = A.jerk_lo, jhi = A.jerk_hi, alo = A.acc_lo, ahi = A.acc_hi;
  __syncthreads();
  // incidence-list range of this lane's cell (time step k of this wave's agent); cells without entries publish nothing
  const int c0 = aok ? cptr[wave * K + min(k, K - 1)] : 0;
  const int c1e = (aok && k < K) ? cptr[wave * K + k + 1] : c0;
  const bool has_rows = c1e > c0;
  u64* my_cell = A.cells + ((size_t)((int64_t)min(k, K - 1) * N + (aok ? agent : 0)) * D) * 2;
  double* my_pt = Pt + (size_t)(wave * D) * RSK + k;   // + d RSK: this lane's slot of column (agent, d)
  double* my_rt = Rt + (size_t)(wave * D) * RSK + k;

  PSTAMP(0);
  bool ok = true;
  int it_done = A.it0;      // ADMM iterations of this solve completed so far
  unsigned steps = 0;       // steps run by this launch
  unsigned exit_code = 0;
  const bool with_dy = A.eps_prim_inf > 0.0;
  if (threadIdx.x < 3 * APB16) (&cert_s[0][0])[threadIdx.x] = 0.0;
  int cad = A.cad0;  // steps between two checks: check_every, or check_fine once the residuals are close / rho has changed
  for (;;) {  // one batch of steps up to the next termination check, then the check and the decision to go on
  int nit = cad - it_done % cad;
  if (it_done + nit > A.max_iter) nit = A.max_iter - it_done;
  for (int it = 0; it < nit; ++it, ++steps) {
    const unsigned tag = A.epoch0 + steps + 1u;
    const bool last = it == nit - 1;
    // The one-double-per-row state presumes z = Pi(v).  Right after scp_qp_reset z = A x0 is NOT projected (and y = 0, so
    // v = z): the first step of a QP therefore takes z = v instead of Pi(v) -- W' = rho (v - F x), v' = alpha F x~ +
    // (1 - alpha) v, which is the standard update from (z, y) = (v, 0) -- and establishes the invariant for all later steps.
    const bool first = A.first_step && steps == 0u;
    u64* gpart = A.gpart + (size_t)(tag & 1u) * nblk * 4;
    const double rv = lastk ? rho * A.rho_eq : rho;   // rho of this lane's velocity / position rows
    // ---- r = -2 x + F^T W' + S0^T G: reverse cumulative sums as suffix scans over the lanes ------------------------
    {
      double r[D];
      double g[D];
#pragma unroll
      for (int d = 0; d < D; ++d) g[d] = 0.0;
      for (int e = c0; e < c1e; ++e) {
        const double ge = e_g[e];
#pragma unroll
        for (int d = 0; d < D; ++d) g[d] += e_c[(size_t)e * D + d] * ge;
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        // W' = rho (z - F x) - y with z = Pi(v), y = rho (v - Pi(v))
        double lo[2], hi[2];
        bounds(d, lo, hi);
        const double xn = lane_above(x[d]);
        double cj = fmin(fmax(v[d][0], jlo), jhi), ca = fmin(fmax(v[d][1], alo), ahi);
        double cv = fmin(fmax(v[d][2], lo[0]), hi[0]), cp = fmin(fmax(v[d][3], lo[1]), hi[1]);
        if (first) { cj = v[d][0]; ca = v[d][1]; cv = v[d][2]; cp = v[d][3]; }  // z of a reset is A x0 itself (y = 0, v = z)
        const double wj = jok ? rho * ((cj - (xn - x[d]) * ih) - (v[d][0] - cj)) : 0.0;
        const double wa = rho * ((ca - x[d]) - (v[d][1] - ca));
        const double wv = rv * ((cv - h * c1[d]) - (v[d][2] - cv));
        const double wp = rv * ((cp - hh * (c2[d] + 0.5 * c1[d])) - (v[d][3] - cp));
        const double u1 = h * wv + 0.5 * hh * (wp - g[d]);
        const double u2 = wp + g[d];
        const double d1 = wave_incl_rsum(u1);
        const double s1 = wave_incl_rsum(u2);
        const double d2 = lane_above(wave_incl_rsum(s1));  // exclusive suffix sum
        const double wjm = lane_below(wj);                  // w_j[k - 1]
        r[d] = (((wjm - wj) * ih + wa) + (d1 + 0.5 * hh * g[d]) + hh * d2) - 2.0 * x[d];
        if (live) my_rt[d * RSK] = r[d];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    PSTAMP(1);
    // ---- p = H_f^{-1} r on the matrix cores: eight waves = 4 row tiles x 2 column tiles, operands streamed from LDS ----
    if (wave < 4 * NCT && (wave & 3) < tK) {
      const int li = lane & 15, lk = lane >> 4;
      const int tile = wave & 3, ct = wave >> 2;
      const double* Mt = Ml + (size_t)tile * nks * 64 + lane;
      const double* Bt = Rt + (size_t)(ct * 16 + li) * RSK;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int s = 0; s < nks; ++s) {
        const int kk = 4 * s + lk;
        const double b = kk < K ? Bt[kk] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[(size_t)s * 64], b, acc, 0, 0, 0);
      }
      double* Ot = Pt + (size_t)(ct * 16 + li) * RSK;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = tile * 16 + lk + 4 * q;
        if (row < K) Ot[row] = acc[q];
      }
    }
    __syncthreads();
    PSTAMP(2);
    // ---- prefix sums of p, S0 p (published where the cell has rows), r.p ------------------------------------------------
    double p[D], s1p[D], s2p[D];
    {
      double rz = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        p[d] = live ? my_pt[d * RSK] : 0.0;
        rz += (live ? my_rt[d * RSK] : 0.0) * p[d];  // r of this lane, re-read from its tile
        const double cs1 = wave_incl_sum(p[d]);
        const double cs2 = lane_below(wave_incl_sum(cs1));
        const double qp = hh * (cs2 - 0.5 * lane_below(cs1));
        s1p[d] = live ? cs1 : 0.0;
        s2p[d] = live ? cs2 : 0.0;
        if (has_rows) {
          st_granules(my_cell + 2 * d, tag, qp);
          my_pt[d * RSK] = qp;  // the row loops read the own agent's S0 p cell here
        }
      }
      rz = wave_incl_sum(rz);
      if (lane == 63) red[0][wave] = rz;
    }
    __syncthreads();
    PSTAMP(3);
    // ---- working rows: partner cells (polled until they carry this step's tag), eta . d(S0 p) ---------------------------
    {
      double sq = 0.0;
      unsigned spins = 0;
      bool bad = false;
      for (int e = threadIdx.x; e < ne; e += NT16) {
        const int code = e_code[e];
        const int ek = code & 63, al = (code >> 6) & 15, side = (code >> 10) & 1, par = code >> 11;
        const u64* pc = A.cells + ((size_t)((int64_t)ek * N + par) * D) * 2;
        u32x4 w[D];
        for (;;) {
          ld_cell<D>(pc, w);
          bool here = true;
#pragma unroll
          for (int d = 0; d < D; ++d) here = here && pair_ok(w[d], tag);
          if (here) break;
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
          s += e_c[(size_t)e * D + d] * (Pt[(size_t)(al * D + d) * RSK + ek] - pp);
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
    // (A two-level form -- group leaders sum 16 partials, everybody polls 16 group sums: a sixteenth of the polls -- was
    // measured at 4096 x 50 and is SLOWER, 4.8 against 4.0 us: what this phase waits for is the slowest workgroup of the
    // step, not the fabric.)
    if (threadIdx.x < 2) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < APB16; ++w) t += red[threadIdx.x][w];
      st_granules(gpart + (size_t)blockIdx.x * 4 + 2 * threadIdx.x, tag, t);
    }
    {
      unsigned spins = 0;
      bool bad = false;
      for (int q = threadIdx.x; q < 2 * nblk; q += NT16) {  // one double (two granules) per thread and pass
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
    // ---- collision rows first (the certificate of the last step needs their delta-y before the lanes' chain) ----------------
    {
      const double irc = 1.0 / rho_c;
      for (int e = threadIdx.x; e < ne; e += NT16) {
        const int code = e_code[e];
        const int ek = code & 63, al = (code >> 6) & 15;
        double tc = 0.0, ax = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double c = e_c[(size_t)e * D + d];
          const double po = Pt[(size_t)(al * D + d) * RSK + ek], pp = e_pp[(size_t)e * D + d];
          const double qo = e_qo[(size_t)e * D + d], qq = e_qp[(size_t)e * D + d];
          tc += c * (fma(a, po, qo) - fma(a, pp, qq));
          const double qon = fma(aa, po, qo), qqn = fma(aa, pp, qq);
          ax += c * (qon - qqn);
          e_qo[(size_t)e * D + d] = qon;
          e_qp[(size_t)e * D + d] = qqn;
        }
        const double zo = e_z[e], yo = e_y[e];
        const double zh = alpha * tc + (1.0 - alpha) * zo;
        const double zn = fmax(zh + yo * irc, e_l[e]);
        const double yn = yo + rho_c * (zh - zn);
        e_z[e] = zn;
        e_y[e] = yn;
        e_g[e] = (rho_c * zn - yn) - rho_c * ax;
        if (last && with_dy) e_pp[(size_t)e * D] = fmin(yn - yo, 0.0);  // delta-y of the batch's last step (u = +inf: polar of
                                                                        // the recession cone), parked until the check
      }
    }
    const bool cert = last && with_dy;
    if (cert) __syncthreads();  // (uniform: every lane of the grid takes it) the rows' delta-y are parked
    double m_ndy = 0.0, m_supp = 0.0, m_natdy = 0.0;  // delta-y of this step, reduced on the spot (certificate)
    // ---- fixed rows: z~ = F (x + a p) from the combined prefix sums, relaxation, projection, duals -----------------------
    {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const double xt = fma(a, p[d], x[d]);
        const double t1 = fma(a, s1p[d], c1[d]), t2 = fma(a, s2p[d], c2[d]);
        const double xtn = lane_above(xt);
        double lo[2], hi[2];
        bounds(d, lo, hi);
        // v' = v + alpha (F x~ - Pi(v)); the new z, y are Pi(v'), rho (v' - Pi(v'))
        double cj = fmin(fmax(v[d][0], jlo), jhi), ca = fmin(fmax(v[d][1], alo), ahi);
        double cv = fmin(fmax(v[d][2], lo[0]), hi[0]), cp = fmin(fmax(v[d][3], lo[1]), hi[1]);
        if (first) { cj = v[d][0]; ca = v[d][1]; cv = v[d][2]; cp = v[d][3]; }
        const double nj = jok ? fma(alpha, (xtn - xt) * ih - cj, v[d][0]) : v[d][0];
        const double na = fma(alpha, xt - ca, v[d][1]);
        const double nv = fma(alpha, h * t1 - cv, v[d][2]);
        const double np = fma(alpha, hh * (t2 + 0.5 * t1) - cp, v[d][3]);
        if (cert) {
          // OSQP's certificate on delta-y = rho [(v' - Pi(v')) - (v - Pi(v))] of this step: |dy|, the support value
          // u.dy+ + l.dy-, and A^T dy by the r chain
          const double dyj = rho * ((nj - fmin(fmax(nj, jlo), jhi)) - (v[d][0] - cj));
          const double dya = rho * ((na - fmin(fmax(na, alo), ahi)) - (v[d][1] - ca));
          const double dyv = rv * ((nv - fmin(fmax(nv, lo[0]), hi[0])) - (v[d][2] - cv));
          const double dyp = rv * ((np - fmin(fmax(np, lo[1]), hi[1])) - (v[d][3] - cp));
          m_ndy = fmax(m_ndy, fmax(fmax(fabs(dyj), fabs(dya)), fmax(fabs(dyv), fabs(dyp))));
          m_supp += (jhi * fmax(dyj, 0.0) + jlo * fmin(dyj, 0.0)) + (ahi * fmax(dya, 0.0) + alo * fmin(dya, 0.0)) +
                    (hi[0] * fmax(dyv, 0.0) + lo[0] * fmin(dyv, 0.0)) + (hi[1] * fmax(dyp, 0.0) + lo[1] * fmin(dyp, 0.0));
          double gd = 0.0;
          for (int e = c0; e < c1e; ++e) gd += e_c[(size_t)e * D + d] * e_pp[(size_t)e * D];
          const double u1 = h * dyv + 0.5 * hh * (dyp - gd);
          const double u2 = dyp + gd;
          const double d1 = wave_incl_rsum(u1);
          const double s1 = wave_incl_rsum(u2);
          const double d2 = lane_above(wave_incl_rsum(s1));
          const double vjm = lane_below(dyj);
          const double at = ((vjm - dyj) * ih + dya) + (d1 + 0.5 * hh * gd) + hh * d2;
          if (live) m_natdy = fmax(m_natdy, fabs(at));
        }
        v[d][0] = nj; v[d][1] = na; v[d][2] = nv; v[d][3] = np;
        x[d] = fma(aa, p[d], x[d]);
        c1[d] = fma(aa, s1p[d], c1[d]);
        c2[d] = fma(aa, s2p[d], c2[d]);
        __builtin_amdgcn_sched_barrier(0);  // one column at a time: interleaving both doubles the temporaries
      }
    }
    if (cert) {
      m_ndy = wave_max_nn(m_ndy);
      m_supp = wave_incl_sum(m_supp);
      m_natdy = wave_max_nn(m_natdy);
      if (lane == 63) { cert_s[0][wave] = m_ndy; cert_s[1][wave] = m_supp; cert_s[2][wave] = m_natdy; }
    }
    __syncthreads();
    PSTAMP(7);
  }

  if (!ok) break;
  it_done += nit;
  // ==== termination check (same quantities as the 8-agent kernel's; x's prefix sums are rebuilt exactly) =========================
  {
    const unsigned ctag = 0x80000000u | (A.epoch0 + steps);
    const double rv = lastk ? rho * A.rho_eq : rho;
    double m[NCHK];
#pragma unroll
    for (int j = 0; j < NCHK; ++j) m[j] = 0.0;
    {
      double gy[D];
#pragma unroll
      for (int d = 0; d < D; ++d) gy[d] = 0.0;
      for (int e = c0; e < c1e; ++e) {
        const double ye = e_y[e];
#pragma unroll
        for (int d = 0; d < D; ++d) gy[d] += e_c[(size_t)e * D + d] * ye;
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const double s1 = wave_incl_sum(x[d]);
        const double s2 = lane_below(wave_incl_sum(s1));
        const double qx = hh * (s2 - 0.5 * lane_below(s1));
        c1[d] = live ? s1 : 0.0;
        c2[d] = live ? s2 : 0.0;
        if (has_rows) {
          st_granules(my_cell + 2 * d, ctag, qx);
          my_pt[d * RSK] = qx;
        }
        double lo[2], hi[2];
        bounds(d, lo, hi);
        const double xn = lane_above(x[d]);
        const double f[4] = {jok ? (xn - x[d]) * ih : 0.0, x[d], h * c1[d], hh * (c2[d] + 0.5 * c1[d])};
        const double zc[4] = {fmin(fmax(v[d][0], jlo), jhi), fmin(fmax(v[d][1], alo), ahi),
                              fmin(fmax(v[d][2], lo[0]), hi[0]), fmin(fmax(v[d][3], lo[1]), hi[1])};  // z = Pi(v)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t == 0 ? jok : live) {
            m[CK_RP] = fmax(m[CK_RP], fabs(f[t] - zc[t]));
            m[CK_NAX] = fmax(m[CK_NAX], fabs(f[t]));
            m[CK_NZ] = fmax(m[CK_NZ], fabs(zc[t]));
          }
        }
        // A^T y: the r chain with W' -> y = rho (v - Pi(v))
        const double vj = jok ? rho * (v[d][0] - zc[0]) : 0.0, va = rho * (v[d][1] - zc[1]);
        const double vv = rv * (v[d][2] - zc[2]), vp = rv * (v[d][3] - zc[3]);
        const double g = gy[d];
        const double u1 = h * vv + 0.5 * hh * (vp - g);
        const double u2 = vp + g;
        const double d1 = wave_incl_rsum(u1);
        const double r1 = wave_incl_rsum(u2);
        const double d2 = lane_above(wave_incl_rsum(r1));
        const double vjm = lane_below(vj);
        const double at = ((vjm - vj) * ih + va) + (d1 + 0.5 * hh * g) + hh * d2;
        if (live) {
          const double px = 2.0 * x[d];
          m[CK_RD] = fmax(m[CK_RD], fabs(px + at));
          m[CK_NPX] = fmax(m[CK_NPX], fabs(px));
          m[CK_NATY] = fmax(m[CK_NATY], fabs(at));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    {  // collision rows: exact S0 x cells of both agents (the carried copies are refreshed), residuals, delta-y
      unsigned spins = 0;
      bool bad = false;
      for (int e = threadIdx.x; e < ne; e += NT16) {
        const int code = e_code[e];
        const int ek = code & 63, al = (code >> 6) & 15, side = (code >> 10) & 1, par = code >> 11;
        const u64* pc = A.cells + ((size_t)((int64_t)ek * N + par) * D) * 2;
        u32x4 w[D];
        for (;;) {
          ld_cell<D>(pc, w);
          bool here = true;
#pragma unroll
          for (int d = 0; d < D; ++d) here = here && pair_ok(w[d], ctag);
          if (here) break;
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
          const double qo = Pt[(size_t)(al * D + d) * RSK + ek];
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
      double v = j == CK_SUPP ? wave_incl_sum(m[j]) : wave_max_nn(m[j]);
      if (lane == 63) {  // + the fixed rows' share of the certificate, reduced when the last step produced it
        if (j == CK_NDY) v = fmax(v, cert_s[0][wave]);
        if (j == CK_SUPP) v += cert_s[1][wave];
        if (j == CK_NATDY) v = fmax(v, cert_s[2][wave]);
        red[j][wave] = v;
      }
    }
    __syncthreads();
    if (fail_s) { ok = false; break; }
    if (threadIdx.x < NCHK) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < APB16; ++w) t = threadIdx.x == CK_SUPP ? t + red[threadIdx.x][w] : fmax(t, red[threadIdx.x][w]);
      st_granules(A.gcheck + ((size_t)blockIdx.x * NCHK + threadIdx.x) * 2, ctag, t);
    }
    {
      unsigned spins = 0;
      bool bad = false;
      for (int q = threadIdx.x; q < NCHK * nblk; q += NT16) {
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
        gck[q] = pair_value(w);
      }
      if (bad) {
        __hip_atomic_store(A.give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fail_s = 1;
      }
    }
    __syncthreads();
    if (fail_s) { ok = false; break; }
    double chk[NCHK];
#pragma unroll
    for (int j = 0; j < NCHK; ++j) {  // the same reduction order in every wave of every workgroup: identical decisions
      double v = 0.0;
      for (int b = lane; b < nblk; b += 64) v = j == CK_SUPP ? v + gck[b * NCHK + j] : fmax(v, gck[b * NCHK + j]);
      chk[j] = read_lane(j == CK_SUPP ? wave_incl_sum(v) : wave_max_nn(v), 63);
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < NCHK; ++j) chk_s[j] = chk[j];
    }
    __syncthreads();  // gck may share its LDS with the r / p tiles of the next step
    if (L.gchk == L.rt) {  // ... which must be zero in the columns beyond the block again
      for (int i = threadIdx.x; i < NC16 * RSK; i += NT16) Rt[i] = 0.0;
      // (Pt: the cells of rows are rewritten by the next step before anybody reads them)
    }
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
      // OSQP's rho estimate snapped to the 2^(1/4) grid; the candidate only SELECTS the host-computed double of the table
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
          {  // y = rho (v - Pi(v)) must survive the switch.  EXACTLY the arithmetic of leaving (write-back: z = Pi(v), y = rho_old
             // (v - Pi(v))) and re-entering with the new rho (load: v = z + y / rho_new), so that a solver object whose cache
             // already holds the new rho's blocks (switch here) and a fresh one (exit, host builds the blocks, relaunch)
             // produce the same bits -- records of pooled and one-at-a-time solves stay identical
#pragma unroll
            for (int d = 0; d < D; ++d) {
              double lo[2], hi[2];
              bounds(d, lo, hi);
              const double zc[4] = {fmin(fmax(v[d][0], jlo), jhi), fmin(fmax(v[d][1], alo), ahi),
                                    fmin(fmax(v[d][2], lo[0]), hi[0]), fmin(fmax(v[d][3], lo[1]), hi[1])};
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const double rr_old = (t >= 2 && lastk) ? rho * A.rho_eq : rho, rr_new = (t >= 2 && lastk) ? nrs * A.rho_eq : nrs;
                const double yv = rr_old * (v[d][t] - zc[t]);
                v[d][t] = zc[t] + yv / rr_new;
              }
            }
          }
          rho = nrs;
          rho_c = rho * A.rho_col_scale;
          for (int i = threadIdx.x; i < tK * nks * 64; i += NT16) Ml[i] = A.tab[slot].pMinv[i];
          for (int e = threadIdx.x; e < ne; e += NT16) {  // row values of the next right-hand side from the exact S0 x cells
            double ax = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) ax += e_c[(size_t)e * D + d] * (e_qo[(size_t)e * D + d] - e_qp[(size_t)e * D + d]);
            e_g[e] = (rho_c * e_z[e] - e_y[e]) - rho_c * ax;
          }
          ++n_rho;
          if (A.check_fine > 0) cad = A.check_fine;
        }
      }
    }
    __syncthreads();  // Rt zeroed, operands / row values of a new rho in place
    PSTAMP(9);
  }
  }  // batches

  // The exit decision is collective (see the 8-agent kernel): a workgroup that got through re-reads the give-up word
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
  // ---- write the state back (F x and S0 x from the exact prefix sums the last check left) ------------------------------------
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double xn = lane_above(x[d]);
    const double c1b = lane_below(c1[d]);
    double lo[2], hi[2];
    bounds(d, lo, hi);
    const double f[4] = {(xn - x[d]) * ih, x[d], h * c1[d], hh * (c2[d] + 0.5 * c1[d])};
    const double zc[4] = {fmin(fmax(v[d][0], jlo), jhi), fmin(fmax(v[d][1], alo), ahi),
                          fmin(fmax(v[d][2], lo[0]), hi[0]), fmin(fmax(v[d][3], lo[1]), hi[1])};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t == 0 ? jok : live) {
        const int row = t == 0 ? k : t * K - 1 + k;
        const int64_t g = (int64_t)row * C + (int64_t)agent * D + d;
        A.zf[g] = zc[t];
        A.yf[g] = ((t >= 2 && lastk) ? rho * A.rho_eq : rho) * (v[d][t] - zc[t]);
        A.fx[g] = f[t];
      }
    }
    if (live) {
      const int64_t g = (int64_t)k * C + (int64_t)agent * D + d;
      A.x[g] = x[d];
      A.Qx[g] = hh * (c2[d] - 0.5 * c1b);
    }
  }
  for (int e = threadIdx.x; e < ne; e += NT16) {
    A.gval[ebase + e] = e_g[e];
    if (!((e_code[e] >> 10) & 1)) {
      const int n = A.ent_code[ebase + e] >> 1;
      A.zc[n] = e_z[e];
      A.yc[n] = e_y[e];
    }
  }
  PSTAMP(8);
#ifdef SCP_PHASE_PROFILE
  if (prof_t)
    for (int i = 0; i < 16; ++i) scp_persist16_clk[i] = i == 15 ? (unsigned long long)steps : pacc_s[i];
#endif
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the nine check results in the slots the host reads (scp_qp::h_scal), then the exit code and the completion word
    const int slot[NCHK] = {SL_RP, SL_NAX, SL_NZ, SL_RD, SL_NPX, SL_NATY, SL_NDY, SL_SUPP, SL_NATDY};
#pragma unroll
    for (int j = 0; j < NCHK; ++j)
      __hip_atomic_store((u64*)(A.host_scal + slot[j]), (u64)__double_as_longlong(chk_s[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_status + 1, (unsigned)it_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_status + 2, n_rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((u64*)A.host_rho, (u64)__double_as_longlong(rho), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_status, exit_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.host_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

#ifdef SCP_PHASE_PROFILE
// developer hook of the profiling build only (not declared in include/scp_hip.h)
extern "C" int scp_debug_persist16_clocks(unsigned long long* out, int n) {
  if (n > 16) n = 16;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scp_persist16_clk), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

size_t scp_persist16_lds_bytes(int K, int cap, int nblk, int D, int apb) {
  const Lds16 L(K, cap, nblk, D, apb);
  const size_t ints = (size_t)cap + (size_t)apb * K + 1;
  return L.n_dbl * sizeof(double) + ((ints + 1) / 2 * 2) * sizeof(int);
}

template <int D, int APB>
static int launch16(scp_ctx* ctx, const PersistArgs& a, int nblk, size_t lds) {
  if (lds > 64 * 1024)
    SCP_HIP_CHECK(ctx, scp_raise_lds_limit(ctx->device, reinterpret_cast<const void*>(cg1_persist16_kernel<D, APB>), lds));
  hipLaunchKernelGGL((cg1_persist16_kernel<D, APB>), dim3(nblk), dim3(64 * APB), lds, ctx->stream, a);
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

int scp_persist16_launch(scp_ctx* ctx, const PersistArgs& a, int nblk, size_t lds, int D, int apb) {
  if (D == 2 && apb == 16) return launch16<2, 16>(ctx, a, nblk, lds);
  if (D == 2 && apb == 8) return launch16<2, 8>(ctx, a, nblk, lds);
  if (D == 3 && apb == 8) return launch16<3, 8>(ctx, a, nblk, lds);
  return scp_fail(ctx, SCP_ERR_INVALID, "persistent kernel: no instantiation for D = %d with %d agents per workgroup", D, apb);
}
