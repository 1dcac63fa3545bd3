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
//   * what crosses workgroups per step is (1) the S0 p cells of the partner agent and (2) the two scalars of the exact
//     line search  a = r.p / (r.p + rho_c sum (eta . d S0 p)^2) : two grid-wide exchanges, each a write-through publish,
//     a sharded arrival counter and relaxed agent-scope polls -- no release / acquire fences (MI355X_MICROARCH.md,
//     inter-workgroup visibility, "valid forms" row 1; measured 1.5 us for the barrier alone, 2.5 us with an all-reduce,
//     tools/grid_sync_bench.hip) against 3 kernel boundaries + 3 cold prologues (21.6 us per step in round 1).
// Every spin is bounded: a workgroup that times out raises a give-up word, every workgroup then leaves WITHOUT writing
// state back, and the host repeats the iterations on the three-launch pipeline.
#include "scp_qp_device.h"

namespace {
using namespace scpdev;

typedef unsigned long long u64;
constexpr int NSHARD = 8;             // arrival counters (workgroup b adds to shard b % 8), one 128-byte line each
constexpr int SHARD_STRIDE = 16;      // u64 per shard
constexpr unsigned SPIN_LIMIT = 1u << 21;

struct PersistArgs {
  int K, N, nblk, nit, emit_dy, ent_cap;
  int64_t C;
  double rho, rho_c, rho_eq, alpha, h;
  const double* pMinv;
  const double *lf, *uf;
  double *zf, *yf, *fx, *x, *Qx, *dyf;
  double* Qp_pub;   // [K][C] exchange slab: S0 p of every column
  double* part;     // [2 parities][rz | sq][nblk * waves] per-wave partials of the line search
  u64* shards;
  unsigned* give_up;
  const int *cell_ptr, *ent_code, *w_k, *w_i, *w_j;
  const double *w_eta, *w_l;
  double *zc, *yc, *dyc, *gval;
  unsigned* host_status;  // mapped host word: 1 = all steps done, 2 = gave up
  u64 epoch0;             // arrivals per workgroup before this launch (the counters are never reset between launches)
};

__device__ inline void st_agent(double* p, double v) {
  __hip_atomic_store((u64*)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double ld_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Grid-wide rendezvous after a write-through publish: every storing wave drains its stores, one lane adds to the
// workgroup's shard, wave 0 polls all shards (relaxed, agent scope).  Returns false when the spin limit was hit or
// another workgroup gave up.
__device__ inline bool grid_rendezvous(u64* shards, int nblk, u64 epoch, unsigned* give_up, int* ok_s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(shards + (blockIdx.x % NSHARD) * SHARD_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *ok_s = 1;
  }
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const u64 per = lane < NSHARD ? (u64)((nblk - lane + NSHARD - 1) / NSHARD) : 0ull;
    const u64 target = per * epoch;
    unsigned spins = 0;
    bool ok = true;
    for (;;) {
      u64 v = target;
      if (lane < NSHARD) v = __hip_atomic_load(shards + lane * SHARD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(v >= target)) break;
      if (++spins > SPIN_LIMIT || __hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok && lane == 0) {
      __hip_atomic_store(give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ok_s = 0;
    }
  }
  __syncthreads();
  return *ok_s != 0;
}

template <int D>
__global__ __launch_bounds__(64 * (CB / D)) void cg1_persist_kernel(PersistArgs A) {
  constexpr int APB = CB / D;    // agents = waves per workgroup
  constexpr int NT = 64 * APB;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red_rz[APB], red_sq[APB];
  __shared__ int ok_s;
  const int K = A.K, N = A.N;
  const int64_t C = A.C;
  const int RSK = pad_col(K);
  const int cap = A.ent_cap;
  double* Rt = lds;                        // [16][RSK]   r, MFMA B operand
  double* Pt = Rt + CB * RSK;              // [16][RSK]   p
  double* Qt = Pt + CB * RSK;              // [APB][64][D] own S0 p cells
  double* e_c = Qt + APB * 64 * D;         // [cap][D] signed eta (+ on agent i's side, - on agent j's)
  double* e_l = e_c + (size_t)cap * D;     // [cap] lower bound
  double* e_z = e_l + cap;                 // [cap]
  double* e_y = e_z + cap;                 // [cap]
  double* e_g = e_y + cap;                 // [cap] row value of the next right-hand side
  double* e_qo = e_g + cap;                // [cap][D] S0 x cell of the own agent
  double* e_qp = e_qo + (size_t)cap * D;   // [cap][D] S0 x cell of the partner agent
  double* e_pp = e_qp + (size_t)cap * D;   // [cap][D] S0 p cell of the partner agent (this step)
  int* e_code = (int*)(e_pp + (size_t)cap * D);  // [cap] k | local agent << 8 | side << 16
  int* e_pad = e_code + cap;               // [cap] index of the partner's cell in the [K][C] slabs
  int* e_row = e_pad + cap;                // [cap] working row n
  int* cptr = e_row + cap;                 // [APB K + 1] cell offsets relative to this workgroup's first entry

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a0 = blockIdx.x * APB;
  const int agent = a0 + wave;
  const bool aok = agent < N;
  const int k = lane;
  const bool live = aok && k < K;
  const double h = A.h, hh = h * h, rho = A.rho, rho_c = A.rho_c, alpha = A.alpha;
  const int tK = (K + 15) >> 4, nks = (K + 3) >> 2;
  const int nparts = A.nblk * APB;

  // ---- entries of this block of agents (contiguous in the agent-major incidence lists) --------------------------
  const int a1 = min(a0 + APB, N);
  const int ebase = A.cell_ptr[cell_of(0, a0, K)];
  const int ne = A.cell_ptr[cell_of(0, a1, K)] - ebase;
  if (ne > cap) {  // cannot happen when the host sized the launch from the entry counts; never spin on it
    if (threadIdx.x == 0) {
      __hip_atomic_store(A.give_up, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A.host_status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  for (int i = threadIdx.x; i <= (a1 - a0) * K; i += NT) cptr[i] = A.cell_ptr[cell_of(0, a0, K) + i] - ebase;
  for (int e = threadIdx.x; e < ne; e += NT) {
    const int code = A.ent_code[ebase + e];
    const int n = code >> 1, side = code & 1;
    const int wi = A.w_i[n], wj = A.w_j[n], wk = A.w_k[n];
    const int own = side ? wj : wi, par = side ? wi : wj;
    e_code[e] = wk | ((own - a0) << 8) | (side << 16);
    e_pad[e] = (int)((int64_t)wk * C + (int64_t)par * D);
    e_row[e] = n;
    const int64_t bo = (int64_t)wk * C + (int64_t)own * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double eta = A.w_eta[(size_t)n * D + d];
      e_c[(size_t)e * D + d] = side ? -eta : eta;
      e_qo[(size_t)e * D + d] = A.Qx[bo + d];
      e_qp[(size_t)e * D + d] = A.Qx[e_pad[e] + d];
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
  double aM[CHB];
  tile_prefetch<CHB>(A.pMinv, nks, wave < tK ? wave : 0, 0, nks, aM);
  __syncthreads();

  u64 epoch = A.epoch0;
  bool ok = true;
  double dy[D][4];
  for (int it = 0; it < A.nit; ++it) {
    const int par = it & 1;
    double* part_rz = A.part + (size_t)(2 * par) * nparts;
    double* part_sq = A.part + (size_t)(2 * par + 1) * nparts;
    // ---- r = -2 x + F^T W' + S0^T G: reverse cumulative sums as suffix scans over the lanes ------------------------
    double r[D];
    {
      const int c0 = aok ? cptr[wave * K + min(k, K - 1)] : 0;
      const int c1 = (aok && k < K) ? cptr[wave * K + k + 1] : c0;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        double g = 0.0;
        for (int e = c0; e < c1; ++e) g += e_c[(size_t)e * D + d] * e_g[e];
        const double wj = rr[0] * (z[d][0] - fx[d][0]) - y[d][0];
        const double wa = rr[1] * (z[d][1] - fx[d][1]) - y[d][1];
        const double wv = rr[2] * (z[d][2] - fx[d][2]) - y[d][2];
        const double wp = rr[3] * (z[d][3] - fx[d][3]) - y[d][3];
        const double u1 = h * wv + 0.5 * hh * (wp - g);
        const double u2 = wp + g;
        const double d1 = wave_incl_rsum(u1);
        const double s1 = wave_incl_rsum(u2);
        const double d2 = lane_above(wave_incl_rsum(s1));  // exclusive suffix sum
        const double wjm = lane_below(wj);                  // w_j[k - 1]
        r[d] = (((wjm - wj) / h + wa) + (d1 + 0.5 * hh * g) + hh * d2) - 2.0 * x[d];
        if (live) Rt[(wave * D + d) * RSK + k] = r[d];
      }
    }
    __syncthreads();
    // ---- p = H_f^{-1} r on the matrix cores: one 16-row tile per wave --------------------------------------------
    if (wave < tK) {
      const int li = lane & 15, lk = lane >> 4;
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
        const int row = wave * 16 + lk + 4 * q;
        if (row < K) Pt[li * RSK + row] = acc[q];
      }
    }
    __syncthreads();
    // ---- S0 p, F p (forward scans), r.p; publish the S0 p cells --------------------------------------------------------
    double p[D], qp[D], fp[D][4];
    double rz = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      p[d] = live ? Pt[(wave * D + d) * RSK + k] : 0.0;
      rz += (live ? r[d] : 0.0) * p[d];
      const double c1 = wave_incl_sum(p[d]);
      const double c2 = lane_below(wave_incl_sum(c1));
      const double c1p = lane_below(c1);
      const double pn = lane_above(p[d]);
      qp[d] = hh * (c2 - 0.5 * c1p);
      fp[d][0] = (live && k < K - 1) ? (pn - p[d]) / h : 0.0;
      fp[d][1] = p[d];
      fp[d][2] = live ? h * c1 : 0.0;  // lanes beyond the horizon hold the running totals: keep their rows at zero,
      fp[d][3] = live ? hh * (c2 + 0.5 * c1) : 0.0;  // or the next step's suffix sums would pick them up
      if (live) {
        Qt[(wave * 64 + k) * D + d] = qp[d];
        st_agent(A.Qp_pub + (int64_t)k * C + (int64_t)agent * D + d, qp[d]);
      }
    }
    rz = wave_incl_sum(rz);
    if (lane == 63) st_agent(part_rz + blockIdx.x * APB + wave, rz);
    if (!grid_rendezvous(A.shards, A.nblk, ++epoch, A.give_up, &ok_s)) { ok = false; break; }
    // ---- working rows: eta . d(S0 p), partials of p.H p; total of r.p -----------------------------------------------------
    {
      double sq = 0.0;
      for (int e = threadIdx.x; e < ne; e += NT) {
        const int code = e_code[e];
        const int ek = code & 0xFF, al = (code >> 8) & 0xFF, side = (code >> 16) & 1;
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double pp = ld_agent(A.Qp_pub + e_pad[e] + d);
          e_pp[(size_t)e * D + d] = pp;
          s += e_c[(size_t)e * D + d] * (Qt[(al * 64 + ek) * D + d] - pp);
        }
        if (!side) sq += s * s;  // every row once
      }
      double v = 0.0;
      for (int b = threadIdx.x; b < nparts; b += NT) v += ld_agent(part_rz + b);
      sq = wave_incl_sum(sq);
      v = wave_incl_sum(v);
      if (lane == 63) {
        st_agent(part_sq + blockIdx.x * APB + wave, sq);
        red_rz[wave] = v;
      }
    }
    if (!grid_rendezvous(A.shards, A.nblk, ++epoch, A.give_up, &ok_s)) { ok = false; break; }
    double a;
    {
      double v = 0.0;
      for (int b = threadIdx.x; b < nparts; b += NT) v += ld_agent(part_sq + b);
      v = wave_incl_sum(v);
      if (lane == 63) red_sq[wave] = v;
      __syncthreads();
      double rzt = 0.0, sqt = 0.0;
#pragma unroll
      for (int w = 0; w < APB; ++w) {
        rzt += red_rz[w];
        sqt += red_sq[w];
      }
      const double pHp = rzt + rho_c * sqt;
      a = (pHp > 0.0 && rzt != 0.0) ? rzt / pHp : 0.0;
    }
    const double aa = alpha * a;
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
    const bool last = it == A.nit - 1;
    for (int e = threadIdx.x; e < ne; e += NT) {
      const int code = e_code[e];
      const int ek = code & 0xFF, al = (code >> 8) & 0xFF, side = (code >> 16) & 1;
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
      if (last && A.emit_dy && !side) A.dyc[e_row[e]] = fmin(yn - yo, 0.0);  // u = +inf: polar of the recession cone
    }
    __syncthreads();
  }
  if (!ok) {
    if (threadIdx.x == 0) __hip_atomic_store(A.host_status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
        if (A.emit_dy) A.dyf[g] = dy[d][t];
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
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(A.host_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// largest number of incidence-list entries of any block of `apb` consecutive agents -> out[0]
__global__ __launch_bounds__(256) void max_block_entries_kernel(int N, int K, int apb, const int* __restrict__ cell_ptr,
                                                                 int* __restrict__ out) {
  const int nblk = (N + apb - 1) / apb;
  int m = 0;
  for (int b = blockIdx.x * 256 + threadIdx.x; b < nblk; b += gridDim.x * 256) {
    const int a0 = b * apb, a1 = min(a0 + apb, N);
    m = max(m, cell_ptr[cell_of(0, a1, K)] - cell_ptr[cell_of(0, a0, K)]);
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

size_t persist_lds_bytes(int K, int D, int cap) {
  const int apb = CB / D;
  size_t dbl = (size_t)2 * CB * pad_col(K) + (size_t)apb * 64 * D + (size_t)cap * (4 * D + 4);
  size_t ints = (size_t)3 * cap + (size_t)apb * K + 1;
  return dbl * sizeof(double) + ((ints + 1) / 2 * 2) * sizeof(int);
}

}  // namespace

// Can the persistent kernel run this QP?  (shape limits; the entry capacity is checked per working set)
bool scp_qp_persist_eligible(const scp_qp* qp) {
  if (!qp->st.persistent || qp->st.cg_iters != 1 || qp->st.use_mfma != 1) return false;
  if (qp->K > 64 || qp->nW <= 0 || qp->persist_off) return false;
  const int apb = CB / qp->D;
  const int nblk = (qp->N + apb - 1) / apb;
  return nblk <= qp->ctx->n_cu;  // one workgroup per CU, all resident (grid-wide rendezvous)
}

// `nit` ADMM iterations in one launch.  Returns SCP_OK and *ran = 1 when the launch was enqueued (its completion status
// arrives in qp->h_persist after the stream has drained: 1 done, 2 gave up -> the caller repeats the iterations on the
// three-launch pipeline), *ran = 0 when this working set does not fit (nothing was enqueued).
int scp_qp_cg1_persist(scp_qp* qp, int nit, bool emit_dy, int* ran) {
  const QpDev& d = qp->d;
  scp_ctx* ctx = qp->ctx;
  hipStream_t s = ctx->stream;
  const int K = qp->K, D = qp->D;
  const int64_t C = qp->C, nx = (int64_t)K * C;
  const int apb = CB / D;
  const int nblk = (qp->N + apb - 1) / apb;
  *ran = 0;
  if (!qp->cg1_ready) {
    int rc = scp_qp_cg1_prepare(qp);
    if (rc) return rc;
  }
  if (qp->persist_cap_nW != qp->nW) {  // working set changed: size the entry tables (one read-back per change)
    SCP_HIP_CHECK(ctx, hipMemsetAsync(d.sync_words + 2 * NSHARD * SHARD_STRIDE, 0, 16, s));
    hipLaunchKernelGGL(max_block_entries_kernel, dim3(4), dim3(256), 0, s, qp->N, K, apb, d.cell_ptr, (int*)(d.sync_words + 2 * NSHARD * SHARD_STRIDE));
    SCP_HIP_CHECK(ctx, hipGetLastError());
    int m = 0;
    SCP_HIP_CHECK(ctx, hipMemcpyAsync(&m, d.sync_words + 2 * NSHARD * SHARD_STRIDE, sizeof(int), hipMemcpyDeviceToHost, s));
    SCP_HIP_CHECK(ctx, hipStreamSynchronize(s));
    qp->persist_cap_nW = qp->nW;
    qp->persist_cap = (m + 63) / 64 * 64;
  }
  const size_t lds = persist_lds_bytes(K, D, qp->persist_cap);
  if (lds > 160 * 1024) return SCP_OK;  // too many rows around one block of agents: three-launch pipeline
  PersistArgs a;
  a.K = K; a.N = qp->N; a.nblk = nblk; a.nit = nit; a.emit_dy = emit_dy ? 1 : 0; a.ent_cap = qp->persist_cap;
  a.C = C;
  a.rho = qp->rho; a.rho_c = qp->rho * qp->st.rho_col_scale; a.rho_eq = qp->st.rho_eq_scale; a.alpha = qp->st.alpha; a.h = qp->h;
  a.pMinv = d.pMinv;
  a.lf = d.lf; a.uf = d.uf; a.zf = d.zf; a.yf = d.yf; a.fx = d.fx; a.x = d.x;
  a.Qx = qp->qx_sel ? d.HQ : d.HQ + nx;
  a.dyf = d.dyf;
  a.Qp_pub = d.hpf;
  a.part = d.part;
  a.shards = (u64*)d.sync_words;
  a.give_up = (unsigned*)(d.sync_words + NSHARD * SHARD_STRIDE);
  a.cell_ptr = d.cell_ptr; a.ent_code = d.ent_code; a.w_k = d.w_k; a.w_i = d.w_i; a.w_j = d.w_j;
  a.w_eta = d.w_eta; a.w_l = d.w_l; a.zc = d.zc; a.yc = d.yc; a.dyc = d.dyc; a.gval = d.gval;
  a.host_status = qp->h_persist_dev;
  *qp->h_persist = 0u;
  if (qp->persist_epoch == 0)  // first launch, or the one after a give-up: every polled word starts from zero
    SCP_HIP_CHECK(ctx, hipMemsetAsync(d.sync_words, 0, (size_t)(NSHARD * SHARD_STRIDE + 2) * sizeof(u64), s));
  a.epoch0 = qp->persist_epoch;
  qp->persist_epoch += 2ull * (u64)nit;  // two rendezvous per ADMM step
  if (D == 2) {
    if (lds > 64 * 1024)
      SCP_HIP_CHECK(ctx, scp_raise_lds_limit(ctx->device, reinterpret_cast<const void*>(cg1_persist_kernel<2>), lds));
    hipLaunchKernelGGL(cg1_persist_kernel<2>, dim3(nblk), dim3(64 * (CB / 2)), lds, s, a);
  } else {
    if (lds > 64 * 1024)
      SCP_HIP_CHECK(ctx, scp_raise_lds_limit(ctx->device, reinterpret_cast<const void*>(cg1_persist_kernel<3>), lds));
    hipLaunchKernelGGL(cg1_persist_kernel<3>, dim3(nblk), dim3(64 * (CB / 3)), lds, s, a);
  }
  SCP_HIP_CHECK(ctx, hipGetLastError());
  *ran = 1;
  return SCP_OK;
}
