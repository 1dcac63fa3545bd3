// Joint QP of the SCP iteration on gfx950 (rows a3 / a6 / a9 of SURVEY.md section 8).
//
//   min ||x||^2   s.t.   l_f <= F x <= u_f  (jerk, acc, vel, pos rows; scp.py:182-257, :332-358)
//                        A_W x >= l_W       (working set of collision rows; scp.py:453-557)
//
// replaces osqp.OSQP().setup/warm_start/solve (scp.py:326-367, :441-449).  The algorithm is OSQP's ADMM
// (rho / sigma / alpha, rho x 1e3 on equality rows, adaptive rho, termination test every 25 iterations on the
// unscaled inf-norm residuals) with an indirect x-update, as OSQP's own GPU backend does:
//
//   ((2 + sigma) I + F^T R_f F + A_W^T R_c A_W) x~ = sigma x + F^T (R_f z_f - y_f) + A_W^T (R_c z_c - y_c)
//
// is solved by PCG.  The fixed part H_f = (2+sigma) I + F^T R_f F is the same K x K block for EVERY
// (agent, axis) column, so its exact inverse (one small dense factorisation per rho) is the
// preconditioner and is applied to all N*D columns at once as a [K x K] x [K x N*D] product on the fp64 MFMA
// units; A_W is never formed: row (k, i, j) is  eta . ((S0 x_i)[k] - (S0 x_j)[k]).
//
// Device layout: every vector lives time-major, [rows][C] with C = N*D columns (c = i*D + d).  Fixed rows are
// stacked as  [0,K-1) jerk | [K-1,2K-1) acc | [2K-1,3K-1) vel | [3K-1,4K-1) pos.
// oracle/qp_oracle.py:admm_structured is the line-by-line CPU statement of this file.
#include "scp_qp_internal.h"
#include "scp_reset_device.h"

#include <chrono>
#include <cmath>
#include <vector>

// ----------------------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------------------
__device__ inline double sum_partials(const double* part) {
  double s = 0.0;
  for (int b = 0; b < NPART; ++b) s += part[b];
  return s;
}

// part[b] = sum over this block's grid-stride share of a.b (fixed tree, deterministic)
__global__ __launch_bounds__(256) void dot_partial_kernel(int64_t n, const double* __restrict__ a,
                                                           const double* __restrict__ b, double* __restrict__ part) {
  __shared__ double s[4];
  double acc = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)NPART * 256) acc += a[t] * b[t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// wf = rho * w[row] * zf - yf   (Rf x C);  rhs = sigma * x  (K x C, first K*C threads)
__global__ __launch_bounds__(256) void admm_rhs_prep_kernel(int64_t nf, int64_t nx, int64_t C, double rho, double sigma,
                                                             const double* __restrict__ wrow,
                                                             const double* __restrict__ zf,
                                                             const double* __restrict__ yf, double* __restrict__ wf,
                                                             const double* __restrict__ x, double* __restrict__ rhs) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < nf) wf[t] = rho * wrow[t / C] * zf[t] - yf[t];
  if (t < nx) rhs[t] = sigma * x[t];
}

enum RowMode { ROW_RHS = 0, ROW_HMUL = 1, ROW_Y = 2, ROW_VEC = 3 };

// r = rhs - Hx
__global__ __launch_bounds__(256) void cg_residual_kernel(int64_t n, const double* __restrict__ rhs,
                                                           const double* __restrict__ Hx, double* __restrict__ r) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) r[t] = rhs[t] - Hx[t];
}

// p = zz ; scal[SL_RZ0] = sum(part)
__global__ __launch_bounds__(256) void cg_start_kernel(int64_t n, const double* __restrict__ zz, double* __restrict__ p,
                                                        const double* __restrict__ part, double* __restrict__ scal) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) p[t] = zz[t];
  if (t == 0) scal[SL_RZ0] = sum_partials(part);
}

// alpha = rz / pHp ; xt += alpha p ; r -= alpha Hp
__global__ __launch_bounds__(256) void cg_update_kernel(int64_t n, int slot, const double* __restrict__ scal,
                                                         const double* __restrict__ part_pHp,
                                                         const double* __restrict__ p, const double* __restrict__ Hp,
                                                         double* __restrict__ xt, double* __restrict__ r) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double rz = scal[slot];
  const double pHp = sum_partials(part_pHp);
  const double alpha = (pHp > 0.0 && rz != 0.0) ? rz / pHp : 0.0;
  if (t < n) {
    xt[t] += alpha * p[t];
    r[t] -= alpha * Hp[t];
  }
}

// beta = rz_new / rz ; p = zz + beta p ; scal[slot^1] = rz_new
__global__ __launch_bounds__(256) void cg_direction_kernel(int64_t n, int slot, double* __restrict__ scal,
                                                            const double* __restrict__ part_rz,
                                                            const double* __restrict__ zz, double* __restrict__ p) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double rz = scal[slot];
  const double rz_new = sum_partials(part_rz);
  const double beta = rz != 0.0 ? rz_new / rz : 0.0;
  if (t < n) p[t] = zz[t] + beta * p[t];
  if (t == 0) scal[slot ^ 1] = rz_new;
}

// fixed rows: relaxation, projection, dual update (OSQP steps 4-6);  x = alpha xt + (1-alpha) x
__global__ __launch_bounds__(256) void admm_fixed_update_kernel(int64_t nf, int64_t nx, int64_t C, double rho,
                                                                 double alpha, const double* __restrict__ wrow,
                                                                 const double* __restrict__ tf,
                                                                 const double* __restrict__ lf,
                                                                 const double* __restrict__ uf, double* __restrict__ zf,
                                                                 double* __restrict__ yf, const double* __restrict__ xt,
                                                                 double* __restrict__ x) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < nf) {
    const double rr = rho * wrow[t / C];
    const double zh = alpha * tf[t] + (1.0 - alpha) * zf[t];
    const double y = yf[t];
    const double zn = fmin(fmax(zh + y / rr, lf[t]), uf[t]);
    yf[t] = y + rr * (zh - zn);
    zf[t] = zn;
  }
  if (t < nx) x[t] = alpha * xt[t] + (1.0 - alpha) * x[t];
}

// collision rows: same update with u = +inf
template <int D>
__global__ __launch_bounds__(256) void admm_row_update_kernel(int64_t nW, int64_t C, double rho, double alpha,
                                                               const int* __restrict__ wk, const int* __restrict__ wi,
                                                               const int* __restrict__ wj,
                                                               const double* __restrict__ weta,
                                                               const double* __restrict__ wl,
                                                               const double* __restrict__ Q, double* __restrict__ zc,
                                                               double* __restrict__ yc) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nW) return;
  const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
  const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
  double tc = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) tc += weta[n * D + d] * (Q[bi + d] - Q[bj + d]);
  const double zh = alpha * tc + (1.0 - alpha) * zc[n];
  const double y = yc[n];
  const double zn = fmax(zh + y / rho, wl[n]);
  yc[n] = y + rho * (zh - zn);
  zc[n] = zn;
}

__device__ inline void atomic_max_nonneg(double* addr, double v) {
  atomicMax((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
}

__device__ inline double block_max(double v) {
  __shared__ double s[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  const double m = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
  __syncthreads();
  return m;
}

// primal residual pieces over the fixed rows: max|Fx - z|, max|Fx|, max|z|
__global__ __launch_bounds__(256) void resid_fixed_kernel(int64_t nf, const double* __restrict__ tf,
                                                           const double* __restrict__ zf, double* __restrict__ scal) {
  double rp = 0.0, na = 0.0, nz = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nf; t += (int64_t)gridDim.x * 256) {
    const double a = tf[t], z = zf[t];
    rp = fmax(rp, fabs(a - z));
    na = fmax(na, fabs(a));
    nz = fmax(nz, fabs(z));
  }
  rp = block_max(rp);
  na = block_max(na);
  nz = block_max(nz);
  if (threadIdx.x == 0) {
    atomic_max_nonneg(scal + SL_RP, rp);
    atomic_max_nonneg(scal + SL_NAX, na);
    atomic_max_nonneg(scal + SL_NZ, nz);
  }
}

template <int D>
__global__ __launch_bounds__(256) void resid_rows_kernel(int64_t nW, int64_t C, const int* __restrict__ wk,
                                                          const int* __restrict__ wi, const int* __restrict__ wj,
                                                          const double* __restrict__ weta,
                                                          const double* __restrict__ Q, const double* __restrict__ zc,
                                                          double* __restrict__ scal) {
  double rp = 0.0, na = 0.0, nz = 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nW; n += (int64_t)gridDim.x * 256) {
    const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
    const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
    double a = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) a += weta[n * D + d] * (Q[bi + d] - Q[bj + d]);
    const double z = zc[n];
    rp = fmax(rp, fabs(a - z));
    na = fmax(na, fabs(a));
    nz = fmax(nz, fabs(z));
  }
  rp = block_max(rp);
  na = block_max(na);
  nz = block_max(nz);
  if (threadIdx.x == 0) {
    atomic_max_nonneg(scal + SL_RP, rp);
    atomic_max_nonneg(scal + SL_NAX, na);
    atomic_max_nonneg(scal + SL_NZ, nz);
  }
}

// dual residual pieces: max|2x + ATy|, max|2x|, max|ATy|
__global__ __launch_bounds__(256) void resid_dual_kernel(int64_t nx, const double* __restrict__ x,
                                                          const double* __restrict__ aty, double* __restrict__ scal) {
  double rd = 0.0, npx = 0.0, nat = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nx; t += (int64_t)gridDim.x * 256) {
    const double px = 2.0 * x[t], a = aty[t];
    rd = fmax(rd, fabs(px + a));
    npx = fmax(npx, fabs(px));
    nat = fmax(nat, fabs(a));
  }
  rd = block_max(rd);
  npx = block_max(npx);
  nat = block_max(nat);
  if (threadIdx.x == 0) {
    atomic_max_nonneg(scal + SL_RD, rd);
    atomic_max_nonneg(scal + SL_NPX, npx);
    atomic_max_nonneg(scal + SL_NATY, nat);
  }
}

// primal infeasibility certificate, fixed rows: dy = y - snapshot (in place), max |dy|, sum u dy+ + l dy-
__global__ __launch_bounds__(256) void dy_fixed_kernel(int64_t nf, const double* __restrict__ yf,
                                                        const double* __restrict__ lf, const double* __restrict__ uf,
                                                        double* __restrict__ dyf, double* __restrict__ scal) {
  __shared__ double ssum[4];
  double mx = 0.0, sup = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nf; t += (int64_t)gridDim.x * 256) {
    const double d = yf[t] - dyf[t];
    dyf[t] = d;
    mx = fmax(mx, fabs(d));
    sup += uf[t] * fmax(d, 0.0) + lf[t] * fmin(d, 0.0);
  }
  mx = block_max(mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sup += __shfl_xor(sup, o);
  if ((threadIdx.x & 63) == 0) ssum[threadIdx.x >> 6] = sup;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_max_nonneg(scal + SL_NDY, mx);
    atomicAdd(scal + SL_SUPP, (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]));
  }
}

// collision rows (u = +inf): dy = min(y - snapshot, 0)
__global__ __launch_bounds__(256) void dy_rows_kernel(int64_t nW, const double* __restrict__ yc,
                                                       const double* __restrict__ wl, double* __restrict__ dyc,
                                                       double* __restrict__ scal) {
  __shared__ double ssum[4];
  double mx = 0.0, sup = 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nW; n += (int64_t)gridDim.x * 256) {
    const double d = fmin(yc[n] - dyc[n], 0.0);
    dyc[n] = d;
    mx = fmax(mx, fabs(d));
    sup += wl[n] * d;
  }
  mx = block_max(mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sup += __shfl_xor(sup, o);
  if ((threadIdx.x & 63) == 0) ssum[threadIdx.x >> 6] = sup;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_max_nonneg(scal + SL_NDY, mx);
    atomicAdd(scal + SL_SUPP, (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]));
  }
}

__global__ __launch_bounds__(256) void max_abs_kernel(int64_t n, const double* __restrict__ v, double* __restrict__ slot) {
  double mx = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) mx = fmax(mx, fabs(v[t]));
  mx = block_max(mx);
  if (threadIdx.x == 0) atomic_max_nonneg(slot, mx);
}

// G0[a][b] = sum_r w_r F[r][a] F[r][b]  (constant per problem shape: once at create)
__global__ __launch_bounds__(256) void build_g0_kernel(int K, int Rf, const double* __restrict__ F,
                                                        const double* __restrict__ wrow, double* __restrict__ G0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= K * K) return;
  const int a = t / K, b = t % K;
  double s = 0.0;
  for (int r = 0; r < Rf; ++r) s += wrow[r] * F[(int64_t)r * K + a] * F[(int64_t)r * K + b];
  G0[t] = s;
}

// Hf[a][b] = (2 + sigma) delta_ab + rho G0[a][b];  HS = [Hf ; S0];  aug = [Hf | I]
__global__ __launch_bounds__(256) void build_hf_kernel(int K, double rho, double sigma, const double* __restrict__ G0,
                                                        const double* __restrict__ S0, double* __restrict__ Hf,
                                                        double* __restrict__ HS, double* __restrict__ aug) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= K * K) return;
  const int a = t / K, b = t % K;
  const double v = rho * G0[t] + (a == b ? 2.0 + sigma : 0.0);
  Hf[t] = v;
  HS[t] = v;
  HS[K * K + t] = S0[t];
  aug[(int64_t)a * 2 * K + b] = v;
  aug[(int64_t)a * 2 * K + K + b] = a == b ? 1.0 : 0.0;
}

// Gauss-Jordan inverse with [Hf | I] resident in LDS (K <= SCP_INV_LDS_MAX_K); same operations as the global one.
// Thread (ty, tx): column tx of the augmented matrix, rows ty, ty + TY, ... (no integer divisions in the pivot loop).
__global__ __launch_bounds__(1024) void spd_inverse_lds_kernel(int K, const double* __restrict__ Hf, double* __restrict__ Minv) {
  extern __shared__ double sh[];  // aug[K][2K] | prow[2K] | col[K]
  const int W = 2 * K;
  double* aug = sh;
  double* prow = aug + K * W;
  double* col = prow + W;
  const int TX = W <= 128 ? 128 : 256, TY = 1024 / TX;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;
  if (tx < W)
    for (int r = ty; r < K; r += TY) aug[r * W + tx] = tx < K ? Hf[r * K + tx] : (tx - K == r ? 1.0 : 0.0);
  __syncthreads();
  for (int p = 0; p < K; ++p) {
    const double piv = aug[p * W + p];
    if (threadIdx.x < W) prow[threadIdx.x] = aug[p * W + threadIdx.x] / piv;
    else if (threadIdx.x >= 512 && threadIdx.x - 512 < K) col[threadIdx.x - 512] = aug[(threadIdx.x - 512) * W + p];
    __syncthreads();
    if (tx < W) {
      const double pr = prow[tx];
      for (int r = ty; r < K; r += TY) {
        if (r == p) aug[r * W + tx] = pr;
        else aug[r * W + tx] -= col[r] * pr;
      }
    }
    __syncthreads();
  }
  if (tx < K)
    for (int r = ty; r < K; r += TY) Minv[r * K + tx] = aug[r * W + K + tx];
}

// Gauss-Jordan inverse for K > SCP_INV_LDS_MAX_K, two small launches per pivot over the whole chip: the pivot row (scaled)
// and the pivot column are first copied out, then every element of aug = [Hf | I] is updated from them -- the same
// operation per element as the one-workgroup kernels (bit-identical result), but a pivot's K x 2K update is spread over
// all CUs instead of dragging the 4 MB matrix (K = 500) through one CU 500 times (61 ms -> ~3 ms per inverse; the
// reference's demo, K = 500, spent 88 % of its 0.65 s there).
__global__ __launch_bounds__(256) void gj_extract_kernel(int K, int p, const double* __restrict__ aug, double* __restrict__ prow,
                                                          double* __restrict__ pcol) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int W = 2 * K;
  if (t < W) prow[t] = aug[(int64_t)p * W + t] / aug[(int64_t)p * W + p];
  else if (t - W < K) pcol[t - W] = aug[(int64_t)(t - W) * W + p];
}
__global__ __launch_bounds__(256) void gj_update_kernel(int K, int p, double* __restrict__ aug, const double* __restrict__ prow,
                                                         const double* __restrict__ pcol) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W = 2 * K;
  if (e >= (int64_t)K * W) return;
  const int r = (int)(e / W), c = (int)(e % W);
  if (r == p) aug[e] = prow[c];
  else aug[e] -= pcol[r] * prow[c];
}
__global__ __launch_bounds__(256) void gj_finish_kernel(int K, const double* __restrict__ aug, double* __restrict__ Minv) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < K * K) Minv[t] = aug[(int64_t)(t / K) * 2 * K + K + (t % K)];
}

// append working rows: decode (k, i, j), copy eta / l, z = max(A x, l), y = 0
// eta_stride == 0: eta_in / l_in are the gathered [n][D] / [n] arrays of scp_gather_rows; otherwise they are the arrays of
// the pairwise pass itself (pair range [q_begin, q_begin + nq)) and the gather happens here.
__global__ __launch_bounds__(256) void add_rows_kernel(int N, int D, int64_t C, int64_t pairs, int64_t base, int64_t n,
                                                        const int64_t* __restrict__ rows,
                                                        const double* __restrict__ eta_in,
                                                        const double* __restrict__ l_in, int64_t eta_stride,
                                                        int64_t q_begin, int64_t nq, const double* __restrict__ Q,
                                                        int64_t* __restrict__ w_row, int* __restrict__ wk,
                                                        int* __restrict__ wi, int* __restrict__ wj,
                                                        double* __restrict__ weta, double* __restrict__ wl,
                                                        double* __restrict__ zc, double* __restrict__ yc) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int64_t r = rows[t];
  const int64_t k = r / pairs, q = r % pairs;
  // lexicographic pair index -> (i, j)
  const double b = 2.0 * N - 1.0;
  int64_t ii = (int64_t)((b - sqrt(b * b - 8.0 * (double)q)) * 0.5);
  if (ii < 0) ii = 0;
  if (ii > N - 2) ii = N - 2;
  while (ii * (2LL * N - ii - 1) / 2 > q) --ii;
  while (ii < N - 2 && (ii + 1) * (2LL * N - ii - 2) / 2 <= q) ++ii;
  const int64_t jj = q - ii * (2LL * N - ii - 1) / 2 + ii + 1;
  const int64_t o = base + t;
  w_row[o] = r;
  wk[o] = (int)k;
  wi[o] = (int)ii;
  wj[o] = (int)jj;
  const int64_t lr = k * nq + (q - q_begin);
  double ax = 0.0;
  for (int d = 0; d < D; ++d) {
    const double e = eta_stride ? eta_in[(int64_t)d * eta_stride + lr] : eta_in[t * D + d];
    weta[o * D + d] = e;
    ax += e * (Q[k * C + ii * D + d] - Q[k * C + jj * D + d]);
  }
  const double lo = eta_stride ? l_in[lr] : l_in[t];
  wl[o] = lo;
  zc[o] = fmax(ax, lo);
  yc[o] = 0.0;
}

// scp_qp_reset in one launch (K <= SCP_FUSED_MAX_K): x0 in reference order [N][K][D] (null: zeros) -> x (time-major),
// z_f = F x, the carried F x and S0 x of the single-step pipeline (exact), y_f = 0.  16 columns per workgroup (256 columns
// of a 128-agent problem still make 16 workgroups), the x tile in LDS, thread = (column, one sixteenth of the rows).
__global__ __launch_bounds__(256) void qp_reset_kernel(int N, int K, int D, int Rf, const double* __restrict__ x0,
                                                        const double* __restrict__ F, const double* __restrict__ S0,
                                                        double* __restrict__ x, double* __restrict__ zf,
                                                        double* __restrict__ fx, double* __restrict__ Qx,
                                                        double* __restrict__ yf) {
  extern __shared__ double reset_xs[];  // [K][RESET_COLS]
  qp_reset_body<false>(threadIdx.x, true, reset_xs, N, K, D, Rf, x0, F, S0, x, zf, fx, Qx, yf);
}

// ----------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------
namespace {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
  char* base;
  size_t off;
  template <typename T>
  T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += align_up(count * sizeof(T));
    return p;
  }
};

size_t kkt_slot_doubles(int K) {  // Hf, HS, Minv, T + packed HS, Minv, T, each 256-byte aligned
  auto al = [](size_t n) { return (n + 31) / 32 * 32; };
  return al((size_t)K * K) * 3 + al((size_t)2 * K * K) + al(scp_packed_count(2 * K, K)) + 2 * al(scp_packed_count(K, K));
}

// Adaptive rho moves on a geometric grid (steps of 2^(1/4)), so a solver object that is reused from scenario to scenario
// (compute-trajectories-batch) keeps meeting the same values: short horizons get enough slots to hold them all.
int kkt_slots(int K) {
  const size_t per = kkt_slot_doubles(K) * sizeof(double);
  const size_t fit = SCP_KKT_POOL_BYTES / per;
  return (int)std::min<size_t>(SCP_KKT_SLOTS_MAX, std::max<size_t>(SCP_KKT_SLOTS, fit));
}

size_t carve(QpDev& d, void* ws, int K, int64_t C, int64_t cap, int D) {
  const int Rf = 4 * K - 1;
  Carver c{static_cast<char*>(ws), 0};
  d.F = c.take<double>((size_t)Rf * K);
  d.Ft = c.take<double>((size_t)Rf * K);
  d.S0 = c.take<double>((size_t)K * K);
  d.S0t = c.take<double>((size_t)K * K);
  d.HS = d.Hf = d.Minv = nullptr;  // rho-dependent blocks live in the cache slots (carve_kkt_slots), set by build_kkt
  d.aug = c.take<double>((size_t)2 * K * K);
  d.gj_tmp = c.take<double>((size_t)3 * K);
  d.wrow = c.take<double>((size_t)Rf);
  d.G0 = c.take<double>((size_t)K * K);
  d.pF = c.take<double>(scp_packed_count(Rf, K));
  d.pFt = c.take<double>(scp_packed_count(K, Rf));
  d.pS0 = c.take<double>(scp_packed_count(K, K));
  d.pS0t = c.take<double>(scp_packed_count(K, K));
  d.pHS = d.pMinv = d.T = d.pT = nullptr;
  d.kkt_pool = c.take<double>((size_t)kkt_slots(K) * kkt_slot_doubles(K));
  const size_t nf = (size_t)Rf * C, nx = (size_t)K * C;
  d.lf = c.take<double>(nf);
  d.uf = c.take<double>(nf);
  d.zf = c.take<double>(nf);
  d.yf = c.take<double>(nf);
  d.wf = c.take<double>(nf);
  d.tf = c.take<double>(nf);
  d.states = c.take<double>((size_t)4 * C);
  d.x = c.take<double>(nx);
  d.xt = c.take<double>(nx);
  d.rhs = c.take<double>(nx);
  d.r = c.take<double>(nx);
  d.p = c.take<double>(nx);
  d.zz = c.take<double>(nx);
  d.G = c.take<double>(nx);
  d.HQ = c.take<double>(2 * nx);
  d.w_row = c.take<int64_t>((size_t)cap);
  d.w_k = c.take<int>((size_t)cap);
  d.w_i = c.take<int>((size_t)cap);
  d.w_j = c.take<int>((size_t)cap);
  d.w_eta = c.take<double>((size_t)cap * D);
  d.w_l = c.take<double>((size_t)cap);
  d.zc = c.take<double>((size_t)cap);
  d.yc = c.take<double>((size_t)cap);
  d.scal = c.take<double>(SL_COUNT + SCP_RESID_CAP);
  d.part = c.take<double>(2 * SCP_PART_CAP);
  d.hpf = c.take<double>(nx);
  d.fx = c.take<double>(nf);
  d.dyf = c.take<double>(nf);
  d.dyc = c.take<double>((size_t)cap);
  const size_t ncell = (size_t)(C / D) * K;
  d.cell_ptr = c.take<int>(ncell + 1);
  d.cell_cur = c.take<int>(ncell);
  d.scan_tot = c.take<int>(ncell / 4096 + 2);
  d.ent_code = c.take<int>((size_t)2 * cap);
  d.coef = c.take<double>((size_t)2 * cap * D);
  d.gval = c.take<double>((size_t)2 * cap);
  d.gval2 = c.take<double>((size_t)2 * cap);
  d.gval3 = c.take<double>((size_t)2 * cap);
  d.pos_i = c.take<int>((size_t)cap);
  d.pos_j = c.take<int>((size_t)cap);
  d.sync_words = c.take<unsigned long long>(SCP_SYNC_WORDS);
  d.cells = c.take<unsigned long long>((size_t)2 * nx);
  d.gpart = c.take<unsigned long long>(SCP_GPART_WORDS);
  d.gcheck = c.take<unsigned long long>(SCP_GCHECK_WORDS);
  return c.off;
}

inline dim3 grid1(int64_t n) { return dim3(scp_cdiv(n, 256)); }

#define QP_CHECK(call)            \
  do {                            \
    int rc_ = (call);             \
    if (rc_ != SCP_OK) return rc_; \
  } while (0)

#define QP_LAUNCHED(qp) SCP_HIP_CHECK((qp)->ctx, hipGetLastError())

int gemm(scp_qp* qp, int R, int M, double alpha, const double* A, const double* X, double beta, double* Y) {
  return scp_launch_gemm(qp->ctx, qp->st.use_mfma, R, M, (int)qp->C, alpha, A, X, beta, Y);
}

// G = A_W^T g over the working rows, g by mode (a gather over the sorted incidence lists: fixed summation order; the
// round-1 version scattered with atomics):
//   ROW_RHS : rho zc - yc            (right-hand side of the x-update)
//   ROW_HMUL: rho eta.(Q_i - Q_j)    (A_W^T R_c A_W v, Q = S0 v)
//   ROW_Y   : yc                     (A_W^T y for the dual residual)
//   ROW_VEC : vec                    (an arbitrary row vector)
template <int MODE>
int row_scatter(scp_qp* qp, const double* Q, const double* vec = nullptr) {
  if (!qp->csr_valid) QP_CHECK(scp_qp_csr_build(qp));
  if (MODE == ROW_HMUL) return scp_qp_rows_gather(qp, false, Q);
  return scp_qp_csr_scatter(qp, MODE == ROW_RHS ? 0 : (MODE == ROW_Y ? 1 : 2), vec);
}

// HQ[0:K] = H v = Hf v + A_W^T R_c A_W v ; HQ[K:2K] = S0 v
int hmul(scp_qp* qp, const double* v) {
  const QpDev& d = qp->d;
  const int K = qp->K;
  QP_CHECK(gemm(qp, 2 * K, K, 1.0, d.HS, v, 0.0, d.HQ));
  if (qp->nW > 0) {
    QP_CHECK(row_scatter<ROW_HMUL>(qp, d.HQ + (size_t)K * qp->C));
    QP_CHECK(gemm(qp, K, K, 1.0, d.S0t, d.G, 1.0, d.HQ));
  }
  return SCP_OK;
}

int dot_partial(scp_qp* qp, const double* a, const double* b, double* part) {
  hipLaunchKernelGGL(dot_partial_kernel, dim3(NPART), dim3(256), 0, qp->ctx->stream, (int64_t)qp->K * qp->C, a, b,
                     part);
  QP_LAUNCHED(qp);
  return SCP_OK;
}

int build_kkt(scp_qp* qp) {
  QpDev& d = qp->d;
  const int K = qp->K;
  hipStream_t s = qp->ctx->stream;
  if (!qp->consts_packed) {
    QP_CHECK(scp_qp_pack_operands(qp, true));
    qp->consts_packed = true;
  }
  // cache lookup: the blocks of this (rho, sigma) may still be resident
  scp_qp::KktSlot* slot = nullptr;
  for (int i = 0; i < qp->n_kkt; ++i)
    if (qp->kkt[i].used && qp->kkt[i].rho == qp->rho && qp->kkt[i].sigma == qp->st.sigma) slot = &qp->kkt[i];
  const bool hit = slot != nullptr;
  if (!hit) {
    slot = &qp->kkt[0];
    for (int i = 0; i < qp->n_kkt; ++i)
      if (qp->kkt[i].used < slot->used) slot = &qp->kkt[i];  // empty (0) or least recently used
  }
  slot->used = ++qp->kkt_clock;
  d.Hf = slot->Hf; d.HS = slot->HS; d.Minv = slot->Minv; d.T = slot->T;
  d.pHS = slot->pHS; d.pMinv = slot->pMinv; d.pT = slot->pT;
  if (hit) return SCP_OK;
  slot->rho = qp->rho;
  slot->sigma = qp->st.sigma;
  hipLaunchKernelGGL(build_hf_kernel, grid1((int64_t)K * K), dim3(256), 0, s, K, qp->rho, qp->st.sigma, d.G0, d.S0, d.Hf,
                     d.HS, d.aug);
  QP_LAUNCHED(qp);
  if (K <= SCP_INV_LDS_MAX_K) {
    const size_t lds = ((size_t)K * 2 * K + 3 * K) * sizeof(double);
    if (lds > 64 * 1024)
      SCP_HIP_CHECK(qp->ctx, scp_raise_lds_limit(qp->ctx->device, reinterpret_cast<const void*>(spd_inverse_lds_kernel), lds));
    hipLaunchKernelGGL(spd_inverse_lds_kernel, dim3(1), dim3(1024), lds, s, K, d.Hf, d.Minv);
  } else {
    double* prow = d.gj_tmp;
    double* pcol = d.gj_tmp + 2 * K;
    for (int p = 0; p < K; ++p) {
      hipLaunchKernelGGL(gj_extract_kernel, grid1((int64_t)3 * K), dim3(256), 0, s, K, p, d.aug, prow, pcol);
      hipLaunchKernelGGL(gj_update_kernel, grid1((int64_t)K * 2 * K), dim3(256), 0, s, K, p, d.aug, prow, pcol);
    }
    hipLaunchKernelGGL(gj_finish_kernel, grid1((int64_t)K * K), dim3(256), 0, s, K, d.aug, d.Minv);
  }
  QP_LAUNCHED(qp);
  // T = S0 H_f^{-1}: the persistent kernel forms S0 p = T r on spare matrix-core waves next to p = H_f^{-1} r
  QP_CHECK(scp_launch_gemm(qp->ctx, 1, K, K, K, 1.0, d.S0, d.Minv, 0.0, d.T));
  return scp_qp_pack_operands(qp, false);
}

int admm_iteration(scp_qp* qp, int* cg_count) {
  const QpDev& d = qp->d;
  scp_ctx* ctx = qp->ctx;
  hipStream_t s = ctx->stream;
  const int K = qp->K, Rf = qp->Rf;
  const int64_t C = qp->C, nf = (int64_t)Rf * C, nx = (int64_t)K * C;
  // rhs = sigma x + F^T (R_f z_f - y_f) + A_W^T (R_c z_c - y_c)
  hipLaunchKernelGGL(admm_rhs_prep_kernel, grid1(nf), dim3(256), 0, s, nf, nx, C, qp->rho, qp->st.sigma, d.wrow, d.zf,
                     d.yf, d.wf, d.x, d.rhs);
  QP_LAUNCHED(qp);
  QP_CHECK(gemm(qp, K, Rf, 1.0, d.Ft, d.wf, 1.0, d.rhs));
  if (qp->nW > 0) {
    QP_CHECK(row_scatter<ROW_RHS>(qp, nullptr));
    QP_CHECK(gemm(qp, K, K, 1.0, d.S0t, d.G, 1.0, d.rhs));
    // PCG on H x~ = rhs, preconditioner Minv, warm start x~ = x
    SCP_HIP_CHECK(ctx, hipMemcpyAsync(d.xt, d.x, nx * sizeof(double), hipMemcpyDeviceToDevice, s));
    QP_CHECK(hmul(qp, d.xt));
    hipLaunchKernelGGL(cg_residual_kernel, grid1(nx), dim3(256), 0, s, nx, d.rhs, d.HQ, d.r);
    QP_LAUNCHED(qp);
    QP_CHECK(gemm(qp, K, K, 1.0, d.Minv, d.r, 0.0, d.zz));
    QP_CHECK(dot_partial(qp, d.r, d.zz, d.part));
    hipLaunchKernelGGL(cg_start_kernel, grid1(nx), dim3(256), 0, s, nx, d.zz, d.p, d.part, d.scal);
    QP_LAUNCHED(qp);
    int slot = SL_RZ0;
    for (int it = 0; it < qp->st.cg_iters; ++it) {
      QP_CHECK(hmul(qp, d.p));
      QP_CHECK(dot_partial(qp, d.p, d.HQ, d.part));
      hipLaunchKernelGGL(cg_update_kernel, grid1(nx), dim3(256), 0, s, nx, slot, d.scal, d.part, d.p, d.HQ, d.xt, d.r);
      QP_LAUNCHED(qp);
      QP_CHECK(gemm(qp, K, K, 1.0, d.Minv, d.r, 0.0, d.zz));
      QP_CHECK(dot_partial(qp, d.r, d.zz, d.part + NPART));
      hipLaunchKernelGGL(cg_direction_kernel, grid1(nx), dim3(256), 0, s, nx, slot, d.scal, d.part + NPART, d.zz, d.p);
      QP_LAUNCHED(qp);
      slot ^= 1;
      ++*cg_count;
    }
  } else {
    QP_CHECK(gemm(qp, K, K, 1.0, d.Minv, d.rhs, 0.0, d.xt));
  }
  // z~ = A x~, relaxation, projection, duals
  QP_CHECK(gemm(qp, Rf, K, 1.0, d.F, d.xt, 0.0, d.tf));
  if (qp->nW > 0) {
    QP_CHECK(gemm(qp, K, K, 1.0, d.S0, d.xt, 0.0, d.HQ + nx));
    if (qp->D == 2)
      hipLaunchKernelGGL(admm_row_update_kernel<2>, grid1(qp->nW), dim3(256), 0, s, qp->nW, C,
                         qp->rho * qp->st.rho_col_scale, qp->st.alpha,
                         d.w_k, d.w_i, d.w_j, d.w_eta, d.w_l, d.HQ + nx, d.zc, d.yc);
    else
      hipLaunchKernelGGL(admm_row_update_kernel<3>, grid1(qp->nW), dim3(256), 0, s, qp->nW, C,
                         qp->rho * qp->st.rho_col_scale, qp->st.alpha,
                         d.w_k, d.w_i, d.w_j, d.w_eta, d.w_l, d.HQ + nx, d.zc, d.yc);
    QP_LAUNCHED(qp);
  }
  hipLaunchKernelGGL(admm_fixed_update_kernel, grid1(nf), dim3(256), 0, s, nf, nx, C, qp->rho, qp->st.alpha, d.wrow,
                     d.tf, d.lf, d.uf, d.zf, d.yf, d.xt, d.x);
  QP_LAUNCHED(qp);
  return SCP_OK;
}

// residuals -> qp->h_scal[SL_RP..SL_NATY]; with_dy: also delta-y = y - snapshot, its max norm and support value
// (SL_NDY, SL_SUPP).  Synchronises the stream.
int residuals(scp_qp* qp, bool with_dy) {
  const QpDev& d = qp->d;
  scp_ctx* ctx = qp->ctx;
  hipStream_t s = ctx->stream;
  const int K = qp->K, Rf = qp->Rf;
  const int64_t C = qp->C, nf = (int64_t)Rf * C, nx = (int64_t)K * C;
  SCP_HIP_CHECK(ctx, hipMemsetAsync(d.scal + SL_RP, 0, 9 * sizeof(double), s));
  if (with_dy) {
    hipLaunchKernelGGL(dy_fixed_kernel, dim3(256), dim3(256), 0, s, nf, d.yf, d.lf, d.uf, d.dyf, d.scal);
    QP_LAUNCHED(qp);
    if (qp->nW > 0) {
      const int blocks = (int)((qp->nW + 255) / 256) < 256 ? (int)((qp->nW + 255) / 256) : 256;
      hipLaunchKernelGGL(dy_rows_kernel, dim3(blocks), dim3(256), 0, s, qp->nW, d.yc, d.w_l, d.dyc, d.scal);
      QP_LAUNCHED(qp);
    }
  }
  QP_CHECK(gemm(qp, Rf, K, 1.0, d.F, d.x, 0.0, d.tf));
  hipLaunchKernelGGL(resid_fixed_kernel, dim3(256), dim3(256), 0, s, nf, d.tf, d.zf, d.scal);
  QP_LAUNCHED(qp);
  // ATy -> rhs (scratch)
  QP_CHECK(gemm(qp, K, Rf, 1.0, d.Ft, d.yf, 0.0, d.rhs));
  if (qp->nW > 0) {
    QP_CHECK(gemm(qp, K, K, 1.0, d.S0, d.x, 0.0, d.HQ + nx));
    const int blocks = (int)((qp->nW + 255) / 256) < 256 ? (int)((qp->nW + 255) / 256) : 256;
    if (qp->D == 2)
      hipLaunchKernelGGL(resid_rows_kernel<2>, dim3(blocks), dim3(256), 0, s, qp->nW, C, d.w_k, d.w_i, d.w_j, d.w_eta,
                         d.HQ + nx, d.zc, d.scal);
    else
      hipLaunchKernelGGL(resid_rows_kernel<3>, dim3(blocks), dim3(256), 0, s, qp->nW, C, d.w_k, d.w_i, d.w_j, d.w_eta,
                         d.HQ + nx, d.zc, d.scal);
    QP_LAUNCHED(qp);
    QP_CHECK(row_scatter<ROW_Y>(qp, nullptr));
    QP_CHECK(gemm(qp, K, K, 1.0, d.S0t, d.G, 1.0, d.rhs));
  }
  hipLaunchKernelGGL(resid_dual_kernel, dim3(128), dim3(256), 0, s, nx, d.x, d.rhs, d.scal);
  QP_LAUNCHED(qp);
  SCP_HIP_CHECK(ctx, hipMemcpyAsync(qp->h_scal, d.scal, SL_COUNT * sizeof(double), hipMemcpyDeviceToHost, s));
  SCP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  return SCP_OK;
}

// second half of the certificate: || A^T dy ||_inf -> h_scal[SL_NATDY] (synchronises)
int certificate_atdy(scp_qp* qp) {
  const QpDev& d = qp->d;
  scp_ctx* ctx = qp->ctx;
  hipStream_t s = ctx->stream;
  const int K = qp->K, Rf = qp->Rf;
  const int64_t nx = (int64_t)K * qp->C;
  SCP_HIP_CHECK(ctx, hipMemsetAsync(d.scal + SL_NATDY, 0, sizeof(double), s));
  QP_CHECK(gemm(qp, K, Rf, 1.0, d.Ft, d.dyf, 0.0, d.rhs));
  if (qp->nW > 0) {
    QP_CHECK(row_scatter<ROW_VEC>(qp, nullptr, d.dyc));
    QP_CHECK(gemm(qp, K, K, 1.0, d.S0t, d.G, 1.0, d.rhs));
  }
  hipLaunchKernelGGL(max_abs_kernel, dim3(128), dim3(256), 0, s, nx, d.rhs, d.scal + SL_NATDY);
  QP_LAUNCHED(qp);
  SCP_HIP_CHECK(ctx, hipMemcpyAsync(qp->h_scal + SL_NATDY, d.scal + SL_NATDY, sizeof(double), hipMemcpyDeviceToHost, s));
  SCP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  return SCP_OK;
}

}  // namespace

// ----------------------------------------------------------------------------------------------------
// C-ABI
// ----------------------------------------------------------------------------------------------------
extern "C" void scp_qp_default_settings(scp_qp_settings* s) {
  if (!s) return;
  s->rho = 0.1;
  s->sigma = 1e-6;
  s->alpha = 1.6;
  s->rho_eq_scale = 1e3;
  s->eps_abs = 1e-3;
  s->eps_rel = 1e-3;
  s->max_iter = 4000;
  s->check_termination = 25;
  s->check_fine = 5;
  s->check_fine_ratio = 2.0;
  s->adaptive_rho = 1;
  // OSQP's own default is a wall-clock rule (the first update once the iterations have cost a fraction of the setup time, i.e.
  // after some multiple of check_termination): not reproducible, and with no factorisation to amortise there is no setup time
  // to measure.  50 is a measured choice (profiles/r03_rho_interval_sweep.txt: 11 problems, 10 ... 4096 agents): the estimate
  // after 25 steps overshoots on the large problems (1024 x 50: rho 0.1 -> 0.011, 500 steps; after 50 steps: 250), 75 / 100
  // cost the small ones.  Total ADMM steps over the sweep 17 225 (25) / 15 500 (50) / 15 500 (75) / 15 950 (100).
  s->adaptive_rho_interval = 50;
  s->adaptive_rho_tolerance = 5.0;
  s->cg_iters = 1;
  s->use_mfma = 1;
  s->rho_col_scale = 10.0;
  s->eps_prim_inf = 1e-4;
  s->persistent = 1;
}

extern "C" size_t scp_qp_workspace_bytes(int N, int K, int D, int64_t row_capacity) {
  if (N <= 0 || K <= 1 || (D != 2 && D != 3) || row_capacity < 0) return 0;
  QpDev d;
  return carve(d, nullptr, K, (int64_t)N * D, row_capacity, D);
}

static int check_settings(scp_ctx* ctx, const scp_qp_settings* s) {
  SCP_REQUIRE(ctx, s->rho > 0 && s->sigma > 0 && s->alpha > 0 && s->alpha < 2 && s->rho_eq_scale > 0 &&
                       s->rho_col_scale > 0,
              "qp settings: rho/sigma/alpha out of range");
  SCP_REQUIRE(ctx, s->max_iter > 0 && s->check_termination > 0 && s->cg_iters >= 1, "qp settings: bad iteration counts");
  SCP_REQUIRE(ctx, s->check_fine >= 0 && s->check_fine_ratio >= 1.0, "qp settings: check_fine >= 0, check_fine_ratio >= 1");
  return SCP_OK;
}

extern "C" int scp_qp_create(scp_ctx* ctx, int N, int K, int D, double h, const scp_qp_settings* s, void* workspace,
                             size_t workspace_bytes, int64_t row_capacity, scp_qp** out) {
  if (!ctx) return SCP_ERR_INVALID;
  SCP_REQUIRE(ctx, out && s && workspace, "qp_create: null pointer");
  SCP_REQUIRE(ctx, N > 0 && K > 1 && (D == 2 || D == 3) && h > 0 && row_capacity >= 0, "qp_create: bad shape");
  SCP_REQUIRE(ctx, (uintptr_t)workspace % 256 == 0, "qp_create: workspace must be 256-byte aligned");
  int rc = check_settings(ctx, s);
  if (rc) return rc;
  const size_t need = scp_qp_workspace_bytes(N, K, D, row_capacity);
  if (workspace_bytes < need)
    return scp_fail(ctx, SCP_ERR_CAPACITY, "qp_create: workspace %zu < %zu bytes", workspace_bytes, need);
  scp_qp* qp = new scp_qp();
  qp->ctx = ctx;
  qp->N = N; qp->K = K; qp->D = D; qp->Rf = 4 * K - 1; qp->C = (int64_t)N * D; qp->h = h;
  qp->st = *s;
  qp->row_cap = row_capacity;
  qp->nW = 0;
  qp->problem_set = qp->reset_done = false;
  qp->cg1_ready = qp->csr_valid = qp->qx_fresh = qp->gval_valid = false;
  qp->qx_sel = 0;
  qp->rho = s->rho;
  carve(qp->d, workspace, K, qp->C, row_capacity, D);
  {
    auto al = [](size_t n) { return (n + 31) / 32 * 32; };
    double* base = qp->d.kkt_pool;
    qp->n_kkt = kkt_slots(K);
    for (int i = 0; i < qp->n_kkt; ++i) {
      auto& k = qp->kkt[i];
      double* q = base;
      k.rho = k.sigma = 0.0;
      k.used = 0;
      k.Hf = q; q += al((size_t)K * K);
      k.Minv = q; q += al((size_t)K * K);
      k.T = q; q += al((size_t)K * K);
      k.HS = q; q += al((size_t)2 * K * K);
      k.pHS = q; q += al(scp_packed_count(2 * K, K));
      k.pMinv = q; q += al(scp_packed_count(K, K));
      k.pT = q;
      base += kkt_slot_doubles(K);
    }
    qp->kkt_clock = 0;
    qp->consts_packed = false;
  }
  qp->check_seq = 0;
  qp->persist_off = false;
  qp->persist_skip_solve = false;
  qp->persist_gave_up_total = 0;
  qp->persist_variant = 0;
  qp->steps_since_reset = 0;
  memset(qp->lim, 0, sizeof(qp->lim));
  qp->persist_fault = 0;
  qp->persist_cap_nW = -1;
  qp->persist_cap = 0;
  qp->persist_epoch = 0;
  if (hipHostMalloc(&qp->h_scal, (SL_COUNT + SCP_RESID_CAP + 4) * sizeof(double),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&qp->h_scal_dev, qp->h_scal, 0) != hipSuccess) {
    delete qp;
    return scp_fail(ctx, SCP_ERR_HIP, "qp_create: hipHostMalloc failed");
  }
  memset(qp->h_scal, 0, (SL_COUNT + SCP_RESID_CAP + 4) * sizeof(double));  // incl. the completion flag
  qp->h_persist = (unsigned*)(qp->h_scal + SL_COUNT + SCP_RESID_CAP + 1);  // status word of the persistent kernel
  qp->h_persist_dev = (unsigned*)(qp->h_scal_dev + SL_COUNT + SCP_RESID_CAP + 1);
  // constant blocks (scp.py:10-28, :198-203, :227-232, :489-491), built on the host once per (K, h)
  const int Rf = qp->Rf;
  std::vector<double> F((size_t)Rf * K, 0.0), Ft((size_t)Rf * K, 0.0), S0((size_t)K * K, 0.0), S0t((size_t)K * K, 0.0),
      w(Rf, 1.0);
  const double hh = h * h;
  for (int k = 0; k < K - 1; ++k) {  // jerk
    F[(size_t)k * K + k] = -1.0 / h;
    F[(size_t)k * K + k + 1] = 1.0 / h;
  }
  for (int k = 0; k < K; ++k) {
    F[(size_t)(K - 1 + k) * K + k] = 1.0;                                           // acc
    for (int m = 0; m <= k; ++m) F[(size_t)(2 * K - 1 + k) * K + m] = h;              // vel (state k+1)
    for (int m = 0; m <= k; ++m) F[(size_t)(3 * K - 1 + k) * K + m] = hh * (k - m + 0.5);  // pos (state k+1)
    for (int m = 0; m < k; ++m) S0[(size_t)k * K + m] = hh * (k - m - 0.5);           // stored sample k
  }
  for (int r = 0; r < Rf; ++r)
    for (int m = 0; m < K; ++m) Ft[(size_t)m * Rf + r] = F[(size_t)r * K + m];
  for (int k = 0; k < K; ++k)
    for (int m = 0; m < K; ++m) S0t[(size_t)m * K + k] = S0[(size_t)k * K + m];
  w[2 * K - 1 + K - 1] = s->rho_eq_scale;  // final velocity equality (scp.py:223-224)
  w[3 * K - 1 + K - 1] = s->rho_eq_scale;  // final position equality (scp.py:256-257)
  hipStream_t st = ctx->stream;
  const QpDev& d = qp->d;
  bool ok = hipMemcpyAsync(d.F, F.data(), F.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(d.Ft, Ft.data(), Ft.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(d.S0, S0.data(), S0.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(d.S0t, S0t.data(), S0t.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
            hipMemcpyAsync(d.wrow, w.data(), w.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess;
  if (!ok) {
    (void)hipHostFree(qp->h_scal);
    delete qp;
    return scp_fail(ctx, SCP_ERR_HIP, "qp_create: constant upload failed");
  }
  hipLaunchKernelGGL(build_g0_kernel, grid1((int64_t)K * K), dim3(256), 0, st, K, Rf, d.F, d.wrow, d.G0);
  if (hipGetLastError() != hipSuccess) {
    (void)hipHostFree(qp->h_scal);
    delete qp;
    return scp_fail(ctx, SCP_ERR_HIP, "qp_create: launch failed");
  }
  *out = qp;
  return SCP_OK;
}

extern "C" void scp_qp_destroy(scp_qp* qp) {
  if (!qp) return;
  (void)hipStreamSynchronize(qp->ctx->stream);
  (void)hipHostFree(qp->h_scal);
  delete qp;
}

extern "C" int scp_qp_update_settings(scp_qp* qp, const scp_qp_settings* s) {
  if (!qp || !s) return SCP_ERR_INVALID;
  int rc = check_settings(qp->ctx, s);
  if (rc) return rc;
  SCP_REQUIRE(qp->ctx, s->rho_eq_scale == qp->st.rho_eq_scale, "qp_update_settings: rho_eq_scale is fixed at create");
  qp->st = *s;
  return SCP_OK;
}

extern "C" int scp_qp_set_problem(scp_qp* qp, const double* limits, const double* space, const double* p0,
                                  const double* v0, const double* pf, const double* vf) {
  if (!qp) return SCP_ERR_INVALID;
  SCP_REQUIRE(qp->ctx, limits && space && p0 && v0 && pf && vf, "qp_set_problem: null pointer");
  QP_CHECK(scp_launch_bounds_time_major(qp->ctx, qp->N, qp->K, qp->D, qp->h, limits, space, p0, v0, pf, vf, qp->d.lf,
                                        qp->d.uf, qp->d.states));
  memcpy(qp->lim, limits, sizeof(qp->lim));
  for (int d = 0; d < 3; ++d) {
    qp->space[d] = d < qp->D ? space[d] : 0.0;
    qp->space[3 + d] = d < qp->D ? space[qp->D + d] : 0.0;
  }
  qp->problem_set = true;
  qp->reset_done = false;
  qp->persist_off = false;  // a give-up is a property of the moment (another kernel held the CUs), not of the object
  return SCP_OK;
}

// eta_stride == 0: gathered rows (the public entry point); > 0: eta / l are the arrays of the pairwise pass over the pair
// range [q_begin, q_begin + nq), gathered by the kernel; at != nullptr: no stored rows at all, eta / l are recomputed from
// the linearisation point
struct RowsAt {
  const double *pos_prev, *p0, *v0;
  double R;
};

// scp_qp_reset; with `at`: the QP's first rows (recomputed from the linearisation point, scp_qp_add_rows_at) are installed
// by the SAME launch when the problem is small (*installed; otherwise only the reset has happened)
static int reset_impl(scp_qp* qp, const double* x0, int64_t n, const int64_t* rows, const RowsAt* at, bool* installed) {
  scp_ctx* ctx = qp->ctx;
  if (installed) *installed = false;
  if (!qp->problem_set) return scp_fail(ctx, SCP_ERR_STATE, "qp_reset: call scp_qp_set_problem first");
  const QpDev& d = qp->d;
  const int64_t nx = (int64_t)qp->K * qp->C, nf = (int64_t)qp->Rf * qp->C;
  const bool one_launch = qp->st.use_mfma == 1 && qp->K <= SCP_FUSED_MAX_K;
  qp->rho = qp->st.rho;
  bool with_rows = false;
  if (one_launch && at && ctx->small_pass && qp->st.cg_iters == 1 && n > 0 && n <= qp->row_cap)
    QP_CHECK(scp_qp_reset_install_small(qp, x0, n, rows, at->pos_prev, at->p0, at->v0, at->R, d.HQ + nx, &with_rows));
  if (with_rows) {
    qp->qx_sel = 0;
  } else if (one_launch) {
    // z = A x (primal warm start, scp.py:443), y = 0 and the single-step pipeline's carried F x, S0 x in one launch
    hipLaunchKernelGGL(qp_reset_kernel, dim3(scp_cdiv(qp->C, RESET_COLS)), dim3(256),
                       (size_t)qp->K * RESET_COLS * sizeof(double), ctx->stream, qp->N, qp->K, qp->D, qp->Rf, x0, d.F, d.S0,
                       d.x, d.zf, d.fx, d.HQ + nx, d.yf);
    QP_LAUNCHED(qp);
    qp->qx_sel = 0;
  } else {
    if (x0) QP_CHECK(scp_launch_to_time_major(ctx, qp->N, qp->K, qp->D, x0, d.x));
    else SCP_HIP_CHECK(ctx, hipMemsetAsync(d.x, 0, nx * sizeof(double), ctx->stream));
    QP_CHECK(gemm(qp, qp->Rf, qp->K, 1.0, d.F, d.x, 0.0, d.zf));  // z = A x  (primal warm start, scp.py:443)
    SCP_HIP_CHECK(ctx, hipMemsetAsync(d.yf, 0, nf * sizeof(double), ctx->stream));
  }
  qp->nW = with_rows ? n : 0;
  qp->persist_cap_nW = -1;
  qp->persist_off = false;  // every new QP tries the persistent path again
  qp->steps_since_reset = 0;
  qp->cg1_ready = false;
  qp->gval_valid = with_rows;  // (incidence lists and row values of the installed rows, at gval_rho_c)
  qp->csr_valid = with_rows;
  qp->qx_fresh = one_launch;  // F x and S0 x of this x are in place
  QP_CHECK(build_kkt(qp));
  qp->reset_done = true;
  if (installed) *installed = with_rows;
  return SCP_OK;
}

extern "C" int scp_qp_reset(scp_qp* qp, const double* x0) {
  if (!qp) return SCP_ERR_INVALID;
  return reset_impl(qp, x0, 0, nullptr, nullptr, nullptr);
}

extern "C" int scp_qp_set_rho(scp_qp* qp, double rho) {
  if (!qp) return SCP_ERR_INVALID;
  if (!qp->reset_done) return scp_fail(qp->ctx, SCP_ERR_STATE, "qp_set_rho: call scp_qp_reset first");
  SCP_REQUIRE(qp->ctx, rho >= 1e-6 && rho <= 1e6, "qp_set_rho: rho out of range");
  qp->rho = rho;
  qp->cg1_ready = false;  // the carried row values depend on rho
  qp->gval_valid = false;
  return build_kkt(qp);
}

static int add_rows_impl(scp_qp* qp, int64_t n, const int64_t* rows, const double* eta, const double* l, int64_t eta_stride,
                         int64_t q_begin, int64_t nq, const RowsAt* at = nullptr) {
  scp_ctx* ctx = qp->ctx;
  if (!qp->reset_done) return scp_fail(ctx, SCP_ERR_STATE, "qp_add_rows: call scp_qp_reset first");
  if (n <= 0) return SCP_OK;
  SCP_REQUIRE(ctx, rows && (at || (eta && l)), "qp_add_rows: null pointer");
  if (qp->nW + n > qp->row_cap)
    return scp_fail(ctx, SCP_ERR_CAPACITY, "qp_add_rows: %lld + %lld rows exceed the capacity %lld",
                    (long long)qp->nW, (long long)n, (long long)qp->row_cap);
  const QpDev& d = qp->d;
  const int64_t nx = (int64_t)qp->K * qp->C;
  // S0 x for z_c = max(A_c x, l): the single-step pipeline's exact copy when there is one (after a reset, after a solve's
  // last check), else computed into the spare slab
  const double* Qx = qp->qx_sel ? d.HQ : d.HQ + nx;
  if (!qp->qx_fresh) {
    QP_CHECK(gemm(qp, qp->K, qp->K, 1.0, d.S0, d.x, 0.0, d.HQ + nx));  // (HQ is scratch whenever qx_fresh is false)
    Qx = d.HQ + nx;
  }
  if (at) {
    bool installed = false;
    if (qp->st.use_mfma == 1 && qp->st.cg_iters == 1 && qp->K <= SCP_FUSED_MAX_K)  // (the single-step pipelines' lists)
      QP_CHECK(scp_qp_install_rows_small(qp, n, rows, at->pos_prev, at->p0, at->v0, at->R, Qx, &installed));
    if (installed) {
      qp->nW += n;
      qp->persist_cap_nW = -1;
      qp->cg1_ready = false;
      return SCP_OK;
    }
    QP_CHECK(scp_launch_add_rows_at(ctx, qp->N, qp->K, qp->D, qp->nW, n, rows, at->pos_prev, at->p0, at->v0, at->R, qp->h, Qx,
                                    d.w_row, d.w_k, d.w_i, d.w_j, d.w_eta, d.w_l, d.zc, d.yc));
  } else {
    hipLaunchKernelGGL(add_rows_kernel, grid1(n), dim3(256), 0, ctx->stream, qp->N, qp->D, qp->C, scp_pairs(qp->N),
                       qp->nW, n, rows, eta, l, eta_stride, q_begin, nq, Qx, d.w_row, d.w_k, d.w_i, d.w_j, d.w_eta, d.w_l,
                       d.zc, d.yc);
    QP_LAUNCHED(qp);
  }
  qp->nW += n;
  qp->persist_cap_nW = -1;
  qp->cg1_ready = false;
  qp->gval_valid = false;
  qp->csr_valid = false;
  return SCP_OK;
}

extern "C" int scp_qp_add_rows(scp_qp* qp, int64_t n, const int64_t* rows, const double* w_eta, const double* w_l) {
  if (!qp) return SCP_ERR_INVALID;
  return add_rows_impl(qp, n, rows, w_eta, w_l, 0, 0, 1);
}

extern "C" int scp_qp_add_rows_at(scp_qp* qp, int64_t n, const int64_t* rows, const double* pos_prev, const double* p0,
                                  const double* v0, double R) {
  if (!qp) return SCP_ERR_INVALID;
  SCP_REQUIRE(qp->ctx, n <= 0 || (pos_prev && p0 && v0), "qp_add_rows_at: null pointer");
  const RowsAt at{pos_prev, p0, v0, R};
  return add_rows_impl(qp, n, rows, nullptr, nullptr, 0, 0, 1, &at);
}

// scp_qp_reset(x0) followed by scp_qp_add_rows_at(rows): one launch for small problems (used by the native SCP loop), the
// two calls otherwise.  Same state, same bits either way.
int scp_qp_reset_add_rows_at(scp_qp* qp, const double* x0, int64_t n, const int64_t* rows, const double* pos_prev,
                             const double* p0, const double* v0, double R) {
  if (!qp) return SCP_ERR_INVALID;
  SCP_REQUIRE(qp->ctx, n <= 0 || (rows && pos_prev && p0 && v0), "qp_reset_add_rows_at: null pointer");
  const RowsAt at{pos_prev, p0, v0, R};
  bool installed = false;
  QP_CHECK(reset_impl(qp, x0, n, rows, &at, &installed));
  if (installed) return SCP_OK;
  return add_rows_impl(qp, n, rows, nullptr, nullptr, 0, 0, 1, &at);
}

// scp_gather_rows + scp_qp_add_rows in one launch (used by the native SCP loop): eta / l_col are the outputs of
// scp_linearize_pairs over the pair range [q_begin, q_end)
int scp_qp_add_rows_from_pass(scp_qp* qp, int64_t n, const int64_t* rows, const double* eta, const double* l_col,
                              int64_t q_begin, int64_t q_end) {
  if (!qp) return SCP_ERR_INVALID;
  SCP_REQUIRE(qp->ctx, q_begin >= 0 && q_end > q_begin && q_end <= scp_pairs(qp->N), "qp_add_rows_from_pass: bad pair range");
  return add_rows_impl(qp, n, rows, eta, l_col, scp_eta_stride(qp->K, q_end - q_begin), q_begin, q_end - q_begin);
}

extern "C" int scp_qp_solve(scp_qp* qp, scp_qp_info* info) {
  if (!qp) return SCP_ERR_INVALID;
  scp_ctx* ctx = qp->ctx;
  if (!qp->reset_done) return scp_fail(ctx, SCP_ERR_STATE, "qp_solve: call scp_qp_reset first");
  SCP_REQUIRE(ctx, info, "qp_solve: null info");
  const bool fused = qp->st.use_mfma == 1 && qp->K <= SCP_FUSED_MAX_K && (qp->C + 15) / 16 <= SCP_PART_CAP / 2;
  const scp_qp_settings& st = qp->st;
  memset(info, 0, sizeof(*info));
  info->status_val = -2;  // OSQP_MAX_ITER_REACHED
  qp->cg1_ready = false;  // settings may have changed between calls
  qp->persist_skip_solve = false;
  const auto wall0 = std::chrono::steady_clock::now();
  if (ctx->timing) SCP_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int cg_total = 0, it = 0;
  int pipes = 0;
  double rp = INFINITY, rd = INFINITY;
  int cad = st.check_termination;  // steps between two checks (settings.check_fine: shorter once the residuals are close)
  // (the fine cadence applies when it divides the coarse one; otherwise the cadence is fixed)
  // and to QPs with collision rows: QP#0 keeps the fixed cadence (its 20 surplus steps are cheap, and a better converged
  // starting point saves the first joint QP of large problems far more: 250 instead of 400 steps at 1024 x 50)
  // ... and up to SCP_FINE_MAX_COLUMNS columns: beyond, it measured slower (4096 x 50: 195 instead of 200 steps, but 0.27 ms
  // more in every repetition; profiles/r03_check_cadence.txt, r03_check_cost.txt)
  constexpr int64_t SCP_FINE_MAX_COLUMNS = 4096;
  const int fine = (st.check_fine > 0 && st.check_fine < st.check_termination && st.check_termination % st.check_fine == 0 && qp->nW > 0 &&
                    qp->C <= SCP_FINE_MAX_COLUMNS) ? st.check_fine : 0;
  while (it < st.max_iter) {
    // fixed rows only (everything is column-local): every iteration up to the next termination check goes into ONE
    // launch; the persistent single-step kernel goes further and runs the checks itself, returning only when the host has
    // something to decide (solved, infeasible, iteration limit, a rho update)
    const bool qp0_it = fused && qp->nW == 0;
    // long horizons (K in 121..1024, e.g. the reference's demo K = 500): the same single-step pipeline with
    // cg1_colK_kernel as its column kernel; its termination check is the generic one (snapshot of the duals)
    const bool bigk = st.use_mfma == 1 && qp->K > SCP_FUSED_MAX_K && qp->K <= SCP_BIGK_MAX_K && qp->C <= SCP_PART_CAP / 2;
    const bool cg1_big = bigk && st.cg_iters == 1 && qp->nW > 0;
    const bool cg1_it = fused && st.cg_iters == 1 && qp->nW > 0;  // its update kernel emits delta-y itself
    bool persist_done = false;
    if (cg1_it && !qp->persist_skip_solve && scp_qp_persist_eligible(qp)) {
      int ran = 0, code = 0, it_done = it;
      QP_CHECK(scp_qp_cg1_persist(qp, it, cad, &ran, &code, &it_done));
      if (ran && code == SCP_PERSIST_GAVE_UP) {
        // its workgroups were not all resident at once (e.g. the device is shared with another process's kernels): nothing
        // was written back, so the solve goes on from the same state on the three-launch pipeline and stays there until
        // the next scp_qp_reset re-arms the persistent path
        qp->persist_off = true;
        qp->persist_epoch = 0;
        ++info->persist_gave_up;
      } else if (ran) {
        ++info->persist_launches;
        info->rho_switches_in_kernel += qp->persist_rho_switches;
        pipes |= 1 << (qp->persist_variant == 1 ? SCP_PIPE_PERSIST16 : (qp->persist_variant == 2 ? SCP_PIPE_PERSIST8L : SCP_PIPE_PERSIST));
        cg_total += it_done - it;
        qp->steps_since_reset += it_done - it;
        it = it_done;
        qp->qx_fresh = true;  // the kernel's last check left F x and S0 x exact
        persist_done = true;
        if (qp->persist_rho_switches > 0) {
          // adaptive-rho updates whose blocks were cached happened inside the kernel (same test, same values as below):
          // adopt the result; build_kkt finds the slot and points d.* at it
          qp->rho = qp->persist_rho;
          info->rho_updates += qp->persist_rho_switches;
          QP_CHECK(build_kkt(qp));
        }
      }
    }
    int n_it = 1;
    if (!persist_done && qp0_it) {
      n_it = cad - it % cad;
      if (it + n_it > st.max_iter) n_it = st.max_iter - it;
    }
    if (!persist_done) {
      it += n_it;
      qp->steps_since_reset += n_it;
    }
    const bool will_check = persist_done || it % cad == 0 || it >= st.max_iter;
    const bool with_dy = will_check && st.eps_prim_inf > 0.0;
    if (!persist_done) {
      if (with_dy && !cg1_it && !qp0_it) {  // snapshot of the duals: delta-y of this iteration feeds the certificate
        SCP_HIP_CHECK(ctx, hipMemcpyAsync(qp->d.dyf, qp->d.yf, (size_t)qp->Rf * qp->C * sizeof(double),
                                          hipMemcpyDeviceToDevice, ctx->stream));
        if (qp->nW > 0)
          SCP_HIP_CHECK(ctx, hipMemcpyAsync(qp->d.dyc, qp->d.yc, (size_t)qp->nW * sizeof(double),
                                            hipMemcpyDeviceToDevice, ctx->stream));
      }
      if (cg1_it) { QP_CHECK(scp_qp_cg1_iteration(qp, &cg_total, with_dy)); pipes |= 1 << SCP_PIPE_CG1; }
      else if (cg1_big) { QP_CHECK(scp_qp_cg1_iteration(qp, &cg_total, false)); pipes |= 1 << SCP_PIPE_CG1_BIGK; }
      else if (qp0_it) { QP_CHECK(scp_qp_qp0_iterations(qp, n_it, with_dy ? qp->d.dyf : nullptr)); pipes |= 1 << SCP_PIPE_QP0; }
      else if (fused) { QP_CHECK(scp_qp_fused_iteration(qp, &cg_total)); pipes |= 1 << SCP_PIPE_FUSED; }
      else { QP_CHECK(admm_iteration(qp, &cg_total)); pipes |= 1 << SCP_PIPE_GENERIC; }
    }
    if (will_check) {
      if (persist_done) {
        // (the kernel left the nine check results in h_scal)
      } else if (cg1_it || qp0_it) {
        QP_CHECK(scp_qp_fused_residuals(qp, with_dy));
      } else {
        QP_CHECK(residuals(qp, with_dy));
      }
      if (!cg1_it) { qp->cg1_ready = false; qp->gval_valid = false; qp->qx_fresh = false; }  // residuals() used G and the Q slabs as scratch (the fused check keeps
                                           // the pipeline's carried state and refreshes S0 x, F x exactly)
      const double* hs = qp->h_scal;
      rp = hs[SL_RP];
      rd = hs[SL_RD];
      const double np = fmax(hs[SL_NAX], hs[SL_NZ]);
      const double nd = fmax(hs[SL_NPX], hs[SL_NATY]);
      const double tol_p = st.eps_abs + st.eps_rel * np, tol_d = st.eps_abs + st.eps_rel * nd;
      if (rp <= tol_p && rd <= tol_d) {
        info->status_val = 1;
        break;
      }
      if (fine)  // close to the tolerances: look again soon (the persistent kernels take the same decision)
        cad = (rp < st.check_fine_ratio * tol_p && rd < st.check_fine_ratio * tol_d) ? fine : st.check_termination;
      // OSQP at max_iter: the same test with ten times the tolerances -> "solved inaccurate" (status 2), which the
      // reference accepts like "solved" (scp.py:363, :446)
      if (it >= st.max_iter && rp <= 10.0 * (st.eps_abs + st.eps_rel * np) && rd <= 10.0 * (st.eps_abs + st.eps_rel * nd))
        info->status_val = 2;
      if (with_dy) {  // OSQP's is_primal_infeasible on the unscaled problem
        const double ndy = hs[SL_NDY], supp = hs[SL_SUPP];
        if (ndy > st.eps_prim_inf && supp < -st.eps_prim_inf * ndy) {
          if (!cg1_it && !qp0_it) QP_CHECK(certificate_atdy(qp));  // the fused check has |A^T dy| already
          if (qp->h_scal[SL_NATDY] < st.eps_prim_inf * ndy) {
            info->status_val = -3;
            break;
          }
        }
      }
      if (st.adaptive_rho && st.adaptive_rho_interval > 0 && it % st.adaptive_rho_interval == 0) {
        const double prim = rp / fmax(np, 1e-10);
        const double dual = rd / fmax(nd, 1e-10);
        double nr = qp->rho * std::sqrt(prim / fmax(dual, 1e-10));
        nr = fmin(fmax(nr, 1e-6), 1e6);
        // snapped to a geometric grid (steps of 2^(1/4)): the estimate is a ratio of small residuals; without the
        // grid 1e-13 of fp noise (atomics order, MFMA vs scalar sums) becomes a percent-level difference in rho and
        // an iterate path that differs from the oracle's at the 1e-4 level although both are valid solutions
        nr = std::exp2(std::round(4.0 * std::log2(nr)) / 4.0);
        if (nr > qp->rho * st.adaptive_rho_tolerance || nr < qp->rho / st.adaptive_rho_tolerance) {
          qp->rho = nr;
          qp->cg1_ready = false;  // the carried row values depend on rho
          qp->gval_valid = false;
          QP_CHECK(build_kkt(qp));
          ++info->rho_updates;
          if (fine) cad = fine;  // (the residuals usually fall below the tolerances within a few steps)
        }
      }
    }
  }
  float ms = 0.f;
  if (ctx->timing) {
    SCP_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    SCP_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    SCP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  } else {  // (every exit of the loop above has read the solve's last check on the host: the device work is done)
    ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  }
  info->iter = it;
  info->cg_iters_total = cg_total;
  info->working_rows = qp->nW;
  info->r_prim = rp;
  info->r_dual = rd;
  info->rho = qp->rho;
  info->solve_ms = ms;
  info->pipeline = pipes;
  return SCP_OK;
}

extern "C" int scp_qp_clone_state(scp_qp* dst, const scp_qp* src) {
  if (!dst || !src) return SCP_ERR_INVALID;
  scp_ctx* ctx = dst->ctx;
  SCP_REQUIRE(ctx, dst->N == src->N && dst->K == src->K && dst->D == src->D && dst->h == src->h,
              "qp_clone_state: shapes differ");
  if (!src->reset_done) return scp_fail(ctx, SCP_ERR_STATE, "qp_clone_state: source has no state");
  if (src->nW > dst->row_cap)
    return scp_fail(ctx, SCP_ERR_CAPACITY, "qp_clone_state: %lld rows exceed the capacity %lld", (long long)src->nW,
                    (long long)dst->row_cap);
  hipStream_t s = ctx->stream;
  const size_t nf = (size_t)src->Rf * src->C * sizeof(double), nx = (size_t)src->K * src->C * sizeof(double);
  const size_t nw = (size_t)src->nW;
  const QpDev &a = src->d, &b = dst->d;
#define CP(field, bytes) SCP_HIP_CHECK(ctx, hipMemcpyAsync(b.field, a.field, (bytes), hipMemcpyDeviceToDevice, s))
  CP(lf, nf); CP(uf, nf); CP(zf, nf); CP(yf, nf); CP(x, nx);
  CP(states, (size_t)4 * src->C * sizeof(double));
  if (nw) {
    CP(w_row, nw * sizeof(int64_t)); CP(w_k, nw * sizeof(int)); CP(w_i, nw * sizeof(int)); CP(w_j, nw * sizeof(int));
    CP(w_eta, nw * src->D * sizeof(double)); CP(w_l, nw * sizeof(double)); CP(zc, nw * sizeof(double));
    CP(yc, nw * sizeof(double));
  }
#undef CP
  memcpy(dst->lim, src->lim, sizeof(dst->lim));
  memcpy(dst->space, src->space, sizeof(dst->space));
  dst->steps_since_reset = src->steps_since_reset;
  dst->nW = src->nW;
  dst->persist_cap_nW = -1;
  dst->rho = src->rho;
  dst->st = src->st;
  dst->problem_set = true;
  dst->cg1_ready = false;
  dst->gval_valid = false;
  dst->csr_valid = false;
  dst->qx_fresh = false;
  QP_CHECK(build_kkt(dst));
  dst->reset_done = true;
  SCP_HIP_CHECK(ctx, hipStreamSynchronize(s));  // src's workspace may be released by the caller right after
  return SCP_OK;
}

extern "C" int scp_qp_get_solution(scp_qp* qp, double* x_out) {
  if (!qp) return SCP_ERR_INVALID;
  SCP_REQUIRE(qp->ctx, x_out, "qp_get_solution: null pointer");
  if (!qp->reset_done) return scp_fail(qp->ctx, SCP_ERR_STATE, "qp_get_solution: no solve yet");
  return scp_launch_from_time_major(qp->ctx, qp->N, qp->K, qp->D, qp->d.x, x_out);
}

const double* scp_qp_solution_tm(const scp_qp* qp) { return qp->d.x; }

extern "C" int scp_qp_get_duals(scp_qp* qp, double* y_fixed, double* y_col) {
  if (!qp) return SCP_ERR_INVALID;
  scp_ctx* ctx = qp->ctx;
  if (!qp->reset_done) return scp_fail(ctx, SCP_ERR_STATE, "qp_get_duals: no solve yet");
  const int N = qp->N, K = qp->K, D = qp->D;
  const int64_t C = qp->C;
  if (y_fixed) {
    // blocks back to the reference stacking order (scp.py:342-358)
    QP_CHECK(scp_launch_from_time_major(ctx, N, K - 1, D, qp->d.yf, y_fixed));
    int64_t src = (int64_t)(K - 1) * C, dst = (int64_t)N * (K - 1) * D;
    for (int b = 0; b < 3; ++b) {
      QP_CHECK(scp_launch_from_time_major(ctx, N, K, D, qp->d.yf + src, y_fixed + dst));
      src += (int64_t)K * C;
      dst += (int64_t)N * K * D;
    }
  }
  if (y_col && qp->nW > 0)
    SCP_HIP_CHECK(ctx, hipMemcpyAsync(y_col, qp->d.yc, qp->nW * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return SCP_OK;
}

// test hook: copy one internal array of the solver to `out` (device pointer, capacity `cap` doubles); *n_out [host] = its
// length.  names: "fx" (carried F x, [4K-1][C]), "qx" (carried S0 x, [K][C]), "gval" (row value per incidence entry),
// "zf", "yf" ([4K-1][C]), "zc", "yc" (per working row), "x" ([K][C]).
extern "C" int scp_qp_peek(scp_qp* qp, const char* name, double* out, int64_t cap, int64_t* n_out) {
  if (!qp) return SCP_ERR_INVALID;
  scp_ctx* ctx = qp->ctx;
  SCP_REQUIRE(ctx, name && out && n_out, "qp_peek: null pointer");
  const QpDev& d = qp->d;
  const int64_t nf = (int64_t)qp->Rf * qp->C, nx = (int64_t)qp->K * qp->C;
  const double* src = nullptr;
  int64_t n = 0;
  if (!strcmp(name, "fx")) { src = d.fx; n = nf; }
  else if (!strcmp(name, "qx")) { src = qp->qx_sel ? d.HQ : d.HQ + nx; n = nx; }
  else if (!strcmp(name, "gval")) { src = d.gval; n = 2 * qp->nW; }
  else if (!strcmp(name, "zf")) { src = d.zf; n = nf; }
  else if (!strcmp(name, "yf")) { src = d.yf; n = nf; }
  else if (!strcmp(name, "zc")) { src = d.zc; n = qp->nW; }
  else if (!strcmp(name, "yc")) { src = d.yc; n = qp->nW; }
  else if (!strcmp(name, "x")) { src = d.x; n = nx; }
  else if (!strcmp(name, "w_eta")) { src = d.w_eta; n = qp->nW * qp->D; }
  else if (!strcmp(name, "w_l")) { src = d.w_l; n = qp->nW; }
  else if (!strcmp(name, "p")) { src = d.p; n = nx; }
  else if (!strcmp(name, "qp")) { src = d.hpf; n = nx; }
  else return scp_fail(ctx, SCP_ERR_INVALID, "qp_peek: unknown array %s", name);
  *n_out = n;
  if (n > cap) return scp_fail(ctx, SCP_ERR_CAPACITY, "qp_peek: %lld doubles needed", (long long)n);
  if (n > 0) SCP_HIP_CHECK(ctx, hipMemcpyAsync(out, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return SCP_OK;
}

// test hook: "persist_fault" = n makes the next n persistent launches wait for a workgroup that does not exist (exercises
// the give-up path); "persist_off" reads (value < 0) or sets whether the solver has fallen back to the three-launch
// pipeline.  Returns the value in effect, or SCP_ERR_INVALID.
extern "C" int scp_qp_debug_set(scp_qp* qp, const char* key, int value) {
  if (!qp || !key) return SCP_ERR_INVALID;
  if (!strcmp(key, "persist_fault")) {
    if (value >= 0) qp->persist_fault = value;
    return qp->persist_fault;
  }
  if (!strcmp(key, "persist_off")) {
    if (value >= 0) qp->persist_off = value != 0;
    return qp->persist_off ? 1 : 0;
  }
  if (!strcmp(key, "persist_gave_up_total")) return qp->persist_gave_up_total;  // (read only)
  return scp_fail(qp->ctx, SCP_ERR_INVALID, "qp_debug_set: unknown key %s", key);
}
