// Column-block kernels of the joint QP (fused path, K <= SCP_FUSED_MAX_K).
//
// Every operation of an ADMM step except the working-row gather/scatter is local to a column c = (agent, axis):
// the fixed rows, the K x K KKT block and the Toeplitz block S0 act along the time index only.
// One workgroup (16 waves) therefore owns 16 columns, keeps their K-vectors as [rows][16] tiles in LDS and chains
// whole sequences of products  tile_out = A . tile_in  (A one of F^T, [H_f; S0], S0^T, H_f^{-1}, F) on the fp64
// matrix cores -- v_mfma_f64_16x16x4_f64, one 16 x 16 output tile per wave and step, the A operand preloaded from
// L2 in chunks of 16 k-steps so that a chain costs one memory latency, not one per step.  Kernel boundaries remain
// only where the algorithm needs the whole grid: the two PCG inner products and the row gather/scatter.
// An ADMM step with n PCG steps is 4n + 3 launches, 7 at the default n = 1 (15n + ... on the generic path of scp_qp.hip, which stays as the fallback
// for K > 128 and as the use_mfma = 0/2 reference); the arithmetic is the same, statement by statement.
#include "scp_qp_device.h"
#include "scp_pair_device.h"
#include "scp_reset_device.h"
#include <cstdlib>

namespace {

using namespace scpdev;
constexpr int FT = 1024;       // threads per workgroup (16 waves: one 16-row output tile per wave in most products)
constexpr int NWV = FT / 64;

// Developer build (make prof -> libscp_hip_prof.so): wall-clock stamps (100 MHz) of one workgroup's phases.
#ifdef SCP_PHASE_PROFILE
__device__ unsigned long long scp_phase_clk[64];
#define PHASE_MARK(slot)                                                                      \
  do {                                                                                        \
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) scp_phase_clk[slot] = wall_clock64(); \
  } while (0)
#else
#define PHASE_MARK(slot) ((void)0)
#endif

// O[R][16] = (ACC ? O : 0) + A[R][M] . V[M][16];  A: global, PACKED in operand order (QpDev::pF ...: [row tile][k step]
// [lane], zero padded);  V, O: distinct LDS tiles.  The waves [w0, w0+nw) of the workgroup share the row tiles; other
// waves return at once, so independent products run side by side on disjoint wave sets.  One operand load is 512
// contiguous bytes per wave (row-major A cost 16 cache lines per load and kept the L1 busy for most of a phase).  The
// operands of the NEXT chunk of 16 k-steps are loaded while the MFMAs of the current chunk issue.
// Operand maps of v_mfma_f64_16x16x4_f64 as in scp_gemm.hip.  Caller synchronises before reading O.
template <bool ACC>
__device__ inline void wg_mm_range(const double* __restrict__ P, int R, int M, int kb, int ke, const double* V, double* O,
                                   int w0, int nw) {
  // O[R][16] (+)= A[R][kb:ke] . V[kb:ke][16]   (kb a multiple of 4; V is indexed by the absolute k)
  const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6) - w0;
  if (wave < 0 || wave >= nw) return;
  const int li = lane & 15, lk = lane >> 4;
  const int tiles = (R + 15) >> 4, nks = (M + 3) >> 2;
  const int ks0 = kb >> 2, ks1 = (ke + 3) >> 2;
  for (int t = wave; t < tiles; t += nw) {
    const int r0 = t * 16;
    const double* Ap = P + (size_t)t * nks * 64 + lane;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    if (ACC) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + lk + 4 * r;
        acc[r] = row < R ? O[row * CB + li] : 0.0;
      }
    }
    double a[16], an[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = Ap[(size_t)min(ks0 + s, ks1 - 1) * 64];  // clamped: straight-line loads
    for (int kc = ks0; kc < ks1; kc += 16) {
      const bool more = kc + 16 < ks1;  // wave-uniform
      if (more) {
#pragma unroll
        for (int s = 0; s < 16; ++s) an[s] = Ap[(size_t)min(kc + 16 + s, ks1 - 1) * 64];
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (kc + s < ks1) {  // wave-uniform
          const int kk = 4 * (kc + s) + lk;
          const double b = kk < ke ? V[kk * CB + li] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b, acc, 0, 0, 0);
        }
      }
      if (more) {
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = an[s];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + lk + 4 * r;
      if (row < R) O[row * CB + li] = acc[r];
    }
  }
}

template <bool ACC>
__device__ inline void wg_mm(const double* __restrict__ A, int R, int M, const double* V, double* O, int w0, int nw) {
  wg_mm_range<ACC>(A, R, M, 0, M, V, O, w0, nw);
}

// tile <-> global ([rows][C] slab, columns c0 .. c0+15); out-of-range columns read as 0 and are not written
__device__ inline void tile_load(double* T, const double* __restrict__ g, int rows, int64_t C, int64_t c0) {
  for (int e = threadIdx.x; e < rows * CB; e += FT) {
    const int r = e >> 4, c = e & 15;
    T[e] = (c0 + c < C) ? g[(int64_t)r * C + c0 + c] : 0.0;
  }
}
__device__ inline void tile_store(const double* T, double* __restrict__ g, int rows, int64_t C, int64_t c0) {
  for (int e = threadIdx.x; e < rows * CB; e += FT) {
    const int r = e >> 4, c = e & 15;
    if (c0 + c < C) g[(int64_t)r * C + c0 + c] = T[e];
  }
}
__device__ inline void tile_zero_global(double* __restrict__ g, int rows, int64_t C, int64_t c0) {
  for (int e = threadIdx.x; e < rows * CB; e += FT) {
    const int r = e >> 4, c = e & 15;
    if (c0 + c < C) g[(int64_t)r * C + c0 + c] = 0.0;
  }
}

// deterministic workgroup sum (fixed tree); result valid in every thread
__device__ inline double wg_sum(double v) {
  __shared__ double s[FT / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) t += s[w];
  __syncthreads();
  return t;
}

// sum of the per-workgroup partials of an inner product, same order in every workgroup
__device__ inline double sum_parts(const double* __restrict__ part, int n) {
  double v = 0.0;
  for (int b = threadIdx.x; b < n; b += FT) v += part[b];
  return wg_sum(v);
}

// ---- K_A: rhs and the fixed part of the PCG start ------------------------------------------------------
// W = rho w_r z_f - y_f ; rhsF = sigma x + F^T W ; [Hx ; Qx] = [H_f ; S0] x
// has_rows: r0 = rhsF - Hx -> rhs, Qx -> Q, xt = x, G = 0        (PCG follows)
// else    : xt = H_f^{-1} rhsF                                    (the x-update is exact)
__global__ __launch_bounds__(FT) void fused_pre_kernel(int K, int Rf, int64_t C, double rho, double sigma, int has_rows,
                                                        const double* __restrict__ Ft, const double* __restrict__ HS,
                                                        const double* __restrict__ Minv,
                                                        const double* __restrict__ wrow, const double* __restrict__ x,
                                                        const double* __restrict__ zf, const double* __restrict__ yf,
                                                        double* __restrict__ rhs, double* __restrict__ Q,
                                                        double* __restrict__ xt, double* __restrict__ G) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* X = lds;                  // [K][16]
  double* W = X + K * CB;           // [Rf][16]
  double* T1 = W + Rf * CB;         // [K][16]
  double* T2 = T1 + K * CB;         // [2K][16]
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  tile_load(X, x, K, C, c0);
  for (int e = threadIdx.x; e < Rf * CB; e += FT) {
    const int r = e >> 4, c = e & 15;
    const int64_t g = (int64_t)r * C + c0 + c;
    W[e] = (c0 + c < C) ? rho * wrow[r] * zf[g] - yf[g] : 0.0;
  }
  __syncthreads();
  if (has_rows) {  // two independent products side by side
    wg_mm<false>(Ft, K, Rf, W, T1, 0, 8);
    wg_mm<false>(HS, 2 * K, K, X, T2, 8, NWV - 8);
  } else {
    wg_mm<false>(Ft, K, Rf, W, T1, 0, NWV);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < K * CB; e += FT) T1[e] += sigma * X[e];  // rhsF
  __syncthreads();
  if (has_rows) {
    for (int e = threadIdx.x; e < K * CB; e += FT) {
      const int r = e >> 4, c = e & 15;
      if (c0 + c < C) {
        const int64_t g = (int64_t)r * C + c0 + c;
        rhs[g] = T1[e] - T2[e];
        Q[g] = T2[K * CB + e];
        xt[g] = X[e];
        G[g] = 0.0;
      }
    }
  } else {
    wg_mm<false>(Minv, K, K, T1, T2, 0, NWV);
    __syncthreads();
    tile_store(T2, xt, K, C, c0);
  }
}

// ---- K_B / K_D: working rows, G = A_W^T g: rows_value_kernel + csr_gather_kernel (scp_qp_rows_gather below) -- a gather
// over the sorted incidence lists, so the sums have a fixed order (the round-1 version scattered with atomics)
// ---- K_C: PCG start ---------------------------------------------------------------------------------------
// r = r0 + S0^T G ; zz = Minv r ; p = zz ; part[b] = r.zz ; [HpF ; Qp] = [H_f ; S0] p ; G = 0
__global__ __launch_bounds__(FT) void fused_cg_init_kernel(int K, int64_t C, const double* __restrict__ S0t,
                                                            const double* __restrict__ Minv,
                                                            const double* __restrict__ HS, const double* __restrict__ r0,
                                                            double* __restrict__ G, double* __restrict__ r,
                                                            double* __restrict__ p, double* __restrict__ hpf,
                                                            double* __restrict__ Q, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Gt = lds;                 // [K][16]
  double* R = Gt + K * CB;          // [K][16]
  double* Z = R + K * CB;           // [K][16]
  double* T2 = Z + K * CB;          // [2K][16]
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  tile_load(Gt, G, K, C, c0);
  tile_load(R, r0, K, C, c0);
  __syncthreads();
  tile_zero_global(G, K, C, c0);
  wg_mm<true>(S0t, K, K, Gt, R, 0, NWV);
  __syncthreads();
  wg_mm<false>(Minv, K, K, R, Z, 0, NWV);
  __syncthreads();
  wg_mm<false>(HS, 2 * K, K, Z, T2, 0, NWV);
  double dot = 0.0;
  for (int e = threadIdx.x; e < K * CB; e += FT) dot += R[e] * Z[e];
  dot = wg_sum(dot);  // (barrier inside: T2 complete afterwards)
  if (threadIdx.x == 0) part[blockIdx.x] = dot;
  for (int e = threadIdx.x; e < K * CB; e += FT) {
    const int rr = e >> 4, c = e & 15;
    if (c0 + c < C) {
      const int64_t g = (int64_t)rr * C + c0 + c;
      r[g] = R[e];
      p[g] = Z[e];
      hpf[g] = T2[e];
      Q[g] = T2[K * CB + e];
    }
  }
}

// ---- K_E: Hp = HpF + S0^T G ; part[b] = p.Hp ; G = 0 -----------------------------------------------------
__global__ __launch_bounds__(FT) void fused_cg_hp_kernel(int K, int64_t C, const double* __restrict__ S0t,
                                                          double* __restrict__ G, const double* __restrict__ hpf,
                                                          const double* __restrict__ p, double* __restrict__ Hp,
                                                          double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Gt = lds;                 // [K][16]
  double* H = Gt + K * CB;          // [K][16]
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  tile_load(Gt, G, K, C, c0);
  tile_load(H, hpf, K, C, c0);
  __syncthreads();
  tile_zero_global(G, K, C, c0);
  wg_mm<true>(S0t, K, K, Gt, H, 0, NWV);
  __syncthreads();
  double dot = 0.0;
  for (int e = threadIdx.x; e < K * CB; e += FT) {
    const int rr = e >> 4, c = e & 15;
    if (c0 + c < C) {
      const int64_t g = (int64_t)rr * C + c0 + c;
      Hp[g] = H[e];
      dot += p[g] * H[e];
    }
  }
  dot = wg_sum(dot);
  if (threadIdx.x == 0) part[blockIdx.x] = dot;
}

// ---- K_F: alpha = rz / pHp ; xt += alpha p ; r -= alpha Hp ; zz = Minv r ; part[b] = r.zz -------------------
// rz: first = 1 -> sum(part_rz) (the PCG start wrote partials), else scal[slot]
__global__ __launch_bounds__(FT) void fused_cg_step_kernel(int K, int64_t C, int nblk, int first, int slot,
                                                            const double* __restrict__ Minv, double* __restrict__ scal,
                                                            const double* __restrict__ part_rz,
                                                            const double* __restrict__ part_php,
                                                            const double* __restrict__ p, const double* __restrict__ Hp,
                                                            double* __restrict__ xt, double* __restrict__ r,
                                                            double* __restrict__ zz, double* __restrict__ part_new) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* R = lds;                  // [K][16]
  double* Z = R + K * CB;           // [K][16]
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  const double rz = first ? sum_parts(part_rz, nblk) : scal[slot];
  const double pHp = sum_parts(part_php, nblk);
  const double alpha = (pHp > 0.0 && rz != 0.0) ? rz / pHp : 0.0;
  if (first && blockIdx.x == 0 && threadIdx.x == 0) scal[slot] = rz;
  for (int e = threadIdx.x; e < K * CB; e += FT) {
    const int rr = e >> 4, c = e & 15;
    double v = 0.0;
    if (c0 + c < C) {
      const int64_t g = (int64_t)rr * C + c0 + c;
      xt[g] += alpha * p[g];
      v = r[g] - alpha * Hp[g];
      r[g] = v;
    }
    R[e] = v;
  }
  __syncthreads();
  wg_mm<false>(Minv, K, K, R, Z, 0, NWV);
  __syncthreads();
  double dot = 0.0;
  for (int e = threadIdx.x; e < K * CB; e += FT) dot += R[e] * Z[e];
  dot = wg_sum(dot);
  if (threadIdx.x == 0) part_new[blockIdx.x] = dot;
  tile_store(Z, zz, K, C, c0);
}

// ---- K_G: beta = rz_new / rz ; p = zz + beta p ; [HpF ; Qp] = [H_f ; S0] p ; scal[slot^1] = rz_new ------------
__global__ __launch_bounds__(FT) void fused_cg_dir_kernel(int K, int64_t C, int nblk, int slot,
                                                           const double* __restrict__ HS, double* __restrict__ scal,
                                                           const double* __restrict__ part_new,
                                                           const double* __restrict__ zz, double* __restrict__ p,
                                                           double* __restrict__ hpf, double* __restrict__ Q) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* P = lds;                  // [K][16]
  double* T2 = P + K * CB;          // [2K][16]
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  const double rz = scal[slot];
  const double rz_new = sum_parts(part_new, nblk);
  const double beta = rz != 0.0 ? rz_new / rz : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[slot ^ 1] = rz_new;
  for (int e = threadIdx.x; e < K * CB; e += FT) {
    const int rr = e >> 4, c = e & 15;
    double v = 0.0;
    if (c0 + c < C) {
      const int64_t g = (int64_t)rr * C + c0 + c;
      v = zz[g] + beta * p[g];
      p[g] = v;
    }
    P[e] = v;
  }
  __syncthreads();
  wg_mm<false>(HS, 2 * K, K, P, T2, 0, NWV);
  __syncthreads();
  for (int e = threadIdx.x; e < K * CB; e += FT) {
    const int rr = e >> 4, c = e & 15;
    if (c0 + c < C) {
      const int64_t g = (int64_t)rr * C + c0 + c;
      hpf[g] = T2[e];
      Q[g] = T2[K * CB + e];
    }
  }
}

// ---- K_U: z~ = F x~ ; relaxation, projection, duals of the fixed rows ; x = alpha x~ + (1-alpha) x ; Q = S0 x~ ---
// fold = 1: the LAST PCG step is applied here, x~ = xt + a p with a = rz / pHp (its residual update and the
// preconditioner product would be dead work), saving one launch per ADMM step.
__global__ __launch_bounds__(FT) void fused_post_kernel(int K, int Rf, int64_t C, double rho, double alpha, int has_rows,
                                                         int fold, int nblk, int first, int slot,
                                                         const double* __restrict__ scal,
                                                         const double* __restrict__ part_rz,
                                                         const double* __restrict__ part_php,
                                                         const double* __restrict__ pdir,
                                                         const double* __restrict__ F, const double* __restrict__ S0,
                                                         const double* __restrict__ wrow, const double* __restrict__ xt,
                                                         const double* __restrict__ lf, const double* __restrict__ uf,
                                                         double* __restrict__ zf, double* __restrict__ yf,
                                                         double* __restrict__ x, double* __restrict__ Q) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* X = lds;                  // [K][16]
  double* T = X + K * CB;           // [Rf][16]
  double* Qt = T + Rf * CB;         // [K][16]
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  if (fold) {
    const double rz = first ? sum_parts(part_rz, nblk) : scal[slot];
    const double pHp = sum_parts(part_php, nblk);
    const double a = (pHp > 0.0 && rz != 0.0) ? rz / pHp : 0.0;
    for (int e = threadIdx.x; e < K * CB; e += FT) {
      const int r = e >> 4, c = e & 15;
      const int64_t g = (int64_t)r * C + c0 + c;
      X[e] = (c0 + c < C) ? xt[g] + a * pdir[g] : 0.0;
    }
  } else {
    tile_load(X, xt, K, C, c0);
  }
  __syncthreads();
  if (has_rows) {
    wg_mm<false>(F, Rf, K, X, T, 0, NWV - 3);
    wg_mm<false>(S0, K, K, X, Qt, NWV - 3, 3);
  } else {
    wg_mm<false>(F, Rf, K, X, T, 0, NWV);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < Rf * CB; e += FT) {
    const int r = e >> 4, c = e & 15;
    if (c0 + c < C) {
      const int64_t g = (int64_t)r * C + c0 + c;
      const double rr = rho * wrow[r];
      const double zh = alpha * T[e] + (1.0 - alpha) * zf[g];
      const double y = yf[g];
      const double zn = fmin(fmax(zh + y / rr, lf[g]), uf[g]);
      yf[g] = y + rr * (zh - zn);
      zf[g] = zn;
    }
  }
  for (int e = threadIdx.x; e < K * CB; e += FT) {
    const int r = e >> 4, c = e & 15;
    if (c0 + c < C) {
      const int64_t g = (int64_t)r * C + c0 + c;
      x[g] = alpha * X[e] + (1.0 - alpha) * x[g];
      if (has_rows) Q[g] = Qt[e];
    }
  }
}

// ---- K_V: collision rows: relaxation, projection (u = +inf), duals ------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void fused_row_update_kernel(int64_t nW, int64_t C, double rho, double alpha,
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

// =====================================================================================================
// Single-PCG-step pipeline (settings.cg_iters == 1, the default): 3 launches per ADMM step.
//
//   x~ = x + a p,  p = Minv r,  r = rhs - H x,  a = (r.p) / (p.H p),   p.H p = r.p + rho sum_rows (eta.dQp)^2
//
// (H_f p = r), so neither H p nor a second scatter is needed.  The slabs Qx = S0 x and Fx = F x are carried:
// x+ = x + alpha a p gives Qx+ = Qx + alpha a Qp, Fx+ = Fx + alpha a Fp (refreshed exactly at every termination
// check), which turns r into -2 x + F^T(rho w (z_f - Fx) - y_f) + S0^T G and everything after a into elementwise work:
//   cg1_col_kernel    : r (wave scans), p = Minv r (MFMA), Qp = S0 p, Fp = F p (wave scans), partials r.p
//   cg1_rows_sq_kernel: partials rho (eta.dQp)^2
//   cg1_update_kernel : a; fixed rows' z, y, Fx+; x+, Qx+ (other buffer); collision rows' z, y and the row values
//                       g = rho zc - yc - rho eta.dQx+ of the next right-hand side (gathered by cg1_col_kernel)
// =====================================================================================================
constexpr int SQ_BLOCKS = 128;
// Column kernel of the single-step pipeline (launch 1 of 3).  With F x carried like S0 x,
//   r = sigma x + A^T(rho z - y) - H x = -2 x + F^T W' + S0^T G,   W' = rho w (z_f - F x) - y_f,
// (H_f = (2 + sigma) I + rho F^T w F), then p = H_f^{-1} r, S0 p, F p.  F = [J ; I ; V ; S] and S0 are the jerk
// stencil and the first / second cumulative sums of the integrator (scp.py:10-28, :489-491), so
//   F^T W' + S0^T G = J^T w_j + w_a + rsum(h w_v + h^2/2 (w_p - g)) + h^2/2 g + h^2 rsum_excl(rsum(w_p + g))
//   V p = h csum(p),  S p = h^2 (csum_excl(csum p) + csum(p)/2),  S0 p = h^2 (csum_excl(csum p) - csum(p)[k-1]/2)
// cost one wave-wide scan each instead of a dense product: the only matrix left is H_f^{-1} (the KKT solve, MFMA).
// Streaming the dense blocks (220 KB per workgroup for 16 columns) from L2 was 7 us of this kernel's 16.
// One wave per column for the scans (E time steps per lane), waves [0, ceil(K/16)) for the MFMA tiles.
template <int E>
__global__ __launch_bounds__(FT) void cg1_col_kernel(int K, int Rf, int64_t C, double rho, double h,
                                                      const double* __restrict__ pMinv, const double* __restrict__ wrow,
                                                      const double* __restrict__ x, const double* __restrict__ Fx,
                                                      const double* __restrict__ zf, const double* __restrict__ yf, int N,
                                                      int D, const int* __restrict__ cell_ptr,
                                                      const double* __restrict__ coef, const double* __restrict__ gval,
                                                      double* __restrict__ p, double* __restrict__ Qp,
                                                      double* __restrict__ Fp, double* __restrict__ part_rz) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double s_rz[NWV];
  const int RSF = pad_col(Rf), RSK = pad_col(K);
  double* Wt = lds;              // [16][RSF]  W', later F p
  double* Gx = Wt + CB * RSF;    // [16][RSK]  gather of eta * g
  double* Xt = Gx + CB * RSK;    // [16][RSK]  x
  double* Rt = Xt + CB * RSK;    // [16][RSK]  r
  double* Pt = Rt + CB * RSK;    // [16][RSK]  p
  double* Qt = Pt + CB * RSK;    // [16][RSK]  S0 p
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tK = (K + 15) >> 4, nks = (K + 3) >> 2;
  const double hh = h * h;
  PHASE_MARK(0);
  // ---- prologue: every global load is issued before the first use --------------------------------------------
  constexpr int WU = 4;  // slab rows per thread and pass: 64 WU >= Rf up to K = 64, a second pass beyond
  const int c = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const bool cok = c0 + c < C;
  const int col = (int)(c0 + c), agent = col / D, dd = col - agent * D;
  // Issue order = return order: the incidence-list bounds first (the gather's second hop depends on them), then x and
  // the slab rows, then the H_f^{-1} operands; the dependent coef / gval loads follow while the rest is in flight.
  // (Measured: the prologue takes 4.1 us either way -- two round trips to data another XCD wrote in the last launch.)
  auto gather = [&](int k, double& xv) {  // incidence-list sum of one (time step, column); returns it, loads x
    int g0 = 0, g1 = 0;
    double acc = 0.0;
    xv = 0.0;
    if (cok && k < K) {
      g0 = cell_ptr[cell_of(k, agent, K)];
      g1 = cell_ptr[cell_of(k, agent, K) + 1];
      xv = x[(int64_t)k * C + c0 + c];
    }
    for (int t = g0; t < g1; ++t) acc += coef[(size_t)t * D + dd] * gval[t];
    return acc;
  };
  int g0 = 0, g1 = 0;
  double xv0 = 0.0;
  if (cok && kg < K) {
    g0 = cell_ptr[cell_of(kg, agent, K)];
    g1 = cell_ptr[cell_of(kg, agent, K) + 1];
    xv0 = x[(int64_t)kg * C + c0 + c];
  }
  double wz[WU], wf[WU], wy[WU];
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int r = kg + u * (FT / CB);
    wz[u] = wf[u] = wy[u] = 0.0;
    if (cok && r < Rf) {
      const int64_t g = (int64_t)r * C + c0 + c;
      wz[u] = zf[g];
      wf[u] = Fx[g];
      wy[u] = yf[g];
    }
  }
  double aM[CHB];
  tile_prefetch<CHB>(pMinv, nks, wave < tK ? wave : 0, 0, nks, aM);
  {
    double acc = 0.0;
    for (int t = g0; t < g1; ++t) acc += coef[(size_t)t * D + dd] * gval[t];
    if (kg < K) {
      Gx[c * RSK + kg] = acc;
      Xt[c * RSK + kg] = xv0;
    }
  }
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int r = kg + u * (FT / CB);
    if (r < Rf) Wt[c * RSF + r] = rho * wrow[r] * (wz[u] - wf[u]) - wy[u];
  }
  for (int k = kg + FT / CB; k < K; k += FT / CB) {  // K > 64
    double xv;
    const double acc = gather(k, xv);
    Gx[c * RSK + k] = acc;
    Xt[c * RSK + k] = xv;
  }
  for (int r = kg + WU * (FT / CB); r < Rf; r += FT / CB) {  // Rf > 256
    const int64_t g = (int64_t)r * C + c0 + c;
    Wt[c * RSF + r] = cok ? rho * wrow[r] * (zf[g] - Fx[g]) - yf[g] : 0.0;
  }
  __syncthreads();
  PHASE_MARK(1);
  // ---- r: one wave per column; index i = lane E + e runs over the time steps in DESCENDING order (k = KM - i), so the
  // reverse cumulative sums of the transposed integrator blocks are ascending scans over the lanes ----------------
  {
    const double* Wc = Wt + wave * RSF;
    const int KM = 64 * E - 1;
    double wj[E], wa[E], u1[E], u2[E], g[E], xk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = KM - (lane * E + e);
      const bool ok = k < K;
      wj[e] = k < K - 1 ? Wc[k] : 0.0;
      wa[e] = ok ? Wc[K - 1 + k] : 0.0;
      const double wv = ok ? Wc[2 * K - 1 + k] : 0.0;
      const double wp = ok ? Wc[3 * K - 1 + k] : 0.0;
      g[e] = ok ? Gx[wave * RSK + k] : 0.0;
      xk[e] = ok ? Xt[wave * RSK + k] : 0.0;
      u1[e] = h * wv + 0.5 * hh * (wp - g[e]);
      u2[e] = wp + g[e];
    }
    double d1[E], d2[E], s1[E], s2[E], wjp[E];
    wave_scan<E>(u1, d1, s1);   // d1 = rsum(u1)
    wave_scan<E>(u2, s1, s2);   // s1 = rsum(u2)
    wave_scan<E>(s1, s2, d2);   // d2 = rsum_excl(rsum(u2))
    wave_next<E>(wj, wjp);      // w_j[k - 1]
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = KM - (lane * E + e);
      const double rk = (((wjp[e] - wj[e]) / h + wa[e]) + (d1[e] + 0.5 * hh * g[e]) + hh * d2[e]) - 2.0 * xk[e];
      if (k < K) Rt[wave * RSK + k] = rk;
    }
  }
  __syncthreads();
  PHASE_MARK(2);
  // ---- p = H_f^{-1} r: one 16-row tile per wave -----------------------------------------------------------------
  for (int t = wave; t < tK; t += NWV) {
    const int li = lane & 15, lk = lane >> 4;
    const double* Ap = pMinv + (size_t)t * nks * 64 + lane;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int kc = 0; kc < nks; kc += CHB) {
      if (kc > 0 || t != wave) {
#pragma unroll
        for (int s = 0; s < CHB; ++s) aM[s] = Ap[(size_t)min(kc + s, nks - 1) * 64];
      }
#pragma unroll
      for (int s = 0; s < CHB; ++s) {
        if (kc + s < nks) {  // wave-uniform
          const int kk = 4 * (kc + s) + lk;
          const double b = kk < K ? Rt[li * RSK + kk] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[s], b, acc, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = t * 16 + lk + 4 * r;
      if (row < K) Pt[li * RSK + row] = acc[r];
    }
  }
  __syncthreads();
  PHASE_MARK(3);
  // ---- S0 p, F p, r.p: one wave per column ------------------------------------------------------------------------
  {
    double* Wc = Wt + wave * RSF;
    double pk[E], c1[E], c2[E], t1[E], t2[E], c1p[E], pn[E];
    double rz = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = lane * E + e;
      pk[e] = k < K ? Pt[wave * RSK + k] : 0.0;
      rz += (k < K ? Rt[wave * RSK + k] : 0.0) * pk[e];
    }
    wave_scan<E>(pk, c1, t1);   // c1 = csum(p)
    wave_scan<E>(c1, t2, c2);   // c2 = csum_excl(csum(p))
    wave_prev<E>(c1, c1p);
    wave_next<E>(pk, pn);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = lane * E + e;
      if (k < K) {
        Qt[wave * RSK + k] = hh * (c2[e] - 0.5 * c1p[e]);
        if (k < K - 1) Wc[k] = (pn[e] - pk[e]) / h;
        Wc[K - 1 + k] = pk[e];
        Wc[2 * K - 1 + k] = h * c1[e];
        Wc[3 * K - 1 + k] = hh * (c2[e] + 0.5 * c1[e]);
      }
    }
    rz = wave_incl_sum(rz);
    if (lane == 63) s_rz[wave] = rz;
  }
  __syncthreads();
  PHASE_MARK(4);
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) t += s_rz[w];
    part_rz[blockIdx.x] = t;
  }
  // ---- coalesced stores ----------------------------------------------------------------------------------------------
  if (cok) {
    for (int k = kg; k < K; k += FT / CB) {
      const int64_t g = (int64_t)k * C + c0 + c;
      p[g] = Pt[c * RSK + k];
      Qp[g] = Qt[c * RSK + k];
    }
    for (int r = kg; r < Rf; r += FT / CB) Fp[(int64_t)r * C + c0 + c] = Wt[c * RSF + r];
  }
  PHASE_MARK(5);
}

// ---- the column kernel for long horizons (K > SCP_FUSED_MAX_K, up to 1024: the reference's demo runs K = 500) ------------
// Same quantities as cg1_col_kernel -- r by reverse cumulative sums, p = H_f^{-1} r, S0 p, F p by forward sums, r.p -- with
// ONE WORKGROUP PER COLUMN and one thread per time step: the sums are block-wide scans (wave scans on DPP + one LDS
// round for the wave totals), and H_f^{-1} (2 MB at K = 500, symmetric: thread k reads column k = row k, coalesced over
// the threads) is streamed from L2 by every column's workgroup instead of living in MFMA operand registers.  No LDS tiles
// of 4K-1 rows x 16 columns, so K is bounded by the thread count only.
__device__ inline double block_scan_incl(double v, bool reverse, double* wtot) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  const double w = reverse ? wave_incl_rsum(v) : wave_incl_sum(v);
  if (lane == (reverse ? 0 : 63)) wtot[wave] = w;
  __syncthreads();
  double off = 0.0;
  if (reverse) {
    for (int q = nw - 1; q > wave; --q) off += wtot[q];
  } else {
    for (int q = 0; q < wave; ++q) off += wtot[q];
  }
  __syncthreads();
  return w + off;
}
// value of thread k + shift (0 outside [0, K)) through an LDS line
__device__ inline double block_shift(double v, int shift, int K, double* line) {
  const int k = threadIdx.x;
  if (k < K) line[k] = v;
  __syncthreads();
  const int src = k + shift;
  const double out = (k < K && src >= 0 && src < K) ? line[src] : 0.0;
  __syncthreads();
  return out;
}

__global__ __launch_bounds__(1024) void cg1_colK_kernel(int K, int Rf, int64_t C, double rho, double h,
                                                         const double* __restrict__ Minv, const double* __restrict__ wrow,
                                                         const double* __restrict__ x, const double* __restrict__ Fx,
                                                         const double* __restrict__ zf, const double* __restrict__ yf, int N,
                                                         int D, const int* __restrict__ cell_ptr,
                                                         const double* __restrict__ coef, const double* __restrict__ gval,
                                                         double* __restrict__ p, double* __restrict__ Qp,
                                                         double* __restrict__ Fp, double* __restrict__ part_rz) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* rbuf = lds;        // [K] r
  double* line = rbuf + K;   // [K] neighbour shifts
  __shared__ double wtot[16];
  const int col = blockIdx.x, k = threadIdx.x;
  const bool live = k < K;
  const int agent = col / D, dd = col - agent * D;
  const double hh = h * h;
  double wj = 0.0, wa = 0.0, wv = 0.0, wp = 0.0, xk = 0.0, g = 0.0;
  if (live) {
    auto wprime = [&](int row) {
      const int64_t o = (int64_t)row * C + col;
      return rho * wrow[row] * (zf[o] - Fx[o]) - yf[o];
    };
    if (k < K - 1) wj = wprime(k);
    wa = wprime(K - 1 + k);
    wv = wprime(2 * K - 1 + k);
    wp = wprime(3 * K - 1 + k);
    xk = x[(int64_t)k * C + col];
    const int c0 = cell_ptr[cell_of(k, agent, K)], c1 = cell_ptr[cell_of(k, agent, K) + 1];
    for (int t = c0; t < c1; ++t) g += coef[(size_t)t * D + dd] * gval[t];
  }
  const double u1 = h * wv + 0.5 * hh * (wp - g);
  const double u2 = wp + g;
  const double d1 = block_scan_incl(u1, true, wtot);
  const double s1 = block_scan_incl(u2, true, wtot);
  const double d2 = block_shift(block_scan_incl(s1, true, wtot), +1, K, line);  // exclusive suffix sum
  const double wjm = block_shift(wj, -1, K, line);                               // w_j[k - 1]
  const double rk = (((wjm - wj) / h + wa) + (d1 + 0.5 * hh * g) + hh * d2) - 2.0 * xk;
  if (live) rbuf[k] = rk;
  __syncthreads();
  double pk = 0.0;
  if (live) {  // p_k = sum_m Minv[m][k] r_m  (Minv symmetric: column k read as the k-th entries of its rows -> coalesced)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int m = 0;
    for (; m + 4 <= K; m += 4) {
      a0 = fma(Minv[(size_t)m * K + k], rbuf[m], a0);
      a1 = fma(Minv[(size_t)(m + 1) * K + k], rbuf[m + 1], a1);
      a2 = fma(Minv[(size_t)(m + 2) * K + k], rbuf[m + 2], a2);
      a3 = fma(Minv[(size_t)(m + 3) * K + k], rbuf[m + 3], a3);
    }
    for (; m < K; ++m) a0 = fma(Minv[(size_t)m * K + k], rbuf[m], a0);
    pk = (a0 + a1) + (a2 + a3);
  }
  const double c1s = block_scan_incl(pk, false, wtot);                                // csum(p)
  const double c2s = block_shift(block_scan_incl(c1s, false, wtot), -1, K, line);     // csum_excl(csum(p))
  const double c1p = block_shift(c1s, -1, K, line);
  const double pn = block_shift(pk, +1, K, line);
  const double rz = block_scan_incl(live ? rk * pk : 0.0, false, wtot);  // total in the last thread
  if (live) {
    p[(int64_t)k * C + col] = pk;
    Qp[(int64_t)k * C + col] = hh * (c2s - 0.5 * c1p);
    if (k < K - 1) Fp[(int64_t)k * C + col] = (pn - pk) / h;
    Fp[(int64_t)(K - 1 + k) * C + col] = pk;
    Fp[(int64_t)(2 * K - 1 + k) * C + col] = h * c1s;
    Fp[(int64_t)(3 * K - 1 + k) * C + col] = hh * (c2s + 0.5 * c1s);
  }
  if (threadIdx.x == blockDim.x - 1) part_rz[col] = rz;
}

template <int D>
__global__ __launch_bounds__(256) void cg1_rows_sq_kernel(int64_t nW, int64_t C, double rho, const int* __restrict__ wk,
                                                           const int* __restrict__ wi, const int* __restrict__ wj,
                                                           const double* __restrict__ weta,
                                                           const double* __restrict__ Qp, double* __restrict__ part_sq) {
  __shared__ double sw[4];
  double acc = 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nW; n += (int64_t)SQ_BLOCKS * 256) {
    const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
    const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
    double ax = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) ax += weta[n * D + d] * (Qp[bi + d] - Qp[bj + d]);
    acc += ax * ax;
  }
  acc = wave_incl_sum(acc);
  if ((threadIdx.x & 63) == 63) sw[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part_sq[blockIdx.x] = rho * ((sw[0] + sw[1]) + (sw[2] + sw[3]));
}

// Fixed rows only (QP#0, or any solve before the first collision row joins): the whole ADMM iteration is column-local,
//   x~ = H_f^{-1} (sigma x + F^T(rho w z - y)),  z~ = F x~,  relaxation, projection, dual update,
// so ONE launch runs `nit` iterations (up to the next termination check) with z, y, l, u of a thread's slab rows in
// registers, x in LDS and the H_f^{-1} operands of a wave's tile resident in registers (K <= 64): four barriers and no
// global memory traffic per iteration, instead of two launches.
template <int E>
__global__ __launch_bounds__(FT) void qp0_col_kernel(int K, int Rf, int64_t C, double rho, double sigma, double alpha,
                                                      double h, int nit, const double* __restrict__ pMinv,
                                                      const double* __restrict__ wrow, const double* __restrict__ lf,
                                                      const double* __restrict__ uf, double* __restrict__ x,
                                                      double* __restrict__ zf, double* __restrict__ yf,
                                                      double* __restrict__ dy_out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int RU = 4 * E;  // slab rows per thread: 64 RU >= Rf = 4K - 1 for K <= 64 E
  const int RSF = pad_col(Rf), RSK = pad_col(K);
  double* Wt = lds;              // [16][RSF]  rho w z - y, then F x~
  double* Xt = Wt + CB * RSF;    // [16][RSK]  x
  double* Rt = Xt + CB * RSK;    // [16][RSK]  right-hand side
  double* Pt = Rt + CB * RSK;    // [16][RSK]  x~
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const bool cok = c0 + c < C;
  const int tK = (K + 15) >> 4, nks = (K + 3) >> 2;
  const double hh = h * h;
  double aM[CHB];
  tile_prefetch<CHB>(pMinv, nks, wave < tK ? wave : 0, 0, nks, aM);
  double z[RU], y[RU], lo[RU], hi[RU], rw[RU];
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const int r = kg + u * (FT / CB);
    z[u] = y[u] = lo[u] = hi[u] = 0.0;
    rw[u] = 1.0;
    if (cok && r < Rf) {
      const int64_t g = (int64_t)r * C + c0 + c;
      z[u] = zf[g]; y[u] = yf[g]; lo[u] = lf[g]; hi[u] = uf[g];
      rw[u] = rho * wrow[r];
    }
  }
  for (int k = kg; k < K; k += FT / CB) Xt[c * RSK + k] = cok ? x[(int64_t)k * C + c0 + c] : 0.0;
  for (int it = 0; it < nit; ++it) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = kg + u * (FT / CB);
      if (r < Rf) Wt[c * RSF + r] = rw[u] * z[u] - y[u];
    }
    __syncthreads();
    {  // right-hand side: time steps in descending order (reverse cumulative sums as ascending scans)
      const double* Wc = Wt + wave * RSF;
      const int KM = 64 * E - 1;
      double wj[E], wa[E], u1[E], u2[E], xk[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int k = KM - (lane * E + e);
        const bool ok = k < K;
        wj[e] = k < K - 1 ? Wc[k] : 0.0;
        wa[e] = ok ? Wc[K - 1 + k] : 0.0;
        const double wv = ok ? Wc[2 * K - 1 + k] : 0.0;
        u2[e] = ok ? Wc[3 * K - 1 + k] : 0.0;
        xk[e] = ok ? Xt[wave * RSK + k] : 0.0;
        u1[e] = h * wv + 0.5 * hh * u2[e];
      }
      double d1[E], d2[E], s1[E], s2[E], wjp[E];
      wave_scan<E>(u1, d1, s1);
      wave_scan<E>(u2, s1, s2);
      wave_scan<E>(s1, s2, d2);
      wave_next<E>(wj, wjp);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int k = KM - (lane * E + e);
        if (k < K) Rt[wave * RSK + k] = (((wjp[e] - wj[e]) / h + wa[e]) + d1[e] + hh * d2[e]) + sigma * xk[e];
      }
    }
    __syncthreads();
    for (int t = wave; t < tK; t += NWV) {  // x~ = H_f^{-1} rhs
      const int li = lane & 15, lk = lane >> 4;
      const double* Ap = pMinv + (size_t)t * nks * 64 + lane;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int kc = 0; kc < nks; kc += CHB) {
        if (nks > CHB || t != wave) {  // operands not resident (K > 64)
#pragma unroll
          for (int s_ = 0; s_ < CHB; ++s_) aM[s_] = Ap[(size_t)min(kc + s_, nks - 1) * 64];
        }
#pragma unroll
        for (int s_ = 0; s_ < CHB; ++s_) {
          if (kc + s_ < nks) {  // wave-uniform
            const int kk = 4 * (kc + s_) + lk;
            const double b = kk < K ? Rt[li * RSK + kk] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[s_], b, acc, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = t * 16 + lk + 4 * r;
        if (row < K) Pt[li * RSK + row] = acc[r];
      }
    }
    __syncthreads();
    {  // z~ = F x~ (forward scans), x+ = alpha x~ + (1 - alpha) x
      double* Wc = Wt + wave * RSF;
      double pk[E], c1[E], c2[E], t1[E], t2[E], pn[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        pk[e] = k < K ? Pt[wave * RSK + k] : 0.0;
      }
      wave_scan<E>(pk, c1, t1);
      wave_scan<E>(c1, t2, c2);
      wave_next<E>(pk, pn);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        if (k < K) {
          if (k < K - 1) Wc[k] = (pn[e] - pk[e]) / h;
          Wc[K - 1 + k] = pk[e];
          Wc[2 * K - 1 + k] = h * c1[e];
          Wc[3 * K - 1 + k] = hh * (c2[e] + 0.5 * c1[e]);
          Xt[wave * RSK + k] = alpha * pk[e] + (1.0 - alpha) * Xt[wave * RSK + k];
        }
      }
    }
    __syncthreads();
    const bool emit = dy_out != nullptr && it == nit - 1;  // delta-y of the last iteration (infeasibility certificate)
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = kg + u * (FT / CB);
      if (cok && r < Rf) {
        const double zh = alpha * Wt[c * RSF + r] + (1.0 - alpha) * z[u];
        const double zn = fmin(fmax(zh + y[u] / rw[u], lo[u]), hi[u]);
        const double yn = y[u] + rw[u] * (zh - zn);
        if (emit) dy_out[(int64_t)r * C + c0 + c] = yn - y[u];
        y[u] = yn;
        z[u] = zn;
      }
    }
  }
  __syncthreads();
  if (cok) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = kg + u * (FT / CB);
      if (r < Rf) {
        const int64_t g = (int64_t)r * C + c0 + c;
        zf[g] = z[u];
        yf[g] = y[u];
      }
    }
    for (int k = kg; k < K; k += FT / CB) x[(int64_t)k * C + c0 + c] = Xt[c * RSK + k];
  }
}

// step length a = r.p / p.H p  from the per-workgroup partials (p.H_f p = p.r because p = H_f^{-1} r); the same
// instruction sequence in every 256-thread workgroup, so every workgroup holds the same bits
__device__ inline double step_length_256(const double* __restrict__ part_rz, int nblk, const double* __restrict__ part_sq) {
  __shared__ double sw[2][4];
  double v = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) v += part_rz[b];
  for (int b = threadIdx.x; b < SQ_BLOCKS; b += 256) q += part_sq[b];
  v = wave_incl_sum(v);  // DPP: totals in lane 63 (an LDS-routed __shfl butterfly costs ten times as much)
  q = wave_incl_sum(q);
  if ((threadIdx.x & 63) == 63) {
    sw[0][threadIdx.x >> 6] = v;
    sw[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  const double rz = (sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]);
  const double pHp = rz + ((sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]));
  return (pHp > 0.0 && rz != 0.0) ? rz / pHp : 0.0;
}

// The same step length with p.Hp's row term summed by the caller itself (every workgroup redundantly, in the same order
// -> the same bits): for small working sets (nW <= SQ_INLINE_MAX) this saves the cg1_rows_sq_kernel launch -- a third
// of an iteration's launches and ~5 us of its latency when a solve is launch bound (N <= ~256).
constexpr int SQ_INLINE_MAX = 2048;
template <int D>
__device__ inline double step_length_inline_256(const double* __restrict__ part_rz, int nblk, int64_t nW, int64_t C, double rho_c,
                                                const int* __restrict__ wk, const int* __restrict__ wi,
                                                const int* __restrict__ wj, const double* __restrict__ weta,
                                                const double* __restrict__ Qp) {
  __shared__ double sw[2][4];
  double v = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) v += part_rz[b];
  for (int64_t n = threadIdx.x; n < nW; n += 256) {
    const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
    const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
    double ax = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) ax += weta[n * D + d] * (Qp[bi + d] - Qp[bj + d]);
    q += ax * ax;
  }
  v = wave_incl_sum(v);
  q = wave_incl_sum(q);
  if ((threadIdx.x & 63) == 63) {
    sw[0][threadIdx.x >> 6] = v;
    sw[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  const double rz = (sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]);
  const double pHp = rz + rho_c * ((sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]));
  return (pHp > 0.0 && rz != 0.0) ? rz / pHp : 0.0;
}

constexpr int UPD_RPT = 2;  // slab rows per thread in the elementwise part of cg1_update_kernel (32 rows per workgroup)

// Launch 3 of 3: everything that follows the step length, elementwise.  Workgroups [0, eblocks): 32 slab
// rows of ONE 16-column block each -- fixed rows: F x~ = F x + a F p, z/y update, F x += alpha a F p; x rows:
// x += alpha a p, S0 x (new buffer) = S0 x + alpha a S0 p.  Workgroup w works on column block w % nblk8 (nblk8 =
// nblk rounded up to a multiple of 8): workgroups go round-robin over the 8 XCDs, so the slabs are rewritten in the L2
// of the XCD whose cg1_col_kernel workgroup reads them next (a row-major split left every read a remote miss: 8 us
// of load latency in the column kernel instead of 2.6).  Workgroups [eblocks, ...): collision rows -- S0 x~ and the
// new S0 x formed on the fly from (S0 x, S0 p) with the same fma, z/y update, then the row values of the NEXT
// right-hand side   g = rho zc - yc - rho eta.d(S0 x)      (A_W^T g is gathered by cg1_col_kernel).
template <int D>
__global__ __launch_bounds__(256) void cg1_update_kernel(int K, int Rf, int64_t C, int nblk, int npart, int eblocks, int nblk8, double rho,
                                                          double rho_c, double alpha, const double* __restrict__ part_rz,
                                                          const double* __restrict__ part_sq,
                                                          const double* __restrict__ wrow, const double* __restrict__ lf,
                                                          const double* __restrict__ uf, double* __restrict__ zf,
                                                          double* __restrict__ yf, double* __restrict__ Fx,
                                                          const double* __restrict__ Fp, double* __restrict__ x,
                                                          const double* __restrict__ pdir,
                                                          const double* __restrict__ Qp, const double* __restrict__ Qx,
                                                          double* __restrict__ Qn, int64_t nW, const int* __restrict__ wk,
                                                          const int* __restrict__ wi, const int* __restrict__ wj,
                                                          const double* __restrict__ weta, const double* __restrict__ wl,
                                                          double* __restrict__ zc, double* __restrict__ yc,
                                                          const int* __restrict__ pos_i, const int* __restrict__ pos_j,
                                                          double* __restrict__ gval, double* __restrict__ dyf,
                                                          double* __restrict__ dyc, int inline_sq) {
  // all operands are loaded BEFORE the step length is reduced (one memory round trip for both)
  if ((int)blockIdx.x < eblocks) {
    const int seg = blockIdx.x / nblk8, blk = blockIdx.x - seg * nblk8;
    const int64_t col = (int64_t)blk * CB + (threadIdx.x & 15);
    const bool live = blk < nblk && col < C;
    const int rows = Rf + K;
    double v0[UPD_RPT], v1[UPD_RPT], v2[UPD_RPT], v3[UPD_RPT], v4[UPD_RPT], v5[UPD_RPT], rr[UPD_RPT];
#pragma unroll
    for (int u = 0; u < UPD_RPT; ++u) {
      const int row = seg * (16 * UPD_RPT) + u * 16 + (threadIdx.x >> 4);
      v0[u] = v1[u] = v2[u] = v3[u] = v4[u] = v5[u] = 0.0;
      rr[u] = 1.0;
      if (live && row < Rf) {
        const int64_t g = (int64_t)row * C + col;
        v0[u] = Fx[g]; v1[u] = Fp[g]; v2[u] = zf[g]; v3[u] = yf[g]; v4[u] = lf[g]; v5[u] = uf[g];
        rr[u] = rho * wrow[row];
      } else if (live && row < rows) {
        const int64_t g = (int64_t)(row - Rf) * C + col;
        v0[u] = x[g]; v1[u] = pdir[g]; v2[u] = Qx[g]; v3[u] = Qp[g];
      }
    }
    const double a = inline_sq ? step_length_inline_256<D>(part_rz, npart, nW, C, rho_c, wk, wi, wj, weta, Qp)
                               : step_length_256(part_rz, npart, part_sq);
    const double aa = alpha * a;
    if (!live) return;
#pragma unroll
    for (int u = 0; u < UPD_RPT; ++u) {
      const int row = seg * (16 * UPD_RPT) + u * 16 + (threadIdx.x >> 4);
      if (row < Rf) {
        const int64_t g = (int64_t)row * C + col;
        const double zh = alpha * fma(a, v1[u], v0[u]) + (1.0 - alpha) * v2[u];
        const double y = v3[u];
        const double zn = fmin(fmax(zh + y / rr[u], v4[u]), v5[u]);
        const double yn = y + rr[u] * (zh - zn);
        yf[g] = yn;
        zf[g] = zn;
        Fx[g] = fma(aa, v1[u], v0[u]);
        if (dyf) dyf[g] = yn - y;  // delta-y of this iteration (primal infeasibility certificate)
      } else if (row < rows) {
        const int64_t g = (int64_t)(row - Rf) * C + col;
        x[g] = fma(aa, v1[u], v0[u]);
        Qn[g] = fma(aa, v3[u], v2[u]);
      }
    }
    return;
  }
  const int64_t n = (int64_t)(blockIdx.x - eblocks) * 256 + threadIdx.x;
  const bool live = n < nW;
  double e[D], qi[D], qj[D], pi[D], pj[D], z = 0.0, y = 0.0, lo = 0.0;
  int posi = 0, posj = 0;
  if (live) {
    const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
    const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      e[d] = weta[n * D + d];
      qi[d] = Qx[bi + d]; qj[d] = Qx[bj + d]; pi[d] = Qp[bi + d]; pj[d] = Qp[bj + d];
    }
    z = zc[n]; y = yc[n]; lo = wl[n];
    posi = pos_i[n]; posj = pos_j[n];
  }
  const double a = inline_sq ? step_length_inline_256<D>(part_rz, npart, nW, C, rho_c, wk, wi, wj, weta, Qp)
                             : step_length_256(part_rz, npart, part_sq);
  const double aa = alpha * a;
  if (!live) return;
  double tc = 0.0, ax = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    tc += e[d] * (fma(a, pi[d], qi[d]) - fma(a, pj[d], qj[d]));
    ax += e[d] * (fma(aa, pi[d], qi[d]) - fma(aa, pj[d], qj[d]));
  }
  const double zh = alpha * tc + (1.0 - alpha) * z;
  const double zn = fmax(zh + y / rho_c, lo);
  const double yn = y + rho_c * (zh - zn);
  yc[n] = yn;
  zc[n] = zn;
  if (dyc) dyc[n] = fmin(yn - y, 0.0);  // u = +inf: projected onto the polar of the recession cone
  const double g = (rho_c * zn - yn) - rho_c * ax;
  gval[posi] = g;
  gval[posj] = g;
}

// row values g written to BOTH incidence-list entries of a row (pos_i, pos_j); the per-cell gathers then need no atomics:
//   INIT: g = (rho zc - yc) - rho eta.(Q_i - Q_j)   right-hand side minus the collision part of H x (Q = S0 x); also the
//         first row values of the single-step pipeline after (x, zc, yc, rho) changed outside it
//   else: g = rho eta.(Q_i - Q_j)                   collision part of H v (Q = S0 v)
template <int D, bool INIT>
__global__ __launch_bounds__(256) void rows_value_kernel(int64_t nW, int64_t C, double rho, const int* __restrict__ wk,
                                                          const int* __restrict__ wi, const int* __restrict__ wj,
                                                          const double* __restrict__ weta, const double* __restrict__ Q,
                                                          const double* __restrict__ zc, const double* __restrict__ yc,
                                                          const int* __restrict__ pos_i, const int* __restrict__ pos_j,
                                                          double* __restrict__ gval) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nW) return;
  const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
  const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
  double ax = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) ax += weta[n * D + d] * (Q[bi + d] - Q[bj + d]);
  const double g = INIT ? (rho * zc[n] - yc[n]) - rho * ax : rho * ax;
  gval[pos_i[n]] = g;
  gval[pos_j[n]] = g;
}

#define FUSED_LAUNCHED(qp) SCP_HIP_CHECK((qp)->ctx, hipGetLastError())

// Tiles beyond 64 KiB of dynamic LDS (K > 50) need the limit raised per (device, kernel): scp_raise_lds_limit.
template <typename Kern>
int allow_lds(scp_qp* qp, Kern kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return SCP_OK;
  SCP_HIP_CHECK(qp->ctx, scp_raise_lds_limit(qp->ctx->device, reinterpret_cast<const void*>(kernel), bytes));
  return SCP_OK;
}

}  // namespace

int scp_qp_fused_iteration(scp_qp* qp, int* cg_count) {
  const QpDev& d = qp->d;
  hipStream_t s = qp->ctx->stream;
  const int K = qp->K, Rf = qp->Rf;
  const int64_t C = qp->C, nx = (int64_t)K * C;
  const int nblk = (int)((C + CB - 1) / CB);
  const int has_rows = qp->nW > 0 ? 1 : 0;
  const double rho_c = qp->rho * qp->st.rho_col_scale;
  const dim3 cgrid(nblk), cblock(FT);
  const dim3 rgrid((unsigned)((qp->nW + 255) / 256)), rblock(256);
  const size_t tile = (size_t)CB * sizeof(double);
  double* Q = d.HQ + nx;   // S0 v slab
  double* Hp = d.HQ;       // H p slab
  double* part_rz = d.part;
  double* part_php = d.part + SCP_PART_CAP;
  int fold = 0, fold_first = 0, fold_slot = SL_RZ0;
  const double* fold_part = part_rz;

  {
    int rc = allow_lds(qp, fused_pre_kernel, (size_t)(4 * K + Rf) * tile);
    if (!rc) rc = allow_lds(qp, fused_cg_init_kernel, (size_t)(5 * K) * tile);
    if (!rc) rc = allow_lds(qp, fused_post_kernel, (size_t)(2 * K + Rf) * tile);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(fused_pre_kernel, cgrid, cblock, (size_t)(4 * K + Rf) * tile, s, K, Rf, C, qp->rho, qp->st.sigma,
                     has_rows, d.pFt, d.pHS, d.pMinv, d.wrow, d.x, d.zf, d.yf, d.rhs, Q, d.xt, d.G);
  FUSED_LAUNCHED(qp);
  if (has_rows) {
    {
      int rc = scp_qp_rows_gather(qp, true, Q);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(fused_cg_init_kernel, cgrid, cblock, (size_t)(5 * K) * tile, s, K, C, d.pS0t, d.pMinv, d.pHS, d.rhs,
                       d.G, d.r, d.p, d.hpf, Q, part_rz);
    FUSED_LAUNCHED(qp);
    int slot = SL_RZ0;
    const int ncg = qp->st.cg_iters;
    for (int it = 0; it < ncg; ++it) {
      {
        int rc = scp_qp_rows_gather(qp, false, Q);
        if (rc) return rc;
      }
      hipLaunchKernelGGL(fused_cg_hp_kernel, cgrid, cblock, (size_t)(2 * K) * tile, s, K, C, d.pS0t, d.G, d.hpf, d.p, Hp,
                         part_php);
      FUSED_LAUNCHED(qp);
      ++*cg_count;
      if (it + 1 == ncg) {  // the last step is folded into the post kernel
        fold = 1;
        fold_first = it == 0 ? 1 : 0;
        fold_slot = slot;
        fold_part = (it & 1) ? part_rz + SCP_PART_CAP / 2 : part_rz;
        break;
      }
      // the step kernel reads part_rz (first step) or scal[slot], writes the new partials to the OTHER half of
      // the rz array so that workgroups still summing the old partials are not disturbed
      double* part_new = (it & 1) ? part_rz : part_rz + SCP_PART_CAP / 2;
      double* part_old = (it & 1) ? part_rz + SCP_PART_CAP / 2 : part_rz;
      hipLaunchKernelGGL(fused_cg_step_kernel, cgrid, cblock, (size_t)(2 * K) * tile, s, K, C, nblk, it == 0 ? 1 : 0, slot,
                         d.pMinv, d.scal, part_old, part_php, d.p, Hp, d.xt, d.r, d.zz, part_new);
      FUSED_LAUNCHED(qp);
      hipLaunchKernelGGL(fused_cg_dir_kernel, cgrid, cblock, (size_t)(3 * K) * tile, s, K, C, nblk, slot, d.pHS, d.scal,
                         part_new, d.zz, d.p, d.hpf, Q);
      FUSED_LAUNCHED(qp);
      slot ^= 1;
    }
  }
  hipLaunchKernelGGL(fused_post_kernel, cgrid, cblock, (size_t)(2 * K + Rf) * tile, s, K, Rf, C, qp->rho, qp->st.alpha,
                     has_rows, fold, nblk, fold_first, fold_slot, d.scal, fold_part, part_php, d.p, d.pF, d.pS0, d.wrow, d.xt,
                     d.lf, d.uf, d.zf, d.yf, d.x, Q);
  FUSED_LAUNCHED(qp);
  if (has_rows) {
    if (qp->D == 2)
      hipLaunchKernelGGL(fused_row_update_kernel<2>, rgrid, rblock, 0, s, qp->nW, C, rho_c, qp->st.alpha, d.w_k, d.w_i,
                         d.w_j, d.w_eta, d.w_l, Q, d.zc, d.yc);
    else
      hipLaunchKernelGGL(fused_row_update_kernel<3>, rgrid, rblock, 0, s, qp->nW, C, rho_c, qp->st.alpha, d.w_k, d.w_i,
                         d.w_j, d.w_eta, d.w_l, Q, d.zc, d.yc);
    FUSED_LAUNCHED(qp);
  }
  return SCP_OK;
}

// `nit` ADMM iterations on the fixed rows alone in one launch (qp->nW == 0); dy_out: where to leave delta-y of the last
// iteration for scp_qp_fused_residuals (NULL: not needed)
int scp_qp_qp0_iterations(scp_qp* qp, int nit, double* dy_out) {
  const QpDev& d = qp->d;
  hipStream_t s = qp->ctx->stream;
  const int K = qp->K, Rf = qp->Rf;
  const int64_t C = qp->C;
  const int nblk = (int)((C + CB - 1) / CB);
  const size_t lds = (size_t)CB * (pad_col(Rf) + 3 * pad_col(K)) * sizeof(double);
  if (K <= 64) {
    int rc = allow_lds(qp, qp0_col_kernel<1>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(qp0_col_kernel<1>, dim3(nblk), dim3(FT), lds, s, K, Rf, C, qp->rho, qp->st.sigma, qp->st.alpha, qp->h,
                       nit, d.pMinv, d.wrow, d.lf, d.uf, d.x, d.zf, d.yf, dy_out);
  } else {
    int rc = allow_lds(qp, qp0_col_kernel<2>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(qp0_col_kernel<2>, dim3(nblk), dim3(FT), lds, s, K, Rf, C, qp->rho, qp->st.sigma, qp->st.alpha, qp->h,
                       nit, d.pMinv, d.wrow, d.lf, d.uf, d.x, d.zf, d.yf, dy_out);
  }
  FUSED_LAUNCHED(qp);
  return SCP_OK;
}

namespace {
constexpr int CSR1_MAX_CELLS = 16384;
constexpr int64_t CSR1_MAX_ROWS = 1 << 18;
// rows [base, base + n) that the kernel recomputes from the linearisation point first (scp_qp_add_rows_at); n = 0: none
struct CsrNewRows {
  int64_t n, base;
  const int64_t* rows;
  const double *pos_prev, *p0, *v0;
  double R, h;
  int N;
  int64_t* w_row;
  double* wl;
};
__global__ void csr_small_kernel(int64_t nW, int K, int ncell, int D, int64_t C, double rho, int* __restrict__ wk,
                                 int* __restrict__ wi, int* __restrict__ wj, double* __restrict__ weta,
                                 const double* __restrict__ Qx, double* __restrict__ zc, double* __restrict__ yc,
                                 int* __restrict__ ptr, int* __restrict__ ent, double* __restrict__ coef,
                                 int* __restrict__ pos_i, int* __restrict__ pos_j, double* __restrict__ gval, CsrNewRows nr);
}  // namespace

// Bring the single-step pipeline's carried state in line with (x, zc, yc, rho): S0 x and F x exact, row values g.
int scp_qp_cg1_prepare(scp_qp* qp) {
  const QpDev& d = qp->d;
  hipStream_t s = qp->ctx->stream;
  const int K = qp->K;
  const int64_t C = qp->C, nx = (int64_t)K * C;
  const double rho_c = qp->rho * qp->st.rho_col_scale;
  if (!qp->qx_fresh) {
    qp->qx_sel = 0;
    qp->gval_valid = false;
    int rc = scp_launch_gemm(qp->ctx, 1, K, K, (int)C, 1.0, d.S0, d.x, 0.0, d.HQ + nx);
    if (!rc) rc = scp_launch_gemm(qp->ctx, 1, qp->Rf, K, (int)C, 1.0, d.F, d.x, 0.0, d.fx);
    if (rc) return rc;
  }
  qp->qx_fresh = false;
  const double* Qx = qp->qx_sel ? d.HQ : d.HQ + nx;
  if (qp->csr_valid && qp->gval_valid && qp->gval_rho_c == rho_c) {  // scp_qp_install_rows_small built lists and values already
    qp->gval_valid = false;  // (the iterations about to run carry them on)
    qp->cg1_ready = true;
    return SCP_OK;
  }
  qp->gval_valid = false;
  if (!qp->csr_valid && qp->nW > 0 && qp->nW <= CSR1_MAX_ROWS && qp->N * K <= CSR1_MAX_CELLS) {
    const int ncell = qp->N * K;
    hipLaunchKernelGGL(csr_small_kernel, dim3(1), dim3(1024), (size_t)ncell * sizeof(int), s, qp->nW, K, ncell, qp->D, C, rho_c,
                       d.w_k, d.w_i, d.w_j, d.w_eta, Qx, d.zc, d.yc, d.cell_ptr, d.ent_code, d.coef, d.pos_i, d.pos_j, d.gval,
                       CsrNewRows{});
    FUSED_LAUNCHED(qp);
    qp->csr_valid = true;
    qp->cg1_ready = true;
    return SCP_OK;
  }
  if (!qp->csr_valid) {
    int rc = scp_qp_csr_build(qp);
    if (rc) return rc;
  }
  const dim3 rgrid((unsigned)((qp->nW + 255) / 256)), rblock(256);
  if (qp->D == 2)
    hipLaunchKernelGGL((rows_value_kernel<2, true>), rgrid, rblock, 0, s, qp->nW, C, rho_c, d.w_k, d.w_i, d.w_j, d.w_eta, Qx,
                       d.zc, d.yc, d.pos_i, d.pos_j, d.gval);
  else
    hipLaunchKernelGGL((rows_value_kernel<3, true>), rgrid, rblock, 0, s, qp->nW, C, rho_c, d.w_k, d.w_i, d.w_j, d.w_eta, Qx,
                       d.zc, d.yc, d.pos_i, d.pos_j, d.gval);
  FUSED_LAUNCHED(qp);
  qp->cg1_ready = true;
  return SCP_OK;
}

int scp_qp_install_rows_small(scp_qp* qp, int64_t n, const int64_t* rows, const double* pos_prev, const double* p0,
                              const double* v0, double R, const double* Qx, bool* done) {
  *done = false;
  const QpDev& d = qp->d;
  const int K = qp->K;
  const int64_t nW = qp->nW + n;
  if (!qp->qx_fresh || n <= 0 || nW > CSR1_MAX_ROWS || qp->N * K > CSR1_MAX_CELLS) return SCP_OK;
  const int ncell = qp->N * K;
  const double rho_c = qp->rho * qp->st.rho_col_scale;
  CsrNewRows nr{n, qp->nW, rows, pos_prev, p0, v0, R, qp->h, qp->N, d.w_row, d.w_l};
  hipLaunchKernelGGL(csr_small_kernel, dim3(1), dim3(1024), (size_t)ncell * sizeof(int), qp->ctx->stream, nW, K, ncell, qp->D,
                     qp->C, rho_c, d.w_k, d.w_i, d.w_j, d.w_eta, Qx, d.zc, d.yc, d.cell_ptr, d.ent_code, d.coef, d.pos_i,
                     d.pos_j, d.gval, nr);
  FUSED_LAUNCHED(qp);
  qp->csr_valid = true;
  qp->gval_valid = true;
  qp->gval_rho_c = rho_c;
  *done = true;
  return SCP_OK;
}

// One ADMM iteration with a single preconditioned CG step from x, in three launches: column blocks (r, then
// [p ; S0 p ; F p]), collision rows (p.H p), elementwise update of every row and column (ping-pong of S0 x).
int scp_qp_cg1_iteration(scp_qp* qp, int* cg_count, bool emit_dy) {
  const QpDev& d = qp->d;
  double* dyf = emit_dy ? d.dyf : nullptr;  // delta-y of this iteration, consumed by scp_qp_fused_residuals(.., 2)
  double* dyc = emit_dy ? d.dyc : nullptr;
  hipStream_t s = qp->ctx->stream;
  const int K = qp->K, Rf = qp->Rf;
  const int64_t C = qp->C, nx = (int64_t)K * C;
  const int nblk = (int)((C + CB - 1) / CB);
  double* Qp = d.hpf;  // S0 p
  double* Fp = d.tf;   // F p
  double* part_rz = d.part;
  double* part_sq = d.part + SCP_PART_CAP / 2;
  const double rho_c = qp->rho * qp->st.rho_col_scale;
  if (!qp->cg1_ready) {
    int rc = scp_qp_cg1_prepare(qp);
    if (rc) return rc;
  }
  double* Qx = qp->qx_sel ? d.HQ : d.HQ + nx;  // S0 x (carried)
  double* Qn = qp->qx_sel ? d.HQ + nx : d.HQ;  // S0 x of the next iteration
  const size_t lds = (size_t)CB * (pad_col(Rf) + 5 * pad_col(K)) * sizeof(double);
  int npart = nblk;  // partial sums of r.p: one per column block, or one per column (long horizons)
  if (K > SCP_FUSED_MAX_K) {
    npart = (int)C;
    hipLaunchKernelGGL(cg1_colK_kernel, dim3((unsigned)C), dim3((unsigned)((K + 63) / 64 * 64)), (size_t)2 * K * sizeof(double), s, K, Rf,
                       C, qp->rho, qp->h, d.Minv, d.wrow, d.x, d.fx, d.zf, d.yf, qp->N, qp->D, d.cell_ptr, d.coef, d.gval, d.p,
                       Qp, Fp, part_rz);
  } else if (K <= 64) {
    int rc = allow_lds(qp, cg1_col_kernel<1>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(cg1_col_kernel<1>, dim3(nblk), dim3(FT), lds, s, K, Rf, C, qp->rho, qp->h, d.pMinv, d.wrow, d.x, d.fx,
                       d.zf, d.yf, qp->N, qp->D, d.cell_ptr, d.coef, d.gval, d.p, Qp, Fp, part_rz);
  } else {
    int rc = allow_lds(qp, cg1_col_kernel<2>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(cg1_col_kernel<2>, dim3(nblk), dim3(FT), lds, s, K, Rf, C, qp->rho, qp->h, d.pMinv, d.wrow, d.x, d.fx,
                       d.zf, d.yf, qp->N, qp->D, d.cell_ptr, d.coef, d.gval, d.p, Qp, Fp, part_rz);
  }
  FUSED_LAUNCHED(qp);
  const dim3 rblock(256);
  const int inline_sq = qp->nW <= SQ_INLINE_MAX ? 1 : 0;  // small working set: the update kernel sums p.Hp's row term itself
  if (!inline_sq) {
    if (qp->D == 2)
      hipLaunchKernelGGL(cg1_rows_sq_kernel<2>, dim3(SQ_BLOCKS), rblock, 0, s, qp->nW, C, rho_c, d.w_k, d.w_i, d.w_j,
                         d.w_eta, Qp, part_sq);
    else
      hipLaunchKernelGGL(cg1_rows_sq_kernel<3>, dim3(SQ_BLOCKS), rblock, 0, s, qp->nW, C, rho_c, d.w_k, d.w_i, d.w_j,
                         d.w_eta, Qp, part_sq);
    FUSED_LAUNCHED(qp);
  }
  const int nblk8 = (nblk + 7) & ~7;
  const int eblocks = nblk8 * ((Rf + K + 16 * UPD_RPT - 1) / (16 * UPD_RPT));
  const dim3 ugrid((unsigned)(eblocks + (qp->nW + 255) / 256));
  if (qp->D == 2)
    hipLaunchKernelGGL(cg1_update_kernel<2>, ugrid, rblock, 0, s, K, Rf, C, nblk, npart, eblocks, nblk8, qp->rho, rho_c, qp->st.alpha,
                       part_rz, part_sq, d.wrow, d.lf, d.uf, d.zf, d.yf, d.fx, Fp, d.x, d.p, Qp, Qx, Qn, qp->nW, d.w_k,
                       d.w_i, d.w_j, d.w_eta, d.w_l, d.zc, d.yc, d.pos_i, d.pos_j, d.gval, dyf, dyc, inline_sq);
  else
    hipLaunchKernelGGL(cg1_update_kernel<3>, ugrid, rblock, 0, s, K, Rf, C, nblk, npart, eblocks, nblk8, qp->rho, rho_c, qp->st.alpha,
                       part_rz, part_sq, d.wrow, d.lf, d.uf, d.zf, d.yf, d.fx, Fp, d.x, d.p, Qp, Qx, Qn, qp->nW, d.w_k,
                       d.w_i, d.w_j, d.w_eta, d.w_l, d.zc, d.yc, d.pos_i, d.pos_j, d.gval, dyf, dyc, inline_sq);
  FUSED_LAUNCHED(qp);
  qp->qx_sel ^= 1;
  ++*cg_count;
  return SCP_OK;
}

// =====================================================================================================
// Incidence lists of the working rows per (time step, agent) cell: the deterministic replacement of the atomic
// row scatter.  Built once per change of the working set (count, scan, fill, sort inside the cells, finish).
// =====================================================================================================
namespace {

__global__ __launch_bounds__(256) void csr_count_kernel(int64_t nW, int K, const int* __restrict__ wk,
                                                         const int* __restrict__ wi, const int* __restrict__ wj,
                                                         int* __restrict__ cnt) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nW) return;
  atomicAdd(cnt + cell_of(wk[n], wi[n], K), 1);  // integer counts: order independent
  atomicAdd(cnt + cell_of(wk[n], wj[n], K), 1);
}

// exclusive scan of cnt[0..ncell) into ptr (in place: cnt and ptr are the same array), cursors = ptr.  Two launches that
// fill the chip instead of one workgroup walking the array (67 us at N K = 51 200, a fortieth of the benchmark step):
// (1) every workgroup scans its own 4096 cells and leaves their total, (2) every workgroup adds the totals before it.
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_mov0_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
constexpr int SCAN_SC = 16;                 // consecutive cells per thread
constexpr int SCAN_CELLS = 256 * SCAN_SC;   // per workgroup
__global__ __launch_bounds__(256) void csr_scan_local_kernel(int ncell, int* __restrict__ ptr, int* __restrict__ blk_tot) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * SCAN_CELLS + SCAN_SC * (int)threadIdx.x;
  int v[SCAN_SC], tot = 0;
#pragma unroll
  for (int e = 0; e < SCAN_SC; ++e) {
    v[e] = c + e < ncell ? ptr[c + e] : 0;
    tot += v[e];
  }
  int incl = tot;
  incl += dpp_mov0_i32<0x111, 0xF>(incl);
  incl += dpp_mov0_i32<0x112, 0xF>(incl);
  incl += dpp_mov0_i32<0x114, 0xF>(incl);
  incl += dpp_mov0_i32<0x118, 0xF>(incl);
  incl += dpp_mov0_i32<0x142, 0xA>(incl);
  incl += dpp_mov0_i32<0x143, 0xC>(incl);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int run = incl - tot;
  for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
  for (int e = 0; e < SCAN_SC; ++e) {
    if (c + e < ncell) ptr[c + e] = run;
    run += v[e];
  }
  if (threadIdx.x == 255) blk_tot[blockIdx.x] = run;
}

__global__ __launch_bounds__(256) void csr_scan_add_kernel(int ncell, int nblk, int* __restrict__ ptr, int* __restrict__ cur,
                                                            const int* __restrict__ blk_tot) {
  int off = 0;
  for (int b = 0; b < (int)blockIdx.x; ++b) off += blk_tot[b];  // (wave-uniform: scalar loads)
  const int c = blockIdx.x * SCAN_CELLS + SCAN_SC * (int)threadIdx.x;
#pragma unroll
  for (int e = 0; e < SCAN_SC; ++e) {
    if (c + e < ncell) {
      const int v = ptr[c + e] + off;
      ptr[c + e] = v;
      cur[c + e] = v;
    }
  }
  if (blockIdx.x == nblk - 1 && threadIdx.x == 0) ptr[ncell] = off + blk_tot[nblk - 1];
}

__global__ __launch_bounds__(256) void csr_fill_kernel(int64_t nW, int K, const int* __restrict__ wk,
                                                        const int* __restrict__ wi, const int* __restrict__ wj,
                                                        int* __restrict__ cur, int* __restrict__ ent) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nW) return;
  ent[atomicAdd(cur + cell_of(wk[n], wi[n], K), 1)] = (int)(2 * n);
  ent[atomicAdd(cur + cell_of(wk[n], wj[n], K), 1)] = (int)(2 * n + 1);
}

// entries of a cell arrive in atomic order: sort them (ascending code) so that every sum has a fixed order
__global__ __launch_bounds__(256) void csr_sort_kernel(int ncell, const int* __restrict__ ptr, int* __restrict__ ent) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncell) return;
  const int b = ptr[c], e = ptr[c + 1];
  for (int i = b + 1; i < e; ++i) {
    const int v = ent[i];
    int j = i - 1;
    while (j >= b && ent[j] > v) {
      ent[j + 1] = ent[j];
      --j;
    }
    ent[j + 1] = v;
  }
}

__global__ __launch_bounds__(256) void csr_finish_kernel(int64_t nent, int D, const int* __restrict__ ent,
                                                          const double* __restrict__ weta, double* __restrict__ coef,
                                                          int* __restrict__ pos_i, int* __restrict__ pos_j) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= nent) return;
  const int code = ent[t];
  const int n = code >> 1, side = code & 1;
  for (int d = 0; d < D; ++d) coef[t * D + d] = side ? -weta[(int64_t)n * D + d] : weta[(int64_t)n * D + d];
  if (side) pos_j[n] = (int)t;
  else pos_i[n] = (int)t;
}

// Small problems (N K <= CSR1_MAX_CELLS cells, e.g. 128 agents x 50 steps): the whole build -- count, scan, fill, sort,
// finish -- and the first row values (rows_value_kernel<D, true>) in ONE workgroup; the cell counters live in LDS.  Same lists,
// same order as the five-launch build.
template <bool COH>
__device__ inline void csr_small_body(int* csr_cnt, int64_t nW, int K, int ncell, int D, int64_t C, double rho,
                                                          int* __restrict__ wk, int* __restrict__ wi,
                                                          int* __restrict__ wj, double* __restrict__ weta,
                                                          const double* __restrict__ Qx, double* __restrict__ zc,
                                                          double* __restrict__ yc, int* __restrict__ ptr,
                                                          int* __restrict__ ent, double* __restrict__ coef,
                                                          int* __restrict__ pos_i, int* __restrict__ pos_j,
                                                          double* __restrict__ gval, CsrNewRows nr) {
  // csr_cnt: [ncell] ints of LDS: counts, then exclusive offsets, then fill cursors (= end of each cell)
  __shared__ int wsum[16];
  constexpr int SC = CSR1_MAX_CELLS / 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (nr.n > 0) {  // scp_qp_add_rows_at's kernel first: the new rows [base, base + n) from the linearisation point
    const int64_t pairs = (int64_t)nr.N * (nr.N - 1) / 2;
    for (int64_t t = tid; t < nr.n; t += 1024) {
      if (D == 2)
        add_row_at<2, COH>(t, nr.N, K, C, pairs, nr.base, nr.rows, nr.pos_prev, nr.p0, nr.v0, nr.R, nr.h, Qx, nr.w_row, wk, wi, wj,
                      weta, nr.wl, zc, yc);
      else
        add_row_at<3, COH>(t, nr.N, K, C, pairs, nr.base, nr.rows, nr.pos_prev, nr.p0, nr.v0, nr.R, nr.h, Qx, nr.w_row, wk, wi, wj,
                      weta, nr.wl, zc, yc);
    }
    __threadfence_block();
    __syncthreads();
  }
  for (int c = tid; c < ncell; c += 1024) csr_cnt[c] = 0;
  __syncthreads();
  for (int64_t n = tid; n < nW; n += 1024) {
    atomicAdd(&csr_cnt[cell_of(wk[n], wi[n], K)], 1);
    atomicAdd(&csr_cnt[cell_of(wk[n], wj[n], K)], 1);
  }
  __syncthreads();
  {  // exclusive scan, SC consecutive cells per thread
    int v[SC], tot = 0;
#pragma unroll
    for (int e = 0; e < SC; ++e) {
      v[e] = SC * tid + e < ncell ? csr_cnt[SC * tid + e] : 0;
      tot += v[e];
    }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int e = 0; e < SC; ++e) {
      if (SC * tid + e < ncell) {
        csr_cnt[SC * tid + e] = run;
        ptr[SC * tid + e] = run;
      }
      run += v[e];
    }
    if (tid == 1023) ptr[ncell] = run;
  }
  __syncthreads();
  for (int64_t n = tid; n < nW; n += 1024) {
    ent[atomicAdd(&csr_cnt[cell_of(wk[n], wi[n], K)], 1)] = (int)(2 * n);
    ent[atomicAdd(&csr_cnt[cell_of(wk[n], wj[n], K)], 1)] = (int)(2 * n + 1);
  }
  __syncthreads();
  for (int c = tid; c < ncell; c += 1024) {  // the cursor of a cell now stands at its end = the next cell's begin
    const int b = c ? csr_cnt[c - 1] : 0, e = csr_cnt[c];
    for (int i = b + 1; i < e; ++i) {
      const int v = ent[i];
      int j = i - 1;
      while (j >= b && ent[j] > v) {
        ent[j + 1] = ent[j];
        --j;
      }
      ent[j + 1] = v;
    }
  }
  __syncthreads();
  for (int64_t t = tid; t < 2 * nW; t += 1024) {
    const int code = ent[t];
    const int n = code >> 1, side = code & 1;
    for (int d = 0; d < D; ++d) coef[t * D + d] = side ? -weta[(int64_t)n * D + d] : weta[(int64_t)n * D + d];
    if (side) pos_j[n] = (int)t;
    else pos_i[n] = (int)t;
  }
  __syncthreads();
  for (int64_t n = tid; n < nW; n += 1024) {  // rows_value_kernel<D, true>
    const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
    const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
    double ax = 0.0;
    for (int d = 0; d < D; ++d)
      ax += weta[n * D + d] * (COH ? load_coherent(Qx + bi + d) - load_coherent(Qx + bj + d) : Qx[bi + d] - Qx[bj + d]);
    const double g = (rho * zc[n] - yc[n]) - rho * ax;
    gval[pos_i[n]] = g;
    gval[pos_j[n]] = g;
  }
}

__global__ __launch_bounds__(1024) void csr_small_kernel(int64_t nW, int K, int ncell, int D, int64_t C, double rho,
                                                          int* __restrict__ wk, int* __restrict__ wi,
                                                          int* __restrict__ wj, double* __restrict__ weta,
                                                          const double* __restrict__ Qx, double* __restrict__ zc,
                                                          double* __restrict__ yc, int* __restrict__ ptr,
                                                          int* __restrict__ ent, double* __restrict__ coef,
                                                          int* __restrict__ pos_i, int* __restrict__ pos_j,
                                                          double* __restrict__ gval, CsrNewRows nr) {
  extern __shared__ int csr_cnt[];
  csr_small_body<false>(csr_cnt, nW, K, ncell, D, C, rho, wk, wi, wj, weta, Qx, zc, yc, ptr, ent, coef, pos_i, pos_j, gval, nr);
}

// scp_qp_reset AND the installation of the QP's first rows in ONE launch: every workgroup resets its 16 columns
// (qp_reset_body: its first 256 threads), S0 x written through; the LAST workgroup to finish (a ticket) then runs
// csr_small_body on all 1024 threads, reading S0 x past its L2.  Same values as the two launches.
struct ResetArgs {
  int N, Rf;
  const double *x0, *F, *S0;
  double *x, *zf, *fx, *yf;
};
__global__ __launch_bounds__(1024) void reset_install_kernel(ResetArgs ra, int64_t nW, int K, int ncell, int D, int64_t C,
                                                              double rho, int* __restrict__ wk, int* __restrict__ wi,
                                                              int* __restrict__ wj, double* __restrict__ weta,
                                                              double* __restrict__ Qx, double* __restrict__ zc,
                                                              double* __restrict__ yc, int* __restrict__ ptr,
                                                              int* __restrict__ ent, double* __restrict__ coef,
                                                              int* __restrict__ pos_i, int* __restrict__ pos_j,
                                                              double* __restrict__ gval, CsrNewRows nr,
                                                              unsigned* __restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) char ri_lds[];  // max([K][16] doubles, [ncell] ints)
  __shared__ int last_sh;
  qp_reset_body<true>(threadIdx.x, threadIdx.x < 256, reinterpret_cast<double*>(ri_lds), ra.N, K, D, ra.Rf, ra.x0, ra.F, ra.S0,
                      ra.x, ra.zf, ra.fx, Qx, ra.yf);
  wait_stores_performed();  // (S0 x, written through, is in place before the ticket is taken)
  __syncthreads();
  if (threadIdx.x == 0) last_sh = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!last_sh) return;
  if (threadIdx.x == 0) *ticket = 0u;  // (the next launch on this stream starts after this kernel has ended)
  csr_small_body<true>(reinterpret_cast<int*>(ri_lds), nW, K, ncell, D, C, rho, wk, wi, wj, weta, Qx, zc, yc, ptr, ent, coef,
                       pos_i, pos_j, gval, nr);
}

}  // namespace

// The launch of scp_qp_reset(x0) + scp_qp_add_rows_at(rows) for small problems (scp_qp.hip: reset_impl keeps the host-side
// state); *done = false: not eligible, nothing launched.  Qx: the slab S0 x goes to.  qp->rho is the QP's starting value.
int scp_qp_reset_install_small(scp_qp* qp, const double* x0, int64_t n, const int64_t* rows, const double* pos_prev,
                               const double* p0, const double* v0, double R, double* Qx, bool* done) {
  *done = false;
  const QpDev& d = qp->d;
  const int K = qp->K;
  if (n <= 0 || n > CSR1_MAX_ROWS || qp->N * K > CSR1_MAX_CELLS) return SCP_OK;
  const int ncell = qp->N * K;
  const double rho_c = qp->rho * qp->st.rho_col_scale;
  const size_t lds = std::max((size_t)K * RESET_COLS * sizeof(double), (size_t)ncell * sizeof(int));
  CsrNewRows nr{n, 0, rows, pos_prev, p0, v0, R, qp->h, qp->N, d.w_row, d.w_l};
  ResetArgs ra{qp->N, qp->Rf, x0, d.F, d.S0, d.x, d.zf, d.fx, d.yf};
  hipLaunchKernelGGL(reset_install_kernel, dim3((unsigned)scp_cdiv(qp->C, RESET_COLS)), dim3(1024), lds, qp->ctx->stream, ra, n,
                     K, ncell, qp->D, qp->C, rho_c, d.w_k, d.w_i, d.w_j, d.w_eta, Qx, d.zc, d.yc, d.cell_ptr, d.ent_code,
                     d.coef, d.pos_i, d.pos_j, d.gval, nr, qp->ctx->d_ticket + 2);  // ([0], [1]: the passes, the checks)
  FUSED_LAUNCHED(qp);
  qp->gval_rho_c = rho_c;
  *done = true;
  return SCP_OK;
}

namespace {
// row values for the residual / certificate scatters
__global__ __launch_bounds__(256) void csr_rowval_kernel(int64_t nW, int mode, double rho, const double* __restrict__ zc,
                                                          const double* __restrict__ yc, const double* __restrict__ vec,
                                                          const int* __restrict__ pos_i, const int* __restrict__ pos_j,
                                                          double* __restrict__ gval, const double* __restrict__ vec2,
                                                          double* __restrict__ gval2) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nW) return;
  if (gval2) {  // a second row vector for the same gather
    const double g2 = vec2[n];
    gval2[pos_i[n]] = g2;
    gval2[pos_j[n]] = g2;
  }
  const double g = mode == 0 ? rho * zc[n] - yc[n] : (mode == 1 ? yc[n] : vec[n]);
  gval[pos_i[n]] = g;
  gval[pos_j[n]] = g;
}

// G[k][col] = sum over the cell's entries of coef * gval
__global__ __launch_bounds__(256) void csr_gather_kernel(int K, int N, int D, const int* __restrict__ ptr,
                                                          const double* __restrict__ coef,
                                                          const double* __restrict__ gval, double* __restrict__ G) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t C = (int64_t)N * D;
  if (t >= C * K) return;
  const int k = (int)(t / C), col = (int)(t % C);
  const int agent = col / D, d = col - agent * D;
  const int cell = cell_of(k, agent, K);
  double acc = 0.0;
  const int t1 = ptr[cell + 1];
  for (int e = ptr[cell]; e < t1; ++e) acc += coef[(size_t)e * D + d] * gval[e];
  G[t] = acc;
}

}  // namespace

int scp_qp_csr_build(scp_qp* qp) {
  const QpDev& d = qp->d;
  hipStream_t s = qp->ctx->stream;
  const int ncell = qp->N * qp->K;
  SCP_REQUIRE(qp->ctx, 2 * qp->nW < 0x3FFFFFFF, "csr_build: too many working rows for 32-bit entry codes");
  SCP_HIP_CHECK(qp->ctx, hipMemsetAsync(d.cell_ptr, 0, (size_t)(ncell + 1) * sizeof(int), s));
  if (qp->nW > 0) {
    const dim3 rgrid((unsigned)((qp->nW + 255) / 256));
    hipLaunchKernelGGL(csr_count_kernel, rgrid, dim3(256), 0, s, qp->nW, qp->K, d.w_k, d.w_i, d.w_j, d.cell_ptr);
    const int sblk = (ncell + SCAN_CELLS - 1) / SCAN_CELLS;
    hipLaunchKernelGGL(csr_scan_local_kernel, dim3(sblk), dim3(256), 0, s, ncell, d.cell_ptr, d.scan_tot);
    hipLaunchKernelGGL(csr_scan_add_kernel, dim3(sblk), dim3(256), 0, s, ncell, sblk, d.cell_ptr, d.cell_cur, d.scan_tot);
    hipLaunchKernelGGL(csr_fill_kernel, rgrid, dim3(256), 0, s, qp->nW, qp->K, d.w_k, d.w_i, d.w_j, d.cell_cur, d.ent_code);
    hipLaunchKernelGGL(csr_sort_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, ncell, d.cell_ptr, d.ent_code);
    hipLaunchKernelGGL(csr_finish_kernel, dim3((unsigned)((2 * qp->nW + 255) / 256)), dim3(256), 0, s, 2 * qp->nW, qp->D,
                       d.ent_code, d.w_eta, d.coef, d.pos_i, d.pos_j);
    FUSED_LAUNCHED(qp);
  }
  qp->csr_valid = true;
  return SCP_OK;
}

int scp_qp_csr_scatter(scp_qp* qp, int mode, const double* vec) {
  const QpDev& d = qp->d;
  hipStream_t s = qp->ctx->stream;
  const int64_t nx = (int64_t)qp->K * qp->C;
  hipLaunchKernelGGL(csr_rowval_kernel, dim3((unsigned)((qp->nW + 255) / 256)), dim3(256), 0, s, qp->nW, mode,
                     qp->rho * qp->st.rho_col_scale, d.zc, d.yc, vec, d.pos_i, d.pos_j, d.gval, (const double*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL(csr_gather_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, qp->K, qp->N, qp->D,
                     d.cell_ptr, d.coef, d.gval, d.G);
  FUSED_LAUNCHED(qp);
  return SCP_OK;
}

// G = A_W^T g with g = (rho zc - yc) - rho A_W v (init) or rho A_W v, Q = S0 v: row values, then the per-cell gather
int scp_qp_rows_gather(scp_qp* qp, bool init, const double* Q) {
  const QpDev& d = qp->d;
  hipStream_t s = qp->ctx->stream;
  if (!qp->csr_valid) {
    int rc = scp_qp_csr_build(qp);
    if (rc) return rc;
  }
  const int64_t C = qp->C, nx = (int64_t)qp->K * C;
  const double rho_c = qp->rho * qp->st.rho_col_scale;
  const dim3 rgrid((unsigned)((qp->nW + 255) / 256)), rblock(256);
#define SCP_ROWS_VALUE(DD, INIT)                                                                                        \
  hipLaunchKernelGGL((rows_value_kernel<DD, INIT>), rgrid, rblock, 0, s, qp->nW, C, rho_c, d.w_k, d.w_i, d.w_j, d.w_eta, Q, \
                     d.zc, d.yc, d.pos_i, d.pos_j, d.gval)
  if (qp->D == 2) {
    if (init) SCP_ROWS_VALUE(2, true);
    else SCP_ROWS_VALUE(2, false);
  } else {
    if (init) SCP_ROWS_VALUE(3, true);
    else SCP_ROWS_VALUE(3, false);
  }
#undef SCP_ROWS_VALUE
  hipLaunchKernelGGL(csr_gather_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, qp->K, qp->N, qp->D,
                     d.cell_ptr, d.coef, d.gval, d.G);
  FUSED_LAUNCHED(qp);
  qp->cg1_ready = false;  // gval is the single-step pipeline's carried row value too
  return SCP_OK;
}

// =====================================================================================================
// Operand packing of the constant blocks (see QpDev::pF ...)
// =====================================================================================================
namespace {
struct PackDesc {
  const double* src;
  double* dst;
  int R, M;
};
struct PackArgs {
  PackDesc m[7];
};
__global__ __launch_bounds__(256) void pack_operands_kernel(PackArgs a) {
  const PackDesc d = a.m[blockIdx.y];
  const int nks = (d.M + 3) >> 2;
  const int64_t total = (int64_t)((d.R + 15) >> 4) * nks * 64;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int lane = (int)(e & 63);
    const int64_t q = e >> 6;
    const int ks = (int)(q % nks), t = (int)(q / nks);
    const int row = t * 16 + (lane & 15), k = 4 * ks + (lane >> 4);
    d.dst[e] = (row < d.R && k < d.M) ? d.src[(size_t)row * d.M + k] : 0.0;
  }
}
}  // namespace

int scp_qp_pack_operands(scp_qp* qp, bool constants) {
  const QpDev& d = qp->d;
  const int K = qp->K, Rf = qp->Rf;
  PackArgs a;
  int n = 0;
  if (constants) {
    a.m[n++] = {d.F, d.pF, Rf, K};
    a.m[n++] = {d.Ft, d.pFt, K, Rf};
    a.m[n++] = {d.S0, d.pS0, K, K};
    a.m[n++] = {d.S0t, d.pS0t, K, K};
  } else {
    a.m[n++] = {d.HS, d.pHS, 2 * K, K};
    a.m[n++] = {d.Minv, d.pMinv, K, K};
    a.m[n++] = {d.T, d.pT, K, K};
  }
  hipLaunchKernelGGL(pack_operands_kernel, dim3(16, n), dim3(256), 0, qp->ctx->stream, a);
  FUSED_LAUNCHED(qp);
  return SCP_OK;
}

// =====================================================================================================
// Termination check of the single-step pipeline, fused: one row launch (gval = yc), one column-block launch
// (F x, S0 x, A^T y = F^T y_f + S0^T gather, delta-y of the fixed rows, all their maxima), one row launch
// (collision rows' residuals and delta-y).  14 launches on the generic path.
// =====================================================================================================
namespace {

__device__ inline double wg_max(double v) {
  __shared__ double s[NWV];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) t = fmax(t, s[w]);
  __syncthreads();
  return t;
}

__device__ inline void atomic_max_nn(double* addr, double v) {
  atomicMax((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
}

// per-workgroup partial results of a check: [rp, |Ax|, |z|, rd, |Px|, |A^T y|, |dy|, supp, |A^T dy|] (maxima / a sum),
// reduced on the host -- 7 same-address atomics per workgroup cost more than the rest of the kernel
constexpr int RS_RP = 0, RS_NAX = 1, RS_NZ = 2, RS_RD = 3, RS_NPX = 4, RS_NATY = 5, RS_NDY = 6, RS_SUPP = 7, RS_NATDY = 8;

// A^T v for one column (one wave), v = (v_f in Vc[0, Rf), v_c gathered in Gc[0, K)), time steps in DESCENDING order
// (k = KM - i): reverse cumulative sums as ascending scans.  Returns the value for step k of index (lane, e).
template <int E>
__device__ inline void column_At(int K, double h, const double* Vc, const double* Gc, double (&out)[E]) {
  const int lane = threadIdx.x & 63, KM = 64 * E - 1;
  const double hh = h * h;
  double yj[E], ya[E], u1[E], u2[E], g[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = KM - (lane * E + e);
    const bool ok = k < K;
    yj[e] = k < K - 1 ? Vc[k] : 0.0;
    ya[e] = ok ? Vc[K - 1 + k] : 0.0;
    const double yv = ok ? Vc[2 * K - 1 + k] : 0.0;
    const double yp = ok ? Vc[3 * K - 1 + k] : 0.0;
    g[e] = ok ? Gc[k] : 0.0;
    u1[e] = h * yv + 0.5 * hh * (yp - g[e]);
    u2[e] = yp + g[e];
  }
  double d1[E], d2[E], s1[E], s2[E], yjp[E];
  wave_scan<E>(u1, d1, s1);
  wave_scan<E>(u2, s1, s2);
  wave_scan<E>(s1, s2, d2);
  wave_next<E>(yj, yjp);
#pragma unroll
  for (int e = 0; e < E; ++e) out[e] = ((yjp[e] - yj[e]) / h + ya[e]) + (d1[e] + 0.5 * hh * g[e]) + hh * d2[e];
}

// Column part of the check, with the integrator blocks as wave scans like cg1_col_kernel (one wave per column):
//   F x, S0 x (stored: the pipeline's carried slabs are refreshed exactly), A^T y = F^T y_f + S0^T G(yc), the maxima of
//   the primal / dual residuals over this block's 16 columns and, with_dy (dyf / gval2 hold delta-y of the last
//   iteration), |dy|, its support value and |A^T dy| of OSQP's primal infeasibility certificate.
template <int E>
__global__ __launch_bounds__(FT) void cg1_resid_col_kernel(int K, int Rf, int64_t C, double h, int N, int D, int with_dy,
                                                            int has_rows, const double* __restrict__ x, const double* __restrict__ zf,
                                                            const double* __restrict__ yf, const double* __restrict__ lf,
                                                            const double* __restrict__ uf, const double* __restrict__ dyf,
                                                            const int* __restrict__ cell_ptr,
                                                            const double* __restrict__ coef,
                                                            const double* __restrict__ gval,
                                                            const double* __restrict__ gval2, double* __restrict__ Qx,
                                                            double* __restrict__ Fx, double* __restrict__ part,
                                                            unsigned* __restrict__ ticket, unsigned long long* flag,
                                                            unsigned long long seq) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int RSF = pad_col(Rf), RSK = pad_col(K);
  double* YF = lds;              // [16][RSF]  y_f, then delta-y_f, then F x (a column is only touched by its own wave)
  double* Gx = YF + CB * RSF;    // [16][RSK]  gather of eta * yc
  double* G2 = Gx + CB * RSK;    // [16][RSK]  gather of eta * delta-yc
  double* Xt = G2 + CB * RSK;    // [16][RSK]  x
  double* Qt = Xt + CB * RSK;    // [16][RSK]  S0 x
  const int64_t c0 = (int64_t)blockIdx.x * CB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const bool cok = c0 + c < C;
  const int col = (int)(c0 + c), agent = col / D, dd_ = col - agent * D;
  const double hh = h * h;
  double ndy = 0.0, supp = 0.0, natdy = 0.0;
  PHASE_MARK(32);
  for (int k = kg; k < K; k += FT / CB) {
    double acc = 0.0, acc2 = 0.0, xv = 0.0;
    if (cok) {
      if (has_rows) {
        const int cell = cell_of(k, agent, K);
        const int t0 = cell_ptr[cell], t1 = cell_ptr[cell + 1];
        for (int t = t0; t < t1; ++t) acc += coef[(size_t)t * D + dd_] * gval[t];
        if (with_dy)
          for (int t = t0; t < t1; ++t) acc2 += coef[(size_t)t * D + dd_] * gval2[t];
      }
      xv = x[(int64_t)k * C + c0 + c];
    }
    Gx[c * RSK + k] = acc;
    G2[c * RSK + k] = acc2;
    Xt[c * RSK + k] = xv;
  }
  for (int r = kg; r < Rf; r += FT / CB) YF[c * RSF + r] = cok ? yf[(int64_t)r * C + c0 + c] : 0.0;
  __syncthreads();
  PHASE_MARK(33);
  double rd = 0.0, npx = 0.0, nat = 0.0;
  {
    double aty[E];
    column_At<E>(K, h, YF + wave * RSF, Gx + wave * RSK, aty);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = (64 * E - 1) - (lane * E + e);
      if (k < K) {
        const double px = 2.0 * Xt[wave * RSK + k];
        rd = fmax(rd, fabs(px + aty[e]));
        npx = fmax(npx, fabs(px));
        nat = fmax(nat, fabs(aty[e]));
      }
    }
  }
  PHASE_MARK(34);
  if (with_dy) {  // wave-uniform
    __syncthreads();
    for (int r = kg; r < Rf; r += FT / CB) {
      double dd = 0.0;
      if (cok) {
        const int64_t g = (int64_t)r * C + c0 + c;
        dd = dyf[g];
        ndy = fmax(ndy, fabs(dd));
        supp += uf[g] * fmax(dd, 0.0) + lf[g] * fmin(dd, 0.0);
      }
      YF[c * RSF + r] = dd;
    }
    __syncthreads();
    double atdy[E];
    column_At<E>(K, h, YF + wave * RSF, G2 + wave * RSK, atdy);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = (64 * E - 1) - (lane * E + e);
      if (k < K) natdy = fmax(natdy, fabs(atdy[e]));
    }
  }
  PHASE_MARK(35);
  {
    // F x and S0 x in ascending order (every lane of this wave is done with its column of YF)
    double* Yc = YF + wave * RSF;
    double pk[E], c1[E], c2[E], t1[E], t2[E], c1p[E], pn[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = lane * E + e;
      pk[e] = k < K ? Xt[wave * RSK + k] : 0.0;
    }
    wave_scan<E>(pk, c1, t1);
    wave_scan<E>(c1, t2, c2);
    wave_prev<E>(c1, c1p);
    wave_next<E>(pk, pn);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = lane * E + e;
      if (k < K) {
        Qt[wave * RSK + k] = hh * (c2[e] - 0.5 * c1p[e]);
        if (k < K - 1) Yc[k] = (pn[e] - pk[e]) / h;
        Yc[K - 1 + k] = pk[e];
        Yc[2 * K - 1 + k] = h * c1[e];
        Yc[3 * K - 1 + k] = hh * (c2[e] + 0.5 * c1[e]);
      }
    }
  }
  __syncthreads();
  PHASE_MARK(36);
  double rp = 0.0, nax = 0.0, nz = 0.0;
  if (cok) {
    for (int r = kg; r < Rf; r += FT / CB) {
      const int64_t g = (int64_t)r * C + c0 + c;
      const double a = YF[c * RSF + r], z = zf[g];
      Fx[g] = a;
      rp = fmax(rp, fabs(a - z));
      nax = fmax(nax, fabs(a));
      nz = fmax(nz, fabs(z));
    }
    for (int k = kg; k < K; k += FT / CB) Qx[(int64_t)k * C + c0 + c] = Qt[c * RSK + k];
  }
  PHASE_MARK(37);
  // nine workgroup reductions with one barrier: DPP inside the waves, lane 63 of every wave to LDS, 9 threads finish
  __shared__ double red[NWV][9];
  {
    const double v[9] = {wave_max_nn(rp),  wave_max_nn(nax), wave_max_nn(nz),     wave_max_nn(rd),   wave_max_nn(npx),
                         wave_max_nn(nat), wave_max_nn(ndy), wave_incl_sum(supp), wave_max_nn(natdy)};
    if (lane == 63) {
#pragma unroll
      for (int j = 0; j < 9; ++j) red[wave][j] = v[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int j = threadIdx.x;
    double t = 0.0;
    for (int w = 0; w < NWV; ++w) t = j == RS_SUPP ? t + red[w][j] : fmax(t, red[w][j]);
    part[(size_t)blockIdx.x * SCP_RESID_STRIDE + j] = t;
  }
  PHASE_MARK(38);
  if (flag) {  // no row kernel follows (QP#0): the LAST workgroup raises the host's completion word itself -- one launch less
    if (threadIdx.x < 9) __threadfence_system();  // (the partials live in mapped host memory)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
      *ticket = 0u;  // (the next launch on this stream starts after this kernel has ended)
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

constexpr int RESID_ROW_BLOCKS = 128;

template <int D>
__global__ __launch_bounds__(256) void cg1_resid_rows_kernel(int64_t nW, int64_t C, int with_dy, const int* __restrict__ wk,
                                                              const int* __restrict__ wi, const int* __restrict__ wj,
                                                              const double* __restrict__ weta,
                                                              const double* __restrict__ wl, const double* __restrict__ Qx,
                                                              const double* __restrict__ zc,
                                                              const double* __restrict__ dyc, double* __restrict__ part) {
  __shared__ double sm[4][5];
  double rp = 0.0, nax = 0.0, nz = 0.0, ndy = 0.0, supp = 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nW; n += (int64_t)RESID_ROW_BLOCKS * 256) {
    const int64_t bi = (int64_t)wk[n] * C + (int64_t)wi[n] * D;
    const int64_t bj = (int64_t)wk[n] * C + (int64_t)wj[n] * D;
    double a = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) a += weta[n * D + d] * (Qx[bi + d] - Qx[bj + d]);
    const double z = zc[n];
    rp = fmax(rp, fabs(a - z));
    nax = fmax(nax, fabs(a));
    nz = fmax(nz, fabs(z));
    if (with_dy) {
      const double dd = dyc[n];  // delta-y of the last iteration, already projected (u = +inf)
      ndy = fmax(ndy, fabs(dd));
      supp += wl[n] * dd;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    rp = fmax(rp, __shfl_xor(rp, o)); nax = fmax(nax, __shfl_xor(nax, o)); nz = fmax(nz, __shfl_xor(nz, o));
    ndy = fmax(ndy, __shfl_xor(ndy, o)); supp += __shfl_xor(supp, o);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[w][0] = rp; sm[w][1] = nax; sm[w][2] = nz; sm[w][3] = ndy; sm[w][4] = supp; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = part + (size_t)blockIdx.x * SCP_RESID_STRIDE;
    o[RS_RP] = fmax(fmax(sm[0][0], sm[1][0]), fmax(sm[2][0], sm[3][0]));
    o[RS_NAX] = fmax(fmax(sm[0][1], sm[1][1]), fmax(sm[2][1], sm[3][1]));
    o[RS_NZ] = fmax(fmax(sm[0][2], sm[1][2]), fmax(sm[2][2], sm[3][2]));
    o[RS_RD] = o[RS_NPX] = o[RS_NATY] = o[RS_NATDY] = 0.0;
    o[RS_NDY] = fmax(fmax(sm[0][3], sm[1][3]), fmax(sm[2][3], sm[3][3]));
    o[RS_SUPP] = (sm[0][4] + sm[1][4]) + (sm[2][4] + sm[3][4]);
  }
}

}  // namespace

namespace {
// last launch of a check: everything before it in the stream has completed and, the partials living in host memory,
// is visible to the host; the host spins on this word instead of sleeping in hipStreamSynchronize (25 us per check)
__global__ void check_done_kernel(unsigned long long* flag, unsigned long long seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

int scp_qp_fused_residuals(scp_qp* qp, bool with_dy) {
  const QpDev& d = qp->d;
  scp_ctx* ctx = qp->ctx;
  hipStream_t s = ctx->stream;
  const int K = qp->K, Rf = qp->Rf;
  const int64_t C = qp->C, nx = (int64_t)K * C;
  const int nblk = (int)((C + CB - 1) / CB);
  double* Qx = qp->qx_sel ? d.HQ : d.HQ + nx;
  double* part = qp->h_scal_dev + SL_COUNT;  // [nblk column blocks | RESID_ROW_BLOCKS row blocks][SCP_RESID_STRIDE]
  const int has_rows = qp->nW > 0 ? 1 : 0;
  if (has_rows)
    hipLaunchKernelGGL(csr_rowval_kernel, dim3((unsigned)((qp->nW + 255) / 256)), dim3(256), 0, s, qp->nW, 1, 0.0, d.zc,
                       d.yc, (const double*)nullptr, d.pos_i, d.pos_j, d.gval2, with_dy ? d.dyc : (const double*)nullptr,
                       with_dy ? d.gval3 : (double*)nullptr);
  const size_t lds = (size_t)CB * (pad_col(Rf) + 4 * pad_col(K)) * sizeof(double);
  const unsigned long long seq = ++qp->check_seq;
  unsigned long long* flag_dev = (unsigned long long*)(qp->h_scal_dev + SL_COUNT + SCP_RESID_CAP);
  unsigned* ticket = ctx->d_ticket + 1;  // ([0]: the small-problem pairwise passes; same stream, never concurrent)
  if (K <= 64) {
    int rc = allow_lds(qp, cg1_resid_col_kernel<1>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(cg1_resid_col_kernel<1>, dim3(nblk), dim3(FT), lds, s, K, Rf, C, qp->h, qp->N, qp->D, with_dy ? 1 : 0,
                       has_rows, d.x, d.zf, d.yf, d.lf, d.uf, d.dyf, d.cell_ptr, d.coef, d.gval2, d.gval3, Qx, d.fx, part,
                       ticket, has_rows ? nullptr : flag_dev, seq);
  } else {
    int rc = allow_lds(qp, cg1_resid_col_kernel<2>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(cg1_resid_col_kernel<2>, dim3(nblk), dim3(FT), lds, s, K, Rf, C, qp->h, qp->N, qp->D, with_dy ? 1 : 0,
                       has_rows, d.x, d.zf, d.yf, d.lf, d.uf, d.dyf, d.cell_ptr, d.coef, d.gval2, d.gval3, Qx, d.fx, part,
                       ticket, has_rows ? nullptr : flag_dev, seq);
  }
  double* rpart = part + (size_t)nblk * SCP_RESID_STRIDE;
  if (!has_rows) {
    // QP#0: no row partials
  } else if (qp->D == 2)
    hipLaunchKernelGGL(cg1_resid_rows_kernel<2>, dim3(RESID_ROW_BLOCKS), dim3(256), 0, s, qp->nW, C, with_dy ? 1 : 0, d.w_k,
                       d.w_i, d.w_j, d.w_eta, d.w_l, Qx, d.zc, d.dyc, rpart);
  else
    hipLaunchKernelGGL(cg1_resid_rows_kernel<3>, dim3(RESID_ROW_BLOCKS), dim3(256), 0, s, qp->nW, C, with_dy ? 1 : 0, d.w_k,
                       d.w_i, d.w_j, d.w_eta, d.w_l, Qx, d.zc, d.dyc, rpart);
  FUSED_LAUNCHED(qp);
  const int npart = nblk + (has_rows ? RESID_ROW_BLOCKS : 0);
  volatile unsigned long long* flag = (volatile unsigned long long*)(qp->h_scal + SL_COUNT + SCP_RESID_CAP);
  if (has_rows) {
    hipLaunchKernelGGL(check_done_kernel, dim3(1), dim3(1), 0, s, flag_dev, seq);
    FUSED_LAUNCHED(qp);
  }
  {
    if (!scp_wait_host_word(flag, seq, 20)) SCP_HIP_CHECK(ctx, hipStreamSynchronize(s));  // a fault surfaces here
    if (*flag != seq) return scp_fail(ctx, SCP_ERR_HIP, "fused check: completion flag not written");
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  double* hs = qp->h_scal;
  static const int slot[9] = {SL_RP, SL_NAX, SL_NZ, SL_RD, SL_NPX, SL_NATY, SL_NDY, SL_SUPP, SL_NATDY};
  for (int j = 0; j < 9; ++j) hs[slot[j]] = 0.0;
  for (int b = 0; b < npart; ++b) {  // fixed order: deterministic
    const double* o = hs + SL_COUNT + (size_t)b * SCP_RESID_STRIDE;
    for (int j = 0; j < 9; ++j) {
      if (j == RS_SUPP) hs[SL_SUPP] += o[j];
      else hs[slot[j]] = fmax(hs[slot[j]], o[j]);
    }
  }
  qp->qx_fresh = true;
  return SCP_OK;
}

#ifdef SCP_PHASE_PROFILE
// developer hook of the profiling build only (not declared in include/scp_hip.h)
extern "C" int scp_debug_phase_clocks(unsigned long long* out, int n) {
  if (n > 64) n = 64;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scp_phase_clk), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
