// Dense K-dimension products of the QP solver on the fp64 matrix cores of gfx950.
//
//   Y[R][C] = alpha * A[R][M] * X[M][C] + beta * Y[R][C]
//
// A is one of the small per-(agent,axis) blocks that every column shares (the inverse of the block-diagonal
// KKT matrix, the stacked fixed-row block F and its transpose, the Toeplitz block S0: SURVEY.md 7.1); X is a
// time-major [M][C] slab with C = N*D columns.  R, M <= 4K-1, C up to a few 10^4: these products are tiny
// (<= 0.1 GFLOP) and latency bound, MFMA is used because it is the natural unit for a 16x16 fp64 tile, not
// for throughput.
//
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md section 3):
//   A fragment: lane l holds A[i = l & 15][k = l >> 4]      (one f64)
//   B fragment: lane l holds B[k = l >> 4][j = l & 15]      (one f64)
//   C/D:        lane l, register r holds D[row = (l >> 4) + 4 r][col = l & 15], r = 0..3
#include "scp_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

// One wave = one 16x16 output tile; 4 waves per block side by side along C (the A tile is the same for all
// of them and stays in L1).
__global__ __launch_bounds__(256) void gemm_f64_mfma_kernel(int R, int M, int C, double alpha,
                                                             const double* __restrict__ A,
                                                             const double* __restrict__ X, double beta,
                                                             double* __restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row0 = blockIdx.y * 16;
  const int col0 = (blockIdx.x * 4 + wave) * 16;
  if (col0 >= C) return;  // whole wave leaves: no partial-exec MFMA
  const int li = lane & 15;
  const int lk = lane >> 4;
  const int arow = row0 + li;
  const int bcol = col0 + li;
  const bool arow_ok = arow < R;
  const bool bcol_ok = bcol < C;
  const double* Ap = A + (size_t)(arow_ok ? arow : 0) * M;
  const double* Xp = X + (bcol_ok ? bcol : 0);
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < M; k0 += 4) {
    const int kk = k0 + lk;
    const bool k_ok = kk < M;
    double a = (arow_ok && k_ok) ? Ap[kk] : 0.0;
    double b = (bcol_ok && k_ok) ? Xp[(size_t)kk * C] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  if (bcol_ok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + lk + 4 * r;
      if (row < R) {
        const size_t o = (size_t)row * C + bcol;
        double v = alpha * acc[r];
        if (beta != 0.0) v += beta * Y[o];
        Y[o] = v;
      }
    }
  }
}

// VALU reference path (settings.use_mfma = 0): one thread per output element, m ascending.
__global__ __launch_bounds__(256) void gemm_f64_valu_kernel(int R, int M, int C, double alpha,
                                                             const double* __restrict__ A,
                                                             const double* __restrict__ X, double beta,
                                                             double* __restrict__ Y) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int row = blockIdx.y;
  if (col >= C) return;
  const double* Ap = A + (size_t)row * M;
  double acc = 0.0;
  for (int m = 0; m < M; ++m) acc = fma(Ap[m], X[(size_t)m * C + col], acc);
  const size_t o = (size_t)row * C + col;
  double v = alpha * acc;
  if (beta != 0.0) v += beta * Y[o];
  Y[o] = v;
}

int scp_launch_gemm(scp_ctx* ctx, int use_mfma, int R, int M, int C, double alpha, const double* A,
                    const double* X, double beta, double* Y) {
  if (R <= 0 || C <= 0) return SCP_OK;
  SCP_REQUIRE(ctx, M > 0 && A && X && Y, "gemm: bad arguments R=%d M=%d C=%d", R, M, C);
  if (use_mfma) {
    dim3 grid(scp_cdiv(C, 64), scp_cdiv(R, 16));
    hipLaunchKernelGGL(gemm_f64_mfma_kernel, grid, dim3(256), 0, ctx->stream, R, M, C, alpha, A, X, beta, Y);
  } else {
    dim3 grid(scp_cdiv(C, 256), R);
    hipLaunchKernelGGL(gemm_f64_valu_kernel, grid, dim3(256), 0, ctx->stream, R, M, C, alpha, A, X, beta, Y);
  }
  SCP_HIP_CHECK(ctx, hipGetLastError());
  return SCP_OK;
}

extern "C" int scp_gemm_f64(scp_ctx* ctx, int use_mfma, int R, int M, int C, double alpha, const double* A,
                            const double* X, double beta, double* Y) {
  if (!ctx) return SCP_ERR_INVALID;
  return scp_launch_gemm(ctx, use_mfma, R, M, C, alpha, A, X, beta, Y);
}
