// Device body of scp_qp_reset's one-launch form, shared by qp_reset_kernel (scp_qp.hip) and by the kernel that resets a QP
// AND installs its first rows in the same launch (scp_qp_fused.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "scp_pair_device.h"

constexpr int RESET_COLS = 16;
// x0 ([N][K][D] or NULL: zeros) -> time-major x, z_f = F x, the carried F x and S0 x of the single-step pipeline (exact),
// y_f = 0.  16 columns per workgroup (256 columns of a 128-agent problem still make 16 workgroups), the x tile in LDS
// (reset_xs: [K][RESET_COLS] doubles), thread = (column, one sixteenth of the rows); the first 256 threads of the workgroup
// work (`active`), all of them must call (one barrier).  COH: S0 x is written through -- the last workgroup of the same
// kernel reads it.
template <bool COH>
__device__ inline void qp_reset_body(int tid, bool active, double* reset_xs, int N, int K, int D, int Rf,
                                     const double* __restrict__ x0, const double* __restrict__ F,
                                     const double* __restrict__ S0, double* __restrict__ x, double* __restrict__ zf,
                                     double* __restrict__ fx, double* __restrict__ Qx, double* __restrict__ yf) {
  constexpr int RG = 256 / RESET_COLS;
  const int64_t C = (int64_t)N * D;
  const int lc = tid & (RESET_COLS - 1), rg = tid / RESET_COLS;
  const int64_t c = (int64_t)blockIdx.x * RESET_COLS + lc;
  const bool live = active && c < C;
  const int64_t agent = live ? c / D : 0;
  const int dim = live ? (int)(c - agent * D) : 0;
  if (active) {
    for (int k = rg; k < K; k += RG) {
      const double v = (live && x0) ? x0[(agent * K + k) * D + dim] : 0.0;
      reset_xs[k * RESET_COLS + lc] = v;
      if (live) x[(int64_t)k * C + c] = v;
    }
  }
  __syncthreads();
  if (!active) return;
  // four rows per thread and pass: one LDS read of x[k] feeds four independent multiply-add chains (one row at a time was
  // a chain of K dependent loads + FMAs per row: 42 us at 1024 agents)
  constexpr int RB = 4;
  for (int r0 = rg * RB; r0 < Rf + K; r0 += RG * RB) {
    const double* __restrict__ row[RB];
    double acc[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = min(r0 + j, Rf + K - 1);
      row[j] = r < Rf ? F + (size_t)r * K : S0 + (size_t)(r - Rf) * K;
      acc[j] = 0.0;
    }
#pragma unroll 2
    for (int k = 0; k < K; ++k) {
      const double xv = reset_xs[k * RESET_COLS + lc];
#pragma unroll
      for (int j = 0; j < RB; ++j) acc[j] += row[j][k] * xv;
    }
    if (!live) continue;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = r0 + j;
      if (r >= Rf + K) break;
      if (r < Rf) {
        zf[(int64_t)r * C + c] = acc[j];
        fx[(int64_t)r * C + c] = acc[j];
        yf[(int64_t)r * C + c] = 0.0;
      } else if (COH) {
        store_coherent(Qx + (int64_t)(r - Rf) * C + c, acc[j]);
      } else {
        Qx[(int64_t)(r - Rf) * C + c] = acc[j];
      }
    }
  }
}
