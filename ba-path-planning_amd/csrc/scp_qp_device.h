// Device-side helpers shared by the column kernels of the joint QP (scp_qp_fused.hip, scp_qp_persist.hip): wave-wide
// scans on data-parallel-primitive (DPP) moves, MFMA operand prefetch, LDS padding.  gfx950 only.
#pragma once
#include "scp_qp_internal.h"

namespace scpdev {

constexpr int CB = 16;    // columns per workgroup of the column kernels (one MFMA tile wide)
constexpr int CHB = 16;   // operand registers (k steps) held at once by a tile product

typedef double double4_t __attribute__((ext_vector_type(4)));

// Operands of the first CH k-steps of row tile t of a packed matrix (QpDev::pMinv ...: [row tile][k step][lane]):
// fetched at kernel entry, long before the vector they multiply exists.
template <int CH>
__device__ inline void tile_prefetch(const double* __restrict__ P, int nks, int t, int ks0, int ks1, double (&a)[CH]) {
  const double* Ap = P + (size_t)t * nks * 64 + (threadIdx.x & 63);
#pragma unroll
  for (int s = 0; s < CH; ++s) a[s] = Ap[(size_t)min(ks0 + s, ks1 - 1) * 64];
}

// ---- wave-wide prefix sums over the time index: the integrator blocks V, S, S0 and their transposes are first and
// second cumulative sums.  DPP moves (no LDS round trip, unlike __shfl): lanes without a source read 0 -----------------
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_mov0(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
  return __hiloint2double(hi, lo);
}
// inclusive sum over the 64 lanes: row_shr 1, 2, 4, 8, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2, 3
__device__ inline double wave_incl_sum(double v) {
  v += dpp_mov0<0x111, 0xF>(v);
  v += dpp_mov0<0x112, 0xF>(v);
  v += dpp_mov0<0x114, 0xF>(v);
  v += dpp_mov0<0x118, 0xF>(v);
  v += dpp_mov0<0x142, 0xA>(v);
  v += dpp_mov0<0x143, 0xC>(v);
  return v;
}
// maximum of non-negative values over the 64 lanes, valid in lane 63
__device__ inline double wave_max_nn(double v) {
  v = fmax(v, dpp_mov0<0x111, 0xF>(v));
  v = fmax(v, dpp_mov0<0x112, 0xF>(v));
  v = fmax(v, dpp_mov0<0x114, 0xF>(v));
  v = fmax(v, dpp_mov0<0x118, 0xF>(v));
  v = fmax(v, dpp_mov0<0x142, 0xA>(v));
  v = fmax(v, dpp_mov0<0x143, 0xC>(v));
  return v;
}
__device__ inline double lane_below(double v) { return dpp_mov0<0x138, 0xF>(v); }  // wave_shr:1 (lane 0 <- 0)
__device__ inline double lane_above(double v) { return dpp_mov0<0x130, 0xF>(v); }  // wave_shl:1 (lane 63 <- 0)

__device__ inline double read_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// inclusive SUFFIX sum over the 64 lanes (lane l <- sum of lanes >= l): row_shl 1, 2, 4, 8 inside the rows of 16, then
// the totals of the higher rows (lanes 16, 32, 48 after the row scans) are added through scalar registers
__device__ inline double wave_incl_rsum(double v) {
  v += dpp_mov0<0x101, 0xF>(v);
  v += dpp_mov0<0x102, 0xF>(v);
  v += dpp_mov0<0x104, 0xF>(v);
  v += dpp_mov0<0x108, 0xF>(v);
  const double t1 = read_lane(v, 16), t2 = read_lane(v, 32), t3 = read_lane(v, 48);
  const int lane = threadIdx.x & 63;
  const double t23 = t2 + t3;
  const double add = lane < 16 ? t1 + t23 : (lane < 32 ? t23 : (lane < 48 ? t3 : 0.0));
  return v + add;
}

// inclusive and exclusive prefix sums over the index i = lane E + e (ascending)
template <int E>
__device__ inline void wave_scan(const double (&v)[E], double (&incl)[E], double (&excl)[E]) {
  double run = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    excl[e] = run;
    run += v[e];
    incl[e] = run;
  }
  const double off = lane_below(wave_incl_sum(run));  // total of the lower lanes
#pragma unroll
  for (int e = 0; e < E; ++e) {
    incl[e] += off;
    excl[e] += off;
  }
}
// value at index i - 1 / i + 1 (0 outside)
template <int E>
__device__ inline void wave_prev(const double (&v)[E], double (&o)[E]) {
  o[0] = lane_below(v[E - 1]);
#pragma unroll
  for (int e = 1; e < E; ++e) o[e] = v[e - 1];
}
template <int E>
__device__ inline void wave_next(const double (&v)[E], double (&o)[E]) {
  o[E - 1] = lane_above(v[0]);
#pragma unroll
  for (int e = 0; e + 1 < E; ++e) o[e] = v[e + 1];
}

// per-column LDS rows are padded to a length = 2 (mod 32) doubles: the (column, time) accesses of the coalesced
// global <-> LDS copies (16 columns x 2 steps per half wave) and of the MFMA operand reads then hit 32 distinct banks
__host__ __device__ inline int pad_col(int n) { return ((n + 29) / 32) * 32 + 2; }

// incidence-list cell of (time step k, agent): agent-major, so that the entries of a block of consecutive agents are one
// contiguous range (the persistent kernel keeps them in LDS)
__host__ __device__ inline int cell_of(int k, int agent, int K) { return agent * K + k; }

}  // namespace scpdev
