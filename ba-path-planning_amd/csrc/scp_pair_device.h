// Device functions of the pairwise geometry shared by the translation units of libscp_hip.so (gfx950 only): the pairwise
// passes (scp_kernels.hip) and the kernel that installs recomputed working rows together with their incidence lists
// (scp_qp_fused.hip) must produce the same bits, so there is ONE definition of each.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// Data one workgroup of a kernel hands to its LAST workgroup (the one-launch pairwise passes of small problems, the kernel that
// resets a QP and installs its rows) (possibly on another XCD, behind
// another L2): written through to the device's coherence point and read past the local L2.  With every such store written
// through, "my stores have been performed" is a wait for their acknowledgements -- no L2 write-back (a __threadfence() per
// workgroup walks the L2 each time: 9 us at 200 workgroups).
__device__ inline void store_coherent(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double load_coherent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ inline void wait_stores_performed() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// Lexicographic pair index q -> (i, j), i < j.  Row i of the triangle starts at off(i) = i (2N - i - 1) / 2.
__device__ __host__ inline int64_t tri_off(int64_t i, int64_t N) { return i * (2 * N - i - 1) / 2; }

__device__ inline void decode_pair(int64_t q, int N, int& i, int& j) {
  const double b = 2.0 * N - 1.0;
  int64_t ii = (int64_t)((b - sqrt(b * b - 8.0 * (double)q)) * 0.5);
  if (ii < 0) ii = 0;
  if (ii > N - 2) ii = N - 2;
  while (tri_off(ii, N) > q) --ii;
  while (ii < N - 2 && tri_off(ii + 1, N) <= q) ++ii;
  i = (int)ii;
  j = (int)(q - tri_off(ii, N) + ii + 1);
}


// c_i[k] = p0 + (k h) v0: the free motion a row's lower bound is measured from (scp.py:543-549).  ONE definition for the
// prep kernel of the passes and for the row-recomputing add kernel, so that both produce the same bits.
__device__ inline double free_motion(double p0, double v0, int k, double h) { return p0 + ((double)k * h) * v0; }


// 1/sqrt(x) to fp64 accuracy from the hardware seed: two Newton steps
__device__ inline double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

template <int D>
struct Pt {
  double v[D];
};

template <int D>
__device__ inline Pt<D> load_pt(const double* base, int idx) {
  Pt<D> r;
  if (D == 2) {
    const double2 t = *reinterpret_cast<const double2*>(base + 2 * idx);
    r.v[0] = t.x;
    r.v[1] = t.y;
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) r.v[d] = base[D * idx + d];
  }
  return r;
}

// The same point through the SCALAR cache, for an index that is the same in all lanes of the wave (the caller made it so with
// readfirstlane): the result lives in SGPRs.  Constant address space = "not written while this kernel runs" (the slices
// are the prep kernel's output).
template <int D>
__device__ inline Pt<D> load_pt_uniform(const double* base, int idx_uniform) {
  typedef const double __attribute__((address_space(4))) cdouble;
  cdouble* q = (cdouble*)(uintptr_t)(base + (int64_t)D * idx_uniform);
  Pt<D> r;
#pragma unroll
  for (int d = 0; d < D; ++d) r.v[d] = q[d];
  return r;
}

// Geometry of one pair at one time step and the compact row derived from it -- the arithmetic of scp.py:498-509, :543-549.
// Shared by the pairwise passes and by add_rows_at_kernel (the row-free loop recomputes the selected rows with it): one
// definition, the same bits.
template <int D>
struct PairGeom {
  double diff[D], ss, inv, raw;
  bool deg;
};
template <int D>
__device__ inline PairGeom<D> pair_geom(const Pt<D>& Pi, const Pt<D>& Pj) {
  PairGeom<D> g;
  g.ss = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    g.diff[d] = Pi.v[d] - Pj.v[d];
    g.ss = fma(g.diff[d], g.diff[d], g.ss);
  }
  g.deg = g.ss < 1e-12;  // dist < 1e-6 (scp.py:503)
  g.inv = rsqrt_nr(fmax(g.ss, 1e-200));
  double raw = g.ss * g.inv;
  g.raw = fma(fma(-raw, raw, g.ss), 0.5 * g.inv, raw);  // one correction step: sqrt to < 1 ulp (0 stays 0)
  return g;
}
// eta (scp.py:509; the fixed direction e_0 and dist := 1 for a degenerate pair, scp.py:503-507), the distance the row uses,
// and  l = R + (eta.diff - dist) - eta.(c_i - c_j) = R - dist + eta.(Q_i - Q_j)  (scp.py:543-549)
template <int D>
__device__ inline void pair_row(const PairGeom<D>& g, const Pt<D>& Qi, const Pt<D>& Qj, double R, double (&eta)[D], double& l,
                                double& dist) {
  dist = g.deg ? 1.0 : g.raw;
  double qd = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double e_d = g.deg ? (d == 0 ? 1.0 : 0.0) : g.diff[d] * g.inv;
    eta[d] = e_d;
    qd = fma(e_d, Qi.v[d] - Qj.v[d], qd);
  }
  l = (R - dist) + qd;
}

// pair_row for a pair that is NOT degenerate (g.deg false): the same operations on the same operands without the selects
// -- the streaming kernel runs this on every row and repairs the (rare) degenerate ones with pair_row afterwards.
template <int D>
__device__ inline void pair_row_regular(const PairGeom<D>& g, const Pt<D>& Qi, const Pt<D>& Qj, double R, double (&eta)[D],
                                        double& l) {
  double qd = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double e_d = g.diff[d] * g.inv;
    eta[d] = e_d;
    qd = fma(e_d, Qi.v[d] - Qj.v[d], qd);
  }
  l = (R - g.raw) + qd;
}


// Working row `base + t` of a QP with eta / l RECOMPUTED from the linearisation point (scp_qp_add_rows_at; the row-free loop:
// scp_select_pairs wrote no rows): decode (k, i, j), the two positions of the pair at step k from pos_prev ([N][K][D]), then
// pair_geom / pair_row -- the very functions of the linearisation kernel, on the same operands (Q = P - free_motion as its
// prep kernel forms it): bit-identical eta and l.  z = max(A x, l), y = 0 as add_rows_kernel.
// COH: Qx was written by other workgroups of the SAME kernel (reset + install in one launch): read past the local L2.
template <int D, bool COH = false>
__device__ inline void add_row_at(int64_t t, int N, int K, int64_t C, int64_t pairs, int64_t base,
                                  const int64_t* __restrict__ rows, const double* __restrict__ pos_prev,
                                  const double* __restrict__ p0, const double* __restrict__ v0, double R, double h,
                                  const double* __restrict__ Qx, int64_t* __restrict__ w_row, int* __restrict__ wk,
                                  int* __restrict__ wi, int* __restrict__ wj, double* __restrict__ weta,
                                  double* __restrict__ wl, double* __restrict__ zc, double* __restrict__ yc) {
  const int64_t r = rows[t];
  const int k = (int)(r / pairs);
  int i, j;
  decode_pair(r % pairs, N, i, j);
  Pt<D> Pi, Pj, Qi, Qj;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    Pi.v[d] = pos_prev[((int64_t)i * K + k) * D + d];
    Pj.v[d] = pos_prev[((int64_t)j * K + k) * D + d];
    Qi.v[d] = Pi.v[d] - free_motion(p0[i * D + d], v0[i * D + d], k, h);
    Qj.v[d] = Pj.v[d] - free_motion(p0[j * D + d], v0[j * D + d], k, h);
  }
  const PairGeom<D> g = pair_geom<D>(Pi, Pj);
  double eta[D], l, dist;
  pair_row<D>(g, Qi, Qj, R, eta, l, dist);
  const int64_t o = base + t;
  w_row[o] = r;
  wk[o] = k;
  wi[o] = i;
  wj[o] = j;
  double ax = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    weta[o * D + d] = eta[d];
    const double qi = COH ? load_coherent(Qx + (int64_t)k * C + (int64_t)i * D + d) : Qx[(int64_t)k * C + (int64_t)i * D + d];
    const double qj = COH ? load_coherent(Qx + (int64_t)k * C + (int64_t)j * D + d) : Qx[(int64_t)k * C + (int64_t)j * D + d];
    ax += eta[d] * (qi - qj);
  }
  wl[o] = l;
  zc[o] = fmax(ax, l);
  yc[o] = 0.0;
}
