/*
 * CPU oracle in plain C -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library
 * (oracle/_build/libscp_oracle_c.so, built by oracle/Makefile).
 *
 * A single-threaded restatement of the same algorithm as oracle/scp_oracle.py + oracle/qp_oracle.py
 * (admm_structured), written the way a careful CPU implementation would be: time-contiguous columns, the fixed
 * rows applied as O(K) scans instead of dense K x K blocks, the working rows as gather/scatter.  It exists so
 * that bench.py's cpu_baseline is a fair single-core number rather than numpy overhead, and it is itself pinned
 * against the numpy oracle (tests/test_oracle_c.py), which is pinned against the reference's golden vectors.
 *
 * Reference lines: kinematics scp.py:371-397; linearisation scp.py:453-557; fixed rows scp.py:182-257;
 * QP definition scp.py:323-369, :399-451; OSQP's algorithm as in oracle/qp_oracle.py ("parity unpinned").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int N, K, D;
  double h, R;
  const double *p0, *v0, *pf, *vf;      /* [N][D] */
  const double *pos_min, *pos_max;      /* [D] */
  double vel_min, vel_max, acc_min, acc_max, jerk_min, jerk_max;
} oc_problem;

typedef struct {
  double rho, sigma, alpha, rho_eq_scale, eps_abs, eps_rel;
  int max_iter, check_termination, adaptive_rho, adaptive_rho_interval;
  double adaptive_rho_tolerance;
  int cg_iters;
  double margin, feas_tol;
  int max_rounds;
  double rho_col_scale; /* rho of the collision rows = rho * rho_col_scale */
  double eps_prim_inf;  /* OSQP's primal infeasibility tolerance */
  int check_fine;          /* adaptive check cadence (qp_oracle.Settings.check_fine): steps to the next check once the */
  double check_fine_ratio; /* residuals are within this factor of their tolerances, or rho has just changed; 0: fixed */
} oc_settings;

typedef struct {
  int status_val, iter, rounds, rho_updates, cg_total;
  int64_t working_rows;
  double r_prim, r_dual, rho;
} oc_info;

static int64_t tri_off(int64_t i, int64_t N) { return i * (2 * N - i - 1) / 2; }

static void decode_pair(int64_t q, int N, int* pi, int* pj) {
  int64_t i = (int64_t)(((2.0 * N - 1.0) - sqrt((2.0 * N - 1.0) * (2.0 * N - 1.0) - 8.0 * (double)q)) * 0.5);
  if (i < 0) i = 0;
  if (i > N - 2) i = N - 2;
  while (tri_off(i, N) > q) --i;
  while (i < N - 2 && tri_off(i + 1, N) <= q) ++i;
  *pi = (int)i;
  *pj = (int)(q - tri_off(i, N) + i + 1);
}

/* a4/a7: scp.py:371-397 (same summation order; compile without FMA contraction for bitwise equality) */
void oc_kinematics(const oc_problem* P, const double* acc, double* pos, double* vel) {
  const int N = P->N, K = P->K, D = P->D;
  const double h = P->h, hh = h * h;
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < K; ++k)
      for (int d = 0; d < D; ++d) {
        const double vi = P->v0[i * D + d];
        double v = vi;
        double p = P->p0[i * D + d] + (h * k) * vi;
        for (int j = 0; j < k; ++j) {
          const double a = acc[((size_t)i * K + j) * D + d];
          v = v + h * a;
          p = p + (hh * ((double)(k - j) - 0.5)) * a;
        }
        pos[((size_t)i * K + k) * D + d] = p;
        if (vel) vel[((size_t)i * K + k) * D + d] = v;
      }
}

/* a5: scp.py:453-557, compact rows eta[r][D], l[r], dist[r]; r = k*pairs + q */
void oc_linearize(const oc_problem* P, const double* pos, double* eta, double* l, double* dist_out) {
  const int N = P->N, K = P->K, D = P->D;
  const int64_t pairs = (int64_t)N * (N - 1) / 2;
  for (int k = 0; k < K; ++k) {
    int64_t r = (int64_t)k * pairs;
    const double kh = k * P->h;
    for (int i = 0; i < N; ++i)
      for (int j = i + 1; j < N; ++j, ++r) {
        double diff[3], ss = 0.0;
        for (int d = 0; d < D; ++d) {
          diff[d] = pos[((size_t)i * K + k) * D + d] - pos[((size_t)j * K + k) * D + d];
          ss += diff[d] * diff[d];
        }
        double dist = D == 2 ? hypot(diff[0], diff[1]) : sqrt(ss);  /* scp.py:501 */
        double e[3];
        if (dist < 1e-6) {  /* scp.py:503-507, fixed direction e_0 */
          dist = 1.0;
          for (int d = 0; d < D; ++d) e[d] = d == 0 ? 1.0 : 0.0;
        } else {
          for (int d = 0; d < D; ++d) e[d] = diff[d] / dist;
        }
        double ip = 0.0, iv = 0.0, lin = 0.0;
        for (int d = 0; d < D; ++d) {
          ip += e[d] * (P->p0[i * D + d] - P->p0[j * D + d]);
          iv += e[d] * (P->v0[i * D + d] - P->v0[j * D + d]);
          lin += e[d] * diff[d];
          eta[r * D + d] = e[d];
        }
        lin -= dist;
        l[r] = P->R + lin - (ip + iv * kh);
        if (dist_out) dist_out[r] = dist;
      }
  }
}

/* ---- column-contiguous helpers: col[c][k], c = i*D + d ------------------------------------------------ */
static void to_cols(int N, int K, int D, const double* x, double* c) {
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < K; ++k)
      for (int d = 0; d < D; ++d) c[((size_t)i * D + d) * K + k] = x[((size_t)i * K + k) * D + d];
}
static void from_cols(int N, int K, int D, const double* c, double* x) {
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < K; ++k)
      for (int d = 0; d < D; ++d) x[((size_t)i * K + k) * D + d] = c[((size_t)i * D + d) * K + k];
}

/* Q = S0 x per column: Q[0] = 0, Q[k] = sum_{m<k} h^2 (k-m-.5) x[m]  (double integration) */
static void s0_apply(int K, double h, const double* x, double* q) {
  double vsum = 0.0, p = 0.0;  /* vsum = h * sum_{m<k} x[m] */
  q[0] = 0.0;
  for (int k = 1; k < K; ++k) {
    p += h * vsum + 0.5 * h * h * x[k - 1];
    vsum += h * x[k - 1];
    q[k] = p;
  }
}
/* out[m] += sum_{k>m} h^2 (k-m-.5) g[k] */
static void s0t_apply_add(int K, double h, const double* g, double* out) {
  double s1 = 0.0, s2 = 0.0;  /* s1 = sum_{k>m} g[k], s2 = sum_{k>m} (k-m-.5) g[k] */
  for (int m = K - 2; m >= 0; --m) {
    s2 += s1 + 0.5 * g[m + 1];
    s1 += g[m + 1];
    out[m] += h * h * s2;
  }
}

/* fixed rows of one column: t = F x  (jerk K-1 | acc K | vel K | pos K) */
static void f_apply(int K, double h, const double* x, double* tj, double* ta, double* tv, double* tp) {
  double vs = 0.0, p = 0.0;
  for (int k = 0; k < K; ++k) {
    if (k < K - 1) tj[k] = (x[k + 1] - x[k]) / h;
    ta[k] = x[k];
    p += h * vs + 0.5 * h * h * x[k];  /* pos row k = state k+1 */
    vs += h * x[k];
    tv[k] = vs;
    tp[k] = p;
  }
}
/* out = F^T w */
static void ft_apply(int K, double h, const double* wj, const double* wa, const double* wv, const double* wp,
                     double* out) {
  double s1 = 0.0, s2 = 0.0, sv = 0.0;
  for (int m = K - 1; m >= 0; --m) {
    /* pos: sum_{k>=m} h^2 (k-m+.5) wp[k] ; vel: h sum_{k>=m} wv[k] */
    s2 += s1 + 0.5 * wp[m];
    s1 += wp[m];
    sv += wv[m];
    double v = wa[m] + h * sv + h * h * s2;
    if (m < K - 1) v -= wj[m] / h;
    if (m > 0) v += wj[m - 1] / h;
    out[m] = v;
    s2 += 0.0;
    /* shift for next m: every (k-m+.5) grows by 1 -> add s1 (done at loop top through s2 += s1) */
  }
}

typedef struct {
  int N, K, D, C;
  double h;
  int64_t nW;
  const int *wk, *wi, *wj;
  const double* weta;
  double* Q; /* [C][K] scratch */
  double* G; /* [C][K] scratch */
} oc_rows;

/* out[n] = eta_n . ((S0 v)_i[k] - (S0 v)_j[k]) */
static void rows_apply(oc_rows* R, const double* v, double* out) {
  const int K = R->K, D = R->D;
  for (int c = 0; c < R->C; ++c) s0_apply(K, R->h, v + (size_t)c * K, R->Q + (size_t)c * K);
  for (int64_t n = 0; n < R->nW; ++n) {
    double a = 0.0;
    for (int d = 0; d < D; ++d)
      a += R->weta[n * D + d] * (R->Q[((size_t)R->wi[n] * D + d) * K + R->wk[n]] - R->Q[((size_t)R->wj[n] * D + d) * K + R->wk[n]]);
    out[n] = a;
  }
}
/* out += A_W^T g */
static void rows_apply_T_add(oc_rows* R, const double* g, double* out) {
  const int K = R->K, D = R->D;
  memset(R->G, 0, sizeof(double) * (size_t)R->C * K);
  for (int64_t n = 0; n < R->nW; ++n)
    for (int d = 0; d < D; ++d) {
      const double c = R->weta[n * D + d] * g[n];
      R->G[((size_t)R->wi[n] * D + d) * K + R->wk[n]] += c;
      R->G[((size_t)R->wj[n] * D + d) * K + R->wk[n]] -= c;
    }
  for (int c = 0; c < R->C; ++c) s0t_apply_add(K, R->h, R->G + (size_t)c * K, out + (size_t)c * K);
}

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* dense K x K helpers (KKT block and its inverse) */
static void build_hf(int K, double h, double sigma, double rho, double eq, double* Hf) {
  /* Hf = (2+sigma) I + rho F^T diag(w) F, column by column through the structured operators */
  double* e = calloc(K, sizeof(double));
  double *tj = malloc(sizeof(double) * K), *ta = malloc(sizeof(double) * K), *tv = malloc(sizeof(double) * K),
         *tp = malloc(sizeof(double) * K), *o = malloc(sizeof(double) * K);
  for (int b = 0; b < K; ++b) {
    memset(e, 0, sizeof(double) * K);
    e[b] = 1.0;
    f_apply(K, h, e, tj, ta, tv, tp);
    for (int k = 0; k < K; ++k) {
      const double w = k == K - 1 ? eq : 1.0;
      if (k < K - 1) tj[k] *= rho;
      ta[k] *= rho;
      tv[k] *= rho * w;
      tp[k] *= rho * w;
    }
    tj[K - 1] = 0.0;
    ft_apply(K, h, tj, ta, tv, tp, o);
    for (int a = 0; a < K; ++a) Hf[(size_t)a * K + b] = o[a] + (a == b ? 2.0 + sigma : 0.0);
  }
  free(e); free(tj); free(ta); free(tv); free(tp); free(o);
}
static void spd_inverse(int K, const double* A, double* Ainv) {
  double* L = malloc(sizeof(double) * K * K);
  memcpy(L, A, sizeof(double) * K * K);
  for (int j = 0; j < K; ++j) {  /* Cholesky, lower */
    for (int k = 0; k < j; ++k)
      for (int i = j; i < K; ++i) L[(size_t)i * K + j] -= L[(size_t)i * K + k] * L[(size_t)j * K + k];
    const double dj = sqrt(L[(size_t)j * K + j]);
    for (int i = j; i < K; ++i) L[(size_t)i * K + j] /= dj;
  }
  double* y = malloc(sizeof(double) * K);
  for (int c = 0; c < K; ++c) {
    for (int i = 0; i < K; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[(size_t)i * K + k] * y[k];
      y[i] = s / L[(size_t)i * K + i];
    }
    for (int i = K - 1; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < K; ++k) s -= L[(size_t)k * K + i] * Ainv[(size_t)k * K + c];
      Ainv[(size_t)i * K + c] = s / L[(size_t)i * K + i];
    }
  }
  free(L); free(y);
}
static void dense_apply(int K, int C, const double* M, const double* v, double* out) {
  for (int c = 0; c < C; ++c) {
    const double* vc = v + (size_t)c * K;
    double* oc = out + (size_t)c * K;
    for (int a = 0; a < K; ++a) {
      const double* row = M + (size_t)a * K;
      double s = 0.0;
      for (int b = 0; b < K; ++b) s += row[b] * vc[b];
      oc[a] = s;
    }
  }
}

/* Joint QP by working-set ADMM; eta/l_col/dist over ALL rows (NULL -> QP#0).  x0, x_out in [N][K][D]. */
int oc_admm(const oc_problem* P, const double* eta, const double* l_col, const double* dist, const double* x0,
            const oc_settings* st, double* x_out, oc_info* info) {
  const int N = P->N, K = P->K, D = P->D, C = N * D;
  const double h = P->h;
  const int64_t pairs = (int64_t)N * (N - 1) / 2, m_col = eta ? pairs * K : 0;
  const size_t nx = (size_t)C * K;
#define NEWV(n) ((double*)calloc((n) ? (n) : 1, sizeof(double)))
  double *x = NEWV(nx), *xt = NEWV(nx), *rhs = NEWV(nx), *r = NEWV(nx), *p = NEWV(nx), *zz = NEWV(nx), *Hp = NEWV(nx);
  double *zj = NEWV(nx), *za = NEWV(nx), *zv = NEWV(nx), *zp = NEWV(nx);
  double *yj = NEWV(nx), *ya = NEWV(nx), *yv = NEWV(nx), *yp = NEWV(nx);
  double *tj = NEWV(nx), *ta = NEWV(nx), *tv = NEWV(nx), *tp = NEWV(nx);
  double *lv = NEWV(nx), *uv = NEWV(nx), *lp = NEWV(nx), *up = NEWV(nx);
  double *Hf = NEWV((size_t)K * K), *Minv = NEWV((size_t)K * K), *Qs = NEWV(nx), *Gs = NEWV(nx);
  double *sj = NEWV(nx), *sa = NEWV(nx), *sv = NEWV(nx), *sp = NEWV(nx), *sc = NULL; /* y snapshots -> dy */
  /* bounds (scp.py:206-257), column layout */
  for (int i = 0; i < N; ++i)
    for (int d = 0; d < D; ++d) {
      const int c = i * D + d;
      const double v0 = P->v0[c], p0 = P->p0[c];
      for (int k = 0; k < K; ++k) {
        const double off = p0 + (h * (k + 1)) * v0;
        if (k < K - 1) {
          lv[(size_t)c * K + k] = P->vel_min - v0;
          uv[(size_t)c * K + k] = P->vel_max - v0;
          lp[(size_t)c * K + k] = P->pos_min[d] - off;
          up[(size_t)c * K + k] = P->pos_max[d] - off;
        } else {
          lv[(size_t)c * K + k] = uv[(size_t)c * K + k] = P->vf[c] - v0;
          lp[(size_t)c * K + k] = up[(size_t)c * K + k] = P->pf[c] - off;
        }
      }
    }
  if (x0) to_cols(N, K, D, x0, x);
  for (int c = 0; c < C; ++c)
    f_apply(K, h, x + (size_t)c * K, zj + (size_t)c * K, za + (size_t)c * K, zv + (size_t)c * K, zp + (size_t)c * K);

  /* working set */
  int64_t nW = 0, capW = 1024;
  int64_t* wrow = malloc(sizeof(int64_t) * capW);
  uint8_t* inW = m_col ? calloc((size_t)m_col, 1) : NULL;
  if (eta)
    for (int64_t rr = 0; rr < m_col; ++rr)
      if (dist[rr] - P->R < st->margin) {
        if (nW == capW) wrow = realloc(wrow, sizeof(int64_t) * (capW *= 2));
        wrow[nW++] = rr;
        inW[rr] = 1;
      }
  int *wk = NULL, *wi = NULL, *wj = NULL;
  double *weta = NULL, *wl = NULL, *zc = NULL, *yc = NULL, *tc = NULL, *gc = NULL, *ax_all = NULL;
  oc_rows R = {N, K, D, C, h, 0, NULL, NULL, NULL, NULL, Qs, Gs};
#define SETUP_ROWS()                                                                                     \
  do {                                                                                                   \
    wk = realloc(wk, sizeof(int) * (nW + 1)); wi = realloc(wi, sizeof(int) * (nW + 1));                  \
    wj = realloc(wj, sizeof(int) * (nW + 1)); weta = realloc(weta, sizeof(double) * (nW + 1) * D);       \
    wl = realloc(wl, sizeof(double) * (nW + 1)); tc = realloc(tc, sizeof(double) * (nW + 1));            \
    gc = realloc(gc, sizeof(double) * (nW + 1));                                                         \
    for (int64_t n = 0; n < nW; ++n) {                                                                   \
      wk[n] = (int)(wrow[n] / pairs);                                                                    \
      decode_pair(wrow[n] % pairs, N, &wi[n], &wj[n]);                                                   \
      for (int d = 0; d < D; ++d) weta[n * D + d] = eta[wrow[n] * D + d];                                \
      wl[n] = l_col[wrow[n]];                                                                            \
    }                                                                                                    \
    R.nW = nW; R.wk = wk; R.wi = wi; R.wj = wj; R.weta = weta;                                           \
  } while (0)
  if (eta) {
    SETUP_ROWS();
    zc = NEWV(nW);
    yc = NEWV(nW);
    rows_apply(&R, x, zc);
    for (int64_t n = 0; n < nW; ++n) zc[n] = dmax(zc[n], wl[n]);
  }

  double rho = st->rho;
  int total_it = 0, status = -2, rho_updates = 0, cg_total = 0, rounds = 0;
  double rp = INFINITY, rd = INFINITY;
  for (int rnd = 0; rnd < st->max_rounds; ++rnd) {
    rounds = rnd + 1;
    build_hf(K, h, st->sigma, rho, st->rho_eq_scale, Hf);
    spd_inverse(K, Hf, Minv);
    status = -2;
    int it = 0;
    int cad = st->check_termination; /* steps between two checks: every round starts on the coarse cadence */
    const int fine = (st->check_fine > 0 && st->check_fine < st->check_termination && st->check_termination % st->check_fine == 0 &&
                      nW > 0 && C <= 4096) ? st->check_fine : 0; /* (when it divides the coarse one, not to QP#0, up to 4096
                                                                     columns: qp_oracle.py) */
    while (total_it < st->max_iter) {
      ++it;
      ++total_it;
      const double rhoc = rho * st->rho_col_scale;
      /* rhs = sigma x + F^T (R z - y) + A_W^T (rho zc - yc) */
      for (int c = 0; c < C; ++c) {
        const size_t o = (size_t)c * K;
        for (int k = 0; k < K; ++k) {
          const double w = k == K - 1 ? rho * st->rho_eq_scale : rho;
          tj[o + k] = rho * zj[o + k] - yj[o + k];
          ta[o + k] = rho * za[o + k] - ya[o + k];
          tv[o + k] = w * zv[o + k] - yv[o + k];
          tp[o + k] = w * zp[o + k] - yp[o + k];
        }
        ft_apply(K, h, tj + o, ta + o, tv + o, tp + o, rhs + o);
        for (int k = 0; k < K; ++k) rhs[o + k] += st->sigma * x[o + k];
      }
      if (nW > 0) {
        for (int64_t n = 0; n < nW; ++n) gc[n] = rhoc * zc[n] - yc[n];
        rows_apply_T_add(&R, gc, rhs);
        /* PCG, preconditioner Minv, warm start xt = x */
#define HMUL(v, out)                                                              \
  do {                                                                            \
    dense_apply(K, C, Hf, (v), (out));                                            \
    rows_apply(&R, (v), tc);                                                      \
    for (int64_t n_ = 0; n_ < nW; ++n_) tc[n_] *= rhoc;                           \
    rows_apply_T_add(&R, tc, (out));                                              \
  } while (0)
        memcpy(xt, x, sizeof(double) * nx);
        HMUL(xt, Hp);
        for (size_t e = 0; e < nx; ++e) r[e] = rhs[e] - Hp[e];
        dense_apply(K, C, Minv, r, zz);
        memcpy(p, zz, sizeof(double) * nx);
        double rz = 0.0;
        for (size_t e = 0; e < nx; ++e) rz += r[e] * zz[e];
        for (int ci = 0; ci < st->cg_iters; ++ci) {
          HMUL(p, Hp);
          double pHp = 0.0;
          for (size_t e = 0; e < nx; ++e) pHp += p[e] * Hp[e];
          if (pHp <= 0.0 || rz == 0.0) break;
          const double a = rz / pHp;
          for (size_t e = 0; e < nx; ++e) {
            xt[e] += a * p[e];
            r[e] -= a * Hp[e];
          }
          dense_apply(K, C, Minv, r, zz);
          double rzn = 0.0;
          for (size_t e = 0; e < nx; ++e) rzn += r[e] * zz[e];
          const double beta = rzn / rz;
          for (size_t e = 0; e < nx; ++e) p[e] = zz[e] + beta * p[e];
          rz = rzn;
          ++cg_total;
        }
      } else {
        dense_apply(K, C, Minv, rhs, xt);
      }
      /* z~ = A x~, relaxation, projection, duals */
      const double al = st->alpha;
      const int will_check = (it % cad == 0) || total_it >= st->max_iter;
      if (will_check) {
        memcpy(sj, yj, sizeof(double) * nx); memcpy(sa, ya, sizeof(double) * nx);
        memcpy(sv, yv, sizeof(double) * nx); memcpy(sp, yp, sizeof(double) * nx);
        if (nW > 0) { sc = realloc(sc, sizeof(double) * nW); memcpy(sc, yc, sizeof(double) * nW); }
      }
      for (int c = 0; c < C; ++c) {
        const size_t o = (size_t)c * K;
        f_apply(K, h, xt + o, tj + o, ta + o, tv + o, tp + o);
        for (int k = 0; k < K; ++k) {
          const double w = k == K - 1 ? rho * st->rho_eq_scale : rho;
          double zh, zn;
          if (k < K - 1) {
            zh = al * tj[o + k] + (1 - al) * zj[o + k];
            zn = dmin(dmax(zh + yj[o + k] / rho, P->jerk_min), P->jerk_max);
            yj[o + k] += rho * (zh - zn);
            zj[o + k] = zn;
          }
          zh = al * ta[o + k] + (1 - al) * za[o + k];
          zn = dmin(dmax(zh + ya[o + k] / rho, P->acc_min), P->acc_max);
          ya[o + k] += rho * (zh - zn);
          za[o + k] = zn;
          zh = al * tv[o + k] + (1 - al) * zv[o + k];
          zn = dmin(dmax(zh + yv[o + k] / w, lv[o + k]), uv[o + k]);
          yv[o + k] += w * (zh - zn);
          zv[o + k] = zn;
          zh = al * tp[o + k] + (1 - al) * zp[o + k];
          zn = dmin(dmax(zh + yp[o + k] / w, lp[o + k]), up[o + k]);
          yp[o + k] += w * (zh - zn);
          zp[o + k] = zn;
        }
      }
      if (nW > 0) {
        rows_apply(&R, xt, tc);
        for (int64_t n = 0; n < nW; ++n) {
          const double zh = al * tc[n] + (1 - al) * zc[n];
          const double zn = dmax(zh + yc[n] / rhoc, wl[n]);
          yc[n] += rhoc * (zh - zn);
          zc[n] = zn;
        }
      }
      for (size_t e = 0; e < nx; ++e) x[e] = al * xt[e] + (1 - al) * x[e];

      if (it % cad == 0 || total_it >= st->max_iter) {
        double nAx = 0.0, nz = 0.0, nPx = 0.0, nATy = 0.0;
        rp = rd = 0.0;
        for (int c = 0; c < C; ++c) {
          const size_t o = (size_t)c * K;
          f_apply(K, h, x + o, tj + o, ta + o, tv + o, tp + o);
          for (int k = 0; k < K; ++k) {
            if (k < K - 1) {
              rp = dmax(rp, fabs(tj[o + k] - zj[o + k])); nAx = dmax(nAx, fabs(tj[o + k])); nz = dmax(nz, fabs(zj[o + k]));
            }
            rp = dmax(rp, fabs(ta[o + k] - za[o + k])); nAx = dmax(nAx, fabs(ta[o + k])); nz = dmax(nz, fabs(za[o + k]));
            rp = dmax(rp, fabs(tv[o + k] - zv[o + k])); nAx = dmax(nAx, fabs(tv[o + k])); nz = dmax(nz, fabs(zv[o + k]));
            rp = dmax(rp, fabs(tp[o + k] - zp[o + k])); nAx = dmax(nAx, fabs(tp[o + k])); nz = dmax(nz, fabs(zp[o + k]));
          }
          ft_apply(K, h, yj + o, ya + o, yv + o, yp + o, rhs + o);  /* rhs := A^T y */
        }
        if (nW > 0) {
          rows_apply(&R, x, tc);
          for (int64_t n = 0; n < nW; ++n) {
            rp = dmax(rp, fabs(tc[n] - zc[n])); nAx = dmax(nAx, fabs(tc[n])); nz = dmax(nz, fabs(zc[n]));
          }
          rows_apply_T_add(&R, yc, rhs);
        }
        for (size_t e = 0; e < nx; ++e) {
          rd = dmax(rd, fabs(2.0 * x[e] + rhs[e])); nPx = dmax(nPx, fabs(2.0 * x[e])); nATy = dmax(nATy, fabs(rhs[e]));
        }
        {
          const double tol_p = st->eps_abs + st->eps_rel * dmax(nAx, nz), tol_d = st->eps_abs + st->eps_rel * dmax(nPx, nATy);
          if (rp <= tol_p && rd <= tol_d) {
            status = 1;
            break;
          }
          if (fine) /* close to the tolerances: look again soon */
            cad = (rp < st->check_fine_ratio * tol_p && rd < st->check_fine_ratio * tol_d) ? fine : st->check_termination;
        }
        /* OSQP at max_iter: 10 x the tolerances -> "solved inaccurate" (scp.py:363, :446 accept it) */
        if (total_it >= st->max_iter && rp <= 10.0 * (st->eps_abs + st->eps_rel * dmax(nAx, nz)) &&
            rd <= 10.0 * (st->eps_abs + st->eps_rel * dmax(nPx, nATy)))
          status = 2;
        { /* primal infeasibility certificate (OSQP is_primal_infeasible), as in qp_oracle.admm_structured */
          double ndy = 0.0, supp = 0.0;
          for (int c = 0; c < C; ++c) {
            const size_t o = (size_t)c * K;
            for (int k = 0; k < K; ++k) {
              double d;
              if (k < K - 1) {
                d = sj[o + k] = yj[o + k] - sj[o + k]; ndy = dmax(ndy, fabs(d));
                supp += P->jerk_max * dmax(d, 0.0) + P->jerk_min * dmin(d, 0.0);
              } else sj[o + k] = 0.0;
              d = sa[o + k] = ya[o + k] - sa[o + k]; ndy = dmax(ndy, fabs(d));
              supp += P->acc_max * dmax(d, 0.0) + P->acc_min * dmin(d, 0.0);
              d = sv[o + k] = yv[o + k] - sv[o + k]; ndy = dmax(ndy, fabs(d));
              supp += uv[o + k] * dmax(d, 0.0) + lv[o + k] * dmin(d, 0.0);
              d = sp[o + k] = yp[o + k] - sp[o + k]; ndy = dmax(ndy, fabs(d));
              supp += up[o + k] * dmax(d, 0.0) + lp[o + k] * dmin(d, 0.0);
            }
          }
          for (int64_t n = 0; n < nW; ++n) {
            const double d = sc[n] = dmin(yc[n] - sc[n], 0.0);
            ndy = dmax(ndy, fabs(d));
            supp += wl[n] * d;
          }
          if (ndy > st->eps_prim_inf && supp < -st->eps_prim_inf * ndy) {
            double nat = 0.0;
            for (int c = 0; c < C; ++c) {
              const size_t o = (size_t)c * K;
              ft_apply(K, h, sj + o, sa + o, sv + o, sp + o, rhs + o);
            }
            if (nW > 0) rows_apply_T_add(&R, sc, rhs);
            for (size_t e = 0; e < nx; ++e) nat = dmax(nat, fabs(rhs[e]));
            if (nat < st->eps_prim_inf * ndy) {
              status = -3;
              break;
            }
          }
        }
        if (st->adaptive_rho && it % st->adaptive_rho_interval == 0) {
          const double prim = rp / dmax(dmax(nAx, nz), 1e-10), dual = rd / dmax(dmax(nPx, nATy), 1e-10);
          double nr = rho * sqrt(prim / dmax(dual, 1e-10));
          nr = dmin(dmax(nr, 1e-6), 1e6);
          nr = exp2(round(4.0 * log2(nr)) / 4.0); /* geometric grid, see qp_oracle.py */
          if (nr > rho * st->adaptive_rho_tolerance || nr < rho / st->adaptive_rho_tolerance) {
            rho = nr;
            build_hf(K, h, st->sigma, rho, st->rho_eq_scale, Hf);
            spd_inverse(K, Hf, Minv);
            ++rho_updates;
            if (fine) cad = fine;
          }
        }
      }
    }
    if (!eta || status == -3) break;
    /* constraint generation: every row outside W checked at the ADMM solution */
    if (!ax_all) ax_all = malloc(sizeof(double) * (size_t)m_col);
    for (int c = 0; c < C; ++c) s0_apply(K, h, x + (size_t)c * K, Qs + (size_t)c * K);
    int64_t added = 0, nW_old = nW;
    for (int k = 0; k < K; ++k) {
      int64_t rr = (int64_t)k * pairs;
      for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j, ++rr) {
          if (inW[rr]) continue;
          double a = 0.0;
          for (int d = 0; d < D; ++d)
            a += eta[rr * D + d] * (Qs[((size_t)i * D + d) * K + k] - Qs[((size_t)j * D + d) * K + k]);
          if (a < l_col[rr] - st->feas_tol) {
            if (nW == capW) wrow = realloc(wrow, sizeof(int64_t) * (capW *= 2));
            wrow[nW++] = rr;
            inW[rr] = 1;
            ax_all[added++] = a;
          }
        }
    }
    if (added == 0 || total_it >= st->max_iter) break;
    /* new rows: z = max(Ax, l), y = 0 (appended; order does not change the mathematics) */
    zc = realloc(zc, sizeof(double) * nW);
    yc = realloc(yc, sizeof(double) * nW);
    SETUP_ROWS();
    for (int64_t n = nW_old; n < nW; ++n) {
      zc[n] = dmax(ax_all[n - nW_old], wl[n]);
      yc[n] = 0.0;
    }
  }
  from_cols(N, K, D, x, x_out);
  info->status_val = status; info->iter = total_it; info->rounds = rounds; info->rho_updates = rho_updates;
  info->cg_total = cg_total; info->working_rows = nW; info->r_prim = rp; info->r_dual = rd; info->rho = rho;
  free(x); free(xt); free(rhs); free(r); free(p); free(zz); free(Hp); free(zj); free(za); free(zv); free(zp);
  free(yj); free(ya); free(yv); free(yp); free(tj); free(ta); free(tv); free(tp); free(lv); free(uv); free(lp);
  free(up); free(Hf); free(Minv); free(Qs); free(Gs); free(wrow); free(inW); free(wk); free(wi); free(wj);
  free(weta); free(wl); free(zc); free(yc); free(tc); free(gc); free(ax_all); free(sj); free(sa); free(sv); free(sp); free(sc);
  return 0;
}
