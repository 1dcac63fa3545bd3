"""CPU oracle for the QP half of the SCP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED.  The reference solves its QPs with the third-party package `osqp` (constraint
`osqp>=0.6`, no lock file, pyproject.toml:19; call sites scp.py:326, :360-362, :441-445).  Its source
is not part of /root/reference, the wheel is not installed in this image and there is no package
index, so none of OSQP's arithmetic can be executed here and the reference holds no tests or golden
vectors at that boundary.  What pins this file instead:

  * the QP *definition* follows the reference call sites exactly: P = 2I, q = 0, constraint stack
    [jerk; acc; vel; pos; collision] (scp.py:329-358, :407-439), accepted statuses {solved,
    solved inaccurate} (scp.py:363, :446), warm start of the primal only (scp.py:443);
  * P = 2I > 0 makes the minimiser unique, so any convergent method must agree on it: the solvers
    below are cross-checked against each other and against scipy's trust-constr at small sizes
    (tests/test_host_cpu.py::test_qp_oracles_agree_and_match_trust_constr) and against the closed-form min-norm solution when no box row is active.

Two solvers:

  osqp_explicit(...)   OSQP's published algorithm (Stellato et al., "OSQP: an operator splitting
                       solver for quadratic programs", Math. Prog. Comp. 2020) on explicit sparse
                       matrices with a direct KKT solve: Ruiz equilibration, rho/sigma/alpha =
                       0.1/1e-6/1.6, rho x 1e3 on equality rows, adaptive rho, eps_abs = eps_rel =
                       1e-3, termination checked every 25 iterations.  The wall-clock driven
                       adaptive-rho interval of osqp 0.6 is replaced by a fixed iteration interval.
  admm_structured(...) The matrix-free working-set ADMM that the HIP path implements (same update
                       order, same PCG, same constants); this is what the GPU result is compared to.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import scp_oracle as so

OSQP_SOLVED = 1
OSQP_SOLVED_INACCURATE = 2
OSQP_MAX_ITER_REACHED = -2
OSQP_PRIMAL_INFEASIBLE = -3
STATUS_TEXT = {
    1: "solved",
    2: "solved inaccurate",
    -2: "maximum iterations reached",
    -3: "primal infeasible",
}


# ---------------------------------------------------------------------------------------------
# OSQP restated on explicit matrices (small sizes)
# ---------------------------------------------------------------------------------------------
def _ruiz(P, q, A, l, u, iters=10):
    n, m = P.shape[0], A.shape[0]
    D = np.ones(n)
    E = np.ones(m)
    c = 1.0
    P = sp.csc_matrix(P, copy=True).astype(float)
    A = sp.csc_matrix(A, copy=True).astype(float)
    q = np.array(q, dtype=float)
    for _ in range(iters):
        # column inf-norms of the KKT matrix [[P, A^T], [A, 0]]
        colP = np.abs(P).max(axis=0).toarray().ravel() if P.nnz else np.zeros(n)
        colA = np.abs(A).max(axis=0).toarray().ravel() if A.nnz else np.zeros(n)
        rowA = np.abs(A).max(axis=1).toarray().ravel() if A.nnz else np.zeros(m)
        dn = np.maximum(colP, colA)
        dn = np.where(dn < 1e-4, 1.0, dn)
        en = np.where(rowA < 1e-4, 1.0, rowA)
        dlt = 1.0 / np.sqrt(np.clip(dn, 1e-4, 1e4))
        elt = 1.0 / np.sqrt(np.clip(en, 1e-4, 1e4))
        Dm = sp.diags(dlt)
        Em = sp.diags(elt)
        P = (Dm @ P @ Dm).tocsc()
        A = (Em @ A @ Dm).tocsc()
        q = dlt * q
        D *= dlt
        E *= elt
        # cost scaling
        colPn = np.abs(P).max(axis=0).toarray().ravel()
        mean_col = colPn.mean()
        qn = np.abs(q).max() if q.size else 0.0
        qn = qn if qn > 1e-4 else 1.0
        g = max(mean_col, qn)
        g = min(max(g, 1e-4), 1e4)
        g = 1.0 / g
        P = P * g
        q = q * g
        c *= g
    return P, q, A, E * l, E * u, D, E, c


def osqp_explicit(P, q, A, l, u, x0=None, rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3,
                  max_iter=4000, check_termination=25, adaptive_rho=True, adaptive_rho_interval=100,
                  adaptive_rho_tolerance=5.0, scaling=10, eps_prim_inf=1e-4):
    """OSQP's ADMM on explicit matrices.  Returns dict(x, y, status_val, status, iter, r_prim, r_dual)."""
    P = sp.csc_matrix(P).astype(float)
    A = sp.csc_matrix(A).astype(float)
    n, m = P.shape[0], A.shape[0]
    q = np.asarray(q, dtype=float)
    l = np.asarray(l, dtype=float)
    u = np.asarray(u, dtype=float)
    if scaling:
        Ps, qs, As, ls, us, D, E, c = _ruiz(P, q, A, l, u, scaling)
    else:
        Ps, qs, As, ls, us, D, E, c = P, q, A, l, u, np.ones(n), np.ones(m), 1.0
    eq = np.abs(ls - us) < 1e-4  # RHO_TOL, rows treated as equalities
    free = (ls < -1e20 * 1e-4) & (us > 1e20 * 1e-4)

    def rho_vec(r):
        v = np.full(m, r)
        v[eq] = 1e3 * r
        v[free] = 1e-6
        return v

    def factor(rv):
        KKT = sp.bmat([[Ps + sigma * sp.eye(n), As.T], [As, -sp.diags(1.0 / rv)]], format="csc")
        return spla.splu(KKT)

    rv = rho_vec(rho)
    lu = factor(rv)
    x = np.zeros(n) if x0 is None else np.asarray(x0, dtype=float) / D
    z = As @ x if x0 is not None else np.zeros(m)
    y = np.zeros(m)
    status = OSQP_MAX_ITER_REACHED
    it = 0
    rp = rd = np.inf
    for it in range(1, max_iter + 1):
        rhs = np.concatenate([sigma * x - qs, z - y / rv])
        sol = lu.solve(rhs)
        xt = sol[:n]
        nu = sol[n:]
        zt = z + (nu - y) / rv
        x_new = alpha * xt + (1 - alpha) * x
        zh = alpha * zt + (1 - alpha) * z
        z_new = np.clip(zh + y / rv, ls, us)
        y_new = y + rv * (zh - z_new)
        dy = y_new - y
        x, z, y = x_new, z_new, y_new
        if it % check_termination == 0 or it == max_iter:
            # unscaled residuals (scaled_termination = False)
            Ax = As @ x
            rp = np.abs((Ax - z) / E).max() if m else 0.0
            Px = Ps @ x
            ATy = As.T @ y
            rd = np.abs((Px + qs + ATy) / D).max() / c
            np_norm = max(np.abs(Ax / E).max(), np.abs(z / E).max()) if m else 0.0
            nd_norm = max(np.abs(Px / D).max(), np.abs(ATy / D).max(), np.abs(qs / D).max()) / c
            if rp <= eps_abs + eps_rel * np_norm and rd <= eps_abs + eps_rel * nd_norm:
                status = OSQP_SOLVED
                break
            if it == max_iter and rp <= 10 * (eps_abs + eps_rel * np_norm) and rd <= 10 * (eps_abs + eps_rel * nd_norm):
                status = OSQP_SOLVED_INACCURATE  # OSQP's approximate-tolerance test at max_iter
            # primal infeasibility certificate (dy)
            dyu = E * dy
            ndy = np.abs(dyu).max()
            if ndy > 1e-30:
                ATdy = np.abs((As.T @ dy) / D).max()
                supp = np.sum(np.where(np.isfinite(us), us, 0.0) * np.maximum(dy, 0)
                              + np.where(np.isfinite(ls), ls, 0.0) * np.minimum(dy, 0))
                bad_inf = np.any((~np.isfinite(us)) & (dy > eps_prim_inf * ndy)) or np.any(
                    (~np.isfinite(ls)) & (dy < -eps_prim_inf * ndy))
                if ATdy <= eps_prim_inf * ndy and supp <= -eps_prim_inf * ndy and not bad_inf:
                    status = OSQP_PRIMAL_INFEASIBLE
                    break
        if adaptive_rho and adaptive_rho_interval and it % adaptive_rho_interval == 0:
            Ax = As @ x
            Px = Ps @ x
            ATy = As.T @ y
            # scaled residuals, as OSQP's compute_rho_estimate
            prim = np.abs(Ax - z).max() / max(np.abs(Ax).max(), np.abs(z).max(), 1e-10)
            dual = np.abs(Px + qs + ATy).max() / max(np.abs(Px).max(), np.abs(ATy).max(), np.abs(qs).max(), 1e-10)
            new = rho * np.sqrt(prim / max(dual, 1e-10))
            new = min(max(new, 1e-6), 1e6)
            if new > rho * adaptive_rho_tolerance or new < rho / adaptive_rho_tolerance:
                rho = new
                rv = rho_vec(rho)
                lu = factor(rv)
    return {
        "x": D * x, "y": E * y / c, "z": z / E, "status_val": status, "status": STATUS_TEXT[status],
        "iter": it, "r_prim": rp, "r_dual": rd, "rho": rho,
    }


# ---------------------------------------------------------------------------------------------
# Structured working-set ADMM: the algorithm of the HIP path, in numpy
# ---------------------------------------------------------------------------------------------
FINE_MAX_COLUMNS = 4096  # (SCP_FINE_MAX_COLUMNS in scp_qp.hip)


@dataclasses.dataclass
class Settings:
    rho: float = 0.1
    sigma: float = 1e-6
    alpha: float = 1.6
    rho_eq_scale: float = 1e3
    eps_abs: float = 1e-3
    eps_rel: float = 1e-3
    max_iter: int = 4000
    check_termination: int = 25
    # adaptive check cadence (scp_qp_settings.check_fine / check_fine_ratio): after a check whose residuals are within
    # `check_fine_ratio` x their tolerances the next check comes after `check_fine` steps (a divisor of check_termination)
    # instead of check_termination; 0: fixed cadence
    check_fine: int = 5
    check_fine_ratio: float = 2.0
    adaptive_rho: bool = True
    adaptive_rho_interval: int = 50  # (scp_qp_default_settings: a measured choice, see there)
    adaptive_rho_tolerance: float = 5.0
    cg_iters: int = 1         # PCG steps per ADMM step (fixed count; warm started at x)
    cg_tol: float = 0.0       # optional early exit: ||r||_2 <= cg_tol * ||rhs||_2
    margin: float = 0.5       # initial working set: rows with dist_prev - R < margin
    feas_tol: float = 1e-6    # a non-working row enters W when (A x)_r < l_r - feas_tol
    max_rounds: int = 20      # constraint-generation rounds
    rho_col_scale: float = 10.0  # rho of the collision rows = rho * rho_col_scale
    eps_prim_inf: float = 1e-4   # OSQP's primal infeasibility tolerance (certificate test at every check)


class FixedOps:
    """Per-(agent, axis) K-column operators; identical for every column."""

    def __init__(self, K, h):
        self.K, self.h = K, h
        self.J, self.Id, self.V, self.S, self.S0 = so.time_blocks(K, h)

    def apply(self, x):
        """x (N,K,D) -> (jerk (N,K-1,D), acc, vel, pos (N,K,D))."""
        e = lambda B: np.einsum("km,imd->ikd", B, x)
        return e(self.J), x.copy(), e(self.V), e(self.S)

    def apply_T(self, wj, wa, wv, wp):
        e = lambda B, w: np.einsum("km,ikd->imd", B, w)
        return e(self.J, wj) + wa + e(self.V, wv) + e(self.S, wp)

    def kkt_matrix(self, sigma, rj, ra, rv, rp):
        """H_f = (2+sigma) I + J^T rj J + ra I + V^T diag(rv) V + S^T diag(rp) S  (K x K);
        rj, ra scalars, rv, rp length-K vectors (last entry carries the equality weight)."""
        K = self.K
        return ((2.0 + sigma + ra) * np.eye(K) + rj * (self.J.T @ self.J)
                + self.V.T @ (rv[:, None] * self.V) + self.S.T @ (rp[:, None] * self.S))


def working_rows(prob, rows):
    iu, ju = so.pair_index(prob.N)
    pairs = iu.size
    k = rows // pairs
    q = rows % pairs
    return k, iu[q], ju[q]


def admm_structured(prob: so.Problem, eta=None, l_col=None, dist=None, x0=None, st: Settings | None = None,
                    rows0=None, trace=None):
    """Joint QP  min ||x||^2  s.t. fixed rows and ALL collision rows (eta, l_col; None -> QP#0).

    Exact constraint generation: ADMM runs on the fixed rows plus a working set W of collision rows;
    after it terminates every row outside W is checked at the solution and violated rows join W
    (duals start at 0), until none is violated.  The final point satisfies the KKT conditions of the
    FULL joint QP (rows outside W are feasible with zero multiplier) to the solver tolerances.
    """
    st = st or Settings()
    N, K, D, h = prob.N, prob.K, prob.D, prob.h
    ops = FixedOps(K, h)
    b = so.fixed_bounds(prob)
    lj, uj = b["jerk"]
    la, ua = b["acc"]
    lv, uv = b["vel"]
    lp, up = b["pos"]
    have_col = eta is not None
    m_col = prob.m_col if have_col else 0

    x = np.zeros((N, K, D)) if x0 is None else np.asarray(x0, dtype=float).reshape(N, K, D).copy()
    # z = A x (warm start of the primal only, duals zero: scp.py:443)
    zj, za, zv, zp = ops.apply(x)
    yj, ya, yv, yp = (np.zeros_like(zj), np.zeros_like(za), np.zeros_like(zv), np.zeros_like(zp))

    if have_col:
        if rows0 is not None:
            W = np.asarray(rows0, dtype=np.int64)
        else:
            W = np.nonzero(dist - prob.R < st.margin)[0].astype(np.int64)
    else:
        W = np.zeros(0, dtype=np.int64)

    def col_setup(W):
        wk, wi, wj_ = working_rows(prob, W)
        return wk, wi, wj_, eta[W], l_col[W]

    def col_apply(x_, wk, wi, wj_, we):
        Q = np.einsum("km,imd->ikd", ops.S0, x_)
        return np.sum(we * (Q[wi, wk, :] - Q[wj_, wk, :]), axis=1)

    def col_apply_T(g, wk, wi, wj_, we):
        G = np.zeros((N, K, D))
        c = we * g[:, None]
        np.add.at(G, (wi, wk), c)
        np.add.at(G, (wj_, wk), -c)
        return np.einsum("km,ikd->imd", ops.S0, G)

    rho = st.rho
    info = {"iter": 0, "rounds": 0, "cg_total": 0, "rho_updates": 0}
    total_it = 0
    status = OSQP_MAX_ITER_REACHED
    zc = yc = None
    if have_col:
        wk, wi, wj_, we, wl = col_setup(W)
        zc = np.maximum(col_apply(x, wk, wi, wj_, we), wl) if W.size else np.zeros(0)
        yc = np.zeros(W.size)

    for rnd in range(st.max_rounds):
        info["rounds"] = rnd + 1

        def build(rho):
            rvv = np.full(K, rho)
            rvv[K - 1] = rho * st.rho_eq_scale
            rpp = rvv.copy()
            Hf = ops.kkt_matrix(st.sigma, rho, rho, rvv, rpp)
            return rvv, rpp, np.linalg.inv(Hf), Hf

        rvv, rpp, M, Hf = build(rho)
        status = OSQP_MAX_ITER_REACHED
        it = 0
        cad = st.check_termination  # steps between two checks (every round starts on the coarse cadence)
        # (the fine cadence applies when it divides the coarse one; otherwise the cadence is fixed)
        # and to QPs with collision rows: QP#0 keeps the fixed cadence (its 20 surplus steps are cheap, and a better converged
        # starting point saves the first joint QP of large problems far more: 250 instead of 400 steps at 1024 x 50)
        # and up to 4096 columns (2048 agents in 2-D): beyond, the GPU measured slower with it (the oracle follows the product)
        fine = st.check_fine if (0 < st.check_fine < st.check_termination and st.check_termination % st.check_fine == 0
                                 and W.size > 0 and N * D <= FINE_MAX_COLUMNS) else 0
        xt = x.copy()
        while total_it < st.max_iter:
            it += 1
            total_it += 1
            rk = lambda r: r[None, :, None]
            # rhs = sigma x + A^T (rho z - y)
            rhs = st.sigma * x + ops.apply_T(rho * zj - yj, rho * za - ya, rk(rvv) * zv - yv, rk(rpp) * zp - yp)
            rho_c = rho * st.rho_col_scale
            if W.size:
                rhs = rhs + col_apply_T(rho_c * zc - yc, wk, wi, wj_, we)

            def Hmul(p):
                out = np.einsum("km,imd->ikd", Hf, p)
                if W.size:
                    out = out + col_apply_T(rho_c * col_apply(p, wk, wi, wj_, we), wk, wi, wj_, we)
                return out

            if W.size:
                # PCG, preconditioner M = H_f^{-1}, warm start at the current iterate x
                xt = x.copy()
                r = rhs - Hmul(xt)
                zz = np.einsum("km,imd->ikd", M, r)
                p = zz.copy()
                rz = float(np.sum(r * zz))
                rhs_n = float(np.sqrt(np.sum(rhs * rhs)))
                for _ in range(st.cg_iters):
                    if st.cg_tol > 0 and np.sqrt(np.sum(r * r)) <= st.cg_tol * rhs_n:
                        break
                    Hp = Hmul(p)
                    pHp = float(np.sum(p * Hp))
                    if pHp <= 0.0 or rz == 0.0:
                        break
                    a = rz / pHp
                    xt = xt + a * p
                    r = r - a * Hp
                    zz = np.einsum("km,imd->ikd", M, r)
                    rz_new = float(np.sum(r * zz))
                    p = zz + (rz_new / rz) * p
                    rz = rz_new
                    info["cg_total"] += 1
            else:
                xt = np.einsum("km,imd->ikd", M, rhs)
            # z~ = A x~ ; relaxation ; projection ; dual update
            will_check = (it % cad == 0) or total_it >= st.max_iter
            if will_check:
                y_prev = (yj, ya, yv, yp, yc)
            tj, ta, tv, tp = ops.apply(xt)
            al = st.alpha
            x_new = al * xt + (1 - al) * x

            def upd(zt, z, y, r, lo, hi):
                zh = al * zt + (1 - al) * z
                zn = np.clip(zh + y / r, lo, hi)
                return zn, y + r * (zh - zn)

            zj, yj = upd(tj, zj, yj, rho, lj, uj)
            za, ya = upd(ta, za, ya, rho, la, ua)
            zv, yv = upd(tv, zv, yv, rk(rvv), lv, uv)
            zp, yp = upd(tp, zp, yp, rk(rpp), lp, up)
            if W.size:
                tc = col_apply(xt, wk, wi, wj_, we)
                zc, yc = upd(tc, zc, yc, rho_c, wl, np.inf)
            x = x_new

            check = (it % cad == 0) or total_it >= st.max_iter
            if check:
                info["checks"] = info.get("checks", 0) + 1
                aj, aa, av, ap = ops.apply(x)
                rp_ = max(np.abs(aj - zj).max(), np.abs(aa - za).max(), np.abs(av - zv).max(), np.abs(ap - zp).max())
                nAx = max(np.abs(aj).max(), np.abs(aa).max(), np.abs(av).max(), np.abs(ap).max())
                nz = max(np.abs(zj).max(), np.abs(za).max(), np.abs(zv).max(), np.abs(zp).max())
                ATy = ops.apply_T(yj, ya, yv, yp)
                if W.size:
                    ac = col_apply(x, wk, wi, wj_, we)
                    rp_ = max(rp_, np.abs(ac - zc).max())
                    nAx = max(nAx, np.abs(ac).max())
                    nz = max(nz, np.abs(zc).max())
                    ATy = ATy + col_apply_T(yc, wk, wi, wj_, we)
                rd_ = np.abs(2.0 * x + ATy).max()
                nPx = np.abs(2.0 * x).max()
                nATy = np.abs(ATy).max()
                info["r_prim"], info["r_dual"] = float(rp_), float(rd_)
                if trace is not None:
                    trace.append((total_it, rho, float(rp_), float(rd_), int(W.size)))
                tol_p, tol_d = st.eps_abs + st.eps_rel * max(nAx, nz), st.eps_abs + st.eps_rel * max(nPx, nATy)
                if rp_ <= tol_p and rd_ <= tol_d:
                    status = OSQP_SOLVED
                    break
                if fine:  # close to the tolerances: look again soon
                    near = rp_ < st.check_fine_ratio * tol_p and rd_ < st.check_fine_ratio * tol_d
                    cad = fine if near else st.check_termination
                # OSQP at max_iter: the same test with 10 x the tolerances -> "solved inaccurate" (accepted by the
                # reference like "solved", scp.py:363, :446)
                if total_it >= st.max_iter and rp_ <= 10 * (st.eps_abs + st.eps_rel * max(nAx, nz)) \
                        and rd_ <= 10 * (st.eps_abs + st.eps_rel * max(nPx, nATy)):
                    status = OSQP_SOLVED_INACCURATE
                # primal infeasibility certificate (OSQP is_primal_infeasible): dy = y - y_prev projected onto the
                # polar of the recession cone (collision rows have u = +inf -> dy := min(dy, 0)); all fixed rows
                # have finite bounds
                dj, da, dv, dp = yj - y_prev[0], ya - y_prev[1], yv - y_prev[2], yp - y_prev[3]
                dc = np.minimum(yc - y_prev[4], 0.0) if W.size else np.zeros(0)
                ndy = max(np.abs(dj).max(), np.abs(da).max(), np.abs(dv).max(), np.abs(dp).max(),
                          np.abs(dc).max() if W.size else 0.0)
                if ndy > st.eps_prim_inf:
                    supp = 0.0
                    for dd, (lo, hi) in ((dj, (lj, uj)), (da, (la, ua)), (dv, (lv, uv)), (dp, (lp, up))):
                        supp += float(np.sum(hi * np.maximum(dd, 0.0) + lo * np.minimum(dd, 0.0)))
                    if W.size:
                        supp += float(np.sum(wl * dc))
                    if supp < -st.eps_prim_inf * ndy:
                        ATdy = ops.apply_T(dj, da, dv, dp)
                        if W.size:
                            ATdy = ATdy + col_apply_T(dc, wk, wi, wj_, we)
                        if np.abs(ATdy).max() < st.eps_prim_inf * ndy:
                            status = OSQP_PRIMAL_INFEASIBLE
                            break
                if st.adaptive_rho and it % st.adaptive_rho_interval == 0:
                    prim = rp_ / max(nAx, nz, 1e-10)
                    dual = rd_ / max(nPx, nATy, 1e-10)
                    new = min(max(rho * np.sqrt(prim / max(dual, 1e-10)), 1e-6), 1e6)
                    # snapped to a geometric grid (steps of 2^(1/4)): the estimate is a ratio of small residuals, so
                    # 1e-13 of fp noise would otherwise become a percent-level difference in rho and send two
                    # implementations of the same algorithm down visibly different (equally valid) iterate paths
                    new = float(2.0 ** (np.round(4.0 * np.log2(new)) / 4.0))
                    if new > rho * st.adaptive_rho_tolerance or new < rho / st.adaptive_rho_tolerance:
                        rho = new
                        rvv, rpp, M, Hf = build(rho)
                        info["rho_updates"] += 1
                        if fine:  # (the residuals usually fall below the tolerances within a few steps of a new rho)
                            cad = fine
        # constraint generation: check every collision row outside W at the ADMM solution
        if not have_col or status == OSQP_PRIMAL_INFEASIBLE:
            break
        ax_all = so.collision_apply(prob, eta, x.ravel())
        viol = ax_all < l_col - st.feas_tol
        viol[W] = False
        new_rows = np.nonzero(viol)[0].astype(np.int64)
        info.setdefault("added", []).append(int(new_rows.size))
        if new_rows.size == 0 or total_it >= st.max_iter:
            break
        # keep (z, y) of the old rows, new rows start at z = max(Ax, l), y = 0
        order = np.argsort(np.concatenate([W, new_rows]), kind="stable")
        zc = np.concatenate([zc, np.maximum(ax_all[new_rows], l_col[new_rows])])[order]
        yc = np.concatenate([yc, np.zeros(new_rows.size)])[order]
        W = np.concatenate([W, new_rows])[order]
        wk, wi, wj_, we, wl = col_setup(W)

    info.update(iter=total_it, status_val=status, status=STATUS_TEXT[status], rho=rho, working_rows=int(W.size))
    y = {"jerk": yj, "acc": ya, "vel": yv, "pos": yp, "col_rows": W, "col": yc}
    return x, y, info


# ---------------------------------------------------------------------------------------------
# SCP outer loop (scp.py:131-180), oracle form
# ---------------------------------------------------------------------------------------------
def scp_solve(prob: so.Problem, max_iterations=15, st: Settings | None = None, force_iterations=None, log=None,
              carry_rho=False):
    """generate_trajectories (scp.py:131-180) on top of admm_structured.

    carry_rho (the solver's opt-in of the same name): the joint QP of iteration n + 1 starts at the rho iteration n ended
    with instead of st.rho (the reference builds a new OSQP object per iteration, scp.py:441).

    `is_feasible` is evaluated once on QP#0's trajectory and never refreshed (scp.py:144, :152).
    force_iterations: run exactly that many loop bodies regardless of the flags (bench mode)."""
    st = st or Settings(max_iter=10000)  # scp.py:442
    st0 = dataclasses.replace(st, max_iter=min(st.max_iter, 4000))  # OSQP default for QP#0 (scp.py:360)
    x, _, info0 = admm_structured(prob, st=st0)
    if info0["status_val"] not in (1, 2):  # scp.py:363-365
        raise RuntimeError(f"OSQP failed: {info0['status']}")
    infos = [info0]
    pos, _ = so.kinematics(prob, x)
    is_feasible, _ = so.check_avoidance(prob, pos)
    it = 0
    converged = False
    rels = []
    while True:
        if force_iterations is not None:
            if it >= force_iterations:
                break
        elif not (it < max_iterations and not converged and not is_feasible):
            break
        prev_pos, _ = so.kinematics(prob, x)
        eta, l_col, dist = so.linearize_pairs(prob, prev_pos)
        st_it = dataclasses.replace(st, rho=infos[-1]["rho"]) if (carry_rho and len(infos) > 1) else st
        x_new, _, info = admm_structured(prob, eta, l_col, dist, x0=x, st=st_it)
        infos.append(info)
        rel = float(np.linalg.norm((x_new - x).ravel()) / np.linalg.norm(x.ravel()))  # scp.py:157-159
        rels.append(rel)
        if log:
            log(f"SCP Iteration {it + 1}: rel_step={rel:.3e} admm_it={info['iter']} W={info['working_rows']}"
                f" rounds={info['rounds']} status={info['status']}")
        if rel <= prob.convergence_tolerance:
            converged = True
        x = x_new
        it += 1
    pos, vel = so.kinematics(prob, x)
    return {"positions": pos, "velocities": vel, "accelerations": x, "iterations": it, "converged": converged,
            "initially_feasible": is_feasible, "rel_steps": rels, "infos": infos}


def scp_solve_explicit(prob: so.Problem, max_iterations=15, log=None, **osqp_kw):
    """generate_trajectories (scp.py:131-180) the way the reference runs it: EXPLICIT sparse matrices, P = 2 I, the
    stack [C_jerk; C_acc; C_vel; C_pos; A_collision] with ALL N(N-1)/2 K collision rows (scp.py:407-439), a fresh
    solver per SCP iteration with max_iter = 10000 and a primal warm start (scp.py:441-443), solved by `osqp_explicit`
    (OSQP's published algorithm with its defaults: Ruiz scaling, eps = 1e-3) in place of the absent `osqp` package.
    The closest available stand-in for the reference's own output; small sizes only (explicit A_collision)."""
    n = prob.n
    P = 2.0 * sp.eye(n, format="csc")
    q = np.zeros(n)
    C, lf, uf = so.stack_fixed(prob)
    r0 = osqp_explicit(P, q, C, lf, uf, **osqp_kw)  # scp.py:360 (defaults: max_iter 4000)
    if r0["status_val"] not in (1, 2):  # scp.py:363-365
        raise RuntimeError(f"OSQP failed: {r0['status']}")
    x = r0["x"]
    infos = [r0]
    pos, _ = so.kinematics(prob, x)
    feasible, _ = so.check_avoidance(prob, pos)
    it, converged, rels = 0, False, []
    kw = dict(osqp_kw)
    kw["max_iter"] = 10000  # scp.py:442
    while it < max_iterations and not converged and not feasible:
        prev_pos, _ = so.kinematics(prob, x)
        eta, l_col, _ = so.linearize_pairs(prob, prev_pos)
        A = sp.vstack([C, so.collision_matrix_explicit(prob, eta)], format="csc")
        l = np.concatenate([lf, l_col])
        u = np.concatenate([uf, np.full(l_col.shape, np.inf)])  # scp.py:479
        r = osqp_explicit(P, q, A, l, u, x0=x, **kw)
        infos.append(r)
        x_new = r["x"]
        rel = float(np.linalg.norm(x_new - x) / np.linalg.norm(x))  # scp.py:157-159
        rels.append(rel)
        if log:
            log(f"SCP Iteration {it + 1}: rel_step={rel:.3e} osqp_it={r['iter']} status={r['status']}")
        converged = rel <= prob.convergence_tolerance
        x = x_new
        it += 1
    pos, vel = so.kinematics(prob, x)
    N, K, D = prob.N, prob.K, prob.D
    return {"positions": pos, "velocities": vel, "accelerations": x.reshape(N, K, D), "iterations": it,
            "converged": converged, "initially_feasible": feasible, "rel_steps": rels, "infos": infos}
