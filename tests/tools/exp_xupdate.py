"""Developer experiment (CPU, numpy): ADMM iteration counts of the first linearised joint QP under different inexact
x-updates.  Not a test; uses the oracle (test infrastructure).

    python tests/tools/exp_xupdate.py 64 128

modes
  exact : x~ = H^{-1} rhs (PCG to 1e-12)
  pcg1  : one PCG step from x, preconditioner H_f^{-1}, exact line search (global scalar)    [round-1 default]
  pcg2  : two PCG steps
  bj1   : one block-Jacobi step from x with the EXACT per-agent blocks
          H_ii = H_f (x) I_D + rho_c sum_{r ni i} (S0^T e_k)(S0^T e_k)^T (x) eta eta^T     (no global scalar)
  bj2   : two such steps
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))

from oracle import qp_oracle as qo  # noqa: E402
from oracle import scp_oracle as so  # noqa: E402
from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402


def run(prob, eta, l_col, dist, x0, mode, st, max_iter=4000, omega=1.0):
    N, K, D, h = prob.N, prob.K, prob.D, prob.h
    ops = qo.FixedOps(K, h)
    b = so.fixed_bounds(prob)
    (lj, uj), (la, ua), (lv, uv), (lp, up) = b["jerk"], b["acc"], b["vel"], b["pos"]
    W = np.nonzero(dist - prob.R < st.margin)[0].astype(np.int64)
    wk, wi, wj = qo.working_rows(prob, W)
    we, wl = eta[W], l_col[W]
    S0 = ops.S0

    def col_apply(x_):
        Q = np.einsum("km,imd->ikd", S0, x_)
        return np.sum(we * (Q[wi, wk, :] - Q[wj, wk, :]), axis=1)

    def col_apply_T(g):
        G = np.zeros((N, K, D))
        c = we * g[:, None]
        np.add.at(G, (wi, wk), c)
        np.add.at(G, (wj, wk), -c)
        return np.einsum("km,ikd->imd", S0, G)

    x = x0.copy()
    zj, za, zv, zp = ops.apply(x)
    yj, ya, yv, yp = (np.zeros_like(zj), np.zeros_like(za), np.zeros_like(zv), np.zeros_like(zp))
    zc = np.maximum(col_apply(x), wl)
    yc = np.zeros(W.size)
    rho = st.rho
    rk = lambda r: r[None, :, None]

    def build(rho):
        rvv = np.full(K, rho)
        rvv[K - 1] = rho * st.rho_eq_scale
        rpp = rvv.copy()
        Hf = ops.kkt_matrix(st.sigma, rho, rho, rvv, rpp)
        M = np.linalg.inv(Hf)
        Dinv = None
        if mode.startswith("bj"):
            rho_c = rho * st.rho_col_scale
            # per-agent blocks, variable order (k, d)
            Hii = np.zeros((N, K * D, K * D))
            base = np.kron(Hf, np.eye(D))
            Hii[:] = base
            for side in (wi, wj):
                for r in range(W.size):
                    u = np.kron(S0[wk[r], :], we[r])  # (K*D,)
                    Hii[side[r]] += rho_c * np.outer(u, u)
            Dinv = np.linalg.inv(Hii)
        return rvv, rpp, M, Hf, Dinv

    rvv, rpp, M, Hf, Dinv = build(rho)
    it = 0
    status = "max_iter"
    rho_updates = 0
    while it < max_iter:
        it += 1
        rho_c = rho * st.rho_col_scale
        rhs = st.sigma * x + ops.apply_T(rho * zj - yj, rho * za - ya, rk(rvv) * zv - yv, rk(rpp) * zp - yp)
        rhs = rhs + col_apply_T(rho_c * zc - yc)

        def Hmul(p):
            return np.einsum("km,imd->ikd", Hf, p) + col_apply_T(rho_c * col_apply(p))

        if mode in ("pcg1", "pcg2", "exact"):
            ncg = {"pcg1": 1, "pcg2": 2, "exact": 200}[mode]
            xt = x.copy()
            r = rhs - Hmul(xt)
            zz = np.einsum("km,imd->ikd", M, r)
            p = zz.copy()
            rz = float(np.sum(r * zz))
            for _ in range(ncg):
                Hp = Hmul(p)
                pHp = float(np.sum(p * Hp))
                if pHp <= 0 or rz == 0 or rz < 1e-26:
                    break
                a = rz / pHp
                xt = xt + a * p
                r = r - a * Hp
                zz = np.einsum("km,imd->ikd", M, r)
                rz_new = float(np.sum(r * zz))
                p = zz + (rz_new / rz) * p
                rz = rz_new
        else:
            nst = int(mode[2:])
            xt = x.copy()
            for _ in range(nst):
                r = rhs - Hmul(xt)
                dx = np.einsum("iab,ib->ia", Dinv, r.reshape(N, K * D)).reshape(N, K, D)
                xt = xt + omega * dx
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
        zc, yc = upd(col_apply(xt), zc, yc, rho_c, wl, np.inf)
        x = x_new
        if it % st.check_termination == 0:
            aj, aa, av, ap = ops.apply(x)
            ac = col_apply(x)
            rp_ = max(np.abs(aj - zj).max(), np.abs(aa - za).max(), np.abs(av - zv).max(), np.abs(ap - zp).max(),
                      np.abs(ac - zc).max())
            nAx = max(np.abs(aj).max(), np.abs(aa).max(), np.abs(av).max(), np.abs(ap).max(), np.abs(ac).max())
            nz = max(np.abs(zj).max(), np.abs(za).max(), np.abs(zv).max(), np.abs(zp).max(), np.abs(zc).max())
            ATy = ops.apply_T(yj, ya, yv, yp) + col_apply_T(yc)
            rd_ = np.abs(2.0 * x + ATy).max()
            nPx = np.abs(2.0 * x).max()
            nATy = np.abs(ATy).max()
            if not np.isfinite(rp_) or rp_ > 1e12:
                status = "diverged"
                break
            if rp_ <= st.eps_abs + st.eps_rel * max(nAx, nz) and rd_ <= st.eps_abs + st.eps_rel * max(nPx, nATy):
                status = "solved"
                break
            if st.adaptive_rho and it % st.adaptive_rho_interval == 0:
                prim = rp_ / max(nAx, nz, 1e-10)
                dual = rd_ / max(nPx, nATy, 1e-10)
                new = min(max(rho * np.sqrt(prim / max(dual, 1e-10)), 1e-6), 1e6)
                new = float(2.0 ** (np.round(4.0 * np.log2(new)) / 4.0))
                if new > rho * st.adaptive_rho_tolerance or new < rho / st.adaptive_rho_tolerance:
                    rho = new
                    rvv, rpp, M, Hf, Dinv = build(rho)
                    rho_updates += 1
    return x, dict(iter=it, status=status, rows=int(W.size), rho=rho, rho_updates=rho_updates)


def main():
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64]
    modes = [a for a in sys.argv[1:] if not a.isdigit()] or ["exact", "pcg1", "pcg2", "bj1", "bj2"]
    for N in sizes:
        K, h, R = 50, 0.2, 0.8
        p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2)
        prob = so.make_problem(N, K * h + 1e-9, h, R, space, p0, pf)
        st0 = qo.Settings(max_iter=4000)
        x0, _, info0 = qo.admm_structured(prob, st=st0)
        pos, _ = so.kinematics(prob, x0)
        eta, l_col, dist = so.linearize_pairs(prob, pos)
        st = qo.Settings(max_iter=10000)
        ref = None
        for mode in modes:
            t0 = time.time()
            x, info = run(prob, eta, l_col, dist, x0, mode, st)
            if ref is None:
                ref = x
            print(f"N={N} {mode:6s} iters={info['iter']:5d} {info['status']:9s} rows={info['rows']} rho={info['rho']:.3g}"
                  f" rho_upd={info['rho_updates']} |x-x_first|={np.abs(x-ref).max():.2e} t={time.time()-t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
