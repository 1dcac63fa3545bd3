"""Developer experiment (CPU): how much does OSQP's Ruiz equilibration buy on the first linearised QP?
ADMM iteration counts of the explicit-matrix OSQP restatement with scaling = 10 / 0 on the working-set QP, next to the
structured solver (no scaling, collision rows at rho x 10).   python tests/tools/exp_scaling.py 64 128"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))
from oracle import c_oracle as co, qp_oracle as qo, scp_oracle as so  # noqa: E402
from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402


def rows_matrix(prob, eta, W):
    """explicit rows W of A_collision (only those)"""
    N, K, D, h = prob.N, prob.K, prob.D, prob.h
    iu, ju = so.pair_index(N)
    pairs = iu.size
    rows, cols, vals = [], [], []
    for n, r in enumerate(W):
        k, q = r // pairs, r % pairs
        if k == 0:
            continue
        m = np.arange(k)
        w = (h * h) * (k - m - 0.5)
        for d in range(D):
            rows.append(np.full(k, n)); cols.append((iu[q] * K + m) * D + d); vals.append(eta[r, d] * w)
            rows.append(np.full(k, n)); cols.append((ju[q] * K + m) * D + d); vals.append(-eta[r, d] * w)
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(len(W), N * K * D)).tocsc()


for N in [int(a) for a in sys.argv[1:]] or [64]:
    K, h, R = 50, 0.2, 0.8
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2)
    prob = so.make_problem(N, K * h + 1e-9, h, R, space, p0, pf)
    x0, i0 = co.admm(prob, st=qo.Settings(max_iter=4000))
    pos, _ = co.kinematics(prob, x0)
    eta, l_col, dist = co.linearize_pairs(prob, pos)
    W = np.nonzero(dist - R < 0.5)[0]
    x1, i1 = co.admm(prob, eta, l_col, dist, x0, qo.Settings(max_iter=10000, max_rounds=1))
    print(f"N={N}: structured (C oracle): {i1['iter']} iterations, {i1['working_rows']} rows, rho {i1['rho']:.3g}", flush=True)
    C, lf, uf = so.stack_fixed(prob)
    A = sp.vstack([C, rows_matrix(prob, eta, W)], format="csc")
    l = np.concatenate([lf, l_col[W]]); u = np.concatenate([uf, np.full(W.size, np.inf)])
    P = 2.0 * sp.eye(prob.n, format="csc"); q = np.zeros(prob.n)
    for scaling, interval in ((10, 25), (0, 25), (10, 100)):
        t = time.time()
        r = qo.osqp_explicit(P, q, A, l, u, x0=x0.ravel(), scaling=scaling, adaptive_rho_interval=interval, max_iter=10000)
        print(f"   explicit OSQP scaling={scaling:2d} rho-interval={interval}: {r['iter']} iterations, status {r['status']}, rho {r['rho']:.3g},"
              f" |x - x_struct| {np.abs(r['x'] - x1.ravel()).max():.2e}  ({time.time()-t:.0f}s)", flush=True)
