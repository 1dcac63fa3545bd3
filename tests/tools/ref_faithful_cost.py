"""Developer tool (CPU): measured cost of the reference-faithful formulation -- explicit sparse matrices with ALL collision
rows, a fresh KKT factorisation per SCP iteration, OSQP's published algorithm (qp_oracle.scp_solve_explicit; the `osqp`
package itself is absent) -- per SCP iteration, next to the structured C oracle on the same scenario.
    python tests/tools/ref_faithful_cost.py 10 20 32"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

from oracle import c_oracle as co, qp_oracle as qo, scp_oracle as so  # noqa: E402
from path_planning.scenarios.position_generator import generate_grid_swap, generate_positions  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [10, 20]:
    if N <= 50:
        p0, pf = generate_positions(N, 0.8, seed=1000 * N + 1)
        space = [0, 0, 20, 20]
    else:
        p0, pf, space = generate_grid_swap(N, seed=1000 * N)
    prob = so.make_problem(N, 10.0, 0.2, 0.8, space, p0, pf)
    x0, _ = co.admm(prob, st=qo.Settings(max_iter=4000))
    pos, _ = co.kinematics(prob, x0)
    t = time.perf_counter()
    eta, l_col, _ = so.linearize_pairs(prob, pos)
    A_col = so.collision_matrix_explicit(prob, eta)
    t_asm = time.perf_counter() - t
    C, lf, uf = so.stack_fixed(prob)
    A = sp.vstack([C, A_col], format="csc")
    l = np.concatenate([lf, l_col]); u = np.concatenate([uf, np.full(l_col.shape, np.inf)])
    P = 2.0 * sp.eye(prob.n, format="csc")
    t = time.perf_counter()
    r = qo.osqp_explicit(P, np.zeros(prob.n), A, l, u, x0=x0.ravel(), max_iter=10000)
    t_qp = time.perf_counter() - t
    t = time.perf_counter()
    eta_c, l_c, dist_c = co.linearize_pairs(prob, pos)
    x1, i1 = co.admm(prob, eta_c, l_c, dist_c, x0, qo.Settings(max_iter=10000))
    t_c = time.perf_counter() - t
    print(f"N={N} K={prob.K}: explicit A_collision {A_col.nnz} nnz assembled in {t_asm:.2f} s (numpy restatement, not the "
          f"reference's Python loops); explicit OSQP (Ruiz, sparse LU of the {A.shape[0] + prob.n}-row KKT): {t_qp:.2f} s, "
          f"{r['iter']} iterations, {r['status']}; structured C oracle, one core: {t_c:.3f} s ({i1['iter']} iterations); "
          f"max |x_explicit - x_structured| = {np.abs(r['x'] - x1.ravel()).max():.2e}", flush=True)
