"""Developer experiment (CPU only, numpy oracle): would carrying the DUALS from one joint QP to the next (fixed rows only /
fixed rows + the collision rows that stay in the working set) save ADMM steps?  The reference hands OSQP the primal warm start
only (scp.py:443).  The oracle's admm_structured is patched in memory (y0=...), nothing in oracle/ changes.
Result (profiles/r03_warm_duals_experiment.txt): 10-45 % fewer steps on the grid-swap scenarios (16 ... 256 agents), but on
the generator scenarios whose linearised QPs run into the iteration limit the carried duals of an unconverged QP make the next
one WORSE (10 000 steps each, 2-3 x the total).  Not built.

    python tools/exp_warm_duals.py
"""
import sys, re, dataclasses, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'ba-path-planning_amd'))
import numpy as np
src = open(os.path.join(ROOT, 'oracle', 'qp_oracle.py')).read()
src = src.replace('from . import scp_oracle as so', 'from oracle import scp_oracle as so')
src = src.replace("rows0=None, trace=None):", "rows0=None, trace=None, y0=None):", 1)
src = src.replace("""    yj, ya, yv, yp = (np.zeros_like(zj), np.zeros_like(za), np.zeros_like(zv), np.zeros_like(zp))
""", """    yj, ya, yv, yp = (np.zeros_like(zj), np.zeros_like(za), np.zeros_like(zv), np.zeros_like(zp))
    if y0 is not None:
        yj, ya, yv, yp = (y0["jerk"].copy(), y0["acc"].copy(), y0["vel"].copy(), y0["pos"].copy())
""", 1)
src = src.replace("""        yc = np.zeros(W.size)

""", """        yc = np.zeros(W.size)
        if y0 is not None and y0.get("col_rows") is not None and len(y0["col_rows"]) and W.size:
            idx = np.searchsorted(y0["col_rows"], W)
            idx[idx >= len(y0["col_rows"])] = 0
            hit = y0["col_rows"][idx] == W
            yc[hit] = y0["col"][idx[hit]]

""", 1)
ns = {}
import types
mod = types.ModuleType('qo_warm'); mod.__dict__['__name__'] = 'qo_warm'
sys.modules['qo_warm'] = mod
exec(compile(src, 'qo_warm.py', 'exec'), mod.__dict__)
qo = mod
from oracle import scp_oracle as so
from path_planning.scenarios.position_generator import generate_grid_swap, generate_positions

def solve(prob, warm, st):
    st0 = dataclasses.replace(st, max_iter=min(st.max_iter, 4000))
    x, y, info0 = qo.admm_structured(prob, st=st0)
    its = [info0['iter']]
    pos, _ = so.kinematics(prob, x)
    feas, _ = so.check_avoidance(prob, pos)
    it = 0; conv = False
    while it < 15 and not conv and not feas:
        prev_pos, _ = so.kinematics(prob, x)
        eta, l_col, dist = so.linearize_pairs(prob, prev_pos)
        y0 = None
        if warm:
            y0 = dict(y)
            if warm == 'fixed': y0["col_rows"] = None
        xn, y, info = qo.admm_structured(prob, eta, l_col, dist, x0=x, st=st, y0=y0)
        its.append(info['iter'])
        rel = float(np.linalg.norm((xn - x).ravel()) / np.linalg.norm(x.ravel()))
        conv = rel <= prob.convergence_tolerance
        x = xn; it += 1
    pos, _ = so.kinematics(prob, x)
    return its, it, conv, x

def make(N, seed, kind):
    K, h, R = 50, 0.2, 0.8
    if kind == 'grid':
        p0, pf, space = generate_grid_swap(N, seed=seed, dim=2)
    else:
        p0, pf = generate_positions(N, 0.8 if N <= 10 else 0.4, seed=seed)
        space = [0, 0, 10, 10]
        R = 0.8 if N <= 10 else 0.4
    return so.make_problem(N, K * h + 1e-9, h, R, space, p0, pf)

if __name__ == '__main__':
    st = qo.Settings(max_iter=10000)
    cases = [(16, 16000, 'grid'), (64, 64000, 'grid'), (128, 128000, 'grid'), (256, 256000, 'grid'), (10, 1, 'gen'), (10, 2, 'gen')]
    for N, seed, kind in cases:
        prob = make(N, seed, kind)
        ref = None
        for warm in (None, 'fixed', 'all'):
            t = time.time()
            its, it, conv, x = solve(prob, warm, st)
            if ref is None: ref = x
            print(f"N={N:4d} {kind:4s} warm={str(warm):5s}: SCP its {it} conv {conv} ADMM {its} total {sum(its)}  max|x - x_cold| {np.abs(x-ref).max():.2e}  ({time.time()-t:.1f}s)", flush=True)
