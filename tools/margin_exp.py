"""Developer experiment: the working-set margin (rows with dist - R < margin enter the QP at linearisation time, the rest only when
the exact constraint generation finds them violated) against rows, rounds, ADMM steps and time of the first SCP iteration and
of the complete solve.  Result (profiles/r03_margin_experiment.txt): no margin is better everywhere -- 0.5 stays."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ba-path-planning_amd'))
import numpy as np, torch
from path_planning.scenarios.position_generator import generate_grid_swap
from path_planning.solvers.scp import SCP
for N in (1024, 4096, 128):
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2)
    for m in (0.3, 0.5, 0.75, 1.0, 1.5):
        s = SCP(N, 10.0, 0.2, 0.8, space, verbose=False, working_set_margin=m)
        s.set_initial_states(p0); s.set_final_states(pf)
        s._precompute_constraint_matrices()
        acc0 = s._solve_initial_trajectory()
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            new, info = s.scp_iteration(acc0)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        t0 = time.perf_counter(); tr = s.generate_trajectories(max_iterations=15); torch.cuda.synchronize(); full = time.perf_counter() - t0
        t0 = time.perf_counter(); tr = s.generate_trajectories(max_iterations=15); torch.cuda.synchronize(); full = time.perf_counter() - t0
        its = [q["iter"] for q in s.last_info["iterations"]]
        print(f"N={N} margin={m}: step {min(ts)*1e3:.3f} ms, ADMM {info['iter']} rows {info['working_rows']} rounds {info['rounds']}; full solve {full*1e3:.2f} ms, its {its} rounds {[q['rounds'] for q in s.last_info['iterations']]}", flush=True)
        s.close()
