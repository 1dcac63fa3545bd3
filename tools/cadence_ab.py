"""Developer tool: fixed check cadence (check_fine = 0) against the adaptive one (check_fine = 5) on the SAME box: the first
SCP iteration (one scp_solver_step, row-free) and the complete solve, per problem size."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [128, 1024, 4096]:
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2)
    for fine, ratio in ((0, 4.0), (5, 4.0), (5, 2.0), (5, 1.5)):
        s = SCP(N, 10.0, 0.2, 0.8, space, verbose=False, qp_settings={"check_fine": fine, "check_fine_ratio": ratio})
        s.set_initial_states(p0)
        s.set_final_states(pf)
        s._precompute_constraint_matrices()
        acc0 = s._solve_initial_trajectory()
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            new, info = s.scp_iteration(acc0)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        fs = []
        for rep in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            s.generate_trajectories(max_iterations=15)
            torch.cuda.synchronize()
            fs.append(time.perf_counter() - t)
        its = [q["iter"] for q in s.last_info["iterations"]]
        print(f"N={N} check_fine={fine} ratio={ratio}: step {min(ts)*1e3:.3f} ms ({info['iter']} ADMM steps, {info['rounds']} rounds, "
              f"{info['rho_updates']} rho updates, QP {info['solve_ms']:.3f} ms); full solve {min(fs)*1e3:.2f} ms, steps {its}", flush=True)
        s.close()
