"""Developer tool: wall time of complete solves (generate_trajectories, max 15 SCP iterations) at bench sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [64, 1024]:
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2)
    s = SCP(N, 10.0, 0.2, 0.8, space, verbose=False)
    times = []
    for rep in range(3):  # rep 0 builds the solver object (device allocations: 10 GB of compact rows at N = 4096) and loads kernels
        s.set_initial_states(p0)
        s.set_final_states(pf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        s.generate_trajectories(max_iterations=15)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    dt = min(times[1:])
    rep_ = s.validate_solution()
    its = [q["iter"] for q in s.last_info["iterations"]]
    print(f"N={N}: {dt*1e3:.1f} ms (first call, with solver creation: {times[0]*1e3:.0f} ms), {s.last_info['n_iterations']} SCP iterations, "
          f"ADMM steps QP#0 {s.last_info['qp0']['iter']} + {its}, "
          f"min distance {rep_['min_pair_distance']:.4f}, collision_free={rep_['collision_free']}")
