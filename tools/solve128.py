"""Developer tool: a few complete 128-agent solves (config 5's unit) -- the target of a rocprofv3 --kernel-trace --stats run
that counts kernel launches per solve."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))
import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
s = None
for r in range(reps):
    p0, pf, space = generate_grid_swap(N, seed=100 + r)
    if s is None:
        s = SCP(N, 10.0, 0.2, 0.8, space, verbose=False)
    else:
        s.set_space_dims(space)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    torch.cuda.synchronize()
    t = time.perf_counter()
    s.generate_trajectories(15)
    dt = time.perf_counter() - t
    print(f"solve {r}: {dt*1e3:.2f} ms, {s.last_info['n_iterations']} SCP iterations, ADMM "
          f"{[s.last_info['qp0']['iter']] + [q['iter'] for q in s.last_info['iterations']]}, rounds {[q['rounds'] for q in s.last_info['iterations']]}")
