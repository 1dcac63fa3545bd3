"""Developer tool: what a NEW SCP object per scenario costs (the reference's own usage: compute_trajectories_batch.py
builds a solver per trial) against a reused one -- construction, first generate_trajectories, second one."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [20, 128, 1024]
    for N in sizes:
        scen = [generate_grid_swap(N, seed=1000 * N + s, dim=2) for s in range(6)]
        rows = []
        for rep, (p0, pf, space) in enumerate(scen):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s = SCP(N, 10.0, 0.2, 0.8, space, verbose=False)
            s.set_initial_states(p0)
            s.set_final_states(pf)
            t1 = time.perf_counter()
            s.generate_trajectories(max_iterations=15)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            s.set_initial_states(p0)
            s.set_final_states(pf)
            s.generate_trajectories(max_iterations=15)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            rows.append((t1 - t0, t2 - t1, t3 - t2))
            if hasattr(s, "close"):
                s.close()
            del s
        for rep, (a, b, c) in enumerate(rows):
            print(f"N={N:5d} object {rep}: construct {a*1e3:6.2f} ms, first solve {b*1e3:7.2f} ms, second solve {c*1e3:6.2f} ms")
        warm = rows[2:]
        print(f"N={N:5d} new object per scenario (objects 2..): {sum(a + b for a, b, _ in warm) / len(warm) * 1e3:.2f} ms per scenario; "
              f"reused object: {sum(c for _, _, c in warm) / len(warm) * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
