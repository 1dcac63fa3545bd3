"""Developer tool: device time per ADMM step of the joint QP for a few shapes (N, dim) -- one SCP iteration from QP#0's
solution through scp_solver_step; prints solve_ms / ADMM iterations."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))
from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402

shapes = [(64, 2), (64, 3), (125, 3), (128, 2), (512, 2), (512, 3), (1024, 2)]
if len(sys.argv) > 1:  # NxD or NxDxP (P = settings.persistent: 2 lean 16-agent kernel, 3 lean 8-agent kernel, 0 three launches)
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for shape in shapes:
    n, dim = shape[:2]
    qp = {"persistent": shape[2]} if len(shape) > 2 else None
    p0, pf, space = generate_grid_swap(n, seed=1000 * n, dim=dim)
    s = SCP(n, 10.0, 0.2, 0.8, space, dim=dim, verbose=False, qp_settings=qp)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc0 = s._solve_initial_trajectory()
    for _ in range(2):
        _, info = s.scp_iteration(acc0)
    print(f"N={n} D={dim} [{info['pipeline']}]: {info['iter']} ADMM steps, {info['working_rows']} rows, {info['rounds']} rounds, "
          f"solve {info['solve_ms']:.3f} ms = {info['solve_ms']*1e3/max(info['iter'],1):.2f} us/step, "
          f"iteration wall {info['time_sec']*1e3:.2f} ms")
