"""Developer tool: what ONE termination check costs inside the persistent kernels -- the same 200 ADMM steps (tolerances
1e-12: never solved; fixed rho) with a check every 200 / 50 / 25 / 10 / 5 steps."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [128, 1024, 4096]:
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2)
    base = None
    s0 = SCP(N, 10.0, 0.2, 0.8, space, verbose=False)
    s0.set_initial_states(p0)
    s0.set_final_states(pf)
    s0._precompute_constraint_matrices()
    acc0 = s0._solve_initial_trajectory().clone()  # (QP#0 with the default settings)
    for c in (200, 50, 25, 10, 5):
        s = SCP(N, 10.0, 0.2, 0.8, space, verbose=False, max_rounds=1,
                qp_settings={"max_iter": 200, "eps_abs": 1e-12, "eps_rel": 1e-12, "adaptive_rho": 0, "check_termination": c,
                             "check_fine": 0, "eps_prim_inf": 1e-4})
        s.set_initial_states(p0)
        s.set_final_states(pf)
        s._precompute_constraint_matrices()
        best = min(s.scp_iteration(acc0)[1]["solve_ms"] for _ in range(4))
        info = s.scp_iteration(acc0)[1]
        checks = 200 // c
        if base is None:
            base = (best, checks)
        extra = (best - base[0]) / max(checks - base[1], 1) * 1e3
        print(f"N={N} check every {c:3d}: {info['iter']} steps, {checks:2d} checks, QP {best:.3f} ms"
              + (f"  -> {extra:5.1f} us per additional check ({extra / (base[0] / 200 * 1e3):.1f} steps)" if checks > base[1] else ""), flush=True)
        s.close()
