"""Developer tool: per-QP statistics of full solves on the reference's own scenario generator (N = 10..20)."""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))
import numpy as np, torch
from path_planning.scenarios.position_generator import generate_positions
from path_planning.solvers.scp import SCP
for N, seed in [(20, 20001), (20, 20002), (18, 18001), (10, 10003)]:
    p0, pf = generate_positions(N, 0.8, seed=seed)
    s = SCP(N, 10.0, 0.2, 0.8, [0, 0, 20, 20], verbose=False)
    s.set_initial_states(p0); s.set_final_states(pf)
    t = time.perf_counter()
    try:
        s.generate_trajectories(max_iterations=15)
        err = None
    except Exception as e:
        err = str(e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    li = s.last_info
    print(N, seed, f"{dt*1e3:.1f} ms", err, {k: li.get(k) for k in ("n_iterations", "converged", "feasible")})
    q0 = li.get("qp0", {})
    print("    qp0", {k: q0.get(k) for k in ("iter", "status", "solve_ms")})
    for q in li.get("iterations", [])[:20]:
        print("   ", {k: q.get(k) for k in ("iter", "rounds", "working_rows", "status", "solve_ms", "rho_updates", "rel_step")})
