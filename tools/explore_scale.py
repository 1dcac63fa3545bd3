"""Developer tool: phase timings of one SCP solve at a given size on cuda:0 (not part of the product path)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--margin", type=float, default=0.5)
    ap.add_argument("--cg", type=int, default=5)
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--block", type=int, default=4)
    ap.add_argument("--min-sep", type=float, default=0.3)
    a = ap.parse_args()
    N = a.agents
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=a.dim, block=a.block, min_sep=a.min_sep)
    s = SCP(N, 10.0, 0.2, 0.8, space, dim=a.dim, verbose=False, working_set_margin=a.margin,
            qp_settings={"cg_iters": a.cg, "eps_abs": a.eps, "eps_rel": a.eps})
    s.set_initial_states(p0)
    s.set_final_states(pf)
    t = time.time()
    s._precompute_constraint_matrices()
    acc = s._solve_initial_trajectory()
    torch.cuda.synchronize()
    print(f"QP#0: {time.time()-t:.3f}s", {k: s._last_qp_info[k] for k in ("iter", "status", "solve_ms", "rho")})
    pos, _ = s._kinematics(acc, want_vel=False)
    print("feasible0:", s._fast_check_avoidance_constraints(pos))
    pp = s._ensure_pairs()
    p0d, v0d, _, _ = s._states()
    for rep in range(3):
        rows, md, fv = pp.linearize(pos, p0d, v0d, a.margin)
        print(f"linearize: rows={pp.rows} sel={rows.numel()} min_dist={md:.4f} kernel_ms={s._ctx.last_pair_ms():.4f}"
              f" -> {pp.rows*8*(a.dim+1)/s._ctx.last_pair_ms()/1e6:.1f} GB/s")
    for it in range(a.iters):
        torch.cuda.synchronize()
        t = time.time()
        new = s._solve_with_avoidance_constraints(acc)
        torch.cuda.synchronize()
        dt = time.time() - t
        rel = s._ctx.rel_step(new, acc)[2]
        i = s._last_qp_info
        print(f"iter {it+1}: {dt:.3f}s rel={rel:.3e} admm={i['iter']} cg={i['cg_iters_total']} W={i['working_rows']} "
              f"rounds={i['rounds']} added={i['added']} qp_ms={i['solve_ms']:.1f} status={i['status']} rho={i['rho']:.3g}")
        acc = new
    pos, _ = s._kinematics(acc, want_vel=False)
    md, fv, _, _ = s._ctx.check_avoidance(N, s.K, s.D, s.R, pos)
    print("final min dist", md)


if __name__ == "__main__":
    main()
