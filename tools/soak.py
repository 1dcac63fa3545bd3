"""Developer tool: many complete solves on random sizes / dimensions / scenario kinds on cuda:0; every result must be finite,
every converged one keep all pairs at >= R - 0.02 (the linearised constraints hold to the ADMM tolerance eps_abs + eps_rel
|Ax| ~ 2e-2 m in a 20 m box, so the reference's own R - 0.01 check can fail by a few mm on a converged solve), and a
repeated solve must be bit-identical (the GPU path has no atomics in its iteration)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

import numpy as np  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap, generate_positions  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    polish = "--polish" in sys.argv  # one tight final QP: converged results must then pass the reference's R - 0.01 check
    large = "--large" in sys.argv    # sizes 1000 .. 4096 (2-D) / 600 .. 2048 (3-D): every persistent-kernel variant and its edges
    floor = 0.8 - (0.01 if polish else 0.02)
    rng = np.random.default_rng(2026)
    bad = 0
    t0 = time.perf_counter()
    for case in range(n_cases):
        kind = rng.choice(["ref", "grid", "grid3d"])
        if large:
            kind = rng.choice(["grid", "grid", "grid3d"])
            dim = 3 if kind == "grid3d" else 2
            N = int(rng.choice([2048, 2049, 4096, 1024, 1025]) if case < 5 and dim == 2 else
                    rng.integers(1000, 4097) if dim == 2 else rng.integers(600, 2049))
            p0, pf, space = generate_grid_swap(N, seed=int(rng.integers(1, 10**6)), dim=dim)
        elif kind == "ref":
            N, dim = int(rng.integers(2, 21)), 2
            p0, pf = generate_positions(N, 0.8, seed=int(rng.integers(1, 10**6)))
            space = [0, 0, 20, 20]
        else:
            dim = 3 if kind == "grid3d" else 2
            N = int(rng.integers(2, 161))
            p0, pf, space = generate_grid_swap(N, seed=int(rng.integers(1, 10**6)), dim=dim)
        T = 10.0 if large else float(rng.choice([6.0, 10.0, 13.0]))
        outs = []
        err = None
        for rep in range(2):
            s = SCP(N, T, 0.2, 0.8, space, dim=dim, verbose=False, polish=polish)
            s.set_initial_states(p0)
            s.set_final_states(pf)
            try:
                outs.append(s.generate_trajectories(max_iterations=15)["positions"].copy())
            except RuntimeError as e:  # "OSQP failed: ..." (infeasible QP#0): a legitimate outcome of the reference too
                err = str(e)
                break
        if err:
            print(f"case {case}: {kind} N={N} D={dim} T={T}: {err}")
            continue
        rep_ = s.validate_solution()
        finite = bool(np.isfinite(outs[0]).all())
        same = bool(np.array_equal(outs[0], outs[1]))
        conv = s.last_info.get("converged")
        ok = finite and same and (rep_["min_pair_distance"] >= floor or not conv)
        if polish and conv:
            worst = max(rep_[k] for k in ("acc_violation", "jerk_violation", "vel_violation", "pos_violation",
                                         "final_position_error", "final_velocity_error"))
            ok = ok and worst < 1e-5
        bad += 0 if ok else 1
        pipes = sorted({p for q in s.last_info["iterations"] for p in q["pipeline"].split("+")})
        print(f"case {case}: {kind} N={N} D={dim} T={T} {'+'.join(pipes)}: iterations={s.last_info['n_iterations']} converged={conv} "
              f"collision_free={rep_['collision_free']} min_dist={rep_['min_pair_distance']:.4f} finite={finite} "
              f"repeatable={same} {'OK' if ok else 'FAIL'}")
    print(f"{n_cases} cases, {bad} failures, {time.perf_counter() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
