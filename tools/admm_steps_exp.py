"""Developer tool: fewer ADMM steps per SCP iteration, measured -- complete solves at N x 50 under
  (i)  carry_rho: the QP of iteration n + 1 starts at the rho iteration n ended with (opt-in, mirrored in the oracles);
  (ii) the termination-check cadence (settings.check_termination; the check runs inside the persistent kernel) and the
       adaptive-rho interval (a multiple of it).
Prints ADMM steps per QP, device time of the QP solves, wall time of the complete solve (second call on the object), the
objective ||a||^2 and the minimum pair distance of the result."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402


def run(N, label, carry=False, **qp):
    p0, pf, space = generate_grid_swap(N, seed=1000 * N)
    s = SCP(N, 10.0 + 1e-9, 0.2, 0.8, space, verbose=False, carry_rho=carry, qp_settings=qp)
    wall = []
    for _ in range(3):
        s.set_initial_states(p0)
        s.set_final_states(pf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        traj = s.generate_trajectories(15)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t)
    its = s.last_info["iterations"]
    rep = s.validate_solution()
    steps = [q["iter"] for q in its]
    ms = sum(q["solve_ms"] for q in its)
    print(f"{label:44s} SCP its {len(its)}  ADMM steps {steps} = {sum(steps):5d}  QP {ms:6.2f} ms ({ms*1e3/max(sum(steps),1):5.2f} us/step)  "
          f"solve {min(wall[1:])*1e3:6.2f} ms  rho end {[round(q['rho'], 4) for q in its]}  ||a||^2 {float((traj['accelerations']**2).sum()):.4f}  "
          f"min dist {rep['min_pair_distance']:.4f}  converged {s.last_info['converged']}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, nargs="*", default=[1024])
    a = ap.parse_args()
    for N in a.agents:
        print(f"== N = {N}")
        run(N, "default (check 25, rho interval 25)")
        run(N, "carry_rho", carry=True)
        for chk, rint in ((10, 50), (10, 30), (10, 20), (10, 10), (15, 30), (15, 15), (5, 25), (50, 50)):
            run(N, f"check {chk}, rho interval {rint}", check_termination=chk, adaptive_rho_interval=rint)
            run(N, f"check {chk}, rho interval {rint}, carry_rho", carry=True, check_termination=chk, adaptive_rho_interval=rint)


if __name__ == "__main__":
    main()
