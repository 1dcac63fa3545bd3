"""Developer tool (VERDICT r2 item 5 ii): ADMM steps and solve time of complete solves as a function of the termination-check
cadence and the adaptive-rho interval, over a spread of problem sizes (grid-swap N = 64 ... 4096, the reference's generator
scenarios at N = 10, 20).  One line per (problem, setting); a summary table of the totals at the end."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap, generate_positions  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402


def problem(tag):
    if tag.startswith("g"):
        n, seed = (int(v) for v in tag[1:].split("s"))
        p0, pf = generate_positions(n, 0.8, seed=seed)
        return n, p0, pf, [0, 0, 20, 20]
    n, seed = (int(v) for v in tag.split("s")) if "s" in tag else (int(tag), None)
    p0, pf, space = generate_grid_swap(n, seed=1000 * n if seed is None else seed)
    return n, p0, pf, space


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", nargs="*", default=["g10s7", "g20s20", "64", "128s1", "128s2", "128s3", "256", "512", "1024", "2048", "4096"])
    ap.add_argument("--combos", nargs="*", default=["25:25", "25:50", "25:75", "25:100", "50:50", "50:100", "15:30", "15:45"])
    a = ap.parse_args()
    combos = [tuple(int(v) for v in c.split(":")) for c in a.combos]
    table = {}
    for tag in a.problems:
        n, p0, pf, space = problem(tag)
        for chk, rint in combos:
            s = SCP(n, 10.0 + 1e-9, 0.2, 0.8, space, verbose=False,
                    qp_settings={"check_termination": chk, "adaptive_rho_interval": rint})
            wall = []
            try:
                for _ in range(2):
                    s.set_initial_states(p0)
                    s.set_final_states(pf)
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    s.generate_trajectories(15)
                    torch.cuda.synchronize()
                    wall.append(time.perf_counter() - t)
            except RuntimeError as e:
                print(f"{tag:8s} check {chk:3d} rho {rint:3d}: {e}", flush=True)
                continue
            its = s.last_info["iterations"]
            steps = [q["iter"] for q in its]
            rep = s.validate_solution()
            ms = sum(q["solve_ms"] for q in its)
            table[(tag, chk, rint)] = (sum(steps), ms, wall[-1] * 1e3, len(its), bool(s.last_info["converged"]))
            print(f"{tag:8s} check {chk:3d} rho {rint:3d}: SCP its {len(its)} ADMM {steps} = {sum(steps):5d}  QP {ms:7.2f} ms  solve "
                  f"{wall[-1]*1e3:7.2f} ms  qp0 {s.last_info['qp0']['iter']}  min dist {rep['min_pair_distance']:.4f}  "
                  f"converged {s.last_info['converged']}  statuses {[q['status_val'] for q in its]}", flush=True)
            del s
    print("\ntotal ADMM steps of the joint QPs / total solve ms, per setting (sum over the problems above):")
    for chk, rint in combos:
        rows = [v for (t, c, r), v in table.items() if (c, r) == (chk, rint)]
        print(f"  check {chk:3d} rho {rint:3d}: steps {sum(v[0] for v in rows):6d}  QP ms {sum(v[1] for v in rows):8.2f}  solve ms "
              f"{sum(v[2] for v in rows):8.2f}  SCP its {sum(v[3] for v in rows):3d}  all converged {all(v[4] for v in rows)}  "
              f"({len(rows)} problems)")


if __name__ == "__main__":
    main()
