"""Developer tool: the reference's compute-trajectories demo (N = 10, T = 100 s, h = 0.2 s -> K = 500) without plots -- the
target of a rocprofv3 run for the long-horizon path."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))
from path_planning.cli import compute_trajectories as c  # noqa: E402

s = c.main(["--seed", "3", "--no-plots"])
if s is not None:
    print([(q["iter"], q["rounds"], q["status"], round(q["solve_ms"], 1)) for q in [s.last_info["qp0"]] + s.last_info["iterations"]])
