"""Developer tool (profiling build: make -C ba-path-planning_amd/csrc prof; SCP_HIP_LIB=.../libscp_hip_prof.so): phase times of
the LAST workgroup of the latest small-problem pairwise pass -- the violations pass of a complete solve's last round, which also
stages the kinematics of the QP's solution, selects around the new positions and leaves the rel-step sums (100 MHz stamps)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))

import torch  # noqa: E402

from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402

NAMES = ["staging (strided loads, kinematics of x_tm)", "pair rows (violations + speculative selection)", "workgroup's own sub-lists",
         "fence + ticket (waiting to be last: other workgroups)", "acquire fence", "partials of all workgroups",
         "emit: violated rows -> list, bitmap", "emit: speculative selection -> list, bitmap (cleared first)",
         "rel-step sums", "statistics, host mirror"]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2)
    s = SCP(N, 10.0, 0.2, 0.8, space, dim=2, verbose=False)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    for _ in range(3):
        s.generate_trajectories(15)
    torch.cuda.synchronize()
    lib = s._ctx.lib
    if not hasattr(lib, "scp_debug_small_clocks"):
        raise SystemExit("profiling build needed: SCP_HIP_LIB=ba-path-planning_amd/lib/libscp_hip_prof.so")
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.scp_debug_small_clocks(buf, 16) == 0
    t = [int(v) for v in buf[:11]]
    print(f"N = {N}: last workgroup of the latest small-problem pass, us per phase:")
    for i, name in enumerate(NAMES):
        print(f"  {name:62s} {(t[i + 1] - t[i]) / 100.0:7.2f}")
    print(f"  {'total inside that workgroup':62s} {(t[10] - t[0]) / 100.0:7.2f}")


if __name__ == "__main__":
    main()
