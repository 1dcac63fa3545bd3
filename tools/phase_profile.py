"""Developer tool: where one workgroup of the column-block ADMM kernels spends its time.

Needs the profiling build (`make -C ba-path-planning_amd/csrc prof`), which stamps a 100 MHz wall clock at the
phase boundaries of cg1_col_kernel / cg1_resid_col_kernel (middle workgroup, last launch).  Not part of the product path.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SCP_HIP_LIB"] = os.path.join(ROOT, "ba-path-planning_amd", "lib", "libscp_hip_prof.so")
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))

import torch  # noqa: E402

from path_planning import _hip  # noqa: E402
from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402

COLA = ["prologue: operand prefetch, x / F x / z / y, gather G", "r: reverse scans (one wave per column)",
        "p = Minv r (MFMA)", "S0 p, F p: forward scans, r.p", "stores p, S0 p, F p"]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=2, block=4, min_sep=0.3)
    qp = {"persistent": int(sys.argv[2])} if len(sys.argv) > 2 else None  # 2: the lean 16-agent kernel at any size
    s = SCP(N, 10.0, 0.2, 0.8, space, dim=2, verbose=False, qp_settings=qp)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc = s._solve_initial_trajectory()
    s._solve_with_avoidance_constraints(acc)
    torch.cuda.synchronize()
    lib = _hip.load_library()
    pb = (C.c_ulonglong * 16)()
    if hasattr(lib, "scp_debug_persist_clocks") and lib.scp_debug_persist_clocks(pb, 16) == 0 and sum(pb):
        names = ["state load (once per launch)", "gather G, W', r: suffix scans", "p = Minv r (MFMA)",
                 "S0 p, F p: forward scans, publish cells", "rows: poll partner cells, eta . dS0p", "all-gather of the partials",
                 "step length", "updates (registers + entries)", "state write-back (once per launch)"]
        steps = max(int(pb[15]), 1)
        pb[15] = 0
        print(f"cg1_persist_kernel, middle workgroup, last launch ({steps} steps incl. their termination checks), us per step:")
        for i, name in enumerate(names):
            v = pb[i] * 0.01
            print(f"  {name:44s} {v if i in (0, 8) else v / steps:8.2f} us{' (total)' if i in (0, 8) else ''}")
        print(f"  {'sum':44s} {sum(pb) * 0.01:8.2f} us per launch")
    if hasattr(lib, "scp_debug_persist16_clocks") and lib.scp_debug_persist16_clocks(pb, 16) == 0 and sum(pb):
        names = ["state load (once per launch)", "gather G, W', r: suffix scans", "p = Minv r (MFMA, operands from LDS)",
                 "prefix sums of p, S0 p, publish cells", "rows: poll partner cells, eta . dS0p", "all-gather of the partials",
                 "step length", "updates (entries, then registers)", "state write-back (once per launch)", "termination checks (total)"]
        steps = max(int(pb[15]), 1)
        pb[15] = 0
        print(f"cg1_persist16_kernel (lean), middle workgroup, last launch ({steps} steps), us per step:")
        for i, name in enumerate(names):
            v = pb[i] * 0.01
            once = i in (0, 8, 9)
            print(f"  {name:44s} {v if once else v / steps:8.2f} us{' (total)' if once else ''}")
        print(f"  {'sum':44s} {sum(pb) * 0.01:8.2f} us per launch")
        return
    buf = (C.c_ulonglong * 64)()
    assert lib.scp_debug_phase_clocks(buf, 64) == 0
    t = list(buf)
    print("cg1_col_kernel, middle workgroup (10 ns ticks):")
    for i, name in enumerate(COLA):
        print(f"  {name:56s} {(t[i + 1] - t[i]) * 0.01:7.2f} us")
    print(f"  {'total inside the workgroup':56s} {(t[5] - t[0]) * 0.01:7.2f} us")
    names = ["loads x, y_f, gathers", "A^T y (scans)", "delta-y loads, A^T dy", "F x, S0 x (scans)", "epilogue loads / stores",
             "reductions"]
    print("cg1_resid_col_kernel, middle workgroup:")
    for i, name in enumerate(names):
        print(f"  {name:56s} {(t[33 + i] - t[32 + i]) * 0.01:7.2f} us")
    print(f"  {'total inside the workgroup':56s} {(t[38] - t[32]) * 0.01:7.2f} us")


if __name__ == "__main__":
    main()
