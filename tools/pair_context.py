"""Developer tool: why is the linearisation kernel slower inside the SCP step than stand-alone?

Runs the SAME kernel (pair_pass_kernel<2, LINEARIZE>) on the SAME input (the positions of QP#0's solution at N x 50) in
different stream contexts and prints the HIP-event duration around that one launch (scp_ctx_last_pair_ms) for each:

  A  back to back (what tools/pair_bench.py measures)
  B  inside scp_solver_step (what bench.py reports in roofline.avg_launch_ms)
  C  back to back with the device idle for ~5 ms before every launch (host sleep)
  D  after a joint-QP solve (the persistent ADMM kernel) on the same stream, Python-driven
  E  after bounds + kinematics kernels only (the prologue of scp_solver_step without a QP before it)
  F  after a 64 MB device memset (dirty lines of another buffer in L2 / MALL)
  G  back to back again (drift of the box during the experiment)
"""
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


CLOCKS = []  # profiling build only: MHz the kernel's workgroups ran at (median over the workgroups), one per launch


def kernel_mhz(lib, n_wg):
    """profiling build (SCP_HIP_LIB=.../libscp_hip_prof.so): shader cycles / 100 MHz ticks per workgroup of the latest
    pairwise kernel -> the clock it ran at"""
    import ctypes

    if not hasattr(lib, "scp_debug_pair_clocks"):
        return None
    n = 2 * min(n_wg, 4096)
    buf = (ctypes.c_ulonglong * n)()
    torch.cuda.synchronize()
    if lib.scp_debug_pair_clocks(buf, n) != 0:
        return None
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(float)
    a = a[a[:, 1] > 0]
    return float(np.median(a[:, 0] / a[:, 1]) * 100.0) if len(a) else None


def stats(name, xs, rows_bytes):
    xs = np.asarray(xs) * 1e3
    clk = ""
    if CLOCKS:
        clk = f"   kernel clock {np.median(CLOCKS):6.0f} MHz (median of {len(CLOCKS)} launches, min {min(CLOCKS):.0f}, max {max(CLOCKS):.0f})"
        CLOCKS.clear()
    print(f"{name:58s} n={len(xs):2d}  min {xs.min():7.1f}  median {np.median(xs):7.1f}  mean {xs.mean():7.1f}  max {xs.max():7.1f} us"
          f"   ({rows_bytes / np.median(xs) / 1e3:6.0f} GB/s at the median){clk}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    N, K, D, h, R = a.agents, 50, 2, 0.2, 0.8
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=D)
    s = SCP(N, K * h + 1e-9, h, R, space, dim=D, verbose=False)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc0 = s._solve_initial_trajectory()
    pp = s._ensure_pairs()
    qp = s._ensure_qp()
    ctx = s._ctx
    P0, V0, PF, VF = s._states()
    pos, _ = s._kinematics(acc0, want_vel=False)
    nbytes = pp.rows * 8 * (D + 1) + 2 * N * K * D * 8
    margin = s.working_set_margin

    n_wg = ((pp.nq + 1 + 8191) // 8192) * K

    def clock():
        c = kernel_mhz(ctx.lib, n_wg)
        if c is not None:
            CLOCKS.append(c)

    def lin():
        pp.linearize(pos, P0, V0, margin)
        clock()
        return pp.last_linearize_ms

    for _ in range(3):
        lin()
    stats("A back to back", [lin() for _ in range(a.reps)], nbytes)

    s.row_free = False  # (the step on the row-writing kernel, as bench.py times it)
    for _ in range(2):
        s.scp_iteration(acc0)
    xs = []
    for _ in range(a.reps):
        _, info = s.scp_iteration(acc0)
        xs.append(info["linearize_ms"])
    # (no clock for B: the last pairwise kernel of a step is its violations pass)
    stats("B inside scp_solver_step (row-writing)", xs, nbytes)
    xs = []
    for _ in range(a.reps):  # B': the same launch sequence as the step's prologue + linearisation, natively, via max_rounds = 0
        opts = s._native_options()
        opts.max_rounds = 0
        _, rec = s._ensure_native().step(s._limits(), s._space(), P0, V0, PF, VF, opts, acc0)
        clock()
        xs.append(float(rec.linearize_ms))
    stats("B' native step cut after the linearisation (max_rounds = 0)", xs, nbytes)

    xs = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        time.sleep(0.005)
        xs.append(lin())
    stats("C back to back, device idle 5 ms before each launch", xs, nbytes)

    # D: a joint QP solve right before (Python-driven: reset, add rows, solve), then the linearisation
    rows, _, _ = pp.linearize(pos, P0, V0, margin)
    w_eta, w_l = pp.gather(rows)
    xs = []
    for _ in range(max(a.reps // 2, 3)):
        qp.update_settings(max_iter=10000)
        qp.reset(acc0)
        qp.add_rows(rows, w_eta, w_l)
        qp.solve()
        xs.append(lin())
    stats("D after a joint-QP solve (persistent kernel) on the stream", xs, nbytes)

    xs = []
    for _ in range(a.reps):
        qp.set_problem(s._limits(), s._space(), P0, V0, PF, VF)
        ctx.kinematics(N, K, D, h, acc0, P0, V0, False)
        xs.append(lin())
    stats("E after the bounds + kinematics kernels", xs, nbytes)

    junk = torch.empty(64 << 20, dtype=torch.uint8, device=ctx.tdev)
    xs = []
    for _ in range(a.reps):
        junk.zero_()
        xs.append(lin())
    stats("F after a 64 MB memset of another buffer", xs, nbytes)

    stats("G back to back (again)", [lin() for _ in range(a.reps)], nbytes)


if __name__ == "__main__":
    main()
