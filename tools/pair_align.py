"""Developer tool: does the duration of the linearisation kernel depend on WHERE its three write streams (eta plane 0, eta
plane 1, l) start relative to each other?  One big device buffer; eta at its start, l at a sweep of byte offsets behind it."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ba-path-planning_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from path_planning import _hip  # noqa: E402
from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402
from path_planning.solvers.scp import SCP  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    K, D = 50, 2
    p0, pf, space = generate_grid_swap(N, seed=1000 * N)
    s = SCP(N, K * 0.2 + 1e-9, 0.2, 0.8, space, verbose=False)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc0 = s._solve_initial_trajectory()
    pos, _ = s._kinematics(acc0, want_vel=False)
    P0, V0, _, _ = s._states()
    ctx = s._ctx
    pp = _hip.PairPass(ctx, N, K, D, 0.8, 0.2)
    rows, stride = pp.rows, pp.stride
    big = torch.empty(D * stride + rows + (1 << 22), dtype=torch.float64, device=ctx.tdev)
    base = big.data_ptr()
    print(f"N = {N}: {rows} rows, eta plane stride {stride * 8} B (mod 4096: {stride * 8 % 4096}, mod 32768: {stride * 8 % 32768}), "
          f"buffer base mod 2 MiB: {base % (1 << 21)}")
    for off in (0, 16, 64, 256, 512, 1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 65536, 131072, 1 << 20):
        e0 = D * stride + off // 8
        pp._eta = big[: D * stride]
        pp._l = big[e0: e0 + rows + 2]
        ts = []
        for _ in range(12):
            pp.linearize(pos, P0, V0, 0.5)
            ts.append(pp.last_linearize_ms * 1e3)
        ts = np.array(ts[2:])
        rel = (pp._l.data_ptr() - base)
        print(f"l at eta + 2 planes + {off:8d} B  (l - eta0 mod 4096 = {rel % 4096:5d}, mod 32768 = {rel % 32768:6d}): "
              f"median {np.median(ts):7.1f} us  min {ts.min():7.1f}  ({rows * 24 / np.median(ts) / 1e3:6.0f} GB/s)", flush=True)
    # separate allocations, as PairPass / the native solver make them
    for rep in range(3):
        pp._eta = ctx.empty(max(D * stride, 2))
        pp._l = ctx.empty(rows + 2)
        ts = []
        for _ in range(12):
            pp.linearize(pos, P0, V0, 0.5)
            ts.append(pp.last_linearize_ms * 1e3)
        ts = np.array(ts[2:])
        print(f"separate allocations #{rep}: eta mod 2 MiB {pp._eta.data_ptr() % (1 << 21)}, l mod 2 MiB {pp._l.data_ptr() % (1 << 21)}, "
              f"l - eta0 mod 32768 = {(pp._l.data_ptr() - pp._eta.data_ptr()) % 32768}: median {np.median(ts):7.1f} us  min {ts.min():7.1f}", flush=True)


if __name__ == "__main__":
    main()
