"""Developer tool: run only the pairwise passes (linearize, violations, check) a few times at a given size --
the target of `rocprofv3 --pmc` runs (HBM traffic, stall breakdown).  Not part of the product path."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))

import numpy as np  # noqa: E402

from path_planning import _hip  # noqa: E402
from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--timesteps", type=int, default=50)
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--margin", type=float, default=0.5)
    a = ap.parse_args()
    N, K, D = a.agents, a.timesteps, a.dim
    ctx = _hip.Context(0)
    p0, pf, _ = generate_grid_swap(N, seed=1000 * N, dim=D)
    s = np.linspace(0.0, 1.0, K)[None, :, None]
    pos = p0[:, None, :] + (3 * s**2 - 2 * s**3) * (pf - p0)[:, None, :]  # smooth straight-line motion
    pos_t, p0_t, v0_t = ctx.tensor(pos), ctx.tensor(p0), ctx.tensor(np.zeros_like(p0))
    pp = _hip.PairPass(ctx, N, K, D, 0.8, 0.2)
    bytes_row = 8 * (D + 1)
    for r in range(a.reps):
        rows, md, fv = pp.linearize(pos_t, p0_t, v0_t, a.margin)
        ms = pp.last_linearize_ms
        print(f"linearize  {ms*1e3:8.1f} us  {pp.rows*bytes_row/ms/1e6:8.1f} GB/s  sel={rows.numel()} min={md:.4f}")
    for r in range(a.reps):  # the row-free form of the same pass (scp_select_pairs): same selection, no row stream
        rows, md, fv = pp.select(pos_t, a.margin)
        ms = pp.last_linearize_ms
        print(f"select     {ms*1e3:8.1f} us  {pp.rows/ms/1e6:8.1f} G rows/s  sel={rows.numel()} min={md:.4f}")
    rows, md, fv = pp.linearize(pos_t, p0_t, v0_t, a.margin)  # (the violations pass works on the bitmap of a linearisation)
    for r in range(a.reps):
        pp.bitmap.zero_()
        new, mv = pp.violations(pos_t, p0_t, v0_t, 1e-6)
        ms = pp.last_violations_ms
        print(f"violations {ms*1e3:8.1f} us  {pp.rows*bytes_row/ms/1e6:8.1f} GB/s  new={new.numel()} maxv={mv:.3e}")
    for r in range(a.reps):
        ctx.check_avoidance(N, K, D, 0.8, pos_t)
        print(f"check      {ctx.last_pair_ms()*1e3:8.1f} us")


if __name__ == "__main__":
    main()
