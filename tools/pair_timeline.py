"""Developer tool (profiling build: SCP_HIP_LIB=.../libscp_hip_prof.so): when did every workgroup of the linearisation kernel
start and end (100 MHz stamps)?  Prints the kernel's span, the workgroup durations by dispatch round, how many workgroups
were resident over time and what the tail costs.  SCP_PAIR_ABLATE=1 runs the same kernel without its stores.
Not part of the product path."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from path_planning import _hip  # noqa: E402
from path_planning.scenarios.position_generator import generate_grid_swap  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--timesteps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    N, K, D = a.agents, a.timesteps, 2
    ctx = _hip.Context(0)
    lib = _hip.load_library() if hasattr(_hip, "load_library") else _hip._lib()
    p0, pf, _ = generate_grid_swap(N, seed=1000 * N, dim=D)
    s = np.linspace(0.0, 1.0, K)[None, :, None]
    pos = p0[:, None, :] + (3 * s**2 - 2 * s**3) * (pf - p0)[:, None, :]
    pos_t, p0_t, v0_t = ctx.tensor(pos), ctx.tensor(p0), ctx.tensor(np.zeros_like(p0))
    pp = _hip.PairPass(ctx, N, K, D, 0.8, 0.2)
    nq = N * (N - 1) // 2
    n_wg = min(-(-(nq + 1) // 8192) * K, 4096)
    print(f"N = {N}, K = {K}: {pp.rows} rows, {n_wg} workgroups (stamps of the first 4096), ablate = {os.environ.get('SCP_PAIR_ABLATE', '0')}")
    for r in range(a.reps):
        pp.linearize(pos_t, p0_t, v0_t, 0.5)
        ms = pp.last_linearize_ms
        torch.cuda.synchronize()
        clk = (ctypes.c_ulonglong * (2 * n_wg))()
        t0 = (ctypes.c_ulonglong * n_wg)()
        if lib.scp_debug_pair_clocks(clk, 2 * n_wg) != 0 or lib.scp_debug_pair_starts(t0, n_wg) != 0:
            print("no stamps: not the profiling build")
            return
        c = np.frombuffer(clk, dtype=np.uint64).reshape(-1, 2).astype(np.int64)
        st = np.frombuffer(t0, dtype=np.uint64).astype(np.int64)
        live = c[:, 1] > 0  # (the linearisation runs a persistent grid: one workgroup per occupancy slot)
        c, st = c[live], st[live]
        n_live = int(live.sum())
        dur = c[:, 1] / 100.0  # us
        mhz = np.median(c[:, 0] / np.maximum(c[:, 1], 1)) * 100.0
        st_us = (st - st.min()) / 100.0
        en_us = st_us + dur
        span = en_us.max()
        order = np.argsort(st_us)
        print(f"launch {r}: events {ms * 1e3:7.1f} us, first start -> last end {span:7.1f} us, clock {mhz:5.0f} MHz, "
              f"sum of workgroup durations / span = {dur.sum() / span:6.1f} resident on average")
        if r != a.reps - 1:
            continue
        q = np.percentile(dur, [5, 50, 95])
        print(f"  {n_live} workgroups; duration: p5 {q[0]:6.1f}  p50 {q[1]:6.1f}  p95 {q[2]:6.1f} us")
        q = np.percentile(en_us, [0, 5, 25, 50, 75, 95, 100])
        print("  end times (us after the first start): min %.1f  p5 %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p95 %.1f  max %.1f" % tuple(q))
        for lo in range(0, n_live, 512):
            idx = order[lo:lo + 512]
            print(f"  workgroups {lo:4d}..{lo + len(idx) - 1:4d} in start order: start {st_us[idx].min():6.1f}..{st_us[idx].max():6.1f} us, "
                  f"duration median {np.median(dur[idx]):6.1f} us, end median {np.median(en_us[idx]):6.1f}")
        step = 5.0
        t = 0.0
        line = []
        while t < span and len(line) < 200:
            line.append(int(np.sum((st_us <= t) & (en_us > t))))
            t += step
        print("  resident workgroups every 5 us: " + " ".join(str(v) for v in line))


if __name__ == "__main__":
    main()
