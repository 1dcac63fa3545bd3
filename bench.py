#!/usr/bin/env python3
"""SCP iterations/sec (N agents x K waypoints) on MI355X -- the metric of BASELINE.json.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W                     (N > 1: one rank per GPU over RCCL)

Workload (config.workload): 1024 agents x 50 timesteps, 2-D, fp64, grid-swap scenario seed 1000*N,
h = 0.2 s, T = 10 s, R = 0.8 m -- BASELINE.json configs[2] (configs[1] "64 x 50" is a parity-test case).
One step = one pass of the SCP loop body (scp.py:152-166): linearise ALL N(N-1)/2*K pairs around the previous
trajectories, solve the joint QP (fixed rows + collision rows, exact constraint generation), relative-step test.
Every step is the FIRST SCP iteration after QP#0 (the heaviest one: largest violations, most ADMM iterations), run
from the same device-resident state, so each timed step is identical work; inputs are in HBM before timing starts.
With N > 1 the same problem is split over the ranks (pair-range sharding of the O(N^2 K) passes, allgather of the
per-shard trajectories and compact working rows): total work is fixed -> "scaling": "strong".

The JSON line also carries
  roofline     : the pairwise linearisation kernel (HBM-write bound): algorithmic bytes per launch / its average
                 duration, measured live with HIP events around that launch on the stream it runs on; next to it the
                 ADMM iteration time of the joint QP (96 % of the step in round 1, latency bound) and the whole step;
  parity_max_abs: max |GPU - C oracle| over the accelerations of the timed step (same x0, same settings);
  cpu_baseline : the CPU oracle (oracle/scp_oracle_c.c, single-threaded C restatement of the same algorithm =
                 "port") timed on rank 0's host on one complete step from the same input state (~30 s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


import numpy as np  # noqa: E402  (cpu_baseline only)


def cpu_baseline(N, K, D, h, T, R, space, p0, pf, margin, step_info, x0, x_gpu):
    """The C oracle (oracle/scp_oracle_c.c: single-threaded CPU statement of the same algorithm, pinned against
    the numpy oracle and through it against the reference's golden vectors) runs ONE complete step -- the same
    step as the GPU, from the same input state x0 -- on one host core."""
    from oracle import c_oracle as co
    from oracle import qp_oracle as qo
    from oracle import scp_oracle as so

    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception:
        pass
    prob = so.make_problem(N, T, h, R, space, p0, pf)
    t0 = time.perf_counter()
    pos, _ = co.kinematics(prob, x0)
    eta, l_col, dist = co.linearize_pairs(prob, pos)
    t1 = time.perf_counter()
    x1, info = co.admm(prob, eta, l_col, dist, x0, qo.Settings(max_iter=10000, margin=margin))
    t2 = time.perf_counter()
    step_s = t2 - t0
    # parity at the benchmarked configuration: the GPU step's result against the C oracle's, same input state
    pos_c, _ = co.kinematics(prob, x1)
    pos_g, _ = co.kinematics(prob, x_gpu)
    parity = {"parity_max_abs": float(np.abs(x_gpu - x1).max()), "parity_max_abs_positions": float(np.abs(pos_g - pos_c).max()),
              "parity_note": ("max |GPU - C oracle| over the accelerations (m/s^2) / positions (m) of the timed step, both from "
                              "the same x0 with the same settings; each side stops at an eps = 1e-3 solution of the same QP")}
    return {
        **parity,
        "value": 1.0 / step_s,
        "unit": "SCP iterations/s",
        "cores": 1,
        "kind": "port",
        "sample": (f"C oracle, 1 thread, one complete step from the same state: linearisation of all rows {t1-t0:.2f}s + "
                   f"joint QP {t2-t1:.2f}s ({info['iter']} ADMM iterations, {info['working_rows']} working rows, "
                   f"{info['rounds']} rounds, status {info['status']}; the GPU step took {step_info['iter']} iterations, "
                   f"{step_info['working_rows']} rows, {step_info['rounds']} rounds)"),
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--timesteps", type=int, default=50)
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo for one-GPU rehearsals")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    N, K, D, h, R = args.agents, args.timesteps, args.dim, 0.2, 0.8
    T = K * h + 1e-9
    p0, pf, space = generate_grid_swap(N, seed=1000 * N, dim=D)
    solver = SCP(N, T, h, R, space, dim=D, device=local_rank, verbose=False, rank=rank, world_size=world)
    assert solver.K == K
    solver.set_initial_states(p0)
    solver.set_final_states(pf)
    # untimed setup: bounds, QP#0, device-resident starting state
    solver._precompute_constraint_matrices()
    acc0 = solver._solve_initial_trajectory()
    pp = solver._ensure_pairs()

    def step():
        """-> (new accelerations, rel. step, device ms of the linearisation kernel, of the last violations kernel)"""
        if world == 1:  # the loop body as ONE library call (scp_solver_step): what generate_trajectories runs per iteration
            new, info = solver.scp_iteration(acc0)
            return new, info["rel_step"], info["linearize_ms"], info["violations_ms"]
        # sharded: the same phases natively (scp_solver_shard_*), split only at the exchanges (allgather of the per-shard
        # trajectories once per iteration, of the selected / violated row ids once per constraint-generation round)
        new, info = solver.scp_iteration_sharded(acc0)
        return new, info["rel_step"], info["linearize_ms"], info["violations_ms"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed step linearises with the ROW-WRITING kernel (scp_linearize_pairs: the roofline kernel, 24 B per row to HBM --
    # what SCP._add_collision_constraints returns); the row-free step (scp_select_pairs + recomputed working rows: the default
    # of generate_trajectories, same bits) is timed right after it and reported beside it.
    solver.row_free = world > 1  # (a sharded step is row-free: a rank holds no other rank's rows)
    for _ in range(args.warmup):
        step()
    lin_ms, viol_ms, infos = [], [], []
    barrier()
    comm0 = (solver.shard.comm_seconds, solver.shard.comm_calls)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, rel, lms, vms = step()
        lin_ms.append(lms)
        viol_ms.append(vms)
        infos.append(dict(solver._last_qp_info, rel_step=rel))
    barrier()
    dt = time.perf_counter() - t0
    comm = ((solver.shard.comm_seconds - comm0[0]) / args.steps * 1e3, (solver.shard.comm_calls - comm0[1]) / args.steps)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    row_free = None
    if world == 1:
        solver.row_free = True
        for _ in range(max(args.warmup, 1)):
            step()
        sel_ms = []
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            _, rel_rf, lms, _ = step()
            sel_ms.append(lms)
        torch.cuda.synchronize()
        dt_rf = time.perf_counter() - t1
        rf_info = dict(solver._last_qp_info)
        row_free = {
            "ms_per_step": dt_rf / args.steps * 1e3, "value": args.steps / dt_rf, "unit": "SCP iterations/s",
            "select_pass_avg_ms": float(np.mean(sel_ms)),
            "same_result": bool(rel_rf == infos[-1]["rel_step"] and rf_info["iter"] == infos[-1]["iter"]
                                and rf_info["working_rows"] == infos[-1]["working_rows"]),
            "note": "the same step with pair_pass_kernel<D,SELECT> (no eta / l planes written or allocated) and the working "
                    "rows recomputed from the linearisation point: the default of generate_trajectories, bit-identical",
        }
        solver.row_free = False

    # HBM traffic of that kernel from the committed PMC passes (profiles/README.md): same workload only, and only while
    # the kernel source is the one the counters were collected on (otherwise null: a stale figure helps nobody)
    traffic, traffic_note, viol_traffic = None, None, None
    tpath = os.path.join(ROOT, "profiles", "r03_pairwise_traffic.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "r02_pairwise_traffic.json")
    if world == 1 and os.path.exists(tpath):
        import hashlib

        with open(tpath) as f:
            tj = json.load(f)
        # (the headline workload at the top level, other workloads -- config 4 -- under "configs")
        tcfg = tj if (N, K, D) == (1024, 50, 2) else tj.get("configs", {}).get(f"{N}x{K}x{D}")
        src = os.path.join(ROOT, "ba-path-planning_amd", "csrc", "scp_kernels.hip")
        sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
        if tcfg is None:
            pass
        elif tj.get("kernel_source_sha256") == sha:
            traffic = tcfg["linearize"]["hbm_bytes"]
            viol_traffic = tcfg.get("violations_recompute", {}).get("hbm_bytes")
        else:
            traffic_note = "scp_kernels.hip changed since the PMC passes were collected: re-run tools/collect_profiles.sh"

    # roofline of the dominant pairwise kernel (per rank: its own shard of rows)
    rows = pp.rows
    alg_bytes = rows * 8 * (D + 1) + 2 * N * K * D * 8  # 24 B/row (D=2) + the two trajectory arrays
    if world > 1:  # the sharded step runs the row-free selection pass: its algorithmic traffic is the trajectory array only
        alg_bytes = N * K * D * 8
    avg_ms = float(np.mean(lin_ms)) if lin_ms else float("nan")
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    out = {
        "metric": "SCP iterations/sec (N agents x K waypoints)",
        "value": args.steps / dt,
        "unit": "SCP iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{N} agents x {K} timesteps, {D}-D, grid-swap seed {1000*N}; step = first SCP iteration after QP#0 "
                        f"(linearise {N*(N-1)//2*K} rows + joint QP + rel-step)",
            "agents": N, "timesteps": K, "dim": D, "collision_rows": N * (N - 1) // 2 * K,
            "parallelism": f"pair-range shard x{world}" if world > 1 else "single GPU",
            "qp": {k: infos[-1][k] for k in ("iter", "cg_iters_total", "working_rows", "rounds", "status", "rho_updates")},
            "rel_step": infos[-1]["rel_step"],
        },
        "roofline": {
            "kernel": ("pair_pass_kernel<D,LINEARIZE> (scp_linearize_pairs)" if world == 1 else
                       "pair_pass_kernel<D,SELECT> (scp_select_pairs over this rank's pair range: the row-free pass writes no "
                       "rows, its only HBM traffic is the trajectory array -- fp64 VALU / LDS bound, frac is not a target here; "
                       "the HBM-bound row-writing kernel is measured by the 1-GPU line)"),
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "bytes_per_launch": alg_bytes, "rows_per_launch": rows, "avg_launch_ms": avg_ms,
            # the rest of the step, for scale: the violations pass recomputes its rows (no HBM stream: fp64 VALU / LDS
            # bound), and the joint QP is a chain of ~500 dependent ADMM steps whose whole state (14 MB) stays on chip:
            # latency bound by construction, so the step as a whole sits at a percent of the HBM roofline
            "violations_pass_avg_ms": float(np.mean(viol_ms)) if viol_ms else None,
            "violations_pass": {
                "kernel": "pair_pass_kernel<D,VIOL_RECOMPUTE> (scp_collision_violations_at)",
                "bound": "fp64 VALU / LDS (eta, R - dist recomputed from the LDS-resident linearisation point: no row stream)",
                "avg_launch_ms": float(np.mean(viol_ms)) if viol_ms else None,
                "rows_per_launch": rows, "traffic": viol_traffic,
                "rows_per_s": rows / (float(np.mean(viol_ms)) * 1e-3) if viol_ms else None,
            },
            "admm": {"iterations": int(infos[-1]["iter"]), "solve_ms": float(infos[-1]["solve_ms"]),
                     "us_per_iteration": float(infos[-1]["solve_ms"]) * 1e3 / max(int(infos[-1]["iter"]), 1),
                     "note": "device time of the QP solve (HIP events) / ADMM iterations: persistent kernel, two "
                             "tagged-granule exchanges per iteration"},
            "whole_step": {"algorithmic_hbm_bytes": alg_bytes, "ms": dt / args.steps * 1e3,
                           "frac": alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
        },
    }
    out["row_free_step"] = row_free
    if world > 1:
        out["exchange"] = {"ms_per_step": comm[0], "collectives_per_step": comm[1], "backend": args.backend,
                           "note": "host wall time of rank 0 inside the exchanges of one step (allgather of the per-shard "
                                   "trajectories, of the selected / violated row ids per constraint-generation round); with "
                                   "gloo this includes the device <-> host staging copies"}
    out["config"]["qp"]["pipeline"] = infos[-1].get("pipeline")
    out["config"]["qp"]["persist_gave_up"] = infos[-1].get("persist_gave_up")
    if traffic_note:
        out["roofline"]["traffic_note"] = traffic_note
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        x_gpu = step()[0]
        out["cpu_baseline"] = cpu_baseline(N, K, D, h, T, R, space, p0, pf, solver.working_set_margin, infos[-1],
                                           acc0.cpu().numpy(), x_gpu.cpu().numpy())
        out["parity_max_abs"] = out["cpu_baseline"]["parity_max_abs"]
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
