"""``compute-trajectories-batch``: timing benchmark over robot counts and trials -> JSON + CSV
(entry point of the reference, /root/reference/src/path_planning/cli/compute_trajectories_batch.py:70-177,
pyproject.toml:54).

The CONFIG dict, the per-run record {N, status, time_sec, error, K, T, h, trial_index}, the summary statistics
and the JSON/CSV schema ("schema_version": "1.0", CSV field order) are the reference's, so ``scp-boxplot``
consumes the files unchanged.  Additions: a per-trial ``seed`` (the reference leaves it as a TODO, :40, :65), the
scenario family, and scenario-parallel execution: with WORLD_SIZE > 1 (torchrun, one process per GPU) the
(N, trial) jobs are dealt round-robin to the ranks, no collective on the data path, and rank 0 merges the
records -- config 5 of BASELINE.json ("512 scenarios x 128 agents, scenario-parallel across 8 GPUs")."""
import argparse
import csv
import json
import os
import time
from datetime import datetime
from pathlib import Path

import numpy as np

from ..scenarios.position_generator import generate_grid_swap, generate_positions
from ..solvers.scp import SCP

# ---------------------------- Config (reference defaults, compute_trajectories_batch.py:14-24) -------------
CONFIG = {
    "Ns": [18, 20],
    "trials_per_N": 10,
    "time_horizon": 10.0,
    "time_step": 0.2,
    "min_distance": 0.8,
    "space_dims": [0, 0, 20, 20],
    "max_iterations": 15,
    "rng_seed": None,
    "results_dir": "data/trial_xxx",
    # additions
    "scenario": "reference",  # or "grid-swap" (needed for N >= 60)
    "dim": 2,
}


def trial_seed(cfg, N, trial):
    """seed = rng_seed + 1000*N + trial (the reference's formula at :108), None when rng_seed is None."""
    if cfg["rng_seed"] is None:
        return None
    return int(cfg["rng_seed"]) + 1000 * N + trial


def make_scenario(N, cfg, seed=None):
    """Start / goal positions and the space box of one trial (compute_trajectories_batch.py:36-41)."""
    if cfg.get("scenario", "reference") == "grid-swap":
        return generate_grid_swap(N, seed=seed or 0, dim=cfg.get("dim", 2))
    init_pos, final_pos = generate_positions(N, cfg["min_distance"], seed=seed)
    return init_pos, final_pos, cfg["space_dims"]


def run_single_trial(N, cfg, rng=None, seed=None, device=None, scenario=None, save_path=None, pool=None):
    """One SCP solve for N vehicles -> result record (compute_trajectories_batch.py:28-67).

    pool: a dict owned by the calling worker (one per thread / stream).  The reference builds a new solver per trial
    (:32-38); here building one -- context, pinned buffers, the QP workspace -- costs more than solving ~100 agents, so
    a worker keeps its solver per problem shape and hands it the next scenario (identical results: every solve starts
    from set_problem / reset)."""
    init_pos, final_pos, space = scenario if scenario is not None else make_scenario(N, cfg, seed)
    solver = None
    t0 = time.perf_counter()
    status = "success"
    err_msg = None
    iters = None
    try:
        key = (N, cfg["time_horizon"], cfg["time_step"], cfg["min_distance"], cfg.get("dim", 2), device)
        solver = pool.get(key) if pool is not None else None
        if solver is None:
            solver = SCP(
                n_vehicles=N,
                time_horizon=cfg["time_horizon"],
                time_step=cfg["time_step"],
                min_distance=cfg["min_distance"],
                space_dims=space,
                dim=cfg.get("dim", 2),
                device=device,
                verbose=cfg.get("verbose", False),
                polish=cfg.get("polish", False),
                carry_rho=cfg.get("carry_rho", False),
                kernel_timing=cfg.get("kernel_timing", True),
                qp_settings=cfg.get("qp_settings"),
            )
            if pool is not None:
                pool[key] = solver
        else:
            solver.set_space_dims(space)
        solver.set_initial_states(init_pos)
        solver.set_final_states(final_pos)
        t0 = time.perf_counter()
        _ = solver.generate_trajectories(max_iterations=cfg["max_iterations"])
        iters = solver.last_info.get("n_iterations")
    except Exception as e:
        status = "error"
        err_msg = str(e)
    t1 = time.perf_counter()
    record = {
        "N": N,
        "status": status,
        "time_sec": t1 - t0,
        "error": err_msg,
        "K": getattr(solver, "K", None),
        "T": getattr(solver, "T", cfg["time_horizon"]),
        "h": getattr(solver, "h", cfg["time_step"]),
        "seed": seed,
        "scp_iterations": iters,
    }
    if status == "success":
        # additions to the reference record: per-iteration wall time, ADMM iterations and final
        # residuals of every QP (QP#0 first), relative steps, convergence flag, minimum pair distance of the result
        info = solver.last_info
        qps = [info["qp0"]] + list(info["iterations"])
        record.update(
            converged=bool(info["converged"]),
            iteration_time_sec=[float(i["time_sec"]) for i in info["iterations"]],
            rel_steps=[float(i["rel_step"]) for i in info["iterations"]],
            qp_iterations=[int(q["iter"]) for q in qps],
            qp_status=[q["status"] for q in qps],
            qp_residuals=[[float(q["r_prim"]), float(q["r_dual"])] for q in qps],
            working_rows=[int(q["working_rows"]) for q in qps],
            # which ADMM pipeline ran each QP, and whether a persistent launch had to give up (CUs taken by another
            # process): a fallen-back solve is slower and differs in the last bits -- visible here, not only in the time
            qp_pipeline=[q.get("pipeline", "none") for q in qps],
            persist_gave_up=int(sum(int(q.get("persist_gave_up", 0)) for q in qps)),
            rho_switches_in_kernel=int(sum(int(q.get("rho_switches_in_kernel", 0)) for q in qps)),
        )
        if cfg.get("validate", False):
            record["min_pair_distance"] = float(solver.validate_solution()["min_pair_distance"])
        if save_path is not None:
            np.savez_compressed(save_path, initial_positions=np.asarray(init_pos), final_positions=np.asarray(final_pos),
                                space_dims=np.asarray(space, dtype=float), **solver.trajectories)
            record["trajectory_file"] = os.path.basename(save_path)
    return record


def summarise(runs, Ns):
    """Per-N statistics over the successful runs (compute_trajectories_batch.py:122-150)."""
    summary = {}
    for N in Ns:
        times = [r["time_sec"] for r in runs if r["N"] == N and r["status"] == "success"]
        errors = sum(1 for r in runs if r["N"] == N and r["status"] != "success")
        if times:
            summary[str(N)] = {
                "count": len(times),
                "errors": errors,
                "min": float(np.min(times)),
                "max": float(np.max(times)),
                "mean": float(np.mean(times)),
                "median": float(np.median(times)),
                "p25": float(np.percentile(times, 25)),
                "p75": float(np.percentile(times, 75)),
                "std": float(np.std(times, ddof=1)) if len(times) > 1 else 0.0,
            }
        else:
            summary[str(N)] = {k: None for k in ("min", "max", "mean", "median", "p25", "p75", "std")}
            summary[str(N)].update(count=0, errors=errors)
    return summary


CSV_FIELDS = ["N", "trial_index", "status", "time_sec", "K", "T", "h", "error"]  # reference order (:158)


def write_results(all_results, json_path, csv_path):
    with open(json_path, "w", encoding="utf-8") as f:
        json.dump(all_results, f, indent=2)
    with open(csv_path, "w", newline="", encoding="utf-8") as f:
        w = csv.DictWriter(f, fieldnames=CSV_FIELDS)
        w.writeheader()
        for r in all_results["runs"]:
            w.writerow({k: r.get(k, None) for k in CSV_FIELDS})


def jobs_for_rank(cfg, rank, world):
    jobs = [(N, t) for N in cfg["Ns"] for t in range(cfg["trials_per_N"])]
    return jobs[rank::world]


def build_parser():
    p = argparse.ArgumentParser(prog="compute-trajectories-batch", description=__doc__.split("\n\n")[0])
    p.add_argument("--Ns", type=int, nargs="+", default=None)
    p.add_argument("--trials", type=int, default=None)
    p.add_argument("--scenario", choices=["reference", "grid-swap"], default=None)
    p.add_argument("--dim", type=int, choices=[2, 3], default=None)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--results-dir", default=None)
    p.add_argument("--max-iterations", type=int, default=None)
    p.add_argument("--save-trajectories", action="store_true",
                   help="write <results-dir>/trajectory_N<N>_t<trial>.npz (positions, velocities, accelerations) per run")
    p.add_argument("--polish", action="store_true",
                   help="one more joint QP at 1e-8 after the SCP loop: the result meets every constraint to ~1e-6")
    p.add_argument("--fresh-solvers", action="store_true",
                   help="build a new solver object for every trial like the reference (:32-38) instead of reusing one per worker")
    p.add_argument("--validate", action="store_true", help="add the minimum pair distance of every result to its record")
    p.add_argument("--warmup", type=int, default=0,
                   help="untimed solves per worker (stream) before the clock starts: the first solve of a worker builds its "
                        "solver object and loads the kernels (~0.1 s); with it the scenarios/s line is the steady-state rate")
    p.add_argument("--qp-persistent", type=int, choices=[0, 1, 2, 3, 4], default=None,
                   help="scp_qp_settings.persistent: 1 = persistent ADMM kernel, variant chosen by size (default), 2 / 3 / 4 = "
                        "force its lean 16-agent form (half the compute units per solve) / lean 8-agent form / round 2's "
                        "kernel, 0 = three launches per ADMM step")
    p.add_argument("--kernel-timing", type=int, choices=[0, 1], default=None,
                   help="HIP events around the pairwise kernels and QP solves (the records' kernel times): default 1 for a single "
                        "stream, 0 with --streams > 1 or several ranks (~25 queue packets fewer per solve; solve_ms is then "
                        "host wall clock, linearize_ms / violations_ms read 0)")
    p.add_argument("--host-wait", type=int, choices=[0, 1, 2], default=None,
                   help="how worker threads wait for the GPU (scp_set_host_wait): 0 spin, 1 spin 20 us then nap (default with "
                        "--streams > 1 or several ranks), 2 nap at once (more workers than host cores)")
    p.add_argument("--carry-rho", action="store_true",
                   help="every joint QP after the first starts at the rho the previous SCP iteration ended with")
    p.add_argument("--streams", type=int, default=1,
                   help="solve this many scenarios concurrently on one GPU, each on its own HIP stream (a solve of "
                        "~100 agents is latency bound and leaves the GPU mostly idle)")
    return p


def main(argv=None):
    args = build_parser().parse_args([] if argv is None else argv)
    if args.streams > 1:
        # HIP maps streams onto 4 hardware queues by default and kernels of streams that share a queue run one after the
        # other -- a persistent QP kernel holds its queue for a millisecond.  Effective only before the runtime starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(max(4, args.streams), 24)))
    cfg = CONFIG.copy()
    for key, val in (("Ns", args.Ns), ("trials_per_N", args.trials), ("scenario", args.scenario), ("dim", args.dim),
                     ("rng_seed", args.seed), ("results_dir", args.results_dir),
                     ("max_iterations", args.max_iterations)):
        if val is not None:
            cfg[key] = val
    cfg["validate"] = bool(args.validate)
    cfg["polish"] = bool(args.polish)
    cfg["carry_rho"] = bool(args.carry_rho)
    many = args.streams > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1
    cfg["kernel_timing"] = bool(args.kernel_timing) if args.kernel_timing is not None else not many
    if args.qp_persistent is not None:
        cfg["qp_settings"] = {"persistent": int(args.qp_persistent)}

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.streams > 1 or world > 1:
        # many solver threads / processes on the host's cores: waiting threads sleep between polls instead of spinning
        from .. import _hip

        _hip.load_library().scp_set_host_wait(1 if args.host_wait is None else args.host_wait)
    elif args.host_wait is not None:
        from .. import _hip

        _hip.load_library().scp_set_host_wait(args.host_wait)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # more ranks than GPUs is a feature: launches of one process serialise in the HIP runtime (8 streams in ONE
        # process top out at ~1.6x one stream at N = 128), several processes per GPU do not
        import torch

        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist

        if not dist.is_initialized():
            dist.init_process_group("gloo")  # control plane only: gathers the result records

    Path(cfg["results_dir"]).mkdir(parents=True, exist_ok=True)
    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    json_path = Path(cfg["results_dir"]) / f"scp_benchmark_{stamp}.json"
    csv_path = Path(cfg["results_dir"]) / f"scp_benchmark_{stamp}.csv"

    if cfg["rng_seed"] is not None:
        np.random.seed(cfg["rng_seed"])

    if rank == 0:
        print("------ WOW SCP Benchmark ------")
        print(f"Robot counts: {cfg['Ns']}, Trials per N: {cfg['trials_per_N']}")
        print(f"T={cfg['time_horizon']}s, h={cfg['time_step']}s, R={cfg['min_distance']}m, space={cfg['space_dims']}")
        print(f"Max SCP iterations: {cfg['max_iterations']}")
        print()

    jobs = jobs_for_rank(cfg, rank, world)
    # scenarios first: the rejection sampling of the grid-swap generator (0.1-0.3 s of host time per scenario at
    # N = 128) would otherwise be what the scenarios/s figure measures
    t_gen = time.perf_counter()
    scenarios = {}
    for N, trial in jobs:
        seed = trial_seed(cfg, N, trial)
        if seed is not None:
            np.random.seed(seed)
        scenarios[(N, trial)] = make_scenario(N, cfg, seed)
    t_gen = time.perf_counter() - t_gen

    import threading

    pools = threading.local()  # one solver pool per worker thread (= per HIP stream)

    def one_job(job):
        N, trial = job
        seed = trial_seed(cfg, N, trial)
        save = str(Path(cfg["results_dir"]) / f"trajectory_N{N}_t{trial}.npz") if args.save_trajectories else None
        if not hasattr(pools, "solvers"):
            pools.solvers = {}
        res = run_single_trial(N, cfg, rng=np.random, seed=seed, device=local_rank, scenario=scenarios[job], save_path=save,
                               pool=None if args.fresh_solvers else pools.solvers)
        res["trial_index"] = trial
        status_str = "OK" if res["status"] == "success" else f"ERR ({res['error']})"
        print(f"  [rank {rank}] N={N} trial {trial+1:02d}/{cfg['trials_per_N']}  time = {res['time_sec']:.3f}s  [{status_str}]")
        return res

    executor = None
    on_stream = None
    if args.streams > 1:
        # scenario-parallel on ONE GPU: a worker thread per HIP stream; ctypes releases the GIL inside the library
        # and every SCP object owns its context, workspace and stream, so the solves overlap on the device
        import threading
        from concurrent.futures import ThreadPoolExecutor

        import torch

        tls = threading.local()

        def on_stream(job):
            if not hasattr(tls, "stream"):
                torch.cuda.set_device(local_rank)
                tls.stream = torch.cuda.Stream(device=local_rank)
            with torch.cuda.stream(tls.stream):
                return one_job(job)

        executor = ThreadPoolExecutor(max_workers=args.streams)
    if args.warmup > 0 and jobs and not args.fresh_solvers:
        import contextlib
        import io

        quiet = io.StringIO()
        with contextlib.redirect_stdout(quiet):
            if executor is None:
                for _ in range(args.warmup):
                    one_job(jobs[0])
            else:
                gate = threading.Barrier(args.streams)  # every worker thread takes exactly one warm-up task

                def warm(_):
                    gate.wait()
                    for _ in range(args.warmup):
                        on_stream(jobs[0])

                list(executor.map(warm, range(args.streams)))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()  # all ranks start their timed loops together (scenario generation takes different times)
    wall_start = time.time()
    t_all = time.perf_counter()
    if executor is None:
        runs = [one_job(j) for j in jobs]
    else:
        with executor as pool:
            runs = list(pool.map(on_stream, jobs))
    wall = time.perf_counter() - t_all
    if jobs:
        print(f"  [rank {rank}] {len(jobs)} scenarios in {wall:.2f}s wall = {len(jobs)/wall:.1f} scenarios/s "
              f"({args.streams} stream(s); scenario generation {t_gen:.2f}s"
              f"{f' and {args.warmup} warm-up solve(s) per stream' if args.warmup > 0 else ''} before the clock)")

    wall_end = time.time()
    if world > 1:
        import torch.distributed as dist

        gathered = [None] * world
        dist.all_gather_object(gathered, (runs, wall_start, wall_end))
        runs = [r for part in gathered for r in part[0]]
        span = max(g[2] for g in gathered) - min(g[1] for g in gathered)
        if rank == 0 and runs:
            print(f"  [all {world} ranks] {len(runs)} scenarios in {span:.2f}s (first start to last end) = "
                  f"{len(runs)/span:.1f} scenarios/s")
    if rank != 0:
        return None
    runs.sort(key=lambda r: (cfg["Ns"].index(r["N"]), r["trial_index"]))

    all_results = {
        "meta": {
            "timestamp": stamp,
            "description": "SCP timing benchmark for multiple N; each entry is a full solve wall time.",
            "config": cfg,
            "schema_version": "1.0",
        },
        "runs": runs,
        "summary": summarise(runs, cfg["Ns"]),
    }
    write_results(all_results, json_path, csv_path)
    print(f"Saved JSON: {json_path}")
    print(f"Saved CSV:  {csv_path}")
    print("\nSummary (success-only times):")
    for N in cfg["Ns"]:
        s = all_results["summary"][str(N)]
        print(f"  N={N}: count={s['count']}, errors={s['errors']}, "
              f"mean={s['mean']}, median={s['median']}, p25={s['p25']}, p75={s['p75']}")
    return all_results


def console_main():
    import sys

    main(sys.argv[1:])


if __name__ == "__main__":
    console_main()
