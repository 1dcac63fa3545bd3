"""BASELINE.json's configurations as they are worded, on the GPU through the product path.

  config 1: "4 agents x 20 timesteps via compute-trajectories"         -> compute_trajectories.main([...])
  config 4: "4096 agents x 50 timesteps" (fits one GPU: 10 GB of rows)  -> tests/test_scp_gpu.py::test_full_size_properties
  config 5: "compute-trajectories-batch: scenarios x 128 agents"        -> compute_trajectories_batch.main([...]) and a
            128-agent grid-swap solve against the C oracle (the numpy oracle needs minutes at this size)
"""
import csv
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import qp_oracle as qo
from oracle import scp_oracle as so

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cg,tol", [(2, 1e-7), (1, 2e-2)])
def test_compute_trajectories_cli_config1(tmp_path, capsys, cg, tol):
    """Config 1 through the entry point: the printed lines of the reference demo (compute_trajectories.py:22-92), the
    two plot files, and the waypoints against the CPU oracle on the same generator scenario -- with two PCG steps iterate
    for iterate (1e-7), on the default single step to the ADMM termination tolerance (on small, nearly
    degenerate QPs that path amplifies the 1e-16 between two summation orders; observed here: 1e-6)."""
    from path_planning.cli import compute_trajectories as cli
    from path_planning.scenarios.position_generator import generate_positions

    pre = str(tmp_path / "demo")
    solver = cli.main(["--n-agents", "4", "--time-horizon", "10", "--time-step", "0.5", "--space", "0", "0", "20", "20",
                       "--seed", "1", "--save-prefix", pre] + (["--cg-iters", "2"] if cg == 2 else []))
    out = capsys.readouterr().out
    for line in ("------ WOW Fleet Collision-Free 2D Trajectory Generation ------", "  Number of vehicles: 4",
                 "Number of timesteps: 20", "Successfully generated positions for 4 vehicles", "Generating trajectories...",
                 "SCP Iteration 1", "Trajectory generation complete!", "Number of time steps: 20",
                 "Total trajectory duration: 10.0 seconds", "Visualizing 2D trajectories...", "Visualizing time snapshots"):
        assert line in out, line
    assert solver is not None and solver.K == 20
    assert os.path.getsize(pre + "_2d.pdf") > 0 and os.path.getsize(pre + "_snapshots.pdf") > 0
    p0, pf = generate_positions(4, 0.8, seed=1)
    prob = so.make_problem(4, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf)
    ref = qo.scp_solve(prob, 15, qo.Settings(max_iter=10000, cg_iters=cg))
    assert solver.last_info["n_iterations"] == ref["iterations"]
    np.testing.assert_allclose(solver.trajectories["positions"], ref["positions"], rtol=0, atol=tol)
    # the reference swallows every error and prints it (compute_trajectories.py:98-99)
    assert cli.main(["--n-agents", "2", "--time-horizon", "1", "--time-step", "0.2", "--space", "0", "0", "200", "200",
                     "--seed", "3", "--no-plots"]) is None
    assert "Error during trajectory generation: OSQP failed" in capsys.readouterr().out


def test_batch_cli_config5(tmp_path, capsys):
    """Config 5 through the entry point with the real solver: 128-agent grid-swap scenarios; JSON / CSV schema of the
    reference (compute_trajectories_batch.py:57-66, :91-100, :122-164), the added per-iteration fields, and one record's
    waypoints against a direct solve of the same scenario."""
    from path_planning.cli import compute_trajectories_batch as cli
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    res = cli.main(["--Ns", "128", "--trials", "2", "--scenario", "grid-swap", "--seed", "7", "--results-dir", str(tmp_path),
                    "--save-trajectories", "--validate"])
    out = capsys.readouterr().out
    assert "------ WOW SCP Benchmark ------" in out and "Saved JSON:" in out and "Summary (success-only times):" in out
    files = sorted(os.listdir(tmp_path))
    jfile = [f for f in files if f.endswith(".json")][0]
    cfile = [f for f in files if f.endswith(".csv")][0]
    data = json.load(open(tmp_path / jfile))
    assert set(data) == {"meta", "runs", "summary"} and data["meta"]["schema_version"] == "1.0"
    assert set(data["meta"]) == {"timestamp", "description", "config", "schema_version"}
    assert len(data["runs"]) == 2 and res["runs"] == data["runs"]
    for r in data["runs"]:
        assert {"N", "status", "time_sec", "error", "K", "T", "h", "trial_index"} <= set(r)  # the reference's record
        assert r["N"] == 128 and r["K"] == 50 and r["status"] == "success" and r["error"] is None
        n_it = r["scp_iterations"]
        assert len(r["iteration_time_sec"]) == len(r["rel_steps"]) == n_it and len(r["qp_iterations"]) == n_it + 1
        assert len(r["qp_residuals"]) == n_it + 1 and all(len(q) == 2 for q in r["qp_residuals"])
        # which ADMM pipeline ran each QP (QP#0: column-local kernel; the joint QPs: the persistent kernel -- 2-D: the lean
        # 8-agent form -- and no fallback)
        assert r["qp_pipeline"][0] == "qp0", r["qp_pipeline"]
        assert all(p == "persistent8-lean" for p in r["qp_pipeline"][1:]), r["qp_pipeline"]
        assert r["persist_gave_up"] == 0 and r["rho_switches_in_kernel"] >= 0
        assert sum(r["iteration_time_sec"]) <= r["time_sec"] and all(s in ("solved", "solved inaccurate") for s in r["qp_status"])
        if r["converged"]:
            assert r["min_pair_distance"] >= 0.8 - 0.02
    s = data["summary"]["128"]
    assert set(s) == {"count", "errors", "min", "max", "mean", "median", "p25", "p75", "std"} and s["count"] == 2
    rows = list(csv.DictReader(open(tmp_path / cfile)))
    assert list(rows[0].keys()) == ["N", "trial_index", "status", "time_sec", "K", "T", "h", "error"] and len(rows) == 2
    # waypoints of trial 1 == a direct solve of the same scenario (seed = rng_seed + 1000 N + trial, :108)
    rec = data["runs"][1]
    saved = np.load(tmp_path / rec["trajectory_file"])
    p0, pf, space = generate_grid_swap(128, seed=rec["seed"])
    np.testing.assert_array_equal(saved["initial_positions"], p0)
    solver = SCP(128, 10.0, 0.2, 0.8, space, verbose=False)
    solver.set_initial_states(p0)
    solver.set_final_states(pf)
    traj = solver.generate_trajectories(15)
    np.testing.assert_array_equal(saved["positions"], traj["positions"])  # deterministic run to run
    assert [int(q["iter"]) for q in [solver.last_info["qp0"]] + solver.last_info["iterations"]] == rec["qp_iterations"]


def test_batch_cli_streams_give_identical_records(tmp_path, capsys):
    """Config 5's scenario-parallel mode (worker threads, one HIP stream and one reused solver object each, warm-up solve)
    against the one-at-a-time path: same scenarios -> the same waypoints bit for bit and the same iteration counts.  (If a
    persistent QP kernel ever gave up under co-scheduling, its solve would continue on the three-launch pipeline and
    differ at the 1e-9 level: the bitwise comparison would show it.)"""
    from path_planning.cli import compute_trajectories_batch as cli

    common = ["--Ns", "128", "--trials", "8", "--scenario", "grid-swap", "--seed", "11", "--save-trajectories"]
    one = cli.main(common + ["--results-dir", str(tmp_path / "one")])
    par = cli.main(common + ["--results-dir", str(tmp_path / "par"), "--streams", "4", "--warmup", "1"])
    out = capsys.readouterr().out
    assert "4 stream(s)" in out and "1 warm-up solve(s) per stream" in out
    assert len(one["runs"]) == len(par["runs"]) == 8
    for a, b in zip(one["runs"], par["runs"]):
        assert a["status"] == b["status"] == "success" and a["seed"] == b["seed"] and a["trial_index"] == b["trial_index"]
        assert a["qp_iterations"] == b["qp_iterations"] and a["scp_iterations"] == b["scp_iterations"]
        assert a["working_rows"] == b["working_rows"] and a["rel_steps"] == b["rel_steps"]
        ta = np.load(tmp_path / "one" / a["trajectory_file"])
        tb = np.load(tmp_path / "par" / b["trajectory_file"])
        for key in ("positions", "velocities", "accelerations"):
            np.testing.assert_array_equal(ta[key], tb[key])


@pytest.mark.parametrize("cg,tol", [(2, 1e-6), (1, 2e-2)])
def test_scp_128_agents_vs_c_oracle(cg, tol):
    """The unit of config 5: one 128-agent grid-swap solve against the C oracle's SCP loop.  Two PCG steps: iterate for
    iterate (equal ADMM counts per QP, waypoints to 1e-6); the default single step: to the ADMM termination tolerance
    (see tests/test_scp_gpu.py::test_scp_sweep_vs_oracle)."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    N = 128
    p0, pf, space = generate_grid_swap(N, seed=4)
    s = SCP(N, 10.0, 0.2, 0.8, space, verbose=False, qp_settings={"cg_iters": cg})
    s.set_initial_states(p0)
    s.set_final_states(pf)
    traj = s.generate_trajectories(15)
    prob = so.make_problem(N, 10.0, 0.2, 0.8, space, p0, pf)
    ref = co.scp_solve(prob, 15, qo.Settings(max_iter=10000, cg_iters=cg))
    assert s.last_info["n_iterations"] == ref["iterations"] and s.last_info["converged"] == ref["converged"]
    gi = [s.last_info["qp0"]["iter"]] + [q["iter"] for q in s.last_info["iterations"]]
    ci = [q["iter"] for q in ref["infos"]]
    if cg > 1:
        assert gi == ci
        assert [q["working_rows"] for q in s.last_info["iterations"]] == [q["working_rows"] for q in ref["infos"][1:]]
    else:
        assert all(abs(a - b) <= 50 for a, b in zip(gi, ci)), (gi, ci)
    np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=tol)
    np.testing.assert_allclose([q["rel_step"] for q in s.last_info["iterations"]], ref["rel_steps"], rtol=0.05, atol=2e-3)


@pytest.mark.parametrize("n,persistent", [(128, 1), (128, 0), (128, 4), (128, 2), (40, 1)])
def test_adaptive_check_cadence(n, persistent):
    """settings.check_fine = 5 (after a check that finds the residuals within 4 x their tolerances, or that changes rho, the
    next check comes after 5 steps instead of 25).  Two PCG steps: the host loop (QP#0 kernel, fused path) follows the C
    oracle's cadence iterate for iterate.  One PCG step: every persistent kernel (default lean 8-agent, lean 16-agent,
    round 2's) takes the same decisions inside the kernel as the host loop of the three-launch pipeline does -- the same
    counts up to the summation-order effect the fixed cadence has too (here: a few fine intervals)."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    p0, pf, space = generate_grid_swap(n, seed=4)
    prob = so.make_problem(n, 10.0, 0.2, 0.8, space, p0, pf)

    def gpu(**qp):
        s = SCP(n, 10.0, 0.2, 0.8, space, verbose=False, qp_settings=dict(check_fine=5, **qp))
        s.set_initial_states(p0)
        s.set_final_states(pf)
        traj = s.generate_trajectories(15)
        return s, traj, [s.last_info["qp0"]["iter"]] + [q["iter"] for q in s.last_info["iterations"]]

    if persistent == 1:  # (a) iterate for iterate against the oracle
        s, traj, gi = gpu(cg_iters=2)
        ref = co.scp_solve(prob, 15, qo.Settings(max_iter=10000, cg_iters=2, check_fine=5))
        assert gi == [q["iter"] for q in ref["infos"]], (gi, [q["iter"] for q in ref["infos"]])
        assert any(i % 25 for i in gi)  # (the fine cadence did end at least one QP between two coarse checks)
        np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=1e-6)
    # (b) the default single PCG step on the chosen pipeline against the C oracle's single step
    s, traj, gi = gpu(persistent=persistent)
    ref = co.scp_solve(prob, 15, qo.Settings(max_iter=10000, check_fine=5))
    ci = [q["iter"] for q in ref["infos"]]
    want = {0: "three-launch", 1: "persistent8-lean", 2: "persistent16", 4: "persistent"}[persistent]
    assert all(want in q["pipeline"] for q in s.last_info["iterations"]), [q["pipeline"] for q in s.last_info["iterations"]]
    assert s.last_info["n_iterations"] == ref["iterations"]
    assert gi[0] == ci[0] and all(abs(a - b) <= 25 for a, b in zip(gi, ci)), (gi, ci)
    assert all(i % 5 == 0 for i in gi)
    np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=2e-2)


def test_config3_full_solve_vs_c_oracle():
    """BASELINE config 3 end to end: the complete 1024 x 50 solve (QP#0 + every SCP iteration, default settings: one PCG
    step, persistent kernel, native loop) against the C oracle's complete solve of the same scenario (~15 s on one host
    core).  The same ADMM counts per QP and waypoints to 1e-6: at this size the single-step path does not separate from
    the oracle (bench.py reports 3.5e-13 on the first step)."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    N, K = 1024, 50
    p0, pf, space = generate_grid_swap(N, seed=1000 * N)
    s = SCP(N, K * 0.2 + 1e-9, 0.2, 0.8, space, verbose=False)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    traj = s.generate_trajectories(15)
    prob = so.make_problem(N, K * 0.2 + 1e-9, 0.2, 0.8, space, p0, pf)
    ref = co.scp_solve(prob, 15, qo.Settings(max_iter=10000))
    assert s.last_info["n_iterations"] == ref["iterations"] and s.last_info["converged"] == ref["converged"]
    gi = [s.last_info["qp0"]["iter"]] + [q["iter"] for q in s.last_info["iterations"]]
    assert gi == [q["iter"] for q in ref["infos"]], (gi, [q["iter"] for q in ref["infos"]])
    assert [q["working_rows"] for q in s.last_info["iterations"]] == [q["working_rows"] for q in ref["infos"][1:]]
    np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=1e-6)
    np.testing.assert_allclose([q["rel_step"] for q in s.last_info["iterations"]], ref["rel_steps"], rtol=1e-6)
