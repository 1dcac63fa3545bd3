"""End-to-end GPU parity: path_planning.solvers.scp.SCP (HIP path through the C-ABI) against the oracle's SCP
loop (oracle/qp_oracle.py:scp_solve) on identical scenarios, plus size-independent properties at the sizes of
BASELINE.json's configs 2 and 3."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import qp_oracle as qo
from oracle import scp_oracle as so

pytestmark = pytest.mark.gpu
TOL = 1e-7  # fp64 tolerance on waypoints, GPU vs CPU oracle, same algorithm and settings, iterate for iterate (cg_iters >= 2)
# The default x-update (ONE warm-started PCG step per ADMM step) is not contractive during the transient: on small, nearly
# degenerate QPs it can amplify the 1e-16 between two summation orders (numpy vs wave scans / MFMA) by up to 1e7 before both
# sides converge to the same point, so a termination check may fall one interval apart and the two stop at two equally valid
# eps = 1e-3 points.  Default-path comparisons therefore use the solver tolerance; every case is ALSO
# run with two PCG steps, where GPU and oracle agree iterate for iterate.
TOL_DEFAULT = 2e-2
CG_CASES = [(2, TOL), (1, TOL_DEFAULT)]


def ref_scenario(n, seed):
    from path_planning.scenarios.position_generator import generate_positions

    return generate_positions(n, 0.8, seed=seed)


def solve_gpu(n, T, h, R, space, p0, pf, max_iterations=15, dim=2, **kw):
    from path_planning.solvers.scp import SCP

    s = SCP(n_vehicles=n, time_horizon=T, time_step=h, min_distance=R, space_dims=space, dim=dim, verbose=False, **kw)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    traj = s.generate_trajectories(max_iterations=max_iterations)
    return s, traj


def test_released_solvers_are_reused():
    """A NEW SCP object per scenario (the reference's usage, compute_trajectories_batch.py:103-117) adopts the native solver
    of a released object of the same shape: same bits as an object that builds its own, on the same and on other scenarios."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers import scp as scp_mod

    scp_mod.clear_native_pool()
    scen = [generate_grid_swap(32, seed=5 + i) for i in range(3)]

    def run(i, **kw):
        p0, pf, space = scen[i]
        s, traj = solve_gpu(32, 10.0, 0.2, 0.8, space, p0, pf, **kw)
        out = (traj["positions"].copy(), [q["iter"] for q in s.last_info["iterations"]], s.last_info["qp0"]["iter"])
        s.close()
        return out

    own = [run(i, reuse_native=False) for i in range(3)]
    before = scp_mod.native_pool_stats()
    assert before["held"] == 0
    first = run(0)  # (misses, builds its own, returns it to the pool)
    again = [run(i) for i in (1, 2, 0)]  # (each adopts the one released just before)
    after = scp_mod.native_pool_stats()
    assert after["hits"] - before["hits"] == 3 and after["misses"] - before["misses"] == 1 and after["held"] == 1
    for got, want in zip([first] + again, [own[0], own[1], own[2], own[0]]):
        np.testing.assert_array_equal(got[0], want[0])
        assert got[1] == want[1] and got[2] == want[2]
    # another shape or other settings: no adoption
    p0, pf, space = scen[0]
    s, _ = solve_gpu(32, 10.0, 0.2, 0.8, space, p0, pf, qp_settings={"cg_iters": 2})
    assert scp_mod.native_pool_stats()["hits"] == after["hits"]
    s.close()
    scp_mod.clear_native_pool()
    assert scp_mod.native_pool_stats()["held"] == 0


@pytest.mark.parametrize("cg,tol", CG_CASES)
@pytest.mark.parametrize("n,seed,T,h", [(4, 1, 10.0, 0.5), (10, 7, 10.0, 0.2)])
def test_scp_matches_oracle(n, seed, T, h, cg, tol):
    p0, pf = ref_scenario(n, seed)
    s, traj = solve_gpu(n, T, h, 0.8, [0, 0, 20, 20], p0, pf, qp_settings={"cg_iters": cg})
    prob = so.make_problem(n, T, h, 0.8, [0, 0, 20, 20], p0, pf)
    out = qo.scp_solve(prob, 15, qo.Settings(max_iter=10000, cg_iters=cg))
    assert s.last_info["n_iterations"] == out["iterations"]
    assert s.last_info["converged"] == out["converged"]
    assert s.last_info["qp0"]["iter"] == out["infos"][0]["iter"]
    for a, b in zip(s.last_info["iterations"], out["infos"][1:]):
        if cg > 1:
            assert a["iter"] == b["iter"] and a["working_rows"] == b["working_rows"] and a["rounds"] == b["rounds"]
        else:
            assert abs(a["iter"] - b["iter"]) <= 50 and a["rounds"] == b["rounds"]
    np.testing.assert_allclose([i["rel_step"] for i in s.last_info["iterations"]], out["rel_steps"],
                               rtol=1e-6 if cg > 1 else 0.05, atol=0 if cg > 1 else 2e-3)
    for key in ("positions", "velocities", "accelerations"):
        assert traj[key].shape == (n, prob.K, 2) and traj[key].dtype == np.float64
        np.testing.assert_allclose(traj[key], out[key], rtol=0, atol=tol if key == "positions" or cg > 1 else 10 * tol)


def check_solution_properties(s, traj, tol=3e-2):
    """Constraints of the reference QP hold at the returned point, to the ADMM termination tolerance
    eps_abs + eps_rel * max(|Ax|, |z|) = 1e-3 * (1 + ~20 m)."""
    N, K, D, h = s.N, s.K, s.D, s.h
    a, v, p = traj["accelerations"], traj["velocities"], traj["positions"]
    p0 = s.initial_positions.reshape(N, D)
    pf = s.final_positions.reshape(N, D)
    # kinematics consistency (bitwise the reference recurrences)
    prob = so.make_problem(N, s.T, h, s.R, list(s.pos_min) + list(s.pos_max), p0, pf)
    pos_o, vel_o = so.kinematics(prob, a)
    np.testing.assert_array_equal(p, pos_o)
    np.testing.assert_array_equal(v, vel_o)
    # stored sample K-1 is NOT the goal; the goal is reached at the unstored state K
    pK = p[:, K - 1] + h * v[:, K - 1] + 0.5 * h * h * a[:, K - 1]
    vK = v[:, K - 1] + h * a[:, K - 1]
    assert np.abs(pK - pf).max() < tol and np.abs(vK).max() < tol
    assert np.abs(a).max() <= 15 + tol and np.abs(v).max() <= 2 + tol
    assert np.abs(np.diff(a, axis=1)).max() / h <= 20 + tol
    assert (p >= s.pos_min - tol).all() and (p <= s.pos_max + tol).all()


def test_scp_grid_swap_64_config2():
    """BASELINE config 2 (64 agents x 50 steps), D=2: solve, properties, and collision-free result."""
    from path_planning.scenarios.position_generator import generate_grid_swap

    p0, pf, space = generate_grid_swap(64, seed=64000)
    s, traj = solve_gpu(64, 10.0, 0.2, 0.8, space, p0, pf)
    check_solution_properties(s, traj)
    prob = so.make_problem(64, 10.0, 0.2, 0.8, space, p0, pf)
    assert s.last_info["n_iterations"] >= 1
    if s.last_info["converged"]:
        assert so.min_pair_distance(prob, traj["positions"]) >= 0.8 - 0.02


@pytest.mark.parametrize("cg,tol", CG_CASES)
def test_scp_3d_z0_metamorphic(cg, tol):
    """D=3 with z == 0 reproduces the D=2 trajectories (the reference is strictly 2-D)."""
    p0, pf = ref_scenario(6, 3)
    s2, t2 = solve_gpu(6, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf, qp_settings={"cg_iters": cg})
    z = np.zeros((6, 1))
    s3, t3 = solve_gpu(6, 10.0, 0.5, 0.8, [0, 0, -5, 20, 20, 5], np.hstack([p0, z]), np.hstack([pf, z]), dim=3,
                       qp_settings={"cg_iters": cg})
    assert s2.last_info["n_iterations"] == s3.last_info["n_iterations"]
    np.testing.assert_allclose(t3["positions"][:, :, :2], t2["positions"], rtol=0, atol=tol)  # summation order differs (C = 12 vs 18 columns)
    assert np.abs(t3["positions"][:, :, 2]).max() < 1e-12


@pytest.mark.parametrize("cg,tol", CG_CASES)
def test_scp_3d_grid_swap_config2(cg, tol):
    """BASELINE config 2 names 3-D: 64 agents x 50 steps in 3-D against the oracle's first iterations."""
    from path_planning.scenarios.position_generator import generate_grid_swap

    p0, pf, space = generate_grid_swap(64, seed=7, dim=3)
    s, traj = solve_gpu(64, 10.0, 0.2, 0.8, space, p0, pf, max_iterations=2, dim=3, qp_settings={"cg_iters": cg})
    check_solution_properties(s, traj)
    prob = so.make_problem(64, 10.0, 0.2, 0.8, space, p0, pf)
    out = qo.scp_solve(prob, 2, qo.Settings(max_iter=10000, cg_iters=cg))
    assert s.last_info["n_iterations"] == out["iterations"]
    np.testing.assert_allclose(traj["positions"], out["positions"], rtol=0, atol=tol)


def test_api_surface_and_errors(capsys):
    from path_planning.solvers.scp import SCP

    s = SCP(n_vehicles=3, time_horizon=3.0, time_step=0.2, min_distance=0.5, space_dims=[-5, -5, 500, 200])
    out = capsys.readouterr().out
    assert "---=== SCP Problem initialized ===---" in out and "Number of timesteps: 15" in out  # scp.py:93-97
    assert (s.N, s.K, s.T, s.h, s.R) == (3, 15, 3.0, 0.2, 0.5)
    assert s.convergence_tolerance == 1.5e-2 and (s.vel_min, s.vel_max, s.acc_max, s.jerk_max) == (-2, 2, 15.0, 20)
    with pytest.raises(ValueError, match="Trajectories not generated yet"):
        s.visualize_trajectories()
    with pytest.raises(ValueError, match="Trajectories not generated yet"):
        s.visualize_time_snapshots()
    with pytest.raises(AssertionError):
        s.set_initial_states(np.zeros((4, 2)))
    # QP#0 infeasible (goal unreachable in T) -> RuntimeError("OSQP failed: primal infeasible") like scp.py:363-365
    s = SCP(n_vehicles=2, time_horizon=1.0, time_step=0.2, min_distance=0.5, verbose=False)
    s.set_initial_states(np.array([[1.0, 1.0], [3.0, 3.0]]))
    s.set_final_states(np.array([[19.0, 19.0], [15.0, 3.0]]))
    with pytest.raises(RuntimeError, match="OSQP failed: primal infeasible"):
        s.generate_trajectories()
    # a later QP that is infeasible only warns and the loop goes on (scp.py:446-447): the reference's __main__ demo
    s = SCP(n_vehicles=3, time_horizon=3.0, time_step=0.2, min_distance=0.5, space_dims=[-5, -5, 500, 200])
    s.set_initial_states(np.array([[-2.0, -2.0], [0.0, -2.0], [2.0, -2.0]]))
    s.set_final_states(np.array([[2.0, 2.0], [0.0, 2.0], [-2.0, 2.0]]))
    capsys.readouterr()
    traj = s.generate_trajectories(max_iterations=2)
    out = capsys.readouterr().out
    assert "Warning: OSQP status primal infeasible" in out and "SCP Iteration 1" in out
    assert np.isfinite(traj["positions"]).all()


def test_initially_feasible_skips_loop():
    """Agents that never come close: is_feasible is True after QP#0 and the loop body never runs (scp.py:152)."""
    p0 = np.array([[2.0, 2.0], [2.0, 18.0]])
    pf = np.array([[18.0, 2.0], [18.0, 18.0]])
    s, traj = solve_gpu(2, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf)
    assert s.last_info["n_iterations"] == 0 and s.last_info["initially_feasible"]
    check_solution_properties(s, traj)


def test_validate_solution_and_refresh_feasibility():
    """device-side feasibility report; opt-in refresh of is_feasible inside the loop."""
    from path_planning.scenarios.position_generator import generate_grid_swap

    p0, pf, space = generate_grid_swap(36, seed=5)
    s, traj = solve_gpu(36, 10.0, 0.2, 0.8, space, p0, pf)
    rep = s.validate_solution()
    prob = so.make_problem(36, 10.0, 0.2, 0.8, space, p0, pf)
    assert abs(rep["min_pair_distance"] - so.min_pair_distance(prob, traj["positions"])) < 1e-12
    ok, first = so.check_avoidance(prob, traj["positions"])
    assert rep["collision_free"] == ok
    if not ok:
        assert (rep["first_violation"]["timestep"],) + rep["first_violation"]["vehicles"] == first[:3]
    for key in ("acc_violation", "jerk_violation", "vel_violation", "pos_violation", "final_position_error",
                "final_velocity_error"):
        assert rep[key] < 3e-2, (key, rep[key])
    s2, _ = solve_gpu(36, 10.0, 0.2, 0.8, space, p0, pf, refresh_feasibility=True)
    assert s2.last_info["n_iterations"] <= s.last_info["n_iterations"]
    assert s2.validate_solution()["min_pair_distance"] >= 0.8 - 0.011 or not s2.last_info["converged"]


def test_capacity_growth_paths():
    """Working-set capacity of the QP and the selection list of the pairwise pass start tiny and must grow on
    demand (restart with every row collected so far) without changing the result."""
    from path_planning import _hip

    p0, pf = ref_scenario(10, 7)
    ref, t_ref = solve_gpu(10, 10.0, 0.2, 0.8, [0, 0, 20, 20], p0, pf, max_iterations=3)
    small, t_small = solve_gpu(10, 10.0, 0.2, 0.8, [0, 0, 20, 20], p0, pf, max_iterations=3, qp_row_capacity=4, native=False)
    assert small._qp.row_capacity > 4  # (Python-driven loop; the native loop's growth path: tests/test_native_gpu.py)
    assert [i["working_rows"] for i in small.last_info["iterations"]] == [i["working_rows"] for i in ref.last_info["iterations"]]
    np.testing.assert_allclose(t_small["positions"], t_ref["positions"], rtol=0, atol=TOL)
    # selection list of the pairwise pass: capacity 8 -> grows, same (sorted) rows
    ctx = ref._ctx
    prob = so.make_problem(10, 10.0, 0.2, 0.8, [0, 0, 20, 20], p0, pf)
    pos = ctx.tensor(t_ref["positions"])
    big = _hip.PairPass(ctx, 10, prob.K, 2, 0.8, 0.2)
    tiny = _hip.PairPass(ctx, 10, prob.K, 2, 0.8, 0.2, sel_cap=8)
    rows_big, _, _ = big.linearize(pos, ctx.tensor(prob.p0), ctx.tensor(prob.v0), 2.0)
    rows_tiny, _, _ = tiny.linearize(pos, ctx.tensor(prob.p0), ctx.tensor(prob.v0), 2.0)
    assert rows_big.numel() > 8 and tiny.sel_cap >= rows_big.numel()
    np.testing.assert_array_equal(rows_tiny.cpu().numpy(), rows_big.cpu().numpy())


def test_plots_headless(tmp_path):
    p0, pf = ref_scenario(4, 1)
    s, _ = solve_gpu(4, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf, max_iterations=1)
    s.visualize_trajectories(save_path=str(tmp_path / "t.pdf"))
    s.visualize_time_snapshots(num_snapshots=3, save_path=str(tmp_path / "s.pdf"))
    assert (tmp_path / "t.pdf").stat().st_size > 0 and (tmp_path / "s.pdf").stat().st_size > 0


def _two_rank_worker(rank, world, port, out_dir):
    import os
    import sys

    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "ba-path-planning_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    p0, pf, space = generate_grid_swap(40, seed=11)
    for tag, native in (("", True), ("py_", False)):
        # native: the iteration natively, split at its exchange points (scp_solver_shard_*: ids only cross ranks);
        # Python-driven: every call from Python, compact rows (ids + eta + l) exchanged, rank 0's solution broadcast
        s = SCP(40, 10.0, 0.2, 0.8, space, verbose=False, device=0, rank=rank, world_size=world, native=native)
        s.set_initial_states(p0)
        s.set_final_states(pf)
        traj = s.generate_trajectories(max_iterations=3)
        np.save(os.path.join(out_dir, f"{tag}pos_{rank}.npy"), traj["positions"])
        np.save(os.path.join(out_dir, f"{tag}its_{rank}.npy"), np.array([i["iter"] for i in s.last_info["iterations"]]))
        if native:
            np.save(os.path.join(out_dir, f"rows_{rank}.npy"), np.array([i["working_rows"] for i in s.last_info["iterations"]]))
    dist.destroy_process_group()


def test_scp_two_ranks_share_one_gpu(tmp_path):
    """The N>1 path end to end on the GPU: 2 ranks (both on cuda:0, gloo standing in for RCCL) shard the pair
    ranges and the agents, exchange trajectories and compact rows, and reproduce the single-rank solve."""
    import os

    import torch.multiprocessing as mp
    from path_planning.scenarios.position_generator import generate_grid_swap

    port = 29600 + os.getpid() % 300
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "pos_0.npy"), np.load(tmp_path / "pos_1.npy")
    np.testing.assert_array_equal(a, b)
    p0, pf, space = generate_grid_swap(40, seed=11)
    s, traj = solve_gpu(40, 10.0, 0.2, 0.8, space, p0, pf, max_iterations=3)
    assert [i["iter"] for i in s.last_info["iterations"]] == np.load(tmp_path / "its_0.npy").tolist()
    assert [i["working_rows"] for i in s.last_info["iterations"]] == np.load(tmp_path / "rows_0.npy").tolist()
    # the native sharded step merges the ranks' row ids in ascending order = the single-rank working set, and the replicated
    # QP is deterministic: two ranks reproduce the one-rank solve BIT FOR BIT
    np.testing.assert_array_equal(a, traj["positions"])
    pa, pb = np.load(tmp_path / "py_pos_0.npy"), np.load(tmp_path / "py_pos_1.npy")
    np.testing.assert_array_equal(pa, pb)
    assert np.load(tmp_path / "py_its_0.npy").tolist() == np.load(tmp_path / "its_0.npy").tolist()
    np.testing.assert_allclose(pa, traj["positions"], rtol=0, atol=1e-8)


def _rccl_one_rank_worker(rank, port, out_dir):
    import os
    import sys

    import torch
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "ba-path-planning_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from path_planning._sharding import Shard

    dev = torch.device("cuda", 0)
    sh = Shard(10, 0, 1, None, run_collectives_alone=True)
    assert dist.get_backend() == "nccl" and not sh.alone
    pos = torch.arange(10 * 4 * 2, dtype=torch.float64, device=dev).reshape(10, 4, 2)
    assert torch.equal(sh.allgather_positions(pos), pos)
    rows = torch.tensor([3, 7, 11], dtype=torch.int64, device=dev)
    eta = torch.arange(6, dtype=torch.float64, device=dev).reshape(3, 2)
    l = torch.tensor([0.5, 0.25, 0.125], dtype=torch.float64, device=dev)
    r, e, ll = sh.allgather_rows(rows, eta, l)
    assert torch.equal(r, rows) and torch.equal(e, eta) and torch.equal(ll, l)
    ids, mx = sh.allgather_ids(torch.tensor([9, 2, 5], dtype=torch.int64, device=dev), extra=0.75)
    assert ids.tolist() == [2, 5, 9] and mx == 0.75 and ids.is_cuda
    sh._ids_cap = 2  # (the capacity retry: a second, longer message)
    ids, _ = sh.allgather_ids(torch.arange(7, dtype=torch.int64, device=dev))
    assert ids.tolist() == list(range(7))
    t = torch.full((5,), 3.0, dtype=torch.float64, device=dev)
    assert torch.equal(sh.broadcast(t.clone()), t)
    assert sh.broadcast_ints([4, -1, 7]) == [4, -1, 7]
    assert sh.all_min(0.5) == 0.5 and sh.all_max(2.5) == 2.5 and sh.all_min_int(123) == 123
    assert sh.comm_calls >= 5 and sh.comm_seconds > 0.0  # (the timed exchanges: positions, rows, ids x 2, broadcast)
    # the sharded SCP iteration with its exchanges on RCCL (trajectories, row ids per round) against scp_solver_step
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    p0, pf, space = generate_grid_swap(40, seed=11)
    s = SCP(40, 10.0, 0.2, 0.8, space, verbose=False, device=0)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc0 = s._solve_initial_trajectory()
    a, ia = s.scp_iteration(acc0)
    s.shard = Shard(40, 0, 1, None, run_collectives_alone=True)
    b, ib = s.scp_iteration_sharded(acc0)
    assert torch.equal(a, b) and ia["iter"] == ib["iter"] and ia["working_rows"] == ib["working_rows"]
    assert s.shard.comm_calls >= 2  # (selected ids, violated ids of at least one round; the trajectories when agents are split)
    s.close()
    open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.destroy_process_group()


def test_rccl_branch_with_one_rank(tmp_path):
    """Every collective of path_planning/_sharding.py through RCCL (backend "nccl": device tensors, no host staging) on a
    communicator of ONE rank -- the code path a multi-GPU node takes, as far as a single GPU can run it (two ranks on one
    device are refused by RCCL; the two-rank semantics are covered over gloo)."""
    import os

    import torch.multiprocessing as mp

    mp.spawn(_rccl_one_rank_worker, args=(29900 + os.getpid() % 90, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()


@pytest.mark.parametrize("n,T,h,dim", [(1, 2.0, 0.2, 2), (2, 0.8, 0.2, 2), (1, 2.0, 0.5, 3), (3, 1.0, 0.5, 3)])
def test_edge_sizes(n, T, h, dim):
    """Smallest shapes: a single agent (no pairs at all), K = 3 time steps, 3-D with few agents."""
    rng = np.random.default_rng(n * 10 + dim)
    p0 = rng.uniform(2, 4, (n, dim)) + np.arange(n)[:, None] * 3.0
    pf = p0 + rng.uniform(-0.1, 0.1, (n, dim))  # reachable within the jerk limit even at K = 3
    space = [0.0] * dim + [20.0] * dim
    s, traj = solve_gpu(n, T, h, 0.8, space, p0, pf, dim=dim)
    prob = so.make_problem(n, T, h, 0.8, space, p0, pf)
    out = qo.scp_solve(prob, 15, qo.Settings(max_iter=10000))
    assert traj["positions"].shape == (n, prob.K, dim)
    assert s.last_info["n_iterations"] == out["iterations"]
    np.testing.assert_allclose(traj["positions"], out["positions"], rtol=0, atol=TOL)
    rep = s.validate_solution()
    assert rep["collision_free"] and rep["final_position_error"] < 3e-2


@pytest.mark.parametrize("T,K", [(12.9, 64), (13.1, 65), (14.0, 70), (24.1, 120), (26.0, 130), (50.1, 250), (100.1, 500)])
def test_large_K_paths(T, K):
    """K = 64: the persistent kernel's limit (one time step per lane); K = 65: two time steps per lane in the wave scans of
    the three-launch column kernels (and 8 slab rows per thread in the QP#0 kernel); K = 70, 120: more than 64 KiB of LDS
    tiles, 120 the largest fused size; K = 130, 250, 500: the long-horizon column kernel (one workgroup per column, block
    scans; the reference's compute-trajectories demo runs K = 500) with the generic QP#0 and check.  Same oracle, same
    tolerance."""
    from path_planning.scenarios.position_generator import generate_positions

    p0, pf = generate_positions(5, 0.8, seed=2)
    s, traj = solve_gpu(5, T, 0.2, 0.8, [0, 0, 20, 20], p0, pf, max_iterations=2)
    prob = so.make_problem(5, T, 0.2, 0.8, [0, 0, 20, 20], p0, pf)
    assert prob.K == K
    out = qo.scp_solve(prob, 2, qo.Settings(max_iter=10000))
    assert s.last_info["n_iterations"] == out["iterations"]
    np.testing.assert_allclose(traj["positions"], out["positions"], rtol=0, atol=TOL)


def test_large_K_3d():
    """K = 65 in 3-D: two time steps per lane in the wave scans together with the D = 3 row kernels (27 columns: two
    column blocks, the second one partly empty)."""
    from path_planning.scenarios.position_generator import generate_grid_swap

    p0, pf, space = generate_grid_swap(9, seed=21, dim=3)
    s, traj = solve_gpu(9, 13.1, 0.2, 0.8, space, p0, pf, max_iterations=2, dim=3)
    prob = so.make_problem(9, 13.1, 0.2, 0.8, space, p0, pf)
    assert prob.K == 65 and prob.D == 3
    out = qo.scp_solve(prob, 2, qo.Settings(max_iter=10000))
    assert s.last_info["n_iterations"] == out["iterations"]
    np.testing.assert_allclose(traj["positions"], out["positions"], rtol=0, atol=TOL)


@pytest.mark.parametrize("cg", [1, 2])
@pytest.mark.parametrize("kind,n,seed", [("ref", 6, 11), ("ref", 8, 12), ("ref", 12, 13), ("ref", 16, 14),
                                         ("grid", 25, 15), ("grid", 36, 16), ("grid3d", 27, 17)])
def test_scp_sweep_vs_oracle(kind, n, seed, cg):
    """Full solves on more scenarios (reference generator and grid-swap, 2-D and 3-D) against the CPU oracle.

    cg = 2 PCG steps per ADMM step: GPU and oracle agree iterate for iterate (same counts, waypoints to 1e-6).
    cg = 1 (the default, twice as fast): the ADMM map with a single inexact x-update step is not contractive
    during the transient -- on nearly degenerate QPs (one pair active over consecutive steps) it amplifies the
    1e-13 difference between two summation orders (numpy vs MFMA) by up to 1e7 before both iterates converge to
    the same solution -- so the two implementations may stop 25 steps apart at two equally valid eps = 1e-3
    solutions: the tolerance there is the ADMM termination tolerance.  (tests/tools/debug_qp.py shows the growth; the GPU
    itself is deterministic run to run.)"""
    from path_planning.scenarios.position_generator import generate_grid_swap

    dim = 3 if kind == "grid3d" else 2
    if kind == "ref":
        p0, pf = ref_scenario(n, seed)
        space = [0, 0, 20, 20]
    else:
        p0, pf, space = generate_grid_swap(n, seed=seed, dim=dim)
    s, traj = solve_gpu(n, 10.0, 0.2, 0.8, space, p0, pf, max_iterations=4, dim=dim,
                        qp_settings={"max_iter": 2000, "cg_iters": cg})
    prob = so.make_problem(n, 10.0, 0.2, 0.8, space, p0, pf)
    out = qo.scp_solve(prob, 4, qo.Settings(max_iter=2000, cg_iters=cg))
    assert s.last_info["n_iterations"] == out["iterations"]
    for a, b in zip(s.last_info["iterations"], out["infos"][1:]):
        assert a["status_val"] == b["status_val"] and a["rounds"] == b["rounds"], (a, b)
        if cg > 1:
            assert a["iter"] == b["iter"] and a["working_rows"] == b["working_rows"], (a, b)
        else:
            # (three coarse check intervals: one of the seven cases lands 60 steps apart since the adaptive check cadence
            # moved where its two constraint-generation rounds end -- 165 against 225, the same final rows and rel. step)
            assert abs(a["iter"] - b["iter"]) <= 75 and abs(a["working_rows"] - b["working_rows"]) <= 2, (a, b)
    # cg = 1: eps_abs + eps_rel * max|Ax| = 1e-3 * (1 + ~20 m), the primal residual either solver may stop at
    np.testing.assert_allclose(traj["positions"], out["positions"], rtol=0, atol=2e-2 if cg == 1 else 1e-6)
    np.testing.assert_allclose([i["rel_step"] for i in s.last_info["iterations"]], out["rel_steps"], rtol=0.05, atol=2e-3)
    check_solution_properties(s, traj)


def test_config4_step_vs_c_oracle():
    """BASELINE config 4 at full size against the oracle: ONE SCP iteration at 4096 x 50 (419 M collision rows; the joint QP on
    the lean 16-agent persistent kernel, 256 workgroups = every compute unit) from QP#0's solution, against the C oracle's
    step from the same x0 (~15 s on one host core): the same working set, the same number of constraint-generation rounds
    and ADMM iterations, accelerations to 1e-9 (bench.py --agents 4096 reports the measured figure)."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    N, K = 4096, 50
    p0, pf, space = generate_grid_swap(N, seed=1000 * N)
    s = SCP(N, K * 0.2 + 1e-9, 0.2, 0.8, space, verbose=False)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc0 = s._solve_initial_trajectory()
    new, info = s.scp_iteration(acc0)
    assert info["status_val"] == 1 and "persistent16" in info["pipeline"] and info["persist_gave_up"] == 0
    prob = so.make_problem(N, K * 0.2 + 1e-9, 0.2, 0.8, space, p0, pf)
    x0 = acc0.cpu().numpy()
    pos, _ = co.kinematics(prob, x0)
    eta, l_col, dist = co.linearize_pairs(prob, pos)
    x1, io = co.admm(prob, eta, l_col, dist, x0, qo.Settings(max_iter=10000, margin=s.working_set_margin))
    assert (info["iter"], info["working_rows"], info["rounds"]) == (io["iter"], io["working_rows"], io["rounds"])
    assert info["rho_updates"] == io["rho_updates"]
    np.testing.assert_allclose(new.cpu().numpy().reshape(x1.shape), x1, rtol=0, atol=1e-9)
    # the step on the three-launch pipeline (what every N > 2048 ran before the lean kernel): the same iterates to rounding
    s3 = SCP(N, K * 0.2 + 1e-9, 0.2, 0.8, space, verbose=False, qp_settings={"persistent": 0})
    s3.set_initial_states(p0)
    s3.set_final_states(pf)
    new3, info3 = s3.scp_iteration(acc0)
    assert info3["pipeline"] == "three-launch" and info3["iter"] == info["iter"]
    np.testing.assert_allclose(new.cpu().numpy(), new3.cpu().numpy(), rtol=0, atol=1e-9)


def test_carry_rho_follows_the_oracle():
    """opt-in carry_rho: the joint QP of SCP iteration n + 1 starts at the rho iteration n ended with.  The oracle mirrors it
    (c_oracle.scp_solve(carry_rho=True)): with two PCG steps the GPU follows it iterate for iterate (equal ADMM counts per
    QP, waypoints 1e-6), and the result stays within the solver tolerance of the restart-at-0.1 trajectories.  (Whether it
    SAVES steps depends on the problem -- tools/admm_steps_exp.py measures it; on this case it costs steps.)"""
    from path_planning.scenarios.position_generator import generate_grid_swap

    n = 96
    p0, pf, space = generate_grid_swap(n, seed=5)
    prob = so.make_problem(n, 10.0, 0.2, 0.8, space, p0, pf)
    st = qo.Settings(max_iter=10000, cg_iters=2)
    ref = co.scp_solve(prob, 15, st, carry_rho=True)
    base = co.scp_solve(prob, 15, st, carry_rho=False)
    s, traj = solve_gpu(n, 10.0, 0.2, 0.8, space, p0, pf, carry_rho=True, qp_settings={"cg_iters": 2})
    assert s.last_info["n_iterations"] == ref["iterations"]
    gi = [q["iter"] for q in s.last_info["iterations"]]
    assert gi == [q["iter"] for q in ref["infos"][1:]], (gi, [q["iter"] for q in ref["infos"][1:]])
    np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=1e-6)
    assert [q["rho"] for q in s.last_info["iterations"]] == [q["rho"] for q in ref["infos"][1:]]
    # two eps = 1e-3 solves of the same problem along different rho paths: OSQP's own accuracy (6e-2 m, see
    # test_against_reference_proxy)
    np.testing.assert_allclose(traj["positions"], base["positions"], rtol=0, atol=6e-2)


def test_3d_beyond_1024_agents_on_the_persistent_path():
    """3-D with more than 1024 agents: the 4-agent 3-D kernel runs out of compute units there and round 2 fell back to three
    launches per ADMM step (45 us); the lean kernel's 8-agent 3-D form takes over up to 2048 agents.  One SCP iteration at
    1100 x 50 x 3-D: the same ADMM count and accelerations (1e-9) as the three-launch pipeline from the same x0."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    n = 1100
    p0, pf, space = generate_grid_swap(n, seed=5, dim=3)
    out = {}
    for persistent in (1, 0):
        s = SCP(n, 10.0 + 1e-9, 0.2, 0.8, space, dim=3, verbose=False, qp_settings={"persistent": persistent})
        s.set_initial_states(p0)
        s.set_final_states(pf)
        s._precompute_constraint_matrices()
        acc0 = s._solve_initial_trajectory()
        new, info = s.scp_iteration(acc0)
        assert info["status_val"] == 1
        out[persistent] = (new.cpu().numpy(), info)
    assert "persistent8-lean" in out[1][1]["pipeline"].split("+") and out[0][1]["pipeline"] == "three-launch"
    assert out[1][1]["iter"] == out[0][1]["iter"] and out[1][1]["working_rows"] == out[0][1]["working_rows"]
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-9)


def test_lean_persistent_kernel_full_solves():
    """Complete solves with the lean 16-agent persistent kernel forced (persistent = 2) at sizes where the 8-agent kernel
    is the default: both follow the C oracle (SCP iteration count, ADMM counts to a check interval or two, waypoints to the
    solver tolerance -- the default single-step path, see test_scp_sweep_vs_oracle), and each other likewise."""
    from path_planning.scenarios.position_generator import generate_grid_swap

    for n, seed in ((40, 11), (150, 3)):
        p0, pf, space = generate_grid_swap(n, seed=seed)
        prob = so.make_problem(n, 10.0, 0.2, 0.8, space, p0, pf)
        ref = co.scp_solve(prob, 15, qo.Settings(max_iter=10000))
        got = {}
        for lean in (False, True, 3):
            s, traj = solve_gpu(n, 10.0, 0.2, 0.8, space, p0, pf, qp_settings={"persistent": {False: 4, True: 2, 3: 3}[lean]})
            assert s.last_info["converged"]
            want = {False: "persistent", True: "persistent16", 3: "persistent8-lean"}[lean]
            pipes = [q["pipeline"] for q in s.last_info["iterations"]]
            assert all(want in q.split("+") for q in pipes), pipes
            assert s.last_info["n_iterations"] == ref["iterations"]
            gi = [q["iter"] for q in s.last_info["iterations"]]
            ci = [q["iter"] for q in ref["infos"][1:]]
            assert all(abs(a - b) <= 50 for a, b in zip(gi, ci)), (gi, ci)
            np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=2e-2)
            got[lean] = (gi, traj["positions"])
        # lean against the 8-agent kernel directly: the association of sums differs, which the single-step path may amplify
        # (see TOL_DEFAULT); their state after 12 steps agrees to 1e-11 (test_persistent_kernel_equals_three_launch_pipeline)
        for other in (True, 3):
            assert all(abs(a - b) <= 50 for a, b in zip(got[False][0], got[other][0])), (got[False][0], got[other][0])
            np.testing.assert_allclose(got[other][1], got[False][1], rtol=0, atol=TOL_DEFAULT)


@pytest.mark.parametrize("N", [1024, 4096])
def test_full_size_properties(N):
    """BASELINE configs 3 and 4 at full size (1024 / 4096 agents x 50 steps: 26.2 M / 419 M collision rows, 0.6 / 10 GB
    of compact rows on one GPU): the oracle cannot finish a solve here in test time, so the check is through
    size-independent properties."""
    import torch

    from path_planning.scenarios.position_generator import generate_grid_swap

    K = 50
    p0, pf, space = generate_grid_swap(N, seed=1000 * N)
    s, traj = solve_gpu(N, K * 0.2 + 1e-9, 0.2, 0.8, space, p0, pf, max_iterations=15)
    assert s.K == K and s.last_info["converged"] and 1 <= s.last_info["n_iterations"] <= 15
    for it in s.last_info["iterations"]:
        assert it["status_val"] == 1
    check_solution_properties(s, traj)  # bitwise kinematics + every fixed row within the ADMM tolerance
    rep = s.validate_solution()
    assert rep["collision_free"] and rep["min_pair_distance"] >= 0.8 - 0.01
    # the compact rows of the LAST linearisation, spot-checked against the oracle formulas on 20 000 random rows
    prob = so.make_problem(N, K * 0.2 + 1e-9, 0.2, 0.8, space, p0, pf)
    ctx, pp = s._ctx, s._ensure_pairs()
    acc = ctx.tensor(traj["accelerations"])
    pos = ctx.tensor(traj["positions"])
    p0d, v0d = ctx.tensor(prob.p0), ctx.tensor(prob.v0)
    rows, min_dist, first = pp.linearize(pos, p0d, v0d, s.working_set_margin)
    assert first == (1 << 64) - 1 and abs(min_dist - rep["min_pair_distance"]) < 1e-12
    rng = np.random.default_rng(0)
    sample = np.unique(rng.integers(0, prob.m_col, 20000))  # (no permutation of 4.2e8 row ids at N = 4096)
    st = torch.as_tensor(sample, dtype=torch.int64, device=ctx.tdev)
    w_eta, w_l = pp.gather(st)
    iu, ju = so.pair_index(N)
    k, q = sample // prob.pairs, sample % prob.pairs
    P = traj["positions"]
    diff = P[iu[q], k] - P[ju[q], k]
    dist = np.hypot(diff[:, 0], diff[:, 1])
    eta = diff / dist[:, None]
    c = so.free_positions(prob)
    l_ref = prob.R + (np.sum(eta * diff, axis=1) - dist) - np.sum(eta * (c[iu[q], k] - c[ju[q], k]), axis=1)
    np.testing.assert_allclose(w_eta.cpu().numpy(), eta, rtol=0, atol=1e-13)
    np.testing.assert_allclose(w_l.cpu().numpy(), l_ref, rtol=0, atol=1e-12)
    # selection == {rows with dist - R < margin} on the sample; and the pass is idempotent
    sel = set(rows.cpu().numpy().tolist())
    want = sample[dist - prob.R < s.working_set_margin]
    assert all(int(r) in sel for r in want) and all(int(r) not in sel for r in sample[dist - prob.R >= s.working_set_margin])
    rows2, _, _ = pp.linearize(pos, p0d, v0d, s.working_set_margin)
    assert torch.equal(rows, rows2)
    # at the returned solution no linearised row (around that same solution) is violated beyond the tolerance
    new_rows, max_v = pp.violations(pos, p0d, v0d, 1e-6)
    assert max_v <= 0.8 - rep["min_pair_distance"] + 1e-9  # l - A x = R - dist at the linearisation point itself


def test_unresolved_rows_are_reported(capsys):
    """Constraint generation cut short (max_rounds = 1, a working set that starts empty): the QP over the working set
    reports "solved" although violated collision rows remain outside it -- the solver must say so (reference-style
    warning line) and record the count, instead of passing the point off as the joint QP's solution."""
    p0, pf = ref_scenario(10, 7)
    from path_planning.solvers.scp import SCP

    s = SCP(10, 10.0, 0.2, 0.8, [0, 0, 20, 20], verbose=True, max_rounds=1, working_set_margin=-1.0)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    capsys.readouterr()
    s.generate_trajectories(max_iterations=1)
    out = capsys.readouterr().out
    info = s.last_info["iterations"][0]
    assert info["unresolved_rows"] > 0 and info["max_violation"] > 1e-3 and info["rounds"] == 1
    assert "Warning: OSQP status constraint generation stopped" in out
    # the default settings resolve every row
    s2, _ = solve_gpu(10, 10.0, 0.2, 0.8, [0, 0, 20, 20], p0, pf, max_iterations=1)
    assert s2.last_info["iterations"][0]["unresolved_rows"] == 0


@pytest.mark.parametrize("kind,n,seed", [("ref", 6, 11), ("ref", 12, 13), ("grid", 25, 15), ("grid", 36, 16)])
def test_default_path_reaches_the_oracle_minimiser(kind, n, seed):
    """The default single-PCG-step path against the oracle at eps = 1e-8: each QP has ONE minimiser (P = 2I > 0), so
    with tight tolerances both implementations must land on it whatever their summation orders -- waypoints to 1e-6
    over whole SCP solves (the 2e-2 of test_scp_sweep_vs_oracle at eps = 1e-3 is the solver tolerance, not a property
    of the path)."""
    from path_planning.scenarios.position_generator import generate_grid_swap

    if kind == "ref":
        p0, pf = ref_scenario(n, seed)
        space = [0, 0, 20, 20]
    else:
        p0, pf, space = generate_grid_swap(n, seed=seed)
    tight = {"eps_abs": 1e-8, "eps_rel": 1e-8, "max_iter": 40000, "max_iter0": 40000}
    s, traj = solve_gpu(n, 10.0, 0.2, 0.8, space, p0, pf, max_iterations=3, qp_settings=tight)
    assert s._native.settings.cg_iters == 1 and s._native.settings.persistent == 1  # the defaults
    prob = so.make_problem(n, 10.0, 0.2, 0.8, space, p0, pf)
    from oracle import c_oracle as co

    ref = co.scp_solve(prob, 3, qo.Settings(max_iter=40000, eps_abs=1e-8, eps_rel=1e-8), max_iter0=40000)
    assert s.last_info["n_iterations"] == ref["iterations"]
    for a, b in zip(s.last_info["iterations"], ref["infos"][1:]):
        assert a["status_val"] == b["status_val"] == 1 and a["working_rows"] == b["working_rows"], (a, b)
    np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(traj["accelerations"], ref["accelerations"], rtol=0, atol=1e-5)


PROXY_TOL = 6e-2  # metres; OSQP's own accuracy at eps = 1e-3 ("distance to the reference")


@pytest.mark.parametrize("n,seed", [(4, 1), (6, 11), (10, 7)])
def test_against_reference_proxy(n, seed):
    """Distance to the reference as far as it can be stated without the `osqp` package: the reference's own loop on
    EXPLICIT matrices with all collision rows and OSQP's published algorithm at its defaults (Ruiz scaling, eps = 1e-3:
    qp_oracle.scp_solve_explicit) against the GPU result at its defaults.  Both stop at eps = 1e-3 solutions of the same
    QPs, so the waypoints agree to OSQP's accuracy, PROXY_TOL; the SCP iteration count may differ by one."""
    p0, pf = ref_scenario(n, seed)
    s, traj = solve_gpu(n, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf)
    prob = so.make_problem(n, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf)
    ref = qo.scp_solve_explicit(prob, 15)
    assert ref["converged"] and s.last_info["converged"]
    assert abs(s.last_info["n_iterations"] - ref["iterations"]) <= 1
    np.testing.assert_allclose(traj["positions"], ref["positions"], rtol=0, atol=PROXY_TOL)
    assert so.min_pair_distance(prob, traj["positions"]) >= 0.8 - 0.01  # scp.py:610


@pytest.mark.parametrize("kind,n,seed", [("ref", 10, 7), ("ref", 17, 331), ("grid", 64, 64000), ("grid3d", 27, 17)])
def test_polish_meets_every_constraint(kind, n, seed):
    """polish=True: one more joint QP at 1e-8 after the loop.  The returned point then satisfies every constraint of the
    reference QP to 1e-6 (fixed rows, final state) and keeps all pairs at >= R - 1e-6 -- hence passes the reference's own
    R - 0.01 check (scp.py:610), which a result at OSQP's 1e-3 can miss by millimetres."""
    from path_planning.scenarios.position_generator import generate_grid_swap

    dim = 3 if kind == "grid3d" else 2
    if kind == "ref":
        p0, pf = ref_scenario(n, seed)
        space = [0, 0, 20, 20]
    else:
        p0, pf, space = generate_grid_swap(n, seed=seed, dim=dim)
    s, traj = solve_gpu(n, 10.0, 0.2, 0.8, space, p0, pf, dim=dim, polish=True)
    assert s.last_info["polish"]["status_val"] in (1, 2) and s.last_info["polish"]["unresolved_rows"] == 0
    check_solution_properties(s, traj, tol=1e-6)
    rep = s.validate_solution()
    assert rep["collision_free"] and rep["min_pair_distance"] >= 0.8 - 1e-6, rep
    for key in ("acc_violation", "jerk_violation", "vel_violation", "pos_violation", "final_position_error",
                "final_velocity_error"):
        assert rep[key] < 1e-6, (key, rep[key])
    # without it the same solve stops at OSQP's tolerance
    s0, t0 = solve_gpu(n, 10.0, 0.2, 0.8, space, p0, pf, dim=dim)
    assert "polish" not in s0.last_info and np.abs(t0["positions"] - traj["positions"]).max() < 5e-2
