"""The natively driven SCP loop (scp_solver_solve, one C call per solve: the default) against the Python-driven loop
(native=False): the same library calls in the same order, so trajectories are bit-identical and the per-QP records
agree field by field."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def solve(native, n, T, h, space, p0, pf, dim=2, max_iterations=15, **kw):
    from path_planning.solvers.scp import SCP

    s = SCP(n, T, h, 0.8, space, dim=dim, verbose=False, native=native, **kw)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    return s, s.generate_trajectories(max_iterations)


def same_records(a, b):
    keys = ("status_val", "iter", "rho_updates", "cg_iters_total", "working_rows", "rounds", "added", "unresolved_rows",
            "status")
    for k in keys:
        assert a[k] == b[k], (k, a[k], b[k])
    for k in ("r_prim", "r_dual", "rho", "max_violation"):
        assert a[k] == b[k] or (np.isnan(a[k]) and np.isnan(b[k])), (k, a[k], b[k])


@pytest.mark.parametrize("case", ["ref10", "grid64", "grid3d27", "polish", "refresh", "tiny_capacity", "single_agent",
                                  "carry_rho"])
def test_native_loop_is_bit_identical(case):
    from path_planning.scenarios.position_generator import generate_grid_swap, generate_positions

    kw, dim, T, h = {}, 2, 10.0, 0.2
    if case in ("ref10", "polish", "refresh", "tiny_capacity"):
        p0, pf = generate_positions(10, 0.8, seed=7)
        n, space = 10, [0, 0, 20, 20]
        kw = {"polish": {"polish": True}, "refresh": {"refresh_feasibility": True}, "tiny_capacity": {"qp_row_capacity": 4}}.get(case, {})
    elif case in ("grid64", "carry_rho"):
        n = 64
        p0, pf, space = generate_grid_swap(n, seed=64000)
        kw = {"carry_rho": True} if case == "carry_rho" else {}
    elif case == "grid3d27":
        n, dim = 27, 3
        p0, pf, space = generate_grid_swap(n, seed=17, dim=3)
    else:
        n, T, h = 1, 2.0, 0.2
        p0, pf, space = np.array([[3.0, 3.0]]), np.array([[3.05, 2.95]]), [0, 0, 20, 20]
    a, ta = solve(True, n, T, h, space, p0, pf, dim=dim, **kw)  # native loop, row-free linearisation (the default)
    b, tb = solve(False, n, T, h, space, p0, pf, dim=dim, **kw)  # Python-driven loop: every row written, working rows gathered
    c, tc = solve(True, n, T, h, space, p0, pf, dim=dim, row_free=False, **kw)  # native loop on the row-writing kernel
    for key in ("positions", "velocities", "accelerations"):
        np.testing.assert_array_equal(ta[key], tb[key])
        np.testing.assert_array_equal(ta[key], tc[key])
    for ra, rc in zip(a.last_info["iterations"], c.last_info["iterations"]):
        same_records(ra, rc)
        assert ra["rel_step"] == rc["rel_step"] and ra["pipeline"] == rc["pipeline"]
    ia, ib = a.last_info, b.last_info
    assert (ia["n_iterations"], ia["converged"], ia["initially_feasible"]) == (ib["n_iterations"], ib["converged"], ib["initially_feasible"])
    assert ia["qp0"]["iter"] == ib["qp0"]["iter"] and ia["qp0"]["status_val"] == ib["qp0"]["status_val"]
    assert len(ia["iterations"]) == len(ib["iterations"])
    for ra, rb in zip(ia["iterations"], ib["iterations"]):
        same_records(ra, rb)
        assert ra["rel_step"] == rb["rel_step"] and ra["time_sec"] > 0
    assert ("polish" in ia) == ("polish" in ib) == (case == "polish")
    if case == "polish":
        same_records(ia["polish"], ib["polish"])
    if case == "tiny_capacity":
        assert ia["iterations"][0]["working_rows"] > 4  # the QP workspace had to grow inside the native loop too


@pytest.mark.parametrize("n,dim,seed", [(10, 2, 3), (64, 2, 64000), (128, 2, 128000), (27, 3, 17), (200, 2, 5)])
def test_context_switches_do_not_change_results(n, dim, seed):
    """scp_ctx_set_option: the one-launch pairwise passes of small problems (incl. the violations pass that derives x and its
    positions from the QP's time-major solution) against the multi-launch form, and kernel timing off: identical solves."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    p0, pf, space = generate_grid_swap(n, seed=seed, dim=dim)
    out = []
    for opts in ({}, {"single_launch_passes": 0}, {"kernel_timing": 0}):
        s = SCP(n, 10.0, 0.2, 0.8, space, dim=dim, verbose=False)
        for k, v in opts.items():
            s._ctx.set_option(k, v)
        s.set_initial_states(p0)
        s.set_final_states(pf)
        out.append((s.generate_trajectories(15), s.last_info))
    for t, info in out[1:]:
        for key in ("positions", "velocities", "accelerations"):
            np.testing.assert_array_equal(t[key], out[0][0][key])
        assert info["n_iterations"] == out[0][1]["n_iterations"] and info["converged"] == out[0][1]["converged"]
        for ra, rb in zip(info["iterations"], out[0][1]["iterations"]):
            same_records(ra, rb)
            assert ra["rel_step"] == rb["rel_step"]
    assert out[2][1]["iterations"][0]["violations_ms"] == 0.0 and out[2][1]["iterations"][0]["solve_ms"] > 0.0
    assert out[0][1]["iterations"][0]["violations_ms"] > 0.0


def test_native_stdout_and_errors(capsys):
    """The reference's printed lines (scp.py:153-163, :446-447, :611-613) and its RuntimeError come out of the native
    loop exactly as out of the Python-driven one."""
    from path_planning.solvers.scp import SCP

    outs = []
    for native in (True, False):
        s = SCP(n_vehicles=3, time_horizon=3.0, time_step=0.2, min_distance=0.5, space_dims=[-5, -5, 500, 200], native=native)
        s.set_initial_states(np.array([[-2.0, -2.0], [0.0, -2.0], [2.0, -2.0]]))
        s.set_final_states(np.array([[2.0, 2.0], [0.0, 2.0], [-2.0, 2.0]]))
        capsys.readouterr()
        s.generate_trajectories(max_iterations=2)
        out = capsys.readouterr().out.splitlines()
        outs.append([ln for ln in out if not ln.startswith("Trajectory generation completed")])
        assert any("Avoidance constraint violation at timestep" in ln for ln in out)
        assert "Warning: OSQP status primal infeasible" in out and "SCP Iteration 1" in out
        s2 = SCP(n_vehicles=2, time_horizon=1.0, time_step=0.2, min_distance=0.5, verbose=False, native=native)
        s2.set_initial_states(np.array([[1.0, 1.0], [3.0, 3.0]]))
        s2.set_final_states(np.array([[19.0, 19.0], [15.0, 3.0]]))
        with pytest.raises(RuntimeError, match="OSQP failed: primal infeasible"):
            s2.generate_trajectories()
    assert outs[0] == outs[1]


def test_scp_iteration_is_the_python_loop_body():
    """scp_solver_step (what bench.py times) = one pass of the Python-driven loop body: bitwise the same accelerations,
    the same record, and the device times of its pairwise kernels reported."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    n = 64
    p0, pf, space = generate_grid_swap(n, seed=64000)
    s = SCP(n, 10.0, 0.2, 0.8, space, verbose=False)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc0 = s._solve_initial_trajectory()
    ref = s._solve_with_avoidance_constraints(acc0)
    ref_info = dict(s._last_qp_info)
    ref_rel = s._ctx.rel_step(ref, acc0)[2]
    for rep in range(3):  # repeatable from the same input (the solver object carries no state from call to call);
        s.row_free = rep != 2  # row-free linearisation (default) and the row-writing kernel give the same bits
        new, info = s.scp_iteration(acc0)
        np.testing.assert_array_equal(new.cpu().numpy(), ref.cpu().numpy())
        same_records(info, ref_info)
        assert info["rel_step"] == ref_rel
        assert 0.0 < info["linearize_ms"] < 5.0 and 0.0 < info["violations_ms"] < 5.0 and info["time_sec"] > 0


@pytest.mark.parametrize("n,dim,seed", [(64, 2, 64000), (27, 3, 17), (1, 2, 0)])
def test_sharded_step_with_one_rank_is_scp_solver_step(n, dim, seed):
    """scp_solver_shard_begin / _qp / _violations / _round_done / _end are the phases scp_solver_step runs back to back: driven
    one by one over the full pair range (a world of one rank, no exchange) they give the same accelerations and the same
    record, bit for bit; a second call shows that no state leaks from step to step."""
    from path_planning.scenarios.position_generator import generate_grid_swap
    from path_planning.solvers.scp import SCP

    if n == 1:
        p0, pf, space, T = np.array([[3.0, 3.0]]), np.array([[3.05, 2.95]]), [0, 0, 20, 20], 2.0
    else:
        p0, pf, space = generate_grid_swap(n, seed=seed, dim=dim)
        T = 10.0
    s = SCP(n, T, 0.2, 0.8, space, dim=dim, verbose=False)
    s.set_initial_states(p0)
    s.set_final_states(pf)
    s._precompute_constraint_matrices()
    acc0 = s._solve_initial_trajectory()
    ref, ref_info = s.scp_iteration(acc0)
    for _ in range(2):
        new, info = s.scp_iteration_sharded(acc0)
        np.testing.assert_array_equal(new.cpu().numpy(), ref.cpu().numpy())
        same_records(info, ref_info)
        assert info["rel_step"] == ref_info["rel_step"] and info["pipeline"] == ref_info["pipeline"]
    s.row_free = False
    with pytest.raises(ValueError):
        s.scp_iteration_sharded(acc0)
