"""SCP trajectory planner -- MI355X-native drop-in for the reference's ``path_planning.solvers.scp.SCP``.

Same constructor, attributes, methods, stdout lines and error behaviour as
/root/reference/src/path_planning/solvers/scp.py (class SCP, :31-616); every method below cites the lines it
replaces.  The numerical work runs in hand-written HIP kernels behind the C-ABI of include/scp_hip.h
(libscp_hip.so, bound with ctypes in path_planning/_hip.py); PyTorch-ROCm tensors are device buffers only.
There is no CPU fallback: constructing the solver without a GPU or without the built library raises.

Differences to the reference that a caller can observe (INTEGRATION.md has the full list):
  * the QP solver is this repo's ADMM (OSQP's algorithm, matrix-free) instead of the `osqp` package, so
    iterates agree with the reference only up to OSQP's own tolerance (eps_abs = eps_rel = 1e-3);
  * ``_add_collision_constraints`` returns the compact form (eta, l) of the constraint rows instead of a
    2.6e9-non-zero CSC matrix; ``C_jerk/C_acc/C_vel/C_pos`` are not materialised (they stay None);
  * new keyword-only arguments: ``dim`` (2 or 3), ``device``, ``qp_settings``, ``working_set_margin``,
    ``feasibility_tol``, ``max_rounds``, ``refresh_feasibility``, ``polish``/``polish_eps``, ``native``, ``row_free``, ``carry_rho``, ``kernel_timing``,
    ``qp_row_capacity``, ``reuse_native``, ``verbose``, ``rank``/``world_size``/``group`` (pair-range-sharded multi-GPU: the SCP iteration natively,
    split at its exchange points -- ``scp_iteration_sharded``).
"""
from __future__ import annotations

import atexit
import threading
import time

import numpy as np

from .. import _hip
from .._sharding import Shard

# The reference's callers build a NEW solver object per scenario (compute_trajectories_batch.py:103-117).  Here an object owns a
# library context and a native solver (device workspace, pinned host words, the K x K blocks cached per rho): ~0.7 ms to
# create, against a 2.5 ms solve at 128 agents.  So the native halves of released SCP objects on the DEFAULT stream are kept,
# per (device, shape, settings), and the next object of the same shape adopts them -- same results bit for bit (a pooled solver is what
# compute-trajectories-batch has always reused per worker; tests/test_scp_gpu.py::test_released_solvers_are_reused).
_POOL = {}
_POOL_LOCK = threading.Lock()
_POOL_LIMIT = 4  # per key
_POOL_STATS = {"hits": 0, "misses": 0, "returned": 0}


def native_pool_stats():
    """{"hits", "misses", "returned", "held"}: how often a new SCP object adopted the native solver of a released one."""
    with _POOL_LOCK:
        return dict(_POOL_STATS, held=sum(len(v) for v in _POOL.values()))


def clear_native_pool():
    """Destroy every pooled native solver and context (also registered with atexit: before the HIP runtime goes away)."""
    with _POOL_LOCK:
        entries = [e for v in _POOL.values() for e in v]
        _POOL.clear()
    for ctx, nat in entries:
        try:
            nat.close()
            ctx.close()
        except Exception:
            pass


atexit.register(clear_native_pool)


class SCP:
    def __init__(
        self,
        n_vehicles=5,
        time_horizon=3.0,
        time_step=0.1,
        min_distance=0.1,
        space_dims=None,
        *,
        dim=2,
        device=None,
        qp_settings=None,
        working_set_margin=0.5,
        feasibility_tol=1e-6,
        max_rounds=20,
        refresh_feasibility=False,
        polish=False,
        polish_eps=1e-8,
        native=True,
        row_free=True,
        carry_rho=False,
        kernel_timing=True,
        qp_row_capacity=None,
        verbose=True,
        rank=0,
        world_size=1,
        group=None,
        reuse_native=True,
    ):
        # --- reference attributes (scp.py:40-91) ---
        self.N = n_vehicles
        self.T = time_horizon
        self.h = time_step
        self.K = int(self.T / self.h)  # scp.py:43
        self.R = min_distance
        self.D = int(dim)
        if self.D not in (2, 3):
            raise ValueError("dim must be 2 or 3")
        if space_dims is None:
            space_dims = [0, 0, 20, 20] if self.D == 2 else [0, 0, 0, 20, 20, 20]
        self.space_dims = space_dims
        if len(space_dims) != 2 * self.D:
            raise ValueError(f"space_dims needs {2 * self.D} entries [min..., max...]")
        self.convergence_tolerance = 1.5e-2
        self.trajectories = None
        self.initial_positions = None
        self.initial_velocities = None
        self.final_positions = None
        self.final_velocities = None
        self.pos_min = np.array(space_dims[: self.D])
        self.pos_max = np.array(space_dims[self.D:])
        self.vel_min = -2
        self.vel_max = 2
        self.acc_min = -15.0
        self.acc_max = 15.0
        self.jerk_min = -20
        self.jerk_max = 20
        self.C_jerk = self.C_acc = self.C_vel = self.C_pos = None  # never materialised on this path
        self.l_jerk, self.u_jerk = [], []
        self.l_acc, self.u_acc = [], []
        self.l_vel, self.u_vel = [], []
        self.l_pos, self.u_pos = [], []

        # --- MI355X path ---
        self.verbose = verbose
        self.working_set_margin = float(working_set_margin)
        self.feasibility_tol = float(feasibility_tol)
        self.max_rounds = int(max_rounds)
        self.refresh_feasibility = bool(refresh_feasibility)
        self.polish = bool(polish)
        self.polish_eps = float(polish_eps)
        self.native = bool(native)  # drive the SCP loop from C++ (scp_solver_solve) instead of from Python: same calls, same bits
        # native loop: linearise WITHOUT writing the 24-byte rows of all N(N-1)/2 K pairs (scp_select_pairs) and recompute eta / l
        # for the selected rows only (scp_qp_add_rows_at); False: scp_linearize_pairs writes every row (what
        # _add_collision_constraints returns, and what the Python-driven loop does).  Same working rows, same bits.
        self.row_free = bool(row_free)
        # opt-in: the joint QP of SCP iteration n + 1 starts at the rho iteration n ended with instead of OSQP's 0.1 (the
        # reference builds a new osqp.OSQP() per iteration, scp.py:441): saves the adaptive-rho transient of every later QP
        self.carry_rho = bool(carry_rho)
        self._rho_start = 0.0
        self._native = None
        self._qp_row_capacity = qp_row_capacity  # initial working-set capacity (grows on demand)
        self._qp_overrides = dict(qp_settings or {})
        self.shard = Shard(self.N, rank, world_size, group)
        if device is None:
            import torch

            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        # the native half of a released object of the same shape on this stream, if there is one (module docstring of _POOL)
        self.reuse_native = bool(reuse_native)
        self._pool_key = None
        if self.reuse_native:
            import torch

            known = {k for k, _ in _hip.QpSettings._fields_}
            st = _hip.default_settings(**{k: v for k, v in self._qp_overrides.items() if k in known})
            tdev = torch.device("cuda", int(device))
            on_default = torch.cuda.is_available() and (torch.cuda.current_stream(tdev).cuda_stream
                                                        == torch.cuda.default_stream(tdev).cuda_stream)
            if on_default:  # (a context is bound to its stream: only the default stream is sure to outlive the pool)
                self._pool_key = (int(device), int(self.N), int(self.K), self.D, float(self.h), float(self.R), bytes(st),
                                  self._qp_row_capacity)
                with _POOL_LOCK:
                    held = _POOL.get(self._pool_key)
                    entry = held.pop() if held else None
                    _POOL_STATS["hits" if entry else "misses"] += 1
                if entry:
                    self._ctx, self._native = entry
        if self._native is None:
            self._ctx = _hip.Context(device)  # raises without a GPU / without libscp_hip.so
        # False: no HIP events around the kernels (linearize_ms / violations_ms read 0, solve_ms is host wall clock): fewer queue
        # packets per solve, for many concurrent solves on one GPU
        self.kernel_timing = bool(kernel_timing)
        self._ctx.set_timing(self.kernel_timing)
        self._qp = None
        self._pairs = None
        self._dev = {}
        self.last_info = {}

        self._print("---=== SCP Problem initialized ===---")
        self._print(f"Number of timesteps: {self.K}")
        self._print(f"Timestep: {self.h}")
        self._print(f"Minimum distance between vehicles: {self.R}")
        self._print(f"Space dimensions: {self.space_dims}")

    def _print(self, *a):
        if self.verbose and self.shard.rank == 0:
            print(*a)

    def set_space_dims(self, space_dims):
        """Change the workspace box of an existing solver (same N, T, h, R, dim).  Together with set_initial_states /
        set_final_states this lets one solver object -- its context, streams, pinned buffers and the QP workspace, whose
        creation costs more than a 100-agent solve -- be reused for the next scenario (compute-trajectories-batch)."""
        if len(space_dims) != 2 * self.D:
            raise ValueError(f"space_dims needs {2 * self.D} entries [min..., max...]")
        self.space_dims = space_dims
        self.pos_min = np.array(space_dims[: self.D])
        self.pos_max = np.array(space_dims[self.D:])
        self.trajectories = None

    # ------------------------------------------------------------------------------------------------
    # a0: states (scp.py:99-129)
    # ------------------------------------------------------------------------------------------------
    def set_initial_states(self, positions, velocities=None):
        """Set initial states for all vehicles in flat format (scp.py:99-113)."""
        if velocities is None:
            velocities = np.zeros((self.N, self.D))
        self.initial_positions = np.asarray(positions, dtype=float).flatten()
        self.initial_velocities = np.asarray(velocities, dtype=float).flatten()
        assert len(self.initial_positions) == len(self.initial_velocities) == self.D * self.N, (
            f"Initial states mismatch"
            f"positions={len(self.initial_positions)}, "
            f"velocities={len(self.initial_velocities)}, "
            f"expected={self.D * self.N}"
        )
        self._dev.pop("p0", None)

    def set_final_states(self, positions, velocities=None):
        """Set final states for all vehicles in flat format (scp.py:115-129)."""
        if velocities is None:
            velocities = np.zeros((self.N, self.D))
        self.final_positions = np.asarray(positions, dtype=float).flatten()
        self.final_velocities = np.asarray(velocities, dtype=float).flatten()
        assert len(self.final_positions) == len(self.final_velocities) == self.D * self.N, (
            f"Final states mismatch"
            f"positions={len(self.final_positions)}, "
            f"velocities={len(self.final_velocities)}, "
            f"expected={self.D * self.N}"
        )
        self._dev.pop("p0", None)

    # ------------------------------------------------------------------------------------------------
    # device-side setup
    # ------------------------------------------------------------------------------------------------
    def _limits(self):
        return [self.vel_min, self.vel_max, self.acc_min, self.acc_max, self.jerk_min, self.jerk_max]

    def _space(self):
        return np.concatenate([np.asarray(self.pos_min, float), np.asarray(self.pos_max, float)])

    def _states(self):
        if "p0" not in self._dev:
            # one upload for the four arrays (each is a contiguous (N, D) view of it)
            packed = self._ctx.tensor(np.stack([self.initial_positions, self.initial_velocities, self.final_positions,
                                                self.final_velocities]).reshape(4, self.N, self.D))
            self._dev["p0"], self._dev["v0"], self._dev["pf"], self._dev["vf"] = packed[0], packed[1], packed[2], packed[3]
        d = self._dev
        return d["p0"], d["v0"], d["pf"], d["vf"]

    def _ensure_qp(self):
        if self._qp is None:
            known = {k for k, _ in _hip.QpSettings._fields_}
            st = _hip.default_settings(**{k: v for k, v in self._qp_overrides.items() if k in known})
            self._qp = _hip.QP(self._ctx, self.N, self.K, self.D, self.h, st, row_capacity=self._qp_row_capacity)
        return self._qp

    def _ensure_pairs(self):
        if self._pairs is None:
            q0, q1 = self.shard.pair_range()
            self._pairs = _hip.PairPass(self._ctx, self.N, self.K, self.D, self.R, self.h, q0, q1)
        return self._pairs

    def _grow_qp(self, need, keep_state=False):
        """Working set outgrew the QP's row capacity: move the solver to a workspace with room for `need` rows
        (keep_state: iterate, duals, working set and rho travel along, so the solve continues unchanged)."""
        old = self._qp
        cap = int(min(self.K * self.shard.pairs, max(need, 2 * old.row_capacity)))
        new = _hip.QP(self._ctx, self.N, self.K, self.D, self.h, old.settings, row_capacity=cap)
        if keep_state:
            new.take_state_of(old)
        else:
            p0, v0, pf, vf = self._states()
            new.set_problem(self._limits(), self._space(), p0, v0, pf, vf)
        old.close()
        self._qp = new
        return new

    # ------------------------------------------------------------------------------------------------
    # a1: SCP loop (scp.py:131-180)
    # ------------------------------------------------------------------------------------------------
    def generate_trajectories(self, max_iterations=15):
        """Main method to generate collision-free trajectories using SCP (scp.py:131-180)."""
        if self.native and self.shard.world == 1:
            return self._generate_trajectories_native(max_iterations)
        is_feasible = False
        start_time = time.time()

        self._precompute_constraint_matrices()
        acc = self._solve_initial_trajectory()
        init_guess_positions, _ = self._kinematics(acc, want_vel=False)
        is_feasible = self._fast_check_avoidance_constraints(init_guess_positions)

        iteration = 0
        converged = False
        self._rho_start = 0.0
        self.last_info = {"iterations": [], "qp0": dict(self._last_qp_info)}
        # `is_feasible` is evaluated once and never refreshed inside the loop (scp.py:144, :152)
        while iteration < max_iterations and not converged and not is_feasible:
            self._print(f"SCP Iteration {iteration+1}")
            t_it = time.perf_counter()
            if self.native and self.shard.world > 1 and self.row_free:
                # multi-rank: the iteration natively, split only at its exchange points (same bits as one rank)
                new_acc, it_info = self.scp_iteration_sharded(acc)
                rel_step_norm = it_info["rel_step"]
                if it_info["status_val"] not in (1, 2):  # scp.py:446-447
                    self._print(f"Warning: OSQP status {it_info['status']}")
                elif it_info["unresolved_rows"]:
                    self._print(f"Warning: OSQP status constraint generation stopped with {it_info['unresolved_rows']} "
                                f"violated collision rows outside the working set (max violation "
                                f"{it_info['max_violation']:.3e})")
            else:
                new_acc = self._solve_with_avoidance_constraints(acc)
                _, _, rel_step_norm = self._ctx.rel_step(new_acc, acc)  # scp.py:157-159 (no zero guard)
            self._print(rel_step_norm)
            self.last_info["iterations"].append(dict(self._last_qp_info, rel_step=rel_step_norm,
                                                     time_sec=time.perf_counter() - t_it))
            if self.carry_rho:
                self._rho_start = float(self._last_qp_info["rho"])
            if rel_step_norm <= self.convergence_tolerance:
                converged = True
                self._print(f"Converged after {iteration+1} iterations.")
            acc = new_acc
            iteration += 1
            if self.refresh_feasibility and not converged:
                # opt-in (the reference leaves this as a TODO, scp.py:150): stop as soon as the new trajectories are
                # collision free instead of waiting for the relative-step test
                pos_now, _ = self._kinematics(acc, want_vel=False)
                verbose, self.verbose = self.verbose, False
                is_feasible = self._fast_check_avoidance_constraints(pos_now)
                self.verbose = verbose

        self._rho_start = 0.0
        if self.polish:
            # opt-in (not in the reference): one more joint QP, linearised at the final trajectories and solved to
            # polish_eps instead of OSQP's 1e-3.  Every linearised row then holds to ~polish_eps, and a satisfied row
            # eta.(p_i - p_j) >= R implies ||p_i - p_j|| >= R, so the result passes the reference's own R - 0.01 check
            # (scp.py:610) and meets the fixed rows to the same accuracy; the reference's loop stops at OSQP's tolerance,
            # which leaves millimetres of violation in a converged result.
            t_p = time.perf_counter()
            acc = self._solve_with_avoidance_constraints(acc, eps=self.polish_eps)
            self.last_info["polish"] = dict(self._last_qp_info, time_sec=time.perf_counter() - t_p)

        positions, velocities = self._kinematics(acc)
        self.trajectories = {
            "positions": positions.cpu().numpy(),  # Shape (N, K, D)
            "velocities": velocities.cpu().numpy(),
            "accelerations": acc.cpu().numpy(),
        }
        self.last_info.update(converged=converged, initially_feasible=bool(is_feasible), n_iterations=iteration)
        end_time = time.time()
        self._print(f"Trajectory generation completed in {end_time - start_time:.3f} seconds")
        return self.trajectories

    def close(self):
        """Release the device objects.  The native solver and its context go to the process-wide pool (reuse_native) for the
        next SCP object of the same shape; everything else is destroyed.  Called by __del__; the object is unusable after."""
        ctx, nat = getattr(self, "_ctx", None), getattr(self, "_native", None)
        self._native = None
        for name in ("_qp", "_pairs"):
            obj = getattr(self, name, None)
            if obj is not None and hasattr(obj, "close"):
                try:
                    obj.close()
                except Exception:
                    pass
            setattr(self, name, None)
        self._dev = {}
        self._ctx = None
        if ctx is None:
            return
        keep = (nat is not None and self._pool_key is not None and getattr(ctx, "h", None)
                and not getattr(ctx, "options_changed", False))
        if keep:
            with _POOL_LOCK:
                held = _POOL.setdefault(self._pool_key, [])
                if len(held) < _POOL_LIMIT:
                    held.append((ctx, nat))
                    _POOL_STATS["returned"] += 1
                    return
        if nat is not None:
            nat.close()
        ctx.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ensure_native(self):
        if self._native is None:
            known = {k for k, _ in _hip.QpSettings._fields_}
            st = _hip.default_settings(**{k: v for k, v in self._qp_overrides.items() if k in known})
            self._native = _hip.NativeSolver(self._ctx, self.N, self.K, self.D, self.h, self.R, st,
                                             row_capacity=self._qp_row_capacity)
        return self._native

    def _native_options(self, max_iterations=15):
        return self._ensure_native().default_options(
            max_iterations=int(max_iterations), max_rounds=self.max_rounds,
            max_iter0=int(self._qp_overrides.get("max_iter0", self._qp_overrides.get("max_iter", 4000))),
            max_iter=int(self._qp_overrides.get("max_iter", 10000)), refresh_feasibility=int(self.refresh_feasibility),
            polish=int(self.polish), working_set_margin=self.working_set_margin, feasibility_tol=self.feasibility_tol,
            polish_eps=self.polish_eps, convergence_tolerance=self.convergence_tolerance, row_free=int(self.row_free),
            carry_rho=int(self.carry_rho))

    def scp_iteration(self, accelerations):
        """ONE pass of the SCP loop body (scp.py:152-166) in one library call (scp_solver_step): linearise around
        `accelerations` (device (N, K, D)), joint QP, relative step.  Returns (new accelerations, info dict).  This is what
        bench.py times; generate_trajectories runs the same code in its native loop."""
        if self.shard.world != 1:
            return self.scp_iteration_sharded(accelerations)
        nat = self._ensure_native()
        p0, v0, pf, vf = self._states()
        new, rec = nat.step(self._limits(), self._space(), p0, v0, pf, vf, self._native_options(),
                            self._to_device_acc(accelerations))
        info = dict(rec.as_dict(), rel_step=float(rec.rel_step), time_sec=float(rec.time_sec))
        self._last_qp_info = info
        return new, info

    def scp_iteration_sharded(self, accelerations, exchange_positions=True):
        """The same iteration with the O(N^2 K) passes sharded over the ranks (scp_solver_shard_*): every rank selects /
        checks the pairs of ITS range, the row ids are allgathered (sorted: the single-rank order), the joint QP is
        replicated -- deterministic, so every rank holds the same bits and nothing is broadcast.  With one rank this is
        scp_solver_step phase by phase (same result, bit for bit); the exchanges are the only difference otherwise.

        exchange_positions: the per-shard trajectories of the linearisation point are computed by the rank that owns the
        agents and allgathered (one collective per SCP iteration, the north-star layout); False: every rank integrates
        all agents itself (N K^2 flops, cheaper than the collective below ~10^4 agents)."""
        nat = self._ensure_native()
        p0, v0, pf, vf = self._states()
        acc = self._to_device_acc(accelerations)
        opts = self._native_options()
        if not opts.row_free:
            raise ValueError("the sharded step is row-free: a rank does not hold the rows of another rank's pairs")
        pos_in = None
        if exchange_positions and not self.shard.alone:
            pos_in, _ = self._kinematics(acc, want_vel=False)  # agent-sharded kinematics + allgather of the trajectories
        q0, q1 = self.shard.pair_range()
        rec, rows = nat.shard_begin(self._limits(), self._space(), p0, v0, pf, vf, opts, acc, pos_in, q0, q1)
        rows, _ = self.shard.allgather_ids(rows)
        more = opts.max_rounds >= 1
        if not more:
            raise ValueError("the sharded step needs max_rounds >= 1")
        while more:
            nat.shard_qp(rows, rec)
            rows, max_v = nat.shard_violations()
            rows, max_v = self.shard.allgather_ids(rows, extra=max_v)
            more = nat.shard_round_done(int(rows.numel()), max_v, rec)
        new = nat.shard_end(rec)
        info = dict(rec.as_dict(), rel_step=float(rec.rel_step), time_sec=float(rec.time_sec))
        self._last_qp_info = info
        return new, info

    def _generate_trajectories_native(self, max_iterations):
        """generate_trajectories with the loop driven by scp_solver_solve (one C call; the Python-driven loop above makes
        the same library calls in the same order and gives bit-identical results).  The reference's stdout lines are
        printed from the returned records, in the reference's order."""
        start_time = time.time()
        self._drop_bound_attributes()  # l_* / u_* are recomputed when somebody reads them (see __getattr__)
        nat = self._ensure_native()
        p0, v0, pf, vf = self._states()
        opts = self._native_options(max_iterations)
        acc, pos, vel, res, recs = nat.solve(self._limits(), self._space(), p0, v0, pf, vf, opts)
        qp0 = dict(recs[0].as_dict(), rounds=1, added=[])
        self._last_qp_info = qp0
        if res.qp0_status not in (1, 2):  # Solved / Solved Inaccurate (scp.py:363-365)
            self._print("not feasible")
            raise RuntimeError(f"OSQP failed: {qp0['status']}")
        if not res.initially_feasible:
            self._print(f"Avoidance constraint violation at timestep {res.first_violation_k} between vehicles "
                        f"{res.first_violation_i} and {res.first_violation_j}: distance = {res.first_violation_distance:.3f}")
        its = []
        for n, r in enumerate(recs[1:1 + res.n_iterations]):
            d = dict(r.as_dict(), rel_step=float(r.rel_step), time_sec=float(r.time_sec))
            self._print(f"SCP Iteration {n+1}")
            if d["status_val"] not in (1, 2):  # scp.py:446-447
                self._print(f"Warning: OSQP status {d['status']}")
            elif d["unresolved_rows"]:
                self._print(f"Warning: OSQP status constraint generation stopped with {d['unresolved_rows']} violated "
                            f"collision rows outside the working set (max violation {d['max_violation']:.3e})")
            self._print(d["rel_step"])
            if d["rel_step"] <= self.convergence_tolerance:
                self._print(f"Converged after {n+1} iterations.")
            its.append(d)
        self.last_info = {"iterations": its, "qp0": qp0, "converged": bool(res.converged),
                          "initially_feasible": bool(res.feasible_at_exit), "n_iterations": int(res.n_iterations)}
        if res.polished:
            r = recs[res.n_records - 1]
            self.last_info["polish"] = dict(r.as_dict(), time_sec=float(r.time_sec))
        if its:
            self._last_qp_info = its[-1]
        # acc, pos, vel are the three slices of one device buffer: one device-to-host copy for all of them
        host = acc._base.cpu().numpy() if acc._base is not None else None
        self.trajectories = {
            "positions": host[1] if host is not None else pos.cpu().numpy(),  # Shape (N, K, D)
            "velocities": host[2] if host is not None else vel.cpu().numpy(),
            "accelerations": host[0] if host is not None else acc.cpu().numpy(),
        }
        self._print(f"Trajectory generation completed in {time.time() - start_time:.3f} seconds")
        return self.trajectories

    # ------------------------------------------------------------------------------------------------
    # a2: fixed rows (scp.py:182-321) -- bounds only; the matrices are the shared K-column blocks
    # ------------------------------------------------------------------------------------------------
    def _precompute_constraint_matrices(self):
        p0, v0, pf, vf = self._fill_bound_attributes()
        self._ensure_qp().set_problem(self._limits(), self._space(), p0, v0, pf, vf)

    _BOUND_ATTRS = ("l_jerk", "u_jerk", "l_acc", "u_acc", "l_vel", "u_vel", "l_pos", "u_pos")

    def _drop_bound_attributes(self):
        for name in self._BOUND_ATTRS:
            self.__dict__.pop(name, None)

    def __getattr__(self, name):
        # The native loop does not need the reference's l_* / u_* arrays (scp.py:189-257) on the host; they are built on
        # first access (one small kernel + a device-to-host copy) instead of once per solve.
        if (name in SCP._BOUND_ATTRS and self.__dict__.get("initial_positions") is not None
                and self.__dict__.get("final_positions") is not None):
            self._fill_bound_attributes()
            return self.__dict__[name]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    def _fill_bound_attributes(self):
        """l_* / u_* attributes of the reference (scp.py:189-257), bitwise equal; returns the device states."""
        N, K, D = self.N, self.K, self.D
        p0, v0, pf, vf = self._states()
        lo, hi = self._ctx.fixed_bounds(N, K, D, self.h, self._limits(), self._space(), p0, v0, pf, vf)
        lo, hi = lo.cpu().numpy(), hi.cpu().numpy()
        nj, na = N * (K - 1) * D, N * K * D
        self.l_jerk, self.u_jerk = lo[:nj], hi[:nj]
        self.l_acc, self.u_acc = lo[nj:nj + na], hi[nj:nj + na]
        self.l_vel, self.u_vel = lo[nj + na:nj + 2 * na], hi[nj + na:nj + 2 * na]
        self.l_pos, self.u_pos = lo[nj + 2 * na:], hi[nj + 2 * na:]
        # size checks of the reference (scp.py:265-299)
        assert self.l_acc.shape == self.u_acc.shape == (D * N * K,)
        assert self.l_jerk.shape == self.u_jerk.shape == (D * N * (K - 1),)
        assert self.l_vel.shape == self.u_vel.shape == (D * N * K,)
        assert self.l_pos.shape == self.u_pos.shape == (D * N * K,)
        return p0, v0, pf, vf

    # ------------------------------------------------------------------------------------------------
    # a3: QP#0 (scp.py:323-369)
    # ------------------------------------------------------------------------------------------------
    def _solve_initial_trajectory(self):
        """Solve initial trajectory without avoidance constraints.  Returns a device tensor (N, K, D)."""
        qp = self._ensure_qp()
        qp.update_settings(max_iter=int(self._qp_overrides.get("max_iter0", self._qp_overrides.get("max_iter", 4000))))
        qp.reset(None)
        info = qp.solve()
        info["status_val"], info["iter"] = self.shard.broadcast_ints([info["status_val"], info["iter"]])
        info["status"] = _hip.STATUS_TEXT.get(info["status_val"], str(info["status_val"]))
        self._last_qp_info = dict(info, rounds=1, added=[])
        if info["status_val"] not in (1, 2):  # Solved / Solved Inaccurate (scp.py:363-365)
            self._print("not feasible")
            raise RuntimeError(f"OSQP failed: {info['status']}")
        x = qp.solution()
        self.shard.broadcast(x)
        return x

    # ------------------------------------------------------------------------------------------------
    # a4 / a7: kinematics (scp.py:371-397, :559-595)
    # ------------------------------------------------------------------------------------------------
    def _kinematics(self, acc, want_vel=True):
        """Agent-sharded kinematics + allgather of the per-shard trajectories (device tensors)."""
        p0, v0, _, _ = self._states()
        i0, i1 = self.shard.agent_range()
        n = i1 - i0
        if self.shard.world == 1:
            return self._ctx.kinematics(self.N, self.K, self.D, self.h, acc, p0, v0, want_vel)
        pos, vel = self._ctx.kinematics(n, self.K, self.D, self.h, acc[i0:i1].contiguous(), p0[i0:i1].contiguous(),
                                        v0[i0:i1].contiguous(), want_vel)
        pos = self.shard.allgather_positions(pos)
        vel = self.shard.allgather_positions(vel) if want_vel else None
        return pos, vel

    def _to_device_acc(self, accelerations):
        import torch

        if isinstance(accelerations, torch.Tensor):
            return accelerations.reshape(self.N, self.K, self.D).to(self._ctx.tdev, torch.float64).contiguous()
        return self._ctx.tensor(np.asarray(accelerations, dtype=float).reshape(self.N, self.K, self.D))

    def _compute_positions_velocities(self, accelerations):
        """positions, velocities (N, K, D) numpy arrays from accelerations (scp.py:371-397)."""
        pos, vel = self._kinematics(self._to_device_acc(accelerations))
        return pos.cpu().numpy(), vel.cpu().numpy()

    def _accelerations_to_positions_velocities(self, accelerations_flat):
        """Same recurrences on the flat vector (scp.py:559-595); bitwise equal to the method above."""
        return self._compute_positions_velocities(accelerations_flat)

    # ------------------------------------------------------------------------------------------------
    # a5: pairwise linearisation (scp.py:453-557)
    # ------------------------------------------------------------------------------------------------
    def _add_collision_constraints(self, previous_solution):
        """Compact form of A_collision: returns (eta (rows, D), l_collision (rows,), u_collision (rows,)) as numpy
        arrays for THIS rank's pair range, rows in the reference's k-major / i / j>i order.  Row r of the
        reference's matrix is eta_r[d] h^2 (k-m-.5) on a_i[m] and the negative on a_j[m], m < k."""
        acc = self._to_device_acc(previous_solution)
        pos, _ = self._kinematics(acc, want_vel=False)
        p0, v0, _, _ = self._states()
        pp = self._ensure_pairs()
        pp.linearize(pos, p0, v0, self.working_set_margin)
        l = pp.l_rows().cpu().numpy()
        return pp.eta_rows().cpu().numpy(), l, np.full(l.shape, np.inf)

    # ------------------------------------------------------------------------------------------------
    # a6: joint QP with collision rows (scp.py:399-451)
    # ------------------------------------------------------------------------------------------------
    def _solve_with_avoidance_constraints(self, accelerations_flat, eps=None):
        """Solve with collision avoidance constraints.  Takes / returns a device tensor (N, K, D).
        eps: termination tolerances of this one QP (the polish step), None = the solver's settings.

        The joint QP over ALL collision rows is solved by exact constraint generation: ADMM runs on the fixed rows
        and a working set (rows with dist - R < working_set_margin at the linearisation point); a full pairwise
        pass then checks every row at the solution and violated rows join the working set until none is left."""
        acc = self._to_device_acc(accelerations_flat)
        p0, v0, _, _ = self._states()
        pp = self._ensure_pairs()
        qp = self._ensure_qp()
        max_iter = int(self._qp_overrides.get("max_iter", 10000))  # scp.py:442
        saved = (qp.settings.eps_abs, qp.settings.eps_rel, qp.settings.max_iter)
        try:
            x, info, total, added, max_v = self._joint_qp_rounds(acc, p0, v0, pp, qp, max_iter, eps)
        finally:
            # the per-QP tolerances (polish) and the per-round iteration budget must not outlive this call, whatever
            # happens inside it (self._qp: the solver may have moved to a larger workspace meanwhile)
            self._qp.update_settings(eps_abs=saved[0], eps_rel=saved[1], max_iter=saved[2])
        self._last_qp_info = dict(info, **total, rounds=len(added), added=added, unresolved_rows=added[-1],
                                  max_violation=max_v)
        if info["status_val"] not in (1, 2):  # scp.py:446-447
            self._print(f"Warning: OSQP status {info['status']}")
        elif added[-1]:
            # constraint generation stopped (max_rounds or the iteration budget) with violated rows still outside the
            # working set: x solves the QP over the working set only, not the full joint QP of scp.py:399-451
            self._print(f"Warning: OSQP status constraint generation stopped with {added[-1]} violated collision rows "
                        f"outside the working set (max violation {max_v:.3e})")
        return x

    def _joint_qp_rounds(self, acc, p0, v0, pp, qp, max_iter, eps):
        """The rounds of _solve_with_avoidance_constraints: linearise, working set, ADMM, violations pass, repeat."""
        if eps is not None:
            qp.update_settings(eps_abs=float(eps), eps_rel=float(eps))
            max_iter = max(max_iter, 40000)  # three more digits take a few times OSQP's budget

        prev_pos, _ = self._kinematics(acc, want_vel=False)
        rows, _, _ = pp.linearize(prev_pos, p0, v0, self.working_set_margin)
        w_eta, w_l = pp.gather(rows)
        rows, w_eta, w_l = self.shard.allgather_rows(rows, w_eta, w_l)

        qp.reset(acc)
        if self._rho_start > 0.0:
            qp.set_rho(self._rho_start)
        try:
            qp.add_rows(rows, w_eta, w_l)
        except _hip.HipError as e:
            if e.code != _hip.SCP_ERR_CAPACITY:
                raise
            qp = self._grow_qp(int(rows.numel()))
            qp.reset(acc)
            if self._rho_start > 0.0:
                qp.set_rho(self._rho_start)
            qp.add_rows(rows, w_eta, w_l)

        used = 0
        added = []
        info = None
        x = acc
        max_v = 0.0
        total = {"iter": 0, "cg_iters_total": 0, "rho_updates": 0, "solve_ms": 0.0, "persist_launches": 0,
                 "persist_gave_up": 0, "rho_switches_in_kernel": 0}
        pipes = set()
        for rnd in range(self.max_rounds):
            qp.update_settings(max_iter=max(max_iter - used, 1))
            info = qp.solve()
            # rank 0's counters drive the loop on every rank (see Shard.broadcast_ints)
            info["status_val"], info["iter"] = self.shard.broadcast_ints([info["status_val"], info["iter"]])
            info["status"] = _hip.STATUS_TEXT.get(info["status_val"], str(info["status_val"]))
            used += info["iter"]
            for k in total:
                total[k] += info[k]
            pipes.update(info["pipeline"].split("+"))
            x = qp.solution()
            self.shard.broadcast(x)
            pos_new, _ = self._kinematics(x, want_vel=False)
            new_rows, max_v = pp.violations(pos_new, p0, v0, self.feasibility_tol)
            n_eta, n_l = pp.gather(new_rows)
            new_rows, n_eta, n_l = self.shard.allgather_rows(new_rows, n_eta, n_l)
            added.append(int(new_rows.numel()))
            if new_rows.numel() == 0 or used >= max_iter:
                break
            try:
                qp.add_rows(new_rows, n_eta, n_l)
            except _hip.HipError as e:
                if e.code != _hip.SCP_ERR_CAPACITY:
                    raise
                # continue in a larger workspace: the state of the solve travels along
                qp = self._grow_qp(qp.n_rows + int(new_rows.numel()), keep_state=True)
                qp.add_rows(new_rows, n_eta, n_l)
        total["pipeline"] = "+".join(n for n in _hip.PIPELINES if n in pipes) or "none"
        return x, info, total, added, max_v

    # ------------------------------------------------------------------------------------------------
    # a8: avoidance check (scp.py:597-615)
    # ------------------------------------------------------------------------------------------------
    def _fast_check_avoidance_constraints(self, positions):
        """True when every pair keeps ||p_i - p_j|| >= R - 0.01 at every stored sample; otherwise prints the first
        violation in k -> i -> j order exactly like the reference (scp.py:602-615)."""
        import torch

        pos = positions if isinstance(positions, torch.Tensor) else self._ctx.tensor(np.asarray(positions, dtype=float))
        pos = pos.reshape(self.N, self.K, self.D).contiguous()
        q0, q1 = self.shard.pair_range()
        _, first, _, _ = self._ctx.check_avoidance(self.N, self.K, self.D, self.R, pos, q0, q1)
        first = self.shard.all_min_int(first)
        if first >= (1 << 63) - 1:
            return True
        pairs = self.shard.pairs
        k, q = divmod(first, pairs)
        i, j = pair_from_index(q, self.N)
        d = float(torch.linalg.vector_norm(pos[i, k] - pos[j, k]).item())
        self._print(
            f"Avoidance constraint violation at timestep {k} between vehicles {i} and {j}: distance = {d:.3f}"
        )
        return False

    # ------------------------------------------------------------------------------------------------
    # post-solve validation: one device pass over all pairs + the fixed rows on the host copy
    # ------------------------------------------------------------------------------------------------
    def validate_solution(self):
        """Feasibility report of the stored trajectories: minimum pair distance over all stored samples and its
        first violation of R - 0.01 (device reduction, generalises scp.py:597-615), worst violation of every bound
        the reference imposes (scp.py:182-257) and of the final-state equalities (state K)."""
        if self.trajectories is None:
            raise ValueError("Trajectories not generated yet")
        N, K, D, h = self.N, self.K, self.D, self.h
        a, v, p = (self.trajectories[k] for k in ("accelerations", "velocities", "positions"))
        q0, q1 = self.shard.pair_range()
        min_dist, first, _, _ = self._ctx.check_avoidance(N, K, D, self.R, self._ctx.tensor(p), q0, q1)
        min_dist = self.shard.all_min(min_dist)
        first = self.shard.all_min_int(first)
        pf = self.final_positions.reshape(N, D)
        vf = self.final_velocities.reshape(N, D)
        vK = v[:, K - 1] + h * a[:, K - 1]
        pK = p[:, K - 1] + h * v[:, K - 1] + 0.5 * h * h * a[:, K - 1]
        over = lambda x, lo, hi: float(max(0.0, (lo - x).max(), (x - hi).max()))  # noqa: E731
        report = {
            "min_pair_distance": min_dist,
            "collision_free": first >= (1 << 63) - 1,
            "first_violation": None,
            "acc_violation": over(a, self.acc_min, self.acc_max),
            "jerk_violation": over(np.diff(a, axis=1) / h, self.jerk_min, self.jerk_max),
            "vel_violation": max(over(v[:, 1:], self.vel_min, self.vel_max), over(vK, self.vel_min, self.vel_max)),
            "pos_violation": over(p[:, 1:], np.asarray(self.pos_min, float), np.asarray(self.pos_max, float)),
            "final_position_error": float(np.abs(pK - pf).max()),
            "final_velocity_error": float(np.abs(vK - vf).max()),
        }
        if not report["collision_free"]:
            k, q = divmod(first, self.shard.pairs)
            i, j = pair_from_index(q, N)
            report["first_violation"] = {"timestep": int(k), "vehicles": (i, j),
                                         "distance": float(np.linalg.norm(p[i, k] - p[j, k]))}
        return report

    # ------------------------------------------------------------------------------------------------
    # visualisation passthroughs (scp.py:644-840): host-side matplotlib over the stored numpy trajectories
    # ------------------------------------------------------------------------------------------------
    def visualize_trajectories(self, show_animation=False, save_path="trajectories.pdf"):
        if self.trajectories is None:
            raise ValueError("Trajectories not generated yet")  # scp.py:646-647
        from ..viz.plot_trajectories import plot_trajectories

        return plot_trajectories(self, show=show_animation, save_path=save_path)

    def visualize_time_snapshots(self, num_snapshots=5, save_path=None):
        if self.trajectories is None:
            raise ValueError("Trajectories not generated yet")  # scp.py:782-783
        from ..viz.plot_trajectories import plot_time_snapshots

        return plot_time_snapshots(self, num_snapshots=num_snapshots, save_path=save_path)


def pair_from_index(q, N):
    """Lexicographic pair index -> (i, j), i < j (host helper, exact integer arithmetic)."""
    i = int(((2 * N - 1) - np.sqrt(float((2 * N - 1) ** 2 - 8 * q))) // 2)
    i = max(0, min(i, N - 2))
    off = lambda a: a * (2 * N - a - 1) // 2
    while off(i) > q:
        i -= 1
    while i < N - 2 and off(i + 1) <= q:
        i += 1
    return i, int(q - off(i) + i + 1)
