#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (run in the build container only).

    MPLBACKEND=Agg python tests/golden/make_golden.py

The reference (/root/reference/src/path_planning) is pure Python and importable here except for
its `import osqp` (scp.py:5): the `osqp` wheel is not installed in this image.  An EMPTY module is
registered under that name so that the import statement succeeds; nothing of OSQP is emulated and
the two reference methods that call osqp.OSQP() (_solve_initial_trajectory,
_solve_with_avoidance_constraints) are never invoked.  Everything stored below is produced by the
reference's own code paths a0, a2, a4, a5, a7, a8 and by its scenario
generator under random.seed(s).

Fixtures are DATA ONLY (inputs + expected outputs, .npz without pickles); no reference source text is
stored.  The reference never travels to the GPU box, these files do.
"""
import contextlib
import hashlib
import io
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def import_reference():
    sys.modules.setdefault("osqp", types.ModuleType("osqp"))  # empty placeholder, see docstring
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from path_planning.scenarios.position_generator import generate_positions
        from path_planning.solvers.scp import SCP
    return SCP, generate_positions


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def quiet(fn, *a, **kw):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **kw)
    return out, buf.getvalue()


def run_case(SCP, name, N, T, h, R, space, p0, pf, v0, vf, acc_seed, acc_scale, full_csc, n_probe=3):
    solver, _ = quiet(SCP, n_vehicles=N, time_horizon=T, time_step=h, min_distance=R, space_dims=space)
    solver.set_initial_states(p0, v0)
    solver.set_final_states(pf, vf)
    K = solver.K
    quiet(solver._precompute_constraint_matrices)
    rng = np.random.default_rng(acc_seed)
    acc = acc_scale * rng.standard_normal(2 * N * K)
    np.random.seed(12345)  # only consumed by the degenerate-pair branch (scp.py:505); fixtures avoid it
    (A_col, l_col, u_col), _ = quiet(solver._add_collision_constraints, acc)
    (pos7, vel7), _ = quiet(solver._accelerations_to_positions_velocities, acc)
    (pos4, vel4), _ = quiet(solver._compute_positions_velocities, acc.reshape(N, K, 2))
    feas, feas_out = quiet(solver._fast_check_avoidance_constraints, pos4)
    out = {
        "N": N, "T": T, "h": h, "R": R, "K": K, "space": np.asarray(space, float),
        "p0": p0, "pf": pf, "v0": v0 if v0 is not None else np.zeros((N, 2)),
        "vf": vf if vf is not None else np.zeros((N, 2)),
        "acc": acc,
        "pos_a4": pos4, "vel_a4": vel4, "pos_a7": pos7, "vel_a7": vel7,
        "l_col": l_col, "u_col_all_inf": bool(np.all(np.isposinf(u_col))),
        "feasible": bool(feas), "feasible_stdout": feas_out,
        "A_col_shape": np.asarray(A_col.shape), "A_col_nnz": A_col.nnz,
    }
    for nm in ("jerk", "acc", "vel", "pos"):
        C = getattr(solver, f"C_{nm}")
        out[f"l_{nm}"] = getattr(solver, f"l_{nm}")
        out[f"u_{nm}"] = getattr(solver, f"u_{nm}")
        out[f"C_{nm}_shape"] = np.asarray(C.shape)
        out[f"C_{nm}_nnz"] = C.nnz
        C.sort_indices()
        if full_csc:
            out[f"C_{nm}_data"] = C.data
            out[f"C_{nm}_indices"] = C.indices.astype(np.int32)
            out[f"C_{nm}_indptr"] = C.indptr.astype(np.int32)
        else:
            out[f"C_{nm}_sha"] = sha(C.data) + sha(C.indices.astype(np.int32)) + sha(C.indptr.astype(np.int32))
    A_col.sort_indices()
    if full_csc:
        out["A_col_data"] = A_col.data
        out["A_col_indices"] = A_col.indices.astype(np.int32)
        out["A_col_indptr"] = A_col.indptr.astype(np.int32)
    # probe products (cheap, size-independent pins of the whole matrix)
    prng = np.random.default_rng(acc_seed + 7)
    X = prng.standard_normal((n_probe, A_col.shape[1]))
    G = prng.standard_normal((n_probe, A_col.shape[0]))
    out["probe_x"] = X
    out["probe_Ax"] = np.stack([A_col @ x for x in X])
    out["probe_g"] = G
    out["probe_ATg"] = np.stack([A_col.T @ g for g in G])
    # fixed-matrix probes
    C = __import__("scipy.sparse", fromlist=["vstack"]).vstack(
        [solver.C_jerk, solver.C_acc, solver.C_vel, solver.C_pos], format="csc")
    out["probe_Cx"] = np.stack([C @ x for x in X])
    Gf = prng.standard_normal((n_probe, C.shape[0]))
    out["probe_gf"] = Gf
    out["probe_CTg"] = np.stack([C.T @ g for g in Gf])
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={N} K={K} rows={A_col.shape[0]} nnz={A_col.nnz} feasible={feas} -> {os.path.getsize(path)/1e3:.1f} KB")


def main():
    SCP, generate_positions = import_reference()

    # (i) N=4, K=20 (T=10, h=0.5): full CSC arrays.  Scenario: reference generator, random.seed(1).
    random.seed(1)
    p0, pf = generate_positions(4, 0.8)
    run_case(SCP, "ref_n4_k20", 4, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf, None, None, 101, 0.5, True)

    # (ii) N=20 and N=40, K=50: hashes + probes instead of the 11-45 MB matrices.
    random.seed(20)
    p0, pf = generate_positions(20, 0.8)
    run_case(SCP, "ref_n20_k50", 20, 10.0, 0.2, 0.8, [0, 0, 20, 20], p0, pf, None, None, 102, 0.3, False, n_probe=2)
    random.seed(40)
    p0, pf = generate_positions(40, 0.8)
    run_case(SCP, "ref_n40_k50", 40, 10.0, 0.2, 0.8, [0, 0, 20, 20], p0, pf, None, None, 103, 0.3, False, n_probe=1)

    # (iii) the reference's __main__ crossing demo (scp.py:844-862), with non-zero v0/vf added in a
    # second variant so that the velocity terms of the bounds / rhs are exercised.
    p0 = np.array([[-2.0, -2.0], [0.0, -2.0], [2.0, -2.0]])
    pf = np.array([[2.0, 2.0], [0.0, 2.0], [-2.0, 2.0]])
    run_case(SCP, "ref_cross3_k15", 3, 3.0, 0.2, 0.5, [-5, -5, 500, 200], p0, pf, None, None, 104, 1.0, True)
    v0 = np.array([[0.3, -0.2], [0.0, 0.5], [-0.4, 0.1]])
    vf = np.array([[0.1, 0.2], [-0.3, 0.0], [0.2, -0.1]])
    run_case(SCP, "ref_cross3_k15_vel", 3, 3.0, 0.2, 0.5, [-5, -5, 500, 200], p0, pf, v0, vf, 105, 1.0, True)

    # (iv) scenario generator under random.seed(s): N in {4, 10, 20, 50}.
    gen = {}
    for n, seed, md in ((4, 1, 0.8), (10, 7, 0.8), (20, 20, 0.8), (50, 3, 0.8), (20, 5, 1.0)):
        random.seed(seed)
        a, b = generate_positions(n, md)
        gen[f"init_n{n}_s{seed}_d{md}"] = a
        gen[f"final_n{n}_s{seed}_d{md}"] = b
    # failure mode: capacity exceeded -> ValueError (position_generator.py:58-59)
    random.seed(0)
    try:
        generate_positions(80, 0.8)
        gen["n80_raises"] = np.asarray(False)
    except ValueError as e:
        gen["n80_raises"] = np.asarray(True)
        gen["n80_message"] = np.asarray(str(e))
    np.savez_compressed(os.path.join(HERE, "ref_generator.npz"), **gen)
    print("ref_generator: ok")


if __name__ == "__main__":
    main()
