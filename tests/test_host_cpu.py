"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol of include/scp_hip.h
(no compute calls without a GPU), the product path fails loudly without a GPU, scenario generators match the
reference's fixtures, the batch CLI keeps the reference's output schema, and the oracle's QP solvers agree with
each other and with an independent optimiser."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scp_hip.h")


def test_library_exports_every_declared_symbol():
    from path_planning import _hip

    path = _hip.library_path()
    assert os.path.exists(path), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(scp_[a-z0-9_]+)\s*\(", text)) - {"scp_eta_stride"}  # static inline helper
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.scp_abi_version() == _hip.ABI_VERSION == int(re.search(r"#define SCP_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    # struct layouts agree with the header
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:  # sizeof as the C compiler sees the header
        src = os.path.join(tmp, "sz.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include "scp_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu", sizeof(scp_qp_settings), '
                    'sizeof(scp_qp_info), sizeof(scp_pair_stats), sizeof(scp_solve_options), sizeof(scp_qp_record), '
                    'sizeof(scp_solve_result));return 0;}\n')
        exe = os.path.join(tmp, "sz")
        subprocess.run(["gcc", "-I", os.path.dirname(HEADER), src, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert [ctypes.sizeof(_hip.QpSettings), ctypes.sizeof(_hip.QpInfo), 32, ctypes.sizeof(_hip.SolveOptions),
            ctypes.sizeof(_hip.QpRecord), ctypes.sizeof(_hip.SolveResult)] == sizes, sizes
    s = _hip.default_settings()
    assert (s.rho, s.sigma, s.alpha, s.eps_abs, s.eps_rel, s.max_iter, s.check_termination) == (
        0.1, 1e-6, 1.6, 1e-3, 1e-3, 4000, 25)  # OSQP defaults


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from path_planning import _hip
    from path_planning.solvers.scp import SCP

    with pytest.raises(_hip.HipError, match="no GPU"):
        SCP(n_vehicles=2, time_horizon=1.0, time_step=0.2, min_distance=0.5, verbose=False)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ba-path-planning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dirpath, f)


def test_generator_matches_reference_fixture(golden_dir):
    import random

    from path_planning.scenarios.position_generator import generate_positions

    g = np.load(os.path.join(golden_dir, "ref_generator.npz"))
    for n, s, md in ((4, 1, 0.8), (10, 7, 0.8), (20, 20, 0.8), (50, 3, 0.8), (20, 5, 1.0)):
        a, b = generate_positions(n, md, seed=s)
        np.testing.assert_array_equal(a, g[f"init_n{n}_s{s}_d{md}"])
        np.testing.assert_array_equal(b, g[f"final_n{n}_s{s}_d{md}"])
        random.seed(s)  # the reference's own way: global random module
        a2, b2 = generate_positions(n, md)
        np.testing.assert_array_equal(a, a2)
    assert bool(g["n80_raises"])
    with pytest.raises(ValueError, match=str(g["n80_message"])):
        generate_positions(80, 0.8, seed=0)


def test_grid_swap_scenarios():
    from path_planning.scenarios.position_generator import generate_grid_swap, straight_line_min_distance

    for n, dim in ((64, 2), (64, 3), (100, 2)):
        a, b, space = generate_grid_swap(n, seed=1000 * n, dim=dim)
        assert a.shape == b.shape == (n, dim) and len(space) == 2 * dim
        a2, b2, _ = generate_grid_swap(n, seed=1000 * n, dim=dim)
        np.testing.assert_array_equal(a, a2)  # seeded
        d0 = np.linalg.norm(a[:, None] - a[None], axis=2) + 10 * np.eye(n)
        assert d0.min() > 0.8 and (np.linalg.norm(b[:, None] - b[None], axis=2) + 10 * np.eye(n)).min() > 0.8
        assert np.linalg.norm(a - b, axis=1).max() < 10.0  # reachable with |v| <= 2 in T = 10
        assert straight_line_min_distance(a, b).min() > 0.1  # no head-on swaps
        assert (a >= np.array(space[:dim])).all() and (b <= np.array(space[dim:])).all()


def test_batch_cli_schema(tmp_path, monkeypatch):
    """JSON/CSV schema of compute_trajectories_batch.py:91-100, :158 (solver stubbed: no GPU here)."""
    import csv
    import json

    from path_planning.cli import compute_trajectories_batch as cli

    def fake_trial(N, cfg, rng=None, seed=None, device=None, scenario=None, save_path=None, pool=None):
        assert scenario is not None and scenario[0].shape == (N, 2)  # generated before the timed loop
        return {"N": N, "status": "success" if seed % 2 == 0 else "error", "time_sec": 0.1 * N + 0.01 * (seed % 7),
                "error": None if seed % 2 == 0 else "boom", "K": 50, "T": cfg["time_horizon"], "h": cfg["time_step"],
                "seed": seed, "scp_iterations": 3}

    monkeypatch.setattr(cli, "run_single_trial", fake_trial)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    res = cli.main(["--Ns", "4", "6", "--trials", "3", "--seed", "10", "--results-dir", str(tmp_path)])
    assert set(res) == {"meta", "runs", "summary"} and res["meta"]["schema_version"] == "1.0"
    assert len(res["runs"]) == 6 and [r["trial_index"] for r in res["runs"]] == [0, 1, 2, 0, 1, 2]
    assert res["runs"][0]["seed"] == 10 + 1000 * 4 + 0  # rng_seed + 1000*N + trial (:108)
    s4 = res["summary"]["4"]
    assert set(s4) == {"count", "errors", "min", "max", "mean", "median", "p25", "p75", "std"}
    assert s4["count"] + s4["errors"] == 3
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 2 and files[0].endswith(".csv") and files[1].endswith(".json")
    rows = list(csv.reader(open(tmp_path / files[0])))
    assert rows[0] == ["N", "trial_index", "status", "time_sec", "K", "T", "h", "error"] and len(rows) == 7
    assert json.load(open(tmp_path / files[1]))["summary"].keys() == {"4", "6"}
    # scenario-parallel job split: every job on exactly one rank
    cfg = dict(cli.CONFIG, Ns=[4, 6], trials_per_N=3)
    jobs = [cli.jobs_for_rank(cfg, r, 4) for r in range(4)]
    assert sorted(j for part in jobs for j in part) == [(n, t) for n in (4, 6) for t in range(3)]


def test_qp_oracles_agree_and_match_trust_constr():
    """QP#0 of the N=4, K=20 plumbing case: structured ADMM == explicit OSQP restatement == scipy trust-constr
    == closed-form min-norm solution (no box row is active there)."""
    from scipy.optimize import LinearConstraint, minimize

    from oracle import qp_oracle as qo
    from oracle import scp_oracle as so
    from path_planning.scenarios.position_generator import generate_positions

    p0, pf = generate_positions(4, 0.8, seed=1)
    prob = so.make_problem(4, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf)
    xs, _, info = qo.admm_structured(prob, st=qo.Settings(eps_abs=1e-9, eps_rel=1e-9))
    assert info["status_val"] == 1
    C, l, u = so.stack_fixed(prob)
    r = qo.osqp_explicit(2 * sp.eye(prob.n, format="csc"), np.zeros(prob.n), C, l, u, eps_abs=1e-9, eps_rel=1e-9,
                         max_iter=20000)
    assert r["status_val"] == 1
    np.testing.assert_allclose(xs.ravel(), r["x"], rtol=0, atol=1e-7)
    # closed form: min ||x||^2 s.t. E x = e (the two equality rows per agent/axis)
    eq = np.nonzero(l == u)[0]
    E = C.tocsr()[eq].toarray()
    x_mn = E.T @ np.linalg.solve(E @ E.T, l[eq])
    assert np.all(C @ x_mn >= l - 1e-9) and np.all(C @ x_mn <= u + 1e-9)
    np.testing.assert_allclose(xs.ravel(), x_mn, rtol=0, atol=1e-7)
    # independent optimiser on a smaller instance (trust-constr is slow on dense 160-variable problems)
    prob2 = so.make_problem(2, 5.0, 0.5, 0.8, [0, 0, 20, 20], p0[:2], pf[:2] * 0.5 + p0[:2] * 0.5)
    C2, l2, u2 = so.stack_fixed(prob2)
    x2, _, i2 = qo.admm_structured(prob2, st=qo.Settings(eps_abs=1e-9, eps_rel=1e-9))
    assert i2["status_val"] == 1
    res = minimize(lambda x: x @ x, np.zeros(prob2.n), jac=lambda x: 2 * x, hess=lambda x: 2 * np.eye(prob2.n),
                   constraints=[LinearConstraint(C2.toarray(), l2, u2)], method="trust-constr",
                   options={"gtol": 1e-12, "xtol": 1e-14, "maxiter": 300})
    np.testing.assert_allclose(res.x, x2.ravel(), rtol=0, atol=1e-6)


def test_structured_vs_explicit_collision_qp():
    """Joint QP with collision rows: constraint generation (working set) reaches the optimum of the FULL QP."""
    from oracle import qp_oracle as qo
    from oracle import scp_oracle as so
    from path_planning.scenarios.position_generator import generate_positions

    p0, pf = generate_positions(4, 0.8, seed=1)
    prob = so.make_problem(4, 10.0, 0.5, 0.8, [0, 0, 20, 20], p0, pf)
    x0, _, _ = qo.admm_structured(prob, st=qo.Settings(eps_abs=1e-9, eps_rel=1e-9))
    pos, _ = so.kinematics(prob, x0)
    eta, l_col, dist = so.linearize_pairs(prob, pos)
    xs, _, info = qo.admm_structured(prob, eta, l_col, dist, x0=x0,
                                     st=qo.Settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=50000, margin=0.05))
    assert info["status_val"] == 1 and info["working_rows"] < prob.m_col
    C, lf, uf = so.stack_fixed(prob)
    A = sp.vstack([C, so.collision_matrix_explicit(prob, eta)], format="csc")
    r = qo.osqp_explicit(2 * sp.eye(prob.n, format="csc"), np.zeros(prob.n), A, np.hstack([lf, l_col]),
                         np.hstack([uf, np.full(l_col.size, np.inf)]), x0=x0.ravel(), eps_abs=1e-10, eps_rel=1e-10,
                         max_iter=200000)
    assert r["status_val"] == 1
    np.testing.assert_allclose(xs.ravel(), r["x"], rtol=0, atol=1e-6)
    assert np.all(A @ xs.ravel() >= np.hstack([lf, l_col]) - 1e-6)


def test_graft_entry_build_runs_on_cpu():
    """The driver's "does it build" check: __graft_entry__.build() compiles libscp_hip.so for gfx950 and the oracle's C
    restatement (incremental: a no-op after the first build), imports the package and checks exports + ABI version."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    g.build()


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirrors of the [host] structs (scp_qp_settings, scp_qp_info, scp_solve_options, scp_qp_record,
    scp_solve_result, scp_pair_stats) have the size and the field offsets gcc gives the C declarations of include/scp_hip.h."""
    import ctypes
    import subprocess

    from path_planning import _hip

    fields = {
        "scp_qp_settings": (_hip.QpSettings, ["rho", "max_iter", "cg_iters", "rho_col_scale", "persistent"]),
        "scp_qp_info": (_hip.QpInfo, ["status_val", "working_rows", "solve_ms", "pipeline", "rho_switches_in_kernel"]),
        "scp_solve_options": (_hip.SolveOptions, ["max_iterations", "working_set_margin", "convergence_tolerance", "row_free",
                                                  "carry_rho"]),
        "scp_qp_record": (_hip.QpRecord, ["status_val", "pipeline", "working_rows", "added", "rel_step", "violations_ms",
                                          "persist_launches", "reserved"]),
        "scp_solve_result": (_hip.SolveResult, ["n_iterations", "first_violation", "time_sec"]),
    }
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "scp_hip.h"', "int main(void) {"]
    for name, (_, fs) in fields.items():
        src.append(f'  printf("{name} %zu", sizeof({name}));')
        for f in fs:
            src.append(f'  printf(" %zu", offsetof({name}, {f}));')
        src.append('  printf("\\n");')
    src.append('  printf("abi %d\\n", SCP_ABI_VERSION);')
    src.append("  return 0; }")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {ln.split()[0]: [int(v) for v in ln.split()[1:]] for ln in out if ln.strip()}
    for name, (cls, fs) in fields.items():
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f in fs]
        assert got[name] == want, (name, got[name], want)
    assert got["abi"] == [_hip.ABI_VERSION]
    assert _hip.pipeline_names(0) == "none" and _hip.pipeline_names((1 << 1) | (1 << 3)) == "persistent+three-launch"
    assert _hip.pipeline_names(1 << 7) == "persistent8-lean"
