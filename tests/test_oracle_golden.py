"""Pin the numpy oracle (oracle/scp_oracle.py) against golden vectors produced by the REAL
reference (tests/golden/make_golden.py): rows a0, a2, a4, a5, a7, a8 of SURVEY.md section 8."""
import hashlib
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import scp_oracle as so

CASES = ["ref_n4_k20", "ref_cross3_k15", "ref_cross3_k15_vel", "ref_n20_k50", "ref_n40_k50"]


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    prob = so.make_problem(int(g["N"]), float(g["T"]), float(g["h"]), float(g["R"]), g["space"],
                           g["p0"], g["pf"], g["v0"], g["vf"])
    return g, prob


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", CASES)
def test_problem_sizes(golden_dir, name):
    g, prob = load(golden_dir, name)
    assert prob.K == int(g["K"])  # K = int(T/h), scp.py:43
    assert tuple(g["A_col_shape"]) == (prob.m_col, prob.n)
    # nnz formulas asserted by the reference itself (scp.py:259-262)
    N, K = prob.N, prob.K
    assert int(g["C_acc_nnz"]) == 2 * N * K
    assert int(g["C_jerk_nnz"]) == 4 * N * (K - 1)
    assert int(g["C_vel_nnz"]) == N * K * (K + 1)
    assert int(g["C_pos_nnz"]) == N * K * (K + 1)
    assert int(g["A_col_nnz"]) == prob.pairs * 2 * K * (K - 1)


@pytest.mark.parametrize("name", CASES)
def test_fixed_bounds_bitwise(golden_dir, name):
    g, prob = load(golden_dir, name)
    b = so.fixed_bounds(prob)
    for nm in ("jerk", "acc", "vel", "pos"):
        np.testing.assert_array_equal(b[nm][0].ravel(), g[f"l_{nm}"])
        np.testing.assert_array_equal(b[nm][1].ravel(), g[f"u_{nm}"])


@pytest.mark.parametrize("name", CASES)
def test_fixed_matrices(golden_dir, name):
    g, prob = load(golden_dir, name)
    mats = dict(zip(("jerk", "acc", "vel", "pos"), so.fixed_matrices_explicit(prob)))
    for nm, C in mats.items():
        C = sp.csc_matrix(C)
        C.sort_indices()
        assert C.shape == tuple(g[f"C_{nm}_shape"])
        assert C.nnz == int(g[f"C_{nm}_nnz"])
        if f"C_{nm}_data" in g.files:
            np.testing.assert_array_equal(C.indices, g[f"C_{nm}_indices"])
            np.testing.assert_array_equal(C.indptr, g[f"C_{nm}_indptr"])
            np.testing.assert_allclose(C.data, g[f"C_{nm}_data"], rtol=0, atol=1e-15)
        else:
            ref = str(g[f"C_{nm}_sha"])
            assert sha(C.indices.astype(np.int32)) == ref[64:128]
            assert sha(C.indptr.astype(np.int32)) == ref[128:192]
    # probes of the stacked matrix
    C, l, u = so.stack_fixed(prob)
    for x, y in zip(g["probe_x"], g["probe_Cx"]):
        np.testing.assert_allclose(C @ x, y, rtol=0, atol=1e-12)
    for gg, y in zip(g["probe_gf"], g["probe_CTg"]):
        np.testing.assert_allclose(C.T @ gg, y, rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", CASES)
def test_kinematics_bitwise(golden_dir, name):
    g, prob = load(golden_dir, name)
    pos, vel = so.kinematics(prob, g["acc"])
    # a4 == a7 bitwise in the reference, and the oracle reproduces both bitwise
    np.testing.assert_array_equal(g["pos_a4"], g["pos_a7"])
    np.testing.assert_array_equal(pos, g["pos_a4"])
    np.testing.assert_array_equal(vel, g["vel_a4"])


@pytest.mark.parametrize("name", CASES)
def test_linearize_pairs(golden_dir, name):
    g, prob = load(golden_dir, name)
    eta, l, dist = so.linearize_pairs(prob, g["pos_a7"])
    assert bool(g["u_col_all_inf"])
    np.testing.assert_allclose(l, g["l_col"], rtol=0, atol=1e-12)
    # compact eta reproduces the explicit matrix: probe products
    for x, y in zip(g["probe_x"], g["probe_Ax"]):
        np.testing.assert_allclose(so.collision_apply(prob, eta, x), y, rtol=0, atol=1e-12)
    for gg, y in zip(g["probe_g"], g["probe_ATg"]):
        np.testing.assert_allclose(so.collision_apply_T(prob, eta, gg), y, rtol=0, atol=1e-11)
    if "A_col_data" in g.files:
        A = so.collision_matrix_explicit(prob, eta)
        A.sort_indices()
        np.testing.assert_array_equal(A.indices, g["A_col_indices"])
        np.testing.assert_array_equal(A.indptr, g["A_col_indptr"])
        np.testing.assert_allclose(A.data, g["A_col_data"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("name", CASES)
def test_check_avoidance(golden_dir, name):
    g, prob = load(golden_dir, name)
    ok, first = so.check_avoidance(prob, g["pos_a4"])
    assert ok == bool(g["feasible"])
    if not ok:
        m = re.search(r"timestep (\d+) between vehicles (\d+) and (\d+): distance = ([0-9.]+)", str(g["feasible_stdout"]))
        assert m, str(g["feasible_stdout"])
        assert (first[0], first[1], first[2]) == (int(m.group(1)), int(m.group(2)), int(m.group(3)))
        assert abs(first[3] - float(m.group(4))) < 6e-4


def test_degenerate_pair_rule():
    # coincident pair -> dist := 1, l = R - 1 + (eta.diff) - ... (scp.py:503-507)
    p0 = np.array([[1.0, 1.0], [1.0, 1.0], [3.0, 1.0]])
    prob = so.make_problem(3, 1.0, 0.2, 0.8, [0, 0, 20, 20], p0, p0 + 1.0)
    pos, _ = so.kinematics(prob, np.zeros(prob.n))
    eta, l, dist = so.linearize_pairs(prob, pos)
    assert dist[0] == 1.0 and np.allclose(eta[0], [1.0, 0.0])
    assert abs(l[0] - (0.8 - 1.0)) < 1e-15
