"""The C oracle (oracle/scp_oracle_c.c, the cpu_baseline of bench.py) against the numpy oracle and the
reference's golden vectors."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import qp_oracle as qo
from oracle import scp_oracle as so


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    prob = so.make_problem(int(g["N"]), float(g["T"]), float(g["h"]), float(g["R"]), g["space"],
                           g["p0"], g["pf"], g["v0"], g["vf"])
    return g, prob


@pytest.mark.parametrize("name", ["ref_n4_k20", "ref_cross3_k15_vel", "ref_n20_k50"])
def test_c_assembly_vs_reference(golden_dir, name):
    g, prob = load(golden_dir, name)
    pos, vel = co.kinematics(prob, g["acc"])
    np.testing.assert_array_equal(pos, g["pos_a4"])  # bitwise the reference (built with -ffp-contract=off)
    np.testing.assert_array_equal(vel, g["vel_a4"])
    eta, l, dist = co.linearize_pairs(prob, g["pos_a7"])
    np.testing.assert_allclose(l, g["l_col"], rtol=0, atol=1e-12)
    eta_o, l_o, dist_o = so.linearize_pairs(prob, g["pos_a7"])
    np.testing.assert_allclose(eta, eta_o, rtol=0, atol=1e-15)
    np.testing.assert_allclose(dist, dist_o, rtol=0, atol=1e-15)


def ref_problem(n, seed, T=10.0, h=0.2):
    from path_planning.scenarios.position_generator import generate_positions

    p0, pf = generate_positions(n, 0.8, seed=seed)
    return so.make_problem(n, T, h, 0.8, [0, 0, 20, 20], p0, pf)


@pytest.mark.parametrize("n,seed,T,h", [(4, 1, 10.0, 0.5), (10, 7, 10.0, 0.2)])
def test_c_admm_vs_numpy(n, seed, T, h):
    prob = ref_problem(n, seed, T, h)
    st = qo.Settings(max_iter=4000)
    x0, _, i0 = qo.admm_structured(prob, st=st)
    xc, ic = co.admm(prob, st=st)
    assert ic["status_val"] == i0["status_val"] == 1 and ic["iter"] == i0["iter"]
    np.testing.assert_allclose(xc, x0, rtol=0, atol=1e-9)
    pos, _ = so.kinematics(prob, x0)
    eta, l_col, dist = so.linearize_pairs(prob, pos)
    # two PCG steps: iterate for iterate; the default single step is not contractive during the transient and can amplify the
    # 1e-16 between two summation orders (numpy einsum vs C loops) by up to 1e7 on small, nearly degenerate QPs before both
    # converge to the same point: the same counts, waypoints to the solver's tolerance class (the adaptive
    # check cadence stops a QP up to 20 steps earlier than the fixed one did, i.e. earlier in that decay: 3e-4 observed on the
    # 4-agent case where the fixed cadence left 1e-7)
    for cg, tol in ((2, 1e-9), (1, 1e-3)):
        st = qo.Settings(max_iter=10000, cg_iters=cg)
        x1, _, i1 = qo.admm_structured(prob, eta, l_col, dist, x0=x0, st=st)
        xc, ic = co.admm(prob, eta, l_col, dist, x0, st)
        assert ic["status_val"] == i1["status_val"] and ic["iter"] == i1["iter"] and ic["rounds"] == i1["rounds"]
        assert ic["working_rows"] == i1["working_rows"]
        np.testing.assert_allclose(xc, x1, rtol=0, atol=tol)
