"""ctypes wrapper of oracle/scp_oracle_c.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The C library is
the single-core CPU statement of the same algorithm as qp_oracle.admm_structured; it is pinned against the numpy
oracle in tests/test_oracle_c.py."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import qp_oracle as qo
from . import scp_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libscp_oracle_c.so")
_lib = None


class _Problem(C.Structure):
    _fields_ = [("N", C.c_int), ("K", C.c_int), ("D", C.c_int), ("h", C.c_double), ("R", C.c_double),
                ("p0", C.c_void_p), ("v0", C.c_void_p), ("pf", C.c_void_p), ("vf", C.c_void_p),
                ("pos_min", C.c_void_p), ("pos_max", C.c_void_p),
                ("vel_min", C.c_double), ("vel_max", C.c_double), ("acc_min", C.c_double), ("acc_max", C.c_double),
                ("jerk_min", C.c_double), ("jerk_max", C.c_double)]


class _Settings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("rho_eq_scale", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("max_iter", C.c_int),
                ("check_termination", C.c_int), ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int),
                ("adaptive_rho_tolerance", C.c_double), ("cg_iters", C.c_int), ("margin", C.c_double),
                ("feas_tol", C.c_double), ("max_rounds", C.c_int), ("rho_col_scale", C.c_double),
                ("eps_prim_inf", C.c_double), ("check_fine", C.c_int), ("check_fine_ratio", C.c_double)]


class _Info(C.Structure):
    _fields_ = [("status_val", C.c_int), ("iter", C.c_int), ("rounds", C.c_int), ("rho_updates", C.c_int),
                ("cg_total", C.c_int), ("working_rows", C.c_int64), ("r_prim", C.c_double), ("r_dual", C.c_double),
                ("rho", C.c_double)]


def load(build=True):
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) and build:
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _lib = C.CDLL(LIB)
        _lib.oc_admm.restype = C.c_int
    return _lib


class CProblem:
    """Keeps the numpy buffers alive next to the C struct."""

    def __init__(self, prob: so.Problem):
        self.prob = prob
        self._keep = [np.ascontiguousarray(a, dtype=np.float64) for a in
                      (prob.p0, prob.v0, prob.pf, prob.vf, prob.pos_min, prob.pos_max)]
        ptr = [a.ctypes.data for a in self._keep]
        self.c = _Problem(prob.N, prob.K, prob.D, prob.h, prob.R, *ptr, prob.vel_min, prob.vel_max, prob.acc_min,
                          prob.acc_max, prob.jerk_min, prob.jerk_max)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def kinematics(prob, acc):
    lib, cp = load(), CProblem(prob)
    acc = np.ascontiguousarray(acc, dtype=np.float64).reshape(prob.N, prob.K, prob.D)
    pos, vel = np.empty_like(acc), np.empty_like(acc)
    lib.oc_kinematics(C.byref(cp.c), _p(acc), _p(pos), _p(vel))
    return pos, vel


def linearize_pairs(prob, pos):
    lib, cp = load(), CProblem(prob)
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    rows = prob.m_col
    eta, l, dist = np.empty((rows, prob.D)), np.empty(rows), np.empty(rows)
    lib.oc_linearize(C.byref(cp.c), _p(pos), _p(eta), _p(l), _p(dist))
    return eta, l, dist


def admm(prob, eta=None, l_col=None, dist=None, x0=None, st: qo.Settings | None = None):
    lib, cp = load(), CProblem(prob)
    st = st or qo.Settings()
    cs = _Settings(st.rho, st.sigma, st.alpha, st.rho_eq_scale, st.eps_abs, st.eps_rel, st.max_iter,
                   st.check_termination, int(st.adaptive_rho), st.adaptive_rho_interval, st.adaptive_rho_tolerance,
                   st.cg_iters, st.margin, st.feas_tol, st.max_rounds, st.rho_col_scale, st.eps_prim_inf, st.check_fine,
                   st.check_fine_ratio)
    x = np.zeros((prob.N, prob.K, prob.D))
    info = _Info()
    keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (eta, l_col, dist, x0)]
    lib.oc_admm(C.byref(cp.c), _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), C.byref(cs), _p(x), C.byref(info))
    d = {k: getattr(info, k) for k, _ in _Info._fields_}
    d["status"] = qo.STATUS_TEXT.get(info.status_val, str(info.status_val))
    return x, d


def scp_solve(prob, max_iterations=15, st: qo.Settings | None = None, max_iter0=4000, carry_rho=False):
    """generate_trajectories (scp.py:131-180) on the C oracle: the same loop as qp_oracle.scp_solve, for sizes the numpy
    oracle is too slow for (a 128-agent solve takes seconds here)."""
    import dataclasses

    st = st or qo.Settings(max_iter=10000)  # scp.py:442
    x, info0 = admm(prob, st=dataclasses.replace(st, max_iter=max_iter0))  # OSQP default for QP#0: 4000 (scp.py:360)
    if info0["status_val"] not in (1, 2):  # scp.py:363-365
        raise RuntimeError(f"OSQP failed: {info0['status']}")
    infos, rels = [info0], []
    pos, vel = kinematics(prob, x)
    feasible = so.check_avoidance(prob, pos)[0]
    it, converged = 0, False
    while it < max_iterations and not converged and not feasible:
        eta, l_col, dist = linearize_pairs(prob, pos)
        st_it = dataclasses.replace(st, rho=infos[-1]["rho"]) if (carry_rho and len(infos) > 1) else st
        xn, info = admm(prob, eta, l_col, dist, x0=x, st=st_it)
        infos.append(info)
        rel = float(np.linalg.norm((xn - x).ravel()) / np.linalg.norm(x.ravel()))  # scp.py:157-159
        rels.append(rel)
        converged = rel <= prob.convergence_tolerance
        x = xn
        pos, vel = kinematics(prob, x)
        it += 1
    return {"positions": pos, "velocities": vel, "accelerations": x, "iterations": it, "converged": converged,
            "initially_feasible": feasible, "rel_steps": rels, "infos": infos}
