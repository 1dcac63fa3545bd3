"""The column kernels of the QP (csrc/scp_qp_fused.hip: cg1_col_kernel, cg1_resid_col_kernel, qp0_col_kernel) apply the
integrator blocks F = [J; I; V; S], S0 and their transposes as cumulative sums instead of dense products.  This pins
the identities they rely on against the explicit blocks of the oracle (oracle/scp_oracle.py:time_blocks, which follows
scp.py:10-28, :198-203, :227-232, :489-491)."""
import numpy as np
import pytest

from oracle import scp_oracle as so


def rsum(v):
    """reverse inclusive cumulative sum"""
    return np.cumsum(v[::-1])[::-1]


def rsum_excl(v):
    return rsum(v) - v


def csum_excl(v):
    return np.cumsum(v) - v


@pytest.mark.parametrize("K,h", [(2, 0.5), (20, 0.5), (50, 0.2), (65, 0.2), (120, 0.2)])
def test_integrator_blocks_are_scans(K, h):
    rng = np.random.default_rng(K)
    J, I, V, S, S0 = so.time_blocks(K, h)
    hh = h * h
    p = rng.standard_normal(K)
    c1 = np.cumsum(p)
    c2 = csum_excl(c1)
    c1_prev = np.concatenate([[0.0], c1[:-1]])
    # forward: F p and S0 p (cg1_col_kernel "S0 p, F p: forward scans")
    np.testing.assert_allclose(J @ p, (p[1:] - p[:-1]) / h, rtol=0, atol=1e-12)
    np.testing.assert_allclose(V @ p, h * c1, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(S @ p, hh * (c2 + 0.5 * c1), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(S0 @ p, hh * (c2 - 0.5 * c1_prev), rtol=1e-12, atol=1e-12)
    # transposed: F^T w + S0^T g in three reverse scans (cg1_col_kernel "r: reverse scans")
    wj, wa, wv, wp, g = (rng.standard_normal(K - 1), rng.standard_normal(K), rng.standard_normal(K),
                         rng.standard_normal(K), rng.standard_normal(K))
    dense = J.T @ wj + wa + V.T @ wv + S.T @ wp + S0.T @ g
    wj_pad = np.concatenate([wj, [0.0]])            # w_j[K-1] = 0
    wj_prev = np.concatenate([[0.0], wj])           # w_j[k-1], w_j[-1] = 0
    u1 = h * wv + 0.5 * hh * (wp - g)
    u2 = wp + g
    scans = (wj_prev - wj_pad) / h + wa + rsum(u1) + 0.5 * hh * g + hh * rsum_excl(rsum(u2))
    np.testing.assert_allclose(scans, dense, rtol=1e-11, atol=1e-11)


def test_right_hand_side_without_Hx_product():
    """r = sigma x + A^T(rho z - y) - H_f x  ==  -2 x + F^T(rho w (z - F x) - y)   (H_f = (2 + sigma) I + rho F^T w F):
    the form the single-step pipeline evaluates with the carried F x slab."""
    K, h, rho, sigma = 30, 0.2, 0.37, 1e-6
    rng = np.random.default_rng(1)
    J, I, V, S, _ = so.time_blocks(K, h)
    F = np.vstack([J, I, V, S])
    w = np.ones(F.shape[0])
    w[2 * K - 1 + K - 1] = w[3 * K - 1 + K - 1] = 1e3  # equality rows (rho_eq_scale)
    x, z, y = rng.standard_normal(K), rng.standard_normal(F.shape[0]), rng.standard_normal(F.shape[0])
    Hf = (2.0 + sigma) * np.eye(K) + rho * F.T @ (w[:, None] * F)
    lhs = sigma * x + F.T @ (rho * w * z - y) - Hf @ x
    rhs = -2.0 * x + F.T @ (rho * w * (z - F @ x) - y)
    np.testing.assert_allclose(rhs, lhs, rtol=1e-9, atol=1e-9)
