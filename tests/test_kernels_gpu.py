"""GPU parity of the stateless HIP entry points (through the C-ABI) against the numpy oracle and the
golden vectors of the real reference: kinematics (a4/a7), bounds (a2), pairwise linearisation (a5),
avoidance check (a8), constraint-generation pass, relative step (a1), fp64 MFMA products."""
import os

import numpy as np
import pytest

from oracle import scp_oracle as so

pytestmark = pytest.mark.gpu

GOLD = ["ref_n4_k20", "ref_cross3_k15", "ref_cross3_k15_vel", "ref_n20_k50", "ref_n40_k50"]
LIMITS = [-2.0, 2.0, -15.0, 15.0, -20.0, 20.0]


@pytest.fixture(scope="module")
def ctx():
    from path_planning import _hip

    c = _hip.Context(0)
    yield c
    c.close()


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    prob = so.make_problem(int(g["N"]), float(g["T"]), float(g["h"]), float(g["R"]), g["space"],
                           g["p0"], g["pf"], g["v0"], g["vf"])
    return g, prob


def synth(N, K, D, seed, h=0.2, R=0.8):
    rng = np.random.default_rng(seed)
    side = max(4.0, 1.6 * N ** (1.0 / D))
    p0 = rng.uniform(0, side, (N, D))
    pf = rng.uniform(0, side, (N, D))
    v0 = rng.uniform(-0.3, 0.3, (N, D))
    vf = rng.uniform(-0.3, 0.3, (N, D))
    space = [-2.0] * D + [side + 2.0] * D
    prob = so.make_problem(N, K * h + 1e-9, h, R, space, p0, pf, v0, vf)
    assert prob.K == K
    acc = 0.3 * rng.standard_normal((N, K, D))
    return prob, acc


@pytest.mark.parametrize("name", GOLD)
def test_kinematics_bitwise_vs_reference(ctx, golden_dir, name):
    g, prob = load(golden_dir, name)
    acc = ctx.tensor(g["acc"].reshape(prob.N, prob.K, 2))
    pos, vel = ctx.kinematics(prob.N, prob.K, 2, prob.h, acc, ctx.tensor(prob.p0), ctx.tensor(prob.v0))
    np.testing.assert_array_equal(pos.cpu().numpy(), g["pos_a4"])
    np.testing.assert_array_equal(vel.cpu().numpy(), g["vel_a4"])


@pytest.mark.parametrize("name", GOLD)
def test_fixed_bounds_bitwise_vs_reference(ctx, golden_dir, name):
    g, prob = load(golden_dir, name)
    lo, hi = ctx.fixed_bounds(prob.N, prob.K, 2, prob.h, LIMITS, g["space"], ctx.tensor(prob.p0), ctx.tensor(prob.v0),
                              ctx.tensor(prob.pf), ctx.tensor(prob.vf))
    l_ref = np.hstack([g[f"l_{k}"] for k in ("jerk", "acc", "vel", "pos")])
    u_ref = np.hstack([g[f"u_{k}"] for k in ("jerk", "acc", "vel", "pos")])
    np.testing.assert_array_equal(lo.cpu().numpy(), l_ref)
    np.testing.assert_array_equal(hi.cpu().numpy(), u_ref)


@pytest.mark.parametrize("name", GOLD)
def test_linearize_vs_reference(ctx, golden_dir, name):
    from path_planning import _hip

    g, prob = load(golden_dir, name)
    pp = _hip.PairPass(ctx, prob.N, prob.K, 2, prob.R, prob.h)
    rows, min_dist, first = pp.linearize(ctx.tensor(g["pos_a7"]), ctx.tensor(prob.p0), ctx.tensor(prob.v0), 0.5)
    l = pp.l_rows().cpu().numpy()
    np.testing.assert_allclose(l, g["l_col"], rtol=0, atol=1e-12)
    eta = pp.eta_rows().cpu().numpy()
    eta_o, l_o, dist_o = so.linearize_pairs(prob, g["pos_a7"])
    np.testing.assert_allclose(eta, eta_o, rtol=0, atol=1e-13)
    # the compact rows reproduce the reference's explicit matrix (probe products)
    for x, y in zip(g["probe_x"], g["probe_Ax"]):
        np.testing.assert_allclose(so.collision_apply(prob, eta, x), y, rtol=0, atol=1e-12)
    # fused selection == oracle selection, fused a8 == reference a8
    sel_o = np.nonzero(dist_o - prob.R < 0.5)[0]
    np.testing.assert_array_equal(np.sort(rows.cpu().numpy()), sel_o)
    assert abs(min_dist - so.min_pair_distance(prob, g["pos_a7"])) < 1e-13
    ok, fv = so.check_avoidance(prob, g["pos_a7"])
    assert ok == bool(g["feasible"])
    if ok:
        assert first == _hip.UINT64_MAX
    else:
        iu, ju = so.pair_index(prob.N)
        q = int(np.nonzero((iu == fv[1]) & (ju == fv[2]))[0][0])
        assert first == fv[0] * prob.pairs + q
    # bitmap marks exactly the selected rows
    bits = pp.bitmap.cpu().numpy().view(np.uint32)
    marked = np.nonzero(np.unpackbits(bits.view(np.uint8), bitorder="little")[: prob.m_col])[0]
    np.testing.assert_array_equal(marked, sel_o)


@pytest.mark.parametrize("N,K,D,seed", [(2, 5, 2, 1), (7, 13, 2, 2), (33, 21, 3, 3), (96, 50, 2, 4), (65, 50, 3, 5),
                                        (130, 17, 2, 6), (300, 7, 2, 7), (257, 5, 3, 8), (256, 4, 2, 9),
                                        (700, 10, 2, 10),  # 2.4 M rows: the three-launch compaction (maps beyond 64 K words)
                                        (1100, 3, 2, 11), (700, 3, 3, 12)])  # slices beyond 16 KB: the L1 / L2 path (scalar loads of agent i)
def test_linearize_vs_oracle_synthetic(ctx, N, K, D, seed):
    from path_planning import _hip

    prob, acc = synth(N, K, D, seed)
    pos, _ = so.kinematics(prob, acc)
    eta_o, l_o, dist_o = so.linearize_pairs(prob, pos)
    pp = _hip.PairPass(ctx, N, K, D, prob.R, prob.h)
    rows, min_dist, first = pp.linearize(ctx.tensor(pos), ctx.tensor(prob.p0), ctx.tensor(prob.v0), 1.0)
    np.testing.assert_allclose(pp.l_rows().cpu().numpy(), l_o, rtol=0, atol=1e-12)
    np.testing.assert_allclose(pp.eta_rows().cpu().numpy(), eta_o, rtol=0, atol=1e-13)
    np.testing.assert_array_equal(np.sort(rows.cpu().numpy()), np.nonzero(dist_o - prob.R < 1.0)[0])
    # a8 on its own entry point
    md, fv, _, _ = ctx.check_avoidance(N, K, D, prob.R, ctx.tensor(pos))
    ok, first_o = so.check_avoidance(prob, pos)
    assert abs(md - so.min_pair_distance(prob, pos)) < 1e-13
    if ok:
        assert fv == _hip.UINT64_MAX
    else:
        iu, ju = so.pair_index(N)
        q = int(np.nonzero((iu == first_o[1]) & (ju == first_o[2]))[0][0])
        assert fv == first_o[0] * prob.pairs + q
    # gather
    w_eta, w_l = pp.gather(rows)
    r = rows.cpu().numpy()
    np.testing.assert_array_equal(w_eta.cpu().numpy(), pp.eta_rows().cpu().numpy()[r])
    np.testing.assert_array_equal(w_l.cpu().numpy(), pp.l_rows().cpu().numpy()[r])


@pytest.mark.parametrize("N,K,cuts", [(41, 19, [0, 101, 102, 500]), (290, 5, [0, 1, 8191, 8193, 20000, 20001])])
def test_linearize_pair_range_shards(ctx, N, K, cuts):
    """Sharded pair ranges (multi-GPU layout) tile the full pass exactly (N = 290: the wave-segment kernel)."""
    from path_planning import _hip

    prob, acc = synth(N, K, 2, 11)
    pos, _ = so.kinematics(prob, acc)
    eta_o, l_o, dist_o = so.linearize_pairs(prob, pos)
    pairs = prob.pairs
    cuts = cuts + [pairs]
    got = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        pp = _hip.PairPass(ctx, prob.N, prob.K, 2, prob.R, prob.h, a, b)
        rows, _, _ = pp.linearize(ctx.tensor(pos), ctx.tensor(prob.p0), ctx.tensor(prob.v0), 0.7)
        got.append(rows.cpu().numpy())
        l = pp.l_rows().cpu().numpy().reshape(prob.K, b - a)
        np.testing.assert_allclose(l, l_o.reshape(prob.K, pairs)[:, a:b], rtol=0, atol=1e-12)
        e = pp.eta_rows().cpu().numpy().reshape(prob.K, b - a, 2)
        np.testing.assert_allclose(e, eta_o.reshape(prob.K, pairs, 2)[:, a:b], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(np.sort(np.concatenate(got)), np.nonzero(dist_o - prob.R < 0.7)[0])


@pytest.mark.parametrize("N,K,D", [(1024, 50, 2), (1300, 12, 2)])
def test_linearize_item_list_shards_tile_the_full_pass(ctx, N, K, D):
    """Large problems: the linearisation grid is a list of work items in up to three chunk sizes (whole time steps per size,
    large first).  Pair-range shards with odd cuts, each with its OWN item list, must reproduce the rows, the selection and
    the statistics of the full pass bit for bit (LDS path at 1024 agents, L1 / L2 path with scalar loads at 1300); the full
    pass itself is held against the oracle on a sample of rows."""
    import torch
    from path_planning import _hip

    prob, acc = synth(N, K, D, 41)
    pos, _ = so.kinematics(prob, acc)
    pos_t, p0, v0 = ctx.tensor(pos), ctx.tensor(prob.p0), ctx.tensor(prob.v0)
    pairs = prob.pairs
    full = _hip.PairPass(ctx, N, K, D, prob.R, prob.h)
    rows_f, md_f, fv_f = full.linearize(pos_t, p0, v0, 0.4)
    eta_f = full.eta_rows().reshape(K, pairs, D)
    l_f = full.l_rows().reshape(K, pairs)
    cuts = [0, 100001, pairs // 2 + 1, pairs]
    got, mds, fvs = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        pp = _hip.PairPass(ctx, N, K, D, prob.R, prob.h, a, b)
        rows, md, fv = pp.linearize(pos_t, p0, v0, 0.4)
        assert torch.equal(pp.l_rows().reshape(K, b - a), l_f[:, a:b])
        assert torch.equal(pp.eta_rows().reshape(K, b - a, D), eta_f[:, a:b])
        got.append(rows)
        mds.append(md)
        fvs.append(fv)
    assert torch.equal(torch.sort(torch.cat(got)).values, rows_f)
    assert min(mds) == md_f and min(fvs) == fv_f
    # a sample of rows of the full pass against the oracle's arithmetic (scp.py:498-509, :543-549)
    rng = np.random.default_rng(42)
    iu, ju = so.pair_index(N)
    ks, qs = rng.integers(0, K, 4000), rng.integers(0, pairs, 4000)
    h = prob.h
    for k, q in zip(ks[:4000], qs[:4000]):
        i, j = int(iu[q]), int(ju[q])
        diff = pos[i, k] - pos[j, k]
        dist = float(np.linalg.norm(diff))
        eta = diff / dist
        ci = prob.p0[i] + (k * h) * prob.v0[i]
        cj = prob.p0[j] + (k * h) * prob.v0[j]
        l = prob.R + (eta @ diff - dist) - eta @ (ci - cj)
        assert abs(float(l_f[k, q]) - l) < 1e-12 and np.abs(eta_f[k, q].cpu().numpy() - eta).max() < 1e-13


@pytest.mark.parametrize("N,K,D,seed,cut", [(2, 5, 2, 1, None), (33, 21, 3, 3, None), (96, 50, 2, 4, None), (65, 50, 3, 5, None),
                                             (300, 7, 2, 7, None), (700, 10, 2, 10, None), (130, 17, 2, 6, (1000, 6001)),
                                             (1100, 3, 2, 11, None), (700, 3, 3, 12, None)])
def test_row_free_linearisation_is_bit_identical(ctx, N, K, D, seed, cut):
    """scp_select_pairs + scp_qp_add_rows_at (the row-free loop: no eta / l planes) against scp_linearize_pairs + gather +
    scp_qp_add_rows: the same selected rows, bitmap and a8 statistics, and BITWISE the same working rows (eta, l, z_c)."""
    import torch
    from path_planning import _hip

    prob, acc = synth(N, K, D, seed)
    pos, _ = so.kinematics(prob, acc)
    q0, q1 = cut if cut else (0, prob.pairs)
    pos_t, p0, v0 = ctx.tensor(pos), ctx.tensor(prob.p0), ctx.tensor(prob.v0)
    a = _hip.PairPass(ctx, N, K, D, prob.R, prob.h, q0, q1)
    b = _hip.PairPass(ctx, N, K, D, prob.R, prob.h, q0, q1)
    rows_a, md_a, fv_a = a.linearize(pos_t, p0, v0, 0.6)
    rows_b, md_b, fv_b = b.select(pos_t, 0.6)
    assert b._eta is None and b._l is None  # nothing was allocated, nothing written
    assert torch.equal(rows_a, rows_b) and md_a == md_b and fv_a == fv_b
    assert torch.equal(a.bitmap, b.bitmap)
    if rows_a.numel() == 0:
        return
    space = np.concatenate([prob.pos_min, prob.pos_max])
    x0 = ctx.tensor(acc.reshape(N, K, D))
    got = {}
    for mode in ("gather", "at"):
        qp = _hip.QP(ctx, N, K, D, prob.h, _hip.default_settings(), row_capacity=int(rows_a.numel()))
        qp.set_problem(LIMITS, space, p0, v0, ctx.tensor(prob.pf), ctx.tensor(prob.vf))
        qp.reset(x0)
        if mode == "gather":
            qp.add_rows(rows_a, *a.gather(rows_a))
        else:
            qp.add_rows_at(rows_b, pos_t, p0, v0, prob.R)
        got[mode] = {k: qp.peek(k).cpu().numpy() for k in ("w_eta", "w_l", "zc", "yc")}
        qp.close()
    for k in got["gather"]:
        np.testing.assert_array_equal(got["at"][k], got["gather"][k], err_msg=k)


@pytest.mark.parametrize("N,K,D", [(300, 4, 2), (1100, 3, 2)])
def test_degenerate_pairs_in_interior_workgroups(ctx, N, K, D):
    """Coincident agents whose rows lie in INTERIOR workgroups of every pass (the streaming form: regular formulas for all
    rows, the degenerate rule of scp.py:503-507 applied behind the group's smallest distance) -- LDS path and L1 / L2 path."""
    from path_planning import _hip

    prob, acc = synth(N, K, D, 31)
    twins = [(5, 100), (150, 200), (N - 40, N - 3)]
    p0, v0 = prob.p0.copy(), prob.v0.copy()
    acc = acc.reshape(N, K, D).copy()
    for a, b in twins:
        p0[b], v0[b], acc[b] = p0[a], v0[a], acc[a]
    prob = so.make_problem(N, K * prob.h + 1e-9, prob.h, prob.R, np.concatenate([prob.pos_min, prob.pos_max]), p0, prob.pf, v0, prob.vf)
    pos, _ = so.kinematics(prob, acc)
    eta_o, l_o, dist_o = so.linearize_pairs(prob, pos)
    assert np.sum(dist_o == 1.0) >= len(twins) * K  # (the rule sets dist := 1 for them)
    pos_t, p0_t, v0_t = ctx.tensor(pos), ctx.tensor(prob.p0), ctx.tensor(prob.v0)
    pp = _hip.PairPass(ctx, N, K, D, prob.R, prob.h)
    rows, min_dist, first = pp.linearize(pos_t, p0_t, v0_t, 0.3)
    np.testing.assert_allclose(pp.l_rows().cpu().numpy(), l_o, rtol=0, atol=1e-12)
    np.testing.assert_allclose(pp.eta_rows().cpu().numpy(), eta_o, rtol=0, atol=1e-13)
    want = np.nonzero(dist_o - prob.R < 0.3)[0]
    np.testing.assert_array_equal(np.sort(rows.cpu().numpy()), want)
    iu, ju = so.pair_index(N)
    q_first = int(np.nonzero((iu == twins[0][0]) & (ju == twins[0][1]))[0][0])
    assert min_dist == 0.0 and first <= q_first  # (the first twin pair at k = 0, unless a closer-than-R pair precedes it)
    sel = _hip.PairPass(ctx, N, K, D, prob.R, prob.h)
    rows_s, md_s, fv_s = sel.select(pos_t, 0.3)
    np.testing.assert_array_equal(rows_s.cpu().numpy(), rows.cpu().numpy())
    assert md_s == min_dist and fv_s == first
    md_c, fv_c, _, _ = ctx.check_avoidance(N, K, D, prob.R, pos_t)
    assert md_c == min_dist and fv_c == first
    # the recomputing violations pass at a displaced point
    x = acc + 0.2 * np.random.default_rng(32).standard_normal(acc.shape)
    pos_new, _ = so.kinematics(prob, x)
    viol = l_o - so.collision_apply(prob, eta_o, x.ravel())
    W = set(rows.cpu().numpy().tolist())
    want_v = sorted(r for r in np.nonzero(viol > 1e-6)[0].tolist() if r not in W)
    new_rows, max_v = pp.violations(ctx.tensor(pos_new), p0_t, v0_t, 1e-6, recompute=True)
    assert sorted(new_rows.cpu().numpy().tolist()) == want_v
    assert abs(max_v - viol.max()) < 1e-11


def test_degenerate_pair(ctx):
    from path_planning import _hip

    p0 = np.array([[1.0, 1.0], [1.0, 1.0], [3.0, 1.0]])
    prob = so.make_problem(3, 1.0, 0.2, 0.8, [0, 0, 20, 20], p0, p0 + 1.0)
    pos, _ = so.kinematics(prob, np.zeros(prob.n))
    eta_o, l_o, _ = so.linearize_pairs(prob, pos)
    pp = _hip.PairPass(ctx, 3, prob.K, 2, prob.R, prob.h)
    rows, _, _ = pp.linearize(ctx.tensor(pos), ctx.tensor(prob.p0), ctx.tensor(prob.v0), 0.5)
    np.testing.assert_allclose(pp.eta_rows().cpu().numpy(), eta_o, rtol=0, atol=1e-15)
    np.testing.assert_allclose(pp.l_rows().cpu().numpy(), l_o, rtol=0, atol=1e-15)
    # the row-free pass applies the same degenerate rule (dist := 1, eta = e_0): same selection, same recomputed rows
    rows2, _, _ = _hip.PairPass(ctx, 3, prob.K, 2, prob.R, prob.h).select(ctx.tensor(pos), 0.5)
    np.testing.assert_array_equal(rows2.cpu().numpy(), rows.cpu().numpy())
    qp = _hip.QP(ctx, 3, prob.K, 2, prob.h, _hip.default_settings(), row_capacity=64)
    qp.set_problem(LIMITS, np.array([0.0, 0, 20, 20]), ctx.tensor(prob.p0), ctx.tensor(prob.v0), ctx.tensor(prob.pf), ctx.tensor(prob.vf))
    qp.reset(None)
    qp.add_rows_at(rows2, ctx.tensor(pos), ctx.tensor(prob.p0), ctx.tensor(prob.v0), prob.R)
    r = rows.cpu().numpy()
    np.testing.assert_array_equal(qp.peek("w_eta").cpu().numpy().reshape(-1, 2), pp.eta_rows().cpu().numpy()[r])
    np.testing.assert_array_equal(qp.peek("w_l").cpu().numpy(), pp.l_rows().cpu().numpy()[r])
    qp.close()


@pytest.mark.parametrize("recompute", [True, False])
@pytest.mark.parametrize("N,K,D,seed", [(9, 12, 2, 21), (40, 50, 2, 22), (30, 25, 3, 23), (280, 6, 2, 24), (270, 3, 3, 25),
                                        (700, 10, 2, 26),  # (three-launch compaction, incl. its overflow path)
                                        (1100, 3, 2, 27), (700, 3, 3, 28)])  # (the L1 / L2 path)
def test_collision_violations_pass(ctx, N, K, D, seed, recompute):
    """Both forms of the pass: recomputing eta / l from the linearisation point (scp_collision_violations_at, the
    default: nothing streamed from HBM) and reading the stored rows (scp_collision_violations)."""
    from path_planning import _hip

    prob, acc = synth(N, K, D, seed)
    pos, _ = so.kinematics(prob, acc)
    eta_o, l_o, dist_o = so.linearize_pairs(prob, pos)
    pp = _hip.PairPass(ctx, N, K, D, prob.R, prob.h)
    p0, v0 = ctx.tensor(prob.p0), ctx.tensor(prob.v0)
    rows, _, _ = pp.linearize(ctx.tensor(pos), p0, v0, 0.2)
    W = set(rows.cpu().numpy().tolist())
    x = acc + 0.2 * np.random.default_rng(seed + 1).standard_normal(acc.shape)
    pos_new, _ = so.kinematics(prob, x)
    ax = so.collision_apply(prob, eta_o, x.ravel())
    viol = l_o - ax
    want = sorted(r for r in np.nonzero(viol > 1e-6)[0].tolist() if r not in W)
    new_rows, max_v = pp.violations(ctx.tensor(pos_new), p0, v0, 1e-6, recompute=recompute)
    assert sorted(new_rows.cpu().numpy().tolist()) == want
    assert abs(max_v - viol.max()) < 1e-11
    # second call: everything already marked
    again, _ = pp.violations(ctx.tensor(pos_new), p0, v0, 1e-6, recompute=recompute)
    assert again.numel() == 0
    # a list that is too short: nothing is merged, the pass is repeated with a longer list and finds the same rows
    if len(want) > 3:
        tiny = _hip.PairPass(ctx, N, K, D, prob.R, prob.h, sel_cap=max(len(W), 1))
        rows_t, _, _ = tiny.linearize(ctx.tensor(pos), p0, v0, 0.2)
        tiny.sel_cap = 2
        got, _ = tiny.violations(ctx.tensor(pos_new), p0, v0, 1e-6, recompute=recompute)
        assert sorted(got.cpu().numpy().tolist()) == want and tiny.sel_cap >= len(want)


@pytest.mark.parametrize("N,K,D,seed", [(2, 5, 2, 1), (3, 4, 3, 2), (33, 21, 3, 3), (96, 50, 2, 4), (128, 50, 2, 8), (65, 50, 3, 5),
                                        (280, 6, 2, 24), (257, 41, 2, 9)])
def test_single_launch_passes_equal_the_multi_launch_form(ctx, N, K, D, seed):
    """Small problems run every pairwise pass as ONE launch (staging from the [N][K][D] arrays, per-workgroup partials, the
    last workgroup reduces, compacts and publishes): same row lists, bitmaps and statistics as prep + pass + compaction."""
    import torch
    from path_planning import _hip

    prob, acc = synth(N, K, D, seed)
    pos, _ = so.kinematics(prob, acc)
    x = acc + 0.2 * np.random.default_rng(seed + 1).standard_normal(acc.shape)
    pos_new, _ = so.kinematics(prob, x)
    pos_t, new_t, p0, v0 = ctx.tensor(pos), ctx.tensor(pos_new), ctx.tensor(prob.p0), ctx.tensor(prob.v0)
    got = {}
    try:
        for flag in (1, 0):
            ctx.set_option("single_launch_passes", flag)
            pp = _hip.PairPass(ctx, N, K, D, prob.R, prob.h)
            rows, md, fv = pp.select(pos_t, 0.2)
            bm0 = pp.bitmap.clone()
            new_rows, max_v = pp.violations(new_t, p0, v0, 1e-6)
            again, max_v2 = pp.violations(new_t, p0, v0, 1e-6)
            chk = ctx.check_avoidance(N, K, D, prob.R, new_t)
            # a list that is too short: nothing merged, the repeat with a longer list finds the same rows
            tiny = _hip.PairPass(ctx, N, K, D, prob.R, prob.h)
            tiny.select(pos_t, 0.2)
            tiny.sel_cap = 1
            short, _ = tiny.violations(new_t, p0, v0, 1e-6)
            got[flag] = (rows, md, fv, bm0, new_rows, max_v, again, max_v2, pp.bitmap.clone(), chk, short, tiny.bitmap.clone())
    finally:
        ctx.set_option("single_launch_passes", 1)
    for u, v in zip(got[1], got[0]):
        if torch.is_tensor(u):
            assert torch.equal(u, v)
        else:
            assert u == v
    assert got[1][6].numel() == 0


def test_rel_step(ctx):
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal(12345), rng.standard_normal(12345)
    d, nb, rel = ctx.rel_step(ctx.tensor(a), ctx.tensor(b))
    assert abs(rel - np.linalg.norm(a - b) / np.linalg.norm(b)) < 1e-13
    assert abs(d - np.linalg.norm(a - b)) < 1e-11 and abs(nb - np.linalg.norm(b)) < 1e-11


@pytest.mark.parametrize("use_mfma", [0, 1])
@pytest.mark.parametrize("R,M,C", [(16, 4, 16), (50, 50, 128), (199, 50, 130), (50, 199, 77), (100, 50, 2048), (7, 3, 5)])
def test_gemm_f64(ctx, use_mfma, R, M, C):
    rng = np.random.default_rng(R * 1000 + M * 10 + C)
    # asymmetric integer data first: exact, catches any operand / accumulator layout mix-up
    A = rng.integers(-8, 9, (R, M)).astype(float)
    X = rng.integers(-8, 9, (M, C)).astype(float)
    Y = ctx.gemm(ctx.tensor(A), ctx.tensor(X), use_mfma)
    np.testing.assert_array_equal(Y.cpu().numpy(), A @ X)
    A = rng.standard_normal((R, M))
    X = rng.standard_normal((M, C))
    Y0 = rng.standard_normal((R, C))
    Y = ctx.gemm(ctx.tensor(A), ctx.tensor(X), use_mfma, alpha=0.7, beta=-1.3, Y=ctx.tensor(Y0))
    np.testing.assert_allclose(Y.cpu().numpy(), 0.7 * (A @ X) - 1.3 * Y0, rtol=0, atol=1e-12 * M)
