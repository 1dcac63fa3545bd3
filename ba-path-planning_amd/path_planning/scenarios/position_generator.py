"""Scenario generation for the SCP planner (host side).

Two generators:

* ``generate_positions`` -- the reference layout (corner circles -> central diamond) with the same
  sampling sequence as /root/reference/src/path_planning/scenarios/position_generator.py:44-75,
  :235-248, so that ``random.seed(s)`` (or the new ``seed=`` argument, which the reference leaves as
  a TODO at compute_trajectories_batch.py:40) reproduces the reference's scenarios bit for bit
  (tests/test_host_cpu.py::test_generator_matches_reference_fixture against tests/golden/ref_generator.npz).  Capacity at R = 0.8 is about
  56 agents; beyond that it raises ``ValueError`` like the reference.
* ``generate_grid_swap`` -- synthetic scenarios for N >= 64: starts on a
  jittered grid, goals = starts permuted inside blocks of ``block`` x ``block`` agents, so that every
  displacement stays feasible for |v| <= 2 m/s, T = 10 s.  D = 2 or 3.
"""
import math
import random as _random

import numpy as np

BOX_SIZE = 20.0
CIRCLE_RADIUS = 2.5
CIRCLE_CENTERS = np.array([[3.5, 3.5], [16.5, 3.5], [3.5, 16.5], [16.5, 16.5]])
DIAMOND_CENTER = np.array([10.0, 10.0])
DIAMOND_SIDE = 6.0
_HALF_DIAG = DIAMOND_SIDE / np.sqrt(2)
DIAMOND_VERTICES = DIAMOND_CENTER + _HALF_DIAG * np.array([[0.0, 1.0], [1.0, 0.0], [0.0, -1.0], [-1.0, 0.0]])


class _Sampler:
    """Draws from ``rng`` in the reference's order: circle = randint(0,3) then uniform(0, 2pi);
    diamond = randint(0,3) then uniform(0,1); goal kind = random() < 0.9."""

    def __init__(self, rng):
        self.rng = rng

    def on_circle(self):
        c = CIRCLE_CENTERS[self.rng.randint(0, 3)]
        ang = self.rng.uniform(0, 2 * np.pi)
        return c + CIRCLE_RADIUS * np.array([np.cos(ang), np.sin(ang)])

    def on_diamond(self):
        e = self.rng.randint(0, 3)
        a, b = DIAMOND_VERTICES[e], DIAMOND_VERTICES[(e + 1) % 4]
        t = self.rng.uniform(0, 1)
        return a + t * (b - a)

    def goal(self):
        if self.rng.random() < 0.9:
            return self.on_diamond()
        return self.on_circle()


def _far_enough(p, chosen, min_dist):
    return all(np.linalg.norm(p - c) >= min_dist for c in chosen)


def _rejection(draw, n, min_dist, max_attempts, what):
    chosen = []
    for _ in range(max_attempts):
        if len(chosen) == n:
            break
        cand = draw()
        if _far_enough(cand, chosen, min_dist):
            chosen.append(cand)
    if len(chosen) < n:
        raise ValueError(f"Could not generate enough {what} positions.")
    return np.array(chosen)


def generate_positions(n_vehicles, min_distance=0.4, max_attempts=1000, seed=None):
    """(initial (N,2), final (N,2)).  seed=None draws from the global ``random`` module like the
    reference; an int seed uses a private ``random.Random(seed)`` (same stream as random.seed(seed))."""
    rng = _random if seed is None else _random.Random(seed)
    s = _Sampler(rng)
    initial = _rejection(s.on_circle, n_vehicles, min_distance, max_attempts, "initial")
    final = _rejection(s.goal, n_vehicles, min_distance, max_attempts, "final")
    return initial, final


def straight_line_min_distance(init, goal, idx_a=None, idx_b=None):
    """Closest approach of the straight-line, equal-time-profile motions p_i(s) = a_i + s (b_i - a_i),
    s in [0, 1]: the relative position is linear in s, so the minimum is a point-segment distance.
    Returns the (len(idx_a), len(idx_b)) matrix (all pairs by default; the diagonal is +inf)."""
    init = np.asarray(init, float)
    goal = np.asarray(goal, float)
    ia = np.arange(len(init)) if idx_a is None else np.asarray(idx_a)
    ib = np.arange(len(init)) if idx_b is None else np.asarray(idx_b)
    r0 = init[ia][:, None, :] - init[ib][None, :, :]
    dr = (goal[ia][:, None, :] - goal[ib][None, :, :]) - r0
    den = np.sum(dr * dr, axis=2)
    s = np.clip(-np.sum(r0 * dr, axis=2) / np.where(den > 0, den, 1.0), 0.0, 1.0)
    d = np.linalg.norm(r0 + s[..., None] * dr, axis=2)
    d[ia[:, None] == ib[None, :]] = np.inf
    return d


def generate_grid_swap(n_agents, seed=0, pitch=2.0, jitter=0.2, block=4, dim=2, layer_gap=2.0, min_sep=0.3,
                       max_tries=8192):
    """Synthetic scenario for large N.  Returns (initial (N,dim), final (N,dim), space_dims).

    dim=2: agents on a ceil(sqrt(N))^2 grid; dim=3: ceil(cbrt(N)) layers ``layer_gap`` apart, each a
    2-D grid.  Goals are the grid cells under a seeded permutation inside ``block`` x ``block`` cells
    (within a layer) with their own jitter, so the displacement is at most
    (block-1)*pitch*sqrt(2) + 2*jitter*sqrt(2) -- feasible for |v| <= 2 m/s, T = 10 s.

    A block's permutation is re-drawn (at most ``max_tries`` times) until the straight-line motions of its agents
    never come closer than ``min_sep``: an exact or near head-on swap (closest approach ~ 0) makes the FIRST
    linearised QP of the SCP infeasible -- the constraint normals flip sign within one time step -- for the
    reference algorithm as much as for this one.  A final pass re-draws blocks that conflict across block borders.
    space_dims = [min_0.., max_0..] with a 2 m rim."""
    rng = np.random.default_rng(seed)
    if dim == 2:
        layers, per = 1, n_agents
    elif dim == 3:
        layers = max(1, math.ceil(round(n_agents ** (1.0 / 3.0), 9)))
        per = math.ceil(n_agents / layers)
    else:
        raise ValueError("dim must be 2 or 3")
    side = math.ceil(math.sqrt(per))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    cells = np.stack([gx.ravel(), gy.ravel()], axis=1)

    def draw_block(cell, xy, idx):
        """goal positions for the agents idx of one block: candidates are drawn 256 at a time"""
        m = idx.size
        if m == 1:
            return cell[idx] * pitch + rng.uniform(-jitter, jitter, size=(1, 2))
        best, best_d = None, -1.0
        off = ~np.eye(m, dtype=bool)
        for _ in range(max(1, max_tries // 256)):
            T = 256
            perm = np.argsort(rng.random((T, m)), axis=1)
            g = cell[idx[perm]] * pitch + rng.uniform(-jitter, jitter, size=(T, m, 2))
            # closest approach of the straight-line motions of every pair, per coordinate (a reduction over a
            # length-2 axis costs numpy more than the arithmetic): r(s) = r0 + s dr, s in [0, 1]
            a = xy[idx]
            r0x = a[:, 0][:, None] - a[:, 0][None, :]
            r0y = a[:, 1][:, None] - a[:, 1][None, :]
            drx = (g[:, :, None, 0] - g[:, None, :, 0]) - r0x
            dry = (g[:, :, None, 1] - g[:, None, :, 1]) - r0y
            den = drx * drx + dry * dry
            sp = np.clip(-(r0x * drx + r0y * dry) / np.where(den > 0, den, 1.0), 0.0, 1.0)
            cx = r0x + sp * drx
            cy = r0y + sp * dry
            d = np.sqrt(cx * cx + cy * cy)
            dmin = np.where(off[None], d, np.inf).reshape(T, -1).min(axis=1)
            ok = np.nonzero(dmin >= min_sep)[0]
            pick = int(ok[0]) if ok.size else int(np.argmax(dmin))
            if dmin[pick] > best_d:
                best, best_d = g[pick], float(dmin[pick])
            if ok.size:
                break
        return best

    init = []
    goal = []
    remaining = n_agents
    for L in range(layers):
        cnt = min(per, remaining)
        remaining -= cnt
        cell = cells[:cnt]
        xy = cell * pitch + rng.uniform(-jitter, jitter, size=(cnt, 2))
        key = (cell[:, 0] // block) * (side // block + 1) + (cell[:, 1] // block)
        gxy = np.empty_like(xy)
        groups = [np.nonzero(key == kk)[0] for kk in np.unique(key)]
        for idx in groups:
            gxy[idx] = draw_block(cell, xy, idx)
        # conflicts across block borders: re-draw one of the two blocks, a few sweeps
        owner = np.empty(cnt, dtype=int)
        for b, idx in enumerate(groups):
            owner[idx] = b
        for _ in range(20):
            bad = set()
            step = 512
            for s0 in range(0, cnt, step):
                ia = np.arange(s0, min(cnt, s0 + step))
                d = straight_line_min_distance(xy, gxy, ia, np.arange(cnt))
                rr, cc = np.nonzero(d < min_sep)
                for r_, c_ in zip(ia[rr], cc):
                    if owner[r_] != owner[c_]:
                        bad.add(int(max(owner[r_], owner[c_])))
            if not bad:
                break
            for b in bad:
                gxy[groups[b]] = draw_block(cell, xy, groups[b])
        if dim == 3:
            z = np.full((cnt, 1), L * layer_gap)
            xy = np.hstack([xy, z])
            gxy = np.hstack([gxy, z])
        init.append(xy)
        goal.append(gxy)
    init = np.vstack(init)
    goal = np.vstack(goal)
    lo = np.minimum(init.min(axis=0), goal.min(axis=0)) - 2.0
    hi = np.maximum(init.max(axis=0), goal.max(axis=0)) + 2.0
    return init, goal, [float(v) for v in lo] + [float(v) for v in hi]
