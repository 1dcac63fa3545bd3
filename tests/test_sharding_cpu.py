"""Multi-GPU plumbing on CPU: world_size 2 over gloo (no GPU needed).

Each rank owns a pair range / an agent range (path_planning/_sharding.py); the oracle stands in for the HIP
kernels so that the test checks exactly what the N>1 path adds: the partitions tile the problem, the padded
allgather of compact rows reproduces the single-rank working set, and the allgather of per-shard trajectories
reproduces the full array."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, K, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import scp_oracle as so
    from path_planning._sharding import Shard
    from path_planning.scenarios.position_generator import generate_grid_swap

    p0, pf, space = generate_grid_swap(N, seed=3)
    prob = so.make_problem(N, K * 0.2 + 1e-9, 0.2, 0.8, space, p0, pf)
    rng = np.random.default_rng(0)
    acc = 0.2 * rng.standard_normal((N, K, 2))
    sh = Shard(N, rank, world)

    # agent-sharded kinematics + allgather of per-shard trajectories
    i0, i1 = sh.agent_range()
    sub = so.make_problem(i1 - i0, K * 0.2 + 1e-9, 0.2, 0.8, space, p0[i0:i1], pf[i0:i1])
    pos_local, _ = so.kinematics(sub, acc[i0:i1])
    pos = sh.allgather_positions(torch.from_numpy(pos_local)).numpy()
    pos_full, _ = so.kinematics(prob, acc)
    assert np.array_equal(pos, pos_full)

    # pair-range shard of the linearisation + allgather of the compact candidate rows
    eta, l, d = so.linearize_pairs(prob, pos)
    q0, q1 = sh.pair_range()
    pairs = prob.pairs
    sel = np.nonzero(d - prob.R < 0.6)[0]
    mine = sel[((sel % pairs) >= q0) & ((sel % pairs) < q1)]
    rows, w_eta, w_l = sh.allgather_rows(torch.from_numpy(mine.astype(np.int64)), torch.from_numpy(eta[mine]),
                                         torch.from_numpy(l[mine]))
    rows = rows.numpy()
    assert np.array_equal(np.sort(rows), sel)
    assert np.array_equal(w_eta.numpy(), eta[rows]) and np.array_equal(w_l.numpy(), l[rows])

    # the row-free exchange: ids only, merged in ascending order (the single-rank order) + the max of one float, ONE collective;
    # a rank whose list outgrows the agreed capacity makes every rank repeat it with a longer message
    ids, mx = sh.allgather_ids(torch.from_numpy(np.sort(mine).astype(np.int64)), extra=0.25 * (rank + 1))
    assert np.array_equal(ids.numpy(), sel) and mx == 0.25 * world
    sh._ids_cap = 2
    ids2, none = sh.allgather_ids(torch.from_numpy(np.sort(mine).astype(np.int64)))
    assert np.array_equal(ids2.numpy(), sel) and none is None and sh._ids_cap >= 2
    empty, mx0 = sh.allgather_ids(torch.zeros(0, dtype=torch.int64), extra=-1.0 - rank)
    assert empty.numel() == 0 and mx0 == -1.0

    # reductions and broadcast
    assert sh.all_min(float(rank + 1)) == 1.0 and sh.all_max(float(rank + 1)) == float(world)
    assert sh.all_min_int((1 << 64) - 1 if rank == 0 else 12345) == 12345
    t = torch.full((3,), float(rank))
    sh.broadcast(t)
    assert torch.all(t == 0)
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows)
    dist.destroy_process_group()


@pytest.mark.parametrize("N,K", [(12, 9), (9, 7)])
def test_two_rank_shards(tmp_path, N, K):
    port = 29500 + (os.getpid() * 7 + N) % 1000
    mp.spawn(_worker, args=(2, port, N, K, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rows_0.npy"), np.load(tmp_path / "rows_1.npy")
    assert np.array_equal(a, b)  # every rank holds the same working set, in the same order


def test_partitions_tile():
    sys.path.insert(0, os.path.join(ROOT, "ba-path-planning_amd"))
    from path_planning._sharding import Shard

    for N in (2, 3, 17, 64, 1024):
        for world in (1, 2, 3, 8):
            pr = [Shard(N, r, 1).__class__.pair_range(_S(N, r, world)) for r in range(world)]
            ar = [_S(N, r, world).agent_range() for r in range(world)]
            assert pr[0][0] == 0 and pr[-1][1] == N * (N - 1) // 2
            assert ar[0][0] == 0 and ar[-1][1] == N
            for (a0, a1), (b0, b1) in zip(pr[:-1], pr[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in pr]
            assert max(sizes) - min(sizes) <= 1  # balanced in pair space


class _S:
    """Shard geometry without a process group."""

    def __init__(self, N, rank, world):
        self.N, self.rank, self.world = N, rank, world
        self.pairs = N * (N - 1) // 2

    def pair_range(self, rank=None):
        r = self.rank if rank is None else rank
        return self.pairs * r // self.world, self.pairs * (r + 1) // self.world

    def agent_range(self, rank=None):
        r = self.rank if rank is None else rank
        return self.N * r // self.world, self.N * (r + 1) // self.world
